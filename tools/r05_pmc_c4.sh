#!/bin/bash
# round 5, review item 7: instruction counters of the C4 kernel (uint16 -> float32 undistort + dense 7x7, 64 x 4K)
# next to the float32 source + 7x7, the headline (float32 + 5x5) and the plain float32 7x7
mkdir -p gpurun_out/r05
export IMGPROC_HIP_PLACE=1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS"
G3="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"
for c in c4 fused7 fused conv7; do
  a=""; [ $c != fused ] && a="--case $c"
  bash tools/pmc.sh r05/c4_$c "$G1" "$G2" "$G3" -- python3 $GRAFT_REPO_ROOT/tools/run_one.py --batch 64 --steps 3 $a > gpurun_out/r05/c4_$c.txt 2>&1
done
for c in c4 fused7 fused conv7; do echo "=== $c"; grep -v "^$" gpurun_out/r05/c4_$c.txt | grep -A22 "wave_stencil\|wave_sep" | grep -v build_map | head -60; done > gpurun_out/r05/c4_counters.txt
cat gpurun_out/r05/c4_counters.txt | head -120
