#!/bin/bash
# round 6: instruction / LDS counters of the kernels the review names - C4 (uint16 + dense 7x7: did the
# 16-byte window reads remove the bank conflicts?), the headline on its separable route, the Lanczos4 and
# bicubic tile kernels at bench.py's C3 homography (what bounds them: the issue fraction), C3 bilinear.
# usage: bash tools/r06_pmc.sh [tag]   -> gpurun_out/r06/pmc_<tag>_<case>.txt
tag=${1:-a}
mkdir -p gpurun_out/r06
export IMGPROC_HIP_PLACE=1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS"
G3="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE"
for spec in "c4:64" "fused:64" "lz4q:16" "cubicq:16" "c3lin:16" "c3cubic:16"; do
  c=${spec%%:*}; b=${spec##*:}
  a=""; [ $c != fused ] && a="--case $c"
  bash tools/pmc.sh r06/pmc_${tag}_$c "$G1" "$G2" "$G3" -- python3 $GRAFT_REPO_ROOT/tools/run_one.py --batch $b --steps 3 $a > gpurun_out/r06/pmc_${tag}_$c.txt 2>&1
done
for c in c4 fused lz4q cubicq c3lin c3cubic; do echo "=== $c"; grep -v "^$" gpurun_out/r06/pmc_${tag}_$c.txt | grep -A24 "wave_stencil\|wave_sep\|tile_warp" | grep -v build_map | head -90; done > gpurun_out/r06/pmc_${tag}_counters.txt
head -150 gpurun_out/r06/pmc_${tag}_counters.txt
