#!/bin/bash
# round 6: the same counter passes as tools/r06_pmc.sh on the kernels it did not cover - uint16 / uint8 Lanczos4 warps,
# LensDistortion.correct's own remap (bilinear, Lanczos4 from the map pair), C5 (bicubic warp + 11 x 11), the plain 11 x 11
tag=${1:-m}
mkdir -p gpurun_out/r06
export IMGPROC_HIP_PLACE=1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS"
G3="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE"
for c in lz4q lz16q lz8q remaplin remaplz4 c5 conv11; do
  bash tools/pmc.sh r06/pmc_${tag}_$c "$G1" "$G2" "$G3" -- python3 $GRAFT_REPO_ROOT/tools/run_one.py --batch 16 --steps 3 --case $c > gpurun_out/r06/pmc_${tag}_$c.txt 2>&1
done
