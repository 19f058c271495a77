#!/bin/bash
# compile fused_k5.hip with the in-tree flags (+ extra defines) and print registers / checker / step mix
cd /root/repo/imgprocessor_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c fused_k5.hip -o /tmp/fused_k5.o -save-temps=obj 2>&1 | grep -v "^$" | head -30
S=/tmp/fused_k5-hip-amdgcn-amd-amdhsa-gfx950.s
python3 /root/repo/tools/regs.py $S "SampleRowSrcIfLi1ENS_8MapCoord"
python3 /root/repo/tools/regs.py $S "LoadRowSrc"
python3 /root/repo/tools/check_pipe_asm.py $S | cut -c1-150
python3 /root/repo/tools/pipe_isa.py $S SampleRowSrcIfLi1ENS_8MapCoordEEELi5 9 | head -3 | cut -c1-400
grep -c "scratch_" $S
