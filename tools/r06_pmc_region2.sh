#!/bin/bash
# round 6, item 1b continued: the same probe with the DIMENSIONED counters (one row per TCC instance and XCC)
# - is the strip-shaped store stream spread evenly over the channels on a slow block?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
G1="TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_BUSY TCC_EA0_WRREQ_LEVEL"
G2="TCC_REQ TCC_TAG_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_CYCLE"
i=0
for g in "$G1" "$G2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $g --kernel-trace -d $R/gpurun_out/r06/region2/p$i -o pmc --output-format csv -- $R/tools/region_pmc.bin 24 4 > $R/gpurun_out/r06/region2.p$i.log 2>&1
  echo "pass $i rc $?"
  grep "^pass\|^fast" $R/gpurun_out/r06/region2.p$i.log
done
python3 $R/tools/pmc_dims.py $R/gpurun_out/r06/region2 > $R/gpurun_out/r06/region2.txt 2>&1
cat $R/gpurun_out/r06/region2.txt
