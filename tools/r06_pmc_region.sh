#!/bin/bash
# round 6, review item 1b: memory-side counters of the strip-shaped store stream on a fast and on a slow block
# (tools/region_pmc.hip), one rocprofv3 --pmc pass per counter group, never combined with a tracing domain
# other than --kernel-trace.  Output: gpurun_out/r06/region/ (+ region.txt = the table).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r06/counters_avail.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" $R/gpurun_out/r06/counters_avail.txt | sort -u > $R/gpurun_out/r06/counter_names.txt
G1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_BUBBLE_sum"
G2="TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
G3="TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum"
G4="TCC_REQ_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
G5="TCC_EA0_WRREQ[0] TCC_EA0_WRREQ[1] TCC_EA0_WRREQ[2] TCC_EA0_WRREQ[3] TCC_EA0_WRREQ[4] TCC_EA0_WRREQ[5] TCC_EA0_WRREQ[6] TCC_EA0_WRREQ[7] TCC_EA0_WRREQ[8] TCC_EA0_WRREQ[9] TCC_EA0_WRREQ[10] TCC_EA0_WRREQ[11] TCC_EA0_WRREQ[12] TCC_EA0_WRREQ[13] TCC_EA0_WRREQ[14] TCC_EA0_WRREQ[15]"
G6="TCC_EA0_WRREQ_STALL[0] TCC_EA0_WRREQ_STALL[1] TCC_EA0_WRREQ_STALL[2] TCC_EA0_WRREQ_STALL[3] TCC_EA0_WRREQ_STALL[4] TCC_EA0_WRREQ_STALL[5] TCC_EA0_WRREQ_STALL[6] TCC_EA0_WRREQ_STALL[7] TCC_EA0_WRREQ_STALL[8] TCC_EA0_WRREQ_STALL[9] TCC_EA0_WRREQ_STALL[10] TCC_EA0_WRREQ_STALL[11] TCC_EA0_WRREQ_STALL[12] TCC_EA0_WRREQ_STALL[13] TCC_EA0_WRREQ_STALL[14] TCC_EA0_WRREQ_STALL[15]"
G7="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"
G8="GRBM_GUI_ACTIVE TCC_CYCLE_sum TCC_EA0_WRREQ_PROBE_COMMAND_sum TCC_EA0_ATOMIC_sum"
i=0
for g in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7" "$G8"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $g --kernel-trace -d $R/gpurun_out/r06/region/p$i -o pmc --output-format csv -- $R/tools/region_pmc.bin 24 4 > $R/gpurun_out/r06/region.p$i.log 2>&1
  echo "pass $i rc $?" >> $R/gpurun_out/r06/region.rc
  tail -3 $R/gpurun_out/r06/region.p$i.log
done
python3 $R/tools/pmc_sum.py $R/gpurun_out/r06/region > $R/gpurun_out/r06/region.txt 2>&1
cat $R/gpurun_out/r06/region.txt
