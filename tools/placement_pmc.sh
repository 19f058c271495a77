#!/bin/bash
# counters of the headline launch on the best / worst placed result buffer (tools/placement_pmc.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for g in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_BUSY_sum" \
         "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $g --kernel-trace -d $R/gpurun_out/plpmc/p$i -o pmc --output-format csv -- python3 $R/tools/placement_pmc.py > $R/gpurun_out/plpmc.p$i.log 2>&1 || echo "pass $i failed"
  grep "candidates ms" $R/gpurun_out/plpmc.p$i.log
done
python3 $R/tools/placement_pmc_sum.py $R/gpurun_out/plpmc
