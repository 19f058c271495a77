#!/bin/bash
# round-3 counter passes (rocprofv3 --pmc, one group per pass, kernel-trace only):
#   tools/r03_pmc.sh <outdir under gpurun_out> -- <program> <args>
out="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum"
 "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_IFETCH_LEVEL"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/$out/p$i" -o pmc --output-format csv -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$out.p$i.log" 2>&1 || echo "pass $i failed: $g"
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_sum.py" "$GRAFT_REPO_ROOT/gpurun_out/$out"
