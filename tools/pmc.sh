#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counters pass 1>" "<counters pass 2>" ... -- python3 script args
# one rocprofv3 --pmc pass per counter group (never combined with sys/runtime tracing)
out="$1"; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/$out/p$i" -o pmc --output-format csv -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$out.p$i.log" 2>&1
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_sum.py" "$GRAFT_REPO_ROOT/gpurun_out/$out"
