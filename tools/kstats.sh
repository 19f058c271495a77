#!/bin/bash
# usage: tools/kstats.sh <outdir> -- python3 script args   -> per-kernel durations (rocprofv3 --kernel-trace --stats)
out="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/$out" -o ks --output-format csv -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$out.log" 2>&1
python3 - "$GRAFT_REPO_ROOT/gpurun_out/$out" <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        print('%-100s calls %5s avg %10.1f us  min %10.1f  max %10.1f' % (n[:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
