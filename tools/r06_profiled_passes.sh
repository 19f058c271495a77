#!/bin/bash
# N fresh processes of `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-configs`: every process draws its own
# pair of batch buffers (their class is reported in the line: config.probe_src_ms / probe_dst_ms), the bench line and the kernel
# statistics of ONE process describe the same launches.  Output: gpurun_out/r06p/pass<i>/ + pass<i>.log (the process's own line).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
N=${1:-6}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  rocprofv3 --kernel-trace --stats -d $O/pass$i -o r06 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs > $O/pass$i.log 2>&1
  python3 - <<PY
import json
ls=[l for l in open("$O/pass$i.log").read().splitlines() if l.startswith("{")]
l=json.loads(ls[-1])
print("pass $i: ms_per_step %.4f frac %.4f probe %.3f / %.3f" % (l["ms_per_step"], l["roofline"]["frac"], l["config"]["probe_src_ms"], l["config"]["probe_dst_ms"]))
PY
done
