#!/bin/bash
# round-2 evidence on the GPU box: bench line, kernel stats, PMC traffic (separate passes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r02_bench.json 2> $R/gpurun_out/r02_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_stats -o r02 --output-format csv -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_hstats -o r02 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs > $R/gpurun_out/r02_hstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r02_fetch -o r02 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r02_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r02_write -o r02 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r02_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r02_cfetch -o r02 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r02_cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r02_cwrite -o r02 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r02_cwrite.log 2>&1
cat $R/gpurun_out/r02_bench.json | head -c 600
