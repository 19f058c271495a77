#!/bin/bash
# round-3 evidence on the GPU box: bench line, kernel stats, PMC traffic (separate passes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r03_bench.json 2> $R/gpurun_out/r03_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_stats -o r03 --output-format csv -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r03_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_hstats -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs > $R/gpurun_out/r03_hstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r03_fetch -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r03_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r03_write -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r03_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r03_cfetch -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r03_cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r03_cwrite -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r03_cwrite.log 2>&1
cat $R/gpurun_out/r03_bench.json | head -c 600
