#!/bin/bash
# round-3 evidence on the GPU box: bench line, kernel stats, PMC traffic (separate passes)
# (the profiled headline-only passes take the first allocation, --placements 1: every launch in their
#  statistics is then a launch on the buffers the run times)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r03_bench.json 2> $R/gpurun_out/r03_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_stats -o r03 --output-format csv -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r03_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_hstats -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --placements 1 > $R/gpurun_out/r03_hstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r03_fetch -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --placements 1 --steps 5 --warmup 2 > $R/gpurun_out/r03_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r03_write -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --placements 1 --steps 5 --warmup 2 > $R/gpurun_out/r03_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r03_cfetch -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r03_cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r03_cwrite -o r03 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r03_cwrite.log 2>&1
cat $R/gpurun_out/r03_bench.json | head -c 600
# the N > 1 launcher path on hardware: two ranks time-sharing this box's one GPU (no scaling claim)
cd $R && python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --steps 50 --warmup 5 > $R/gpurun_out/r03_bench_2ranks_one_gpu.json 2> $R/gpurun_out/r03_bench_2ranks.err
tail -c 400 $R/gpurun_out/r03_bench_2ranks_one_gpu.json
