#!/bin/bash
# round-5 evidence on the GPU box: bench line, kernel stats, PMC traffic (separate passes), the
# other_configs kernels, the --gpus 2 self-launch.  Every pass runs bench.py as a caller of the library
# does: batch buffers as the driver hands them out (block placement is opt-in since round 5).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r05e/bench.json 2> $R/gpurun_out/r05e/bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05e/hstats -o r05 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs > $R/gpurun_out/r05e/hstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r05e/fetch -o r05 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r05e/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r05e/write -o r05 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r05e/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05e/cstats -o r05 --output-format csv -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r05e/cstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r05e/cfetch -o r05 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r05e/cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r05e/cwrite -o r05 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r05e/cwrite.log 2>&1
head -c 400 $R/gpurun_out/r05e/bench.json
# the N > 1 path on hardware: `python bench.py --gpus 2` starts its two ranks itself (they time-share
# this box's one GPU: the launcher, rendezvous, barrier and max-over-ranks timing - no scaling claim)
cd $R && python3 $R/bench.py --gpus 2 --steps 50 --warmup 5 > $R/gpurun_out/r05e/bench_2ranks_one_gpu.json 2> $R/gpurun_out/r05e/bench_2ranks.err
tail -c 300 $R/gpurun_out/r05e/bench_2ranks_one_gpu.json
