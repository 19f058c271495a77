#!/bin/bash
# round-6 evidence on the GPU box: bench line, kernel stats, PMC traffic (separate passes), the
# other_configs kernels, the --gpus 2 self-launch with its end-to-end leg.  Every pass runs bench.py as a
# caller of the library does: batch buffers as the driver hands them out.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/hstats -o r06 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs > $O/hstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o r06 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o r06 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/cstats -o r06 --output-format csv -- python3 $R/bench.py --no-cpu > $O/cstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/cfetch -o r06 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $O/cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/cwrite -o r06 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $O/cwrite.log 2>&1
head -c 600 $O/bench.json
# the N > 1 path on hardware: `python bench.py --gpus 2` starts its two ranks itself (they time-share
# this box's one GPU: launcher, rendezvous, barrier, per-rank arrays, the end-to-end leg - no scaling claim)
cd $R && python3 $R/bench.py --gpus 2 --steps 50 --warmup 5 --e2e-frames 32 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
tail -c 1500 $O/bench_2ranks_one_gpu.json
tail -5 $O/bench_2ranks.err
