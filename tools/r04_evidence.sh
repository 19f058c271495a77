#!/bin/bash
# round-4 evidence on the GPU box: bench line, kernel stats, PMC traffic (separate passes).
# Every pass runs bench.py as a caller of the library does: the context's block pool places the batch
# buffers (device.py::_alloc_placed), so the statistics of the headline kernel are launches on
# placed buffers; the pool's probe launches show up as wave_stencil_kernel<LoadRowSrc, 3>.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r04_bench.json 2> $R/gpurun_out/r04_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_stats -o r04 --output-format csv -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r04_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_hstats -o r04 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs > $R/gpurun_out/r04_hstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r04_fetch -o r04 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r04_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r04_write -o r04 --output-format csv -- python3 $R/bench.py --no-cpu --no-configs --steps 5 --warmup 2 > $R/gpurun_out/r04_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r04_cfetch -o r04 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r04_cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r04_cwrite -o r04 --output-format csv -- python3 $R/bench.py --no-cpu --steps 5 --warmup 2 > $R/gpurun_out/r04_cwrite.log 2>&1
python3 $R/tools/angle_sweep.py 0 0.5 1 2 4 7 15 30 45 90 > $R/gpurun_out/r04_angles.txt 2>&1
head -c 600 $R/gpurun_out/r04_bench.json
# the N > 1 launcher path on hardware: two ranks time-sharing this box's one GPU (no scaling claim)
cd $R && python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --steps 50 --warmup 5 > $R/gpurun_out/r04_bench_2ranks_one_gpu.json 2> $R/gpurun_out/r04_bench_2ranks.err
tail -c 400 $R/gpurun_out/r04_bench_2ranks_one_gpu.json
