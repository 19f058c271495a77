"""What the host side can feed: the C4 chain (4K uint16 frame -> undistort + 7x7 -> float32 frame)
host -> host through page-locked buffers on ONE GPU, for 1..8 overlapped workers, the worker
threads unpinned, pinned to the GPU's NUMA node and pinned to another node; plus the raw copy
rates.  GPU box only.    python tools/c4_host_feed.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.sharding import FramePipeline, numa_cpus_of_device  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
h, w = 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
k7 = np.random.default_rng(123).random((7, 7))
k7 /= k7.sum()
ctx = ia.default_context(0)
print('host: %d CPUs; NUMA nodes: %s' % (os.cpu_count(), ', '.join(
    '%s=%s' % (n, open('/sys/devices/system/node/%s/cpulist' % n).read().strip())
    for n in sorted(os.listdir('/sys/devices/system/node')) if n.startswith('node'))))
local = numa_cpus_of_device(0)
print('GPU 0 NUMA-local CPUs: %s' % (sorted(local) if local else 'not reported'))
allc = set(range(os.cpu_count()))
other = (allc - local) if local and allc - local else None

# raw copy rates, one stream, page-locked
hin = ctx.pinned_empty((8, h, w), np.uint16)
hout = ctx.pinned_empty((8, h, w), np.float32)
hin[...] = 7
din, dout = ctx.empty((8, h, w), np.uint16), ctx.empty((8, h, w), np.float32)
for name, f, nb in (('H2D', lambda: din.set(hin), hin.nbytes), ('D2H', lambda: dout.get(hout), hout.nbytes)):
    f(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f()
    ctx.synchronize()
    print('%s page-locked, one stream: %.1f GB/s' % (name, 5 * nb / (time.perf_counter() - t0) / 1e9))
del hin, hout, din, dout

frame = np.round(np.random.default_rng(0).random((h, w)) * 4095).astype(np.uint16)
for label, cpus in (('unpinned', None), ('GPU-local node', local), ('other node(s)', other)):
    if label != 'unpinned' and not cpus:
        continue
    for depth in (1, 2, 3, 4, 6, 8):
        pipe = FramePipeline(0, depth, cpus=cpus)
        maps = {id(c): ops.build_undistort_map(K, dist, K, h, w, ctx=c, device=True)
                for c in pipe.contexts}
        if cpus:
            os.sched_setaffinity(0, cpus)   # the buffers are touched from the same node
        fin = pipe.pinned_empty((N, h, w), np.uint16)
        fout = pipe.pinned_empty((N, h, w), np.float32)
        fin[...] = frame
        os.sched_setaffinity(0, allc)

        def fn(c, d, o):
            mx, my = maps[id(c)]
            ops.remap_conv2d(d, mx, my, k7, out=o)
        pipe.run(fin[:depth], fout[:depth], fn)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            pipe.run(fin, fout, fn)
            best = min(best, time.perf_counter() - t0)
        print('%-16s %d workers: %.3f ms/frame = %.1f Gpix/s, PCIe %.1f GB/s (in + out)'
              % (label, depth, best / N * 1e3, N * h * w / best / 1e9, 6 * N * h * w / best / 1e9))
        del fin, fout, pipe, maps
