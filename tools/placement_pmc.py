"""Counters of the headline launch on a well placed and on a badly placed result buffer: run
under `rocprofv3 --pmc ... --kernel-trace`; the launches on the best candidate come first, then a
device copy as a marker, then the launches on the worst.  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
s0 = ctx.to_device(np.concatenate([one] * 4))
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)


def timeit(fn, n=8, warm=4):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


cands = [ctx.empty((B, h, w), np.float32) for _ in range(10)]
for _ in range(40):
    ops.remap_conv2d(s0, dmx, dmy, k5, out=cands[0])
t = [timeit(lambda: ops.remap_conv2d(s0, dmx, dmy, k5, out=d)) for d in cands]
bi, wi = int(np.argmin(t)), int(np.argmax(t))
print('candidates ms: %s; best %d, worst %d' % (' '.join('%.4f' % x for x in t), bi, wi), flush=True)
marker = ctx.empty((1024, 1024), np.float32)
m2 = ctx.empty((1024, 1024), np.float32)
m2.copy_from(marker)     # marker 1: the candidate search is over
ctx.synchronize()
for _ in range(12):
    ops.remap_conv2d(s0, dmx, dmy, k5, out=cands[bi])
ctx.synchronize()
m2.copy_from(marker)     # marker 2: best -> worst
ctx.synchronize()
for _ in range(12):
    ops.remap_conv2d(s0, dmx, dmy, k5, out=cands[wi])
ctx.synchronize()
