"""Does the MAP pair's placement matter?  Fixed source and result, 10 map pairs allocated in
turn (with 1 GiB spacers between them), the headline launch timed with each.  GPU box only."""
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
d0 = ctx.empty((B, h, w), np.float32)
hx, hy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx)


def timeit(fn, n=16, warm=5):
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


held = []
mx, my = ctx.to_device(hx), ctx.to_device(hy)
for _ in range(60):
    ops.remap_conv2d(s0, mx, my, k5, out=d0)
for i in range(10):
    t = timeit(lambda: ops.remap_conv2d(s0, mx, my, k5, out=d0))
    print('map pair %d at %#x / %#x: %.4f ms' % (i, mx.ptr.value, my.ptr.value, t), flush=True)
    held += [mx, my, ctx.empty((1 << 30,), np.uint8)]
    mx, my = ctx.to_device(hx), ctx.to_device(hy)
