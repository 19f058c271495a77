"""C2 (64 x 1080p float32, undistort from maps + 5x5): the strip height of the shared-record loop
(knob strip_h; 0 = the library's choice) and the frame-group chunk, alternated in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import imgprocessor_amd as ia
from imgprocessor_amd import ops

ctx = ia.default_context(0)
h, w, B = 1080, 1920, 64
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2); g /= g.sum(); k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
dst = ctx.empty((B, h, w), np.float32)


def t(n=60):
    for _ in range(10): ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(n): ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    e1.record(); ctx.synchronize()
    return e0.elapsed_ms(e1) / n


for _ in range(300): ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
res = {}
for rnd in range(3):
    for sh in (0, 20, 24, 30, 36, 40, 45, 54, 60, 72, 90, 108, 120, 135, 180, 216, 270):
        old = ctx.set_tuning(strip_h=sh)
        res.setdefault(sh, []).append(t())
        ctx.set_tuning(**old)
for sh, v in res.items():
    print('strip_h %3d: %s  min %.4f' % (sh, '  '.join('%.4f' % x for x in v), min(v)), flush=True)
