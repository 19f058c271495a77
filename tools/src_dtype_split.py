"""fused undistort (maps) + 5x5 / 7x7 on 64 x 4K frames by source dtype: how much of the kernel's time
belongs to the tap gathers (float32: 4 dword gathers per sample, uint16: 2, uint8: 2 ushort)"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from bench_micro import timeit  # noqa: E402

ctx = ia.default_context(0)
B, h, w = int(os.environ.get('FRAMES', 64)), 2160, 3840
rng = np.random.default_rng(0)
Kc = np.array([[3840., 0, 1919.5], [0, 3840., 1079.5], [0, 0, 1]])
dc = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
dst = ctx.empty((B, h, w), np.float32)
for K in (5, 7):
    g = np.exp(-0.5 * np.arange(-(K // 2), K // 2 + 1) ** 2.0)
    g /= g.sum()
    k2 = np.outer(g, g)
    for dt in (np.float32, np.uint16, np.uint8):
        a = rng.random((B, h, w), dtype=np.float32)
        src = ctx.to_device(a if dt == np.float32 else (a * (4095 if dt == np.uint16 else 255)).astype(dt))
        del a
        t = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k2, out=dst))
        print('%dx%d %-8s %8.1f us  (%6.1f Gpx/s)' % (K, K, np.dtype(dt).name, t, B * h * w / t / 1e3))
        del src
