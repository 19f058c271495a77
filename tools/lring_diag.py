"""clean strips / strips of the LDS-ring plan for the 4K headline and a few other sources (GPU box)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h, w = 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
src = ctx.to_device(np.random.default_rng(0).random((batch, h, w), dtype=np.float32))
dst = ctx.empty((batch, h, w), np.float32)
ctx.set_tuning(lring=2)
for sh in (0, 72, 36):
    ctx.set_tuning(strip_h=sh)
    ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    ctx.synchronize()
    print('headline maps, strip_h %3d: clean %d of %d strips' % (sh, ctx.get_tuning('lring_clean'), ctx.get_tuning('lring_strips')))
ctx.set_tuning(strip_h=0)
