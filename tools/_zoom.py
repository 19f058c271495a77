import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops
ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
g = np.exp(-0.5 * np.arange(-2, 3) ** 2); g /= g.sum(); k5 = np.outer(g, g)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32)); dst = ctx.empty((B, h, w), np.float32)
print([(len(b['ms']), min(b['ms'])) for b in ctx.placement_log])
def timed(fn, n=30, warm=100):
    for _ in range(warm): fn()
    ctx.synchronize(); e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(n): fn()
    e1.record(); ctx.synchronize(); return e0.elapsed_ms(e1) / n
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
from imgprocessor_amd.utils.geometry import getOptimalNewCameraMatrix
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
for name, dist in (('barrel k1=-0.12', [-0.12, 0.03, 1e-3, -5e-4, 0.0]), ('pincushion k1=+0.08', [0.08, 0.01, 1e-3, -5e-4, 0.0])):
    dist = np.array(dist)
    for alpha in (None, 0.0, 0.5, 1.0):
        newK = K if alpha is None else getOptimalNewCameraMatrix(K, dist, (w, h), alpha)[0]
        mx, my = ops.build_undistort_map(K, dist, newK, h, w, ctx=ctx, device=True)
        mxh = mx.get(); myh = my.get()
        outside = ((mxh < 0) | (mxh > w - 1) | (myh < 0) | (myh > h - 1)).mean()
        print('%-20s newK %-12s outside %.3f: %.4f ms' % (name, 'K' if alpha is None else 'alpha=%.1f' % alpha, outside,
              timed(lambda: ops.remap_conv2d(src, mx, my, k5, out=dst))), flush=True)
