"""What footprints on the border of the source cost: the remap and remap + filter kernels on a lens map
that stays inside the source (newK = K) and on the reference's default - getOptimalNewCameraMatrix with
alpha = 1 (camera/LensDistortion.py:350-357), 2.8 % of the output pixels outside the source along the
rim.  4K frames, batches of 16 / 15 / 1.
"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from imgprocessor_amd.utils.geometry import getOptimalNewCameraMatrix
ctx = ia.default_context(0)
h, w = 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
maps = {'inside (newK = K)': ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True),
        'alpha = 1': ops.build_undistort_map(K, dist, getOptimalNewCameraMatrix(K, dist, (w, h), 1.0)[0], h, w, ctx=ctx, device=True)}
def timed(fn, n=20, warm=30):
    for _ in range(warm): fn()
    ctx.synchronize(); e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(n): fn()
    e1.record(); ctx.synchronize(); return e0.elapsed_ms(e1) / n
rng = np.random.default_rng(0)
def kern(k):
    a = rng.random((k, k)); return a / a.sum()
g9 = ops.gaussian_kernel1d(1.0)
for B in (16, 15, 1):
    src = ctx.to_device(rng.random((B, h, w), dtype=np.float32)); dst = ctx.empty((B, h, w), np.float32)
    u16 = ctx.to_device(rng.integers(0, 4096, (B, h, w)).astype(np.uint16))
    cases = {'remap linear': lambda mx, my: ops.remap(src, mx, my, 'linear', out=dst),
             'remap cubic': lambda mx, my: ops.remap(src, mx, my, 'cubic', out=dst),
             'remap lanczos4': lambda mx, my: ops.remap(src, mx, my, 'lanczos4', out=dst),
             'remap u16 linear': lambda mx, my: ops.remap(u16, mx, my, 'linear', out_dtype=np.float32, out=dst),
             'fused 3x3': lambda mx, my: ops.remap_conv2d(src, mx, my, kern(3), out=dst),
             'fused 5x5': lambda mx, my: ops.remap_conv2d(src, mx, my, kern(5), out=dst),
             'fused 7x7': lambda mx, my: ops.remap_conv2d(src, mx, my, kern(7), out=dst),
             'fused 11x11': lambda mx, my: ops.remap_conv2d(src, mx, my, kern(11), out=dst),
             'fused sep 9+9': lambda mx, my: ops.remap_sepconv2d(src, mx, my, g9, g9, out=dst),
             'fused u16 7x7': lambda mx, my: ops.remap_conv2d(u16, mx, my, kern(7), out=dst)}
    for name, fn in cases.items():
        r = [timed(lambda: fn(*maps[m])) for m in maps]
        print('n=%-2d %-18s inside %.4f   alpha=1 %.4f   (%+.0f %%)' % (B, name, r[0], r[1], 100 * (r[1] / r[0] - 1)), flush=True)
    del src, dst, u16
