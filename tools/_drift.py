import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops
ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
g = np.exp(-0.5 * np.arange(-2, 3) ** 2); g /= g.sum(); k5 = np.outer(g, g)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32)); dst = ctx.empty((B, h, w), np.float32)
print(ctx.placement_log)
def timed(fn, n=40, warm=150):
    for _ in range(warm): fn()
    ctx.synchronize(); e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(n): fn()
    e1.record(); ctx.synchronize(); return e0.elapsed_ms(e1) / n
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
cases = {}
for name, dist in (('lens k1=-0.12 (bench)', [-0.12, 0.03, 1e-3, -5e-4, 0.0]), ('lens k1=-0.03', [-0.03, 0, 0, 0, 0.0]), ('lens k1=0 (identity)', [0, 0, 0, 0, 0.0])):
    cases[name] = ops.build_undistort_map(K, np.array(dist), K, h, w, ctx=ctx, device=True)
cases['shift 3.3 px, 2.7 rows'] = (ctx.to_device(xx + 3.3), ctx.to_device(yy + 2.7))
cases['zoom 0.97 about the centre'] = (ctx.to_device((xx - w / 2) * 0.97 + w / 2), ctx.to_device((yy - h / 2) * 0.97 + h / 2))
for deg in (0.25, 0.5, 1.0, 2.0):
    a = np.deg2rad(deg); cx, cy = w / 2, h / 2
    cases['rotation %.2f deg' % deg] = (ctx.to_device((np.cos(a) * (xx - cx) - np.sin(a) * (yy - cy) + cx).astype(np.float32)),
                                        ctx.to_device((np.sin(a) * (xx - cx) + np.cos(a) * (yy - cy) + cy).astype(np.float32)))
for rnd in range(2):
    for name, (mx, my) in cases.items():
        print('%-28s %.4f ms' % (name, timed(lambda: ops.remap_conv2d(src, mx, my, k5, out=dst))), flush=True)
