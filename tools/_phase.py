import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops, _lib as L
from imgprocessor_amd.device import dtype_id
ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
g = np.exp(-0.5 * np.arange(-2, 3) ** 2); g /= g.sum(); k5 = np.outer(g, g)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
dst = ctx.empty((B + 1, h, w), np.float32)     # room for offsets
print(ctx.placement_log)
kv = L.dbl(np.ravel(k5), 25)
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
maps = {'identity': (ctx.to_device(xx), ctx.to_device(yy)),
        'lens (bench)': ops.build_undistort_map(K, np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0]), K, h, w, ctx=ctx, device=True),
        'zoom 0.97': (ctx.to_device((xx - w / 2) * 0.97 + w / 2), ctx.to_device((yy - h / 2) * 0.97 + h / 2))}
def run(mx, my, off_bytes, fstride=None):
    fs = h * w if fstride is None else fstride
    def fn():
        ctx._check(ctx._lib.ipa_remap_conv2d_dev(ctx.handle, src.ptr, dtype_id(np.float32), h, w, w, mx.ptr, my.ptr, w, kv, 5, 5,
                   C.c_void_p(dst.ptr.value + off_bytes), dtype_id(np.float32), h, w, w, B, h * w, fs,
                   ops.interp_id('linear'), ops.border_id('constant'), 0.0, ops.border_id('reflect'), ops.border_id('reflect')), 'x')
    for _ in range(100): fn()
    ctx.synchronize(); e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(30): fn()
    e1.record(); ctx.synchronize(); return e0.elapsed_ms(e1) / 30
row = w * 4
for name, (mx, my) in maps.items():
    res = []
    for off in (0, 256, 4096, 8 * row, 37 * row, 64 * row, 137 * row, 500 * row, 1080 * row, 2000 * row):
        res.append('%d:%.3f' % (off // row if off >= row else off, run(mx, my, off)))
    print('%-14s dst offset (rows; first entries bytes): %s' % (name, '  '.join(res)), flush=True)
