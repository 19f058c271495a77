import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import imgprocessor_amd as ia
from imgprocessor_amd import ops
ctx = ia.default_context(0)
for (h, w, B) in ((1080, 1920, 64), (1080, 1920, 16), (2160, 3840, 32), (4320, 7680, 8)):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2); g /= g.sum(); k5 = np.outer(g, g)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
    dst = ctx.empty((B, h, w), np.float32)
    def t(n=30):
        for _ in range(10): ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event(); e0.record()
        for _ in range(n): ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
        e1.record(); ctx.synchronize()
        return e0.elapsed_ms(e1) / n
    for _ in range(100): ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    res = {}
    for rnd in range(2):
        for gc in (0, -1, 1, 2, 4, 8):
            ctx.set_tuning(group_chunk=gc)
            res.setdefault(gc, []).append(t())
    ctx.set_tuning(group_chunk=-1)
    print('%dx%d x %d frames: ' % (w, h, B) + '  '.join('gc %d: %.4f' % (gc, min(v)) for gc, v in res.items()), flush=True)
    del src, dst, dmx, dmy
