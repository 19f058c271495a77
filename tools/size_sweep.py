"""time per launch of the main calls over odd frame sizes and batch sizes: looks for dispatch cliffs
(a size or batch that falls off a fast path).  GPU box only.  python tools/size_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
g9 = np.exp(-0.5 * np.arange(-4, 5) ** 2)
g9 /= g9.sum()


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3


rng = np.random.default_rng(0)
print('%-22s %9s %9s %9s %9s %9s   us per launch (ns per pixel)' % ('frames x h x w', 'fused5', 'conv5', 'sep9', 'remap', 'fusedsep9'))
for (n, h, w) in ((16, 2160, 3840), (16, 2160, 3838), (16, 2160, 3841), (16, 2161, 3840), (16, 2159, 3836),
                  (16, 1080, 1920), (16, 1080, 1922), (16, 1079, 1919), (64, 540, 960), (64, 541, 963),
                  (16, 2160, 4096), (16, 2160, 256), (16, 2160, 250), (16, 100, 3840), (256, 100, 100), (1, 4320, 7680)):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    src = ctx.to_device(rng.random((n, h, w), dtype=np.float32))
    dst = ctx.empty((n, h, w), np.float32)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    t = [timeit(lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst)),
         timeit(lambda: ops.conv2d(src, k5, out=dst)),
         timeit(lambda: ops.sepconv2d(src, g9, g9, out=dst)),
         timeit(lambda: ops.remap(src, dmx, dmy, out=dst)),
         timeit(lambda: ops.remap_sepconv2d(src, dmx, dmy, g9, g9, out=dst))]
    px = n * h * w
    print('%-22s %s' % ('%d x %d x %d' % (n, h, w), ' '.join('%9.1f' % x for x in t)) + '   (' +
          ' '.join('%.4f' % (x * 1e3 / px) for x in t) + ')', flush=True)
    del src, dst, dmx, dmy
