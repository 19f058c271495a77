"""C3 split: the fused warp + separable 9+9 with the homography evaluated per pixel and frame against the
same chain reading the coordinates from a float32 map pair (timing only: float32 maps round the
double coordinates), and against the 5x5 / 9x9 dense forms; 16 x 4K float32"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.utils import getPerspectiveTransform  # noqa: E402
from bench_micro import timeit  # noqa: E402

ctx = ia.default_context(0)
B, h, w = int(os.environ.get('FRAMES', 16)), 2160, 3840
rng = np.random.default_rng(0)
src = ctx.to_device(rng.random((B, h, w), dtype=np.float32))
dst = ctx.empty((B, h, w), np.float32)
quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
den = Hm[2, 0] * xs + Hm[2, 1] * ys + Hm[2, 2]
mx = ((Hm[0, 0] * xs + Hm[0, 1] * ys + Hm[0, 2]) / den).astype(np.float32)
my = ((Hm[1, 0] * xs + Hm[1, 1] * ys + Hm[1, 2]) / den).astype(np.float32)
dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
g = np.exp(-0.5 * (np.arange(-4, 5) / 2.0) ** 2)
g /= g.sum()
g5 = np.exp(-0.5 * np.arange(-2, 3) ** 2.0)
g5 /= g5.sum()
px = B * h * w
for name, fn in (
        ('warp + sep 9+9 (homography per pixel)', lambda: ops.warp_perspective_sepconv2d(src, Hm, (h, w), g, g, out=dst)),
        ('remap(maps) + sep 9+9', lambda: ops.remap_sepconv2d(src, dmx, dmy, g, g, out=dst)),
        ('warp + dense 5x5', lambda: ops.warp_perspective_conv2d(src, Hm, (h, w), np.outer(g5, g5), out=dst)),
        ('remap(maps) + dense 5x5', lambda: ops.remap_conv2d(src, dmx, dmy, np.outer(g5, g5), out=dst)),
        ('warp alone', lambda: ops.warp_perspective(src, Hm, (h, w), 'linear', out=dst)),
        ('remap(maps) alone', lambda: ops.remap(src, dmx, dmy, 'linear', out=dst)),
        ('sep 9+9 alone', lambda: ops.sepconv2d(src, g, g, out=dst)),
):
    t = timeit(ctx, fn)
    print('%-42s %8.1f us  (%6.1f Gpx/s)' % (name, t, px / t / 1e3))
