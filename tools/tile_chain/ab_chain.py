"""A/B: perspective warp + separable 9+9 as two launches through the workspace (tile_chain = 0)
against the one launch of csrc/tile_chain.hpp (1), alternated in one process on one pair of buffers;
16 x 4K float32: the bench's quadrilateral (C3) and the picture rotated by 15 degrees, bilinear and
bicubic.  Knob sweeps: python tools/ab_chain.py [chain_steps=2,4,8] [chain_frames=4,8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from imgprocessor_amd.utils import getPerspectiveTransform

sweeps = {}
for a in sys.argv[1:]:
    k, v = a.split('=')
    sweeps[k] = [int(x) for x in v.split(',')]
ctx = ia.default_context(0)
h, w, B = 2160, 3840, 16
quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
a = np.deg2rad(15.0)
cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
              [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy], [0, 0, 1.0]])
Hr = np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]]) @ R
g9 = ops.gaussian_kernel1d(1.0)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
dst = ctx.empty((B, h, w), np.float32)


def t(fn, n=20):
    for _ in range(5): fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(n): fn()
    e1.record(); ctx.synchronize()
    return e0.elapsed_ms(e1) / n


variants = [('two launches', dict(tile_chain=0)), ('one launch', dict(tile_chain=2))]
for k, vals in sweeps.items():
    for v in vals:
        variants.append(('one launch %s=%d' % (k, v), {'tile_chain': 2, k: v}))
for _ in range(40): ops.warp_perspective_sepconv2d(src, Hm, (h, w), g9, g9, 'cubic', out=dst)
for name, interp, M in (('C3 bicubic', 'cubic', Hm), ('C3 bilinear', 'linear', Hm),
                        ('rotated 15 bilinear', 'linear', Hr), ('rotated 15 bicubic', 'cubic', Hr)):
    res, bits = {}, {}
    for rnd in range(3):
        for vn, knobs in variants:
            old = ctx.set_tuning(**knobs)
            try:
                res.setdefault(vn, []).append(t(lambda: ops.warp_perspective_sepconv2d(src, M, (h, w), g9, g9, interp, out=dst)))
                if rnd == 0:
                    bits[vn] = dst.frame(3).get()
            finally:
                ctx.set_tuning(**old)
    ref = bits[variants[0][0]]
    print(name)
    for vn, _ in variants:
        same = np.array_equal(bits[vn].view(np.uint32), ref.view(np.uint32))
        print('  %-28s %s   min %.4f ms   %s' % (vn, '  '.join('%.4f' % v for v in res[vn]), min(res[vn]),
                                                 'same bits' if same else 'DIFFERENT BITS'), flush=True)
