#!/usr/bin/env python3
"""The standalone bilinear remap of uint16 frames into float32 on the marching strips (wave_sep_kernel with K = 1: no
filter, 256-px strips; knob strip_remap) against the gather kernels it replaces: bits and time, one process.  The left
column runs the entry point with strip_remap = 0, the right one the K = 1 chain directly.
    python tools/strip_remap_probe.py [batch ...]
(float32 frames were measured with a build that had their K = 1 kernels too - profiles/r06_micro.txt: level with the
tile kernel on maps, 15 - 19 % slower under a homography - and are not built.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.utils import getPerspectiveTransform  # noqa: E402

ctx = ia.default_context(0)
h, w = 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
one = np.array([1.0])
rng = np.random.default_rng(0)
base = rng.random((16, h, w), dtype=np.float32)


def t(fn, n):
    for _ in range(max(5, n // 3)):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


for batch in [int(a) for a in sys.argv[1:]] or [16, 64]:
    src = ctx.to_device(np.concatenate([base] * (batch // 16)) if batch >= 16 else base[:batch].copy())
    u16 = ctx.to_device((np.concatenate([base] * (batch // 16)) * 4095).astype(np.uint16) if batch >= 16
                        else (base[:batch] * 4095).astype(np.uint16))
    dst, dst2 = ctx.empty((batch, h, w), np.float32), ctx.empty((batch, h, w), np.float32)
    pairs = [
        ('maps u16 -> f32', lambda o: ops.remap(u16, dmx, dmy, out_dtype=np.float32, out=o),
         lambda o: ops.remap_sepconv2d(u16, dmx, dmy, one, one, out=o)),
        ('maps q5 u16 -> f32', lambda o: ops.remap(u16, dmx, dmy, 'linear_cv_q5', out_dtype=np.float32, out=o),
         lambda o: ops.remap_sepconv2d(u16, dmx, dmy, one, one, 'linear_cv_q5', out=o)),
        ('lens u16 -> f32', lambda o: ops.undistort(u16, K, dist, K, out_dtype=np.float32, out=o),
         lambda o: ops.undistort_sepconv2d(u16, K, dist, K, one, one, out=o)),
        ('homography u16 -> f32', lambda o: ops.warp_perspective(u16, Hm, (h, w), 'linear', out_dtype=np.float32, out=o),
         lambda o: ops.warp_perspective_sepconv2d(u16, Hm, (h, w), one, one, 'linear', out=o)),
    ]
    n = max(10, 1200 // batch)
    ctx.set_tuning(strip_remap=0)
    for _ in range(200 // max(1, batch // 16)):
        pairs[0][1](dst)
    print('batch %d x 4K, ms per launch: entry point / strips (K = 1), 3 rounds alternated' % batch)
    for name, a, b in pairs:
        a(dst)
        b(dst2)
        same = np.array_equal(dst.get().view(np.uint32), dst2.get().view(np.uint32))
        ra, rb = [], []
        for _ in range(3):
            ra.append(t(lambda: a(dst), n))
            rb.append(t(lambda: b(dst2), n))
        print('%-18s %s   |   %s   min %.4f / %.4f  (%+.1f %%)  %s' % (
            name, '  '.join('%.4f' % v for v in ra), '  '.join('%.4f' % v for v in rb), min(ra), min(rb),
            100 * (min(rb) / min(ra) - 1), 'identical bits' if same else 'BITS DIFFER'), flush=True)
    del src, u16, dst, dst2
