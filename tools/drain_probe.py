"""Is the ~80 us a 64-frame headline launch pays over the marginal cost of 64 more frames the DRAIN of its last
round of workgroups?  2 T(64) - T(128) on the same 128 frames for strip heights 144 / 72 / 48 / 36: the drain
model says it scales with the duration of a workgroup = with the strip height.   python tools/drain_probe.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.device import DeviceArray  # noqa: E402

ctx = ia.default_context(0)
h, w = 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
N = 128
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
src = ctx.to_device(np.concatenate([one] * (N // 16)))
dst = ctx.empty((N, h, w), np.float32)


def window(a, first, n):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype, v.nbytes = ctx, (n, h, w), a.dtype, n * h * w * 4
    v.ptr = ctypes.c_void_p(a.ptr.value + first * h * w * 4)
    v._owner = False
    v._base = a
    return v


def t(fn, n=20):
    for _ in range(6):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


a0, b0 = window(src, 0, 64), window(dst, 0, 64)
a1, b1 = window(src, 64, 64), window(dst, 64, 64)
for _ in range(100):
    ops.remap_conv2d(a0, dmx, dmy, k5, out=b0)
print('strip_h   T(first 64)  T(second 64)   T(128)    2 x 64 - 128 (us)')
for rnd in range(2):
    for sh in (0, 144, 108, 72, 48, 36, 24):
        ctx.set_tuning(strip_h=sh)
        t0 = t(lambda: ops.remap_conv2d(a0, dmx, dmy, k5, out=b0))
        t1 = t(lambda: ops.remap_conv2d(a1, dmx, dmy, k5, out=b1))
        t2 = t(lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst), 12)
        print('%5d     %9.4f    %9.4f   %9.4f    %7.1f' % (sh, t0, t1, t2, (t0 + t1 - t2) * 1e3))
ctx.set_tuning(strip_h=0)
