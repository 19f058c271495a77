"""ms per frame of the headline launch (maps + 5x5 outer product, 4K float32) against the frames per launch:
is the 5 % between 64 and 128 frames per launch a fixed cost per launch, or the quantisation of the launch's
workgroups into rounds of resident ones?   python tools/batch_sweep.py [knob=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

knobs = {k: int(v) for k, v in (a.split('=') for a in sys.argv[1:] if '=' in a and not a.startswith('--'))}
NS = (16, 32, 48, 56, 60, 64, 68, 72, 80, 96, 112, 128, 160, 192, 256)
for a in sys.argv[1:]:
    if a.startswith('--n='):
        NS = tuple(int(v) for v in a[4:].split(','))
ctx = ia.default_context(0)
ctx.set_tuning(**knobs)
h, w = 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
NMAX = 256
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
src = ctx.empty((NMAX, h, w), np.float32)
dst = ctx.empty((NMAX, h, w), np.float32)
host = np.concatenate([one] * (NMAX // 16))
src.set(host)
del host


def view(a, n):
    from imgprocessor_amd.device import DeviceArray
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype, v.nbytes = ctx, (n, h, w), a.dtype, n * h * w * 4
    v.ptr = a.ptr
    v._owner = False
    v._base = a
    return v


def t(fn, n):
    for _ in range(max(3, n // 3)):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


# settle the clocks
s64, d64 = view(src, 64), view(dst, 64)
for _ in range(100):
    ops.remap_conv2d(s64, dmx, dmy, k5, out=d64)
print('knobs', knobs)
print('frames   ms/launch   us/frame   groups  chunk')
for rnd in range(2):
    for n in NS:
        s, d = view(src, n), view(dst, n)
        ms = t(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d), max(8, 1200 // n))
        print('%5d   %9.4f   %8.3f   %5d  %5d' % (n, ms, ms / n * 1e3, n // 4, ctx.get_tuning('group_chunk_used')))


# the same launches on WINDOWS of the 256-frame buffers: where a window lies in the device memory moves a launch by
# several per cent - is the batch effect above a property of the launch or of the place?
def window(a, first, n):
    from imgprocessor_amd.device import DeviceArray
    import ctypes
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype, v.nbytes = ctx, (n, h, w), a.dtype, n * h * w * 4
    v.ptr = ctypes.c_void_p(a.ptr.value + first * h * w * 4)
    v._owner = False
    v._base = a
    return v


print('window (first frame, frames)   ms/launch   us/frame')
for n in (64, 128, 256):
    for first in range(0, NMAX, n):
        s, d = window(src, first, n), window(dst, first, n)
        ms = t(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d), max(8, 1200 // n))
        print('   %3d + %3d                  %9.4f   %8.3f' % (first, n, ms, ms / n * 1e3))
