"""Where the result batch lies relative to the source batch: the headline launch (64 x 4K,
undistort + 5x5) timed with both batches inside ONE allocation at varying distances, and on
separately allocated buffers (several allocations).  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.device import DeviceArray  # noqa: E402
import ctypes as C  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
host = np.concatenate([one] * 4)
nb = host.nbytes


def view(base, off, shape, dtype):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype = ctx, shape, np.dtype(dtype)
    v.nbytes = int(np.prod(shape)) * v.dtype.itemsize
    v.ptr = C.c_void_p(base.ptr.value + off)
    v._owner = False
    v._base = base
    return v


def timeit(fn, n=25, warm=8):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


print('separate allocations (src, dst allocated in turn):')
keep = []
for i in range(4):
    s = ctx.to_device(host)
    d = ctx.empty((B, h, w), np.float32)
    t = timeit(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d))
    tc = timeit(lambda: d.copy_from(s))
    print('  allocation %d: src %#x dst %#x  fused %.4f ms  copy %.4f ms' % (i, s.ptr.value, d.ptr.value, t, tc))
    keep.append((s, d))   # kept alive so that the next pair lands elsewhere
del keep
big = ctx.empty((2 * nb + (1 << 30),), np.uint8)
src = view(big, 0, (B, h, w), np.float32)
src.set(host)
print('one allocation at %#x, dst at src + batch + delta:' % big.ptr.value)
for delta in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 16 << 20, 64 << 20, (64 << 20) + 8192,
              256 << 20, 512 << 20, 1000 << 20):
    dst = view(big, nb + delta, (B, h, w), np.float32)
    t = timeit(lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst))
    tc = timeit(lambda: dst.copy_from(src))
    print('  delta %10d: fused %.4f ms  copy %.4f ms' % (delta, t, tc))
