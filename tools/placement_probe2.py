"""Is it the ORDER of an allocation or its POSITION in a big early slab?  One 40 GB slab allocated
first thing in the process, the headline pair placed at several offsets inside it; then separate
allocations afterwards.  GPU box only."""
import os
import sys
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.device import DeviceArray  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
nb = B * h * w * 4
GB = 1 << 30
slab_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
big = ctx.empty((slab_gb * GB,), np.uint8)          # the FIRST allocation of the process
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
host = np.concatenate([one] * 4)


def view(off):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype = ctx, (B, h, w), np.dtype(np.float32)
    v.nbytes = nb
    v.ptr = C.c_void_p(big.ptr.value + off)
    v._owner = False
    v._base = big
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


s0 = view(0)
s0.set(host)
d0 = view(nb + (2 << 20))
for _ in range(60):
    ops.remap_conv2d(s0, dmx, dmy, k5, out=d0)     # clocks up
print('slab of %d GB at %#x (first allocation of the process); pair at offset (GB):' % (slab_gb, big.ptr.value))
step = 2 * nb + (4 << 20)
off = 0
while off + step <= slab_gb * GB:
    s, d = view(off), view(off + nb + (2 << 20))
    s.copy_from(s0) if off else None
    t = timeit(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d))
    tc = timeit(lambda: d.copy_from(s))
    print('  %6.2f: fused %.4f ms  copy %.4f ms' % (off / GB, t, tc))
    off += step
print('separate allocations afterwards:')
keep = []
for i in range(3):
    s = ctx.to_device(host)
    d = ctx.empty((B, h, w), np.float32)
    t = timeit(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d))
    tc = timeit(lambda: d.copy_from(s))
    print('  allocation %d: fused %.4f ms  copy %.4f ms' % (i, t, tc))
    keep.append((s, d))
