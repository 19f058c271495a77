"""The 8 GiB structure (tools/placement_probe5.py) and the headline launch: source, result and
the map pair placed at chosen GiB offsets of one slab.  GPU box only."""
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
slab = ctx.empty((56 * GB,), np.uint8)
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
hx, hy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx)
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
host = np.concatenate([one] * 4)


def view(off, shape):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype = ctx, shape, np.dtype(np.float32)
    v.nbytes = int(np.prod(shape)) * 4
    v.ptr = C.c_void_p(slab.ptr.value + int(off))
    v._owner = False
    v._base = slab
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


def run(so, do, mo):
    s, d = view(so * GB, (B, h, w)), view(do * GB, (B, h, w))
    mx, my = view(mo * GB, (h, w)), view(mo * GB + (64 << 20), (h, w))
    s.set(host)
    mx.set(hx)
    my.set(hy)
    t = timeit(lambda: ops.remap_conv2d(s, mx, my, k5, out=d))
    tc = timeit(lambda: d.copy_from(s))
    print('src %4.1f  dst %4.1f  maps %4.1f GiB: fused %.4f ms  copy %.4f ms' % (so, do, mo, t, tc), flush=True)


for _ in range(2):
    run(0, 2.5, 5)        # everything in the first 8 GiB
    run(0, 8, 5)          # result in the next 8 GiB
    run(0, 8, 12)         # result and maps there
    run(0, 16, 5)         # result two regions on (same class as the source)
    run(0, 2.5, 12)       # only the maps in the other class
    run(8, 10.5, 13)      # everything in the second region
    run(8, 16, 13)
    run(3, 11, 6)
    run(4, 12, 20)
