"""Does the SIZE handed to hipMalloc change where / how a batch buffer is placed?  Each trial
allocates a fresh (source, result) pair with the size rounded up to a granule and times the
headline launch and a copy.  GPU box only.   python tools/placement_probe3.py"""
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
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
host = np.concatenate([one] * 4)
order = sys.argv[1] if len(sys.argv) > 1 else 'exact,1g,4g,exact,1g,4g'


def view(base):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype, v.nbytes = ctx, (B, h, w), np.dtype(np.float32), nb
    v.ptr = C.c_void_p(base.ptr.value)
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


def rounded(kind):
    gran = {'exact': 1, '2m': 2 << 20, '1g': 1 << 30, '4g': 4 << 30}[kind]
    return (nb + gran - 1) // gran * gran


dmx = dmy = None
keep = []
for i, kind in enumerate(order.split(',')):
    size = rounded(kind)
    sb, db = ctx.empty((size,), np.uint8), ctx.empty((size,), np.uint8)
    s, d = view(sb), view(db)
    s.set(host)
    if dmx is None:   # (maps after the first pair, as bench.py does)
        dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
        for _ in range(60):
            ops.remap_conv2d(s, dmx, dmy, k5, out=d)
    t = timeit(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d))
    tc = timeit(lambda: d.copy_from(s))
    print('trial %d %-5s (%.2f GiB each): fused %.4f ms  copy %.4f ms   src %#x dst %#x'
          % (i, kind, size / 2 ** 30, t, tc, sb.ptr.value, db.ptr.value), flush=True)
    keep.append((sb, db))
