"""Do the virtual addresses of 'good' and 'bad' allocations differ in some bit?  12 source and 12
result candidates (separate hipMallocs), each timed against a fixed partner.  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
host = np.concatenate([one] * 4)
s0 = ctx.to_device(host)
d0 = ctx.empty((B, h, w), np.float32)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)


def timeit(fn, n=16, warm=5):
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


for _ in range(60):
    ops.remap_conv2d(s0, dmx, dmy, k5, out=d0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
print('maps at %#x %#x; fixed src %#x, fixed dst %#x' % (dmx.ptr.value, dmy.ptr.value, s0.ptr.value, d0.ptr.value))
held = []
for i in range(N):
    s = ctx.empty((B, h, w), np.float32)
    s.copy_from(s0)
    t = timeit(lambda: ops.remap_conv2d(s, dmx, dmy, k5, out=d0))
    a = s.ptr.value
    print('src %2d at %#x  (GiB %7.3f, mod 16 GiB %6.3f, MiB mod 2048 = %4d): %.4f ms' % (i, a, a / 2 ** 30, (a % (16 << 30)) / 2 ** 30, (a >> 20) % 2048, t), flush=True)
    held.append(s)
del held
ctx.trim()
held = []
for i in range(N):
    d = ctx.empty((B, h, w), np.float32)
    t = timeit(lambda: ops.remap_conv2d(s0, dmx, dmy, k5, out=d))
    a = d.ptr.value
    print('dst %2d at %#x  (GiB %7.3f, mod 16 GiB %6.3f, MiB mod 2048 = %4d): %.4f ms' % (i, a, a / 2 ** 30, (a % (16 << 30)) / 2 ** 30, (a >> 20) % 2048, t), flush=True)
    held.append(d)
