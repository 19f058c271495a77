"""Which buffer's placement matters?  4 source and 4 result allocations, the headline launch timed
for all 16 combinations (same maps).  GPU box only."""
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
N = 4
srcs, dsts = [], []
for i in range(N):   # interleaved, as separate calls allocate them
    srcs.append(ctx.to_device(host))
    dsts.append(ctx.empty((B, h, w), np.float32))
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)


def timeit(fn, n=20, warm=6):
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
    ops.remap_conv2d(srcs[0], dmx, dmy, k5, out=dsts[0])
print('rows: source allocation, columns: result allocation; fused ms (copy ms)')
for i in range(N):
    row = []
    for j in range(N):
        t = timeit(lambda: ops.remap_conv2d(srcs[i], dmx, dmy, k5, out=dsts[j]))
        tc = timeit(lambda: dsts[j].copy_from(srcs[i]))
        row.append('%.4f (%.3f)' % (t, tc))
    print('  src %d: %s' % (i, '   '.join(row)))
# a second map pair: does the maps' placement matter?
dmx2, dmy2 = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
print('second map pair, src 0..3 -> dst 0..3 diagonal: %s' % '  '.join(
    '%.4f' % timeit(lambda: ops.remap_conv2d(srcs[i], dmx2, dmy2, k5, out=dsts[i])) for i in range(N)))
