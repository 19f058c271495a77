"""BASELINE C2 (64 x 1080p float32, LensDistortion maps + 5x5 Gaussian) under strip-height / tail knobs, one process.
    python tools/c2_knobs.py name:knob=v,knob=v ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

sets = []
for a in sys.argv[1:]:
    name, _, kv = a.partition(':')
    sets.append((name, {k: int(v) for k, v in (x.split('=') for x in kv.split(',') if x)}))
sets = sets or [('base', {})]
ctx = ia.default_context(0)
h, w, B = 1080, 1920, 64
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
g /= g.sum()
k5 = np.outer(g, g)
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
dst = ctx.empty((B, h, w), np.float32)


def t(n=60):
    for _ in range(20):
        ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


base = {k: ctx.get_tuning(k) for _, kn in sets for k in kn}
for _ in range(300):
    ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
res = {n: [] for n, _ in sets}
for rnd in range(3):
    for name, kn in sets:
        ctx.set_tuning(**base)
        ctx.set_tuning(**kn)
        res[name].append(t())
ctx.set_tuning(**base)
for name, kn in sets:
    print('%-10s %-40s %s  min %.4f' % (name, kn, '  '.join('%.4f' % v for v in res[name]), min(res[name])))
