"""A/B timing of the fused 4K headline on the per-frame kernel with the waves of a workgroup on four
strips of a frame / on four frames of a strip (knob frames_wg); AB_SPLIT=1 adds the sampler + filter
wave pair, AB_RING=1 the ring kernel, AB_GROUP=1 the frame-group kernel.  Alternated inside one
process.  GPU box only.

    python tools/ab_group.py [batch ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def timeit(ctx, fn, n, warm):
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


def main():
    batches = [int(a) for a in sys.argv[1:]] or [16, 64]
    ctx = ia.default_context(0)
    h, w = 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    rng = np.random.default_rng(0)
    variants = [('strips in WG', dict(group=0, ring=0, pair=0, frames_wg=0)),
                ('frames in WG', dict(group=0, ring=0, pair=0, frames_wg=1))]
    if os.environ.get('AB_SPLIT'):
        variants += [('sampler+filter', dict(group=0, ring=0, pair=2, frames_wg=0))]
    if os.environ.get('AB_RING'):
        variants += [('ring+per-frame', dict(group=0, ring=1, pair=0))]
    if os.environ.get('AB_GROUP'):
        variants += [('group/gather', dict(group=1, ring=0, group_ring=0)),
                     ('group/ring', dict(group=1, ring=0, group_ring=1))]
    for B in batches:
        src = ctx.to_device(rng.random((B, h, w), dtype=np.float32))
        dst = ctx.empty((B, h, w), np.float32)
        n = max(10, 1600 // B)
        for rep in range(4):
            for name, knobs in variants:
                ctx.set_tuning(**knobs)
                t = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst), n, n // 4)
                ta = timeit(ctx, lambda: ops.undistort_conv2d(src, K, dist, K, k5, out=dst), n,
                            n // 4)
                comp = (8 * B + 8) * h * w
                print('B=%3d %-13s map %8.3f ms (%6.1f us/16fr, compulsory %.2f TB/s = %.3f)   '
                      'analytic %8.3f ms (%.3f)'
                      % (B, name, t, t * 1e3 * 16 / B, comp / t / 1e9, comp / t / 1e9 / 8000,
                         ta, 8 * B * h * w / ta / 1e9 / 8000), flush=True)
        del src, dst


if __name__ == '__main__':
    main()
