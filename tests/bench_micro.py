"""micro-benchmark helper (GPU box): times single entry points with HIP events.

    python tests/bench_micro.py conv|copy|strip|remap|fused
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def timeit(ctx, fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3  # us


if __name__ == '__main__':
    ctx = ia.default_context(0)
    B, h, w = 16, 2160, 3840
    rng = np.random.default_rng(0)
    src = ctx.to_device(rng.random((B, h, w), dtype=np.float32))
    dst = ctx.empty((B, h, w), np.float32)
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    what = sys.argv[1] if len(sys.argv) > 1 else 'conv'
    px = B * h * w
    Kc = np.array([[3840., 0, 1919.5], [0, 3840., 1079.5], [0, 0, 1]])
    dc = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    if what == 'conv':
        for K in (3, 5, 7, 9, 11):
            k = rng.random((K, K))
            t = timeit(ctx, lambda: ops.conv2d(src, k, out=dst))
            print('conv %2dx%-2d  %8.1f us  %7.1f Gpx/s  %6.0f GB/s(8B/px)'
                  % (K, K, t, px / t / 1e3, 8 * px / t / 1e3))
        kk = np.exp(-0.5 * np.arange(-4, 5) ** 2)
        kk /= kk.sum()
        t = timeit(ctx, lambda: ops.sepconv2d(src, kk, kk, out=dst))
        print('sepconv 9+9  %8.1f us  %7.1f Gpx/s  %6.0f GB/s(8B/px)'
              % (t, px / t / 1e3, 8 * px / t / 1e3))
    if what == 'copy':
        t = timeit(ctx, lambda: ctx._check(ctx._lib.ipa_memcpy_d2d(ctx.handle, dst.ptr, src.ptr,
                                                                   src.nbytes)))
        print('d2d copy     %8.1f us  %6.0f GB/s (r+w)' % (t, 2 * src.nbytes / t / 1e3))
    if what == 'strip':
        k = rng.random((5, 5))
        sh = os.environ.get('IPA_STRIP_H')
        t = timeit(ctx, lambda: ops.conv2d(src, k, out=dst))
        print('conv5     strip_h=%s  %8.1f us %6.0f GB/s(8B/px)' % (sh, t, 8 * px / t / 1e3))
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        t = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst))
        print('fused_map strip_h=%s  %8.1f us %6.0f GB/s(16B/px)' % (sh, t, 16 * px / t / 1e3))
    if what == 'remap':
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        for interp in ('nearest', 'linear', 'cubic', 'lanczos4'):
            t = timeit(ctx, lambda: ops.remap(src, dmx, dmy, interp, out=dst))
            print('remap %-9s %8.1f us  %7.1f Gpx/s %6.0f GB/s(16B/px)'
                  % (interp, t, px / t / 1e3, 16 * px / t / 1e3))
        t = timeit(ctx, lambda: ops.undistort(src, Kc, dc, Kc, out=dst))
        print('undistort(analytic) %8.1f us  %7.1f Gpx/s %6.0f GB/s(8B/px)'
              % (t, px / t / 1e3, 8 * px / t / 1e3))
