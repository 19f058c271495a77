"""micro-benchmark helper (GPU box): times single entry points with HIP events.

    python tests/bench_micro.py conv|copy|strip|remap|ringremap|intremap|configs|stencils|host|pipeline

conv / remap / strip: the filter, remap and fused kernels on 16 x 4K float32 frames
(IPA_STRIP_H, IPA_FRAMES_INNER, IMGPROC_HIP_LIB select variants); configs: the BASELINE
configurations C3..C5 kernel-only; stencils: the secondary stencils on one 4K frame;
host / pipeline: PCIe-inclusive host-to-host paths.  Results: profiles/r01_micro.txt.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def timeit(ctx, fn, n=10, warm=2):
    # IPA_MICRO_SCALE=30 times 30x more launches after 30x more warm-up: steady-state clocks
    scale = int(os.environ.get('IPA_MICRO_SCALE', '1'))
    n, warm = n * scale, warm * scale
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
    if what == 'intremap':
        # integer frames: every product and sum in double, results equal to the oracle bit for bit
        from imgprocessor_amd.utils import getPerspectiveTransform
        quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
        rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
        Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        for dt in (np.uint8, np.uint16):
            a = rng.random((B, h, w), dtype=np.float32)
            isrc = ctx.to_device((a * (255 if dt == np.uint8 else 4095)).astype(dt))
            idst = ctx.empty((B, h, w), dt)
            for interp in ('linear', 'linear_cv_q5', 'cubic', 'cubic_cv', 'lanczos4'):
                t1 = timeit(ctx, lambda: ops.remap(isrc, dmx, dmy, interp, out=idst))
                t2 = timeit(ctx, lambda: ops.warp_perspective(isrc, Hm, (h, w), interp, out=idst))
                print('%-7s %-13s remap(maps) %8.1f us   warp %8.1f us  (%6.1f Gpx/s)'
                      % (np.dtype(dt).name, interp, t1, t2, px / t2 / 1e3))
    if what == 'ringremap':
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        from imgprocessor_amd.utils import getPerspectiveTransform
        quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
        rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
        Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))  # the C3 / C5 warp
        for interp in ('linear', 'cubic', 'lanczos4'):
            for name, fn in (('map', lambda: ops.remap(src, dmx, dmy, interp, out=dst)),
                             ('homography', lambda: ops.warp_perspective(src, Hm, (h, w), interp,
                                                                         out=dst)),
                             ('lens model', lambda: ops.undistort(src, Kc, dc, Kc, interp,
                                                                  out=dst))):
                ts = []
                for on in (0, 2):
                    old = ctx.set_tuning(ring_remap=on)
                    ts.append(timeit(ctx, fn))
                    ctx.set_tuning(**old)
                print('%-9s %-11s gather %8.1f us   ring %8.1f us  (%5.1f Gpx/s)'
                      % (interp, name, ts[0], ts[1], px / ts[1] / 1e3))
    if what == 'remap':
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        for interp in ('nearest', 'linear', 'cubic', 'lanczos4'):
            t = timeit(ctx, lambda: ops.remap(src, dmx, dmy, interp, out=dst))
            print('remap %-9s %8.1f us  %7.1f Gpx/s %6.0f GB/s(16B/px)'
                  % (interp, t, px / t / 1e3, 16 * px / t / 1e3))
        t = timeit(ctx, lambda: ops.undistort(src, Kc, dc, Kc, out=dst))
        print('undistort(analytic) %8.1f us  %7.1f Gpx/s %6.0f GB/s(8B/px)'
              % (t, px / t / 1e3, 8 * px / t / 1e3))
    if what == 'host':
        # PCIe-inclusive: host ndarray in, host ndarray out (pageable numpy buffers), one 4K frame
        import time
        from imgprocessor_amd.camera.LensDistortion import LensDistortion
        from imgprocessor_amd.filters import filter as ipa_filter
        img = src.get()[0]
        ld = LensDistortion(newCameraMatrix='same', ctx=ctx)
        ld.setCameraParams(Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2], dc[0], dc[1], dc[4], dc[2], dc[3])
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)

        def wall(fn, n=10):
            fn()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            return (time.perf_counter() - t0) / n * 1e3
        t = wall(lambda: ipa_filter(ld.correct(img, keepSize=True), k5))
        print('host: correct() then filter(), 2 round trips   %7.2f ms/frame  %6.2f Gpx/s'
              % (t, h * w / t / 1e6))
        t = wall(lambda: ops.remap_conv2d(ctx.to_device(img), dmx, dmy, k5).get())
        print('host: fused remap_conv2d, 1 round trip          %7.2f ms/frame  %6.2f Gpx/s'
              % (t, h * w / t / 1e6))
        batch = src.get()
        t = wall(lambda: ops.remap_conv2d(ctx.to_device(batch), dmx, dmy, k5).get(), n=3)
        print('host: fused, 16-frame batch, 1 round trip       %7.2f ms/frame  %6.2f Gpx/s'
              % (t / B, B * h * w / t / 1e6))
    if what == 'configs':
        # kernel-only timings of the BASELINE configurations C3..C5 at batch 16 (C5: 4 x 8K)
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        k7 = np.random.default_rng(123).random((7, 7))
        k7 /= k7.sum()
        u16 = ctx.to_device((rng.random((B, h, w)) * 4095).astype(np.uint16))
        t = timeit(ctx, lambda: ops.remap_conv2d(u16, dmx, dmy, k7, out=dst))
        print('C4 u16->f32 undistort + 7x7 fused   %8.1f us  %7.1f Gpx/s' % (t, px / t / 1e3))
        t = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k7, out=dst))
        print('   f32      undistort + 7x7 fused   %8.1f us  %7.1f Gpx/s' % (t, px / t / 1e3))
        from imgprocessor_amd.utils import getPerspectiveTransform
        quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
        rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
        Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
        g9 = ops.gaussian_kernel1d(1.0)
        tmp = ctx.empty((B, h, w), np.float32)
        for interp in ('linear', 'cubic'):
            t1 = timeit(ctx, lambda: ops.warp_perspective(src, Hm, (h, w), interp, out=tmp))
            t2 = timeit(ctx, lambda: ops.sepconv2d(tmp, g9, g9, out=dst))
            print('C3 warp %-6s %8.1f us + sep 9+9 %8.1f us  %7.1f Gpx/s'
                  % (interp, t1, t2, px / (t1 + t2) / 1e3))
            t = timeit(ctx, lambda: ops.warp_perspective_sepconv2d(src, Hm, (h, w), g9, g9, interp,
                                                                   out=dst))
            print('C3 warp %-6s -> sep 9+9 in one kernel %8.1f us  %7.1f Gpx/s'
                  % (interp, t, px / t / 1e3))
        k11 = np.random.default_rng(321).random((11, 11))
        k11 /= k11.sum()
        t = timeit(ctx, lambda: ops.warp_perspective_conv2d(src, Hm, (h, w), k11, 'cubic', out=dst))
        print('C5-like 4K bicubic warp + 11x11 (2 launches) %8.1f us  %7.1f Gpx/s'
              % (t, px / t / 1e3))
        # map-based: one kernel (fused_big.hip) unless IPA_BIG_FUSED=0
        dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
        k9 = np.random.default_rng(322).random((9, 9))
        k9 /= k9.sum()
        for interp, kk in (('linear', k9), ('linear', k11), ('cubic', k9), ('cubic', k11)):
            t = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, kk, interp, out=dst))
            print('map remap %-6s + %2dx%-2d %8.1f us  %7.1f Gpx/s'
                  % (interp, len(kk), len(kk), t, px / t / 1e3))
    if what == 'pipeline':
        # end to end, host to host, with page-locked arrays and overlapped workers (C4 style)
        import time
        from imgprocessor_amd.sharding import FramePipeline
        k7 = np.random.default_rng(123).random((7, 7))
        k7 /= k7.sum()
        N = 48
        for depth in (1, 2, 3, 4):
            pipe = FramePipeline(0, depth)
            maps = {id(c): ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=c, device=True)
                    for c in pipe.contexts}
            for dt in (np.uint16, np.float32):
                fin = pipe.pinned_empty((N, h, w), dt)
                fout = pipe.pinned_empty((N, h, w), np.float32)
                fin[...] = (rng.random((1, h, w)) * 4095).astype(dt)

                def fn(c, d, o):
                    mx, my = maps[id(c)]
                    ops.remap_conv2d(d, mx, my, k7, out=o)
                pipe.run(fin[:depth], fout[:depth], fn)  # warm
                t0 = time.perf_counter()
                pipe.run(fin, fout, fn)
                t = (time.perf_counter() - t0) / N * 1e3
                print('pipeline depth %d  %-7s -> f32 undistort + 7x7  %6.2f ms/frame  %6.2f Gpx/s'
                      % (depth, np.dtype(dt).name, t, h * w / t / 1e6))
                del fin, fout
    if what == 'stencils':
        # the secondary stencils on ONE 4K frame (float32 and float64)
        from imgprocessor_amd.filters import (standardDeviation2d, varYSizeGaussianFilter,
                                              maskedFilter, nan_maximum_filter, medianThreshold)
        from imgprocessor_amd.interpolate import interpolate2dStructuredIDW
        for dt in (np.float32, np.float64):
            img = ctx.to_device((0.2 + rng.random((h, w))).astype(dt))
            blurred = ops.gaussian_filter(img, (5, 5))
            m = ctx.to_device((rng.random((h, w)) < 0.05).astype(np.uint8))
            one = h * w
            t = timeit(ctx, lambda: ops.local_std(img, blurred, (5, 5)))
            print('%-8s local_std k5            %8.1f us  %6.2f Gpx/s' % (np.dtype(dt).name, t, one / t / 1e3))
            t = timeit(ctx, lambda: ops.gaussian_filter(img, (5, 5)))
            print('%-8s gaussian sigma 5 (41+41) %8.1f us  %6.2f Gpx/s' % (np.dtype(dt).name, t, one / t / 1e3))
            t = timeit(ctx, lambda: ops.median_threshold(img, 0.1))
            print('%-8s medianThreshold 3x3      %8.1f us  %6.2f Gpx/s' % (np.dtype(dt).name, t, one / t / 1e3))
            t = timeit(ctx, lambda: ops.nan_max(img, 7))
            print('%-8s nan_max k7               %8.1f us  %6.2f Gpx/s' % (np.dtype(dt).name, t, one / t / 1e3))
            work = ctx.empty((h, w), dt)

            def fill():
                work.copy_from(img)
                ops.masked_mean(work, m, 30, True, fn='mean')
            t = timeit(ctx, fill)
            print('%-8s maskedFilter mean k30 5%%  %8.1f us  %6.2f Gpx/s' % (np.dtype(dt).name, t, one / t / 1e3))
            t = timeit(ctx, lambda: varYSizeGaussianFilter(img, (0, 4), 1), n=3, warm=1)
            print('%-8s varYSizeGaussian (0,4),1 %8.1f us  %6.2f Gpx/s' % (np.dtype(dt).name, t, one / t / 1e3))
