"""Round-3 additions (interpolate/ fills, fastFilter / fastMean, cv2.resize) timed on device-resident
arrays next to the CPU oracle on the same input.  GPU box only; the oracle is the checker here.

    python tools/bench_more.py [--no-cpu]
"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.filters import fastFilter, fastMean  # noqa: E402
from oracle import oracle  # noqa: E402

ctx = ia.default_context(0)


def gpu_us(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3


NO_CPU = '--no-cpu' in sys.argv   # (profiling passes: kernels only)


def cpu_ms(fn):
    if NO_CPU:
        return float('nan'), None
    t = time.perf_counter()
    r = fn()
    return (time.perf_counter() - t) * 1e3, r


def line(name, us, cms, nbytes, ok):
    print('%-58s %9.1f us  %7.1f GB/s  oracle %9.1f ms  x%-7.0f %s' % (
        name, us, nbytes / us / 1e3, cms, cms * 1e3 / us, 'same' if ok else 'DIFFERS'), flush=True)


def close(a, b, tol):
    if b is None:
        return True
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(a)
    return bool(np.all(np.abs(a[m] - b[m]) <= tol * max(1.0, float(np.abs(b[m]).max()))))


rng = np.random.default_rng(3)

# interpolate2dUnstructuredIDW: the reference's demo size, and 4K
for (h, w, n) in ((1000, 2000, 30), (2160, 3840, 30), (2160, 3840, 300)):
    x, y, v = rng.integers(0, h, n), rng.integers(0, w, n), rng.random(n)
    d = ctx.empty((h, w), np.float32)
    us = gpu_us(lambda: ops.unstructured_idw(x, y, v, d, 2))
    cms, want = cpu_ms(lambda: oracle.interpolate2dUnstructuredIDW(x, y, v, np.zeros((h, w), np.float32), 2))
    line('unstructured IDW %dx%d f32, %d points' % (h, w, n), us, cms, h * w * 4, close(d.get(), want, 2.5e-7))

# hole fills: 4K float32, one 25 % hole + 2 % scattered
h, w = 2160, 3840
grid = (rng.random((h, w)) + np.linspace(1, 2, w)[None, :]).astype(np.float32)
mask = rng.random((h, w)) < 0.02
mask[h // 2 - 100:h // 2 + 100, w // 2 - 200:w // 2 + 200] = True
mask[0, :] = False
mask[:, 0] = False
dm = ctx.to_device(mask.astype(np.uint8))
for k in (5, 15):
    d = ctx.to_device(grid)
    dsrc = ctx.to_device(grid)

    def run_cross():
        d.copy_from(dsrc) if hasattr(d, 'copy_from') else None
        ops.cross_avg_fill(d, dm, k, 2)
    d = ctx.to_device(grid)
    ops.cross_avg_fill(d, dm, k, 2)
    got = d.get()
    us = gpu_us(lambda: ops.cross_avg_fill(d, dm, k, 2), n=10)
    cms, want = cpu_ms(lambda: oracle.interpolate2dStructuredCrossAvg(grid.copy(), mask, k, 2))
    line('cross average fill 4K f32, kernel %d' % k, us, cms, h * w * 9, close(got, want, 2.5e-7))
    sq = np.ascontiguousarray(grid[:, :h])
    msq = np.ascontiguousarray(mask[:, :h])
    d = ctx.to_device(sq)
    dmsq = ctx.to_device(msq.astype(np.uint8))
    ops.circular_idw_fill(d, dmsq, k, 2, 1, 0.5, h // 2, h // 2)
    got = d.get()
    us = gpu_us(lambda: ops.circular_idw_fill(d, dmsq, k, 2, 1, 0.5, h // 2, h // 2), n=10)
    cms, want = cpu_ms(lambda: oracle.interpolateCircular2dStructuredIDW(sq.copy(), msq, k, 2, 1, 0.5, h // 2, h // 2))
    line('circular IDW fill 2160x2160 f32, kernel %d' % k, us, cms, h * h * 9, close(got, want, 2.5e-7))

# cv2.resize
img = rng.random((h, w)).astype(np.float32)
dimg = ctx.to_device(img)
for (name, oi, dsz) in (('linear', oracle.RESIZE_LINEAR, (1080, 1920)), ('linear', oracle.RESIZE_LINEAR, (4320, 7680)),
                        ('cubic', oracle.RESIZE_CUBIC, (4320, 7680)), ('lanczos4', oracle.RESIZE_LANCZOS4, (4320, 7680)),
                        ('area', oracle.RESIZE_AREA, (216, 384)), ('area', oracle.RESIZE_AREA, (1000, 1777))):
    out = ctx.empty(dsz, np.float32)
    us = gpu_us(lambda: ops.resize(dimg, dsz, name, out=out))
    cms, want = cpu_ms(lambda: oracle.resize(img, dsz, oi))
    line('resize 4K f32 -> %dx%d %s' % (dsz[0], dsz[1], name), us, cms, (h * w + dsz[0] * dsz[1]) * 4,
         NO_CPU or np.array_equal(out.get(), want, equal_nan=True))

# fastFilter window statistics (the reference's defaults: ksize 30, every = ksize // 3 ... ) and fastMean
for (fn, ks, ev) in (('median', 30, 10), ('mean', 30, 10), ('median', 60, 20), ('nanmedian', 30, 10)):
    a = img.astype(np.float64)
    if fn.startswith('nan'):
        a[rng.random(a.shape) < 0.05] = np.nan
    da = ctx.to_device(a)
    got = ops.fast_filter_stat(da, ks, ev, fn)
    got = got.get() if hasattr(got, 'get') else got
    us = gpu_us(lambda: ops.fast_filter_stat(da, ks, ev, fn), n=10)
    cms, want = cpu_ms(lambda: oracle.fastFilter(a, ks, ev, False, fn))
    line('fastFilter statistics 4K f64 %s ksize %d every %d' % (fn, ks, ev), us, cms, h * w * 8,
         close(got[:-1, :-1], want, 1e-12))   # (the reference drops the last row / column of cells)
    us = gpu_us(lambda: fastFilter(da, ks, ev, fn=fn, ctx=ctx), n=10)
    got = fastFilter(da, ks, ev, fn=fn, ctx=ctx).get()
    cms, want = cpu_ms(lambda: oracle.fastFilter(a, ks, ev, True, fn))
    line('fastFilter 4K f64 %s ksize %d (statistics + Lanczos4 enlargement)' % (fn, ks), us, cms, h * w * 16,
         close(got, want, 1e-11))
us = gpu_us(lambda: fastMean(dimg, 10, ctx=ctx), n=10)
got = fastMean(dimg, 10, ctx=ctx)
got = got.get()
cms, want = cpu_ms(lambda: oracle.fastMean(img, 10))
line('fastMean 4K f32 f=10 (area down + linear up)', us, cms, h * w * 8, NO_CPU or np.array_equal(got, want))
