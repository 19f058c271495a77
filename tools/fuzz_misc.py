#!/usr/bin/env python3
"""Randomised check of the round-3 additions against the CPU oracle (GPU box): cv2.resize
restatement (bit for bit), fastFilter statistics, the three interpolate/ fills.
usage: python tools/fuzz_misc.py [n] [seed]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops, interpolate  # noqa: E402
from oracle import oracle  # noqa: E402


def close(a, b, tol, atol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(a)
    return bool(np.all(np.abs(a[m] - b[m]) <= tol * np.abs(b[m]) + atol))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    fails = 0
    for case in range(n):
        dt = rng.choice([np.float32, np.float64])
        tol = 2.5e-7 if dt == np.float32 else 1e-11
        sh, sw = int(rng.integers(1, 90)), int(rng.integers(1, 130))
        a = rng.standard_normal((sh, sw)).astype(dt)
        # resize: any size either way for the kernels, downscale for area
        dh, dw = int(rng.integers(1, 140)), int(rng.integers(1, 200))
        for name, oid in (('linear', 1), ('cubic', 2), ('lanczos4', 4)):
            g, o = ops.resize(a, (dh, dw), name), oracle.resize(a, (dh, dw), oid)
            # (equal_nan: a Lanczos4 position that rounds to fraction 1.0f divides by zero in the
            # published coefficient formula - NaN in every restatement alike)
            if not np.array_equal(g, o, equal_nan=True):
                fails += 1
                print('MISMATCH resize %s %s -> %s %s: max %g' % (name, a.shape, (dh, dw), dt.__name__,
                                                                   np.abs(g - o).max()))
        ah, aw = int(rng.integers(1, sh + 1)), int(rng.integers(1, sw + 1))
        g, o = ops.resize(a, (ah, aw), 'area'), oracle.resize(a, (ah, aw), 3)
        if not np.array_equal(g, o):
            fails += 1
            print('MISMATCH area %s -> %s %s: max %g' % (a.shape, (ah, aw), dt.__name__, np.abs(g - o).max()))
        # fastFilter statistics
        k = int(rng.integers(1, 40))
        ev = int(rng.integers(1, max(2, k // 3 + 1)))
        b = a.copy()
        b[rng.random(b.shape) < 0.15] = np.nan
        if ((2 * k + ev - 1) // ev) ** 2 <= 4096:
            for fn in ('median', 'nanmedian', 'mean', 'nanmean'):
                src = b if fn.startswith('nan') else a
                g = ops.fast_filter_stat(src, k, ev, fn)
                o = np.empty(g.shape)
                oracle._chk(oracle.lib().orc_fast_filter_stat(
                    oracle._p(np.ascontiguousarray(src)), oracle._dt(src), oracle.C.c_long(sh),
                    oracle.C.c_long(sw), oracle.C.c_long(k), oracle.C.c_long(ev),
                    oracle.C.c_int(oracle._FF_FN[fn]), oracle._p(o)), 'stat')
                # (means of zero-mean data cancel: judged against the data's scale; medians are exact)
                if not close(g, o, 0 if 'median' in fn else 1e-12, 0 if 'median' in fn else 1e-13):
                    fails += 1
                    print('MISMATCH stat %s k=%d every=%d %s %s' % (fn, k, ev, a.shape, dt.__name__))
        # the fills
        h, w = int(rng.integers(3, 80)), int(rng.integers(3, 120))
        grid = rng.random((h, w)).astype(dt) + 1
        mask = rng.random((h, w)) < rng.choice([0.05, 0.3, 0.7])
        kk, pw = int(rng.integers(1, 12)), float(rng.choice([1, 2, 3, 1.5]))
        g = interpolate.interpolate2dStructuredCrossAvg(grid.copy(), mask, kk, pw)
        o = oracle.interpolate2dStructuredCrossAvg(grid.copy(), mask, kk, pw)
        if not close(g, o, tol):
            fails += 1
            print('MISMATCH cross %s k=%d p=%g %s dens %.2f' % ((h, w), kk, pw, dt.__name__, mask.mean()))
        if w >= h:
            cx, cy = float(rng.integers(0, h)), float(rng.integers(0, w))
            g = interpolate.interpolateCircular2dStructuredIDW(grid.copy(), mask, kk, pw, 1.0, 0.3, cx + 0.5, cy + 0.5)
            o = oracle.interpolateCircular2dStructuredIDW(grid.copy(), mask, kk, pw, 1.0, 0.3, cx + 0.5, cy + 0.5)
            if not close(g, o, 1e-6 if dt == np.float32 else 1e-9):
                fails += 1
                print('MISMATCH circular %s k=%d p=%g %s' % ((h, w), kk, pw, dt.__name__))
        npnt = int(rng.integers(1, 40))
        x, y, v = rng.random(npnt) * h, rng.random(npnt) * w, rng.standard_normal(npnt)
        x[::4] = np.floor(x[::4])
        y[::4] = np.floor(y[::4])
        g = interpolate.interpolate2dUnstructuredIDW(x, y, v, np.zeros((h, w), dt), pw)
        o = oracle.interpolate2dUnstructuredIDW(x, y, v, np.zeros((h, w), dt), pw)
        if not close(g, o, 1e-6 if dt == np.float32 else 1e-9):
            fails += 1
            print('MISMATCH unstructured %s n=%d p=%g %s' % ((h, w), npnt, pw, dt.__name__))
        if (case + 1) % 25 == 0:
            print('%d cases, %d mismatches' % (case + 1, fails), flush=True)
    print('done: %d cases, %d mismatches' % (n, fails))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
