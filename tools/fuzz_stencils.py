#!/usr/bin/env python3
"""Randomised check of the filters/ and interpolate/ stencils against the CPU oracle (GPU box):
dense and separable filters (every border mode, masks, float32 / float64, odd and rectangular
kernels), the row-dependent Gaussian, local standard deviation, masked mean / median, NaN
maximum, median threshold, closest distance, position uncertainty, IDW / fast IDW.
usage: python tools/fuzz_stencils.py [n] [seed]"""
import io
import contextlib
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops, filters, interpolate  # noqa: E402
from imgprocessor_amd.render import closestDirectDistance  # noqa: E402
from imgprocessor_amd.uncertainty import positionToIntensityUncertainty  # noqa: E402
from oracle import oracle  # noqa: E402


def close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    inf = np.isinf(a) | np.isinf(b)
    if not np.array_equal(a[inf], b[inf]):      # (e.g. std2d's zero divisors on the rim)
        return False
    m = ~np.isnan(a) & ~inf
    return bool(np.all(np.abs(a[m] - b[m]) <= rtol * np.abs(b[m]) + atol))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    fails = 0

    def check(name, got, want, rtol, atol, info):
        nonlocal fails
        if not close(got, want, rtol, atol):
            fails += 1
            g, w_ = np.asarray(got, np.float64), np.asarray(want, np.float64)
            d = np.abs(np.nan_to_num(g) - np.nan_to_num(w_)) if g.shape == w_.shape else None
            print('MISMATCH %s %s: %s' % (name, info, 'shape %s vs %s' % (g.shape, w_.shape) if d is None
                                          else 'max |d| %g at %s' % (d.max(), np.unravel_index(d.argmax(), d.shape))),
                  flush=True)

    modes = ['reflect', 'constant', 'wrap', 'mirror', 'nearest']
    for case in range(n):
        dt = rng.choice([np.float32, np.float64])
        r, a0 = (1e-5, 1e-5) if dt == np.float32 else (1e-11, 1e-12)
        h, w = int(rng.integers(12, 200)), int(rng.integers(12, 330))
        img = (rng.random((h, w)) * 4 - 1).astype(dt)
        scale = 4.0
        kh, kw = [int(v) for v in rng.choice([1, 3, 5, 7, 9, 11, 2, 4, 13], 2)]
        kh, kw = min(kh, h - 1), min(kw, w - 1)
        if rng.random() < 0.6:
            kw = kh = min(kh | 1, min(h, w) - 1 | 1) if (kh | 1) < min(h, w) else 3
        k = rng.standard_normal((kh, kw))
        mode = str(rng.choice(modes))
        cval = float(rng.choice([0.0, 0.7]))
        info = '%s %dx%d k %dx%d %s' % (np.dtype(dt).name, h, w, kh, kw, mode)
        check('conv2d', ops.conv2d(img, k, mode, cval), oracle.conv2d(img, k, mode, cval),
              r, a0 * scale * np.abs(k).sum(), info)
        if kh == kw and kh % 2 == 1 and rng.random() < 0.5:
            mask = rng.random((h, w)) < 0.4
            check('conv2d mask', ops.conv2d(img, k, mode, cval, mask=mask),
                  oracle.conv2d(img, k, mode, cval, mask=mask), r, a0 * scale * np.abs(k).sum(), info)
            if mode == 'reflect':   # (the reference's modey test accepts nothing else)
                check('maskedConvolve', filters.maskedConvolve(img, k, mask, mode),
                      oracle.maskedConvolve(img, k, mask, mode), r, a0 * scale * np.abs(k).sum(), info)
        ny, nx = int(rng.choice([1, 3, 5, 9, 15, 31])), int(rng.choice([1, 3, 5, 9, 17, 41]))
        ny, nx = min(ny, 2 * (h // 2) - 1), min(nx, 2 * (w // 2) - 1)
        ky, kx = rng.random(ny), rng.random(nx)
        check('sepconv2d', ops.sepconv2d(img, ky, kx, mode, cval), oracle.sepconv2d(img, ky, kx, mode, cval),
              r, a0 * scale * ky.sum() * kx.sum(), info + ' sep %d+%d' % (ny, nx))
        sig = float(rng.choice([0.5, 1.0, 2.3, 4.0]))
        if int(4 * sig + 0.5) < min(h, w) // 2:
            check('gaussian', filters.gaussian_filter(img, sig, 'reflect'), oracle.gaussian_filter(img, sig),
                  r, a0 * scale, info + ' sigma %g' % sig)
        ks = int(rng.choice([3, 5, 9]))
        if 4 * ks + 1 < min(h, w):
            with contextlib.redirect_stdout(io.StringIO()):
                got = filters.standardDeviation2d(img, ks)
            check('std2d', got, oracle.standardDeviation2d(img, ks), max(r, 1e-5) if dt == np.float32 else 1e-9,
                  a0 * scale, info + ' ksize %d' % ks)
        smax = float(rng.choice([1.0, 2.0, 4.0]))
        if int(smax * 2.5) + 2 < h:
            stdx = float(rng.choice([0, 1.0]))
            with contextlib.redirect_stdout(io.StringIO()):
                got = filters.varYSizeGaussianFilter(img, (0, smax), stdx)
                want = oracle.varYSizeGaussianFilter(img, (0, smax), stdx)
            check('varYSizeGaussianFilter', got, want, r, a0 * scale, info + ' stdy %g stdx %g' % (smax, stdx))
        # masked filters
        mk = int(rng.choice([3, 6, 10, 30]))
        msk = rng.random((h, w)) < rng.choice([0.05, 0.3])
        for fn in ('mean', 'median'):
            for fill in (True, False):
                got = filters.maskedFilter(img.copy(), msk, mk, fill, fn)
                want = oracle.maskedFilter(img.copy(), msk, mk, fill, fn)
                check('maskedFilter %s fill=%s' % (fn, fill), got, want, r if fn == 'mean' else 0,
                      a0 * scale if fn == 'mean' else 0, info + ' ksize %d' % mk)
        nimg = img.copy()
        nimg[rng.random((h, w)) < 0.2] = np.nan
        nk = int(rng.choice([3, 5, 10, 21]))
        check('nan_maximum_filter', filters.nan_maximum_filter(nimg, nk), oracle.nan_maximum_filter(nimg, nk),
              0, 0, info + ' ksize %d' % nk)
        thr = float(rng.choice([0.05, 0.3]))
        cond = str(rng.choice(['>', '<']))
        g1, i1 = filters.medianThreshold(np.abs(img) + 0.1, thr, condition=cond)
        o1, j1 = oracle.medianThreshold(np.abs(img) + 0.1, thr, condition=cond)
        check('medianThreshold', g1, o1, 0, 0, info + ' thr %g %s' % (thr, cond))
        if not np.array_equal(np.asarray(i1), np.asarray(j1)):
            fails += 1
            print('MISMATCH medianThreshold indices %s' % info, flush=True)
        msz = int(rng.choice([2, 4, 5, 7, 8, 11]))
        if msz < min(h, w):
            g2, i2 = filters.medianThreshold(np.abs(img) + 0.1, thr, size=msz, condition=cond)
            o2, j2 = oracle.medianThreshold(np.abs(img) + 0.1, thr, size=msz, condition=cond)
            check('medianThreshold size %d' % msz, g2, o2, 0, 0, info + ' thr %g %s' % (thr, cond))
            if not np.array_equal(np.asarray(i2), np.asarray(j2)):
                fails += 1
                print('MISMATCH medianThreshold size %d indices %s' % (msz, info), flush=True)
        # point-spread IDW (sweeps with in-order dependence; more columns than rows or square:
        # where the source's window is defined)
        if w >= h:
            from imgprocessor_amd.interpolate import interpolate2dStructuredPointSpreadIDW as psidw
            pm = rng.random((h, w)) < rng.choice([0.1, 0.5, 0.9])
            pm[h // 3:2 * h // 3, w // 3:2 * w // 3] = True
            pm[0, 0] = False
            pk, pp = int(rng.choice([2, 5, 15])), float(rng.choice([1, 2, 3.5]))
            check('pointSpreadIDW', psidw(img, pm, pk, pp), oracle.interpolate2dStructuredPointSpreadIDW(img, pm, pk, pp),
                  3e-5 if dt == np.float32 else 1e-9, 0, info + ' kernel %d power %g' % (pk, pp))
        b = rng.random((h, w)) < 0.02
        ck = int(rng.choice([3, 10, 30]))
        check('closestDirectDistance', closestDirectDistance(b, ck), oracle.closestDirectDistance(b, ck), 0, 0,
              info + ' ksize %d' % ck)
        ps = int(rng.choice([3, 5, 7]))
        sx_, sy_ = float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.3, 1.5))
        check('positionToIntensityUncertainty const', positionToIntensityUncertainty(img, sx_, sy_, ps),
              oracle.positionToIntensityUncertainty(img, sx_, sy_, ps), max(r, 1e-5) if dt == np.float32 else 1e-9,
              a0 * scale, info + ' k %d' % ps)
        sxm, sym = rng.uniform(0.3, 1.5, (h, w)), rng.uniform(0.3, 1.5, (h, w))
        check('positionToIntensityUncertainty maps', positionToIntensityUncertainty(img, sxm, sym, ps),
              oracle.positionToIntensityUncertainty(img, sxm, sym, ps), max(r, 1e-5) if dt == np.float32 else 1e-9,
              a0 * scale, info + ' k %d' % ps)
        ik, pw = int(rng.choice([2, 5, 15])), float(rng.choice([1, 2, 3]))
        check('IDW', interpolate.interpolate2dStructuredIDW(img.copy(), msk, ik, pw),
              oracle.interpolate2dStructuredIDW(img.copy(), msk, ik, pw), max(r, 2e-6) if dt == np.float32 else 1e-10,
              a0, info + ' k %d p %g' % (ik, pw))
        mn = int(rng.choice([1, 3, 5, 9]))
        check('FastIDW', interpolate.interpolate2dStructuredFastIDW(img.copy(), msk, ik, pw, mn),
              oracle.interpolate2dStructuredFastIDW(img.copy(), msk, ik, pw, mn),
              max(r, 2e-6) if dt == np.float32 else 1e-10, a0, info + ' k %d p %g n %d' % (ik, pw, mn))
        if (case + 1) % 20 == 0:
            print('%d cases, %d mismatches' % (case + 1, fails), flush=True)
    print('done: %d cases, %d mismatches' % (n, fails))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
