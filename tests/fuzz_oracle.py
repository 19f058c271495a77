#!/usr/bin/env python3
"""Randomised check of the remap family against the CPU oracle (GPU box): float32 within
1e-5 of the data range, integer results bit for bit.  Not collected by pytest (run by hand on
the GPU box).  usage: python tests/fuzz_oracle.py [n] [seed]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from oracle import oracle  # noqa: E402

INTERPS = {'nearest': oracle.NEAREST, 'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS,
           'cubic_cv': oracle.CUBIC_CV, 'linear_cv_q5': oracle.LINEAR | oracle.Q5,
           'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5, 'lanczos4': oracle.LANCZOS4}
BORDERS = {'constant': oracle.CONSTANT, 'replicate': oracle.REPLICATE, 'reflect': oracle.REFLECT,
           'wrap': oracle.WRAP, 'reflect101': oracle.REFLECT101}


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = ia.default_context(0)
    fails = 0
    worst = 0.0
    for case in range(n_cases):
        big = bool(os.environ.get('FUZZ_BIG'))    # frames up to 1300 x 2700, batches of 4 / 8
        hm, wm = (1300, 2700) if big else (300, 700)
        h, w = int(rng.integers(20, hm)), int(rng.integers(40, wm))
        dh, dw = (h, w) if rng.random() < 0.6 else (int(rng.integers(20, hm)), int(rng.integers(40, wm)))
        n = int(rng.choice([4, 8])) if big else int(rng.integers(1, 5))
        dt = rng.choice([np.float32, np.float32, np.uint8, np.uint16])
        a = rng.random((n, h, w))
        src = a.astype(np.float32) if dt == np.float32 else np.round(
            a * (255 if dt == np.uint8 else 4095)).astype(dt)
        ang = np.deg2rad(rng.choice([0, 0, 3, -8, 45, 90]) + rng.normal(0, 0.5))
        sc = rng.choice([1.0, 0.9, 1.15, 0.5, 2.0])
        yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
        x0, y0 = xx - dw / 2, yy - dh / 2
        mx = (sc * (np.cos(ang) * x0 - np.sin(ang) * y0) + w / 2 + rng.normal(0, 10) +
              2 * np.sin(yy / 31.0)).astype(np.float32)
        my = (sc * (np.sin(ang) * x0 + np.cos(ang) * y0) + h / 2 + rng.normal(0, 10) +
              2 * np.cos(xx / 47.0)).astype(np.float32)
        if rng.random() < 0.15:
            mx[dh // 2, dw // 3:dw // 3 + 5] = np.nan
        iname = str(rng.choice(list(INTERPS)))
        bname = str(rng.choice(list(BORDERS)))
        cval = float(rng.choice([0.0, 0.3, 17.0])) if dt != np.float32 else float(rng.choice([0.0, 0.3]))
        sep = rng.choice([3, 5, 7, 9, 13], 2)       # (the separable chain's taps, see below)
        sepw = rng.random(13) + 0.1
        pp = rng.normal(0, 2e-5, 2)                # (the warp's perspective terms, see below)
        K = int(rng.choice([3, 5, 7, 9]))          # (the fused chain's filter, see below)
        kern = rng.random((K, K))
        # every third case: an exact outer product - the library then takes its separable K + K loops (knob
        # rank1_sep, round 6) and must still be within the tolerance of the oracle's dense double sum
        outer = rng.random(2 * K)
        if case % 3 == 0:
            kern = np.outer(outer[:K] - 0.3, outer[K:] + 0.1)
        kern /= kern.sum()
        cmode = str(rng.choice(['reflect', 'constant', 'wrap', 'mirror', 'nearest']))
        if os.environ.get('FUZZ_ONLY') and int(os.environ['FUZZ_ONLY']) != case:
            continue   # (every random draw of the case is above: the stream stays in step)
        got = ops.remap(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), iname, bname,
                        cval).get()
        for f in range(n):
            want = oracle.remap(src[f], mx, my, INTERPS[iname], BORDERS[bname], cval)
            if dt == np.float32:
                ok = np.isnan(got[f]) == np.isnan(want)
                d = np.abs(np.nan_to_num(got[f]) - np.nan_to_num(want)).max() if ok.all() else np.inf
                worst = max(worst, float(d))
                bad = d > 1e-5 * max(1.0, float(np.abs(np.nan_to_num(want)).max()))
            else:
                bad = not np.array_equal(got[f], want)
            if bad:
                fails += 1
                print('MISMATCH case %d frame %d: %s %dx%d -> %dx%d %s %s cval %g'
                      % (case, f, np.dtype(dt).name, h, w, dh, dw, iname, bname, cval))
                df = np.abs(got[f].astype(np.float64) - want.astype(np.float64))
                idx = np.argwhere(df > 0)
                print('   %d values differ, max |d| %g; first at %s: got %r want %r, map (%r, %r)'
                      % (len(idx), np.nanmax(df), idx[0].tolist(), got[f][tuple(idx[0])],
                         want[tuple(idx[0])], mx[tuple(idx[0])], my[tuple(idx[0])]))
                break
        # the fused chain (remap -> K x K filter in one kernel) on the same case: float32 and uint16
        # frames, every interpolation the fused entry point takes
        if iname != 'nearest':     # (round 6: every element type - two launches where no fused kernel is built)
            try:
                gotf = ops.remap_conv2d(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), kern,
                                        iname, bname, cval, cmode).get()
            except NotImplementedError:
                gotf = None
            if gotf is not None:
                for f in range(n):
                    mid = oracle.remap(src[f], mx, my, INTERPS[iname], BORDERS[bname], cval,
                                       out_dtype=np.float32)
                    want = oracle.conv2d(mid, kern, cmode)
                    ok = np.isnan(gotf[f]) == np.isnan(want)
                    d = np.abs(np.nan_to_num(gotf[f]) - np.nan_to_num(want)).max() if ok.all() else np.inf
                    scale = max(1.0, float(np.abs(np.nan_to_num(want)).max()))
                    worst = max(worst, float(d) / scale)
                    if d > 1e-5 * scale:
                        fails += 1
                        print('MISMATCH fused case %d frame %d: %s %dx%d -> %dx%d n=%d %s %s cval %g K=%d %s: '
                              'max |d| %g' % (case, f, np.dtype(dt).name, h, w, dh, dw, n, iname, bname, cval,
                                              K, cmode, d))
                        break
        # ... and the separable form (remap -> ky then kx), 3 / 5 / 7 / 9 / 13 taps per axis
        if iname != 'nearest':
            ny_, nx_ = int(sep[0]), int(sep[1])
            if ny_ < dh and nx_ < dw:
                ky_, kx_ = sepw[:ny_] / sepw[:ny_].sum(), sepw[:nx_][::-1] / sepw[:nx_].sum()
                try:
                    gots = ops.remap_sepconv2d(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), ky_, kx_,
                                               iname, bname, cval, cmode).get()
                except NotImplementedError:
                    gots = None
                if gots is not None:
                    for f in range(n):
                        mid = oracle.remap(src[f], mx, my, INTERPS[iname], BORDERS[bname], cval,
                                           out_dtype=np.float32)
                        want = oracle.sepconv2d(mid, ky_, kx_, cmode)
                        ok = np.isnan(gots[f]) == np.isnan(want)
                        d = np.abs(np.nan_to_num(gots[f]) - np.nan_to_num(want)).max() if ok.all() else np.inf
                        if d > 1e-5 * max(1.0, float(np.abs(np.nan_to_num(want)).max())):
                            fails += 1
                            print('MISMATCH sep case %d frame %d: %dx%d -> %dx%d n=%d %s %s cval %g taps %d+%d %s: '
                                  'max |d| %g' % (case, f, h, w, dh, dw, n, iname, bname, cval, ny_, nx_, cmode, d))
                            break
        # the same case as a homography warp (coordinates evaluated in the kernel - or, for batches
        # of bicubic / Lanczos4 warps, once into stored coordinates) against the oracle's
        M = np.array([[sc * np.cos(ang), -sc * np.sin(ang), w / 2 - sc * (np.cos(ang) * dw / 2 - np.sin(ang) * dh / 2)],
                      [sc * np.sin(ang), sc * np.cos(ang), h / 2 - sc * (np.sin(ang) * dw / 2 + np.cos(ang) * dh / 2)],
                      [pp[0], pp[1], 1.0]])
        gotw = ops.warp_perspective(ctx.to_device(src), M, (dh, dw), iname, bname, cval).get()
        for f in range(n):
            want = oracle.warp_perspective(src[f], M, (dh, dw), INTERPS[iname], BORDERS[bname], cval)
            if dt == np.float32:
                ok = np.isnan(gotw[f]) == np.isnan(want)
                d = np.abs(np.nan_to_num(gotw[f]) - np.nan_to_num(want)).max() if ok.all() else np.inf
                bad = d > 1e-5 * max(1.0, float(np.abs(np.nan_to_num(want)).max()))
            else:
                bad = not np.array_equal(gotw[f], want)
            if bad:
                fails += 1
                df = np.abs(gotw[f].astype(np.float64) - want.astype(np.float64))
                print('MISMATCH warp case %d frame %d: %s %dx%d -> %dx%d n=%d %s %s cval %g: %d values, max |d| %g'
                      % (case, f, np.dtype(dt).name, h, w, dh, dw, n, iname, bname, cval,
                         int((df > 0).sum()), np.nanmax(df)))
                break
        if (case + 1) % 50 == 0:
            print('%d cases, %d mismatches, worst float error %.2e' % (case + 1, fails, worst), flush=True)
    print('done: %d cases, %d mismatches, worst float error %.2e' % (n_cases, fails, worst))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
