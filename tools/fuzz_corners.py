#!/usr/bin/env python3
"""Randomised check AROUND THE SOURCE'S CORNERS against the CPU oracle (GPU box).  The general fuzzers
(tests/fuzz_oracle.py, tools/fuzz_paths.py) place a source corner inside the output picture now and then; the
footprints on a corner are where the border rules of every kernel meet (round 6: pixel (0, 0) lost on the
hand-scheduled loops, one or two samples per frame - found after ~2000 general cases).  Here EVERY case has all
four corners of the source inside the output picture, at a random sub-pixel position, lane and strip: a shift (+ a
small rotation / zoom) of a source smaller than the output; standalone remap and warp for every interpolation,
border mode and element type, the fused chains (dense K x K, separable K + K) where they are built; batches of
1 .. 12 frames so that the per-frame, the shared-footprint, the ring and the tile kernels all take cases.

    python tools/fuzz_corners.py [n_cases] [seed]      (FUZZ_ONLY=<case>: one case of the sequence; FUZZ_DUMP=<file.npz>)
float32 within 1e-5 of the data range, integer results bit for bit."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.utils.geometry import getOptimalNewCameraMatrix  # noqa: E402
from oracle import oracle  # noqa: E402

INTERPS = {'nearest': oracle.NEAREST, 'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS,
           'cubic_cv': oracle.CUBIC_CV, 'linear_cv_q5': oracle.LINEAR | oracle.Q5,
           'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5, 'lanczos4': oracle.LANCZOS4}
BORDERS = {'constant': oracle.CONSTANT, 'replicate': oracle.REPLICATE, 'reflect': oracle.REFLECT,
           'wrap': oracle.WRAP, 'reflect101': oracle.REFLECT101}


def differs(got, want, is_float):
    if is_float:
        if not np.array_equal(np.isnan(got), np.isnan(want)):
            return np.inf
        d = float(np.abs(np.nan_to_num(got.astype(np.float64)) - np.nan_to_num(want.astype(np.float64))).max())
        return d if d > 1e-5 * max(1.0, float(np.abs(np.nan_to_num(want)).max())) else 0.0
    return 0.0 if np.array_equal(got, want) else float(np.abs(got.astype(np.int64) - want.astype(np.int64)).max())


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = ia.default_context(0)
    fails = 0
    for case in range(n_cases):
        h, w = int(rng.integers(8, 120)), int(rng.integers(8, 700))
        # the output picture: larger than the source on every side by 2 .. 40 px
        ox, oy = float(rng.uniform(2, 40)), float(rng.uniform(2, 40))
        dh, dw = h + int(oy) + int(rng.integers(3, 40)), w + int(ox) + int(rng.integers(3, 300))
        n = int(rng.choice([1, 2, 3, 4, 4, 8, 8, 12]))
        dt = [np.float32, np.float32, np.uint16, np.uint8][int(rng.integers(0, 4))]
        a = rng.random((n, h, w))
        src = a.astype(np.float32) if dt == np.float32 else np.round(a * (255 if dt == np.uint8 else 4095)).astype(dt)
        for f in range(n):   # corner pixels that cannot be mistaken for their neighbours
            src[f, 0, 0] = src[f, 0, -1] = src[f, -1, 0] = src[f, -1, -1] = src.max()
        ang = np.deg2rad(float(rng.choice([0, 0, 0, 1, -2, 5])) * rng.random())
        sc = float(rng.choice([1.0, 1.0, 1.0, 0.97, 1.04]))
        M = np.array([[sc * np.cos(ang), -sc * np.sin(ang), -ox],
                      [sc * np.sin(ang), sc * np.cos(ang), -oy], [0.0, 0.0, 1.0]])
        if rng.random() < 0.3:
            M[2, :2] = rng.normal(0, 2e-5, 2)
        yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
        W = M[2, 0] * xx + M[2, 1] * yy + M[2, 2]
        mx = ((M[0, 0] * xx + M[0, 1] * yy + M[0, 2]) / W).astype(np.float32)
        my = ((M[1, 0] * xx + M[1, 1] * yy + M[1, 2]) / W).astype(np.float32)
        iname = str(rng.choice(['linear', 'linear', 'linear', 'nearest', 'cubic', 'cubic_cv', 'linear_cv_q5',
                                'cubic_cv_q5', 'lanczos4']))
        bname = str(rng.choice(['constant', 'constant', 'constant', 'replicate', 'reflect', 'wrap', 'reflect101']))
        cval = float(rng.choice([0.0, 0.25])) if dt == np.float32 else float(rng.choice([0, 17]))
        K = int(rng.choice([3, 5, 7, 9, 11]))
        kern = rng.random((K, K))
        if case % 3 == 0:
            kern = np.outer(rng.random(K) + 0.1, rng.random(K) + 0.1)
        kern /= kern.sum()
        g = rng.random(K) + 0.1
        g /= g.sum()
        cmode = str(rng.choice(['reflect', 'constant', 'wrap', 'mirror', 'nearest']))
        k1 = float(rng.uniform(-0.25, 0.1))
        if os.environ.get('FUZZ_ONLY') and int(os.environ['FUZZ_ONLY']) != case:
            continue
        lens = None
        if h >= 24 and w >= 24:
            fxy = float(max(h, w))
            Kc = np.array([[fxy, 0, (w - 1) / 2.0], [0, fxy, (h - 1) / 2.0], [0, 0, 1.0]])
            dist = np.array([k1, 0.02, 1e-3, -5e-4, 0.0])
            newK, _ = getOptimalNewCameraMatrix(Kc, dist, (w, h), 1)
            lmx, lmy = oracle.build_undistort_map(Kc, dist, newK, h, w)
            lens = (Kc, dist, newK, lmx, lmy)
        what = '%s %dx%d -> %dx%d n=%d %s %s cval %g, corner at (%.2f, %.2f), K=%d %s' % (
            np.dtype(dt).name, h, w, dh, dw, n, iname, bname, cval, ox, oy, K, cmode)
        d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
        isf = dt == np.float32
        runs = [('remap', lambda: ops.remap(d_src, dmx, dmy, iname, bname, cval),
                 lambda f: oracle.remap(src[f], mx, my, INTERPS[iname], BORDERS[bname], cval), isf),
                ('warp', lambda: ops.warp_perspective(d_src, M, (dh, dw), iname, bname, cval),
                 lambda f: oracle.warp_perspective(src[f], M, (dh, dw), INTERPS[iname], BORDERS[bname], cval), isf)]
        if dt != np.float32:
            # integer frames INTO float32 (transformations.toFloatArray's ingest; uint16 + bilinear: the strip remap, round 6)
            runs += [('remap -> f32', lambda: ops.remap(d_src, dmx, dmy, iname, bname, cval, out_dtype=np.float32),
                      lambda f: oracle.remap(src[f], mx, my, INTERPS[iname], BORDERS[bname], cval, out_dtype=np.float32), True),
                     ('warp -> f32', lambda: ops.warp_perspective(d_src, M, (dh, dw), iname, bname, cval, out_dtype=np.float32),
                      lambda f: oracle.warp_perspective(src[f], M, (dh, dw), INTERPS[iname], BORDERS[bname], cval,
                                                        out_dtype=np.float32), True)]
        if dt in (np.float32, np.uint16) and iname != 'nearest' and K < min(dh, dw):
            mid_m = lambda f: oracle.remap(src[f], mx, my, INTERPS[iname], BORDERS[bname], cval, out_dtype=np.float32)  # noqa: E731
            mid_w = lambda f: oracle.warp_perspective(src[f], M, (dh, dw), INTERPS[iname], BORDERS[bname], cval,  # noqa: E731
                                                      out_dtype=np.float32)
            runs += [('remap + KxK', lambda: ops.remap_conv2d(d_src, dmx, dmy, kern, iname, bname, cval, cmode),
                      lambda f: oracle.conv2d(mid_m(f), kern, cmode), True),
                     ('warp + KxK', lambda: ops.warp_perspective_conv2d(d_src, M, (dh, dw), kern, iname, bname, cval, cmode),
                      lambda f: oracle.conv2d(mid_w(f), kern, cmode), True)]
            if K <= 9:
                runs += [('remap + K+K', lambda: ops.remap_sepconv2d(d_src, dmx, dmy, g, g[::-1].copy(), iname, bname, cval, cmode),
                          lambda f: oracle.sepconv2d(mid_m(f), g, g[::-1].copy(), cmode), True),
                         ('warp + K+K', lambda: ops.warp_perspective_sepconv2d(d_src, M, (dh, dw), g, g[::-1].copy(), iname,
                                                                               bname, cval, cmode),
                          lambda f: oracle.sepconv2d(mid_w(f), g, g[::-1].copy(), cmode), True)]
        # the lens model with the reference's own camera matrix (getOptimalNewCameraMatrix, alpha = 1: every source
        # pixel kept - all four corners inside the undistorted picture), coordinates evaluated in the kernels
        if lens is not None:
            Kc, dist, newK, lmx, lmy = lens
            runs.append(('undistort', lambda: ops.undistort(d_src, Kc, dist, newK, iname, bname, cval),
                         lambda f: oracle.remap(src[f], lmx, lmy, INTERPS[iname], BORDERS[bname], cval), isf))
            if dt != np.float32:
                runs.append(('undistort -> f32', lambda: ops.undistort(d_src, Kc, dist, newK, iname, bname, cval, out_dtype=np.float32),
                             lambda f: oracle.remap(src[f], lmx, lmy, INTERPS[iname], BORDERS[bname], cval, out_dtype=np.float32), True))
            if dt in (np.float32, np.uint16) and iname != 'nearest' and K < min(h, w):
                mid_l = lambda f: oracle.remap(src[f], lmx, lmy, INTERPS[iname], BORDERS[bname], cval, out_dtype=np.float32)  # noqa: E731
                runs.append(('undistort + KxK', lambda: ops.undistort_conv2d(d_src, Kc, dist, newK, kern, iname, bname, cval, cmode),
                             lambda f: oracle.conv2d(mid_l(f), kern, cmode), True))
                if K <= 9:
                    runs.append(('undistort + K+K', lambda: ops.undistort_sepconv2d(d_src, Kc, dist, newK, g, g[::-1].copy(), iname,
                                                                                     bname, cval, cmode),
                                 lambda f: oracle.sepconv2d(mid_l(f), g, g[::-1].copy(), cmode), True))
        for name, fn, ref, is_float in runs:
            try:
                got = fn().get()
            except NotImplementedError:
                continue          # (a chain the C ABI does not build, e.g. uint16 frames + homography)
            for f in range(n):
                want = ref(f)
                d = differs(got[f], want, is_float)
                if d:
                    fails += 1
                    idx = np.argwhere(np.nan_to_num(got[f].astype(np.float64)) != np.nan_to_num(want.astype(np.float64)))
                    if os.environ.get('FUZZ_DUMP'):
                        np.savez_compressed(os.environ['FUZZ_DUMP'], src=src, M=M, mx=mx, my=my, kern=kern, g=g, got=got,
                                            want=want, frame=f, what=what, name=name)
                    print('MISMATCH case %d %s frame %d: %s: max |d| %g, %d values differ, rows %d..%d cols %d..%d' % (
                        case, name, f, what, d, len(idx), idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max()),
                        flush=True)
                    break
        if (case + 1) % 50 == 0:
            print('%d cases, %d mismatches' % (case + 1, fails), flush=True)
    print('done: %d cases, %d mismatches' % (n_cases, fails))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
