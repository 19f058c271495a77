#!/usr/bin/env python3
"""Randomised differential check of the alternative kernels against the plain ones (GPU):
ring remap, ring big, frame-pair kernel, lens map cache - every result must have the bits of
the per-frame / gather kernels.  usage: python tools/fuzz_paths.py [n_cases] [seed] [big]
(big: frames up to 2200 x 3900 instead of 420 x 1300; FUZZ_ONLY=<case> runs one case of the sequence,
FUZZ_DUMP=<file.npz> stores the inputs and both results of the last mismatch)"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def same(a, b):
    if a.shape != b.shape:
        return False
    if a.dtype.kind == 'f':
        av, bv = a.view(np.uint32 if a.itemsize == 4 else np.uint64), b.view(
            np.uint32 if b.itemsize == 4 else np.uint64)
        bad = (av != bv) & ~(np.isnan(a) & np.isnan(b))
        return not bad.any()
    return np.array_equal(a, b)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    hmax, wmax = (2200, 3900) if big else (420, 1300)
    ctx = ia.default_context(0)
    fails = 0
    for case in range(n_cases):
        h = int(rng.integers(40, hmax))
        w = int(rng.integers(70, wmax))
        n = int(rng.integers(1, 5 if big else 9))
        dh = h if rng.random() < 0.7 else int(rng.integers(30, hmax))
        dw = w if rng.random() < 0.7 else int(rng.integers(60, wmax))
        src = rng.random((n, h, w), dtype=np.float32)
        if rng.random() < 0.1:
            src[0, h // 2, w // 3] = np.nan
        d_src = ctx.to_device(src)
        # a random projective map dst -> src: rotation, scale, shear, shift, perspective
        a = np.deg2rad(rng.choice([0, 0, 0, 2, -3, 10, 35, 90, 180]) + rng.normal(0, 0.5))
        sc = rng.choice([1.0, 1.0, 0.9, 1.1, 0.5, 2.0]) * (1 + rng.normal(0, 0.01))
        M = np.array([[sc * np.cos(a), -sc * np.sin(a) + rng.normal(0, 0.01), rng.normal(0, 20)],
                      [sc * np.sin(a), sc * np.cos(a), rng.normal(0, 20)],
                      [rng.normal(0, 2e-5), rng.normal(0, 2e-5), 1.0]])
        c = np.array([[1, 0, -dw / 2], [0, 1, -dh / 2], [0, 0, 1.0]])
        cs = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
        M = cs @ M @ c
        yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
        W = M[2, 0] * xx + M[2, 1] * yy + M[2, 2]
        mx = ((M[0, 0] * xx + M[0, 1] * yy + M[0, 2]) / W).astype(np.float32)
        my = ((M[1, 0] * xx + M[1, 1] * yy + M[1, 2]) / W).astype(np.float32)
        if rng.random() < 0.3:  # a smooth distortion on top
            mx += (3 * np.sin(yy / 37.0)).astype(np.float32)
            my += (2 * np.cos(xx / 53.0)).astype(np.float32)
        if rng.random() < 0.1:
            mx[dh // 3, dw // 4:dw // 4 + 9] = np.nan
        dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
        interp = str(rng.choice(['linear', 'linear_cv_q5', 'cubic', 'cubic_cv_q5', 'lanczos4']))
        border = str(rng.choice(['constant', 'replicate', 'reflect', 'wrap', 'reflect101']))
        cval = float(rng.choice([0.0, 0.25]))
        K = int(rng.choice([3, 5, 7, 9, 11]))
        k = rng.random((K, K))
        k /= k.sum()
        cmode = str(rng.choice(['reflect', 'constant', 'wrap', 'mirror', 'nearest']))
        fx = float(w * rng.uniform(0.8, 1.3))
        Kc = np.array([[fx, 0, (w - 1) / 2.0], [0, fx, (h - 1) / 2.0], [0, 0, 1.0]])
        dist = np.array([rng.normal(0, 0.1), rng.normal(0, 0.02), rng.normal(0, 1e-3),
                         rng.normal(0, 1e-3), 0.0])
        finterp = interp if interp in ('linear', 'linear_cv_q5', 'cubic', 'cubic_cv_q5') else 'linear'
        if os.environ.get('FUZZ_ONLY') and int(os.environ['FUZZ_ONLY']) != case:
            continue
        print('case %d: %dx%d -> %dx%d n=%d angle-ish M=%s' % (case, h, w, dh, dw, n, np.round(M, 3).tolist()), flush=True)
        calls = {
            'remap': lambda: ops.remap(d_src, dmx, dmy, interp, border, cval),
            'warp': lambda: ops.warp_perspective(d_src, M, (dh, dw), interp, border, cval),
            'undistort': (lambda: ops.undistort(d_src, Kc, dist, Kc, interp, border, cval))
            if (dh, dw) == (h, w) else None,
            'remap_conv': lambda: ops.remap_conv2d(d_src, dmx, dmy, k, finterp, border, cval, cmode),
            'warp_conv': lambda: ops.warp_perspective_conv2d(d_src, M, (dh, dw), k, finterp, border,
                                                             cval, cmode),
            'undistort_conv': (lambda: ops.undistort_conv2d(d_src, Kc, dist, Kc, k, finterp, border,
                                                            cval, cmode))
            if (dh, dw) == (h, w) else None,
        }
        plain = dict(ring_remap=0, lens_cache=0, ring_min=1, frames_wg=0, stored_coords=0, pipe=1, tile_warp=0, strip_remap=0)
        alts = [dict(ring_remap=2), dict(lens_cache=1), dict(frames_wg=1),
                dict(ring_remap=2, lens_cache=1, frames_wg=1),
                dict(frames_wg=1, stored_coords=1),     # homography coordinates stored once per batch
                dict(pipe=0),                           # compiler-scheduled loops everywhere
                dict(tile_warp=2),                      # homography warps with the tile's box in LDS
                dict(tile_warp=1, ring_remap=1, stored_coords=4)]   # the default policy
        for name, fn in calls.items():
            if fn is None:
                continue
            old = ctx.set_tuning(**plain)
            try:
                ref = fn().get()
                # (last: the library's OWN defaults for every knob `plain` touched - the combination callers get)
                for alt in alts + [dict(old)]:
                    ctx.set_tuning(**plain)
                    ctx.set_tuning(**alt)
                    for rep in range(2):  # the second call takes the cached plan / hint
                        got = fn().get()
                        if not same(got, ref):
                            fails += 1
                            bad = (got.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(got) & np.isnan(ref))
                            idx = np.argwhere(bad.reshape((-1,) + bad.shape[-2:]))
                            print('   %d values differ; frames %s rows %d..%d cols %d..%d; first %s got %r want %r'
                                  % (bad.sum(), sorted(set(idx[:, 0].tolist())), idx[:, 1].min(), idx[:, 1].max(),
                                     idx[:, 2].min(), idx[:, 2].max(), idx[0].tolist(),
                                     got.reshape((-1,) + got.shape[-2:])[tuple(idx[0])],
                                     ref.reshape((-1,) + ref.shape[-2:])[tuple(idx[0])]))
                            rows = np.bincount(idx[:, 1], minlength=bad.shape[-2])
                            print('   rows with mismatches (row: count): %s' % ', '.join(
                                '%d: %d' % (r, c) for r, c in enumerate(rows) if c)[:600])
                            if os.environ.get('FUZZ_DUMP'):   # the inputs and both results, for a look on the CPU
                                np.savez_compressed(os.environ['FUZZ_DUMP'], src=src, M=M, k=k, mx=mx, my=my, Kc=Kc,
                                                    dist=dist, got=got, ref=ref, cval=cval, what=name, alt=repr(alt),
                                                    interp=finterp, border=border, cmode=cmode)
                            print('MISMATCH case %d %s %r rep %d: %dx%d -> %dx%d n=%d interp=%s '
                                  'border=%s K=%d cmode=%s' % (case, name, alt, rep, h, w, dh, dw,
                                                               n, interp, border, K, cmode))
            finally:
                ctx.set_tuning(**old)
        if (case + 1) % 25 == 0:
            print('%d cases, %d mismatches' % (case + 1, fails), flush=True)
    print('done: %d cases, %d mismatches' % (n_cases, fails))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
