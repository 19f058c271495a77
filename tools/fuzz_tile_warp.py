"""Random homography warps on the tile kernel (csrc/tile_warp.hpp, knob tile_warp = 2) against the
gather / ring kernels (tile_warp = 0), bit for bit: sizes 1 .. 900, batches 1 .. 20, any rotation,
zooms 0.3 .. 2.5, perspective up to a horizon inside the picture, shifts beyond the source, every
interpolation and border mode, float32 and uint16 frames (OpenCV's 16U arithmetic).

    python3 tools/fuzz_tile_warp.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = ia.default_context(0)
    fails = 0
    for case in range(n_cases):
        h = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 65, int(rng.integers(3, 900))]))
        w = int(rng.choice([1, 3, 63, 64, 65, 128, 129, int(rng.integers(3, 900))]))
        n = int(rng.choice([1, 1, 2, 3, 4, 7, 8, 9, 16, 20]))
        if h * w * n > 6e6:
            n = max(1, int(6e6 // (h * w)))
        dh = h if rng.random() < 0.6 else int(rng.integers(1, 900))
        dw = w if rng.random() < 0.6 else int(rng.integers(1, 900))
        a = np.deg2rad(rng.choice([0, 0, 1, -2, 5, 17, 45, 90, 135, 180, 270, rng.uniform(0, 360)]) + rng.normal(0, 0.3))
        sc = float(rng.choice([1.0, 1.0, 1.0, 0.9, 1.1, 0.3, 0.5, 0.7, 1.4, 1.9, 2.05, 2.5])) * (1 + rng.normal(0, 0.01))
        pp = rng.normal(0, 1.0, 2) * float(rng.choice([0, 1e-6, 1e-5, 1e-4, 1e-3, 3e-3]))
        sh = rng.normal(0, 1.0, 2) * float(rng.choice([0, 5, 40, 400, 5000]))
        M = np.array([[sc * np.cos(a), -sc * np.sin(a) + rng.normal(0, 0.01), sh[0]],
                      [sc * np.sin(a), sc * np.cos(a), sh[1]],
                      [pp[0], pp[1], 1.0]])
        c = np.array([[1, 0, -dw / 2], [0, 1, -dh / 2], [0, 0, 1.0]])
        cs = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
        M = cs @ M @ c
        u16 = rng.random() < 0.3
        if u16:
            src = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
            interp = str(rng.choice(['cubic_cv_q5', 'lanczos4']))
            cval = float(rng.choice([0, 1000, 65535, 70000, -5]))
        else:
            src = rng.random((n, h, w), dtype=np.float32)
            if rng.random() < 0.1:
                src[0, h // 2, w // 3] = np.nan
            interp = str(rng.choice(['linear', 'linear_cv_q5', 'cubic', 'cubic_cv', 'cubic_cv_q5', 'lanczos4']))
            cval = float(rng.choice([0.0, 0.25, -1.5]))
        border = str(rng.choice(['constant', 'constant', 'replicate', 'reflect', 'wrap', 'reflect101']))
        d = ctx.to_device(src)
        res = []
        for tw in (0, 2):
            ctx.set_tuning(tile_warp=tw)
            for rep in range(2 if tw == 2 else 1):   # the second call takes the cached box
                res.append(ops.warp_perspective(d, M, (dh, dw), interp, border, border_value=cval).get())
        ctx.set_tuning(tile_warp=1)
        for k in (1, 2):
            x, y = res[0], res[k]
            if u16:
                same = np.array_equal(x, y)
            else:
                same = np.array_equal(x.view(np.uint32)[~(np.isnan(x) & np.isnan(y))],
                                      y.view(np.uint32)[~(np.isnan(x) & np.isnan(y))])
            if not same:
                fails += 1
                bad = x != y
                if not u16:
                    bad &= ~(np.isnan(x) & np.isnan(y))
                idx = np.argwhere(bad)
                print('MISMATCH case %d call %d: %s %dx%d -> %dx%d n=%d %s %s cval %g: %d values, first %s got %r want %r\n   M=%r'
                      % (case, k, src.dtype.name, h, w, dh, dw, n, interp, border, cval, bad.sum(), idx[0].tolist(),
                         y[tuple(idx[0])], x[tuple(idx[0])], M.tolist()), flush=True)
                break
        if (case + 1) % 100 == 0:
            print('%d cases, %d mismatches' % (case + 1, fails), flush=True)
    print('done: %d cases, %d mismatches' % (n_cases, fails))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
