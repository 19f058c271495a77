#!/usr/bin/env python3
"""Randomised differential check of the one-launch warp + separable filter (csrc/tile_chain.hpp, knob
tile_chain = 1) against the two launches through the workspace (0): random projective maps, sizes,
batches, tap counts, interpolations, border modes of the warp and of the filter, steps and frames
per workgroup - every result must have the same bits.
usage: python tools/fuzz_chain.py [n_cases] [seed] [big]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

BORDERS = ('constant', 'replicate', 'reflect', 'wrap', 'reflect101')
INTERPS = ('linear', 'cubic', 'cubic_cv', 'linear_cv_q5', 'cubic_cv_q5')


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    rng = np.random.default_rng(seed)
    hmax, wmax = (2200, 3900) if big else (420, 900)
    ctx = ia.default_context(0)
    fails = taken = 0
    for case in range(n_cases):
        h, w = int(rng.integers(8, hmax)), int(rng.integers(8, wmax))
        n = int(rng.integers(1, 4 if big else 12))
        dh = h if rng.random() < 0.6 else int(rng.integers(1, hmax))
        dw = w if rng.random() < 0.6 else int(rng.integers(1, wmax))
        src = rng.random((n, h, w), dtype=np.float32)
        if rng.random() < 0.1:
            src[0, h // 2, w // 3] = np.nan
        a = np.deg2rad(rng.choice([0, 0, 2, -3, 10, 35, 90, 180, 217]) + rng.normal(0, 0.5))
        sc = rng.choice([1.0, 1.0, 0.9, 1.1, 0.6, 1.7]) * (1 + rng.normal(0, 0.01))
        M = np.array([[sc * np.cos(a), -sc * np.sin(a) + rng.normal(0, 0.01), rng.normal(0, 20)],
                      [sc * np.sin(a), sc * np.cos(a), rng.normal(0, 20)],
                      [rng.normal(0, 2e-5), rng.normal(0, 2e-5), 1.0]])
        M = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]]) @ M @ np.array([[1, 0, -dw / 2], [0, 1, -dh / 2], [0, 0, 1.0]])
        K = int(rng.choice([3, 5, 7, 9]))
        ky, kx = rng.random(K) + 0.05, rng.random(K) + 0.05
        ky, kx = ky / ky.sum(), kx / kx.sum()
        interp = str(rng.choice(INTERPS))
        border, conv = str(rng.choice(BORDERS)), str(rng.choice(BORDERS))
        cval = float(rng.choice([0.0, 0.5, np.nan])) if border == 'constant' else 0.0
        knobs = dict(chain_steps=int(rng.choice([0, 0, 1, 2, 3, 5, 9])), chain_frames=int(rng.choice([0, 0, 1, 2, 5, 8])))
        d = ctx.to_device(src)
        out = []
        try:
            for tc in (0, 1):
                ctx.set_tuning(tile_chain=2 * tc, **knobs)
                before = ctx.get_tuning('chain_launches')
                out.append(ops.warp_perspective_sepconv2d(d, M, (dh, dw), ky, kx, interp, border, cval, conv).get())
                taken += ctx.get_tuning('chain_launches') - before
        finally:
            ctx.set_tuning(tile_chain=0, chain_steps=0, chain_frames=0)
        a0, a1 = out
        bad = (a0.view(np.uint32) != a1.view(np.uint32)) & ~(np.isnan(a0) & np.isnan(a1))
        if bad.any():
            fails += 1
            print('MISMATCH case %d: %dx%d -> %dx%d x %d, %s, K %d, warp %s, filter %s, %s: %d values, first at %s'
                  % (case, h, w, dh, dw, n, interp, K, border, conv, knobs, bad.sum(), np.argwhere(bad)[0]), flush=True)
    print('fuzz_chain: %d cases (%d on the chain kernel), %d mismatches' % (n_cases, taken, fails))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
