"""How the perspective-warp kernels depend on the in-plane rotation of the homography
(16 x 4K frames; rotation about the frame centre + a mild perspective term): the row-walking kernels
(gather / ring, tile_warp = 0) against the default policy, which takes the tile kernel of
csrc/tile_warp.hpp where it pays.

    python3 tools/angle_sweep.py [angles...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def rot_persp(h, w, deg):
    a = np.deg2rad(deg)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
                  [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy],
                  [0, 0, 1.0]])
    P = np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]])
    return P @ R


def timed(ctx, fn, n=20, warm=30):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


def main():
    angles = [float(a) for a in sys.argv[1:]] or [0, 0.5, 1, 2, 4, 7, 15, 30, 45, 90]
    ctx = ia.default_context(0)
    B, h, w = 16, 2160, 3840
    src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
    dst = ctx.empty((B, h, w), np.float32)
    g9 = ops.gaussian_kernel1d(1.0)
    print('16 x 4K float32, ms per launch: row-walking kernels (tile_warp = 0) / default policy')
    print('%6s %13s %13s %13s %15s %15s' % ('deg', 'linear', 'cubic', 'lanczos4', 'linear+sep9', 'cubic+sep9'))
    for deg in angles:
        M = rot_persp(h, w, deg)
        row = []
        for it in ('linear', 'cubic', 'lanczos4'):
            for tw in (0, 1):
                ctx.set_tuning(tile_warp=tw)
                row.append(timed(ctx, lambda: ops.warp_perspective(src, M, (h, w), it, out=dst)))
        for it in ('linear', 'cubic'):
            for tw in (0, 1):
                ctx.set_tuning(tile_warp=tw)
                row.append(timed(ctx, lambda: ops.warp_perspective_sepconv2d(src, M, (h, w), g9, g9, it, out=dst)))
        print('%6.1f ' % deg + ' '.join('%6.3f/%-6.3f' % (row[2 * i], row[2 * i + 1]) for i in range(3)) + ' ' +
              ' '.join('%7.3f/%-7.3f' % (row[2 * i], row[2 * i + 1]) for i in range(3, 5)), flush=True)
    # the camera's uint16 frames in OpenCV's 16U arithmetic
    u16 = ctx.to_device((src.get() * 65535).astype(np.uint16))
    d16 = ctx.empty((B, h, w), np.uint16)
    print('16 x 4K uint16 (1/32-px coordinates, 16U arithmetic), ms per launch: gather kernel / tile kernel')
    print('%6s %15s %15s' % ('deg', 'cubic_cv_q5', 'lanczos4'))
    for deg in angles:
        M = rot_persp(h, w, deg)
        row = []
        for it in ('cubic_cv_q5', 'lanczos4'):
            for tw in (0, 1):
                ctx.set_tuning(tile_warp=tw)
                row.append(timed(ctx, lambda: ops.warp_perspective(u16, M, (h, w), it, out=d16), n=5, warm=3))
        print('%6.1f ' % deg + ' '.join('%7.3f/%-7.3f' % (row[2 * i], row[2 * i + 1]) for i in range(2)), flush=True)
    ctx.set_tuning(tile_warp=1)


if __name__ == '__main__':
    main()
