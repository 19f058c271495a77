"""The LDS-ring fused kernel (wave_lring.hpp) against the gather kernel, bit for bit, over cases
that exercise clean strips, rim strips, the fallback strips and the plan cache.  GPU box only.

    python tools/lring_check.py [--big]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def lens_maps(ctx, h, w, dist, f=None):
    f = float(w) if f is None else f
    K = np.array([[f, 0, (w - 1) / 2.0], [0, f, (h - 1) / 2.0], [0, 0, 1.0]])
    return K, ops.build_undistort_map(K, np.asarray(dist, float), K, h, w, ctx=ctx, device=True)


def main():
    big = '--big' in sys.argv
    ctx = ia.default_context(0)
    rng = np.random.default_rng(7)
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    k3 = rng.random((3, 3))
    k3 /= k3.sum()
    k5r = rng.random((5, 5)) - 0.3
    bad = 0

    def both(name, fn):
        nonlocal bad
        ctx.set_tuning(lring=0)
        a = fn().get()
        ctx.set_tuning(lring=2)
        b = fn().get()
        b2 = fn().get()   # (a second call: cached plans)
        ctx.set_tuning(lring=1)
        c = fn().get()
        same = np.array_equal(a, b, equal_nan=True) and np.array_equal(a, b2, equal_nan=True) and \
            np.array_equal(a, c, equal_nan=True)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        print('%-64s %s' % (name, 'identical bits' if same else
                            'DIFFERS: %d px, max |d| %.3g, first at %s' %
                            (int((d > 0).sum()), float(np.nanmax(d)), np.argwhere(d > 0)[:3].tolist())))
        sys.stdout.flush()
        bad += 0 if same else 1

    sizes = [(2160, 3840, 8), (1080, 1920, 8), (300, 500, 8), (777, 1028, 12), (64, 260, 8),
             (2160, 3840, 16)] if big else [(1080, 1920, 8), (300, 500, 8), (777, 1028, 12), (64, 260, 8)]
    for (h, w, n) in sizes:
        src = ctx.to_device(rng.random((n, h, w), dtype=np.float32))
        for dist in ([-0.12, 0.03, 1e-3, -5e-4, 0.0], [0.0, 0.01, 0.1, 0.01, 0.001], [0.2, 0.05, 0, 0, 0]):
            K, (dmx, dmy) = lens_maps(ctx, h, w, dist)
            for kern, kn in ((k5, 'k5'), (k3, 'k3'), (k5r, 'k5r')):
                for cm in ('reflect', 'constant', 'wrap', 'nearest', 'mirror'):
                    if not big and (kn, cm) not in (('k5', 'reflect'), ('k3', 'constant'), ('k5r', 'wrap'),
                                                    ('k3', 'nearest'), ('k5', 'mirror')):
                        continue
                    both('maps %dx%dx%d dist %s %s %s' % (n, h, w, dist[:2], kn, cm),
                         lambda: ops.remap_conv2d(src, dmx, dmy, kern, conv_mode=cm))
            both('lens model %dx%dx%d dist %s k5' % (n, h, w, dist[:2]),
                 lambda: ops.undistort_conv2d(src, K, np.asarray(dist, float), K, k5))
        # homography: mild perspective (clean), rotation (mostly fallback)
        from imgprocessor_amd.utils import getPerspectiveTransform
        quad = np.array([(0.05 * w, 0.05 * h), (0.95 * w, 0.025 * h), (0.975 * w, 0.975 * h), (0.025 * w, 0.95 * h)])
        rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
        Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
        both('homography %dx%dx%d k5' % (n, h, w), lambda: ops.warp_perspective_conv2d(src, Hm, (h, w), k5))
        ang = np.deg2rad(7.0)
        R7 = np.array([[np.cos(ang), -np.sin(ang), 0.04 * w], [np.sin(ang), np.cos(ang), -0.05 * h],
                       [4e-6, -2e-6, 1.0]])
        both('rotation 7 deg %dx%dx%d k3' % (n, h, w), lambda: ops.warp_perspective_conv2d(src, R7, (h, w), k3))
        # maps that leave the source (zoom out) and a vertical stretch (two new rows per row)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        zx = ctx.to_device((xx - w / 2) * 1.3 + w / 2)
        zy = ctx.to_device((yy - h / 2) * 1.3 + h / 2)
        both('zoom-out maps %dx%dx%d k5' % (n, h, w), lambda: ops.remap_conv2d(src, zx, zy, k5))
        sx = ctx.to_device(xx * 0.5 + 3.25)
        sy = ctx.to_device(yy * 0.5 + 1.75)
        both('magnifying maps %dx%dx%d k5' % (n, h, w), lambda: ops.remap_conv2d(src, sx, sy, k5))
        # maps rewritten in place between calls (plans of caller-owned maps must not be reused)
        wob = ctx.to_device(xx + 3.0 * np.sin(yy / 37.0).astype(np.float32))
        both('wobble maps %dx%dx%d k5' % (n, h, w), lambda: ops.remap_conv2d(src, wob, zy if False else ctx.to_device(yy), k5))
        del src
    print('lring_check: %d case(s) differ' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
