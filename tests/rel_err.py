"""Measured float32 error margins of the BASELINE configurations against the oracle (double
accumulation), GPU box: max |gpu - oracle| / max|oracle| (the form the 1e-5 tolerance of the
parity tests takes: rtol 1e-5 + atol 1e-5 * max|oracle|) and the max pointwise relative error
over pixels with |oracle| >= 1e-3 * max|oracle|.

    python tests/rel_err.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.utils import getPerspectiveTransform  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.conftest import synth  # noqa: E402


def report(name, got, want):
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    m = np.abs(want).max()
    err = np.abs(got - want)
    big = np.abs(want) >= 1e-3 * m
    print('%-58s max|d|/max|ref| %.2e   max pointwise rel %.2e' %
          (name, err.max() / m, (err[big] / np.abs(want[big])).max()), flush=True)


def main():
    orc.build()
    orc.set_threads(min(32, orc.max_threads(), len(os.sched_getaffinity(0))))
    ctx = ia.default_context(0)
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)

    def cam(h, w):
        return (np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]]),
                np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0]))

    # C1 512x512 box3
    img = synth((512, 512), 0)
    report('C1 512^2 box3 (maskedConvolve == filter)', ops.conv2d(img, np.ones((3, 3)) / 9),
           orc.conv2d(img, np.ones((3, 3)) / 9))
    # C2 / headline: undistort + 5x5
    for h, w, name in ((1080, 1920, 'C2 1080p undistort + 5x5 Gaussian'),
                       (2160, 3840, 'headline 4K undistort + 5x5 Gaussian')):
        K, d = cam(h, w)
        img = synth((h, w), 1)
        mx, my = orc.build_undistort_map(K, d, K, h, w)
        got = ops.remap_conv2d(ctx.to_device(img), ctx.to_device(mx), ctx.to_device(my), k5).get()
        report(name, got, orc.conv2d(orc.remap(img, mx, my), k5))
    # C3 perspective + separable 9+9
    h, w = 2160, 3840
    quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
    rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
    Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
    g9 = ops.gaussian_kernel1d(1.0)
    img = synth((h, w), 2)
    for interp, oi in (('linear', orc.LINEAR), ('cubic', orc.CUBIC_KEYS), ('lanczos4', orc.LANCZOS4)):
        got = ops.warp_perspective_sepconv2d(ctx.to_device(img), Hm, (h, w), g9, g9, interp).get()
        want = orc.sepconv2d(orc.warp_perspective(img, Hm, (h, w), oi), g9, g9)
        report('C3 4K perspective (%s) + separable 9+9' % interp, got, want)
    # C4 uint16 -> float32 + 7x7
    K, d = cam(h, w)
    k7 = np.random.default_rng(123).random((7, 7))
    k7 /= k7.sum()
    u16 = np.round(synth((h, w), 3, np.float64) * 4095).astype(np.uint16)
    mx, my = orc.build_undistort_map(K, d, K, h, w)
    got = ops.remap_conv2d(ctx.to_device(u16), ctx.to_device(mx), ctx.to_device(my), k7).get()
    report('C4 4K uint16 -> float32 undistort + dense 7x7', got,
           orc.conv2d(orc.remap(u16, mx, my, out_dtype=np.float32), k7))
    # C5 8K bicubic + 11x11 (normalised random kernel, all positive)
    h, w = 4320, 7680
    a = np.deg2rad(7.0)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
                  [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy], [0, 0, 1.0]])
    M = np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]]) @ R
    k11 = np.random.default_rng(321).random((11, 11))
    k11 /= k11.sum()
    img = synth((h, w), 4)
    got = ops.warp_perspective_conv2d(ctx.to_device(img), M, (h, w), k11, 'cubic').get()
    report('C5 8K bicubic warp + dense 11x11', got,
           orc.conv2d(orc.warp_perspective(img, M, (h, w), orc.CUBIC_KEYS), k11))
    # mixed-sign 11x11 (the case where the absolute term of the tolerance matters)
    ks = np.random.default_rng(9).standard_normal((11, 11))
    img = synth((1080, 1920), 5)
    report('1080p dense 11x11, mixed-sign kernel', ops.conv2d(img, ks), orc.conv2d(img, ks))


if __name__ == '__main__':
    main()
