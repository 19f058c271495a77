"""In-process A/B of context knobs on the 4K headline (map-based undistort + 5x5), the plain 5x5
filter and the standalone remap: every knob set is timed in turn, several rounds, so that the
box's clock state is shared.  GPU box only.

    python tools/ab_knobs.py [--batch 64] [--rounds 3] name:knob=v,knob=v ...
e.g. python tools/ab_knobs.py base: inner0:frames_inner=0 wg0:frames_wg=0 sh48:strip_h=48
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def timeit(ctx, fn, n, warm):
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
    args = sys.argv[1:]
    batch, rounds, sets, what = 64, 3, [], ['fused5', 'conv5']
    while args:
        a = args.pop(0)
        if a == '--batch':
            batch = int(args.pop(0))
        elif a == '--rounds':
            rounds = int(args.pop(0))
        elif a == '--what':
            what = args.pop(0).split(',')
        else:
            name, _, kv = a.partition(':')
            sets.append((name, {k: int(v) for k, v in (x.split('=') for x in kv.split(',') if x)}))
    sets = sets or [('base', {})]
    ctx = ia.default_context(0)
    h, w = 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    rng = np.random.default_rng(0)
    one = rng.random((16, h, w), dtype=np.float32)
    src = ctx.to_device(np.concatenate([one] * (batch // 16)) if batch >= 16 else one[:batch])
    dst = ctx.empty((batch, h, w), np.float32)
    k7 = rng.random((7, 7))
    k7 /= k7.sum()
    u16 = None
    if {'c4', 'c4g7', 'c4g5', 'c4sep9', 'c4sep5', 'c3u16'} & set(what):
        u16 = ctx.to_device(np.round(np.concatenate([one] * (batch // 16)) * 4095).astype(np.uint16)
                            if batch >= 16 else np.round(one[:batch] * 4095).astype(np.uint16))
    Hm = None
    if {'c3', 'c3u16', 'warp5', 'lz4', 'cubic', 'warplin'} & set(what):
        from imgprocessor_amd.utils import getPerspectiveTransform
        quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
        rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
        Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
    g9 = ops.gaussian_kernel1d(1.0)
    k11 = np.random.default_rng(321).random((11, 11))
    k11 /= k11.sum()
    ang = np.deg2rad(7.0)   # C5's warp: rotation + mild perspective
    R7 = np.array([[np.cos(ang), -np.sin(ang), 150.0], [np.sin(ang), np.cos(ang), -100.0],
                   [4e-6, -2e-6, 1.0]])
    r7x = r7y = None
    if 'rot7cubicmap' in what or 'rot7lz4map' in what or 'rot7linearmap' in what:
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        W = R7[2, 0] * xx + R7[2, 1] * yy + R7[2, 2]
        r7x = ctx.to_device(((R7[0, 0] * xx + R7[0, 1] * yy + R7[0, 2]) / W).astype(np.float32))
        r7y = ctx.to_device(((R7[1, 0] * xx + R7[1, 1] * yy + R7[1, 2]) / W).astype(np.float32))
        del yy, xx, W
    gk, gk1 = {}, {}
    for kk in (3, 5, 7, 9):
        t = np.exp(-0.5 * (np.arange(kk) - kk // 2) ** 2)
        t /= t.sum()
        gk[kk] = np.outer(t, t)
        gk1[kk] = t
    calls = {
        'c3': lambda: ops.warp_perspective_sepconv2d(src, Hm, (h, w), g9, g9, 'linear', out=dst),
        'warp5': lambda: ops.warp_perspective_conv2d(src, Hm, (h, w), k5, 'linear', out=dst),
        'sep9map': lambda: ops.remap_sepconv2d(src, dmx, dmy, g9, g9, out=dst),
        'c4': lambda: ops.remap_conv2d(u16, dmx, dmy, k7, out=dst),
        # uint16 frames + Gaussian outer products / separable taps (knob sep_u16: one kernel against two launches / dense)
        'c4g7': lambda: ops.remap_conv2d(u16, dmx, dmy, gk[7], out=dst),
        'c4g5': lambda: ops.remap_conv2d(u16, dmx, dmy, gk[5], out=dst),
        'c3u16': lambda: ops.warp_perspective_sepconv2d(u16, Hm, (h, w), g9, g9, 'linear', out=dst),
        'c4sep9': lambda: ops.remap_sepconv2d(u16, dmx, dmy, g9, g9, out=dst),
        'c4sep5': lambda: ops.remap_sepconv2d(u16, dmx, dmy, gk1[5], gk1[5], out=dst),
        'fused7': lambda: ops.remap_conv2d(src, dmx, dmy, k7, out=dst),
        'fused5': lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst),
        'conv5': lambda: ops.conv2d(src, k5, out=dst),
        # Gaussian outer products of 3 / 7 / 9 taps (knob rank1_sep: dense loop against the K + K loop)
        'gconv3': lambda: ops.conv2d(src, gk[3], out=dst),
        'gconv7': lambda: ops.conv2d(src, gk[7], out=dst),
        'gconv9': lambda: ops.conv2d(src, gk[9], out=dst),
        'fusedg3': lambda: ops.remap_conv2d(src, dmx, dmy, gk[3], out=dst),
        'fusedg7': lambda: ops.remap_conv2d(src, dmx, dmy, gk[7], out=dst),
        'fusedg9': lambda: ops.remap_conv2d(src, dmx, dmy, gk[9], out=dst),
        'gsep3': lambda: ops.sepconv2d(src, gk1[3], gk1[3], out=dst),
        'gsep5': lambda: ops.sepconv2d(src, gk1[5], gk1[5], out=dst),
        'gsep7': lambda: ops.sepconv2d(src, gk1[7], gk1[7], out=dst),
        'gsep9': lambda: ops.sepconv2d(src, gk1[9], gk1[9], out=dst),
        'undist5': lambda: ops.undistort_conv2d(src, K, dist, K, k5, out=dst),
        'conv11': lambda: ops.conv2d(src, k11, out=dst),
        'lz4': lambda: ops.warp_perspective(src, Hm, (h, w), 'lanczos4', out=dst),
        'cubic': lambda: ops.warp_perspective(src, Hm, (h, w), 'cubic', out=dst),
        'rot7cubic': lambda: ops.warp_perspective(src, R7, (h, w), 'cubic', out=dst),
        'rot7lz4': lambda: ops.warp_perspective(src, R7, (h, w), 'lanczos4', out=dst),
        'c5': lambda: ops.warp_perspective_conv2d(src, R7, (h, w), k11, 'cubic', out=dst),
        'rot7cubicmap': lambda: ops.remap(src, r7x, r7y, 'cubic', out=dst),
        'rot7lz4map': lambda: ops.remap(src, r7x, r7y, 'lanczos4', out=dst),
        'rot7linear': lambda: ops.warp_perspective(src, R7, (h, w), 'linear', out=dst),
        'rot7linearmap': lambda: ops.remap(src, r7x, r7y, 'linear', out=dst),
        'remap': lambda: ops.remap(src, dmx, dmy, out=dst),
        'remapcubic': lambda: ops.remap(src, dmx, dmy, 'cubic', out=dst),
        'remaplz4': lambda: ops.remap(src, dmx, dmy, 'lanczos4', out=dst),
        'undist': lambda: ops.undistort(src, K, dist, K, out=dst),
        'undistcubic': lambda: ops.undistort(src, K, dist, K, 'cubic', out=dst),
        'warplin': lambda: ops.warp_perspective(src, Hm, (h, w), 'linear', out=dst),
        'copy': lambda: dst.copy_from(src),
    }
    n = max(10, 1600 // batch)
    base = {k: ctx.get_tuning(k) for _, kn in sets for k in kn}
    res = {(name, c): [] for name, _ in sets for c in what}
    for r in range(rounds):
        for name, knobs in sets:
            ctx.set_tuning(**base)
            ctx.set_tuning(**knobs)
            for c in what:
                res[(name, c)].append(timeit(ctx, calls[c], n, n // 3))
    ctx.set_tuning(**base)
    print('batch %d x 4K float32, ms per launch, %d rounds alternated in one process' % (batch, rounds))
    for c in what:
        for name, knobs in sets:
            v = res[(name, c)]
            print('%-8s %-10s %-40s %s   min %.4f' % (c, name, knobs, '  '.join('%.4f' % x for x in v),
                                                      min(v)))
    # the results of every knob set against the first one, bit for bit (last frame of the batch)
    for c in what:
        if c == 'copy':
            continue
        ref = None
        for name, knobs in sets:
            ctx.set_tuning(**base)
            ctx.set_tuning(**knobs)
            calls[c]()
            got = dst.frame(batch - 1).get() if batch > 1 else dst.get()
            if ref is None:
                ref = got
            same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
            print('%-8s %-10s %s' % (c, name, 'identical bits' if same else 'DIFFERS from %s (max |d| %.3g, %d px)'
                                     % (sets[0][0], np.nanmax(np.abs(got - ref)), int((got != ref).sum()))))
    ctx.set_tuning(**base)


if __name__ == '__main__':
    main()
