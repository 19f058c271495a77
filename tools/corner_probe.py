#!/usr/bin/env python3
"""What makes the top-left-corner footprint of tools/fuzz_paths.py seed 63 case 66 come out wrong on the shared-footprint
loop (round 6)?  Variations of that case, shared loop (frames_wg = 1) against the per-frame kernels (frames_wg = 0):
    IMGPROC_HIP_LIB=.../libimgproc_hip_r5corner.so python tools/corner_probe.py [dump.npz]
(the r5corner build keeps round 5's rule: make VARIANT=r5corner DEFS=-DIPA_DEBUG_CORNER_AS_ROUND5 ONLY="fused_k3 ...")"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd import _lib  # noqa: E402

if len(sys.argv) > 1:          # the fuzzer's own dump (FUZZ_DUMP=... of a run that still mismatches)
    d = np.load(sys.argv[1])
    src, M, k7 = d['src'], d['M'], d['k']
    cval = float(d['cval'])
else:                          # the case as it was drawn: geometry and matrix of seed 63 / case 66, fresh random frames
    rng0 = np.random.default_rng(66)
    src = rng0.random((4, 143, 1052), dtype=np.float32)
    M = np.array([[1.005, -0.037, -1.459], [0.04, 1.006, -170.734], [0.0, 0.0, 0.997]])
    k7 = rng0.random((7, 7))
    k7 /= k7.sum()
    cval = 0.25
ctx = ia.default_context(0)
print('library:', _lib.LIB_PATH)
plain = dict(ring_remap=0, lens_cache=0, ring_min=1, frames_wg=0, stored_coords=0, pipe=1, tile_warp=0)


def run(name, src, M, shape, k, cv=cval, maps=False):
    dsrc = ctx.to_device(src)
    if maps:
        dh, dw = shape
        yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
        W = M[2, 0] * xx + M[2, 1] * yy + M[2, 2]
        mx = ((M[0, 0] * xx + M[0, 1] * yy + M[0, 2]) / W).astype(np.float32)
        my = ((M[1, 0] * xx + M[1, 1] * yy + M[1, 2]) / W).astype(np.float32)
        dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
        fn = lambda: ops.remap_conv2d(dsrc, dmx, dmy, k, 'linear', 'constant', cv, 'reflect')   # noqa: E731
    else:
        fn = lambda: ops.warp_perspective_conv2d(dsrc, M, shape, k, 'linear', 'constant', cv, 'reflect')   # noqa: E731
    old = ctx.set_tuning(**plain)
    try:
        ref = fn().get()
        ctx.set_tuning(frames_wg=1)
        got = fn().get()
    finally:
        ctx.set_tuning(**old)
    bad = np.argwhere(got != ref)
    if len(bad):
        print('%-46s WRONG  %4d values, rows %d..%d cols %d..%d, max %.3g' % (
            name, len(bad), bad[:, 1].min(), bad[:, 1].max(), bad[:, 2].min(), bad[:, 2].max(), np.abs(got - ref).max()))
    else:
        print('%-46s same bits' % name)


def kern(K):
    k = np.random.default_rng(5).random((K, K))
    return k / k.sum()


h, w = src.shape[1:]
run('case 66 as drawn', src, M, (416, w), k7)
run('... as a map pair', src, M, (416, w), k7, maps=True)
for K in (3, 5):
    run('... %dx%d' % (K, K), src, M, (416, w), kern(K))
    run('... %dx%d as a map pair' % (K, K), src, M, (416, w), kern(K), maps=True)
run('... border value 0', src, M, (416, w), k7, cv=0.0)
run('... 8 frames', np.concatenate([src, src]), M, (416, w), k7)
run('... 300 output rows', src, M, (300, w), k7)
run('... 200 output rows', src, M, (200, w), k7)
run('... 500 output columns', src, M, (416, 500), k7)
M2 = M.copy(); M2[2, :2] = 0; M2[2, 2] = 1
run('... affine (no perspective row)', src, M2 / 1.0, (416, w), k7)
M3 = M2.copy(); M3[0, 1] = 0; M3[1, 0] = 0
run('... no rotation: scale + shift', src, M3, (416, w), k7)
M4 = np.array([[1.0, 0, M3[0, 2]], [0, 1.0, M3[1, 2]], [0, 0, 1.0]])
run('... pure shift (%.3f, %.3f)' % (M4[0, 2], M4[1, 2]), src, M4, (416, w), k7)
M5 = np.array([[1.0, 0, -7.3], [0, 1.0, -5.4], [0, 0, 1.0]])
run('... pure shift (-7.3, -5.4)', src, M5, (416, w), k7)
run('... pure shift (-7.3, -5.4), 5x5', src, M5, (416, w), kern(5))
run('... pure shift (-7.3, -170.4)', src, np.array([[1.0, 0, -7.3], [0, 1.0, -170.4], [0, 0, 1.0]]), (416, w), k7)
M6 = M4.copy(); M6[1, 1] = 1.006
run('... shift + vertical scale 1.006', src, M6, (416, w), k7)
M7 = M4.copy(); M7[0, 1] = -0.037
run('... shift + x drift -0.037 per row', src, M7, (416, w), k7)
M8 = M4.copy(); M8[1, 0] = 0.04
run('... shift + y drift 0.04 per column', src, M8, (416, w), k7)
