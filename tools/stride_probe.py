"""frame stride against the memory channels: the fused undistort + 5x5 on 64 x 4K float32 frames with the
frames of the batch contiguous (stride 33 177 600 B = 2^13 * 4050) and with padded frame strides"""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.device import dtype_id  # noqa: E402
from bench_micro import timeit  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 64, 2160, 3840
Kc = np.array([[3840., 0, 1919.5], [0, 3840., 1079.5], [0, 0, 1]])
dc = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
g = np.exp(-0.5 * np.arange(-2, 3) ** 2.0)
g /= g.sum()
k = np.ascontiguousarray(np.outer(g, g), dtype=np.float64)
cb = ops.border_id('reflect')
for pad_elems in (0, 64, 1024 + 64, 3840, 3840 * 2 + 192, 5 * 3840 + 320):
    fs = h * w + pad_elems
    src = ctx.empty((B * fs,), np.float32)
    dst = ctx.empty((B * fs,), np.float32)
    ctx._check(ctx._lib.ipa_memset(ctx.handle, src.ptr, 0, src.nbytes), 'memset')

    def run():
        ctx._check(ctx._lib.ipa_remap_conv2d_dev(
            ctx.handle, src.ptr, dtype_id(np.float32), h, w, w, dmx.ptr, dmy.ptr, w,
            k.ctypes.data_as(C.POINTER(C.c_double)), 5, 5, dst.ptr, dtype_id(np.float32), h, w, w,
            B, fs, fs, ops.interp_id('linear'), ops.border_id('constant'), 0.0, cb, cb), 'remap_conv2d')
    for i in range(2):
        t = timeit(ctx, run, n=30, warm=30)
    print('frame stride + %6d floats (%9d B): %8.1f us' % (pad_elems, fs * 4, t), flush=True)
    del src, dst
