"""Hypothesis H (round 5): what makes a pair of buffers slow for the strip-shaped kernels is the
PHYSICAL distance between the read stream and the write stream (both march at the same offset of
their buffer), slow when it is a multiple of a large power of two.  Test inside ONE 10 GiB
allocation (physically contiguous if the driver has such a block): source at offset 0, result at
4 GiB + d for a sweep of d; plain 3x3 / 5x5 and the headline launch, 64 x 4K.  GPU box only.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import _lib as L  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

B, H, W = 64, 2160, 3840
DENSE = B * H * W * 4


def main():
    ctx = ia.default_context(0)
    ctx._place_n = 1
    lib = ctx._lib
    big = C.c_void_p()
    L.check(lib.ipa_malloc(ctx.handle, 10 << 30, C.byref(big)), ctx.handle, 'malloc')
    L.check(lib.ipa_memset(ctx.handle, big, 0x3c, 10 << 30), ctx.handle, 'memset')
    K = np.array([[float(W), 0, (W - 1) / 2.0], [0, float(W), (H - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.ascontiguousarray(np.outer(g, g), dtype=np.float64)
    k3 = np.full((3, 3), 1.0 / 9)
    dmx, dmy = ops.build_undistort_map(K, dist, K, H, W, ctx=ctx, device=True)

    def fused(sp, dp):
        L.check(lib.ipa_remap_conv2d_dev(
            ctx.handle, sp, L.F32, H, W, W, dmx.ptr, dmy.ptr, W,
            k5.ctypes.data_as(C.POINTER(C.c_double)), 5, 5, dp, L.F32, H, W, W, B, H * W, H * W,
            L.INTER_LINEAR, L.BORDER_CONSTANT, 0.0, L.BORDER_REFLECT, L.BORDER_REFLECT),
            ctx.handle, 'remap_conv2d')

    def conv3(sp, dp):
        L.check(lib.ipa_conv2d_dev(
            ctx.handle, sp, L.F32, H, W, W, k3.ctypes.data_as(C.POINTER(C.c_double)), 3, 3, None, 0,
            dp, W, B, H * W, H * W, L.BORDER_REFLECT, L.BORDER_REFLECT, 0.0), ctx.handle, 'conv2d')

    def copy(sp, dp):
        L.check(lib.ipa_memcpy_d2d(ctx.handle, dp, sp, DENSE), ctx.handle, 'copy')

    def timeit(fn, sp, dp, warm=5, n=16):
        for _ in range(warm):
            fn(sp, dp)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        e0.record()
        for _ in range(n):
            fn(sp, dp)
        e1.record()
        ctx.synchronize()
        return e0.elapsed_ms(e1) / n

    at = lambda off: C.c_void_p(big.value + off)
    for _ in range(200):
        fused(at(0), at(4 << 30))
    ctx.synchronize()
    print('big block at 0x%x; source at +0, result at +4 GiB + d; ms per 64 x 4K' % big.value)
    print('%12s %10s %10s %10s   reversed: %10s %10s' % ('d', 'conv3', 'fused', 'copy', 'conv3', 'fused'))
    ds = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19,
          1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29,
          1 << 30, 3 << 29, (1 << 30) + (1 << 20), 12345 * 256, 0]
    for d in ds:
        sp, dp = at(0), at((4 << 30) + d)
        print('%12d %10.4f %10.4f %10.4f   reversed: %10.4f %10.4f'
              % (d, timeit(conv3, sp, dp), timeit(fused, sp, dp), timeit(copy, sp, dp),
                 timeit(conv3, dp, sp), timeit(fused, dp, sp)), flush=True)
    print('source at +s, result at +4 GiB')
    for s in [0, 4096, 1 << 16, 1 << 20, 1 << 24, 1 << 28, 1 << 30]:
        sp, dp = at(s), at(4 << 30)
        print('%12d %10.4f %10.4f' % (s, timeit(conv3, sp, dp), timeit(fused, sp, dp)), flush=True)
    print('both moved by the same m (distance 4 GiB)')
    for m in [0, 4096, 1 << 16, 1 << 20, 1 << 24, 1 << 28, 1 << 30]:
        sp, dp = at(m), at((4 << 30) + m)
        print('%12d %10.4f %10.4f' % (m, timeit(conv3, sp, dp), timeit(fused, sp, dp)), flush=True)


if __name__ == '__main__':
    main()
