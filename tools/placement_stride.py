"""Does the LAYOUT of a batch inside its allocation (frame stride, row pitch) change what the
"class" of the allocation costs?  (round 5, review item 2)

The frames of a launch march in lock-step: at any time the 64 frames of a strip are read and
written at the same (row, column), i.e. at addresses that differ by multiples of the frame
stride (33 177 600 B for dense 4K float32 = 2025 x 16 KiB).  If what makes an allocation slow is
how those concurrent streams fall on the memory channels / banks, padding the frame stride or
the row pitch must move the slow blocks and leave the fast ones alone.

    python tools/placement_stride.py [n_blocks]

Blocks of 2.4 GB are drawn as they come (placement off), probed by the pool's 3x3 probe and
listed in ALLOCATION ORDER (is the class periodic in the order?); then the headline launch
(64 x 4K undistort + 5x5 through the C ABI with explicit strides) on the fastest and the slowest
block as source / result with padded frame strides and pitches.  GPU box only.
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
BLOCK = 2400 << 20


def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = ia.default_context(0)
    ctx._place_n = 1
    lib = ctx._lib
    ptrs = []
    for _ in range(nblk):
        p = C.c_void_p()
        L.check(lib.ipa_malloc(ctx.handle, BLOCK, C.byref(p)), ctx.handle, 'malloc')
        ptrs.append(p)
    dense = B * H * W * 4
    t = [ctx._probe_block(p, dense) for p in ptrs]
    t2 = [ctx._probe_block(p, dense) for p in ptrs]
    print('allocation order: address, probe ms (two passes)')
    for i, p in enumerate(ptrs):
        print('  %2d  0x%x  %.4f  %.4f' % (i, p.value, t[i], t2[i]))
    order = np.argsort(t)
    fast, fast2, slow, slow2 = order[0], order[1], order[-1], order[-2]
    print('fast blocks %d %d (%.4f %.4f), slow blocks %d %d (%.4f %.4f)'
          % (fast, fast2, t[fast], t[fast2], slow, slow2, t[slow], t[slow2]))

    K = np.array([[float(W), 0, (W - 1) / 2.0], [0, float(W), (H - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.ascontiguousarray(np.outer(g, g), dtype=np.float64)
    dmx, dmy = ops.build_undistort_map(K, dist, K, H, W, ctx=ctx, device=True)
    # contents do not matter for the timing; fill the blocks once so that no page is untouched
    for p in ptrs:
        L.check(lib.ipa_memset(ctx.handle, p, 0x3c, BLOCK), ctx.handle, 'memset')
    ctx.synchronize()

    def launch(sp, dp, spitch, dpitch, sstride, dstride, n=B):
        L.check(lib.ipa_remap_conv2d_dev(
            ctx.handle, sp, L.F32, H, W, spitch, dmx.ptr, dmy.ptr, W,
            k5.ctypes.data_as(C.POINTER(C.c_double)), 5, 5, dp, L.F32, H, W, dpitch, n,
            sstride, dstride, L.INTER_LINEAR, L.BORDER_CONSTANT, 0.0, L.BORDER_REFLECT,
            L.BORDER_REFLECT), ctx.handle, 'remap_conv2d')

    def conv(sp, dp, spitch, dpitch, sstride, dstride, n=B):
        L.check(lib.ipa_conv2d_dev(
            ctx.handle, sp, L.F32, H, W, spitch, k5.ctypes.data_as(C.POINTER(C.c_double)), 5, 5,
            None, 0, dp, dpitch, n, sstride, dstride, L.BORDER_REFLECT, L.BORDER_REFLECT, 0.0),
            ctx.handle, 'conv2d')

    def timeit(fn, *a):
        for _ in range(12):
            fn(*a)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        e0.record()
        for _ in range(30):
            fn(*a)
        e1.record()
        ctx.synchronize()
        return e0.elapsed_ms(e1) / 30

    # settle the clocks
    for _ in range(300):
        launch(ptrs[fast], ptrs[fast2], W, W, H * W, H * W)
    ctx.synchronize()

    pairs = [('fast->fast', fast, fast2), ('fast->slow', fast, slow), ('slow->fast', slow, fast),
             ('slow->slow', slow, slow2)]
    pads = [0, 64, 256, 1024, 4096, 3 * 4096, 5 * 4096, 16384, 16384 + 4096, 65536 + 4096,
            262144 + 4096, (1 << 20) - 4096, 1 << 20]
    for kname, fn in (('fused undistort + 5x5', launch), ('plain 5x5', conv)):
        print('--- %s, 64 x 4K: frame stride = dense + pad bytes (source and result alike), ms per launch' % kname)
        print('%-12s' % 'pad' + ''.join('%12s' % n for n, _, _ in pairs))
        for pad in pads:
            st = H * W + pad // 4
            row = [timeit(fn, ptrs[s], ptrs[d], W, W, st, st) for _, s, d in pairs]
            print('%-12d' % pad + ''.join('%12.4f' % v for v in row), flush=True)
        print('--- %s: padded only on one side (pad 20480 B)' % kname)
        st = H * W + 20480 // 4
        for name, s, d in pairs:
            print('%-12s source padded %.4f   result padded %.4f' % (
                name, timeit(fn, ptrs[s], ptrs[d], W, W, st, H * W),
                timeit(fn, ptrs[s], ptrs[d], W, W, H * W, st)), flush=True)
        print('--- %s: row pitch (elements), frame stride = pitch x rows' % kname)
        print('%-12s' % 'pitch' + ''.join('%12s' % n for n, _, _ in pairs))
        for pitch in (3840, 3840 + 32, 3840 + 64, 3840 + 128, 3840 + 192, 4096, 4096 + 64):
            st = H * pitch
            row = [timeit(fn, ptrs[s], ptrs[d], pitch, pitch, st, st) for _, s, d in pairs]
            print('%-12d' % pitch + ''.join('%12.4f' % v for v in row), flush=True)
    # fewer frames in lock-step: the same 64 frames as 4 launches of 16 (what the lock-step costs)
    print('--- 4 launches of 16 frames instead of one of 64 (dense), ms per 64 frames')
    for name, s, d in pairs:
        def four(sp, dp):
            for q in range(4):
                off = q * 16 * H * W * 4
                launch(C.c_void_p(sp.value + off), C.c_void_p(dp.value + off), W, W, H * W, H * W, 16)
        print('%-12s %.4f' % (name, timeit(four, ptrs[s], ptrs[d])), flush=True)


if __name__ == '__main__':
    main()
