"""Does a cheap probe on a fresh 2 GiB allocation predict how the fused kernel runs on it?
For N fresh allocations: the plain 3x3 filter in place (reads and writes the candidate), a device
copy into / out of it, and the 4K headline with the candidate as source / as result.  GPU box only.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def view(base, off, shape):
    import ctypes as C
    from imgprocessor_amd.device import DeviceArray
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype = base.ctx, tuple(shape), base.dtype
    v.nbytes = int(np.prod(shape, dtype=np.int64)) * v.dtype.itemsize
    v.ptr = C.c_void_p(base.ptr.value + off)
    v._owner = False
    v._base = base
    return v


def timeit(ctx, fn, n=10, warm=3):
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
    ctx = ia.default_context(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B, h, w = 64, 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    k3 = np.ones((3, 3)) / 9
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    fsrc = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
    fdst = ctx.empty((B, h, w), np.float32)
    for _ in range(30):
        ops.remap_conv2d(fsrc, dmx, dmy, k5, out=fdst)
    print('fixed pair %.4f' % timeit(ctx, lambda: ops.remap_conv2d(fsrc, dmx, dmy, k5, out=fdst)))
    keep = []
    print('%-16s %9s %9s %9s %9s %9s %9s' % ('allocation', 'conv3 in', 'conv3 rd', 'conv3 wr', 'copy', 'fused src', 'fused dst'))
    for i in range(N):
        a = ctx.empty((B, h, w), np.float32)
        a.copy_from(fsrc)
        keep.append(a)
        half = B // 2
        lo, hi = view(a, 0, (half, h, w)), view(a, half * h * w * 4, (half, h, w))
        t_in = timeit(ctx, lambda: ops.conv2d(lo, k3, out=hi))   # first half -> second half
        a.copy_from(fsrc)
        t_rd = timeit(ctx, lambda: ops.conv2d(a, k3, out=fdst))
        t_wr = timeit(ctx, lambda: ops.conv2d(fsrc, k3, out=a))
        t_cp = timeit(ctx, lambda: a.copy_from(fsrc))
        a.copy_from(fsrc)
        t_s = timeit(ctx, lambda: ops.remap_conv2d(a, dmx, dmy, k5, out=fdst))
        t_d = timeit(ctx, lambda: ops.remap_conv2d(fsrc, dmx, dmy, k5, out=a))
        print('%#-16x %9.4f %9.4f %9.4f %9.4f %9.4f %9.4f' % (a.ptr.value, t_in, t_rd, t_wr, t_cp, t_s, t_d))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
