"""Does the ADDRESS of a batch buffer decide how fast the fused kernel runs on it?  One 8 GiB
allocation; the 2 GiB source (then the result) of the 4K headline placed at different byte offsets
inside it, the partner buffer fixed.  GPU box only.

    python tools/placement_offset.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.device import DeviceArray  # noqa: E402


def view(base, off, shape, dtype):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype = base.ctx, tuple(shape), np.dtype(dtype)
    v.nbytes = int(np.prod(shape, dtype=np.int64)) * v.dtype.itemsize
    v.ptr = C.c_void_p(base.ptr.value + off)
    v._owner = False
    v._base = base
    return v


def timeit(ctx, fn, n=12, warm=4):
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
    B, h, w = 64, 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    nb = B * h * w * 4
    big = ctx.empty((8 << 30,), np.uint8)
    fixed_src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
    fixed_dst = ctx.empty((B, h, w), np.float32)
    print('big at %#x, fixed source %#x, fixed result %#x' % (big.ptr.value, fixed_src.ptr.value, fixed_dst.ptr.value))
    for _ in range(30):
        ops.remap_conv2d(fixed_src, dmx, dmy, k5, out=fixed_dst)
    offs = [0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20,
            1 << 30, (1 << 30) + (2 << 20), 2 << 30, 3 << 30, (3 << 30) + (512 << 20), 4 << 30, 5 << 30]
    offs = [o for o in offs if o + nb <= (8 << 30)]
    for role in ('source', 'result'):
        print('--- the %s inside the big allocation, ms per 64 x 4K launch' % role)
        for rep in range(2):
            row = []
            for o in offs:
                v = view(big, o, (B, h, w), np.float32)
                if role == 'source':
                    v.copy_from(fixed_src)
                    t = timeit(ctx, lambda: ops.remap_conv2d(v, dmx, dmy, k5, out=fixed_dst))
                else:
                    t = timeit(ctx, lambda: ops.remap_conv2d(fixed_src, dmx, dmy, k5, out=v))
                row.append(t)
            print('  '.join('%#x:%.4f' % (o, t) for o, t in zip(offs, row)))
    print('fixed pair: %.4f' % timeit(ctx, lambda: ops.remap_conv2d(fixed_src, dmx, dmy, k5, out=fixed_dst)))
    # fresh allocations for comparison
    for i in range(4):
        a = ctx.empty((B, h, w), np.float32)
        a.copy_from(fixed_src)
        print('fresh allocation %#x as source: %.4f, as result: %.4f' % (
            a.ptr.value, timeit(ctx, lambda: ops.remap_conv2d(a, dmx, dmy, k5, out=fixed_dst)),
            timeit(ctx, lambda: ops.remap_conv2d(fixed_src, dmx, dmy, k5, out=a))))
        globals()['keep%d' % i] = a


if __name__ == '__main__':
    main()
