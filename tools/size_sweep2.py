"""integer frames over odd sizes: uint8 / uint16 remaps and the uint16 -> float32 fused chain.
GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)
k7 = np.random.default_rng(1).random((7, 7))
k7 /= k7.sum()


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3


rng = np.random.default_rng(0)
print('%-20s %9s %9s %9s %9s %9s %9s  us per launch' % ('frames x h x w', 'u8 lin', 'u8 lz4', 'u16 lin', 'u16->f32', 'u16 fused7', 'u8->f32'))
for (n, h, w) in ((16, 2160, 3840), (16, 2160, 3838), (16, 2160, 3841), (16, 1080, 1920), (16, 1079, 1919), (4, 2160, 3840), (4, 2160, 3839)):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    u8 = ctx.to_device(rng.integers(0, 256, (n, h, w)).astype(np.uint8))
    u16 = ctx.to_device(rng.integers(0, 4096, (n, h, w)).astype(np.uint16))
    o8, o16, of = ctx.empty((n, h, w), np.uint8), ctx.empty((n, h, w), np.uint16), ctx.empty((n, h, w), np.float32)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    t = [timeit(lambda: ops.remap(u8, dmx, dmy, 'linear', out=o8)),
         timeit(lambda: ops.remap(u8, dmx, dmy, 'lanczos4', out=o8)),
         timeit(lambda: ops.remap(u16, dmx, dmy, 'linear_cv_q5', out=o16)),
         timeit(lambda: ops.remap(u16, dmx, dmy, 'linear', out_dtype=np.float32, out=of)),
         timeit(lambda: ops.remap_conv2d(u16, dmx, dmy, k7, out=of)),
         timeit(lambda: ops.remap(u8, dmx, dmy, 'linear', out_dtype=np.float32, out=of))]
    print('%-20s %s' % ('%d x %d x %d' % (n, h, w), ' '.join('%9.1f' % x for x in t)), flush=True)
    del u8, u16, o8, o16, of, dmx, dmy
