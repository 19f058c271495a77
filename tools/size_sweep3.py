"""uint8 Lanczos4 remap around 1080 x 1920: which of height / width makes the 1079 x 1919 launch slow.
GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)


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
for (n, h, w) in ((16, 1080, 1920), (16, 1079, 1920), (16, 1080, 1919), (16, 1079, 1919), (16, 1080, 1918),
                  (16, 1081, 1921), (16, 1080, 1792), (16, 1080, 1793), (16, 1080, 2047), (16, 1080, 2048),
                  (16, 1072, 1919), (16, 1088, 1919), (8, 1080, 1919), (32, 1080, 1919)):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    u8 = ctx.to_device(rng.integers(0, 256, (n, h, w)).astype(np.uint8))
    o8 = ctx.empty((n, h, w), np.uint8)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    t = timeit(lambda: ops.remap(u8, dmx, dmy, 'lanczos4', out=o8))
    t2 = timeit(lambda: ops.remap(u8, dmx, dmy, 'linear', out=o8))
    print('%-20s lz4 %8.1f  lin %8.1f us  (%.2f / %.2f ns per Mpx)' % ('%d x %d x %d' % (n, h, w), t, t2, t / (n * h * w) * 1e3, t2 / (n * h * w) * 1e3), flush=True)
    del u8, o8, dmx, dmy
