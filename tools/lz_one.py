"""one Lanczos4 warp of 16 x 4K float32 frames, a few times (for counter passes)"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.utils import getPerspectiveTransform  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 16, 2160, 3840
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
dst = ctx.empty((B, h, w), np.float32)
quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
interp = sys.argv[1] if len(sys.argv) > 1 else 'lanczos4'
for _ in range(4):
    ops.warp_perspective(src, Hm, (h, w), interp, out=dst)
ctx.synchronize()
