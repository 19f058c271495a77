"""one uint8 remap of 16 x 4K frames from maps, a few times (for counter passes): interpolation = argv[1]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

ctx = ia.default_context(0)
B, h, w = 16, 2160, 3840
rng = np.random.default_rng(0)
src = ctx.to_device((rng.random((B, h, w), dtype=np.float32) * 255).astype(np.uint8))
dst = ctx.empty((B, h, w), np.uint8)
Kc = np.array([[3840., 0, 1919.5], [0, 3840., 1079.5], [0, 0, 1]])
dc = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
dmx, dmy = ops.build_undistort_map(Kc, dc, Kc, h, w, ctx=ctx, device=True)
interp = sys.argv[1] if len(sys.argv) > 1 else 'cubic_cv'
for _ in range(4):
    ops.remap(src, dmx, dmy, interp, out=dst)
ctx.synchronize()
