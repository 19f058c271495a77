import os, sys, subprocess
if len(sys.argv) == 1:
    for ab, fwg, lds in ((0, 8, 150000), (0, 8, 75000), (0, 8, 50000), (0, 8, 39000), (0, 8, 31000), (0, 8, 26000), (0, 8, 20000), (2, 8, 150000), (2, 8, 50000), (2, 8, 31000), (2, 8, 20000)):
        env = dict(os.environ, IPA_TW_ABLATE=str(ab), IPA_TW_FWG=str(fwg), IPA_TW_LDS=str(lds))
        out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout
        print('ablate %d fwg %2d lds %6d: %s' % (ab, fwg, lds, out.strip()), flush=True)
    sys.exit(0)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from tools.angle_sweep import rot_persp, timed
ctx = ia.default_context(0)
ctx._place_n = 1
B, h, w = 16, 2160, 3840
src = ctx.to_device(np.random.default_rng(1).random((B, h, w), dtype=np.float32)); dst = ctx.empty((B, h, w), np.float32)
row = []
for it in ('linear',):
  for deg in (0, 15):
    M = rot_persp(h, w, deg)
    row.append('%s@%d:%.3f' % (it[:3], deg, timed(ctx, lambda: ops.warp_perspective(src, M, (h, w), it, out=dst), n=10, warm=5)))
import time
M = rot_persp(h, w, 0.0)
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.warp_perspective(src, M, (h, w), 'linear', out=dst)
t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
row.append('host/call %.3f ms, drain %.3f ms' % ((t1 - t0) * 50, (t2 - t1) * 1e3))
print(' '.join(row))
