import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from tools.angle_sweep import rot_persp, timed
ctx = ia.default_context(0)
B, h, w = 16, 2160, 3840
src = ctx.to_device(np.random.default_rng(1).random((B, h, w), dtype=np.float32)); dst = ctx.empty((B, h, w), np.float32)
print('%6s %s' % ('deg', ' '.join('%9s/%-9s' % (it, 'tile') for it in ('linear', 'cubic', 'lanczos4'))))
for deg in (0, 1, 2, 4, 7, 10, 15, 22, 30, 37, 45, 52, 60, 75, 90, -4, -15, -30, -45, -60, 135, 180):
    M = rot_persp(h, w, deg); row = []
    for it in ('linear', 'cubic', 'lanczos4'):
        for tw in (0, 2):
            ctx.set_tuning(tile_warp=tw)
            row.append(timed(ctx, lambda: ops.warp_perspective(src, M, (h, w), it, out=dst), n=10, warm=5))
    print('%6.1f ' % deg + ' '.join('%9.3f/%-9.3f' % (row[2*i], row[2*i+1]) for i in range(3)), flush=True)
