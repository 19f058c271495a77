import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from tools.angle_sweep import rot_persp, timed
ctx = ia.default_context(0)
rng = np.random.default_rng(1)
bad = 0; n = 0
for (h, w, B) in ((301, 517, 3), (33, 200, 5)):
    src = ctx.to_device(rng.random((B, h, w), dtype=np.float32))
    for deg in (0, 3, 17, 45, 90, 133, 180, 271):
        M = rot_persp(h, w, deg)
        M[2, 0] *= 50; M[2, 1] *= 50
        if deg == 17: M[0, 2] += 40; M[1, 2] -= 25
        for it in ('linear', 'cubic', 'cubic_cv', 'linear_cv_q5', 'cubic_cv_q5', 'lanczos4'):
            for bm in ('constant', 'replicate', 'reflect', 'wrap', 'reflect101'):
                for oshape in ((h, w), (h + 13, w - 7)):
                    res = []
                    for tw in (0, 2):
                        ctx.set_tuning(tile_warp=tw)
                        res.append(ops.warp_perspective(src, M, oshape, it, bm, border_value=0.25).get())
                    n += 1
                    if not np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32)):
                        bad += 1
                        d = np.abs(res[0] - res[1])
                        if bad < 20: print('MISMATCH', h, w, B, deg, it, bm, oshape, d.max(), np.argwhere(d > 0)[:3])
print('cases', n, 'mismatches', bad, flush=True)
B, h, w = 16, 2160, 3840
src = ctx.to_device(rng.random((B, h, w), dtype=np.float32)); dst = ctx.empty((B, h, w), np.float32)
print('%6s %s' % ('deg', ' '.join('%9s/%-9s' % (it, 'tile') for it in ('linear', 'cubic', 'lanczos4'))))
for deg in (0, 1, 4, 7, 15, 30, 45, 90):
    M = rot_persp(h, w, deg); row = []
    for it in ('linear', 'cubic', 'lanczos4'):
        for tw in (0, 2):
            ctx.set_tuning(tile_warp=tw)
            row.append(timed(ctx, lambda: ops.warp_perspective(src, M, (h, w), it, out=dst)))
    print('%6.1f ' % deg + ' '.join('%9.3f/%-9.3f' % (row[2*i], row[2*i+1]) for i in range(3)), flush=True)
