import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from oracle import oracle as orc
orc.build()
ctx = ia.default_context(0)
h, w, n = 540, 1920, 4
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
rng = np.random.default_rng(0)
u16 = np.round(rng.random((n, h, w)) * 4095).astype(np.uint16)
d = ctx.to_device(u16)
k = rng.random((7, 7)); k /= k.sum()
a = ops.remap_conv2d(d, dmx, dmy, k).get()
ctx.set_tuning(frames_wg=0)
b = ops.remap_conv2d(d, dmx, dmy, k).get()
want = orc.conv2d(orc.remap(u16[0], dmx.get(), dmy.get(), out_dtype=np.float32), k)
np.set_printoptions(linewidth=200, precision=1, suppress=True)
for r in (31, 32, 33, 64):
    print('row', r, 'shared ', a[0, r, 300:312])
    print('row', r, 'perframe', b[0, r, 300:312])
    print('row', r, 'oracle  ', want[r, 300:312])
# which single-row kernel contribution is off: fit a[32] - b[32] against rows of the remapped image
rem = orc.remap(u16[0], dmx.get(), dmy.get(), out_dtype=np.float32)
diff = (a[0, 32] - b[0, 32])[300:1500]
import numpy.linalg as la
# candidate: extra term = sum_j k[i, j] * rem[32 - 3 + i + s, x + j - 3] for row shifts s
for s in range(-8, 9):
    for i in range(7):
        rr = 32 - 3 + i + s
        if 0 <= rr < h:
            pred = sum(k[i, j] * rem[rr, 300 + j - 3:1500 + j - 3] for j in range(7))
            c = np.corrcoef(pred, diff)[0, 1]
            if abs(c) > 0.5:
                print('diff correlates with kernel row %d applied to remapped row %d (shift %d): %.3f, scale %.3f' % (i, rr, s, c, (diff @ pred) / (pred @ pred)))
print('mean diff', diff.mean(), 'mean b', b[0, 32, 300:1500].mean())
