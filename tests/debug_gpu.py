import numpy as np, sys
sys.path.insert(0, '.')
import imgprocessor_amd as ia
from oracle import oracle as orc
rng = np.random.default_rng(0)
img = rng.random((16, 24)).astype(np.float32)
yy, xx = np.mgrid[0:16, 0:24].astype(np.float32)
for interp in ('nearest', 'linear', 'cubic', 'lanczos4'):
    got = ia.ops.remap(img, xx, yy, interp)
    print(interp, 'identity maxdiff', np.abs(got - img).max())
got = ia.ops.remap(img, xx + 0.5, yy, 'linear')
want = orc.remap(img, xx + 0.5, yy)
print('half px shift maxdiff', np.abs(got - want).max())
print(got[3, :6], want[3, :6], img[3, :7])
got = ia.ops.remap(img, xx + 1, yy, 'linear')
print('int shift x', np.abs(got - orc.remap(img, xx + 1, yy)).max())
got = ia.ops.remap(img, xx, yy + 1, 'linear')
print('int shift y', np.abs(got - orc.remap(img, xx, yy + 1)).max())
got = ia.ops.remap(img, xx, yy + 0.25, 'linear')
print('y quarter', np.abs(got - orc.remap(img, xx, yy + 0.25)).max())
