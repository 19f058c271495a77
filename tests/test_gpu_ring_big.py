"""GPU: remap -> 7x7 / 9x9 / 11x11 of float32 batches in one kernel with the taps in LDS
(csrc/ring_big.hpp, BASELINE configuration C5) against the existing paths (per-frame one-kernel
form for bilinear from maps, remap kernel -> filter kernel through the workspace for the rest):
the same arithmetic in the same order, so the results must agree BIT FOR BIT.
"""
import numpy as np
import pytest

from .conftest import assert_close
from .test_gpu_group import frames, kern, radial_maps, rot_maps, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    ctx = imgprocessor_amd.default_context(0)
    if not ctx.get_tuning('experimental'):
        pytest.skip('ring_big_kernel is a shelved round-2 kernel: build with make EXPERIMENTAL=1')
    return imgprocessor_amd


def both(ia, fn):
    ctx = ia.default_context(0)
    old = ctx.set_tuning(ring_big=0, ring_min=1)
    try:
        ref = fn().get()
        ctx.set_tuning(ring_big=2)
        got = fn().get()
    finally:
        ctx.set_tuning(**old)
    return ref, got


HM = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])


@pytest.mark.parametrize('K', [7, 9, 11])
@pytest.mark.parametrize('interp', ['linear', 'cubic', 'cubic_cv_q5', 'linear_cv_q5'])
@pytest.mark.parametrize('shape,n', [((150, 300), 1), ((131, 517), 3), ((200, 1030), 6)])
def test_ring_big_sources(ia, oracle, K, interp, shape, n):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w)
    mx, my, Kc, dist = radial_maps(h, w)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    k = kern(K, K)
    ref, got = both(ia, lambda: ops.remap_conv2d(d_src, dmx, dmy, k, interp))
    same_bits(got, ref, 'maps %s K=%d' % (interp, K))
    ref, got = both(ia, lambda: ops.warp_perspective_conv2d(d_src, HM, (h, w), k, interp))
    same_bits(got, ref, 'homography %s K=%d' % (interp, K))
    ref, got = both(ia, lambda: ops.undistort_conv2d(d_src, Kc, dist, Kc, k, interp))
    same_bits(got, ref, 'lens model %s K=%d' % (interp, K))
    if interp in ('linear', 'cubic'):
        oi = oracle.LINEAR if interp == 'linear' else oracle.CUBIC_KEYS
        want = oracle.conv2d(oracle.remap(src[n - 1], mx, my, oi), k)
        assert_close(got[n - 1], want, 2e-5, 2e-5 * np.abs(want).max(), 'vs oracle')


@pytest.mark.parametrize('interp', ['linear', 'cubic'])
@pytest.mark.parametrize('case', ['rot3', 'rot20', 'rot90', 'shift_out', 'pincushion', 'flipx',
                                  'zoom_out', 'zoom_in'])
def test_ring_big_geometries(ia, interp, case):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 170, 700, 5
    src = frames(n, h, w)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    if case.startswith('rot'):
        mx, my = rot_maps(h, w, float(case[3:]))
    elif case == 'shift_out':
        mx, my, _, _ = radial_maps(h, w, shift=-40.5)
    elif case == 'pincushion':
        mx, my, _, _ = radial_maps(h, w, k1=0.25)
    elif case == 'flipx':
        mx, my = (w - 1 - x + 0.25).astype(np.float32), (y + 0.5).astype(np.float32)
    elif case == 'zoom_out':
        mx, my = (x * 1.7 - 100).astype(np.float32), (y * 1.7 - 30).astype(np.float32)
    else:
        mx, my = (x * 0.31 + 7.3).astype(np.float32), (y * 0.31 + 3.1).astype(np.float32)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    for K, kw in ((11, {}), (9, {'conv_mode': 'constant'}),
                  (7, {'border_mode': 'reflect', 'conv_mode': 'wrap'}),
                  (11, {'border_mode': 'replicate', 'border_value': 0.5, 'conv_mode': 'mirror'})):
        k = kern(K, 9)
        ref, got = both(ia, lambda: ops.remap_conv2d(d_src, dmx, dmy, k, interp, **kw))
        same_bits(got, ref, '%s %s K=%d %r' % (case, interp, K, kw))


def test_ring_big_nan_and_far_coordinates(ia):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 120, 600, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    mx = mx.copy(); my = my.copy()
    mx[10, 50:60] = np.nan
    my[20, 300:310] = np.inf
    mx[30, 400:420] = 3e7
    mx[40:44, :] = -5.0
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    for interp in ('linear', 'cubic'):
        ref, got = both(ia, lambda: ops.remap_conv2d(d_src, dmx, dmy, kern(9), interp,
                                                     border_value=0.25))
        same_bits(got, ref, interp)


def test_ring_big_plan_reuse(ia):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 150, 900, 4
    d_src = ctx.to_device(frames(n, h, w))
    Ms = [HM, np.array([[1.02, -0.01, -3.0], [0.015, 0.99, 1.5], [-1e-5, 1e-5, 1.0]])]
    k11, k9 = kern(11, 1), kern(9, 2)
    old = ctx.set_tuning(ring_big=0, ring_min=1)
    try:
        want = {(i, K): ops.warp_perspective_conv2d(d_src, M, (h, w), k, 'cubic').get()
                for i, M in enumerate(Ms) for K, k in ((11, k11), (9, k9))}
        ctx.set_tuning(ring_big=2)
        for i, K in [(0, 11), (0, 11), (1, 11), (0, 9), (0, 9), (1, 9), (1, 11), (0, 11)]:
            got = ops.warp_perspective_conv2d(d_src, Ms[i], (h, w), k11 if K == 11 else k9,
                                              'cubic').get()
            same_bits(got, want[(i, K)], 'M%d K=%d' % (i, K))
            # the standalone remap shares the plan buffer
            ops.warp_perspective(d_src, Ms[i], (h, w), 'lanczos4')
    finally:
        ctx.set_tuning(**old)


@pytest.mark.parametrize('interp', ['linear', 'cubic'])
def test_ring_big_4k(ia, oracle, interp):
    from imgprocessor_amd import ops
    from imgprocessor_amd.utils import getPerspectiveTransform
    ctx = ia.default_context(0)
    h, w, n = 2160, 3840, 3
    src = frames(n, h, w)
    d_src = ctx.to_device(src)
    quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
    rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
    Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
    k = kern(11, 3)
    ref, got = both(ia, lambda: ops.warp_perspective_conv2d(d_src, Hm, (h, w), k, interp))
    same_bits(got, ref, '4K homography ' + interp)
    mx, my, _, _ = radial_maps(h, w)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    ref, got = both(ia, lambda: ops.remap_conv2d(d_src, dmx, dmy, k, interp))
    same_bits(got, ref, '4K maps ' + interp)
