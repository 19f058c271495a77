"""GPU: the standalone ring remap (csrc/ring_remap.hpp: batches of float32 frames, clean strips
sampled from an LDS ring of source rows, bilinear / bicubic / Lanczos4) against the gather
kernel (csrc/remap_impl.hpp) - the same arithmetic in the same order, so the results must agree
BIT FOR BIT - and against the oracle.
"""
import numpy as np
import pytest

from .conftest import assert_close
from .gpu_helpers import frames, radial_maps, rot_maps, same_bits

pytestmark = pytest.mark.gpu

INTERPS = ['linear', 'linear_cv_q5', 'cubic', 'cubic_cv_q5', 'lanczos4']


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def orc_interp(oracle, name):
    return {'linear': oracle.LINEAR, 'linear_cv_q5': oracle.LINEAR | oracle.Q5,
            'cubic': oracle.CUBIC_KEYS, 'cubic_cv': oracle.CUBIC_CV,
            'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5, 'lanczos4': oracle.LANCZOS4}[name]


def both(ia, fn):
    """fn() with the gather kernel alone, then with the ring kernel taking the clean strips"""
    ctx = ia.default_context(0)
    old = ctx.set_tuning(ring_remap=0, ring_min=1)
    try:
        ref = fn().get()
        ctx.set_tuning(ring_remap=2)
        got = fn().get()
    finally:
        ctx.set_tuning(**old)
    return ref, got


@pytest.mark.parametrize('interp', INTERPS)
@pytest.mark.parametrize('shape,n', [((96, 300), 1), ((131, 517), 3), ((200, 1030), 6)])
def test_ring_remap_radial(ia, oracle, interp, shape, n):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    ref, got = both(ia, lambda: ops.remap(d_src, dmx, dmy, interp))
    same_bits(got, ref, '%s %r x%d' % (interp, shape, n))
    want = oracle.remap(src[n - 1], mx, my, orc_interp(oracle, interp))
    assert_close(got[n - 1], want, 1e-5, 1e-5 * np.abs(want).max(), interp + ' vs oracle')


@pytest.mark.parametrize('interp', INTERPS)
@pytest.mark.parametrize('case', ['rot3', 'rot20', 'rot90', 'shift_out', 'pincushion', 'flipx',
                                  'zoom_out', 'zoom_in'])
def test_ring_remap_geometries(ia, interp, case):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 150, 700, 5
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
    for kw in ({}, {'border_mode': 'reflect'}, {'border_mode': 'replicate', 'border_value': 0.5}):
        ref, got = both(ia, lambda: ops.remap(d_src, dmx, dmy, interp, **kw))
        same_bits(got, ref, '%s %s %r' % (case, interp, kw))


@pytest.mark.parametrize('interp', INTERPS)
def test_ring_remap_analytic_sources(ia, interp):
    """lens model and homography evaluated in the kernel"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 140, 900, 6
    src = frames(n, h, w)
    _, _, K, dist = radial_maps(h, w)
    d_src = ctx.to_device(src)
    M = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])
    ref, got = both(ia, lambda: ops.undistort(d_src, K, dist, K, interp))
    same_bits(got, ref, 'undistort ' + interp)
    ref, got = both(ia, lambda: ops.warp_perspective(d_src, M, (h, w), interp))
    same_bits(got, ref, 'homography ' + interp)
    ref, got = both(ia, lambda: ops.warp_perspective(d_src, M, (h + 37, w - 101), interp))
    same_bits(got, ref, 'homography, other output size ' + interp)


def test_ring_remap_nan_and_far_coordinates(ia):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 80, 600, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    mx = mx.copy(); my = my.copy()
    mx[10, 50:60] = np.nan
    my[20, 300:310] = np.inf
    mx[30, 400:420] = 3e7
    mx[40:44, :] = -5.0
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    for interp in INTERPS:
        ref, got = both(ia, lambda: ops.remap(d_src, dmx, dmy, interp, border_value=0.25))
        same_bits(got, ref, interp)


@pytest.mark.parametrize('interp', ['linear', 'cubic', 'lanczos4'])
def test_ring_remap_4k(ia, oracle, interp):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 2160, 3840, 3
    src = frames(n, h, w)
    d_src = ctx.to_device(src)
    M = np.array([[0.97, 0.02, 20.0], [-0.015, 1.02, 12.5], [2e-6, -3e-6, 1.0]])
    ref, got = both(ia, lambda: ops.warp_perspective(d_src, M, (h, w), interp))
    same_bits(got, ref, '4K homography ' + interp)
    mx, my, _, _ = radial_maps(h, w)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    ref, got = both(ia, lambda: ops.remap(d_src, dmx, dmy, interp))
    same_bits(got, ref, '4K map ' + interp)
    sub = slice(1000, 1100)
    want = oracle.remap(src[2], mx, my, orc_interp(oracle, interp))
    assert_close(got[2][sub], want[sub], 1e-5, 1e-5 * np.abs(want).max(), '4K vs oracle')


def test_ring_remap_other_dtypes_take_the_gather_kernel(ia):
    """integer frames and float64 are not covered by the ring kernel: same results either way"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 100, 400, 3
    mx, my, _, _ = radial_maps(h, w)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    for dt in (np.uint16, np.float64):
        d_src = ctx.to_device(frames(n, h, w, dt) if dt == np.uint16 else frames(n, h, w).astype(dt))
        ref, got = both(ia, lambda: ops.remap(d_src, dmx, dmy, 'cubic'))
        assert np.array_equal(ref, got)


def test_ring_remap_plan_reuse(ia):
    """sources given by value keep their plan + coordinates for the next call with the same
    parameters: repeated, alternating and changed parameters all give the gather kernel's bits"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 140, 900, 4
    src = frames(n, h, w)
    d_src = ctx.to_device(src)
    Ms = [np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]]),
          np.array([[1.02, -0.01, -3.0], [0.015, 0.99, 1.5], [-1e-5, 1e-5, 1.0]])]
    old = ctx.set_tuning(ring_remap=0, ring_min=1)
    try:
        want = {(i, interp): ops.warp_perspective(d_src, M, (h, w), interp).get()
                for i, M in enumerate(Ms) for interp in ('cubic', 'lanczos4')}
        ctx.set_tuning(ring_remap=2)
        for i, interp in [(0, 'cubic'), (0, 'cubic'), (1, 'cubic'), (0, 'cubic'), (0, 'lanczos4'),
                          (0, 'lanczos4'), (1, 'lanczos4'), (1, 'cubic'), (1, 'cubic')]:
            got = ops.warp_perspective(d_src, Ms[i], (h, w), interp).get()
            same_bits(got, want[(i, interp)], 'M%d %s' % (i, interp))
        # another output size with the same matrix
        ctx.set_tuning(ring_remap=0)
        w2 = ops.warp_perspective(d_src, Ms[1], (h - 9, w - 130), 'cubic').get()
        ctx.set_tuning(ring_remap=2)
        same_bits(ops.warp_perspective(d_src, Ms[1], (h - 9, w - 130), 'cubic').get(), w2, 'size')
        # a different source size with the same matrix and output size (another inside test)
        src2 = ctx.to_device(frames(n, h - 20, w - 40))
        ctx.set_tuning(ring_remap=0)
        w3 = ops.warp_perspective(src2, Ms[1], (h - 9, w - 130), 'cubic').get()
        ctx.set_tuning(ring_remap=2)
        same_bits(ops.warp_perspective(src2, Ms[1], (h - 9, w - 130), 'cubic').get(), w3, 'src size')
    finally:
        ctx.set_tuning(**old)


def test_lens_map_cache_matches_per_pixel_model(ia):
    """fused undistort + filter: the cached float32 maps (default) against the lens model
    evaluated in the kernel (lens_cache=0) - bit for bit, across changes of model and size"""
    from imgprocessor_amd import ops
    from .gpu_helpers import kern
    ctx = ia.default_context(0)
    h, w, n = 140, 900, 3
    d_src = ctx.to_device(frames(n, h, w))
    _, _, K, dist = radial_maps(h, w)
    K2 = K.copy(); K2[0, 2] += 3.5
    dist2 = dist * 0.5
    g = ops.gaussian_kernel1d(1.0)
    calls = [(K, dist, K), (K, dist, K), (K2, dist, K), (K, dist2, K2), (K, dist, K)]
    for k in (kern(5, 2), kern(9, 4)):
        old = ctx.set_tuning(lens_cache=0)
        try:
            want = [ops.undistort_conv2d(d_src, a, b, c, k).get() for a, b, c in calls]
            want_s = [ops.undistort_sepconv2d(d_src, a, b, c, g, g).get() for a, b, c in calls]
            ctx.set_tuning(lens_cache=1)
            for i, (a, b, c) in enumerate(calls):
                same_bits(ops.undistort_conv2d(d_src, a, b, c, k).get(), want[i], 'conv %d' % i)
                same_bits(ops.undistort_sepconv2d(d_src, a, b, c, g, g).get(), want_s[i], 'sep %d' % i)
            # a smaller frame with the same model
            d2 = ctx.to_device(frames(n, h - 13, w - 77))
            ctx.set_tuning(lens_cache=0)
            w2 = ops.undistort_conv2d(d2, K, dist, K, k).get()
            ctx.set_tuning(lens_cache=1)
            same_bits(ops.undistort_conv2d(d2, K, dist, K, k).get(), w2, 'other size')
        finally:
            ctx.set_tuning(**old)


def test_ring_remap_hint_skips_the_ring_for_rotations(ia):
    """a source that leaves no clean strips (strong rotation): from the second call on the ring
    path is skipped on the hint of the first; results stay those of the gather kernel"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 300, 900, 4
    src = frames(n, h, w)
    d_src = ctx.to_device(src)
    mx, my = rot_maps(h, w, 25.0)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    a = np.deg2rad(25.0)
    M = np.array([[np.cos(a), -np.sin(a), 100.0], [np.sin(a), np.cos(a), -80.0], [0, 0, 1.0]])
    old = ctx.set_tuning(ring_remap=0, ring_min=1)
    try:
        want_m = ops.remap(d_src, dmx, dmy, 'cubic').get()
        want_h = ops.warp_perspective(d_src, M, (h, w), 'lanczos4').get()
        ctx.set_tuning(ring_remap=2)
        for i in range(4):
            same_bits(ops.remap(d_src, dmx, dmy, 'cubic').get(), want_m, 'maps, call %d' % i)
            ctx.synchronize()
        for i in range(4):
            same_bits(ops.warp_perspective(d_src, M, (h, w), 'lanczos4').get(), want_h,
                      'homography, call %d' % i)
        # back to a source with clean strips
        mx2, my2, _, _ = radial_maps(h, w)
        d2x, d2y = ctx.to_device(mx2), ctx.to_device(my2)
        ctx.set_tuning(ring_remap=0)
        want_r = ops.remap(d_src, d2x, d2y, 'cubic').get()
        ctx.set_tuning(ring_remap=2)
        for i in range(3):
            same_bits(ops.remap(d_src, d2x, d2y, 'cubic').get(), want_r, 'radial, call %d' % i)
    finally:
        ctx.set_tuning(**old)


def test_stored_coordinates_have_the_bits_of_the_per_pixel_evaluation(oracle):
    """batches of bicubic / Lanczos4 warps that the ring kernel declines (rotation): the
    homography / lens model is evaluated once into the plan buffer and every frame reads it
    (knob stored_coords) - identical bits to evaluating it per pixel and frame, and the oracle's
    values"""
    import imgprocessor_amd as ia
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    rng = np.random.default_rng(13)
    n, h, w = 6, 150, 331
    src = rng.random((n, h, w), dtype=np.float32)
    d = ctx.to_device(src)
    a = np.deg2rad(11.0)
    M = np.array([[np.cos(a), -np.sin(a), 40.0], [np.sin(a), np.cos(a), -30.0], [3e-5, -2e-5, 1.0]])
    Kc = np.array([[300.0, 0, (w - 1) / 2.0], [0, 300.0, (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.2, 0.05, 1e-3, -1e-3, 0.0])
    oi = {'cubic': oracle.CUBIC_KEYS, 'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5,
          'lanczos4': oracle.LANCZOS4}
    old = ctx.set_tuning(stored_coords=0)
    try:
        for interp in ('cubic', 'cubic_cv_q5', 'lanczos4'):
            for border in ('constant', 'reflect'):
                for name, call in (
                        ('warp', lambda: ops.warp_perspective(d, M, (h - 7, w + 5), interp, border, 0.25)),
                        ('undistort', lambda: ops.undistort(d, Kc, dist, Kc, interp, border, 0.25))):
                    ctx.set_tuning(stored_coords=0)
                    ref = call().get()
                    ctx.set_tuning(stored_coords=4)
                    for rep in range(2):      # the second call reuses the stored coordinates
                        got = call().get()
                        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
                            (name, interp, border, rep)
                    if name == 'warp':
                        want = oracle.warp_perspective(src[2], M, (h - 7, w + 5), oi[interp],
                                                       border, 0.25)
                        assert np.abs(got[2] - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
        # below the batch threshold nothing is stored (and nothing changes)
        ctx.set_tuning(stored_coords=0)
        ref = ops.warp_perspective(d, M, (h, w), 'lanczos4').get()
        ctx.set_tuning(stored_coords=8)
        assert np.array_equal(ops.warp_perspective(d, M, (h, w), 'lanczos4').get(), ref)
    finally:
        ctx.set_tuning(**old)
