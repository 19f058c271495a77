"""GPU: the remap -> filter chain entry points accept whatever the standalone entry points accept.  Fused kernels are
built for float32 frames (bilinear, both bicubics) and uint16 frames (bilinear; maps or the lens model) with square
3 .. 11 kernels; every other combination returned IPA_ERR_UNSUPPORTED until round 6 ("use ipa_remap_dev +
ipa_conv2d_dev") - uint16 frames under a homography with a 3 / 5 / 7 kernel did, with a 9 / 11 kernel did not.  Now the
library does that itself: remap into the context workspace, then the filter - the bits of the two calls
(the reference always makes two calls: cv2.remap / cv2.warpPerspective, camera/LensDistortion.py:323-326,
camera/PerspectiveCorrection.py:401-405, then a filter of filters/)."""
import numpy as np
import pytest

from .gpu_helpers import frames, kern, radial_maps, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


M = np.array([[1.01, 0.02, -3.0], [-0.015, 0.99, 2.0], [1e-5, -2e-5, 1.0]])

CASES = [
    # dtype, interpolation, kernel shape, coordinate sources that were unsupported
    (np.uint16, 'linear', (5, 5), ('warp',)),
    (np.uint16, 'linear', (3, 3), ('warp',)),
    (np.uint16, 'cubic', (5, 5), ('map', 'warp', 'lens')),
    (np.uint16, 'lanczos4', (7, 7), ('map', 'warp', 'lens')),
    (np.uint8, 'linear', (3, 3), ('warp',)),                      # (with maps: one kernel since the end of round 6)
    (np.uint8, 'linear_cv_q5', (9, 9), ('map', 'warp')),
    (np.float32, 'lanczos4', (5, 5), ('map', 'warp', 'lens')),
    (np.float32, 'nearest', (3, 3), ('map', 'warp')),
    (np.float32, 'linear', (3, 5), ('map', 'warp', 'lens')),     # rectangular
    (np.float32, 'linear', (13, 13), ('map', 'warp')),           # larger than the fused kernels
    (np.float32, 'cubic_cv', (1, 7), ('map',)),
]


@pytest.mark.parametrize('n', [1, 4])
@pytest.mark.parametrize('case', CASES, ids=lambda c: '%s-%s-%dx%d' % (np.dtype(c[0]).name, c[1], c[2][0], c[2][1]))
def test_chain_is_remap_then_filter(ia, oracle, case, n):
    from imgprocessor_amd import ops
    dtype, interp, (kh, kw), sources = case
    ctx = ia.default_context(0)
    h, w = 90, 410
    src = frames(n, h, w, np.float32)
    if dtype != np.float32:
        src = np.round(src * (255 if dtype == np.uint8 else 4095)).astype(dtype)
    mx, my, Kc, dist = radial_maps(h, w)
    k = np.random.default_rng(3).random((kh, kw))
    k /= k.sum()
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    cval = 7.0
    for source in sources:
        if source == 'map':
            got = ops.remap_conv2d(d, dmx, dmy, k, interp, 'constant', cval, 'reflect')
            mid = ops.remap(d, dmx, dmy, interp, 'constant', cval, out_dtype=np.float32)
            omid = lambda f: oracle.remap(src[f], mx, my, oracle_interp(oracle, interp), oracle.CONSTANT, cval, out_dtype=np.float32)  # noqa: E731
        elif source == 'warp':
            got = ops.warp_perspective_conv2d(d, M, (h, w), k, interp, 'constant', cval, 'reflect')
            mid = ops.warp_perspective(d, M, (h, w), interp, 'constant', cval, out_dtype=np.float32)
            omid = lambda f: oracle.warp_perspective(src[f], M, (h, w), oracle_interp(oracle, interp), oracle.CONSTANT, cval,  # noqa: E731
                                                     out_dtype=np.float32)
        else:
            got = ops.undistort_conv2d(d, Kc, dist, Kc, k, interp, 'constant', cval, 'reflect')
            mid = ops.undistort(d, Kc, dist, Kc, interp, 'constant', cval, out_dtype=np.float32)
            omid = lambda f: oracle.remap(src[f], mx, my, oracle_interp(oracle, interp), oracle.CONSTANT, cval, out_dtype=np.float32)  # noqa: E731
        two = ops.conv2d(mid, k, 'reflect').get()
        got = got.get()
        assert got.dtype == np.float32 and got.shape == two.shape
        same_bits(got.reshape(two.shape), two, '%s chain against the two calls' % source)
        f = n - 1
        want = oracle.conv2d(omid(f), k, 'reflect')
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got.reshape((n, h, w))[f] - want).max() <= 1e-5 * scale, source


def oracle_interp(oracle, name):
    return {'nearest': oracle.NEAREST, 'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS, 'cubic_cv': oracle.CUBIC_CV,
            'linear_cv_q5': oracle.LINEAR | oracle.Q5, 'lanczos4': oracle.LANCZOS4}[name]


def test_what_the_two_calls_reject_is_still_rejected(ia):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    src = ctx.to_device(frames(1, 40, 80))
    mx, my, _, _ = radial_maps(40, 80)
    with pytest.raises(ValueError):
        ops.remap_conv2d(src, ctx.to_device(mx), ctx.to_device(my), kern(3), 'bogus')
    with pytest.raises(ValueError):
        ops.remap_conv2d(src, ctx.to_device(mx), ctx.to_device(my), kern(3), 'linear', 'bogus')
    f64 = ctx.to_device(frames(1, 40, 80).astype(np.float64))     # (float64 frames: remap -> float32 is not a conversion
    with pytest.raises((NotImplementedError, ValueError, TypeError)):   #  the standalone entry point makes either)
        ops.remap_conv2d(f64, ctx.to_device(mx), ctx.to_device(my), kern(3))


@pytest.mark.parametrize('n', [1, 4, 7, 8])
@pytest.mark.parametrize('K', [3, 5, 7])
def test_uint8_frames_dense_chain_in_one_kernel(ia, oracle, K, n):
    """8-bit camera frames + maps + a dense K x K kernel (round 6, the last addition): the chain kernels of the uint16
    frames with 16-bit tap loads; against the oracle, and against remap -> filter as two calls within summation order"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 150, 610
    src = np.round(frames(n, h, w, np.float32) * 255).astype(np.uint8)
    mx, my, Kc, dist = radial_maps(h, w, shift=3.3)
    mx = mx - np.float32(20.0)
    k = np.random.default_rng(8).random((K, K))
    k /= k.sum()
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    for border, cmode in (('constant', 'reflect'), ('replicate', 'constant'), ('reflect', 'wrap')):
        got = ops.remap_conv2d(d, dmx, dmy, k, 'linear', border, 9.0, cmode).get().reshape((n, h, w))
        got_l = ops.undistort_conv2d(d, Kc, dist, Kc, k, 'linear', border, 9.0, cmode).get().reshape((n, h, w))
        two = ops.conv2d(ops.remap(d, dmx, dmy, 'linear', border, 9.0, out_dtype=np.float32), k, cmode).get().reshape((n, h, w))
        assert np.abs(got - two).max() <= 2e-6 * 255
        bo = {'constant': oracle.CONSTANT, 'replicate': oracle.REPLICATE, 'reflect': oracle.REFLECT}[border]
        umx, umy = oracle.build_undistort_map(Kc, dist, Kc, h, w)
        for f in (0, n - 1):
            want = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR, bo, 9.0, out_dtype=np.float32), k, cmode)
            assert np.abs(got[f] - want).max() <= 1e-5 * 255, (K, n, border, f)
            want = oracle.conv2d(oracle.remap(src[f], umx, umy, oracle.LINEAR, bo, 9.0, out_dtype=np.float32), k, cmode)
            assert np.abs(got_l[f] - want).max() <= 1e-5 * 255, (K, n, border, f, 'lens')
