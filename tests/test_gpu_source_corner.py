"""GPU: the footprint on the source's TOP-LEFT corner (column -1, row 0 or -1) under BORDER_CONSTANT - the
reference's border mode (cv2.remap / cv2.warpPerspective as camera/LensDistortion.py:323-326 and
camera/PerspectiveCorrection.py:401-405 call them): its one inside tap is pixel (0, 0) of the frame.

Round 6, tools/fuzz_paths.py seed 63 case 66: the hand-scheduled loops fetch the right-hand tap of a float32 tap
row as `buffer_load_dword ... offen offset:4`; where the tap row starts at byte -4 the hardware range-checks the
unsigned sum without wrapping it and returned 0 for pixel (0, 0) - one or two samples per frame, wherever a map
crosses that corner (0.12 off at a border value of 0.25).  border_tap_bits now sends that footprint through the
tap-by-tap sampler for every element type.  Every chain kernel, frame count and coordinate source below has an
output pixel whose footprint is exactly that one, in every frame; checked against the oracle point by point."""
import numpy as np
import pytest

from .conftest import assert_close
from .gpu_helpers import frames, kern

pytestmark = pytest.mark.gpu

SHIFT_X, SHIFT_Y = 7.3, 5.4      # source (0, 0) lies at output (7.3, 5.4): pixels (7, 5) and (7, 6) straddle it


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def shift_maps(h, w):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    return (x - SHIFT_X).astype(np.float32), (y - SHIFT_Y).astype(np.float32)


def shift_matrix():
    return np.array([[1.0, 0.0, -SHIFT_X], [0.0, 1.0, -SHIFT_Y], [0.0, 0.0, 1.0]])


def gauss(k):
    g = np.exp(-0.5 * (np.arange(k) - k // 2) ** 2)
    return g / g.sum()


def corner_samples_matter(oracle, src0, mx, my, cval):
    """the two samples in question are what this test is about: they must differ from the border value and from
    what a zero in place of pixel (0, 0) would give"""
    w = oracle.remap(src0.astype(np.float32), mx, my, oracle.LINEAR, oracle.CONSTANT, cval)
    z = src0.astype(np.float32).copy()
    z[0, 0] = 0
    wz = oracle.remap(z, mx, my, oracle.LINEAR, oracle.CONSTANT, cval)
    assert abs(w[5, 7] - wz[5, 7]) > 1e-3 * max(1.0, abs(w[5, 7])) and abs(w[6, 7] - wz[6, 7]) > 1e-3 * max(1.0, abs(w[6, 7]))


@pytest.mark.parametrize('n', [1, 3, 4, 8])
@pytest.mark.parametrize('dtype', [np.float32, np.uint16])
def test_plain_remap_and_warp(ia, oracle, n, dtype):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 70, 300
    src = frames(n, h, w, dtype)
    src[:, 0, 0] = src.max()             # a pixel that cannot be mistaken for its neighbours
    cval = 0.25 if dtype == np.float32 else 1000
    mx, my = shift_maps(h, w)
    corner_samples_matter(oracle, src[0], mx, my, float(cval))
    d = ctx.to_device(src)
    got = ops.remap(d, ctx.to_device(mx), ctx.to_device(my), 'linear', 'constant', cval).get()
    gotw = ops.warp_perspective(d, shift_matrix(), (h, w), 'linear', 'constant', cval).get()
    for f in range(n):
        want = oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, float(cval))
        wantw = oracle.warp_perspective(src[f], shift_matrix(), (h, w), oracle.LINEAR, oracle.CONSTANT, float(cval))
        if dtype == np.float32:
            assert_close(got[f], want, 1e-5, 1e-5, 'remap frame %d' % f)
            assert_close(gotw[f], wantw, 1e-5, 1e-5, 'warp frame %d' % f)
        else:
            assert np.array_equal(got[f], want) and np.array_equal(gotw[f], wantw), f


@pytest.mark.parametrize('n', [1, 4, 6, 8])
@pytest.mark.parametrize('K', [3, 5, 7, 9])
@pytest.mark.parametrize('dtype', [np.float32, np.uint16])
def test_chains(ia, oracle, n, K, dtype):
    """map, homography and lens model as coordinate sources; dense and separable filters; one frame (the
    per-frame loop), whole workgroups of frames (the shared-footprint loop) and a ragged count"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 66, 530                        # three strips wide: the corner lies in a rim strip
    src = frames(n, h, w, dtype)
    src[:, 0, 0] = src.max()
    cval = 0.25 if dtype == np.float32 else 1000.0
    mx, my = shift_maps(h, w)
    M = shift_matrix()
    d = ctx.to_device(src)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    dense, g = kern(K), gauss(K)
    scale = float(src.max())
    for what, k in (('dense', dense), ('outer product', np.outer(g, g))):
        got_m = ops.remap_conv2d(d, dmx, dmy, k, 'linear', 'constant', cval).get()
        got_w = ops.warp_perspective_conv2d(d, M, (h, w), k, 'linear', 'constant', cval).get()
        for f in range(n):
            wm = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, cval, out_dtype=np.float32), k)
            assert_close(got_m[f], wm, 1e-5, 1e-5 * scale, 'map + %s %dx%d, frame %d of %d' % (what, K, K, f, n))
            if got_w is not None:
                ww = oracle.conv2d(oracle.warp_perspective(src[f], M, (h, w), oracle.LINEAR, oracle.CONSTANT, cval,
                                                           out_dtype=np.float32), k)
                assert_close(got_w[f], ww, 1e-5, 1e-5 * scale, 'warp + %s %dx%d, frame %d of %d' % (what, K, K, f, n))
    got_s = ops.remap_sepconv2d(d, dmx, dmy, g, g, 'linear', 'constant', cval).get()
    for f in range(n):
        want = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, cval, out_dtype=np.float32),
                             np.outer(g, g))
        assert_close(got_s[f], want, 1e-5, 1e-5 * scale, 'map + separable %d+%d, frame %d of %d' % (K, K, f, n))


# the knob sets that pin a chain to one loop (the defaults choose among them by batch and size)
LOOPS = {
    'default policy': {},
    'shared-footprint loop': dict(ring_remap=0, lens_cache=0, ring_min=1, stored_coords=0, tile_warp=0, frames_wg=1, pipe=1),
    'per-frame loop': dict(ring_remap=0, lens_cache=0, ring_min=1, stored_coords=0, tile_warp=0, frames_wg=0, pipe=1),
    'compiler-scheduled loops': dict(ring_remap=0, lens_cache=0, ring_min=1, stored_coords=0, tile_warp=0, frames_wg=0, pipe=0),
}


@pytest.mark.parametrize('loop', list(LOOPS))
@pytest.mark.parametrize('K', [3, 5, 7])
def test_every_lane_of_the_rim_strip(ia, oracle, loop, K):
    """the geometry the fuzzer found it on (143 x 1052 source, 4 frames, the corner in the leftmost strip), the
    corner moved through 16 columns so that its footprint visits every lane position of a quad and of the strip's
    reflected halo lanes: which lanes lost pixel (0, 0) depended on their neighbours' offsets (column 7: wrong,
    column 1: right).  Round 5's rule (make VARIANT=r5corner DEFS=-DIPA_DEBUG_CORNER_AS_ROUND5 ONLY="fused_k3 fused_k5 ...")
    fails the default policy and the shared-footprint loop at 3x3 and 5x5, first with the corner in column 3."""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, dh = 143, 1052, 120
    src = frames(4, h, w)
    src[:, 0, 0] = 2.0
    d = {4: ctx.to_device(src), 1: ctx.to_device(src[:1])}
    k = kern(K)
    y, x = np.mgrid[0:dh, 0:w].astype(np.float64)
    old = ctx.set_tuning(**LOOPS[loop])
    try:
        for c in range(16):
            for n in (4, 1):
                mx, my = (x - (c + 0.3)).astype(np.float32), (y - 5.4).astype(np.float32)
                got = ops.remap_conv2d(d[n], ctx.to_device(mx), ctx.to_device(my), k, 'linear', 'constant', 0.25).get()
                for f in (0, n - 1):
                    want = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, 0.25), k)
                    assert_close(got[f], want, 1e-5, 2e-5, '%s, %dx%d, corner in column %d, frame %d of %d' % (loop, K, K, c, f, n))
        # the fuzzer's own matrix (a rotation of 2 degrees on top), as a homography
        M = np.array([[1.005, -0.037, -1.459], [0.04, 1.006, -50.734], [0.0, 0.0, 0.997]])
        got = ops.warp_perspective_conv2d(d[4], M, (dh, w), k, 'linear', 'constant', 0.25).get()
        for f in (0, 3):
            want = oracle.conv2d(oracle.warp_perspective(src[f], M, (dh, w), oracle.LINEAR, oracle.CONSTANT, 0.25), k)
            assert_close(got[f], want, 1e-5, 2e-5, '%s, %dx%d, homography, frame %d' % (loop, K, K, f))
    finally:
        ctx.set_tuning(**old)


@pytest.mark.parametrize('loop', list(LOOPS))
def test_last_pixel_of_uint16_frames(ia, oracle, loop):
    """the OPPOSITE corner, uint16 frames (one dword holds both taps of a tap row): the footprint whose left tap is
    the last pixel of the last row - that dword ends two bytes past the frame and the range check can drop it whole,
    the pixel with it (tools/fuzz_corners.py, round 6: 405 of 4095 off in 2 samples per frame, every frame of the
    batch - at 12 x 170 px and 116 x 65 px frames, not at 23 x 170: it depends on where the frame ends, so the
    frame size is swept through every residue of 16 bytes and both failing geometries are here as drawn)."""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    k = kern(3)
    cases = [(12, 170, 71, 340, 12, np.array([[9.99999272e-01, 1.20700418e-03, -4.71179433], [-1.20700418e-03, 9.99999272e-01, -38.236679],
                                                [0.0, 0.0, 1.0]])),
             (116, 65, 160, 108, 8, np.array([[1.04, 0.0, -7.25123345], [0.0, 1.04, -15.95355581], [0.0, 0.0, 1.0]]))]
    for h in (12, 23):
        for w in range(160, 177):
            cases.append((h, w, h + 20, w + 40, 4, np.array([[1.0, 0.0, -7.3], [0.0, 1.0, -5.4], [0.0, 0.0, 1.0]])))
    old = ctx.set_tuning(**LOOPS[loop])
    try:
        for h, w, dh, dw, n, M in cases:
            y, x = np.mgrid[0:dh, 0:dw].astype(np.float64)
            mx = (M[0, 0] * x + M[0, 1] * y + M[0, 2]).astype(np.float32)
            my = (M[1, 0] * x + M[1, 1] * y + M[1, 2]).astype(np.float32)
            dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
            src = frames(n, h, w, np.uint16)
            src[:, -1, -1] = 4095
            d = ctx.to_device(src)
            plain = ops.remap(d, dmx, dmy, 'linear', 'constant', 17).get()
            got = ops.remap_conv2d(d, dmx, dmy, k, 'linear', 'constant', 17.0).get()
            for f in (0, n - 1):
                assert np.array_equal(plain[f], oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, 17.0)), (h, w, n, f)
                mid = oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, 17.0, out_dtype=np.float32)
                assert_close(got[f], oracle.conv2d(mid, k), 1e-5, 0.04, '%s, %d frames of %d x %d px, frame %d' % (loop, n, h, w, f))
    finally:
        ctx.set_tuning(**old)


def test_lens_model_through_the_corner(ia, oracle):
    """LensDistortion with the reference's own camera matrix (alpha = 1 keeps every source pixel: the source's
    corners lie INSIDE the undistorted picture) - the rim the constant border fills passes all four corners"""
    from imgprocessor_amd import ops
    from imgprocessor_amd.utils.geometry import getOptimalNewCameraMatrix
    ctx = ia.default_context(0)
    h, w = 240, 520
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.2, 0.05, 1e-3, -5e-4, 0.0])
    newK, _ = getOptimalNewCameraMatrix(K, dist, (w, h), 1)
    src = frames(4, h, w)
    src[:, 0, 0] = src[:, 0, -1] = src[:, -1, 0] = src[:, -1, -1] = 3.0
    mx, my = oracle.build_undistort_map(K, dist, newK, h, w)
    k5 = kern(5)
    got = ops.undistort_conv2d(ctx.to_device(src), K, dist, newK, k5, 'linear', 'constant', 0.25).get()
    plain = ops.undistort(ctx.to_device(src), K, dist, newK, 'linear', 'constant', 0.25).get()
    for f in range(4):
        w0 = oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, 0.25)
        assert_close(plain[f], w0, 1e-5, 3e-5, 'undistort frame %d' % f)
        assert_close(got[f], oracle.conv2d(w0, k5), 1e-5, 3e-5, 'undistort + 5x5 frame %d' % f)
