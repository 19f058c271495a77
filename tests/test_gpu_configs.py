"""GPU: the secondary stencils against the reference fixtures, and the
BASELINE.json configurations C2..C5 at FULL size — against the oracle directly where it
finishes in seconds (it is plain C with OpenMP) and through size-independent properties.
"""
import os

import numpy as np
import pytest

from .conftest import load_golden, assert_close, synth

pytestmark = pytest.mark.gpu

RT = 1e-5


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


@pytest.fixture(scope='module')
def orc(oracle):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    oracle.set_threads(max(1, min(n, oracle.max_threads(), 32)))
    yield oracle
    oracle.set_threads(1)


def close32(got, want, what='', scale=None):
    want = np.asarray(want, dtype=np.float64)
    s = np.nanmax(np.abs(want)) if scale is None else scale
    assert_close(got, want, RT, RT * s, what)


def close32p(got, want, what='', scale=None):
    """close32 + the POINTWISE bound of the north star: every pixel within 1e-5 relative to
    max(|ref|, 1e-3 max|ref|) - close32's absolute term alone would pass a 1e-5 max|ref| error on
    a dark pixel (the C1 - C4 kernels are positive: no cancellation to excuse it)"""
    close32(got, want, what, scale)
    want = np.asarray(want, dtype=np.float64)
    floor = 1e-3 * float(np.nanmax(np.abs(want)))
    rel = np.abs(np.asarray(got, dtype=np.float64) - want) / np.maximum(np.abs(want), floor)
    worst = float(np.nanmax(rel))
    assert worst <= RT, '%s: pointwise relative error %.3g > %g at %s' % (
        what, worst, RT, np.unravel_index(np.nanargmax(rel), rel.shape))


# ------------------------------------------------------- a9 / a10 stencils ----
def test_var_y_gauss_golden(ia):
    from imgprocessor_amd.filters import varYSizeGaussianFilter
    g = load_golden('var_y_gauss.npz')
    assert_close(varYSizeGaussianFilter(g['arr'], (0, 4), 1), g['out_0_4_1'], 1e-12, 1e-14)
    assert_close(varYSizeGaussianFilter(g['arr'], 3, 0), g['out_3_0'], 1e-12, 1e-14)
    assert_close(varYSizeGaussianFilter(g['arr_nan'], (0, 4), 1), g['out_nan_0_4_1'], 1e-12, 1e-14)
    assert_close(varYSizeGaussianFilter(g['arr'], (1, 3), 2, modex='reflect'),
                 g['out_1_3_2_reflect'], 1e-12, 1e-14)
    a32 = g['arr'].astype(np.float32)
    got = varYSizeGaussianFilter(a32, (0, 4), 1)
    assert got.dtype == np.float32
    close32(got, g['out_0_4_1'], 'f32', scale=1.0)
    with pytest.raises(UnboundLocalError):  # the reference's ndarray branch is broken
        varYSizeGaussianFilter(g['arr'], np.ones(40))
    # the device-built separable tables against the host tables through the generic kernel
    # (arbitrary-table path), on a frame with several workgroups per axis, NaNs included
    from imgprocessor_amd.filters.varYSizeGaussianFilter import _row_kernels
    big = synth((301, 700), 9, np.float64)
    big[17, 100:140] = np.nan
    big[200:204, 650:] = np.nan
    for dt, tol in ((np.float64, 1e-13), (np.float32, 1e-6)):
        b = big.astype(dt)
        for (rng, stdx, modex) in (((0, 10), 1, 'wrap'), ((2, 6), 0, 'reflect'), ((0, 3), 2.5, 'wrap')):
            mn, mx = rng
            kx = int(stdx * 2.5); kx += 1 - kx % 2
            ky = int(mx * 2.5); ky += 1 - ky % 2
            tab = _row_kernels(np.linspace(mn, mx, b.shape[0]), stdx, ky, kx)
            want = ia.ops.conv_ydep(b, tab, modex=modex)
            got = varYSizeGaussianFilter(b, rng, stdx, modex=modex)
            assert got.dtype == dt
            assert_close(got, want, tol, tol, 'device tables %s %s %s' % (dt.__name__, rng, stdx))


def test_var_y_gauss_large_window(ia, orc):
    """stdyrange beyond the LDS tile of the separable kernel (ky > 57 float32 / > 27 float64):
    the whole table is expanded on the device and the generic kernel runs it; the reference
    accepts any stdyrange (filters/varYSizeGaussianFilter.py:9-50)"""
    from imgprocessor_amd.filters import varYSizeGaussianFilter
    img = synth((150, 300), 31, np.float64)
    img[40, 100:120] = np.nan
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
        for rng, stdx in (((0, 30), 1), (30, 0), ((5, 12), 2)):
            b = img.astype(dt)
            got = varYSizeGaussianFilter(b, rng, stdx)
            assert got.dtype == dt and got.shape == b.shape
            want = orc.varYSizeGaussianFilter(b, rng, stdx)
            assert_close(got, want, tol, tol, 'large window %s %s %s' % (dt.__name__, rng, stdx))


def test_std2d_golden(ia):
    from imgprocessor_amd.filters import standardDeviation2d
    g = load_golden('std2d.npz')
    for k in (5, 11):
        assert_close(standardDeviation2d(g['img'], k), g['std_k%d' % k], 1e-10, 1e-13, 'k%d' % k)
    close32(standardDeviation2d(g['img32'], 5), g['std32_k5'], 'f32')
    ctx = ia.default_context(0)
    d = standardDeviation2d(ctx.to_device(g['img']), 5)
    assert isinstance(d, ia.DeviceArray)
    assert_close(d.get(), g['std_k5'], 1e-10, 1e-13)


def test_long_separable_kernels(ia, orc):
    """sigma 11 -> 89 taps: the two-launch fallback of ipa_sepconv2d_dev"""
    img = synth((150, 210), 3)
    from imgprocessor_amd.filters import gaussian_filter
    close32(gaussian_filter(img, 11), orc.gaussian_filter(img, 11), 'sigma 11')
    close32(gaussian_filter(img, (0, 9)), orc.gaussian_filter(img, (0, 9)), 'x only, sigma 9')
    i64 = img.astype(np.float64)
    assert_close(gaussian_filter(i64, 6), orc.gaussian_filter(i64, 6), 1e-12, 1e-13, 'f64 sigma 6')


# ------------------------------------------------------------ full-size configs ----
def camera(h, w):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    return K, np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])


def gauss(n, sigma=1.0):
    g = np.exp(-0.5 * (np.arange(n) - n // 2) ** 2 / sigma ** 2)
    return g / g.sum()


def persp(quad, h, w):
    from imgprocessor_amd.utils import getPerspectiveTransform
    dst = np.array([[0, 0], [w, 0], [w, h], [0, h]], float)
    return np.linalg.inv(getPerspectiveTransform(np.array(quad, float), dst))


def test_c1_512_masked_convolve_box3(ia, orc):
    """C1 (the reference's own CPU-runnable case): 512x512 float32, maskedConvolve with
    ones(3,3)/9 and an all-true mask == filter(img, box3) == scipy uniform_filter(3, 'reflect')
    (SURVEY §8c probe: 5.6e-16 against the reference itself)"""
    import io
    import contextlib
    import scipy.ndimage as ndi
    from imgprocessor_amd.filters import maskedConvolve, filter as ipa_filter
    img = synth((512, 512), 0)
    box = np.ones((3, 3)) / 9
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints the padded shape
        got = maskedConvolve(img, box, np.ones(img.shape, bool))
    assert got.dtype == np.float32 and got.shape == img.shape
    close32p(got, orc.maskedConvolve(img, box, np.ones(img.shape, bool)), 'C1 vs oracle', scale=1.0)
    close32p(got, ndi.uniform_filter(img.astype(np.float64), 3, mode='reflect'), 'C1 vs scipy',
            scale=1.0)
    assert np.array_equal(got, ipa_filter(img, box))  # the identity of SURVEY §8b
    half = np.zeros(img.shape, bool)
    half[:, :256] = True
    with contextlib.redirect_stdout(io.StringIO()):
        m = maskedConvolve(img, box, half)
    assert np.array_equal(m[:, :256], got[:, :256]) and not m[:, 256:].any()


def test_c2_1080p_undistort_gauss5(ia, orc):
    """C2: 1080p float32, LensDistortion radial undistort + 5x5 Gaussian"""
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    from imgprocessor_amd.filters import filter as ipa_filter
    h, w = 1080, 1920
    img = synth((h, w), 0)
    K, d = camera(h, w)
    k5 = np.outer(gauss(5), gauss(5))
    ld = LensDistortion(newCameraMatrix='same')
    ld.setCameraParams(K[0, 0], K[1, 1], K[0, 2], K[1, 2], d[0], d[1], d[4], d[2], d[3])
    und = ld.correct(img, keepSize=True)
    out = ipa_filter(und, k5)
    mx, my = orc.build_undistort_map(K, d, K, h, w)
    assert np.array_equal(np.stack(ld.getUndistortRectifyMap(w, h)), np.stack([mx, my]))
    want_u = orc.remap(img, mx, my)
    close32p(und, want_u, 'C2 undistort', scale=1.0)
    close32p(out, orc.conv2d(want_u, k5), 'C2 filter', scale=1.0)
    ctx = ia.default_context(0)
    fused = ia.ops.remap_conv2d(ctx.to_device(img), ctx.to_device(mx), ctx.to_device(my), k5).get()
    close32p(fused, orc.conv2d(want_u, k5), 'C2 fused', scale=1.0)
    an = ia.ops.undistort_conv2d(ctx.to_device(img), K, d, K, k5).get()
    assert np.array_equal(an, fused)  # analytic == map-based, bit for bit
    # size-independent properties: linearity in the image, constant image stays constant inside
    img2 = synth((h, w), 1)
    a, b = 0.75, -1.25
    lhs = ipa_filter(ld.correct((a * img + b * img2).astype(np.float32), keepSize=True), k5)
    rhs = a * out + b * ipa_filter(ld.correct(img2, keepSize=True), k5)
    assert np.abs(lhs - rhs).max() < 5e-6
    const = ipa_filter(ld.correct(np.full((h, w), 0.625, np.float32), keepSize=True), k5)
    inner = (mx > 2) & (mx < w - 3) & (my > 2) & (my < h - 3)
    inner[:3] = inner[-3:] = False
    inner[:, :3] = inner[:, -3:] = False
    import scipy.ndimage as ndi
    inner = ndi.binary_erosion(inner, iterations=3)
    assert np.abs(const[inner] - 0.625).max() < 2e-6


def test_c3_4k_perspective_sep9(ia, orc):
    """C3: 4K float32, perspective warp (bilinear and bicubic) + separable 9x9 Gaussian"""
    h, w = 2160, 3840
    img = synth((h, w), 0)
    M = persp([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], h, w)
    g9 = gauss(9)
    ctx = ia.default_context(0)
    d_img = ctx.to_device(img)
    for interp, iid in (('linear', orc.LINEAR), ('cubic', orc.CUBIC_KEYS)):
        warped = ia.ops.warp_perspective(d_img, M, (h, w), interp)
        want_w = orc.warp_perspective(img, M, (h, w), iid)
        # (the pointwise bound where the weights are positive: a bicubic / Lanczos4 sample that
        # cancels to ~0 carries the float32 rounding of its O(1) terms - 7e-8 absolute here)
        (close32p if interp == 'linear' else close32)(warped.get(), want_w, 'C3 warp ' + interp, scale=1.0)
        out = ia.ops.sepconv2d(warped, g9, g9).get()
        close32p(out, orc.sepconv2d(want_w, g9, g9), 'C3 sep9 ' + interp, scale=1.0)
        # the chain as ONE kernel (remap -> separable 9+9)
        fused = ia.ops.warp_perspective_sepconv2d(d_img, M, (h, w), g9, g9, interp).get()
        close32p(fused, orc.sepconv2d(want_w, g9, g9), 'C3 fused sep9 ' + interp, scale=1.0)
        assert np.abs(fused - out).max() < 2e-6
        # the same filter as a dense 9x9 (fused chain, K=9 -> two launches inside)
        dense = ia.ops.warp_perspective_conv2d(d_img, M, (h, w), np.outer(g9, g9), interp).get()
        assert np.abs(dense - out).max() < 5e-6
    # PerspectiveCorrection class at full size, default Lanczos4
    from imgprocessor_amd.camera.PerspectiveCorrection import PerspectiveCorrection
    pc = PerspectiveCorrection(img.shape, new_size=(h, w))
    pc.setReference([(192, 108), (3648, 54), (3744, 2106), (96, 2052)])
    got = pc.correct(img)
    close32(got, orc.warp_perspective(img, np.linalg.inv(pc.homography), (h, w), orc.LANCZOS4),
            'C3 lanczos4', scale=1.0)


def test_c4_u16_batch_undistort_k7(ia, orc):
    """C4 (a slice of it): 4K uint16 -> float32 frames, undistort + dense 7x7, batch sharded"""
    from imgprocessor_amd.sharding import frame_block
    h, w, n = 2160, 3840, 4
    K, d = camera(h, w)
    k7 = np.random.default_rng(123).random((7, 7))
    k7 /= k7.sum()
    frames = np.stack([np.round(synth((h, w), s, np.float64) * 4095).astype(np.uint16)
                       for s in range(n)])
    ctx = ia.default_context(0)
    dmx, dmy = ia.ops.build_undistort_map(K, d, K, h, w, device=True)
    mx, my = dmx.get(), dmy.get()
    outs = []
    for rank in range(2):  # two "ranks" of the batch sharder on one device
        a, b = frame_block(n, 2, rank)
        d_fr = ctx.to_device(frames[a:b])
        outs.append(ia.ops.remap_conv2d(d_fr, dmx, dmy, k7).get())
    got = np.concatenate(outs)
    assert got.dtype == np.float32 and got.shape == (n, h, w)
    for i in (0, n - 1):
        want = orc.conv2d(orc.remap(frames[i], mx, my, out_dtype=np.float32), k7)
        close32p(got[i], want, 'C4 frame %d' % i)
    # one launch over the whole batch == per-block launches
    whole = ia.ops.remap_conv2d(ctx.to_device(frames), dmx, dmy, k7).get()
    assert np.array_equal(whole, got)


def test_c5_8k_bicubic_k11(ia, orc):
    """C5: 8K float32, bicubic warp under rotation+perspective, dense 11x11 (LDS-pressure case)"""
    h, w = 4320, 7680
    img = synth((h, w), 0)
    a = np.deg2rad(7.0)
    cx, cy = (w - 1) / 2, (h - 1) / 2
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.]])
    T = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.]])
    P = np.array([[1, 0, 0], [0, 1, 0], [1.5e-6, -1e-6, 1.]])
    M = T @ R @ P @ np.linalg.inv(T)
    k11 = np.random.default_rng(321).random((11, 11))
    k11 /= k11.sum()
    ctx = ia.default_context(0)
    d_img = ctx.to_device(img)
    got = ia.ops.warp_perspective_conv2d(d_img, M, (h, w), k11, 'cubic').get()
    want_w = orc.warp_perspective(img, M, (h, w), orc.CUBIC_KEYS)
    # Why close32 (1e-5 relative + 1e-5 of the image's scale) and not the pointwise close32p here: a
    # bicubic (Keys, a = -0.5) footprint has NEGATIVE lobes - its 16 products cancel in part - and 121
    # more float32 products follow; against the oracle's double sums the worst pixel of this frame is
    # 1.1e-5 relative (tests/rel_err.py), 7.9e-7 of the scale.  The chains whose
    # weights are all positive (C1 - C4) are held to the pointwise bound; this one to 2e-5 pointwise
    # on top of close32.
    want = orc.conv2d(want_w, k11)
    close32(got, want, 'C5', scale=1.0)
    floor = 1e-3 * float(np.max(np.abs(want)))
    worst = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))
    assert worst <= 2e-5, 'C5: max pointwise relative error %.3g > 2e-5' % worst
    # row-band split (single huge frame over G GPUs): bands of OUTPUT rows with an
    # 11//2 halo computed from the same source equal the monolithic result
    bands = []
    G = 4
    for r in range(G):
        y0, y1 = r * h // G, (r + 1) * h // G
        lo, hi = max(y0 - 5, 0), min(y1 + 5, h)
        Mb = M @ np.array([[1, 0, 0], [0, 1, lo], [0, 0, 1.]])  # output row offset
        part = ia.ops.conv2d(ia.ops.warp_perspective(d_img, Mb, (hi - lo, w), 'cubic'), k11).get()
        bands.append(part[y0 - lo:y0 - lo + (y1 - y0)])
    banded = np.concatenate(bands)
    # (rows within 5 of a band's own edge see that band's reflect border, and are cropped away)
    assert np.abs(banded - got).max() < 5e-6


# ------------------------------------------------- transform/ wrappers (§8 f3) ----
def test_transform_wrappers(ia, orc):
    from imgprocessor_amd.transform import (rotate, simplePerspectiveTransform, linearToPolar,
                                            polarToLinear, linearToPolarMaps)
    from imgprocessor_amd.transform.rotate import rotation_matrix_2d
    # rotate: the reference's own check (transform/rotate.py:26-32)
    a = np.tile(np.linspace(0, 1, 50), (50, 1))
    c = rotate(rotate(a, 14), -14)
    assert np.abs(a - c).mean() < 0.005
    # non-square image: (width, height) = (s0, s1) as the reference passes image.shape to cv2
    img = synth((60, 90), 4)
    r = rotate(img, 30)
    assert r.shape == (90, 60)
    M = np.vstack([rotation_matrix_2d((59 / 2., 89 / 2.), 30), [0, 0, 1.]])
    close32(r, orc.warp_perspective(img, np.linalg.inv(M), (90, 60), orc.CUBIC_CV | orc.Q5,
                                    orc.REFLECT), 'rotate', scale=1.0)
    # simplePerspectiveTransform: quad -> rectangle of the average edge lengths, and back
    quad = np.array([(8, 2), (80, 6), (82, 55), (5, 57)], float)
    out = simplePerspectiveTransform(img, quad)
    assert out.shape == (52, 75)
    from imgprocessor_amd.utils import getPerspectiveTransform
    H = getPerspectiveTransform(quad, [[0, 0], [75, 0], [75, 52], [0, 52]])
    close32(out, orc.warp_perspective(img, np.linalg.inv(H), (52, 75)), 'simplePersp', scale=1.0)
    out2 = simplePerspectiveTransform(img, quad, shape=(40, 64), interpolation='cubic')
    assert out2.shape == (40, 64)
    # polar maps: the reference's round trip (transform/polarTransform.py:121-127) in spirit:
    # concentric rings survive linear -> polar -> linear
    y, x = np.mgrid[0:257, 0:257]
    rings = (0.5 + 0.5 * np.cos(np.hypot(x - 128, y - 128) / 6.0)).astype(np.float32)
    pol = linearToPolar(rings)
    mY, mX = linearToPolarMaps(rings.shape)
    assert pol.shape == mY.shape
    close32(pol, orc.remap(rings, mY, mX, orc.LINEAR, orc.REFLECT), 'linearToPolar', scale=1.0)
    # rings are (nearly) constant along phi in the polar image
    # (radii < 120 px: beyond that the circle leaves the 257 px image and BORDER_REFLECT fills in)
    assert np.abs(pol - pol.mean(axis=1, keepdims=True))[5:120].max() < 0.12
    back = polarToLinear(pol, shape=rings.shape)
    assert back.shape == rings.shape


def test_strip_geometry_sweep(ia, orc):
    """ragged shapes around the strip geometry of the marching kernels (248-px strip step,
    256-px strip width, 32-row strips, chunked rows): filter, separable filter and the fused
    chains against the oracle, single frames and small batches"""
    ctx = ia.default_context(0)
    rng = np.random.default_rng(17)
    widths = (1, 3, 247, 248, 249, 256, 257, 495, 496, 497, 505, 760)
    heights = (1, 2, 5, 31, 32, 33, 37, 65)
    cases = [(h, w) for h in heights for w in widths]
    rng.shuffle(cases)
    for n, (h, w) in enumerate(cases):
        img = synth((h, w), n)
        ksz = (3, 5, 7)[n % 3]
        k = rng.random((ksz, ksz))
        k /= k.sum()
        mode = ('reflect', 'constant', 'wrap', 'mirror', 'nearest')[n % 5]
        close32(ia.ops.conv2d(img, k, mode, 0.25), orc.conv2d(img, k, mode, 0.25),
                'conv %dx%d k%d %s' % (h, w, ksz, mode))
        g = rng.random(ksz)
        g /= g.sum()
        close32(ia.ops.sepconv2d(img, g, g, mode, 0.25), orc.sepconv2d(img, g, g, mode, 0.25),
                'sep %dx%d k%d %s' % (h, w, ksz, mode))
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        mx = (xx + 1.7 * np.sin(yy / 9 + 0.3) - 0.4).astype(np.float32)
        my = (yy + 1.3 * np.cos(xx / 11) + 0.2).astype(np.float32)
        want_r = orc.remap(img, mx, my, orc.LINEAR, orc.CONSTANT, 0.1)
        d_img, dmx, dmy = ctx.to_device(img), ctx.to_device(mx), ctx.to_device(my)
        close32(ia.ops.remap_conv2d(d_img, dmx, dmy, k, 'linear', 'constant', 0.1, mode).get(),
                orc.conv2d(want_r, k, mode), 'fused %dx%d k%d %s' % (h, w, ksz, mode))
        close32(ia.ops.remap_sepconv2d(d_img, dmx, dmy, g, g, 'linear', 'constant', 0.1,
                                       mode).get(),
                orc.sepconv2d(want_r, g, g, mode), 'fused sep %dx%d k%d %s' % (h, w, ksz, mode))
        close32(ia.ops.remap(img, mx, my, 'linear', 'constant', 0.1), want_r,
                'remap %dx%d' % (h, w), scale=1.0)
    # batches of odd sizes through the frame-inner dispatch order
    for nb, (h, w) in ((3, (37, 505)), (5, (65, 760)), (2, (33, 256))):
        batch = np.stack([synth((h, w), 50 + i) for i in range(nb)])
        k = rng.random((5, 5))
        k /= k.sum()
        got = ia.ops.conv2d(batch, k)
        for i in range(nb):
            close32(got[i], orc.conv2d(batch[i], k), 'batch conv %d' % i)


def test_tall_strips_of_large_launches(ia, orc):
    """launches of many frames march in 48- and 72-row strips (wave_strip_height); the same
    geometry forced on small frames through the context's strip_h knob, against the oracle"""
    ctx = ia.default_context(0)
    rng = np.random.default_rng(23)
    old = ctx.set_tuning(strip_h=0)
    try:
        for sh in (48, 64, 72):
            ctx.set_tuning(strip_h=sh)
            assert ctx.get_tuning('strip_h') == sh
            for n, (h, w) in enumerate(((sh - 1, 249), (sh, 505), (sh + 1, 256), (2 * sh + 3, 760))):
                img = synth((h, w), 70 + n)
                ksz = {48: (3, 5, 7, 9), 64: (9, 11, 9, 11), 72: (7, 3, 5, 11)}[sh][n]
                k = rng.random((ksz, ksz))
                k /= k.sum()
                g = rng.random(ksz)
                g /= g.sum()
                mode = ('reflect', 'constant', 'wrap', 'nearest')[n]
                close32(ia.ops.conv2d(img, k, mode, 0.25), orc.conv2d(img, k, mode, 0.25),
                        'conv strip %d %dx%d' % (sh, h, w))
                close32(ia.ops.sepconv2d(img, g, g, mode, 0.25), orc.sepconv2d(img, g, g, mode, 0.25),
                        'sep strip %d %dx%d' % (sh, h, w))
                yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
                mx = (xx + 1.7 * np.sin(yy / 9 + 0.3) - 0.4).astype(np.float32)
                my = (yy + 1.3 * np.cos(xx / 11) + 0.2).astype(np.float32)
                want_r = orc.remap(img, mx, my, orc.LINEAR, orc.CONSTANT, 0.1)
                d_img, dmx, dmy = ctx.to_device(img), ctx.to_device(mx), ctx.to_device(my)
                close32(ia.ops.remap_conv2d(d_img, dmx, dmy, k, 'linear', 'constant', 0.1,
                                            mode).get(),
                        orc.conv2d(want_r, k, mode), 'fused strip %d %dx%d' % (sh, h, w))
                close32(ia.ops.remap_sepconv2d(d_img, dmx, dmy, g, g, 'linear', 'constant', 0.1,
                                               mode).get(),
                        orc.sepconv2d(want_r, g, g, mode), 'fused sep strip %d %dx%d' % (sh, h, w))
    finally:
        ctx.set_tuning(**old)
    assert ctx.get_tuning('strip_h') == old['strip_h']


def test_big_kernels_on_the_marching_wave(ia, orc):
    """9x9 / 11x11: the plain filter and the map-based bilinear remap -> filter chain in one
    kernel (coefficient rows streamed through SGPRs), batches through the frame-inner order"""
    ctx = ia.default_context(0)
    rng = np.random.default_rng(29)
    h, w = 150, 700
    batch = np.stack([synth((h, w), 90 + i) for i in range(3)])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    mx = (xx + 2.1 * np.sin(yy / 13 + 0.3) - 0.6).astype(np.float32)
    my = (yy + 1.6 * np.cos(xx / 17) + 0.4).astype(np.float32)
    d_b, dmx, dmy = ctx.to_device(batch), ctx.to_device(mx), ctx.to_device(my)
    for ksz, mode in ((9, 'reflect'), (11, 'constant'), (11, 'wrap'), (9, 'nearest')):
        k = rng.random((ksz, ksz)) - 0.2
        got_c = ia.ops.conv2d(d_b, k, mode, 0.5).get()
        got_f = ia.ops.remap_conv2d(d_b, dmx, dmy, k, 'linear', 'constant', 0.1, mode).get()
        for i in range(3):
            close32(got_c[i], orc.conv2d(batch[i], k, mode, 0.5), 'conv %d %s #%d' % (ksz, mode, i))
            want_r = orc.remap(batch[i], mx, my, orc.LINEAR, orc.CONSTANT, 0.1)
            close32(got_f[i], orc.conv2d(want_r, k, mode), 'fused %d %s #%d' % (ksz, mode, i))


def test_single_frame_row_bands(ia, orc):
    """C5 alternative: one frame split into output row bands, one band per GPU,
    no collective — the concatenated bands equal the whole-frame chain bit for bit"""
    from imgprocessor_amd.sharding import row_bands, remap_filter_band
    h, w = 601, 1100
    img = synth((h, w), 3)
    K, d = camera(h, w)
    ctx = ia.default_context(0)
    d_img = ctx.to_device(img)
    dmx, dmy = ia.ops.build_undistort_map(K, d, K, h, w, device=True)
    for ksz, interp in ((5, 'linear'), (11, 'cubic')):
        k = np.random.default_rng(ksz).random((ksz, ksz))
        k /= k.sum()
        whole = ia.ops.conv2d(ia.ops.remap(d_img, dmx, dmy, interp), k).get()
        for g in (2, 3, 8):
            parts = [remap_filter_band(d_img, dmx, dmy, k, band, interp)
                     for band in row_bands(h, g, halo=ksz // 2)]
            got = np.concatenate([p.get() for p in parts if p is not None])
            assert np.array_equal(got, whole), (ksz, g)
    mx, my = dmx.get(), dmy.get()
    close32(whole, orc.conv2d(orc.remap(img, mx, my, orc.CUBIC_KEYS), k), 'bands vs oracle',
            scale=1.0)
    with pytest.raises(ValueError):   # halo too small for the kernel
        remap_filter_band(d_img, dmx, dmy, k, row_bands(h, 2, halo=1)[0])


def test_frame_pipeline_end_to_end(ia, orc):
    """host-to-host streaming with pinned arrays and overlapped workers == frame-by-frame calls"""
    from imgprocessor_amd.sharding import FramePipeline
    h, w, n = 270, 480, 7
    K, d = camera(h, w)
    k5 = np.outer(gauss(5), gauss(5))
    pipe = FramePipeline(0, depth=3)
    maps = {id(c): ia.ops.build_undistort_map(K, d, K, h, w, ctx=c, device=True)
            for c in pipe.contexts}
    fin = pipe.pinned_empty((n, h, w), np.uint16)
    fout = pipe.pinned_empty((n, h, w), np.float32)
    for i in range(n):
        fin[i] = np.round(synth((h, w), i, np.float64) * 4095).astype(np.uint16)

    def fn(c, d_in, d_out):
        mx, my = maps[id(c)]
        ia.ops.remap_conv2d(d_in, mx, my, k5, out=d_out)
    assert pipe.run(fin, fout, fn) is fout
    mx, my = orc.build_undistort_map(K, d, K, h, w)
    for i in (0, 3, n - 1):
        want = orc.conv2d(orc.remap(np.array(fin[i]), mx, my, out_dtype=np.float32), k5)
        close32(fout[i], want, 'pipeline frame %d' % i)
    # pageable lists of frames work as well
    outs = [np.empty((h, w), np.float32) for _ in range(n)]
    pipe.run([np.array(f) for f in fin], outs, fn)
    assert all(np.array_equal(o, f) for o, f in zip(outs, fout))
    with pytest.raises(ValueError):
        pipe.run(fin, fout[:2], fn)
