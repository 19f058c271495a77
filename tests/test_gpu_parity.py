"""GPU: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Tolerances: float32 images within 1e-5 relative (BASELINE.json north_star;
`rtol` on each element plus an `atol` of 1e-5 x the image's value range so
that cancelling sums are judged against the data scale); float64 images
within 1e-12; uint8 fixed-point and pure index work bit-exact.
"""
import io
import contextlib

import numpy as np
import pytest

from .conftest import load_golden, assert_close, synth, load_cv_golden

pytestmark = pytest.mark.gpu

RT = 1e-5


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)  # raises without a gfx950 device: no fallback
    return imgprocessor_amd


def close32(got, want, what='', scale=None):
    want = np.asarray(want, dtype=np.float64)
    s = np.nanmax(np.abs(want)) if scale is None else scale
    assert_close(got, want, RT, RT * s, what)


# ---------------------------------------------------------------- remap ----
CASES = ('zero', 'radial', 'synthdefault', 'strong')


def test_undistort_map_bit_exact(ia, oracle):
    g = load_golden('remap_scipy.npz')
    H, W = g['img'].shape
    for c in CASES:
        mx, my = ia.ops.build_undistort_map(g['K_' + c], g['dist_' + c], g['newK_' + c], H, W)
        omx, omy = oracle.build_undistort_map(g['K_' + c], g['dist_' + c], g['newK_' + c], H, W)
        assert np.array_equal(mx, omx) and np.array_equal(my, omy), c
        assert_close(mx, g['mapx_' + c], 1.2e-7, 1e-6, 'golden mapx ' + c)
    # non-affine newK (perspective row) exercises the 1/w branch
    nK = np.array([[100., 2, 60], [1, 105., 50], [1e-4, -2e-4, 1.0]])
    mx, my = ia.ops.build_undistort_map(g['K_strong'], g['dist_strong'], nK, 50, 70)
    omx, omy = oracle.build_undistort_map(g['K_strong'], g['dist_strong'], nK, 50, 70)
    assert np.array_equal(mx, omx) and np.array_equal(my, omy)


def test_remap_linear_golden(ia, oracle):
    g = load_golden('remap_scipy.npz')
    img, img16 = g['img'], g['img16']
    for c in CASES:
        mx, my = g['mapx_' + c], g['mapy_' + c]
        for cname, cv in (('c0', 0.0), ('cnan', np.nan), ('c037', 0.37)):
            key = 'lin_%s_%s' % (c, cname)
            if key not in g:
                continue
            got = ia.ops.remap(img, mx, my, 'linear', 'constant', cv)
            close32(got, g[key], key, scale=1.0)
            close32(got, oracle.remap(img, mx, my, oracle.LINEAR, oracle.CONSTANT, cv), key + ' orc',
                    scale=1.0)
        got = ia.ops.remap(img16, mx, my, 'linear', out_dtype=np.float32)
        close32(got, g['lin16_' + c], 'lin16 ' + c)
        # analytic kernel: same float32 coordinates -> identical to the map-based kernel
        an = ia.ops.undistort(img, g['K_' + c], g['dist_' + c], g['newK_' + c])
        mb = ia.ops.remap(img, *ia.ops.build_undistort_map(g['K_' + c], g['dist_' + c],
                                                           g['newK_' + c], *img.shape))
        assert np.array_equal(an, mb), 'analytic != map-based ' + c
    mx, my = g['mapx_strong'], g['mapy_strong']
    for smode, b in (('nearest', 'replicate'), ('reflect', 'reflect'), ('mirror', 'reflect101'),
                     ('grid-wrap', 'wrap')):
        close32(ia.ops.remap(img, mx, my, 'linear', b), g['lin_strong_' + smode], smode)


def test_remap_all_modes_vs_oracle(ia, oracle):
    rng = np.random.default_rng(3)
    H, W = 61, 83  # odd sizes: scalar tails, unaligned rows
    img = synth((H, W), 9)
    yy, xx = np.mgrid[0:70, 0:90].astype(np.float32)
    mx = (xx * 0.93 - 2.2 + 3 * np.sin(yy / 9)).astype(np.float32)
    my = (yy * 0.9 - 3.1 + 2 * np.cos(xx / 7)).astype(np.float32)
    mx[5, 7] = np.nan
    my[9, 3] = np.inf
    mx[11, 11] = 1e7
    interps = {'nearest': oracle.NEAREST, 'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS,
               'cubic_cv': oracle.CUBIC_CV, 'linear_cv_q5': oracle.LINEAR | oracle.Q5,
               'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5, 'lanczos4': oracle.LANCZOS4}
    borders = {'constant': oracle.CONSTANT, 'replicate': oracle.REPLICATE,
               'reflect': oracle.REFLECT, 'wrap': oracle.WRAP, 'reflect101': oracle.REFLECT101}
    for iname, iid in interps.items():
        for bname, bid in borders.items():
            got = ia.ops.remap(img, mx, my, iname, bname, 0.25)
            want = oracle.remap(img, mx, my, iid, bid, 0.25)
            close32(got, want, '%s/%s' % (iname, bname), scale=1.0)
    # float64 images compute in double
    img64 = img.astype(np.float64)
    for iname in ('linear', 'cubic', 'lanczos4'):
        got = ia.ops.remap(img64, mx, my, iname, 'reflect')
        assert got.dtype == np.float64
        assert_close(got, oracle.remap(img64, mx, my, interps[iname], oracle.REFLECT), 1e-12, 1e-12,
                     'f64 ' + iname)
    # integer destinations are computed in double like the oracle, then rounded half-even and
    # saturated (cv::saturate_cast): bit-exact for every interpolation and border mode
    u16 = rng.integers(0, 4096, (H, W), dtype=np.uint16)
    for iname, iid in interps.items():
        for bname, bid in (('constant', oracle.CONSTANT), ('reflect', oracle.REFLECT)):
            got = ia.ops.remap(u16, mx, my, iname, bname, 17.5)
            want = oracle.remap(u16, mx, my, iid, bid, 17.5)
            assert got.dtype == np.uint16
            assert np.array_equal(got, want), 'u16->u16 %s/%s' % (iname, bname)
    # uint8 -> float32 ingest
    u8 = rng.integers(0, 256, (H, W), dtype=np.uint8)
    close32(ia.ops.remap(u8, mx, my, 'cubic', out_dtype=np.float32),
            oracle.remap(u8, mx, my, oracle.CUBIC_KEYS, out_dtype=np.float32), 'u8->f32')


def test_remap_frames_narrower_than_the_footprint(ia, oracle):
    """sources of 1 .. 5 pixels across or down: every footprint touches the border, and the tap rows
    of a bicubic / Lanczos4 footprint start before the frame or run past its end even in its
    middle rows (found by tools/fuzz_tile_warp.py on a 7 x 1 source)"""
    rng = np.random.default_rng(12)
    interps = {'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS, 'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5,
               'lanczos4': oracle.LANCZOS4}
    for (h, w) in ((7, 1), (1, 7), (1, 1), (2, 3), (3, 2), (5, 4), (4, 9), (9, 3)):
        img = rng.random((h, w), dtype=np.float32)
        yy, xx = np.mgrid[0:23, 0:37].astype(np.float32)
        mx = (xx * (w + 6) / 37.0 - 3.3 + 0.2 * np.sin(yy)).astype(np.float32)
        my = (yy * (h + 6) / 23.0 - 2.7 + 0.3 * np.cos(xx)).astype(np.float32)
        for iname, iid in interps.items():
            for bname, bid in (('constant', oracle.CONSTANT), ('replicate', oracle.REPLICATE)):
                for src in (img, np.stack([img, img[::-1, ::-1].copy(), img * 0.5])):   # single frame, batch
                    got = ia.ops.remap(src, mx, my, iname, bname, 0.25)
                    frames_ = src if src.ndim == 3 else src[None]
                    for f in range(frames_.shape[0]):
                        want = oracle.remap(frames_[f], mx, my, iid, bid, 0.25)
                        close32(got[f] if src.ndim == 3 else got, want,
                                '%dx%d %s/%s frame %d' % (h, w, iname, bname, f), scale=1.0)


def test_remap_uint8_bit_exact(ia, oracle):
    rng = np.random.default_rng(4)
    H, W = 97, 131
    u8 = rng.integers(0, 256, (H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    mx = (xx + 4 * np.sin(yy / 11) - 1.3).astype(np.float32)
    my = (yy * 1.02 - 2.7).astype(np.float32)
    for bname, bid in (('constant', oracle.CONSTANT), ('reflect', oracle.REFLECT),
                       ('replicate', oracle.REPLICATE)):
        for cv in (0, 17.6):
            got = ia.ops.remap(u8, mx, my, 'linear', bname, cv)
            want = oracle.remap(u8, mx, my, oracle.LINEAR, bid, cv)
            assert got.dtype == np.uint8 and np.array_equal(got, want), (bname, cv)
    assert np.array_equal(ia.ops.remap(u8, xx, yy), u8)
    # "uint8 after rounding": float32 frames remapped INTO an integer array round the double
    # sum once, like the oracle - bit-exact, ties included (values k/2 make ties frequent)
    f = (u8.astype(np.float32) * 0.5 + 3.0).astype(np.float32)
    hx = (xx + 0.5).astype(np.float32)
    for mxx, myy in ((mx, my), (hx, yy)):
        for dt in (np.uint8, np.uint16):
            got = ia.ops.remap(f, mxx, myy, 'linear', out_dtype=dt)
            want = oracle.remap(f, mxx, myy, out_dtype=dt)
            assert got.dtype == dt and np.array_equal(got, want), dt
    # uint8 through the wide interpolations (PerspectiveCorrection.correct's defaults)
    for iname, iid in (('cubic', oracle.CUBIC_KEYS), ('cubic_cv', oracle.CUBIC_CV),
                       ('cubic_cv_q5', oracle.CUBIC_CV | oracle.Q5), ('lanczos4', oracle.LANCZOS4),
                       ('nearest', oracle.NEAREST)):
        for bname, bid in (('constant', oracle.CONSTANT), ('reflect', oracle.REFLECT)):
            got = ia.ops.remap(u8, mx, my, iname, bname, 3.0)
            want = oracle.remap(u8, mx, my, iid, bid, 3.0)
            assert got.dtype == np.uint8 and np.array_equal(got, want), (iname, bname)


def test_uint8_tables_on_edge_shapes(ia, oracle):
    """the uint8 table kernels (bicubic / Lanczos4 weights in LDS, aligned tap dwords) on shapes
    that leave every fast path: sources smaller than a footprint, destination widths that are not
    multiples of 4 / 256, odd pitches, several frames, coordinates far outside, NaN coordinates"""
    rng = np.random.default_rng(77)
    borders = (('constant', oracle.CONSTANT), ('replicate', oracle.REPLICATE),
               ('reflect', oracle.REFLECT), ('wrap', oracle.WRAP),
               ('reflect101', oracle.REFLECT101))
    interps = (('cubic_cv', oracle.CUBIC_CV), ('lanczos4', oracle.LANCZOS4),
               ('linear_cv_q5', oracle.LINEAR | oracle.Q5))
    shapes = (((3, 5), (7, 9)), ((8, 8), (8, 8)), ((9, 257), (9, 257)), ((33, 261), (40, 515)),
              ((130, 513), (129, 1031)), ((21, 1023), (17, 256)), ((6, 2), (5, 300)))
    for (h, w), (dh, dw) in shapes:
        n = 3
        src = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float64)
        mx = (xx * (w / dw) * 1.07 - 2.3 + 1.5 * np.sin(yy / 5.0)).astype(np.float32)
        my = (yy * (h / dh) * 1.05 - 1.9 + 1.5 * np.cos(xx / 9.0)).astype(np.float32)
        mx[0, 0] = np.nan
        my[-1, -1] = 1e9
        mx[dh // 2, dw // 2] = -1e7
        for (iname, iid) in interps:
            for (bname, bid) in borders:
                got = ia.ops.remap(src, mx, my, iname, bname, 7.0)
                for f in range(n):
                    want = oracle.remap(src[f], mx, my, iid, bid, 7.0)
                    assert np.array_equal(got[f], want), (h, w, dh, dw, iname, bname, f)
        # the homography's double coordinates through the same kernels
        M = np.array([[w / dw * 0.97, 0.02, -0.6], [-0.015, h / dh * 1.03, 0.4], [1e-5, -2e-5, 1.0]])
        for (iname, iid) in interps[:2]:
            got = ia.ops.warp_perspective(src, M, (dh, dw), iname, 'reflect')
            for f in range(n):
                want = oracle.warp_perspective(src[f], M, (dh, dw), iid, oracle.REFLECT)
                assert np.array_equal(got[f], want), (h, w, dh, dw, iname, f)


def test_known_answers(ia):
    rng = np.random.default_rng(5)
    img = rng.random((33, 47)).astype(np.float32)
    yy, xx = np.mgrid[0:33, 0:47].astype(np.float32)
    for interp in ('nearest', 'linear', 'cubic', 'cubic_cv', 'lanczos4', 'linear_cv_q5'):
        assert np.array_equal(ia.ops.remap(img, xx, yy, interp), img), interp
    got = ia.ops.remap(img, xx + 3, yy - 2, 'linear', 'constant', 7.0)
    want = np.full_like(img, 7.0)
    want[2:, :-3] = img[:-2, 3:]
    assert np.array_equal(got, want)
    K = np.array([[50., 0, 23], [0, 50., 16], [0, 0, 1]])
    assert np.array_equal(ia.ops.undistort(img, K, np.zeros(5), K), img)
    assert np.array_equal(ia.ops.warp_perspective(img, np.eye(3), img.shape), img)


def test_warp_perspective_golden(ia, oracle):
    g = load_golden('warp_skimage.npz')
    img = g['img']
    for name in ('quad', 'quadb', 'rot7'):
        M = g['M_' + name]
        shp = tuple(int(v) for v in g['shape_' + name])
        for order, interp in ((1, 'linear'), (3, 'cubic')):
            for cname, cv in (('c0', 0.0), ('c05', 0.5)):
                want = g['warp_%s_o%d_%s' % (name, order, cname)]
                got = ia.ops.warp_perspective(img, M, shp, interp, 'constant', cv)
                close32(got, want, '%s o%d %s' % (name, order, cname), scale=1.0)
        got = ia.ops.warp_perspective(img, M, shp, 'linear', 'replicate')
        close32(got, g['warp_%s_o1_edge' % name], name + ' edge')
        got64 = ia.ops.warp_perspective(img.astype(np.float64), M, shp, 'cubic', 'constant', 0.5)
        assert_close(got64, g['warp_%s_o3_c05' % name], 1e-11, 1e-12, name + ' f64')
        for interp, iid in (('lanczos4', oracle.LANCZOS4), ('cubic_cv_q5', oracle.CUBIC_CV | oracle.Q5)):
            close32(ia.ops.warp_perspective(img, M, shp, interp),
                    oracle.warp_perspective(img, M, shp, iid), name + interp, scale=1.0)


def test_batch_and_device_arrays(ia, oracle):
    ctx = ia.default_context(0)
    n, H, W = 5, 72, 100
    frames = np.stack([synth((H, W), s) for s in range(n)])
    K = np.array([[100., 0, 49.5], [0, 100., 35.5], [0, 0, 1]])
    d = np.array([-0.2, 0.05, 1e-3, 2e-3, 0.01])
    mx, my = oracle.build_undistort_map(K, d, K, H, W)
    want = np.stack([oracle.remap(f, mx, my) for f in frames])
    d_fr = ctx.to_device(frames)
    dmx, dmy = ia.ops.build_undistort_map(K, d, K, H, W, device=True)
    got = ia.ops.remap(d_fr, dmx, dmy).get()
    close32(got, want, 'batch remap', scale=1.0)
    got = ia.ops.undistort(d_fr, K, d, K).get()
    close32(got, want, 'batch undistort', scale=1.0)
    assert np.array_equal(ia.ops.remap(frames, mx, my), ia.ops.remap(d_fr, dmx, dmy).get())
    # roi window of the maps == crop of the full result
    full = ia.ops.remap(d_fr, dmx, dmy).get()
    part = ia.ops.remap(d_fr, dmx, dmy, map_roi=(7, 5, 61, 40)).get()
    assert np.array_equal(part, full[:, 5:45, 7:68])
    # uint16 batch -> float32 (C4-style ingest)
    f16 = np.round(frames * 4095).astype(np.uint16)
    got = ia.ops.remap(ctx.to_device(f16), dmx, dmy, out_dtype=np.float32).get()
    want = np.stack([oracle.remap(f, mx, my, out_dtype=np.float32) for f in f16])
    close32(got, want, 'u16 batch')


# -------------------------------------------------------------- filters ----
def test_conv2d_golden(ia, oracle):
    g = load_golden('remap_scipy.npz')
    img = g['img']
    close32(ia.ops.conv2d(img, np.ones((3, 3)) / 9), g['corr_box3'], 'box3')
    close32(ia.ops.conv2d(img, g['k7']), g['corr_k7'], 'k7')
    close32(ia.ops.conv2d(img, g['k11']), g['corr_k11'], 'k11')
    close32(ia.ops.conv2d(img, g['k5']), oracle.conv2d(img, g['k5']), 'k5')
    close32(ia.ops.conv2d(img, g['k7'][:3, :]), g['corr_k3x7'], 'k3x7 (generic path)')
    close32(ia.ops.conv2d(img, g['k7'][:6, :4]), g['corr_k6x4'], 'k6x4 (generic path)')
    for smode in ('nearest', 'mirror', 'wrap', 'constant'):
        close32(ia.ops.conv2d(img, g['k7'], smode, cval=0.25), g['corr_k7_' + smode], smode)
    # 9x9 and per-axis borders against the oracle
    k9 = np.random.default_rng(9).random((9, 9))
    k9 /= k9.sum()
    close32(ia.ops.conv2d(img, k9, 'wrap', mode_y='reflect'),
            oracle.conv2d(img, k9, 'wrap', mode_y='reflect'), 'k9 wrap-x/reflect-y')
    # float64
    i64 = img.astype(np.float64)
    for k in (g['k5'], g['k7'], g['k11']):
        assert_close(ia.ops.conv2d(i64, k), oracle.conv2d(i64, k), 1e-12, 1e-13, 'f64 conv')


def test_conv2d_shapes_and_masks(ia, oracle):
    rng = np.random.default_rng(10)
    for (H, W) in ((1, 1), (3, 5), (31, 33), (32, 128), (33, 129), (130, 70)):
        img = rng.random((H, W)).astype(np.float32)
        for K in (3, 5):
            if K // 2 >= min(H, W) and (H, W) != (1, 1):
                continue
            k = rng.random((K, K))
            got = ia.ops.conv2d(img, k, 'reflect')
            close32(got, oracle.conv2d(img, k, 'reflect'), 'shape %s k%d' % ((H, W), K))
    img = synth((70, 90), 1)
    mask = rng.random((70, 90)) < 0.3
    k = rng.random((5, 5))
    got = ia.ops.conv2d(img, k, mask=mask)
    want = oracle.conv2d(img, k, mask=mask)
    close32(got, want, 'masked')
    assert np.all(got[~mask] == 0)
    # batch
    fr = np.stack([synth((40, 52), s) for s in range(3)])
    got = ia.ops.conv2d(fr, k)
    for i in range(3):
        close32(got[i], oracle.conv2d(fr[i], k), 'batch %d' % i)


def test_masked_convolve_golden(ia):
    from imgprocessor_amd.filters import maskedConvolve, filter as ipa_filter
    g = load_golden('masked_convolve.npz')
    n = 0
    for key, want in g.items():
        if not key.startswith('out_'):
            continue
        _, im, kn, mk = key.split('_')
        img = g['img_' + im]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            got = maskedConvolve(img, g['kernel_' + kn], g['mask_' + mk])
        h = g['kernel_' + kn].shape[0] // 2
        assert buf.getvalue().strip() == str((img.shape[0] + 2 * h, img.shape[1] + 2 * h))
        assert got.dtype == want.dtype
        if img.dtype == np.float32:
            close32(got, want, key)
        else:
            assert_close(got, want, 1e-12, 1e-13, key)
        # the identity of SURVEY §8(b)
        f = ipa_filter(img, np.fft.fftshift(g['kernel_' + kn]))
        assert np.array_equal(np.where(g['mask_' + mk], f, 0), got)
        n += 1
    assert n == 13
    with contextlib.redirect_stdout(io.StringIO()):
        got = maskedConvolve(g['selftest_arr'], np.eye(5), g['selftest_mask'])
    assert_close(got, g['selftest_out'], 1e-12, 1e-13, 'reference selftest (maskedConvolve.py:56-73)')
    with pytest.raises(Exception):
        maskedConvolve(g['selftest_arr'], np.eye(5), g['selftest_mask'], mode='wrap')


def test_extend_array_golden(ia):
    from imgprocessor_amd.filters import extendArrayForConvolution
    g = load_golden('extend_array.npz')
    arr = g['arr']
    for key, want in g.items():
        if not key.startswith('ext_'):
            continue
        _, kx, ky, modex = key.split('_')
        got = extendArrayForConvolution(arr, (int(kx[2:]), int(ky[2:])), modex=modex)
        assert np.array_equal(got, want), key
    with pytest.raises(Exception):
        extendArrayForConvolution(arr, (3, 3), modey='wrap')


def test_gaussian_golden(ia, oracle):
    from imgprocessor_amd.filters import gaussian_filter, box_filter
    g = load_golden('remap_scipy.npz')
    img = g['img']
    for s in (0.5, 1.0, 1.25, 2.0):
        close32(gaussian_filter(img, s), g['gauss_s%s' % str(s).replace('.', 'p')], 'gauss %s' % s)
    close32(gaussian_filter(img, (1.0, 2.5)), g['gauss_s1_2p5'], 'gauss (1,2.5)')
    assert_close(gaussian_filter(img.astype(np.float64), 1.0), g['gauss64_s1'], 1e-12, 1e-13)
    for mode in ('constant', 'nearest', 'mirror', 'wrap'):
        k = oracle.gaussian_kernel1d(1.5)
        close32(ia.ops.sepconv2d(img, k, k, mode, 0.3), oracle.sepconv2d(img, k, k, mode, 0.3),
                'sep ' + mode)
    close32(ia.ops.sepconv2d(img, None, oracle.gaussian_kernel1d(2.0)),
            oracle.sepconv2d(img, None, oracle.gaussian_kernel1d(2.0)), 'x only')
    close32(box_filter(img, 3), oracle.conv2d(img, np.ones((3, 3)) / 9, 'mirror'), 'cv2.blur 3x3')
    big = synth((200, 300), 2)
    close32(gaussian_filter(big, 3.0), oracle.gaussian_filter(big, 3.0), 'sigma 3 (25 taps)')


def test_masked_filter_and_nan_max(ia, oracle):
    from imgprocessor_amd.filters import maskedFilter, nan_maximum_filter
    g = load_golden('masked_filter.npz')
    for ks in (6, 11, 30):
        a = g['arr'].copy()
        assert maskedFilter(a, g['mask'], ks, fn='mean') is a  # in place like the reference
        assert_close(a, g['mean_fill_k%d' % ks], 1e-13, 1e-15)
        got, want = maskedFilter(g['arr'].copy(), g['mask'], ks, fill_mask=False, fn='mean'), \
            g['mean_nofill_k%d' % ks]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert_close(np.nan_to_num(got), np.nan_to_num(want), 1e-13, 1e-15)
    a32 = g['arr'].astype(np.float32)
    close32(maskedFilter(a32, g['mask'], 6, fn='mean'), g['mean32_fill_k6'], 'mean f32')
    for ks in (3, 6, 9):
        assert np.array_equal(nan_maximum_filter(g['arr_nan'], ks), g['nanmax_k%d' % ks],
                              equal_nan=True)
    # device arrays + a bigger frame against the oracle (ragged size, all-masked columns)
    big = synth((301, 517), 21, np.float64)
    m = np.random.default_rng(3).random(big.shape) < 0.3
    m[:, 200:240] = True
    d = ia.default_context().to_device(big)
    dm = ia.default_context().to_device(m.astype(np.uint8))
    assert maskedFilter(d, dm, 30, fn='mean') is d
    assert_close(d.get(), oracle.maskedFilter(big.copy(), m, 30, fn='mean'), 1e-13, 1e-15)
    bn = big.copy()
    bn[m] = np.nan
    assert np.array_equal(nan_maximum_filter(bn.astype(np.float32), 7),
                          oracle.nan_maximum_filter(bn.astype(np.float32), 7), equal_nan=True)
    # median: pure selection -> bit-identical to the reference fixtures and to the oracle
    for ks in (6, 11):
        a = g['arr'].copy()
        assert maskedFilter(a, g['mask'], ks, fn='median') is a
        assert np.array_equal(a, g['median_fill_k%d' % ks])
        got = maskedFilter(g['arr'].copy(), g['mask'], ks, fill_mask=False, fn='median')
        assert np.array_equal(got, g['median_nofill_k%d' % ks], equal_nan=True)
    a32 = g['arr'].astype(np.float32)
    assert np.array_equal(maskedFilter(a32, g['mask'], 6, fn='median'), g['median32_fill_k6'])
    for dt in (np.float64, np.float32):
        b = big.astype(dt)
        want = oracle.maskedFilter(b.copy(), m, 30, True, 'median')
        assert np.array_equal(maskedFilter(b, m, 30, fn='median'), want)
    bn = big.copy()
    bn[::9, ::13] = np.nan   # a NaN among the window values makes the median NaN (np.median)
    want = oracle.maskedFilter(bn.copy(), m, 8, True, 'median')
    assert np.isnan(want).any()
    assert np.array_equal(maskedFilter(bn, m, 8, fn='median'), want, equal_nan=True)
    with pytest.raises(NotImplementedError):
        maskedFilter(big, m, 200, fn='median')   # window larger than the per-wave LDS buffer
    # the reference's default is the MEDIAN, and any fn other than 'mean' selects it too
    # (filters/maskedFilter.py:12-13, 30-35)
    a = g['arr'].copy()
    assert maskedFilter(a, g['mask'], 6) is a
    assert np.array_equal(a, g['median_fill_k6'])
    assert np.array_equal(maskedFilter(g['arr'].copy(), g['mask'], 6, fn='mode'),
                          g['median_fill_k6'])


def test_median_threshold_golden(ia, oracle):
    from imgprocessor_amd.filters import medianThreshold
    g = load_golden('median_threshold.npz')
    for key, thr, cond in (('thr0p1_gt', 0.1, '>'), ('thr0p5_gt', 0.5, '>'),
                           ('thr0p05_lt', 0.05, '<')):
        out, ind = medianThreshold(g['img'], thr, condition=cond)
        assert np.array_equal(out, g['out_' + key]), key
        assert ind.dtype == bool and np.array_equal(ind, g['ind_' + key]), key
    out, ind = medianThreshold(g['img'].astype(np.float32), 0.1)
    assert out.dtype == np.float32 and np.array_equal(out, g['out32_thr0p1_gt'])
    assert np.array_equal(ind, g['ind32_thr0p1_gt'])
    out, ind = medianThreshold(g['img_zero'], 0.1)
    assert np.array_equal(out, g['out_zero']) and np.array_equal(ind, g['ind_zero'])
    img = g['img'].copy()
    assert medianThreshold(img, 0.1, copy=False)[0] is img
    assert np.array_equal(img, g['out_thr0p1_gt'])
    assert medianThreshold(img, 0.0) == (img, None)
    # any window size (round 4): odd, even (scipy's shifted origin), larger - the reference's output
    for size in (5, 4, 2, 7, 9):
        out, ind = medianThreshold(g['img'], 0.1, size=size)
        assert np.array_equal(out, g['out_s%d' % size]), size
        assert np.array_equal(ind, g['ind_s%d' % size]), size
    out, ind = medianThreshold(g['img'].astype(np.float32), 0.1, size=5)
    assert out.dtype == np.float32 and np.array_equal(out, g['out32_s5'])
    assert np.array_equal(ind, g['ind32_s5'])
    with pytest.raises(NotImplementedError):
        medianThreshold(img, 0.1, size=(3, 5))
    # ragged sizes (tile edges in both directions) against the oracle, device arrays
    big = 0.2 + synth((203, 391), 5, np.float64)
    big[::7, ::11] *= 4
    d = ia.default_context().to_device(big)
    dout, dind = medianThreshold(d, 0.2)
    want, wind = oracle.medianThreshold(big, 0.2)
    assert np.array_equal(dout.get(), want) and np.array_equal(dind.get().astype(bool), wind)
    for size in (5, 6, 11):
        dout, dind = medianThreshold(d, 0.2, size=size)
        want, wind = oracle.medianThreshold(big, 0.2, size=size)
        assert np.array_equal(dout.get(), want) and np.array_equal(dind.get().astype(bool), wind), size
    one = synth((1, 5), 1, np.float32) + 0.5  # single row: every vertical neighbour is the row
    assert np.array_equal(medianThreshold(one, 0.01)[0], oracle.medianThreshold(one, 0.01)[0])


def test_closest_distance_and_position_uncertainty(ia, oracle):
    from imgprocessor_amd.render import closestDirectDistance
    from imgprocessor_amd.uncertainty import positionToIntensityUncertainty
    g = load_golden('render_uncertainty.npz')
    for ks in (4, 9):
        got = closestDirectDistance(g['cdd_arr'], ks)
        assert got.dtype == np.uint16 and np.array_equal(got, g['cdd_k%d' % ks])
    big = np.random.default_rng(8).random((301, 517)) > 0.997
    assert np.array_equal(closestDirectDistance(big, 30), oracle.closestDirectDistance(big, 30))
    assert np.array_equal(closestDirectDistance(big, 12, np.float64),
                          oracle.closestDirectDistance(big, 12, np.float64))

    def same(got, want, rt=1e-12):
        assert got.dtype == np.float64
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert_close(np.nan_to_num(got), np.nan_to_num(want), rt, rt)
    img = g['piu_img']
    same(positionToIntensityUncertainty(img, 1.5, 0.7, 7), g['piu_const_1p5_0p7_k7'])
    same(positionToIntensityUncertainty(img, 2, 2, 5), g['piu_const_2_2_k5'])
    same(positionToIntensityUncertainty(img, g['piu_sx'], g['piu_sy'], 7), g['piu_vari_k7'])
    same(positionToIntensityUncertainty(g['piu_u16'], 1, 1, 5), g['piu_u16_const_1_1_k5'])
    i32 = np.nan_to_num(img).astype(np.float32)   # float32 frames: differences in float64
    same(positionToIntensityUncertainty(i32, 1.5, 0.7, 7),
         oracle.positionToIntensityUncertainty(i32, 1.5, 0.7, 7))
    same(positionToIntensityUncertainty(img, 1.0, 1.0), oracle.positionToIntensityUncertainty(
        img, 1.0, 1.0, 5))  # kernelSize None -> max(3, 4*std+1)
    with pytest.raises(AssertionError):
        positionToIntensityUncertainty(img, g['piu_sx'], g['piu_sy'][:5], 7)


def test_camera_calibration_correct(ia, oracle, capsys):
    """CameraCalibration.correct: stages 2-4 pinned by the golden chain, stage 5 = the lens remap"""
    from imgprocessor_amd.camera.CameraCalibration import CameraCalibration
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    g = load_golden('median_threshold.npz')
    raw, bg, ff = g['cal_raw'], g['cal_bg'], g['cal_ff']
    for thr, key in ((0.1, 'cal_out_thr0p1'), (0.0, 'cal_out_thr0p0')):
        got = ia.ops.calib_prefilter(raw, bg, ff, thr)
        assert np.array_equal(got, g[key], equal_nan=True), key
    assert np.array_equal(ia.ops.calib_prefilter(raw, None, None, 0.0), raw, equal_nan=True)
    r32 = raw.astype(np.float32)
    assert np.array_equal(ia.ops.calib_prefilter(r32, bg, ff, 0.1),
                          oracle.calib_prefilter(r32, bg, ff, 0.1))

    cal = CameraCalibration()
    cal.addDarkCurrent(bg, date='02 Nov 15 - 10:00')
    cal.addFlatField(ff, date='03 Nov 15 - 10:00')
    out = cal.correct(raw, exposure_time=1.0, threshold=0.1)
    assert 'CORRECT CAMERA' in capsys.readouterr().out
    assert out.dtype == np.float64 and np.array_equal(out, g['cal_out_thr0p1'])
    # explicit background image wins over the stored dark current (:481-498)
    out = cal.correct(raw, bgImages=[bg + 1.0], threshold=0.1)
    assert np.array_equal(out, oracle.calib_prefilter(raw, bg + 1.0, ff, 0.1))
    # + lens stage: equals the oracle chain prefilter -> undistort (keep_size True and False)
    h, w = raw.shape
    from imgprocessor_amd.utils import getOptimalNewCameraMatrix
    ld = LensDistortion()
    ld.setCameraParams(60., 60., (w - 1) / 2., (h - 1) / 2., -0.12, 0.03, 0.0, 1e-3, -5e-4)
    ld.coeffs['shape'] = (h, w)
    cal.addLens(ld, date='04 Nov 15 - 10:00')
    out_lens = cal.correct(raw, exposure_time=1.0, threshold=0.1, keep_size=True)
    K, d = ld.coeffs['cameraMatrix'], ld.coeffs['distortionCoeffs'].ravel()
    nK, roi = getOptimalNewCameraMatrix(K, d, (w, h), 1, (w, h))
    want = oracle.undistort(g['cal_out_thr0p1'], K, d, nK)
    assert out_lens.shape == (h, w)
    assert_close(out_lens, want, 1e-12, 1e-9)
    crop = cal.correct(raw, exposure_time=1.0, threshold=0.1, keep_size=False)
    x, y, rw, rh = roi
    assert_close(crop, want[y:y + rh, x:x + rw], 1e-12, 1e-9)
    # the reference's dark-current model is broken for (slope, intercept): stage reported+skipped
    cal2 = CameraCalibration()
    cal2.addDarkCurrent(bg, np.zeros_like(bg), date='02 Nov 15 - 10:00')
    out = cal2.correct(raw, exposure_time=1.0, threshold=0.0)
    assert 'Error:' in capsys.readouterr().out
    assert np.array_equal(out, raw, equal_nan=True)
    # float32 frames (extension) and device-resident frames
    out32 = cal.correct(r32, exposure_time=1.0, threshold=0.1, dtype=np.float32)
    assert out32.dtype == np.float32 and np.abs(out32 - want).max() < 2e-3 * np.abs(want).max()
    dev = cal.correct(ia.default_context().to_device(raw), exposure_time=1.0, threshold=0.1)
    assert np.array_equal(dev.get(), out_lens)


# ---------------------------------------------------------------- fused ----
def test_fused_chain(ia, oracle):
    ctx = ia.default_context(0)
    g = load_golden('remap_scipy.npz')
    img = g['img']
    d_img = ctx.to_device(img)
    mx, my = g['mapx_radial'], g['mapy_radial']
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    got = ia.ops.remap_conv2d(d_img, dmx, dmy, g['k5']).get()
    close32(got, g['chain_radial_k5'], 'fused vs scipy chain')
    rng = np.random.default_rng(2)
    K, d, nK = g['K_strong'], g['dist_strong'], g['newK_strong']
    M = load_golden('warp_skimage.npz')['M_rot7']
    for ksz in (3, 5, 7, 9, 11):
        k = rng.random((ksz, ksz))
        k /= k.sum()
        for interp, iid in (('linear', oracle.LINEAR), ('cubic', oracle.CUBIC_KEYS)):
            for cmode in ('reflect', 'wrap', 'constant'):
                want = oracle.conv2d(oracle.remap(img, mx, my, iid, oracle.CONSTANT, 0.1), k, cmode)
                got = ia.ops.remap_conv2d(d_img, dmx, dmy, k, interp, 'constant', 0.1, cmode).get()
                close32(got, want, 'fused map k%d %s %s' % (ksz, interp, cmode))
        want = oracle.conv2d(oracle.undistort(img, K, d, nK), k)
        close32(ia.ops.undistort_conv2d(d_img, K, d, nK, k).get(), want, 'fused undistort k%d' % ksz)
        want = oracle.conv2d(oracle.warp_perspective(img, M, (80, 100), oracle.CUBIC_KEYS), k)
        close32(ia.ops.warp_perspective_conv2d(d_img, M, (80, 100), k, 'cubic').get(), want,
                'fused warp k%d' % ksz)
    # two launches == one launch
    two = ia.ops.conv2d(ia.ops.remap(d_img, dmx, dmy), g['k5']).get()
    close32(ia.ops.remap_conv2d(d_img, dmx, dmy, g['k5']).get(), two, 'fused vs two launches')
    # uint16 frames -> float32 (C4)
    d16 = ctx.to_device(g['img16'])
    k7 = g['k7']
    want = oracle.conv2d(oracle.remap(g['img16'], mx, my, out_dtype=np.float32), k7)
    close32(ia.ops.remap_conv2d(d16, dmx, dmy, k7).get(), want, 'fused u16')


def test_fused_separable_chain(ia, oracle):
    """remap -> separable filter in one kernel == oracle remap then sepconv2d (and == two launches)"""
    ctx = ia.default_context(0)
    g = load_golden('remap_scipy.npz')
    img = g['img']
    d_img = ctx.to_device(img)
    mx, my = g['mapx_radial'], g['mapy_radial']
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    K, d, nK = g['K_strong'], g['dist_strong'], g['newK_strong']
    M = load_golden('warp_skimage.npz')['M_rot7']
    rng = np.random.default_rng(5)
    for n in (3, 5, 7, 9, 11, 13):   # 11, 13: two launches inside
        ky, kx = rng.random(n), rng.random(n)
        ky /= ky.sum()
        kx /= kx.sum()
        for interp, iid in (('linear', oracle.LINEAR), ('cubic', oracle.CUBIC_KEYS)):
            for cmode in ('reflect', 'wrap', 'constant', 'mirror'):
                want = oracle.sepconv2d(oracle.remap(img, mx, my, iid, oracle.CONSTANT, 0.1),
                                        ky, kx, cmode)
                got = ia.ops.remap_sepconv2d(d_img, dmx, dmy, ky, kx, interp, 'constant', 0.1,
                                             cmode).get()
                close32(got, want, 'fused sep map n%d %s %s' % (n, interp, cmode))
        want = oracle.sepconv2d(oracle.undistort(img, K, d, nK), ky, kx)
        close32(ia.ops.undistort_sepconv2d(d_img, K, d, nK, ky, kx).get(), want,
                'fused sep undistort n%d' % n)
        want = oracle.sepconv2d(oracle.warp_perspective(img, M, (80, 100), oracle.CUBIC_KEYS), ky, kx)
        close32(ia.ops.warp_perspective_sepconv2d(d_img, M, (80, 100), ky, kx, 'cubic').get(), want,
                'fused sep warp n%d' % n)
    g9 = oracle.gaussian_kernel1d(1.0)
    two = ia.ops.sepconv2d(ia.ops.remap(d_img, dmx, dmy), g9, g9).get()
    assert np.array_equal(ia.ops.remap_sepconv2d(d_img, dmx, dmy, g9, g9).get(), two)
    # cv2-style quantised coordinates, a batch, and a frame wide enough for interior strips
    big = synth((200, 900), 3)
    yy, xx = np.mgrid[0:200, 0:900].astype(np.float32)
    bmx = (xx + 3.3 * np.sin(yy / 40)).astype(np.float32)
    bmy = (yy + 2.1 * np.cos(xx / 90)).astype(np.float32)
    batch = np.stack([big, big[::-1].copy(), synth((200, 900), 4)])
    for interp, iid in (('linear', oracle.LINEAR), ('linear_cv_q5', oracle.LINEAR | oracle.Q5)):
        got = ia.ops.remap_sepconv2d(ctx.to_device(batch), ctx.to_device(bmx), ctx.to_device(bmy),
                                     g9, g9, interp).get()
        for i in range(3):
            want = oracle.sepconv2d(oracle.remap(batch[i], bmx, bmy, iid), g9, g9)
            close32(got[i], want, 'fused sep batch %d %s' % (i, interp))
    with pytest.raises(ValueError):
        ia.ops.remap_sepconv2d(d_img, dmx, dmy, np.ones(4) / 4, g9)


# ------------------------------------------------------------------ IDW ----
def test_idw_golden(ia):
    from imgprocessor_amd.interpolate import (interpolate2dStructuredIDW,
                                               interpolate2dStructuredFastIDW)
    from imgprocessor_amd.interpolate.interpolate2dStructuredFastIDW import growPositions
    g = load_golden('idw.npz')
    grid = g['grid']
    n = 0
    for key, want in g.items():
        if key.startswith('idw_k'):
            p = key.split('_')
            kern, power = int(p[1][1:]), int(p[2][1:])
            fx, fy = (2, 0.5) if len(p) > 3 else (1, 1)
            gg = grid.copy()
            got = interpolate2dStructuredIDW(gg, g['mask_k%d' % kern], kern, power, fx, fy)
            assert got is gg  # in place, returns the grid
        elif key.startswith('fidw_k'):
            p = key.split('_')
            kern, power, minn = int(p[1][1:]), int(p[2][1:]), int(p[3][1:])
            got = interpolate2dStructuredFastIDW(grid.copy(), g['mask_k%d' % kern], kern, power, minn)
        else:
            continue
        assert_close(got, want, 1e-12, 1e-14, key)
        n += 1
    assert n == 21
    close32(interpolate2dStructuredIDW(grid.astype(np.float32), g['mask_k5'], 5, 2),
            g['idw32_k5_p2'], 'idw32')
    assert_close(interpolate2dStructuredIDW(grid.copy(), g['mask_block'], 3, 2), g['idw_block_k3'],
                 1e-12)
    assert_close(interpolate2dStructuredFastIDW(grid.copy(), g['mask_block'], 3, 2, 5),
                 g['fidw_block_k3'], 1e-12)
    pos, dist = growPositions(4)
    assert np.array_equal(pos, g['grow4_pos']) and np.array_equal(dist, g['grow4_dist'])


def test_idw_edges_vs_oracle(ia, oracle):
    from imgprocessor_amd.interpolate import (interpolate2dStructuredIDW,
                                               interpolate2dStructuredFastIDW)
    rng = np.random.default_rng(8)
    for (H, W) in ((40, 150), (65, 64), (7, 9)):
        grid = rng.random((H, W))
        mask = rng.random((H, W)) < 0.3  # masked pixels right up to every edge
        k = 4
        assert_close(interpolate2dStructuredIDW(grid.copy(), mask, k, 2),
                     oracle.interpolate2dStructuredIDW(grid.copy(), mask, k, 2), 1e-12, 1e-14)
        assert_close(interpolate2dStructuredFastIDW(grid.copy(), mask, k, 2, 5),
                     oracle.interpolate2dStructuredFastIDW(grid.copy(), mask, k, 2, 5), 1e-12, 1e-14)


# ------------------------------------------------------- drop-in surface ----
def test_lens_distortion_class(ia, oracle):
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    H, W = 120, 160
    img = synth((H, W), 3)
    fx = fy = 160.0
    ld = LensDistortion()
    ld.setCameraParams(fx, fy, 79.5, 59.5, -0.12, 0.03, 0.0, 1e-3, -5e-4)
    assert ld.getCameraParams() == (fx, fy, 79.5, 59.5, -0.12, 0.03, 0.0, 1e-3, -5e-4)
    out = ld.correct(img)                      # keepSize=False: cropped to roi
    xx, yy, ww, hh = ld.roi
    assert out.shape == (hh, ww) and out is ld.img
    full = ld.correct(img, keepSize=True)
    assert full.shape == img.shape
    assert np.array_equal(out, full[yy:yy + hh, xx:xx + ww])
    mx, my = ld.getUndistortRectifyMap(W, H)
    assert mx.dtype == np.float32 and mx.shape == (H, W)
    assert ld.getUndistortRectifyMap(W, H)[0] is mx  # cached per shape
    K = np.array([[fx, 0, 79.5], [0, fy, 59.5], [0, 0, 1]])
    d = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    omx, omy = oracle.build_undistort_map(K, d, ld.newCameraMatrix, H, W)
    assert np.array_equal(mx, omx) and np.array_equal(my, omy)
    close32(full, oracle.remap(img, omx, omy), 'correct', scale=1.0)
    close32(ld.correct(img, keepSize=True, borderValue=0.5),
            oracle.remap(img, omx, omy, cval=0.5), 'borderValue', scale=1.0)
    # colour image = independent channels; uint8 stays uint8
    rgb = (np.stack([synth((H, W), s) for s in (1, 2, 3)], axis=2) * 255).astype(np.uint8)
    o = ld.correct(rgb, keepSize=True)
    assert o.shape == rgb.shape and o.dtype == np.uint8
    for c in range(3):
        assert np.array_equal(o[..., c], oracle.remap(np.ascontiguousarray(rgb[..., c]), omx, omy))
    # the camera's uint16 frames: cv2.remap's 16U arithmetic (float32 table weights at 1/32 px),
    # as the reference's correct() gets from cv2 (camera/LensDistortion.py:323-326)
    f16 = np.round(synth((H, W), 8, np.float64) * 4095).astype(np.uint16)
    o16 = ld.correct(f16, keepSize=True)
    assert o16.dtype == np.uint16
    assert np.array_equal(o16, oracle.remap(f16, omx, omy, oracle.LINEAR | oracle.Q5))
    # distortImage: first-order inverse map as written in the reference
    posy, posx = np.mgrid[0:H, 0:W].astype(np.float32)
    close32(ld.distortImage(img), oracle.remap(img, posx + (posx - omx), posy + (posy - omy)),
            'distortImage', scale=1.0)
    # undistort then re-distort is close to identity in the interior (approximate inverse)
    rt = ld.distortImage(ld.correct(img, keepSize=True))
    assert np.abs(rt - img)[30:-30, 30:-30].mean() < 0.04
    # zero distortion, newK = K -> identity
    ld0 = LensDistortion(newCameraMatrix='same')
    ld0.setCameraParams(fx, fy, 79.5, 59.5, 0, 0, 0, 0, 0)
    assert np.array_equal(ld0.correct(img, keepSize=True), img)
    # device arrays stay on the device
    ctx = ia.default_context(0)
    d_out = ld.correct(ctx.to_device(img), keepSize=True)
    assert isinstance(d_out, ia.DeviceArray) and np.array_equal(d_out.get(), full)


def test_perspective_correction_class(ia, oracle, capsys):
    from imgprocessor_amd.camera.PerspectiveCorrection import PerspectiveCorrection
    g = load_golden('warp_skimage.npz')
    img = g['img']
    quad = np.array([(8, 2), (120, 6), (122, 90), (5, 93)], float)
    pc = PerspectiveCorrection(img.shape, new_size=(96, 128))
    pc.setReference(quad[[2, 0, 3, 1]])  # any order: corners are sorted
    assert_close(pc.homography, g['H_quad'], 1e-9, 1e-9, 'homography')
    out = pc.correct(img)
    assert 'CORRECT PERSPECTIVE' in capsys.readouterr().out
    assert out.shape == (96, 128)
    Minv = np.linalg.inv(pc.homography)
    close32(out, oracle.warp_perspective(img, Minv, (96, 128), oracle.LANCZOS4), 'lanczos4',
            scale=1.0)
    pcl = PerspectiveCorrection(img.shape, new_size=(96, 128), interpolation='linear')
    pcl.setReference(quad)
    close32(pcl.correct(img), g['warp_quad_o1_c0'], 'linear vs skimage', scale=1.0)
    pcb = PerspectiveCorrection(img.shape, new_size=(80, 100), border=7, interpolation='cubic')
    pcb.setReference(quad)
    close32(pcb.correct(img), g['warp_quadb_o3_c0'], 'border=7 cubic vs skimage', scale=1.0)
    # uncorrect: INTER_CUBIC | WARP_INVERSE_MAP
    un = pc.uncorrect(out)
    assert un.shape == out.shape
    close32(un, oracle.warp_perspective(out, pc.homography, out.shape,
                                        oracle.CUBIC_CV | oracle.Q5), 'uncorrect', scale=1.0)
    # round trip correct -> uncorrect is close to the original inside the quad
    assert np.abs(un - img)[20:-20, 20:-20].mean() < 0.03
    pts = pc.correctPoints(quad)
    assert_close(pts[0], [[0, 0], [128, 0], [128, 96], [0, 96]], 0, 1e-3)
    pch = PerspectiveCorrection(img.shape, new_size=(96, 128), interpolation='linear')
    pch.setReference(g['H_rot7'])
    close32(pch.correct(img), g['warp_rot7_o1_c0'], 'explicit homography', scale=1.0)


def test_rotate_roundtrip_bound(ia):
    """the reference's own check, transform/rotate.py:26-32: a 50x50 ramp rotated by +14 deg
    then -14 deg (INTER_CUBIC, BORDER_REFLECT, about the image centre) stays within 0.005"""
    a = np.tile(np.linspace(0, 1, 50), (50, 1))

    def rotate(img, angle):
        s0, s1 = img.shape
        cx, cy = (s0 - 1) / 2., (s1 - 1) / 2.
        t = np.deg2rad(angle)
        al, be = np.cos(t), np.sin(t)  # cv2.getRotationMatrix2D
        M = np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy],
                      [0, 0, 1.]])
        return ia.ops.warp_perspective(img, np.linalg.inv(M), img.shape, 'cubic_cv_q5', 'reflect')
    for dt in (np.float64, np.float32):
        c = rotate(rotate(a.astype(dt), 14), -14)
        assert np.abs(a - c).mean() < 0.005


def test_errors(ia):
    img = np.zeros((8, 8), np.float32)
    with pytest.raises(ValueError):
        ia.ops.remap(img, np.zeros((4, 4), np.float32), np.zeros((4, 5), np.float32))
    with pytest.raises(ValueError):
        ia.ops.remap(img, np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32), 'bogus')
    with pytest.raises(TypeError):
        ia.ops.remap(img.astype(np.int32), np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))
    with pytest.raises(NotImplementedError):
        ia.ops.remap(img.astype(np.float64), np.zeros((4, 4), np.float32),
                     np.zeros((4, 4), np.float32), out_dtype=np.float32)
    with pytest.raises(ValueError):
        ia.ops.undistort(img, np.eye(3), np.zeros(5), np.zeros((3, 3)))  # singular newK


def test_cv_u16_arithmetic(ia, oracle):
    """uint16 -> uint16 in the cv2 modes (linear_cv_q5, cubic_cv_q5, lanczos4) = OpenCV's 16U
    arithmetic (float32 table weights, float32 accumulation, cvRound): bit for bit the oracle for
    every border mode and border value, and the independent numpy restatement (Lanczos4: up to
    the last-bit differences of the 1-D tables).  This is what LensDistortion.correct /
    PerspectiveCorrection.correct do on the camera's uint16 frames (camera/LensDistortion.py:
    323-326, camera/PerspectiveCorrection.py:401-405)."""
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    img16 = g['img16']
    big = np.round(synth((300, 700), 77, np.float64) * 65535).astype(np.uint16)
    yy, xx = np.mgrid[0:300, 0:700].astype(np.float32)
    bmx = (xx * 1.03 - 14.3 + 2.0 * np.sin(yy / 23)).astype(np.float32)
    bmy = (yy * 0.97 + 6.1 + 1.5 * np.cos(xx / 31)).astype(np.float32)
    modes = (('linear', 'linear_cv_q5', oracle.LINEAR | oracle.Q5),
             ('cubic', 'cubic_cv_q5', oracle.CUBIC_CV | oracle.Q5),
             ('lanczos4', 'lanczos4', oracle.LANCZOS4))
    for name in ('radial', 'strong'):
        mx, my = g['mapx_' + name], g['mapy_' + name]
        for kind, iname, iid in modes:
            for key, cv in (('u16cv', 0), ('u16cv1000', 1000)):
                got = ia.ops.remap(img16, mx, my, iname, 'constant', cv)
                assert got.dtype == np.uint16
                assert np.array_equal(got, oracle.remap(img16, mx, my, iid, oracle.CONSTANT, cv)), (kind, name, key)
                want = g['%s_%s_%s' % (key, kind, name)]
                if kind == 'lanczos4':
                    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
                    assert d.max() <= 1 and (d != 0).mean() < 0.01
                else:
                    assert np.array_equal(got, want), (kind, name, key)
            for bname, bid in (('replicate', oracle.REPLICATE), ('reflect', oracle.REFLECT),
                               ('reflect101', oracle.REFLECT101), ('wrap', oracle.WRAP)):
                assert np.array_equal(ia.ops.remap(img16, mx, my, iname, bname),
                                      oracle.remap(img16, mx, my, iid, bid)), (kind, name, bname)
    for kind, iname, iid in modes:   # a frame with several tiles per axis, batch of 3
        batch = np.stack([np.roll(big, 13 * i, axis=1) for i in range(3)])
        got = ia.ops.remap(batch, bmx, bmy, iname, 'constant', 70)
        for i in range(3):
            assert np.array_equal(got[i], oracle.remap(batch[i], bmx, bmy, iid, oracle.CONSTANT, 70)), (kind, i)
    # the analytic sources take the same arithmetic
    M = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])
    for kind, iname, iid in modes:
        assert np.array_equal(ia.ops.warp_perspective(big, M, (300, 700), iname),
                              oracle.warp_perspective(big, M, (300, 700), iid)), kind


def test_cv_modes_vs_independent_restatements(ia, oracle):
    """the GPU path against the second, independently written numpy restatements of the
    cv2-specific modes (cv_modes.npz; the oracle is checked against the same fixtures in
    test_oracle_golden.py).  These are the DEFAULTS of the reference's classes:
    PerspectiveCorrection.correct = Lanczos4, uncorrect / distort = bicubic a=-0.75 at 1/32 px,
    cv2.remap on uint8 = 15-bit fixed point."""
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    img, img8 = g['img'], g['img8']
    for name in ('radial', 'strong'):
        mx, my = g['mapx_' + name], g['mapy_' + name]
        for cname, cv in (('c0', 0.0), ('c037', 0.37)):
            close32(ia.ops.remap(img, mx, my, 'linear_cv_q5', 'constant', cv),
                    g['q5lin_%s_%s' % (name, cname)], 'q5 vs scipy ' + name, scale=1.0)
        for key, iname in (('cubic075', 'cubic_cv'), ('cubic075q5', 'cubic_cv_q5'),
                           ('cubic05', 'cubic'), ('lanczos4', 'lanczos4')):
            close32(ia.ops.remap(img, mx, my, iname), g['%s_%s' % (key, name)],
                    key + ' ' + name, scale=1.0)
        assert np.array_equal(ia.ops.remap(img8, mx, my), g['u8fix_' + name])
        assert np.array_equal(ia.ops.remap(img8, mx, my, border_value=17), g['u8fix17_' + name])
        # uint8 bicubic (a=-0.75) / Lanczos4 = OpenCV's short-weight tables: equal to the oracle
        # for every coordinate source and border, and to the independent numpy restatement
        # (Lanczos4: its float32 1-D weights differ from the oracle's in the last bit - two of
        # the 1024 integer tables differ by one in one entry)
        for kind, iname, frac in (('cubic', 'cubic_cv', 0.0), ('lanczos4', 'lanczos4', 0.01)):
            got = ia.ops.remap(img8, mx, my, iname)
            iid = oracle.CUBIC_CV if kind == 'cubic' else oracle.LANCZOS4
            assert np.array_equal(got, oracle.remap(img8, mx, my, iid))
            for bname, bid in (('replicate', oracle.REPLICATE), ('reflect101', oracle.REFLECT101),
                               ('wrap', oracle.WRAP)):
                assert np.array_equal(ia.ops.remap(img8, mx, my, iname, bname),
                                      oracle.remap(img8, mx, my, iid, bid)), (kind, bname)
            for key, cv in (('u8tab', 0), ('u8tab17', 17)):
                want = g['%s_%s_%s' % (key, kind, name)]
                d = np.abs(ia.ops.remap(img8, mx, my, iname, border_value=cv).astype(np.int32) - want)
                assert d.max() <= 1 and (d != 0).mean() <= frac, (kind, name, key)


def test_cv2_warp_perspective_vectors_when_present(ia):
    """the GPU path against cv2.warpPerspective's own output (PerspectiveCorrection.correct: Lanczos4,
    uncorrect: INTER_CUBIC | WARP_INVERSE_MAP) - only where tests/golden/gen_cv2_golden.py has run
    under an OpenCV; without it the test states "cv2-unpinned" and returns"""
    from .test_oracle_golden import cv2_warp_cases
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    if 'warp_H' not in g:
        assert pin.startswith('cv2-unpinned')
        return
    H = g['warp_H']
    names = {'linear': 'linear_cv_q5', 'cubic': 'cubic_cv_q5', 'lanczos4': 'lanczos4'}
    for key, img, kind, inverse in cv2_warp_cases(g):
        M = H if inverse else np.linalg.inv(H)
        got = ia.ops.warp_perspective(img, M, img.shape, names[kind])
        if img.dtype == np.float32:
            close32(got, g[key], key, scale=1.0)
        else:
            d = np.abs(got.astype(np.int64) - g[key].astype(np.int64))
            assert d.max() <= 1 and (d != 0).mean() <= 0.01, (key, d.max(), (d != 0).mean())


def test_u8_lanczos4_footprints_ending_on_the_last_pixel(ia, oracle):
    """uint8 Lanczos4 (OpenCV's 8U table arithmetic): a footprint whose last tap is the LAST pixel of
    the frame - its aligned tap dwords would reach past the frame's buffer descriptor - must still
    be bit-exact (found by tests/fuzz_oracle.py in the bottom-right corner of a 204 x 155 frame)"""
    rng = np.random.default_rng(21)
    for (h, w) in ((204, 155), (37, 53), (40, 64), (33, 67)):
        for n in (1, 3):
            src = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
            yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
            mx = (xx * ((w - 4.2) / (w - 1))).astype(np.float32)      # last column -> x = w - 4.2
            my = (yy * ((h - 4.22) / (h - 1))).astype(np.float32)     # last row -> footprint ends on h - 1
            got = ia.ops.remap(src, mx, my, 'lanczos4', 'constant', 0.3)
            for f in range(n):
                want = oracle.remap(src[f], mx, my, oracle.LANCZOS4, oracle.CONSTANT, 0.3)
                assert np.array_equal(got[f], want), (h, w, n, f)
