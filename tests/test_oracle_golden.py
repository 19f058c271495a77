"""CPU: the oracle (oracle/oracle.c) against the golden fixtures.

Fixtures come from the reference itself (in-tree numba stencils run through an
identity shim), scipy.ndimage and skimage — see tests/golden/gen_golden.py.
"""
import numpy as np
import pytest

from .conftest import load_golden, assert_close, load_cv_golden

F32 = 2.5e-7  # one float32 ulp-ish: oracle accumulates in double, output is float32


def test_extend_array(oracle):
    g = load_golden('extend_array.npz')
    arr = g['arr']
    n = 0
    for k, want in g.items():
        if not k.startswith('ext_'):
            continue
        _, kx, ky, modex = k.split('_')
        got = oracle.extendArrayForConvolution(arr, (int(kx[2:]), int(ky[2:])), modex=modex)
        assert np.array_equal(got, want), k
        n += 1
    assert n == 10
    # numpy equivalents
    assert np.array_equal(oracle.extendArrayForConvolution(arr, (9, 5)),
                          np.pad(arr, ((2, 2), (4, 4)), mode='symmetric'))
    with pytest.raises(RuntimeError):  # modey='wrap' is not supported by the reference (:57)
        oracle.extendArrayForConvolution(arr, (3, 3), modey='wrap')


def test_masked_convolve(oracle):
    g = load_golden('masked_convolve.npz')
    n = 0
    for k, want in g.items():
        if not k.startswith('out_'):
            continue
        _, im, kn, mk = k.split('_')
        img = g['img_' + im]
        got = oracle.maskedConvolve(img, g['kernel_' + kn], g['mask_' + mk])
        assert got.dtype == want.dtype
        assert_close(got, want, rtol=F32 if img.dtype == np.float32 else 1e-13,
                     atol=1e-7 if img.dtype == np.float32 else 1e-15, what=k)
        # identity of SURVEY §8(b): maskedConvolve == where(mask, correlate(arr, fftshift(k)), 0)
        c = oracle.conv2d(img, np.fft.fftshift(g['kernel_' + kn]), 'reflect', mask=g['mask_' + mk])
        assert np.array_equal(c, got), k
        n += 1
    assert n == 13
    got = oracle.maskedConvolve(g['selftest_arr'], np.eye(5), g['selftest_mask'])
    assert_close(got, g['selftest_out'], rtol=1e-13, atol=1e-15, what='selftest')


def test_var_y_gauss(oracle):
    g = load_golden('var_y_gauss.npz')
    assert_close(oracle.varYSizeGaussianFilter(g['arr'], (0, 4), 1), g['out_0_4_1'], 1e-12, 1e-15)
    assert_close(oracle.varYSizeGaussianFilter(g['arr'], 3, 0), g['out_3_0'], 1e-12, 1e-15)
    assert_close(oracle.varYSizeGaussianFilter(g['arr_nan'], (0, 4), 1), g['out_nan_0_4_1'],
                 1e-12, 1e-15)
    assert_close(oracle.varYSizeGaussianFilter(g['arr'], (1, 3), 2, modex='reflect'),
                 g['out_1_3_2_reflect'], 1e-12, 1e-15)


def test_std2d(oracle):
    g = load_golden('std2d.npz')
    for k in (5, 11):
        assert_close(oracle.standardDeviation2d(g['img'], k), g['std_k%d' % k], 1e-11, 1e-14)
    assert_close(oracle.standardDeviation2d(g['img32'], 5), g['std32_k5'], 2e-6, 1e-7)


def test_masked_filter_and_nan_max(oracle):
    g = load_golden('masked_filter.npz')
    for ks in (6, 11, 30):
        got = oracle.maskedFilter(g['arr'].copy(), g['mask'], ks, True, 'mean')
        assert_close(got, g['mean_fill_k%d' % ks], 1e-13, 1e-15)
        got = oracle.maskedFilter(g['arr'].copy(), g['mask'], ks, False, 'mean')
        want = g['mean_nofill_k%d' % ks]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert_close(np.nan_to_num(got), np.nan_to_num(want), 1e-13, 1e-15)
    got = oracle.maskedFilter(g['arr'].astype(np.float32), g['mask'], 6, True, 'mean')
    assert got.dtype == np.float32
    assert_close(got, g['mean32_fill_k6'], 1e-6, 1e-7)
    # default = median (the reference's default)
    assert np.array_equal(oracle.maskedFilter(g['arr'].copy(), g['mask'], 6), g['median_fill_k6'])
    for ks in (6, 11):   # median: selection only -> bit-exact
        got = oracle.maskedFilter(g['arr'].copy(), g['mask'], ks, True, 'median')
        assert np.array_equal(got, g['median_fill_k%d' % ks])
        got = oracle.maskedFilter(g['arr'].copy(), g['mask'], ks, False, 'median')
        assert np.array_equal(got, g['median_nofill_k%d' % ks], equal_nan=True)
    got = oracle.maskedFilter(g['arr'].astype(np.float32), g['mask'], 6, True, 'median')
    assert got.dtype == np.float32 and np.array_equal(got, g['median32_fill_k6'])
    for ks in (3, 6, 9):
        got, want = oracle.nan_maximum_filter(g['arr_nan'], ks), g['nanmax_k%d' % ks]
        assert np.isnan(want).any() or ks > 3
        assert np.array_equal(got, want, equal_nan=True)


def test_c1_512_box3(oracle):
    """BASELINE config C1 at full size: maskedConvolve(ones(3,3)/9, all-true mask) equals
    scipy's uniform_filter(3, 'reflect') (the SURVEY §8c probe against the reference itself)"""
    import scipy.ndimage as ndi
    from .conftest import synth
    img = synth((512, 512), 0)
    got = oracle.maskedConvolve(img, np.ones((3, 3)) / 9, np.ones(img.shape, bool))
    assert got.dtype == np.float32
    assert_close(got, ndi.uniform_filter(img.astype(np.float64), 3, mode='reflect'), 1e-6, 1e-7)


def test_median_threshold_and_calibration_stages(oracle):
    """bit-exact against the reference's medianThreshold and the numpy statements of
    CameraCalibration.correct stages 2-4 (selection + IEEE division only)"""
    g = load_golden('median_threshold.npz')
    for key, thr, cond in (('thr0p1_gt', 0.1, '>'), ('thr0p5_gt', 0.5, '>'),
                           ('thr0p05_lt', 0.05, '<')):
        out, ind = oracle.medianThreshold(g['img'], thr, condition=cond)
        assert np.array_equal(out, g['out_' + key]) and np.array_equal(ind, g['ind_' + key])
        assert 0 < ind.sum() < ind.size
    out, ind = oracle.medianThreshold(g['img'].astype(np.float32), 0.1)
    assert out.dtype == np.float32 and np.array_equal(out, g['out32_thr0p1_gt'])
    assert np.array_equal(ind, g['ind32_thr0p1_gt'])
    out, ind = oracle.medianThreshold(g['img_zero'], 0.1)  # median == 0: inf replaced, nan kept
    assert np.array_equal(out, g['out_zero']) and np.array_equal(ind, g['ind_zero'])
    img = g['img'].copy()
    assert oracle.medianThreshold(img, 0.1, copy=False)[0] is img
    assert np.array_equal(img, g['out_thr0p1_gt'])
    assert oracle.medianThreshold(img, 0.0) == (img, None)
    for size in (5, 4, 2, 7, 9):   # other window sizes, even ones with scipy's shifted origin
        out, ind = oracle.medianThreshold(g['img'], 0.1, size=size)
        assert np.array_equal(out, g['out_s%d' % size]) and np.array_equal(ind, g['ind_s%d' % size])
    out, ind = oracle.medianThreshold(g['img'].astype(np.float32), 0.1, size=5)
    assert np.array_equal(out, g['out32_s5']) and np.array_equal(ind, g['ind32_s5'])
    for thr, key in ((0.1, 'cal_out_thr0p1'), (0.0, 'cal_out_thr0p0')):
        got = oracle.calib_prefilter(g['cal_raw'], g['cal_bg'], g['cal_ff'], thr)
        assert np.array_equal(got, g[key], equal_nan=True), key
    assert np.isnan(g['cal_out_thr0p0']).any() and np.isfinite(g['cal_out_thr0p1']).all()


def test_closest_distance_and_position_uncertainty(oracle):
    g = load_golden('render_uncertainty.npz')
    for ks in (4, 9):
        got = oracle.closestDirectDistance(g['cdd_arr'], ks)
        assert got.dtype == np.uint16 and np.array_equal(got, g['cdd_k%d' % ks])
    f = oracle.closestDirectDistance(g['cdd_arr'], 9, np.float64)
    assert np.array_equal(f.astype(np.uint16), g['cdd_k9']) and (f[g['cdd_arr']] == 0).all()

    def same(got, want):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert_close(np.nan_to_num(got), np.nan_to_num(want), 1e-12, 1e-12)
    same(oracle.positionToIntensityUncertainty(g['piu_img'], 1.5, 0.7, 7), g['piu_const_1p5_0p7_k7'])
    same(oracle.positionToIntensityUncertainty(g['piu_img'], 2, 2, 5), g['piu_const_2_2_k5'])
    same(oracle.positionToIntensityUncertainty(g['piu_img'], g['piu_sx'], g['piu_sy'], 7),
         g['piu_vari_k7'])
    same(oracle.positionToIntensityUncertainty(g['piu_u16'], 1, 1, 5), g['piu_u16_const_1_1_k5'])
    assert np.isnan(g['piu_vari_k7']).any() and (g['piu_vari_k7'][:3] == 0).all()


def test_idw(oracle):
    g = load_golden('idw.npz')
    grid = g['grid']
    n = 0
    for k, want in g.items():
        if k.startswith('idw_k'):
            p = k.split('_')
            kern, power = int(p[1][1:]), int(p[2][1:])
            fx, fy = (2, 0.5) if len(p) > 3 else (1, 1)
            got = oracle.interpolate2dStructuredIDW(grid.copy(), g['mask_k%d' % kern], kern,
                                                    power, fx, fy)
        elif k.startswith('fidw_k'):
            p = k.split('_')
            kern, power, minn = int(p[1][1:]), int(p[2][1:]), int(p[3][1:])
            got = oracle.interpolate2dStructuredFastIDW(grid.copy(), g['mask_k%d' % kern], kern,
                                                        power, minn)
        else:
            continue
        assert_close(got, want, 1e-12, 1e-15, what=k)
        n += 1
    assert n == 21
    assert_close(oracle.interpolate2dStructuredIDW(grid.astype(np.float32), g['mask_k5'], 5, 2),
                 g['idw32_k5_p2'], F32, 0, 'idw32')
    assert_close(oracle.interpolate2dStructuredIDW(grid.copy(), g['mask_block'], 3, 2),
                 g['idw_block_k3'], 1e-12)
    assert_close(oracle.interpolate2dStructuredFastIDW(grid.copy(), g['mask_block'], 3, 2, 5),
                 g['fidw_block_k3'], 1e-12)
    pos, dist = oracle.growPositions(4)
    assert np.array_equal(pos, g['grow4_pos']) and np.array_equal(dist, g['grow4_dist'])


CASES = ('zero', 'radial', 'synthdefault', 'strong')


def test_undistort_map(oracle):
    g = load_golden('remap_scipy.npz')
    H, W = g['img'].shape
    for c in CASES:
        mx, my = oracle.build_undistort_map(g['K_' + c], g['dist_' + c], g['newK_' + c], H, W)
        # independent numpy float64 evaluation of the documented formula; allow 1 ulp of f32
        assert_close(mx, g['mapx_' + c], 1.2e-7, 1e-6, 'mapx ' + c)
        assert_close(my, g['mapy_' + c], 1.2e-7, 1e-6, 'mapy ' + c)
    mx, my = oracle.build_undistort_map(g['K_zero'], g['dist_zero'], g['newK_zero'], H, W)
    yy, xx = np.mgrid[0:H, 0:W]
    assert np.abs(mx - xx).max() < 1e-4 and np.abs(my - yy).max() < 1e-4


def test_remap_linear_scipy(oracle):
    g = load_golden('remap_scipy.npz')
    img, img16 = g['img'], g['img16']
    n = 0
    for c in CASES:
        mx, my = g['mapx_' + c], g['mapy_' + c]
        for cname, cv in (('c0', 0.0), ('cnan', np.nan), ('c037', 0.37)):
            key = 'lin_%s_%s' % (c, cname)
            if key not in g:
                continue
            got = oracle.remap(img, mx, my, oracle.LINEAR, oracle.CONSTANT, cv)
            assert_close(got, g[key], F32, 1e-7, key)
            n += 1
        got = oracle.remap(img16, mx, my, oracle.LINEAR, oracle.CONSTANT, 0, out_dtype=np.float32)
        assert_close(got, g['lin16_' + c], F32, 1e-4, 'lin16 ' + c)
        # analytic undistort == map-based remap (same float32 coordinates)
        an = oracle.undistort(img, g['K_' + c], g['dist_' + c], g['newK_' + c])
        assert_close(an, g['lin_%s_c0' % c], 2e-6, 2e-6, 'analytic ' + c)
    assert n == 8
    mx, my = g['mapx_strong'], g['mapy_strong']
    for smode, b in (('nearest', oracle.REPLICATE), ('reflect', oracle.REFLECT),
                     ('mirror', oracle.REFLECT101), ('grid-wrap', oracle.WRAP)):
        got = oracle.remap(img, mx, my, oracle.LINEAR, b)
        assert_close(got, g['lin_strong_' + smode], F32, 1e-7, smode)


def test_filters_scipy(oracle):
    g = load_golden('remap_scipy.npz')
    img = g['img']
    assert_close(oracle.conv2d(img, np.ones((3, 3)) / 9), g['corr_box3'], F32, 0, 'box3')
    assert_close(oracle.conv2d(img, g['k7']), g['corr_k7'], F32, 0, 'k7')
    assert_close(oracle.conv2d(img, g['k11']), g['corr_k11'], F32, 0, 'k11')
    assert_close(oracle.conv2d(img, g['k7'][:3, :]), g['corr_k3x7'], F32, 0, 'k3x7')
    assert_close(oracle.conv2d(img, g['k7'][:6, :4]), g['corr_k6x4'], F32, 0, 'k6x4')
    for smode in ('nearest', 'mirror', 'wrap', 'constant'):
        assert_close(oracle.conv2d(img, g['k7'], smode, cval=0.25), g['corr_k7_' + smode], F32, 0,
                     smode)
    for s in (0.5, 1.0, 1.25, 2.0):
        assert_close(oracle.gaussian_filter(img, s), g['gauss_s%s' % str(s).replace('.', 'p')],
                     F32, 0, 'gauss %s' % s)
    assert_close(oracle.gaussian_filter(img, (1.0, 2.5)), g['gauss_s1_2p5'], F32, 0)
    assert_close(oracle.gaussian_filter(img.astype(np.float64), 1.0), g['gauss64_s1'], 1e-13, 0)
    # kernel radius rule of scipy: 5 taps <-> sigma 0.5, 9 taps <-> sigma 1.0
    assert oracle.gaussian_kernel1d(0.5).size == 5 and oracle.gaussian_kernel1d(1.0).size == 9
    # headline chain
    und = oracle.remap(img, g['mapx_radial'], g['mapy_radial'])
    assert_close(oracle.conv2d(und, g['k5']), g['chain_radial_k5'], 4e-7, 0, 'chain')
    assert_close(oracle.remap_conv2d(img, g['mapx_radial'], g['mapy_radial'], g['k5']),
                 g['chain_radial_k5'], 4e-7, 0, 'chain fused entry')


def test_warp_skimage(oracle):
    g = load_golden('warp_skimage.npz')
    img = g['img'].astype(np.float64)
    n = 0
    for name in ('quad', 'quadb', 'rot7'):
        M = g['M_' + name]
        shp = tuple(int(v) for v in g['shape_' + name])
        for order, interp in ((1, oracle.LINEAR), (3, oracle.CUBIC_KEYS)):
            for cname, cv in (('c0', 0.0), ('c05', 0.5)):
                want = g['warp_%s_o%d_%s' % (name, order, cname)]
                got = oracle.warp_perspective(img, M, shp, interp, oracle.CONSTANT, cv)
                assert_close(got, want, 1e-11, 1e-12, '%s o%d %s' % (name, order, cname))
                n += 1
        got = oracle.warp_perspective(img, M, shp, oracle.LINEAR, oracle.REPLICATE)
        assert_close(got, g['warp_%s_o1_edge' % name], 1e-11, 1e-12, name + ' edge')
        # H maps the quad onto the destination rectangle (getPerspectiveTransform semantics)
    assert n == 12
    quad = np.array([(8, 2), (120, 6), (122, 90), (5, 93)], float)
    dst = np.array([[0, 0], [128, 0], [128, 96], [0, 96]], float)
    H = oracle.get_perspective_transform(quad, dst)
    assert_close(H, g['H_quad'], 1e-10, 1e-12)
    p = H @ np.c_[quad, np.ones(4)].T
    assert_close((p[:2] / p[2]).T, dst, 0, 1e-9)


def test_known_answers(oracle):
    rng = np.random.default_rng(5)
    img = rng.random((33, 47)).astype(np.float32)
    yy, xx = np.mgrid[0:33, 0:47].astype(np.float32)
    for interp in (oracle.NEAREST, oracle.LINEAR, oracle.CUBIC_CV, oracle.CUBIC_KEYS,
                   oracle.LANCZOS4, oracle.LINEAR | oracle.Q5, oracle.CUBIC_CV | oracle.Q5):
        got = oracle.remap(img, xx, yy, interp)
        assert np.array_equal(got, img), interp  # identity map -> bit-exact copy
    # integer translation with constant border
    got = oracle.remap(img, xx + 3, yy - 2, oracle.LINEAR, oracle.CONSTANT, 7.0)
    want = np.full_like(img, 7.0)
    want[2:, :-3] = img[:-2, 3:]
    assert np.array_equal(got, want)
    # uint8 fixed-point path: identity and half-pixel average with rounding
    u8 = rng.integers(0, 256, (33, 47), dtype=np.uint8)
    assert np.array_equal(oracle.remap(u8, xx, yy), u8)
    got = oracle.remap(u8, xx + 0.5, yy)
    a = u8.astype(np.int64)
    want = np.zeros_like(a)
    want[:, :-1] = (a[:, :-1] * 16384 + a[:, 1:] * 16384 + 16384) >> 15
    want[:, -1] = (a[:, -1] * 16384 + 16384) >> 15
    assert np.array_equal(got, want.astype(np.uint8))
    # Lanczos4 weights sum to 1 and reproduce constants
    c = np.full((20, 20), 3.25, np.float32)
    yy2, xx2 = np.mgrid[0:20, 0:20].astype(np.float32)
    got = oracle.remap(c, xx2 * 0.5 + 5.3, yy2 * 0.5 + 5.1, oracle.LANCZOS4)
    assert_close(got, c, 1e-6)
    # the reference's own check (transform/rotate.py:26-32): 50x50 ramp, +14 deg then -14 deg,
    # INTER_CUBIC + BORDER_REFLECT about the centre, mean |a-c| < 0.005
    a = np.tile(np.linspace(0, 1, 50), (50, 1))

    def rotate(img, angle):
        s0, s1 = img.shape
        cx, cy = (s0 - 1) / 2., (s1 - 1) / 2.
        t = np.deg2rad(angle)
        al, be = np.cos(t), np.sin(t)  # cv2.getRotationMatrix2D
        M = np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy],
                      [0, 0, 1.]])
        return oracle.warp_perspective(img, np.linalg.inv(M), img.shape,
                                       oracle.CUBIC_CV | oracle.Q5, oracle.REFLECT)
    assert np.abs(a - rotate(rotate(a, 14), -14)).mean() < 0.005


# ---- cv2-specific modes against the independent numpy restatements (cv_modes.npz) ----------
def _lanczos_table(oracle):
    import ctypes as C
    lib = oracle.lib()
    tab = np.zeros((32, 8), np.float32)
    for k in range(32):
        row = np.zeros(8, np.float32)
        lib.orc_lanczos4_weights(C.c_float(k / 32.0), row.ctypes.data_as(C.c_void_p))
        tab[k] = row
    return tab


def test_cv_modes_independent_restatements(oracle):
    """linear_cv_q5 == scipy at coordinates rounded to 1/32 px; Keys a=-0.75 / a=-0.5, Lanczos4
    and the uint8 fixed-point bilinear against second, independently written numpy
    restatements (tests/golden/gen_golden.py::gen_cv_modes).  Measured deviations are printed:
    they are the margin these modes have against a different implementation of the same
    definition - cv2 itself remains unavailable."""
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    img, img8 = g['img'], g['img8']
    worst = {}
    for name in ('radial', 'strong'):
        mx, my = g['mapx_' + name], g['mapy_' + name]
        for cname, cv in (('c0', 0.0), ('c037', 0.37)):
            got = oracle.remap(img, mx, my, oracle.LINEAR | oracle.Q5, oracle.CONSTANT, cv)
            want = g['q5lin_%s_%s' % (name, cname)]
            assert_close(got, want, 0, 3e-7, 'q5 ' + name + cname)
        for key, iid in (('cubic075', oracle.CUBIC_CV), ('cubic075q5', oracle.CUBIC_CV | oracle.Q5),
                         ('cubic05', oracle.CUBIC_KEYS), ('lanczos4', oracle.LANCZOS4)):
            got = oracle.remap(img, mx, my, iid)
            want = g['%s_%s' % (key, name)]
            err = float(np.abs(got.astype(np.float64) - want).max())
            worst[key] = max(worst.get(key, 0.0), err)
            assert_close(got, want, 0, 2e-6, key + ' ' + name)
        assert np.array_equal(oracle.remap(img8, mx, my), g['u8fix_' + name])
        assert np.array_equal(oracle.remap(img8, mx, my, cval=17), g['u8fix17_' + name])
    tab = _lanczos_table(oracle)
    assert_close(tab, g['lanczos_tab'], 0, 2e-7, 'Lanczos4 table')
    print('max |oracle - independent restatement|:', worst)


def _close_u8(got, want, what, max_frac):
    """uint8 results: equal, or (Lanczos4: the two restatements' float32 1-D weights differ in
    the last bit, 2 of the 1024 integer tables in one entry) off by one on a few pixels"""
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1, '%s: max |diff| %d' % (what, d.max())
    frac = float((d != 0).mean())
    assert frac <= max_frac, '%s: %.4f %% of the pixels differ' % (what, 100 * frac)
    return frac


def test_cv_u8_fixed_point_tables(oracle):
    """OpenCV's 8U bicubic / Lanczos4 (short weights with the sum fix-up, integer accumulation):
    the oracle against the independent numpy restatement - tables and remapped images"""
    import ctypes as C
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    lib = oracle.lib()
    for which, kind, ks in ((0, 'cubic', 4), (1, 'lanczos4', 8)):
        t = np.zeros((32, ks), np.float32)
        lib.orc_fixed_tab1d(which, t.ctypes.data_as(C.c_void_p))
        if kind == 'cubic':
            assert np.array_equal(t, g['fixtab1d_cubic'])
        else:
            assert_close(t, g['fixtab1d_lanczos4'], 0, 2.5e-7, 'Lanczos4 float32 weights')
        bad = 0
        it = np.zeros(ks * ks, np.int32)
        for fy in range(32):
            for fx in range(32):
                lib.orc_fixed_weights_2d(ks, fy, fx, it.ctypes.data_as(C.c_void_p))
                assert int(it.sum()) == 32768, (kind, fy, fx)
                d = np.abs(it.reshape(ks, ks) - g['fixtab2d_' + kind][fy, fx])
                assert d.max() <= 1
                bad += int(d.max() != 0)
        assert bad <= (0 if kind == 'cubic' else 8), (kind, bad)
    img8 = g['img8']
    for name in ('radial', 'strong'):
        mx, my = g['mapx_' + name], g['mapy_' + name]
        for kind, iid, frac in (('cubic', oracle.CUBIC_CV, 0.0), ('lanczos4', oracle.LANCZOS4, 0.01)):
            _close_u8(oracle.remap(img8, mx, my, iid), g['u8tab_%s_%s' % (kind, name)],
                      kind + ' ' + name, frac)
            _close_u8(oracle.remap(img8, mx, my, iid, cval=17), g['u8tab17_%s_%s' % (kind, name)],
                      kind + ' cval 17 ' + name, frac)
            # the 1/32-px flag changes nothing: 8U coordinates are always rounded
            assert np.array_equal(oracle.remap(img8, mx, my, iid | oracle.Q5),
                                  oracle.remap(img8, mx, my, iid))


def test_cv_u16_arithmetic_independent_restatement(oracle):
    """OpenCV's remap arithmetic on CV_16U (float32 table weights at 1/32 px, float32
    accumulation, cvRound) - what cv2.remap does on the camera's uint16 frames in
    LensDistortion.correct (camera/LensDistortion.py:323-326): the oracle's restatement against
    the second, independently written float32-numpy one (gen_golden.py::remap_u16_cv_np).
    Bilinear and bicubic bit for bit; Lanczos4 up to the last-bit differences of the two 1-D
    weight tables (off by one on a few pixels).  cv2 itself is absent: unpinned."""
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    img16 = g['img16']
    for name in ('radial', 'strong'):
        mx, my = g['mapx_' + name], g['mapy_' + name]
        for kind, iid in (('linear', oracle.LINEAR | oracle.Q5), ('cubic', oracle.CUBIC_CV | oracle.Q5),
                          ('lanczos4', oracle.LANCZOS4)):
            for key, cv in (('u16cv', 0), ('u16cv1000', 1000)):
                got = oracle.remap(img16, mx, my, iid, oracle.CONSTANT, cv)
                want = g['%s_%s_%s' % (key, kind, name)]
                assert got.dtype == np.uint16
                if kind == 'lanczos4':
                    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
                    assert d.max() <= 1 and (d != 0).mean() < 0.01, (kind, name, d.max(), (d != 0).mean())
                else:
                    assert np.array_equal(got, want), (kind, name, key)
    # the exact-coordinate modes keep the double sums (not the cv2 arithmetic)
    a = oracle.remap(img16, g['mapx_radial'], g['mapy_radial'], oracle.LINEAR)
    b = oracle.remap(img16, g['mapx_radial'], g['mapy_radial'], oracle.LINEAR | oracle.Q5)
    assert not np.array_equal(a, b)


def cv2_warp_cases(g):
    """(key, image, interpolation name, inverse-map flag) of the cv2.warpPerspective vectors that only the
    cv2-generated file carries (gen_cv2_golden.py): PerspectiveCorrection.correct / uncorrect as called"""
    for tag, img in (('f32', g['img']), ('u8', g['img8']), ('u16', g['img16'])):
        for kind in ('linear', 'cubic', 'lanczos4'):
            yield 'warp_%s_%s' % (kind, tag), img, kind, False
        yield 'warpinv_cubic_' + tag, img, 'cubic', True


def test_cv2_warp_perspective_vectors_when_present(oracle):
    """cv2.warpPerspective itself (camera/PerspectiveCorrection.py:401-405 Lanczos4, :377-378
    INTER_CUBIC | WARP_INVERSE_MAP) - only where tests/golden/gen_cv2_golden.py has run under an OpenCV;
    here (no cv2) the test states that and returns"""
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    if 'warp_H' not in g:
        assert pin.startswith('cv2-unpinned')
        return
    H = g['warp_H']
    ids = {'linear': oracle.LINEAR | oracle.Q5, 'cubic': oracle.CUBIC_CV | oracle.Q5, 'lanczos4': oracle.LANCZOS4}
    for key, img, kind, inverse in cv2_warp_cases(g):
        M = H if inverse else np.linalg.inv(H)      # the library takes the destination -> source matrix
        got = oracle.warp_perspective(img, M, img.shape, ids[kind])
        if img.dtype == np.float32:
            assert_close(got, g[key], 0, 2e-6, key)
        else:
            d = np.abs(got.astype(np.int64) - g[key].astype(np.int64))
            assert d.max() <= 1 and (d != 0).mean() <= 0.01, (key, d.max(), (d != 0).mean())


def interp_more_cases(g):
    """(key, callable(api) -> result) for every fixture of interp_more.npz; `api` is the oracle
    module here and the product's interpolate package in the GPU tests"""
    shape = tuple(int(v) for v in g['u_shape'])
    cases = []
    for power in (1, 2, 3):
        cases.append(('u_int_p%d' % power, lambda a, p=power: a.interpolate2dUnstructuredIDW(
            g['u_xi'], g['u_yi'], g['u_vi'], np.zeros(shape), p)))
        cases.append(('u_flt_p%d' % power, lambda a, p=power: a.interpolate2dUnstructuredIDW(
            g['u_xf'], g['u_yf'], g['u_vf'], np.zeros(shape), p)))
    cases.append(('u_int32_p2', lambda a: a.interpolate2dUnstructuredIDW(
        g['u_xi'], g['u_yi'], g['u_vi'], np.zeros(shape, np.float32), 2)))
    for name in ('sq', 'wide'):
        cx, cy = (int(v) for v in g['c_centre_' + name])
        for kern, power, fr, fphi in ((5, 2, 1, 0.2), (15, 2, 1, 1), (7, 1, 2, 0.5)):
            key = 'c_%s_k%d_p%d_fr%g_fphi%g' % (name, kern, power, fr, fphi)
            cases.append((key, lambda a, n=name, k=kern, p=power, r=fr, f=fphi, x=cx, y=cy:
                          a.interpolateCircular2dStructuredIDW(g['c_grid_' + n].copy(),
                                                               g['c_mask_' + n], k, p, r, f, x, y)))
    cases.append(('c32_sq_k5', lambda a: a.interpolateCircular2dStructuredIDW(
        g['c_grid_sq'].astype(np.float32), g['c_mask_sq'], 5, 2, 1, 0.2, 25, 25)))
    for name, kern in (('sq', 5), ('tall', 4), ('wide', 6)):
        for power in (2, 1):
            cases.append(('x_%s_k%d_p%d' % (name, kern, power), lambda a, n=name, k=kern, p=power:
                          a.interpolate2dStructuredCrossAvg(g['x_grid_' + n].copy(),
                                                            g['x_mask_' + n], k, p)))
    return cases


def test_interp_more(oracle):
    """interpolate2dUnstructuredIDW / interpolateCircular2dStructuredIDW /
    interpolate2dStructuredCrossAvg against the reference's own output"""
    g = load_golden('interp_more.npz')
    cases = interp_more_cases(g)
    assert len(cases) == 20
    for key, run in cases:
        want = g[key]
        got = run(oracle)
        assert got.dtype == want.dtype and got.shape == want.shape, key
        tol = F32 if want.dtype == np.float32 else 1e-12
        if key == 'c32_sq_k5':
            # the interpreted source sums `python float * np.float32` in float32 (NEP 50 weak
            # scalars); numba types the accumulators float64, which the oracle follows
            tol = 2e-6
        assert_close(got, want, tol, 1e-15 if tol < 1e-9 else 0, what=key)
    # float32 grids: _localAvg accumulates in float64 under numba (int 0 + float32 unifies to
    # float64) - the interpreted shim would accumulate in float32, so this case is compared
    # with the float64 fixture at float32 resolution
    got = oracle.interpolate2dStructuredCrossAvg(g['x_grid_sq'].astype(np.float32),
                                                 g['x_mask_sq'], 5, 2)
    assert got.dtype == np.float32
    assert_close(got, g['x_sq_k5_p2'], 1e-6, 0, what='x32')
    # a grid with fewer columns than rows indexes out of bounds in the circular source
    with pytest.raises(RuntimeError):
        oracle.interpolateCircular2dStructuredIDW(np.zeros((8, 6)), np.zeros((8, 6), bool))


def fast_filter_cases(g):
    """(key, image, ksize, every, fn, smooth) of fast_filter.npz"""
    cases = []
    for key in g:
        if not key.startswith('ff_'):
            continue
        p = key.split('_')
        every = None if p[3] == 'eNone' else int(p[3][1:])
        cases.append((key, g['img_nan' if p[1] == 'nan' else 'img'], int(p[2][1:]), every, p[4],
                      2 if key.endswith('smooth2') else 0))
    return cases


def test_fast_filter_statistics(oracle):
    """filters/fastFilter.py with resize=False against the reference's own output"""
    g = load_golden('fast_filter.npz')
    cases = fast_filter_cases(g)
    assert len(cases) == 21
    for key, img, ksize, every, fn, smooth in cases:
        got = oracle.fastFilter(img, ksize, every, resize_=False, fn=fn, smoothksize=smooth)
        assert_close(got, g[key], 1e-12, 0, key)


def test_cv_resize_restatement_properties(oracle):
    """cv2.resize cannot run here (cv2-unpinned): what the restatement must satisfy whatever the
    OpenCV version - identities of the published algorithm, checked against plain numpy"""
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        a = rng.random((37, 53)).astype(dt)
        L, Cb, A, Z = (oracle.RESIZE_LINEAR, oracle.RESIZE_CUBIC, oracle.RESIZE_AREA,
                       oracle.RESIZE_LANCZOS4)
        # same size: every position has fraction 0 -> the image itself, for every kernel
        for interp in (L, Cb, Z, A):
            assert np.array_equal(oracle.resize(a, a.shape, interp), a), interp
        # integer-factor area = block mean; sums grouped by four, scaled by float32(1 / area)
        b = rng.random((36, 52)).astype(dt)
        want = b.reshape(18, 2, 26, 2).transpose(0, 2, 1, 3).reshape(18, 26, 4).astype(dt)
        want = ((want[..., 0] + want[..., 1] + want[..., 2] + want[..., 3]) *
                dt(np.float32(0.25))).astype(dt)
        assert np.array_equal(oracle.resize(b, (18, 26), A), want)
        # (the scale is float32(1 / 12) also for float64 images)
        assert_close(oracle.resize(b, (12, 13), A), b.reshape(12, 3, 13, 4).mean(axis=(1, 3)), 1e-6)
        # fractional area: weights of a destination cell sum to 1 -> constants stay constant,
        # the mean is kept up to rounding
        c = oracle.resize(a, (10, 17), A)
        assert abs(c.mean() - a.mean()) < 2e-3
        assert_close(oracle.resize(np.full((37, 53), 3.5, dt), (10, 17), A), np.full((10, 17), 3.5),
                     1e-6)
        # bilinear upscaling by 2 of a ramp: interior values are the ramp at (d + 0.5) / 2 - 0.5
        ramp = np.tile(np.arange(20, dtype=dt), (6, 1))
        up = oracle.resize(ramp, (12, 40), L)
        x = (np.arange(40) + 0.5) / 2 - 0.5
        assert_close(up[3, 1:-1], x[1:-1], 1e-6)
        assert up[3, 0] == 0 and up[3, -1] == 19            # clamped to the edge pixels
        # kernels that sum to 1 keep constants (float32 coefficients: to rounding)
        for interp in (L, Cb, Z):
            assert_close(oracle.resize(np.full((9, 11), 2.0, dt), (31, 40), interp),
                         np.full((31, 40), 2.0), 1e-6)
    with pytest.raises(RuntimeError):   # INTER_AREA upscaling is a different algorithm in OpenCV
        oracle.resize(np.zeros((4, 4), np.float32), (8, 8), oracle.RESIZE_AREA)
    # fastFilter / fastMean end to end run (shape and smoothness only: cv2-unpinned)
    img = rng.random((120, 171)) * 100
    out = oracle.fastFilter(img, 30)
    assert out.shape == img.shape and np.isfinite(out).all()
    fm = oracle.fastMean(img.astype(np.float32), 10)
    assert fm.shape == img.shape and fm.dtype == np.float32 and abs(fm.mean() - img.mean()) < 1.0


def cv_resize_cases(g):
    ids = {'linear': 1, 'cubic': 2, 'area': 3, 'lanczos4': 4}
    for key in g:
        if key.startswith('img_') or key.startswith('aimg_'):
            continue
        kind, tag, size = key.split('_')
        dh, dw = (int(v) for v in size.split('x'))
        yield key, g[('aimg_' if kind == 'area' else 'img_') + tag], (dh, dw), kind, ids[kind]


def test_cv_resize_independent_restatement(oracle):
    """cv2.resize: oracle.c against the second, vectorised numpy restatement
    (tests/golden/gen_golden.py::resize_np).  Bilinear, bicubic and both INTER_AREA forms agree bit
    for bit; Lanczos4 to the last bits of the float32 coefficients (numpy's vectorised sin / cos
    against libm's, as for the remap table in cv_modes.npz)."""
    g, pin = load_cv_golden('cv_resize.npz')
    print(pin)
    n = 0
    for key, src, dsize, kind, oid in cv_resize_cases(g):
        got = oracle.resize(src, dsize, oid)
        if kind == 'lanczos4':
            assert_close(got, g[key], 0, 2e-6, key)
        else:
            assert np.array_equal(got, g[key]), key
        n += 1
    assert n == 34


def test_point_spread_idw_golden():
    """oracle/oracle.c::orc_point_spread_idw against the reference's own output
    (interpolate2dStructuredPointSpreadIDW run through the numba shim, point_spread.npz): float64
    bit for bit - border quirks, in-sweep dependence, window limits included"""
    from oracle import oracle
    g = load_golden('point_spread.npz')
    for name in ('sq', 'wide'):
        grid, mask = g['ps_grid_' + name], g['ps_mask_' + name]
        for kern, power in ((5, 2), (3, 1), (8, 3)):
            got = oracle.interpolate2dStructuredPointSpreadIDW(grid, mask, kern, power)
            assert np.array_equal(got, g['ps_%s_k%d_p%d' % (name, kern, power)]), (name, kern, power)
        assert mask.any()     # copy=True left the caller's mask alone
    got = oracle.interpolate2dStructuredPointSpreadIDW(g['ps_grid_edge'], g['ps_mask_edge'], 4, 2)
    assert np.array_equal(got, g['ps_edge_k4_p2'])
    # the rows that start / end masked: unmasked last pixels of rows / columns were recomputed
    assert (got != g['ps_grid_edge'])[~g['ps_mask_edge']].any()
    # float32 grid: the fixture ran under numpy 2 (python floats are weak: float32 sums), numba
    # and the oracle sum in double
    got = oracle.interpolate2dStructuredPointSpreadIDW(g['ps_grid_sq'].astype(np.float32),
                                                       g['ps_mask_sq'], 5, 2)
    assert got.dtype == np.float32
    assert_close(got, g['ps32_sq_k5_p2'], 2e-6, 0, 'float32')
    # copy=False: in place, the mask ends up empty
    gr, m = g['ps_grid_sq'].copy(), g['ps_mask_sq'].copy()
    r = oracle.interpolate2dStructuredPointSpreadIDW(gr, m, 5, 2, copy=False)
    assert r is gr and not m.any() and np.array_equal(gr, g['ps_sq_k5_p2'])
