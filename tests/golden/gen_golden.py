#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ — runs ONLY in the build
container (needs /root/reference, scipy; the skimage part needs
/opt/conda/bin/python3.9 and is run as a child process).

How the reference is executed: its in-tree numba stencils are plain
Python+numpy under ``@jit(nopython=True)``; a throw-away identity shim for the
``numba`` module (created in a temp dir, never committed) lets the SAME source
run interpretively.  The cv2-backed parts (cv2 is an un-vendored, un-pinned
dependency that is not installable here) cannot be imported; for those the
fixtures hold the outputs of the libraries the north-star names as the parity
target: scipy.ndimage.map_coordinates(order=1, 'grid-constant') and
skimage.transform.warp(order=1/3).

Fixtures are DATA (inputs + expected outputs as .npz); no reference source,
bytecode or media file is copied.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz
"""
import os
import subprocess
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
warnings.filterwarnings('ignore')


def install_shim():
    d = tempfile.mkdtemp(prefix='numba_shim_')
    os.makedirs(os.path.join(d, 'numba'))
    with open(os.path.join(d, 'numba', '__init__.py'), 'w') as f:
        f.write(
            "import types\n"
            "def _ident(*a, **k):\n"
            "    if len(a) == 1 and isinstance(a[0], types.FunctionType) and not k:\n"
            "        return a[0]\n"
            "    return lambda fn: fn\n"
            "jit = njit = vectorize = guvectorize = _ident\n"
            "class _Sig(object):\n"
            "    # a type name of a numba signature: boolean(boolean[:, :], boolean[:, :])\n"
            "    def __getitem__(self, k):\n"
            "        return self\n"
            "    def __call__(self, *a, **k):\n"
            "        return self\n"
            "boolean = _Sig()\n")
    sys.path.insert(0, d)
    sys.path.insert(1, REF)


def synth(shape, seed, dtype=np.float32):
    """SURVEY §8(d) image content: smooth + noise, clipped to [0,1]"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
    img = 0.5 + 0.25 * np.sin(2 * np.pi * x / 97) + 0.25 * np.cos(2 * np.pi * y / 61)
    img += 0.05 * rng.standard_normal(shape)
    return np.clip(img, 0, 1).astype(dtype)


def gauss2d(n, sigma=1.0):
    g = np.exp(-0.5 * (np.arange(n) - n // 2) ** 2 / sigma ** 2)
    g /= g.sum()
    return np.outer(g, g)


def gen_stencils():
    import io
    import contextlib
    from imgProcessor.filters._extendArrayForConvolution import extendArrayForConvolution
    from imgProcessor.filters.maskedConvolve import maskedConvolve
    from imgProcessor.filters.varYSizeGaussianFilter import varYSizeGaussianFilter
    from imgProcessor.filters.standardDeviation import standardDeviation2d
    from imgProcessor.interpolate.interpolate2dStructuredIDW import interpolate2dStructuredIDW
    from imgProcessor.interpolate.interpolate2dStructuredFastIDW import \
        interpolate2dStructuredFastIDW
    from imgProcessor.utils.growPositions import growPositions

    # (2) extendArrayForConvolution ------------------------------------
    out = {}
    arr = synth((20, 26), 7, np.float64)
    out['arr'] = arr
    for (kx, ky) in [(3, 3), (5, 5), (9, 5), (1, 5), (5, 1)]:
        for modex in ('reflect', 'wrap'):
            out['ext_kx%d_ky%d_%s' % (kx, ky, modex)] = extendArrayForConvolution(
                arr, (kx, ky), modex=modex, modey='reflect')
    np.savez_compressed(os.path.join(HERE, 'extend_array.npz'), **out)

    # (1) maskedConvolve -------------------------------------------------
    out = {}
    H, W = 48, 64
    rng = np.random.default_rng(11)
    kernels = {
        'box3': np.ones((3, 3)) / 9,
        'eye5': np.eye(5),
        'asym5': rng.random((5, 5)),
        'gauss5c': gauss2d(5),
        'gauss5s': np.fft.ifftshift(gauss2d(5)),
        'rand7': rng.random((7, 7)) / 49,
        'rand11': rng.random((11, 11)) / 121,
    }
    masks = {'all': np.ones((H, W), bool)}
    m = np.zeros((H, W), bool)
    m[20:24] = True
    m[:, 30:34] = True
    masks['cross'] = m
    masks['rand10'] = rng.random((H, W)) < 0.1
    imgs = {'f32s0': synth((H, W), 0, np.float32), 'f32s1': synth((H, W), 1, np.float32),
            'f64s2': synth((H, W), 2, np.float64)}
    for k, v in kernels.items():
        out['kernel_' + k] = v
    for k, v in masks.items():
        out['mask_' + k] = v
    for k, v in imgs.items():
        out['img_' + k] = v
    cases = []
    for kn in kernels:
        cases.append(('f32s0', kn, 'all'))
    for kn in ('asym5', 'rand7'):
        cases += [('f32s1', kn, 'cross'), ('f64s2', kn, 'rand10'), ('f64s2', kn, 'all')]
    for (im, kn, mk) in cases:
        with contextlib.redirect_stdout(io.StringIO()):  # the reference prints the padded shape
            o = maskedConvolve(imgs[im], kernels[kn], masks[mk])
        out['out_%s_%s_%s' % (im, kn, mk)] = o
    # the reference's own pinned assertion (filters/maskedConvolve.py:56-73), scaled down
    a = np.fromfunction(lambda x, y: np.sin(x) + np.cos(y), (60, 80))
    mk = np.zeros_like(a, dtype=bool)
    mk[40:44] = True
    mk[:, 40:44] = True
    with contextlib.redirect_stdout(io.StringIO()):
        o = maskedConvolve(a, np.eye(5), mk)
    from scipy.ndimage import convolve
    o2 = convolve(a, np.eye(5))
    o2[~mk] = 0
    assert np.allclose(o, o2)
    out['selftest_arr'] = a
    out['selftest_mask'] = mk
    out['selftest_out'] = o
    np.savez_compressed(os.path.join(HERE, 'masked_convolve.npz'), **out)

    # (3) varYSizeGaussianFilter ---------------------------------------
    out = {}
    a = synth((40, 40), 3, np.float64)
    out['arr'] = a
    out['out_0_4_1'] = varYSizeGaussianFilter(a.copy(), (0, 4), 1)
    out['out_3_0'] = varYSizeGaussianFilter(a.copy(), 3, 0)
    an = a.copy()
    an[10, :] = np.nan
    an[25, 3:9] = np.nan
    out['arr_nan'] = an
    out['out_nan_0_4_1'] = varYSizeGaussianFilter(an.copy(), (0, 4), 1)
    out['out_1_3_2_reflect'] = varYSizeGaussianFilter(a.copy(), (1, 3), 2, modex='reflect')
    np.savez_compressed(os.path.join(HERE, 'var_y_gauss.npz'), **out)

    # (4) standardDeviation2d ---------------------------------------------
    out = {}
    a = synth((40, 52), 4, np.float64)
    out['img'] = a
    for k in (5, 11):
        out['std_k%d' % k] = standardDeviation2d(a, ksize=k)
    a32 = synth((40, 52), 5, np.float32)
    out['img32'] = a32
    out['std32_k5'] = standardDeviation2d(a32, ksize=5)
    np.savez_compressed(os.path.join(HERE, 'std2d.npz'), **out)

    # (5) IDW / FastIDW ------------------------------------------------------
    out = {}
    H = W = 64
    g = synth((H, W), 6, np.float64)
    out['grid'] = g
    rng = np.random.default_rng(12)
    for kern in (3, 5, 15):
        m = rng.random((H, W)) < 0.15
        m[10:16, 20:27] = True  # a hole bigger than kernel=3/5 windows can bridge only partly
        m[H - kern - 1:, :] = False  # keep masked px >= kernel+1 from bottom/right (reference OOB quirk)
        m[:, W - kern - 1:] = False
        out['mask_k%d' % kern] = m
        for power in (1, 2, 3):
            gg = g.copy()
            out['idw_k%d_p%d' % (kern, power)] = interpolate2dStructuredIDW(gg, m, kern, power)
        gg = g.copy()
        out['idw_k%d_p2_fx2_fy05' % kern] = interpolate2dStructuredIDW(gg, m, kern, 2, 2, 0.5)
        for power, minn in ((2, 5), (1, 3), (3, 9)):
            gg = g.copy()
            out['fidw_k%d_p%d_n%d' % (kern, power, minn)] = interpolate2dStructuredFastIDW(
                gg, m, kern, power, minn)
    g32 = g.astype(np.float32)
    out['idw32_k5_p2'] = interpolate2dStructuredIDW(g32.copy(), out['mask_k5'], 5, 2)
    # fully masked block in the middle bigger than the window: those px stay untouched
    m = np.zeros((H, W), bool)
    m[20:40, 20:40] = True
    out['mask_block'] = m
    out['idw_block_k3'] = interpolate2dStructuredIDW(g.copy(), m, 3, 2)
    out['fidw_block_k3'] = interpolate2dStructuredFastIDW(g.copy(), m, 3, 2, 5)
    pos, dist = growPositions(4)
    out['grow4_pos'] = pos
    out['grow4_dist'] = dist
    np.savez_compressed(os.path.join(HERE, 'idw.npz'), **out)

    # (f2) maskedFilter (mean) and nan_maximum_filter ------------------------------
    from imgProcessor.filters.maskedFilter import maskedFilter
    from imgProcessor.filters.nan_maximum_filter import nan_maximum_filter
    out = {}
    a = synth((40, 52), 8, np.float64)
    rng = np.random.default_rng(13)
    m = rng.random(a.shape) < 0.2
    m[5:12, 8:20] = True
    out['arr'] = a
    out['mask'] = m
    for ks in (6, 11, 30):
        out['mean_fill_k%d' % ks] = maskedFilter(a.copy(), m, ksize=ks, fill_mask=True, fn='mean')
        out['mean_nofill_k%d' % ks] = maskedFilter(a.copy(), m, ksize=ks, fill_mask=False,
                                                   fn='mean')
    for ks in (6, 11):
        out['median_fill_k%d' % ks] = maskedFilter(a.copy(), m, ksize=ks, fill_mask=True,
                                                   fn='median')
        out['median_nofill_k%d' % ks] = maskedFilter(a.copy(), m, ksize=ks, fill_mask=False,
                                                     fn='median')
    a32 = a.astype(np.float32)
    out['median32_fill_k6'] = maskedFilter(a32.copy(), m, ksize=6, fill_mask=True, fn='median')
    out['mean32_fill_k6'] = maskedFilter(a32.copy(), m, ksize=6, fill_mask=True, fn='mean')
    an = a.copy()
    an[rng.random(a.shape) < 0.3] = np.nan
    an[20:30, 30:45] = np.nan  # a block bigger than the small windows: stays NaN inside
    out['arr_nan'] = an
    for ks in (3, 6, 9):
        out['nanmax_k%d' % ks] = nan_maximum_filter(an, ks)
    np.savez_compressed(os.path.join(HERE, 'masked_filter.npz'), **out)

    # (f2/f4) medianThreshold and the pre-lens stages of CameraCalibration.correct -------
    from imgProcessor.filters.medianThreshold import medianThreshold
    if not hasattr(np, 'asfarray'):  # removed in numpy 2; documented behaviour restored
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)
    out = {}
    rng = np.random.default_rng(14)
    a = 0.2 + synth((45, 61), 9, np.float64)
    spikes = rng.random(a.shape) < 0.03
    a[spikes] *= rng.choice([0.2, 3.0, 10.0], size=spikes.sum())
    a[7, 9] = 0.0
    a[0, 0] = 5.0     # corners / edges: reflect border of the median
    a[44, 60] = 0.01
    out['img'] = a
    for thr, cond in ((0.1, '>'), (0.5, '>'), (0.05, '<')):
        o, ind = medianThreshold(a, thr, condition=cond)
        key = 'thr%s_%s' % (str(thr).replace('.', 'p'), 'gt' if cond == '>' else 'lt')
        out['out_' + key] = o
        out['ind_' + key] = ind
    a32 = a.astype(np.float32)
    o, ind = medianThreshold(a32, 0.1)
    out['out32_thr0p1_gt'], out['ind32_thr0p1_gt'] = o, ind
    z = a.copy()
    z[20:24, 30:34] = 0.0   # 3x3 median == 0 inside: (img-0)/0 = nan (kept) / inf (replaced)
    z[21, 31] = 1.0
    out['img_zero'] = z
    out['out_zero'], out['ind_zero'] = medianThreshold(z, 0.1)
    # other window sizes (round 4): odd, even (scipy shifts the window's origin), larger
    for size in (5, 4, 2, 7, 9):
        out['out_s%d' % size], out['ind_s%d' % size] = medianThreshold(a, 0.1, size=size)
    out['out32_s5'], out['ind32_s5'] = medianThreshold(a32, 0.1, size=5)
    # CameraCalibration.correct stages 2-4 as written at camera/CameraCalibration.py:505
    # (image -= bg), :527-528 (image[i] /= d[i], i = d != 0), :566-567 (nan_to_num, then the
    # reference's medianThreshold in place).  The module itself needs cv2 to import, so the three
    # numpy statements are restated here around the reference's own medianThreshold.
    raw = 40 + 1000 * a
    bg = 40 + rng.standard_normal(a.shape)
    ff = 0.6 + 0.4 * synth((45, 61), 10, np.float64)
    ff[3, 4] = 0.0          # untouched by the division
    raw[10, 10] = np.nan    # -> 0 by nan_to_num
    raw[30, 40] = np.inf    # -> finfo.max by nan_to_num
    out['cal_raw'], out['cal_bg'], out['cal_ff'] = raw, bg, ff
    for thr in (0.1, 0.0):
        image = raw.copy()
        image -= bg
        i = ff != 0
        image[i] /= ff[i]
        if thr > 0:
            image = np.nan_to_num(image)
            medianThreshold(image, thr, copy=False)
        out['cal_out_thr%s' % str(thr).replace('.', 'p')] = image
    np.savez_compressed(os.path.join(HERE, 'median_threshold.npz'), **out)

    # (f2) closestDirectDistance and positionToIntensityUncertainty --------------------------
    from imgProcessor.render.closestDirectDistance import closestDirectDistance
    from imgProcessor.uncertainty.positionToIntensityUncertainty import \
        positionToIntensityUncertainty
    out = {}
    rng = np.random.default_rng(15)
    arr = rng.random((48, 60)) > 0.985
    arr[5, 5] = True
    out['cdd_arr'] = arr
    for ks in (4, 9):
        out['cdd_k%d' % ks] = closestDirectDistance(arr, ksize=ks)
    img = 100 * synth((40, 46), 11, np.float64)
    img[12, 13] = np.nan          # NaN centre: skipped; NaN neighbour: propagates
    out['piu_img'] = img
    out['piu_const_1p5_0p7_k7'] = positionToIntensityUncertainty(img, 1.5, 0.7, 7)
    out['piu_const_2_2_k5'] = positionToIntensityUncertainty(img, 2, 2, 5)
    sxm = 0.5 + 1.5 * synth((40, 46), 12, np.float64)
    sym = 0.4 + 1.0 * synth((40, 46), 13, np.float64)
    out['piu_sx'], out['piu_sy'] = sxm, sym
    out['piu_vari_k7'] = positionToIntensityUncertainty(img, sxm, sym, 7)
    u16 = np.round(img.clip(0, 100) * 40).astype(np.uint16)
    u16[12, 13] = 7
    out['piu_u16'] = u16
    out['piu_u16_const_1_1_k5'] = positionToIntensityUncertainty(u16, 1, 1, 5)
    np.savez_compressed(os.path.join(HERE, 'render_uncertainty.npz'), **out)


# ---------------------------------------------------------------------------
def undistort_map_np(K, d, newK, h, w):
    """documented cv2.initUndistortRectifyMap formula (R = I), float64 numpy,
    written independently of oracle.c"""
    k1, k2, p1, p2, k3 = d
    ir = np.linalg.inv(np.asarray(newK, float))
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    X = ir[0, 0] * u + ir[0, 1] * v + ir[0, 2]
    Y = ir[1, 0] * u + ir[1, 1] * v + ir[1, 2]
    Wd = ir[2, 0] * u + ir[2, 1] * v + ir[2, 2]
    x = X / Wd
    y = Y / Wd
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    K = np.asarray(K, float)
    return ((K[0, 0] * xd + K[0, 2]).astype(np.float32),
            (K[1, 1] * yd + K[1, 2]).astype(np.float32))


def perspective_from_quad(quad, dst):
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i, ((x, y), (u, v)) in enumerate(zip(quad, dst)):
        A[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(A, b), 1).reshape(3, 3)


REMAP_CASES = {
    # name: (fx, fy, cx, cy, [k1,k2,p1,p2,k3])
    'zero': (128.0, 128.0, 63.5, 47.5, [0, 0, 0, 0, 0]),
    'radial': (128.0, 128.0, 63.5, 47.5, [-0.12, 0.03, 1e-3, -5e-4, 0]),
    # in-tree synthetic default of camera/lens/estimateSystematicErrorLensCorrection.py:99-114
    'synthdefault': (96.0, 96.0, 48.0, 64.0, [0.0, 0.01, 0.1, 0.01, 0.001]),
    'strong': (100.0, 110.0, 60.0, 50.0, [0.35, -0.1, 5e-3, 3e-3, 0.02]),
}


def gen_remap_scipy():
    from scipy.ndimage import map_coordinates, correlate, gaussian_filter
    H, W = 96, 128
    out = {}
    img = synth((H, W), 0, np.float32)
    img16 = np.round(synth((H, W), 1, np.float64) * 4095).astype(np.uint16)
    out['img'] = img
    out['img16'] = img16
    for name, (fx, fy, cx, cy, d) in REMAP_CASES.items():
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.]])
        newK = K.copy()
        if name == 'strong':
            newK = np.array([[90., 0, 66], [0, 95., 45], [0, 0, 1.]])
        mx, my = undistort_map_np(K, d, newK, H, W)
        out['K_' + name] = K
        out['newK_' + name] = newK
        out['dist_' + name] = np.array(d, float)
        out['mapx_' + name] = mx
        out['mapy_' + name] = my
        for cv, cname in ((0.0, 'c0'), (np.nan, 'cnan'), (0.37, 'c037')):
            if cname != 'c0' and name not in ('radial', 'strong'):
                continue
            out['lin_%s_%s' % (name, cname)] = map_coordinates(
                img, [my, mx], order=1, mode='grid-constant', cval=cv, output=np.float32)
        out['lin16_%s' % name] = map_coordinates(
            img16.astype(np.float32), [my, mx], order=1, mode='grid-constant', cval=0,
            output=np.float32)
        # other border modes of the same gather (scipy names -> cv2 ids in the test)
        if name == 'strong':
            for smode in ('nearest', 'reflect', 'mirror', 'grid-wrap'):
                out['lin_%s_%s' % (name, smode)] = map_coordinates(
                    img, [my, mx], order=1, mode=smode, output=np.float32)
    # headline chain at fixture size: undistort then 5x5 explicit Gaussian
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    out['k5'] = k5
    und = out['lin_radial_c0']
    out['chain_radial_k5'] = correlate(und, k5.astype(np.float64), mode='reflect')
    # filters as the reference obtains them
    out['corr_box3'] = correlate(img, np.ones((3, 3)) / 9, mode='reflect')
    rk = np.random.default_rng(123).random((7, 7))
    rk /= rk.sum()
    out['k7'] = rk
    out['corr_k7'] = correlate(img, rk, mode='reflect')
    for smode in ('nearest', 'mirror', 'wrap', 'constant'):
        out['corr_k7_' + smode] = correlate(img, rk, mode=smode, cval=0.25)
    rk11 = np.random.default_rng(321).random((11, 11))
    rk11 /= rk11.sum()
    out['k11'] = rk11
    out['corr_k11'] = correlate(img, rk11, mode='reflect')
    out['corr_k3x7'] = correlate(img, rk[:3, :], mode='reflect')
    out['corr_k6x4'] = correlate(img, rk[:6, :4], mode='reflect')  # even sizes: centre k//2
    for s in (0.5, 1.0, 1.25, 2.0):
        out['gauss_s%s' % str(s).replace('.', 'p')] = gaussian_filter(img, s)
    out['gauss_s1_2p5'] = gaussian_filter(img, (1.0, 2.5))
    out['gauss64_s1'] = gaussian_filter(img.astype(np.float64), 1.0)
    np.savez_compressed(os.path.join(HERE, 'remap_scipy.npz'), **out)


# ---------------------------------------------------------------------------------------
# cv2-specific behaviour.  cv2 cannot be imported here, so these
# fixtures are SECOND, INDEPENDENT restatements of OpenCV's published definitions, written in
# vectorised numpy from the textbook forms (piecewise Keys kernel, sinc-product Lanczos kernel,
# integer bilinear weights, the 9 x 9 rectangle search) - not transcriptions of oracle/oracle.c
# or of the product - plus the scipy identity for the 1/32-px coordinate rule.  They pin the
# oracle and the GPU path against a differently-written implementation of the same definition;
# they do NOT pin either against cv2 itself (DESIGN.md section 2 says so).
# ---------------------------------------------------------------------------------------
def _gather(img, iy, ix, cval):
    """img[iy, ix] with a constant border (cv2.BORDER_CONSTANT per tap)"""
    h, w = img.shape
    ok = (iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)
    out = np.full(iy.shape, float(cval))
    out[ok] = img[iy[ok], ix[ok]]
    return out


def keys_kernel(x, a):
    """Keys' cubic convolution kernel, textbook piecewise form"""
    x = np.abs(x)
    return np.where(x <= 1, (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1,
                    np.where(x < 2, a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a, 0.0))


def lanczos4_kernel(x):
    """Lanczos kernel with a = 4: sinc(x) sinc(x / 4) on |x| < 4"""
    return np.where(np.abs(x) < 4, np.sinc(x) * np.sinc(x / 4.0), 0.0)


def remap_separable_np(img, mx, my, kernel, ntaps, q5, cval=0.0, normalise=False):
    """dst = sum_r sum_c w_y[r] w_x[c] src[iy0 + r, ix0 + c] in float64; coordinates exact or
    rounded to 1/32 px (cvRound(c * 32) / 32, round-half-even)"""
    x = mx.astype(np.float64)
    y = my.astype(np.float64)
    if q5:
        x = np.rint(x * 32) / 32
        y = np.rint(y * 32) / 32
    fx, fy = np.floor(x), np.floor(y)
    tx, ty = x - fx, y - fy
    first = -(ntaps // 2 - 1)           # -1 for 4 taps, -3 for 8
    offs = np.arange(first, first + ntaps)
    wx = np.stack([kernel(tx - o) for o in offs])   # tap at fx + o is (tx - o) away
    wy = np.stack([kernel(ty - o) for o in offs])
    if normalise:
        wx = (wx / wx.sum(axis=0)).astype(np.float32).astype(np.float64)
        wy = (wy / wy.sum(axis=0)).astype(np.float32).astype(np.float64)
    out = np.zeros(x.shape)
    src = img.astype(np.float64)
    for r, oy in enumerate(offs):
        row = np.zeros(x.shape)
        for c, ox in enumerate(offs):
            row += wx[c] * _gather(src, (fy + oy).astype(np.int64), (fx + ox).astype(np.int64), cval)
        out += wy[r] * row
    return out


def remap_u8_fixed_np(img8, mx, my, cval8=0):
    """cv2.remap on CV_8U, INTER_LINEAR: coordinates to 1/32 px, integer weights
    (32 - fx)(32 - fy) * 32 ... (they sum to 2^15), rounded shift by 15"""
    qx = np.rint(mx.astype(np.float64) * 32).astype(np.int64)
    qy = np.rint(my.astype(np.float64) * 32).astype(np.int64)
    ix, iy, fx, fy = qx >> 5, qy >> 5, qx & 31, qy & 31
    src = img8.astype(np.int64)
    h, w = src.shape

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = np.full(yy.shape, int(cval8), np.int64)
        v[ok] = src[yy[ok], xx[ok]]
        return v
    acc = (tap(iy, ix) * (32 - fx) * (32 - fy) + tap(iy, ix + 1) * fx * (32 - fy) +
           tap(iy + 1, ix) * (32 - fx) * fy + tap(iy + 1, ix + 1) * fx * fy) * 32
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def cv_fixed_tables_np(kind):
    """OpenCV's tables for remap on CV_8U with INTER_CUBIC / INTER_LANCZOS4, written from the
    published scheme in float32 numpy arithmetic: 1-D weights of the 32 fractions, then per
    fraction pair the ks x ks weights round(wy * wx * 2^15) as shorts; where they do not sum to
    2^15 the difference goes to one weight of the 2 x 2 block starting at (ks/2, ks/2) - the
    largest when the sum is short, the smallest when it is over (first one in row-major order).
    Returns (tab1d (32, ks) float32, itab (32, 32, ks, ks) int64)."""
    f = np.float32
    x = np.arange(32, dtype=f) * f(1.0 / 32)
    if kind == 'cubic':
        A = f(-0.75)
        x1 = x + f(1)
        u = f(1) - x
        c0 = ((A * x1 - f(5) * A) * x1 + f(8) * A) * x1 - f(4) * A
        c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
        c2 = ((A + f(2)) * u - (A + f(3))) * u * u + f(1)
        c3 = f(1) - c0 - c1 - c2
        tab = np.stack([c0, c1, c2, c3], axis=1).astype(f)
    else:
        # Lanczos a = 4 at the 8 taps -3..4, float32 coefficients, normalised in float32 with a
        # sequentially accumulated float32 sum; fraction 0 is the unit impulse
        t = x.astype(np.float64)[:, None] - np.arange(-3, 5)[None, :]
        w = lanczos4_kernel(t).astype(f)
        tab = np.zeros((32, 8), f)
        for i in range(32):
            if x[i] < np.finfo(f).eps:
                tab[i, 3] = 1
                continue
            ssum = f(0)
            for k in range(8):
                ssum = f(ssum + w[i, k])
            inv = f(1) / ssum
            tab[i] = w[i] * inv
    ks = tab.shape[1]
    prod = (tab[:, None, :, None] * tab[None, :, None, :]).astype(f)   # [fy, fx, k1, k2]
    itab = np.clip(np.rint((prod * f(32768)).astype(np.float64)), -32768, 32767).astype(np.int64)
    isum = itab.sum(axis=(2, 3))
    h = ks // 2
    blk = itab[:, :, h:h + 2, h:h + 2].reshape(32, 32, 4)      # row-major scan order
    imin, imax = blk.argmin(axis=2), blk.argmax(axis=2)          # first occurrence
    diff = isum - 32768
    target = np.where(diff < 0, imax, imin)
    for fy in range(32):
        for fx in range(32):
            if diff[fy, fx] != 0:
                k1, k2 = h + target[fy, fx] // 2, h + target[fy, fx] % 2
                v = itab[fy, fx, k1, k2] - diff[fy, fx]
                itab[fy, fx, k1, k2] = ((v + 32768) % 65536) - 32768    # (short)
    return tab, itab


def remap_u8_tab_np(img8, mx, my, kind, cval8=0):
    """cv2.remap on CV_8U with INTER_CUBIC / INTER_LANCZOS4 and BORDER_CONSTANT: coordinates to
    1/32 px, integer weights from cv_fixed_tables_np, (sum + 2^14) >> 15, saturated"""
    _, itab = cv_fixed_tables_np(kind)
    ks = itab.shape[2]
    qx = np.rint(mx.astype(np.float64) * 32).astype(np.int64)
    qy = np.rint(my.astype(np.float64) * 32).astype(np.int64)
    ix0, iy0, fx, fy = (qx >> 5) - (ks // 2 - 1), (qy >> 5) - (ks // 2 - 1), qx & 31, qy & 31
    src = img8.astype(np.int64)
    h, w = src.shape
    acc = np.zeros(mx.shape, np.int64)
    wts = itab[fy, fx]                                            # (H, W, ks, ks)
    for r in range(ks):
        for c in range(ks):
            yy, xx = iy0 + r, ix0 + c
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            v = np.full(yy.shape, int(cval8), np.int64)
            v[ok] = src[yy[ok], xx[ok]]
            acc += v * wts[:, :, r, c]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255)
    # a footprint wholly outside the source is the border value itself
    outside = (ix0 >= w) | (ix0 + ks <= 0) | (iy0 >= h) | (iy0 + ks <= 0)
    out[outside] = int(cval8)
    return out.astype(np.uint8)


def remap_u16_cv_np(img16, mx, my, kind, cval16=0):
    """cv2.remap on CV_16U with BORDER_CONSTANT, written from the published scheme in float32
    numpy arithmetic (every product and every sum an IEEE single, no fused multiply-add):
    coordinates to 1/32 px, 1-D float32 weights (linear: 1 - x, x; cubic / Lanczos4: the tables
    of cv_fixed_tables_np), 2-D weight = the float32 product wy * wx;
      linear             v00 w00 + v01 w01 + v10 w10 + v11 w11 left to right (border value
                         substituted for taps outside);
      cubic, inside      the 16 products summed left to right, row-major;
      Lanczos4, inside   row sums (8 products left to right), added row by row;
      taps outside       sum = cv, then sum += (S - cv) * w for the taps inside the frame;
    cvRound + saturation to uint16."""
    f = np.float32
    ks = {'linear': 2, 'cubic': 4, 'lanczos4': 8}[kind]
    qx = np.rint(mx.astype(np.float64) * 32).astype(np.int64)
    qy = np.rint(my.astype(np.float64) * 32).astype(np.int64)
    ix0, iy0, fx, fy = (qx >> 5) - (ks // 2 - 1), (qy >> 5) - (ks // 2 - 1), qx & 31, qy & 31
    if kind == 'linear':
        x = np.arange(32, dtype=f) * f(1.0 / 32)
        tab = np.stack([f(1) - x, x], axis=1).astype(f)
    else:
        tab = cv_fixed_tables_np(kind)[0]
    wx, wy = tab[fx], tab[fy]                       # (H, W, ks) float32
    src = img16.astype(f)
    h, w = src.shape
    cv = f(cval16)
    S = np.empty((ks, ks) + mx.shape, f)
    ok = np.empty((ks, ks) + mx.shape, bool)
    for r in range(ks):
        for c in range(ks):
            yy, xx = iy0 + r, ix0 + c
            o = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            v = np.full(mx.shape, cv, f)
            v[o] = src[yy[o], xx[o]]
            S[r, c], ok[r, c] = v, o
    W2 = [[(wy[..., r] * wx[..., c]).astype(f) for c in range(ks)] for r in range(ks)]
    inside = ok.all(axis=(0, 1))
    if kind == 'linear':
        out = (S[0, 0] * W2[0][0]).astype(f)
        for (r, c) in ((0, 1), (1, 0), (1, 1)):
            out = (out + (S[r, c] * W2[r][c]).astype(f)).astype(f)
    else:
        if kind == 'cubic':
            ins = None
            for r in range(ks):
                for c in range(ks):
                    pr = (S[r, c] * W2[r][c]).astype(f)
                    ins = pr if ins is None else (ins + pr).astype(f)
        else:
            ins = np.zeros(mx.shape, f)
            for r in range(ks):
                rs = None
                for c in range(ks):
                    pr = (S[r, c] * W2[r][c]).astype(f)
                    rs = pr if rs is None else (rs + pr).astype(f)
                ins = (ins + rs).astype(f)
        brd = np.full(mx.shape, cv, f)
        for r in range(ks):
            for c in range(ks):
                term = ((S[r, c] - cv).astype(f) * W2[r][c]).astype(f)
                brd = np.where(ok[r, c], (brd + term).astype(f), brd)
        out = np.where(inside, ins, brd)
    res = np.clip(np.rint(out.astype(np.float64)), 0, 65535)
    outside = (ix0 >= w) | (ix0 + ks <= 0) | (iy0 >= h) | (iy0 + ks <= 0)
    res[outside] = int(cval16)
    return res.astype(np.uint16)


def optimal_new_camera_matrix_np(K, d, size, alpha):
    """cv2.getOptimalNewCameraMatrix(K, d, (w, h), alpha, (w, h)), OpenCV 4.x definition:
    a 9 x 9 grid of image points is undistorted to ideal coordinates; `outer` bounds all of
    them, `inner` is bounded by the grid's edge points; the new matrix scales the (alpha-blended)
    rectangle onto the (W - 1) x (H - 1) frame; the roi is the inner rectangle under the new
    matrix (ceil of the origin, floor of the extent).  The inverse lens model is iterated to
    convergence here (OpenCV stops after 5 fixed-point steps)."""
    k1, k2, p1, p2, k3 = [float(v) for v in d]
    w, h = size
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    gx, gy = np.meshgrid(np.arange(9) * (w - 1) / 8.0, np.arange(9) * (h - 1) / 8.0)

    def ideal(P):
        x0, y0 = (gx - cx) / fx, (gy - cy) / fy
        x, y = x0.copy(), y0.copy()
        for _ in range(60):
            r2 = x * x + y * y
            rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (x0 - dx) / rad, (y0 - dy) / rad
        if P is not None:
            x, y = x * P[0, 0] + P[0, 2], y * P[1, 1] + P[1, 2]
        return x, y

    def rects(P):
        x, y = ideal(P)
        outer = (x.min(), y.min(), x.max() - x.min(), y.max() - y.min())
        il, ir = x[:, 0].max(), x[:, -1].min()
        it, ib = y[0, :].max(), y[-1, :].min()
        return (il, it, ir - il, ib - it), outer
    inner, outer = rects(None)
    f0 = ((w - 1) / inner[2], (h - 1) / inner[3])
    f1 = ((w - 1) / outer[2], (h - 1) / outer[3])
    c0 = (-f0[0] * inner[0], -f0[1] * inner[1])
    c1 = (-f1[0] * outer[0], -f1[1] * outer[1])
    a = float(alpha)
    M = np.array([[f0[0] * (1 - a) + f1[0] * a, 0, c0[0] * (1 - a) + c1[0] * a],
                  [0, f0[1] * (1 - a) + f1[1] * a, c0[1] * (1 - a) + c1[1] * a], [0, 0, 1.0]])
    inner, _ = rects(M)
    x0, y0 = int(np.ceil(inner[0])), int(np.ceil(inner[1]))
    x1, y1 = x0 + int(np.floor(inner[2])), y0 + int(np.floor(inner[3]))
    x0c, y0c, x1c, y1c = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    return M, np.array([x0c, y0c, max(x1c - x0c, 0), max(y1c - y0c, 0)])


def gen_cv_modes():
    from scipy.ndimage import map_coordinates
    H, W = 96, 128
    out = {}
    img = synth((H, W), 4, np.float32)
    img8 = np.round(synth((H, W), 5, np.float64) * 255).astype(np.uint8)
    out['img'] = img
    out['img8'] = img8
    img16 = np.round(synth((H, W), 6, np.float64) * 65535).astype(np.uint16)
    out['img16'] = img16
    for name in ('radial', 'strong'):
        fx, fy, cx, cy, d = REMAP_CASES[name]
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.]])
        newK = K.copy()
        if name == 'strong':
            newK = np.array([[90., 0, 66], [0, 95., 45], [0, 0, 1.]])
        mx, my = undistort_map_np(K, d, newK, H, W)
        out['mapx_' + name] = mx
        out['mapy_' + name] = my
        # (i) the 1/32-px rule through scipy: linear_cv_q5 == map_coordinates at rounded coords
        qx = np.rint(mx.astype(np.float64) * 32) / 32
        qy = np.rint(my.astype(np.float64) * 32) / 32
        for cv, cname in ((0.0, 'c0'), (0.37, 'c037')):
            out['q5lin_%s_%s' % (name, cname)] = map_coordinates(
                img.astype(np.float64), [qy, qx], order=1, mode='grid-constant',
                cval=cv).astype(np.float32)
        # (ii) independent restatements
        out['cubic075_%s' % name] = remap_separable_np(img, mx, my, lambda t: keys_kernel(t, -0.75),
                                                       4, q5=False).astype(np.float32)
        out['cubic075q5_%s' % name] = remap_separable_np(img, mx, my,
                                                         lambda t: keys_kernel(t, -0.75), 4,
                                                         q5=True).astype(np.float32)
        out['cubic05_%s' % name] = remap_separable_np(img, mx, my, lambda t: keys_kernel(t, -0.5),
                                                      4, q5=False).astype(np.float32)
        out['lanczos4_%s' % name] = remap_separable_np(img, mx, my, lanczos4_kernel, 8, q5=True,
                                                       normalise=True).astype(np.float32)
        out['u8fix_%s' % name] = remap_u8_fixed_np(img8, mx, my, 0)
        out['u8fix17_%s' % name] = remap_u8_fixed_np(img8, mx, my, 17)
        for kind in ('cubic', 'lanczos4'):
            out['u8tab_%s_%s' % (kind, name)] = remap_u8_tab_np(img8, mx, my, kind, 0)
            out['u8tab17_%s_%s' % (kind, name)] = remap_u8_tab_np(img8, mx, my, kind, 17)
        # OpenCV's 16U arithmetic (float32 table weights, float32 accumulation)
        for kind in ('linear', 'cubic', 'lanczos4'):
            out['u16cv_%s_%s' % (kind, name)] = remap_u16_cv_np(img16, mx, my, kind, 0)
            out['u16cv1000_%s_%s' % (kind, name)] = remap_u16_cv_np(img16, mx, my, kind, 1000)
    # the Lanczos4 table itself: rows k / 32, taps -3..4, normalised
    t = np.arange(32) / 32.0
    tab = np.stack([lanczos4_kernel(t - o) for o in range(-3, 5)], axis=1)
    out['lanczos_tab'] = tab / tab.sum(axis=1, keepdims=True)
    # OpenCV's 8U tables (1-D float32 weights, 2-D short weights with the sum fix-up)
    for kind in ('cubic', 'lanczos4'):
        t1, t2 = cv_fixed_tables_np(kind)
        out['fixtab1d_' + kind] = t1
        out['fixtab2d_' + kind] = t2.astype(np.int16)
    # getOptimalNewCameraMatrix(alpha = 0, 1) for three cameras
    cams = {'barrel': (640, 480, [[600., 0, 319.5], [0, 600., 239.5], [0, 0, 1]],
                       [-0.2, 0.05, 0, 0, 0]),
            'pincushion': (640, 480, [[700., 0, 330.0], [0, 690., 250.0], [0, 0, 1]],
                           [0.12, 0.01, 1e-3, -5e-4, 0.0]),
            'c2': (1920, 1080, [[1920., 0, 959.5], [0, 1920., 539.5], [0, 0, 1]],
                   [-0.12, 0.03, 1e-3, -5e-4, 0.0])}
    for name, (w, h, K, d) in cams.items():
        K = np.array(K)
        out['optK_in_%s' % name] = np.concatenate([K.ravel(), d, [w, h]])
        for alpha in (0, 1):
            M, roi = optimal_new_camera_matrix_np(K, d, (w, h), alpha)
            out['optK_%s_a%d' % (name, alpha)] = M
            out['optroi_%s_a%d' % (name, alpha)] = roi
    np.savez_compressed(os.path.join(HERE, 'cv_modes.npz'), **out)


SKIMAGE_CHILD = r'''
import sys, warnings
import numpy as np
warnings.filterwarnings('ignore')
from skimage.transform import warp
d = dict(np.load(sys.argv[1]))
out = {}
img = d['img']
for name in [k[2:] for k in d if k.startswith('M_')]:
    M = d['M_' + name]
    shp = tuple(int(v) for v in d['shape_' + name])
    for order in (1, 3):
        for cname, cv in (('c0', 0.0), ('c05', 0.5)):
            out['warp_%s_o%d_%s' % (name, order, cname)] = warp(
                img.astype(np.float64), M, output_shape=shp, order=order, mode='constant',
                cval=cv, clip=False, preserve_range=True)
    out['warp_%s_o1_edge' % name] = warp(img.astype(np.float64), M, output_shape=shp, order=1,
                                         mode='edge', clip=False, preserve_range=True)
np.savez_compressed(sys.argv[2], **out)
'''


def gen_warp_skimage():
    H, W = 96, 128
    img = synth((H, W), 2, np.float32)
    d = {'img': img}
    # demo quad of camera/PerspectiveCorrection.py:866-869 scaled to the fixture, -> full rect
    quad = np.array([(8, 2), (120, 6), (122, 90), (5, 93)], float)
    for name, (sy, sx, b) in {'quad': (96, 128, 0), 'quadb': (80, 100, 7)}.items():
        dst = np.array([[b, b], [sx - b, b], [sx - b, sy - b], [b, sy - b]], float)
        Hm = perspective_from_quad(quad, dst)
        d['H_' + name] = Hm
        d['M_' + name] = np.linalg.inv(Hm)  # dst -> src
        d['shape_' + name] = np.array([sy, sx])
    # rotation by 7 deg + mild perspective (SURVEY C5 style)
    a = np.deg2rad(7.0)
    R = np.array([[np.cos(a), -np.sin(a), 10.0], [np.sin(a), np.cos(a), -6.0], [2e-4, -1e-4, 1.0]])
    d['M_rot7'] = R
    d['H_rot7'] = np.linalg.inv(R)
    d['shape_rot7'] = np.array([H, W])
    tmp_in = os.path.join(tempfile.mkdtemp(), 'in.npz')
    np.savez(tmp_in, **d)
    tmp_out = tmp_in.replace('in.npz', 'out.npz')
    child = tmp_in.replace('in.npz', 'child.py')
    with open(child, 'w') as f:
        f.write(SKIMAGE_CHILD)
    subprocess.check_call(['/opt/conda/bin/python3.9', child, tmp_in, tmp_out])
    o = dict(np.load(tmp_out))
    d.update(o)
    np.savez_compressed(os.path.join(HERE, 'warp_skimage.npz'), **d)


# ---- cv2.resize, second restatement (numpy, vectorised; independent of oracle.c) -------------
def _resize_axis_np(ssize, dsize, kind, clamp_x):
    """per destination index: first-tap offset and float32 coefficients (resize.cpp)"""
    ks = {'linear': 2, 'cubic': 4, 'lanczos4': 8}[kind]
    scale = 1.0 / (float(dsize) / float(ssize))
    f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    f = (f - s0.astype(np.float32)).astype(np.float32)
    xmax = dsize
    if clamp_x:
        if kind == 'linear':
            lo = s0 < 0
            f[lo], s0[lo] = 0, 0
        over = s0 + ks // 2 >= ssize
        if over.any():
            xmax = int(np.argmax(over))
        if kind == 'linear':
            hi = s0 >= ssize - 1
            f[hi], s0[hi] = 0, ssize - 1
    one = np.float32(1)
    if kind == 'linear':
        co = np.stack([one - f, f], 1)
    elif kind == 'cubic':
        A = np.float32(-0.75)
        c0 = ((A * (f + one) - np.float32(5) * A) * (f + one) + np.float32(8) * A) * (f + one) - np.float32(4) * A
        c1 = ((A + np.float32(2)) * f - (A + np.float32(3))) * f * f + one
        g = one - f
        c2 = ((A + np.float32(2)) * g - (A + np.float32(3))) * g * g + one
        co = np.stack([c0, c1, c2, one - c0 - c1 - c2], 1).astype(np.float32)
    else:
        s45 = 0.70710678118654752440084436210485
        cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
        x = f.astype(np.float64)
        y0 = -(x + 3) * np.pi * 0.25
        s_, c_ = np.sin(y0), np.cos(y0)
        i = np.arange(8)
        y = -(x[:, None] + 3 - i[None, :]) * np.pi * 0.25
        with np.errstate(divide='ignore', invalid='ignore'):
            co = ((cs[None, :, 0] * s_[:, None] + cs[None, :, 1] * c_[:, None]) / (y * y)).astype(np.float32)
        tot = np.zeros(dsize, np.float32)
        for j in range(8):
            tot = tot + co[:, j]
        co = (co * (one / tot)[:, None]).astype(np.float32)
        small = f < np.finfo(np.float32).eps
        co[small] = 0
        co[small, 3] = 1
    return s0, co.astype(np.float32), ks, xmax


def resize_np(img, dsize_hw, kind):
    """cv2.resize(img, (w, h), interpolation=kind) for 2-D float32 / float64, numpy restatement"""
    img = np.asarray(img)
    wt = img.dtype.type
    sh, sw = img.shape
    dh, dw = dsize_hw
    if kind == 'area':
        sx_, sy_ = 1.0 / (float(dw) / sw), 1.0 / (float(dh) / sh)
        ix, iy = int(np.rint(sx_)), int(np.rint(sy_))
        eps = np.finfo(np.float64).eps
        if abs(sx_ - ix) < eps and abs(sy_ - iy) < eps:
            out = np.zeros((dh, dw), img.dtype)
            w1 = sw // ix
            for dy in range(dh):
                sy0 = dy * iy
                if sy0 >= sh:
                    continue
                wfull = min(w1 if sy0 + iy <= sh else 0, dw)
                for dx in range(dw):
                    sx0 = dx * ix
                    if dx < wfull:
                        blk = img[sy0:sy0 + iy, sx0:sx0 + ix].reshape(-1)   # row-major = ofs[k]
                        acc = wt(0)
                        k = 0
                        while k <= blk.size - 4:
                            acc = wt(acc + wt(wt(wt(blk[k] + blk[k + 1]) + blk[k + 2]) + blk[k + 3]))
                            k += 4
                        while k < blk.size:
                            acc = wt(acc + blk[k])
                            k += 1
                        out[dy, dx] = wt(acc * wt(np.float32(1) / np.float32(ix * iy)))
                    elif sx0 < sw:
                        blk = img[sy0:min(sy0 + iy, sh), sx0:min(sx0 + ix, sw)].reshape(-1)
                        acc = wt(0)
                        for v in blk:
                            acc = wt(acc + v)
                        out[dy, dx] = wt(np.float32(acc) / np.float32(blk.size))
            return out

        def tab(ssize, dsize, scale):
            t = []
            for d in range(dsize):
                f1 = d * scale
                f2 = f1 + scale
                cell = min(scale, ssize - f1)
                s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
                s2 = min(s2, ssize - 1)
                s1 = min(s1, s2)
                if s1 - f1 > 1e-3:
                    t.append((d, s1 - 1, np.float32((s1 - f1) / cell)))
                for s in range(s1, s2):
                    t.append((d, s, np.float32(1.0 / cell)))
                if f2 - s2 > 1e-3:
                    t.append((d, s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
            return t
        xt, yt = tab(sw, dw, sx_), tab(sh, dh, sy_)
        out = np.zeros((dh, dw), img.dtype)
        for dy in range(dh):
            acc = np.zeros(dw, img.dtype)
            for (d, sy, beta) in [e for e in yt if e[0] == dy]:
                buf = np.zeros(dw, img.dtype)
                for (dx, sxx, alpha) in xt:
                    buf[dx] = wt(buf[dx] + wt(img[sy, sxx] * wt(alpha)))
                acc = (acc + wt(beta) * buf).astype(img.dtype)
            out[dy] = acc
        return out
    sx0, ax, ks, xmax = _resize_axis_np(sw, dw, kind, True)
    sy0, ay, _, _ = _resize_axis_np(sh, dh, kind, False)
    tmp = np.zeros((sh, dw), img.dtype)
    if ks == 2:
        cols = np.arange(dw)
        inner = cols < xmax
        a0, a1 = ax[:, 0].astype(img.dtype), ax[:, 1].astype(img.dtype)
        sx1 = np.minimum(sx0 + 1, sw - 1)
        tmp[:] = img[:, sx0] * wt(1)
        tmp[:, inner] = (img[:, sx0[inner]] * a0[inner] + img[:, sx1[inner]] * a1[inner])
    else:
        for j in range(ks):
            idx = np.clip(sx0 - (ks // 2 - 1) + j, 0, sw - 1)
            tmp = (tmp + img[:, idx] * ax[:, j].astype(img.dtype)[None, :]).astype(img.dtype)
    out = None
    for k in range(ks):
        idx = np.clip(sy0 - ks // 2 + 1 + k, 0, sh - 1)
        term = (tmp[idx, :] * ay[:, k].astype(img.dtype)[:, None]).astype(img.dtype)
        out = term if out is None else (out + term).astype(img.dtype)
    return out


def gen_cv_resize():
    """cv_resize.npz: inputs and the numpy restatement's outputs (cv2 itself cannot run here)"""
    out = {}
    rng = np.random.default_rng(31)
    for dt in (np.float32, np.float64):
        tag = 'f32' if dt == np.float32 else 'f64'
        a = rng.standard_normal((37, 53)).astype(dt)
        out['img_' + tag] = a
        for kind in ('linear', 'cubic', 'lanczos4'):
            for (dh, dw) in ((120, 171), (20, 31), (37, 53), (50, 40)):
                out['%s_%s_%dx%d' % (kind, tag, dh, dw)] = resize_np(a, (dh, dw), kind)
        b = rng.random((36, 52)).astype(dt)
        out['aimg_' + tag] = b
        for (dh, dw) in ((18, 26), (12, 13), (10, 17), (36, 52), (17, 25)):
            out['area_%s_%dx%d' % (tag, dh, dw)] = resize_np(b, (dh, dw), 'area')
    np.savez_compressed(os.path.join(HERE, 'cv_resize.npz'), **out)


def gen_fast_filter():
    """fast_filter.npz: the strided window statistics of filters/fastFilter.py run from the
    reference's own source (resize=False: the cv2.resize that follows cannot run here).  The
    module imports cv2 for the DEFAULT VALUES of two keyword arguments; a throw-away stand-in
    that only holds those constants lets the import succeed - no cv2 function is called."""
    d = tempfile.mkdtemp(prefix='cv2_consts_')
    with open(os.path.join(d, 'cv2.py'), 'w') as f:
        f.write("INTER_LANCZOS4 = 4\nBORDER_REFLECT = 2\n"
                "def resize(*a, **k):\n    raise RuntimeError('no cv2 here')\n")
    sys.path.insert(0, d)
    try:
        from imgProcessor.filters.fastFilter import fastFilter
    finally:
        sys.path.remove(d)
        sys.modules.pop('cv2', None)
    out = {}
    img = synth((120, 171), 4, np.float64) * 100
    nan = img.copy()
    nan[30:34] = np.nan          # the reference's demo adds NaN stripes (:129-130)
    nan[:, 110:113] = np.nan
    nan[0:40, 0:45] = np.nan     # and a window that is all NaN
    out['img'] = img
    out['img_nan'] = nan
    for ksize, every in ((30, None), (12, 4), (40, 2), (9, 3), (5, 1)):
        for fn in ('median', 'nanmedian', 'mean', 'nanmean'):
            src = nan if fn.startswith('nan') or (ksize, every) == (12, 4) else img
            key = 'ff_%s_k%d_e%s_%s' % ('nan' if src is nan else 'img', ksize, every, fn)
            out[key] = fastFilter(src, ksize, every, resize=False, fn=fn)
    out['ff_img_k12_e4_median_smooth2'] = fastFilter(img, 12, 4, resize=False, fn='median',
                                                     smoothksize=2)
    np.savez_compressed(os.path.join(HERE, 'fast_filter.npz'), **out)


def gen_interp_more():
    """interp_more.npz: the three interpolate/ functions SURVEY §2 lists next to the IDW pair,
    run from the reference's own source through the numba identity shim"""
    from imgProcessor.interpolate.interpolate2dStructuredCrossAvg import \
        interpolate2dStructuredCrossAvg
    from imgProcessor.interpolate.interpolate2dUnstructuredIDW import interpolate2dUnstructuredIDW
    from imgProcessor.interpolate.interpolateCircular2dStructuredIDW import \
        interpolateCircular2dStructuredIDW
    out = {}
    rng = np.random.default_rng(21)
    # unstructured IDW: x is the ROW coordinate (grid[i, j], i against x) -------------------
    gx, gy, n = 40, 56, 12
    xi, yi = rng.integers(0, gx, n), rng.integers(0, gy, n)   # on pixels: exact hits exist
    vi = rng.integers(0, 10, n)
    xf, yf = rng.random(n) * gx, rng.random(n) * gy
    vf = rng.standard_normal(n)
    out.update(u_xi=xi, u_yi=yi, u_vi=vi, u_xf=xf, u_yf=yf, u_vf=vf, u_shape=np.array([gx, gy]))
    for power in (1, 2, 3):
        out['u_int_p%d' % power] = interpolate2dUnstructuredIDW(
            xi, yi, vi, np.zeros((gx, gy)), power)
        out['u_flt_p%d' % power] = interpolate2dUnstructuredIDW(
            xf, yf, vf, np.zeros((gx, gy)), power)
    out['u_int32_p2'] = interpolate2dUnstructuredIDW(xi, yi, vi, np.zeros((gx, gy), np.float32), 2)
    # circular IDW: square grid and one with more columns than rows (columns >= shape[0] are
    # left alone, gy = shape[0] in the source) ---------------------------------------------
    for name, shape in (('sq', (48, 48)), ('wide', (40, 56))):
        g = synth(shape, 8, np.float64)
        m = rng.random(shape) < 0.2
        m[10:18, 12:22] = True
        out['c_grid_' + name] = g
        out['c_mask_' + name] = m
        cx, cy = shape[0] // 2 + 1, shape[1] // 2 + 1
        out['c_centre_' + name] = np.array([cx, cy])
        for kern, power, fr, fphi in ((5, 2, 1, 0.2), (15, 2, 1, 1), (7, 1, 2, 0.5)):
            key = 'c_%s_k%d_p%d_fr%g_fphi%g' % (name, kern, power, fr, fphi)
            out[key] = interpolateCircular2dStructuredIDW(g.copy(), m, kern, power, fr, fphi,
                                                          cx, cy)
    g = out['c_grid_sq'].astype(np.float32)
    out['c32_sq_k5'] = interpolateCircular2dStructuredIDW(g.copy(), out['c_mask_sq'], 5, 2, 1, 0.2,
                                                          25, 25)
    # cross average: masks kept where the source is well defined - every unmasked pixel a
    # search can find lies more than `kernel` px from the bottom / right edge (window clamped to
    # gx, not gx-1), and the first masked pixel in raster order has an unmasked left neighbour
    # (slot 2 is np.empty garbage before that).  Row prefixes masked later exercise the stale
    # slot, rows >= gy - 1 of the tall grid the skipped look-right. ----------------------------
    for name, shape, kern in (('sq', (64, 64), 5), ('tall', (64, 40), 4), ('wide', (40, 64), 6)):
        g = synth(shape, 9, np.float64) + np.linspace(5, 10, shape[1])[None, :]
        m = np.zeros(shape, bool)
        lim0, lim1 = shape[0] - kern - 2, shape[1] - kern - 2
        m[:lim0, :lim1] = rng.random((lim0, lim1)) < 0.25
        m[12:26, 10:24] = True            # the "large empty area" of the docstring
        m[0:3, :] = False
        m[:, 0] = False
        m[30, 0:5] = True                 # no unmasked pixel to the left: stale slot 2
        m[31, 0:2] = True
        out['x_grid_' + name] = g
        out['x_mask_' + name] = m
        for power in (2, 1):
            out['x_%s_k%d_p%d' % (name, kern, power)] = interpolate2dStructuredCrossAvg(
                g.copy(), m, kern, power)
    np.savez_compressed(os.path.join(HERE, 'interp_more.npz'), **out)


def gen_point_spread():
    """point_spread.npz: interpolate2dStructuredPointSpreadIDW run from the reference's own source
    (numba identity shim).  Its wrapper asks for ``np.bool``, which numpy >= 1.24 no longer has:
    the alias is restored for the call, nothing else is touched.  Grids: square, and more columns
    than rows (``if ymx > gx: ymx = gy`` compares the column limit with the ROW count: with
    gy > gx the window then runs to the end of the row - defined behaviour; with gy < gx it could
    leave the array - no such case here)."""
    if not hasattr(np, 'bool'):
        np.bool = bool
    from imgProcessor.interpolate.interpolate2dStructuredPointSpreadIDW import \
        interpolate2dStructuredPointSpreadIDW
    out = {}
    rng = np.random.default_rng(33)
    for name, shape in (('sq', (48, 48)), ('wide', (40, 56))):
        g = synth(shape, 12, np.float64) + np.linspace(0, 2, shape[1])[None, :]
        m = rng.random(shape) < 0.3
        m[14:30, 10:26] = True            # a bigger connected area: several sweeps
        m[0, 0] = False
        out['ps_grid_' + name] = g
        out['ps_mask_' + name] = m
        for kern, power in ((5, 2), (3, 1), (8, 3)):
            out['ps_%s_k%d_p%d' % (name, kern, power)] = interpolate2dStructuredPointSpreadIDW(
                g, m, kern, power)
    # a mask that starts and ends rows on masked pixels: the scan's carry across row / column ends
    # marks the LAST pixel of a row / column (border[i, -1]) - unmasked pixels get recomputed
    g = synth((32, 32), 13, np.float64)
    m = np.zeros((32, 32), bool)
    m[5:9, 0:6] = True
    m[12:20, 26:32] = True
    m[0:4, 10:14] = True
    m[28:32, 18:25] = True
    out.update(ps_grid_edge=g, ps_mask_edge=m)
    out['ps_edge_k4_p2'] = interpolate2dStructuredPointSpreadIDW(g, m, 4, 2)
    g32 = out['ps_grid_sq'].astype(np.float32)
    out['ps32_sq_k5_p2'] = interpolate2dStructuredPointSpreadIDW(g32, out['ps_mask_sq'], 5, 2)
    # copy=False: grid and mask are modified in place
    g, m = out['ps_grid_sq'].copy(), out['ps_mask_sq'].copy()
    r = interpolate2dStructuredPointSpreadIDW(g, m, 5, 2, copy=False)
    assert r is g and not m.any()
    np.savez_compressed(os.path.join(HERE, 'point_spread.npz'), **out)


if __name__ == '__main__':
    install_shim()
    if sys.argv[1:] == ['point_spread']:
        gen_point_spread()
        sys.exit(0)
    if sys.argv[1:] == ['interp_more']:
        gen_interp_more()
        sys.exit(0)
    if sys.argv[1:] == ['fast_filter']:
        gen_fast_filter()
        sys.exit(0)
    if sys.argv[1:] == ['cv_resize']:
        gen_cv_resize()
        sys.exit(0)
    gen_stencils()
    gen_interp_more()
    gen_point_spread()
    gen_fast_filter()
    gen_cv_resize()
    gen_remap_scipy()
    gen_cv_modes()
    gen_warp_skimage()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('%-24s %7.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))
