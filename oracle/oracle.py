"""ctypes front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of oracle.c.  Nothing under
``imgprocessor_amd/`` imports this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do.

The functions mirror the reference call surface so that tests read like the
reference's own ``__main__`` checks (file:line cited per function).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

U8, U16, F32, F64 = 0, 1, 2, 3
NEAREST, LINEAR, CUBIC_CV, LANCZOS4, CUBIC_KEYS, Q5 = 0, 1, 2, 4, 5, 0x100
CONSTANT, REPLICATE, REFLECT, WRAP, REFLECT101 = 0, 1, 2, 3, 4

_DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16,
       np.dtype(np.float32): F32, np.dtype(np.float64): F64}
_MODES = {'constant': CONSTANT, 'nearest': REPLICATE, 'replicate': REPLICATE,
          'reflect': REFLECT, 'symmetric': REFLECT, 'wrap': WRAP,
          'mirror': REFLECT101, 'reflect101': REFLECT101}


def build(force=False):
    """compile liboracle.so with gcc (oracle/Makefile)"""
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'liboracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        for name in ('orc_remap', 'orc_build_undistort_map', 'orc_undistort',
                     'orc_warp_perspective', 'orc_extend_array', 'orc_conv2d',
                     'orc_masked_convolve', 'orc_gaussian_kernel1d', 'orc_sepconv2d',
                     'orc_conv_ydep', 'orc_std2d', 'orc_idw', 'orc_fast_idw',
                     'orc_remap_conv2d', 'orc_masked_mean', 'orc_nan_max',
                     'orc_masked_median',
                     'orc_median_threshold', 'orc_median_threshold_size', 'orc_calib_prefilter', 'orc_closest_distance',
                     'orc_pos_to_intensity_unc'):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().orc_get_max_threads())


def _dt(a):
    try:
        return _DT[a.dtype]
    except KeyError:
        raise TypeError('oracle: unsupported dtype %s' % a.dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _dbl(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _mode(m):
    return _MODES[m] if isinstance(m, str) else int(m)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError('oracle %s failed rc=%d' % (what, rc))


# ---------------------------------------------------------------- remap ----
def remap(src, mapx, mapy, interp=LINEAR, border=CONSTANT, cval=0.0, out_dtype=None):
    """cv2.remap as called at camera/LensDistortion.py:323-326"""
    src = np.ascontiguousarray(src)
    mapx = np.ascontiguousarray(mapx, dtype=np.float32)
    mapy = np.ascontiguousarray(mapy, dtype=np.float32)
    dh, dw = mapx.shape
    dst = np.empty((dh, dw), dtype=out_dtype or src.dtype)
    _chk(lib().orc_remap(_p(src), _dt(src), C.c_long(src.shape[0]), C.c_long(src.shape[1]),
                         C.c_long(src.shape[1]), _p(mapx), _p(mapy), C.c_long(dw), _p(dst),
                         _dt(dst), C.c_long(dh), C.c_long(dw), C.c_long(dw), C.c_int(interp),
                         C.c_int(_mode(border)), C.c_double(cval)), 'remap')
    return dst


def build_undistort_map(K, dist5, newK, h, w):
    """cv2.initUndistortRectifyMap(K, d, None, newK, (w,h), CV_32FC1) —
    camera/LensDistortion.py:355-357"""
    K, d, nK = _dbl(K, 9), _dbl(dist5, 5), _dbl(newK, 9)
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    _chk(lib().orc_build_undistort_map(_p(K), _p(d), _p(nK), C.c_long(h), C.c_long(w), _p(mx),
                                       _p(my), C.c_long(w)), 'build_undistort_map')
    return mx, my


def undistort(src, K, dist5, newK, interp=LINEAR, border=CONSTANT, cval=0.0, out_dtype=None,
              out_shape=None):
    src = np.ascontiguousarray(src)
    K, d, nK = _dbl(K, 9), _dbl(dist5, 5), _dbl(newK, 9)
    dh, dw = out_shape or src.shape
    dst = np.empty((dh, dw), dtype=out_dtype or src.dtype)
    _chk(lib().orc_undistort(_p(src), _dt(src), C.c_long(src.shape[0]), C.c_long(src.shape[1]),
                             C.c_long(src.shape[1]), _p(K), _p(d), _p(nK), _p(dst), _dt(dst),
                             C.c_long(dh), C.c_long(dw), C.c_long(dw), C.c_int(interp),
                             C.c_int(_mode(border)), C.c_double(cval)), 'undistort')
    return dst


def warp_perspective(src, M_dst2src, out_shape, interp=LINEAR, border=CONSTANT, cval=0.0,
                     out_dtype=None):
    """cv2.warpPerspective with M already the dst->src matrix —
    camera/PerspectiveCorrection.py:377-378 (WARP_INVERSE_MAP) / :401-405 (H inverted)"""
    src = np.ascontiguousarray(src)
    M = _dbl(M_dst2src, 9)
    dh, dw = out_shape
    dst = np.empty((dh, dw), dtype=out_dtype or src.dtype)
    _chk(lib().orc_warp_perspective(_p(src), _dt(src), C.c_long(src.shape[0]),
                                    C.c_long(src.shape[1]), C.c_long(src.shape[1]), _p(M),
                                    _p(dst), _dt(dst), C.c_long(dh), C.c_long(dw), C.c_long(dw),
                                    C.c_int(interp), C.c_int(_mode(border)), C.c_double(cval)),
         'warp_perspective')
    return dst


# -------------------------------------------------------------- filters ----
def extendArrayForConvolution(arr, kernelXY, modex='reflect', modey='reflect'):
    """filters/_extendArrayForConvolution.py:5-97"""
    arr = np.ascontiguousarray(arr)
    kx, ky = kernelXY
    h, w = arr.shape
    out = np.empty((h + 2 * (ky // 2), w + 2 * (kx // 2)), arr.dtype)
    _chk(lib().orc_extend_array(_p(arr), _dt(arr), C.c_long(h), C.c_long(w), C.c_long(kx),
                                C.c_long(ky), C.c_int(_mode(modex)), C.c_int(_mode(modey)),
                                _p(out)), 'extend_array')
    return out


def conv2d(img, kernel, mode='reflect', cval=0.0, mask=None, mode_y=None):
    """centred correlation == scipy.ndimage.correlate(img, kernel, mode=mode)"""
    img = np.ascontiguousarray(img)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    h, w = img.shape
    out = np.empty_like(img)
    m = None
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
    _chk(lib().orc_conv2d(_p(img), _dt(img), C.c_long(h), C.c_long(w), C.c_long(w), _p(k),
                          C.c_long(k.shape[0]), C.c_long(k.shape[1]),
                          _p(m) if m is not None else None, C.c_long(w), _p(out), C.c_long(w),
                          C.c_int(_mode(mode)), C.c_int(_mode(mode_y if mode_y else mode)),
                          C.c_double(cval)), 'conv2d')
    return out


def maskedConvolve(arr, kernel, mask, mode='reflect'):
    """filters/maskedConvolve.py:13-43 (wrapped kernel indices and all)"""
    arr = np.ascontiguousarray(arr)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    assert k.shape[0] == k.shape[1]
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty_like(arr)
    _chk(lib().orc_masked_convolve(_p(arr), _dt(arr), C.c_long(arr.shape[0]),
                                   C.c_long(arr.shape[1]), _p(k), C.c_long(k.shape[0]), _p(m),
                                   C.c_int(_mode(mode)), _p(out)), 'masked_convolve')
    return out


def gaussian_kernel1d(sigma, radius=None, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d"""
    if radius is None:
        radius = int(truncate * float(sigma) + 0.5)
    out = np.empty(2 * radius + 1)
    _chk(lib().orc_gaussian_kernel1d(C.c_double(sigma), C.c_long(radius), _p(out)), 'gk1d')
    return out


def sepconv2d(img, ky, kx, mode='reflect', cval=0.0):
    """scipy.ndimage.gaussian_filter-style: axis 0 with ky, then axis 1 with kx"""
    img = np.ascontiguousarray(img)
    h, w = img.shape
    ky = _dbl(ky) if ky is not None else np.zeros(0)
    kx = _dbl(kx) if kx is not None else np.zeros(0)
    out = np.empty_like(img)
    _chk(lib().orc_sepconv2d(_p(img), _dt(img), C.c_long(h), C.c_long(w), C.c_long(w), _p(ky),
                             C.c_long(ky.size), _p(kx), C.c_long(kx.size), _p(out), C.c_long(w),
                             C.c_int(_mode(mode)), C.c_int(_mode(mode)), C.c_double(cval)),
         'sepconv2d')
    return out


def gaussian_filter(img, sigma, mode='reflect', truncate=4.0):
    """scipy.ndimage.gaussian_filter(img, sigma) for 2-D input"""
    if np.isscalar(sigma):
        sigma = (sigma, sigma)
    ks = [gaussian_kernel1d(s, truncate=truncate) if s > 1e-15 else None for s in sigma]
    return sepconv2d(img, ks[0], ks[1], mode)


def conv_ydep(ext, kernels, h, w):
    """filters/varYSizeGaussianFilter.py:53-68 on the padded array"""
    ext = np.ascontiguousarray(ext)
    kernels = np.ascontiguousarray(kernels, dtype=np.float64)
    out = np.empty((h, w), ext.dtype)
    _chk(lib().orc_conv_ydep(_p(ext), _dt(ext), C.c_long(h), C.c_long(w), _p(kernels),
                             C.c_long(kernels.shape[1]), C.c_long(kernels.shape[2]), _p(out)),
         'conv_ydep')
    return out


def varYSizeGaussianFilter(arr, stdyrange, stdx=0, modex='wrap', modey='reflect'):
    """filters/varYSizeGaussianFilter.py:9-50 (int / tuple stdyrange only:
    the ndarray form raises UnboundLocalError in the reference)"""
    arr = np.ascontiguousarray(arr)
    s0 = arr.shape[0]
    if type(stdyrange) not in (list, tuple):
        stdyrange = (0, stdyrange)
    mn, mx = stdyrange
    stdys = np.linspace(mn, mx, s0)
    kx = int(stdx * 2.5)
    kx += 1 - kx % 2
    ky = int(mx * 2.5)
    ky += 1 - ky % 2
    arr2 = extendArrayForConvolution(arr, (kx, ky), modex, modey)
    inp = np.zeros((ky, kx))
    inp[ky // 2, kx // 2] = 1
    kernels = np.empty((s0, ky, kx))
    for i in range(s0):
        kernels[i] = gaussian_filter(inp, (stdys[i], stdx))
    return conv_ydep(arr2, kernels, *arr.shape)


def standardDeviation2d(img, ksize=5, blurred=None):
    """filters/standardDeviation.py:9-31 — ksize is always expanded to (k,k)
    and used as sigma for the blur (reference quirk, :19-23)"""
    img = np.ascontiguousarray(img)
    ksize = (ksize, ksize)
    if blurred is None:
        blurred = gaussian_filter(img, ksize)
    blurred = np.ascontiguousarray(blurred, dtype=img.dtype)
    std = np.empty_like(img)
    _chk(lib().orc_std2d(_p(img), _dt(img), C.c_long(img.shape[0]), C.c_long(img.shape[1]),
                         C.c_long(ksize[0]), C.c_long(ksize[1]), _p(blurred), _p(std)), 'std2d')
    return std


def maskedFilter(arr, mask, ksize=30, fill_mask=True, fn='median'):
    """filters/maskedFilter.py:12-37: fn == 'mean' -> mean, anything else (default) -> median"""
    fn = 'mean' if fn == 'mean' else 'median'
    mask = np.ascontiguousarray(mask, dtype=bool)
    if fill_mask:
        sel, out = mask, arr
        assert arr.flags.c_contiguous
    else:
        sel, out = ~mask, np.full_like(arr, fill_value=np.nan)
    use = np.ascontiguousarray(~mask, dtype=np.uint8)
    sel = np.ascontiguousarray(sel, dtype=np.uint8)
    arr = np.ascontiguousarray(arr)
    f = lib().orc_masked_mean if fn == 'mean' else lib().orc_masked_median
    _chk(f(_p(arr), _dt(arr), _p(sel), _p(use), C.c_long(arr.shape[0]), C.c_long(arr.shape[1]),
           C.c_long(ksize // 2), _p(out)), 'masked_' + fn)
    return out


def nan_maximum_filter(arr, ksize):
    """filters/nan_maximum_filter.py:6-14"""
    arr = np.ascontiguousarray(arr)
    out = np.empty_like(arr)
    _chk(lib().orc_nan_max(_p(arr), _dt(arr), C.c_long(arr.shape[0]), C.c_long(arr.shape[1]),
                           C.c_long(ksize // 2), _p(out)), 'nan_max')
    return out


def medianThreshold(img, threshold=0.1, size=3, condition='>', copy=True):
    """filters/medianThreshold.py:7-30 -> (img, indices)"""
    if not threshold > 0:
        return img, None
    src = np.ascontiguousarray(img)
    out = np.empty_like(src)
    idx = np.empty(src.shape, np.uint8)
    _chk(lib().orc_median_threshold_size(_p(src), _dt(src), C.c_long(src.shape[0]),
                                         C.c_long(src.shape[1]), C.c_int(int(size)),
                                         C.c_double(threshold), C.c_int(condition != '>'), _p(out),
                                         _p(idx)), 'median_threshold')
    if copy:
        return out, idx.astype(bool)
    img[...] = out
    return img, idx.astype(bool)


def calib_prefilter(img, bg=None, ff=None, threshold=0.1):
    """camera/CameraCalibration.py:416-437 + :505, :527-528, :566-567: dark current,
    flat field, nan_to_num, thresholded 3x3 median — returns a new array"""
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    b = None if bg is None else np.ascontiguousarray(bg, dtype=img.dtype)
    f = None if ff is None else np.ascontiguousarray(ff, dtype=img.dtype)
    _chk(lib().orc_calib_prefilter(_p(img), _dt(img), _p(b) if b is not None else None,
                                   _p(f) if f is not None else None, C.c_long(img.shape[0]),
                                   C.c_long(img.shape[1]), C.c_double(threshold), _p(out)),
         'calib_prefilter')
    return out


def closestDirectDistance(arr, ksize=30, dtype=np.uint16):
    """render/closestDirectDistance.py:6-41"""
    a = np.ascontiguousarray(np.asarray(arr) != 0, dtype=np.uint8)
    out = np.zeros(a.shape, dtype=dtype)
    _chk(lib().orc_closest_distance(_p(a), C.c_long(a.shape[0]), C.c_long(a.shape[1]),
                                    C.c_long(ksize), _p(out), _dt(out)), 'closest_distance')
    return out


def positionToIntensityUncertainty(image, sx, sy, kernelSize=None):
    """uncertainty/positionToIntensityUncertainty.py:52-89 (integer kernelSize)"""
    vari = isinstance(sx, np.ndarray)
    if kernelSize is None:
        kernelSize = max(3, 4 * (max(sx.max(), sy.max()) if vari else max(sx, sy)) + 1)
    size = max(1, int(kernelSize) // 2)
    image = np.ascontiguousarray(image)
    if image.dtype.kind in 'ui':
        image = image.astype(np.float64)
    sxa = np.ascontiguousarray(sx, dtype=np.float64).reshape(-1)
    sya = np.ascontiguousarray(sy, dtype=np.float64).reshape(-1)
    sint = np.zeros(image.shape)
    _chk(lib().orc_pos_to_intensity_unc(_p(image), _dt(image), C.c_long(image.shape[0]),
                                        C.c_long(image.shape[1]), _p(sxa), _p(sya),
                                        C.c_int(int(vari)), C.c_long(size), _p(sint)),
         'pos_to_intensity_unc')
    return sint


# ---------------------------------------------------------- interpolate ----
def idw_weights(kernel, power=2, fx=1, fy=1):
    """interpolate/interpolate2dStructuredIDW.py:16-21 (centre left at 0 here;
    it is np.empty garbage in the reference and never read)"""
    w = np.zeros((2 * kernel + 1, 2 * kernel + 1))
    for xi in range(-kernel, kernel + 1):
        for yi in range(-kernel, kernel + 1):
            dist = ((fx * xi) ** 2 + (fy * yi) ** 2)
            if dist:
                w[xi + kernel, yi + kernel] = 1 / dist ** (0.5 * power)
    return w


def interpolate2dStructuredIDW(grid, mask, kernel=15, power=2, fx=1, fy=1):
    """interpolate/interpolate2dStructuredIDW.py:8-65 (in place, returns grid)"""
    assert grid.flags.c_contiguous
    w = idw_weights(kernel, power, fx, fy)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    _chk(lib().orc_idw(_p(grid), _dt(grid), _p(m), C.c_long(grid.shape[0]),
                       C.c_long(grid.shape[1]), C.c_long(kernel), _p(w)), 'idw')
    return grid


def growPositions(ksize):
    """utils/growPositions.py:5-31 — offsets sorted by distance (stable order
    of numpy argsort on the same distances), centre dropped"""
    i = ksize * 2 + 1
    dist = np.fromfunction(lambda x, y: ((x - ksize) ** 2 + (y - ksize) ** 2) ** 0.5, (i, i))
    pos = np.dstack(np.unravel_index(np.argsort(dist.ravel()), (i, i)))[0, 1:]
    return pos - ksize, dist[pos[:, 0], pos[:, 1]]


def interpolate2dStructuredFastIDW(grid, mask, kernel=15, power=2, minnvals=5):
    """interpolate/interpolate2dStructuredFastIDW.py:9-63"""
    assert grid.flags.c_contiguous
    indices, dist = growPositions(kernel)
    weights = np.ascontiguousarray(1 / dist ** (0.5 * power))
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    _chk(lib().orc_fast_idw(_p(grid), _dt(grid), _p(m), C.c_long(grid.shape[0]),
                            C.c_long(grid.shape[1]), _p(idx), _p(weights), C.c_long(len(weights)),
                            C.c_long(minnvals - 1)), 'fast_idw')
    return grid


# -------------------------------------------------------- headline chain ----
def interpolate2dUnstructuredIDW(x, y, v, grid, power=2):
    """interpolate/interpolate2dUnstructuredIDW.py:7-38 (in place, returns grid)"""
    assert grid.flags.c_contiguous
    x, y, v = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, v))
    _chk(lib().orc_unstructured_idw(_p(grid), _dt(grid), C.c_long(grid.shape[0]),
                                    C.c_long(grid.shape[1]), _p(x), _p(y), _p(v),
                                    C.c_long(len(v)), C.c_double(power)), 'unstructured_idw')
    return grid


def interpolateCircular2dStructuredIDW(grid, mask, kernel=15, power=2, fr=1, fphi=1, cx=0, cy=0):
    """interpolate/interpolateCircular2dStructuredIDW.py:7-69 (in place, returns grid)"""
    assert grid.flags.c_contiguous
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    _chk(lib().orc_circular_idw(_p(grid), _dt(grid), _p(m), C.c_long(grid.shape[0]),
                                C.c_long(grid.shape[1]), C.c_long(kernel), C.c_double(power),
                                C.c_double(fr), C.c_double(fphi), C.c_double(cx),
                                C.c_double(cy)), 'circular_idw')
    return grid


def interpolate2dStructuredPointSpreadIDW(grid, mask, kernel=15, power=2, maxIter=1e5, copy=True):
    """interpolate/interpolate2dStructuredPointSpreadIDW.py:7-141 (copy=False: grid and mask are
    modified in place, as there)"""
    assert grid.shape == mask.shape, 'grid and mask shape are different'
    if copy:
        grid, mask = grid.copy(), mask.copy()
    assert grid.flags.c_contiguous and mask.flags.c_contiguous and mask.dtype == np.bool_
    m = mask.view(np.uint8)
    _chk(lib().orc_point_spread_idw(_p(grid), _dt(grid), _p(m), C.c_long(grid.shape[0]),
                                    C.c_long(grid.shape[1]), C.c_long(kernel), C.c_double(power),
                                    C.c_long(int(min(maxIter, 2 ** 62)))), 'point_spread_idw')
    return grid


def interpolate2dStructuredCrossAvg(grid, mask, kernel=15, power=2):
    """interpolate/interpolate2dStructuredCrossAvg.py:7-115 (in place, returns grid)"""
    assert grid.flags.c_contiguous
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    _chk(lib().orc_cross_avg(_p(grid), _dt(grid), _p(m), C.c_long(grid.shape[0]),
                             C.c_long(grid.shape[1]), C.c_long(kernel), C.c_double(power)),
         'cross_avg')
    return grid


RESIZE_LINEAR, RESIZE_CUBIC, RESIZE_AREA, RESIZE_LANCZOS4 = 1, 2, 3, 4
_FF_FN = {'median': 0, 'nanmedian': 1, 'mean': 2, 'nanmean': 3}


def resize(img, dsize_hw, interpolation=RESIZE_LINEAR):
    """cv2.resize(img, (w, h), interpolation=...) for 2-D float32 / float64 images - restated
    from OpenCV's published algorithm, cv2-unpinned (oracle.c: orc_resize)"""
    img = np.ascontiguousarray(img)
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    out = np.empty((dh, dw), img.dtype)
    _chk(lib().orc_resize(_p(img), _dt(img), C.c_long(img.shape[0]), C.c_long(img.shape[1]),
                          _p(out), C.c_long(dh), C.c_long(dw), C.c_int(interpolation)), 'resize')
    return out


def fastFilter(arr, ksize=30, every=None, resize_=True, fn='median',
               interpolation=RESIZE_LANCZOS4, smoothksize=0):
    """filters/fastFilter.py:9-48 (borderMode is accepted and unused there)"""
    if every is None:
        every = max(ksize // 3, 1)
    else:
        assert ksize >= 3 * every
    arr = np.ascontiguousarray(arr)
    s0, s1 = arr.shape[:2]
    ss0 = s0 // every
    every = s0 // ss0
    n0, n1 = -(-s0 // every), -(-s1 // every)
    out = np.empty((n0, n1))
    _chk(lib().orc_fast_filter_stat(_p(arr), _dt(arr), C.c_long(s0), C.c_long(s1),
                                    C.c_long(ksize), C.c_long(every), C.c_int(_FF_FN[fn]),
                                    _p(out)), 'fast_filter_stat')
    out = np.ascontiguousarray(out[:n0 - 1, :n1 - 1])   # the loops' LAST indices used as sizes
    if smoothksize:
        out = gaussian_filter(out, smoothksize)
    if not resize_:
        return out
    return resize(out, (s0, s1), interpolation)


def fastMean(img, f=10):
    """filters/fastMean.py:5-19: INTER_AREA down to round(shape / f), INTER_LINEAR back up"""
    s0, s1 = img.shape[:2]
    small = resize(img, (int(round(s0 / f)), int(round(s1 / f))), RESIZE_AREA)
    return resize(small, (s0, s1), RESIZE_LINEAR)


def remap_conv2d(src, mapx, mapy, kernel, interp=LINEAR, border=CONSTANT, cval=0.0,
                 cmode='reflect', out_dtype=np.float32):
    """undistort (map-based) then K x K centred correlation: the benchmark chain"""
    src = np.ascontiguousarray(src)
    mapx = np.ascontiguousarray(mapx, dtype=np.float32)
    mapy = np.ascontiguousarray(mapy, dtype=np.float32)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    h, w = src.shape
    tmp = np.empty((h, w), out_dtype)
    dst = np.empty((h, w), out_dtype)
    _chk(lib().orc_remap_conv2d(_p(src), _dt(src), C.c_long(h), C.c_long(w), _p(mapx), _p(mapy),
                                _p(k), C.c_long(k.shape[0]), C.c_long(k.shape[1]), _p(tmp),
                                _p(dst), _dt(dst), C.c_int(interp), C.c_int(_mode(border)),
                                C.c_double(cval), C.c_int(_mode(cmode)), C.c_int(_mode(cmode))),
         'remap_conv2d')
    return dst


# ------------------------------------------------ host geometry helpers ----
def get_perspective_transform(src_pts, dst_pts):
    """cv2.getPerspectiveTransform semantics: H with H·src ~ dst (8x8 solve, h22=1) —
    camera/PerspectiveCorrection.py:149-150"""
    s = np.asarray(src_pts, dtype=np.float64)
    d = np.asarray(dst_pts, dtype=np.float64)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = s[i]
        u, v = d[i]
        A[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i] = u
        b[i + 4] = v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)
