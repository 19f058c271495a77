"""Functional layer over the C ABI: every function takes either host numpy
arrays (staged through the context workspace, result returned as numpy) or
``DeviceArray``s (enqueued on the context stream, result stays in HBM).

Shapes: (h, w) image or (n, h, w) batch of independent frames.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .device import DeviceArray, default_context, dtype_id, as_frames

INTERPOLATIONS = {
    # exact-coordinate forms (scipy / scikit-image semantics)
    'nearest': L.INTER_NEAREST,
    'linear': L.INTER_LINEAR,                 # == map_coordinates(order=1), skimage order=1
    'cubic': L.INTER_CUBIC_KEYS,              # Keys a=-0.5 == skimage order=3
    'cubic_cv': L.INTER_CUBIC_CV,             # Keys a=-0.75, exact coordinates
    # cv2-style: coordinates rounded to 1/32 px (INTER_BITS=5)
    'linear_cv_q5': L.INTER_LINEAR | L.INTER_Q5,
    'cubic_cv_q5': L.INTER_CUBIC_CV | L.INTER_Q5,
    'lanczos4': L.INTER_LANCZOS4,
}
BORDERS = {
    'constant': L.BORDER_CONSTANT, 'replicate': L.BORDER_REPLICATE, 'nearest': L.BORDER_REPLICATE,
    'edge': L.BORDER_REPLICATE, 'reflect': L.BORDER_REFLECT, 'symmetric': L.BORDER_REFLECT,
    'wrap': L.BORDER_WRAP, 'grid-wrap': L.BORDER_WRAP, 'mirror': L.BORDER_REFLECT101,
    'reflect101': L.BORDER_REFLECT101,
}


def interp_id(v):
    if isinstance(v, str):
        try:
            return INTERPOLATIONS[v]
        except KeyError:
            raise ValueError('unknown interpolation %r (one of %s)' % (v, sorted(INTERPOLATIONS)))
    return int(v)


def border_id(v):
    if isinstance(v, str):
        try:
            return BORDERS[v]
        except KeyError:
            raise ValueError('unknown border mode %r (one of %s)' % (v, sorted(BORDERS)))
    return int(v)


def _is_dev(a):
    return isinstance(a, DeviceArray)


def _ctx_of(*arrs, **kw):
    ctx = kw.get('ctx')
    for a in arrs:
        if _is_dev(a):
            if ctx is not None and a.ctx is not ctx:
                raise ValueError('arrays live on different contexts')
            ctx = a.ctx
    return ctx or default_context()


def _host(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    dtype_id(a.dtype)
    return a


def _p(a):
    return a.ptr if _is_dev(a) else a.ctypes.data_as(C.c_void_p)


def _out_dtype(src_dtype, out_dtype):
    return np.dtype(src_dtype if out_dtype is None else out_dtype)


def _dev_out(ctx, out, shape, dtype):
    if out is None:
        return DeviceArray(ctx, shape, dtype)
    if not _is_dev(out) or out.shape != tuple(shape) or out.dtype != np.dtype(dtype):
        raise ValueError('out must be a DeviceArray of shape %s dtype %s' % (shape, dtype))
    return out


# ------------------------------------------------------------------ maps --
def to_planes(img, ctx=None):
    """(H, W, C) image (host or device) -> device batch (C, H, W), the layout copy on the device
    (ipa_deinterleave_dev): what cv2.remap / warpPerspective do channel by channel
    (camera/LensDistortion.py:323-326, camera/PerspectiveCorrection.py:401-405)"""
    ctx = _ctx_of(img, ctx=ctx)
    d = img if _is_dev(img) else ctx.to_device(img)
    if d.ndim != 3:
        raise ValueError('to_planes expects a (H,W,C) image')
    h, w, c = d.shape
    out = ctx.empty((c, h, w), d.dtype)
    ctx._check(ctx._lib.ipa_deinterleave_dev(ctx.handle, d.ptr, dtype_id(d.dtype), h, w, c, w * c,
                                             out.ptr, w, h * w), 'deinterleave')
    return out


def from_planes(planes, out=None, ctx=None):
    """device batch (C, H, W) -> device (H, W, C) image (ipa_interleave_dev)"""
    ctx = _ctx_of(planes, ctx=ctx)
    if not _is_dev(planes) or planes.ndim != 3:
        raise TypeError('from_planes expects a device batch (C,H,W)')
    c, h, w = planes.shape
    dst = _dev_out(ctx, out, (h, w, c), planes.dtype)
    ctx._check(ctx._lib.ipa_interleave_dev(ctx.handle, planes.ptr, dtype_id(planes.dtype), h, w, c, w,
                                           h * w, dst.ptr, w * c), 'interleave')
    return dst


def build_undistort_map(K, dist5, newK, h, w, ctx=None, device=False):
    """cv2.initUndistortRectifyMap(K, d, None, newK, (w,h), CV_32FC1)"""
    ctx = ctx or default_context()
    lib = ctx._lib
    K, d, nK = L.dbl(np.ravel(K), 9), L.dbl(np.ravel(dist5)[:5], 5), L.dbl(np.ravel(newK), 9)
    if device:
        mx, my = DeviceArray(ctx, (h, w), np.float32), DeviceArray(ctx, (h, w), np.float32)
        ctx._check(lib.ipa_build_undistort_map_dev(ctx.handle, K, d, nK, h, w, mx.ptr, my.ptr, w),
                   'build_undistort_map')
        return mx, my
    mx, my = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    ctx._check(lib.ipa_build_undistort_map(ctx.handle, K, d, nK, h, w, _p(mx), _p(my)),
               'build_undistort_map')
    return mx, my


def _check_dev_maps(mapx, mapy):
    """device maps are read with plain (not range-checked) loads: refuse anything that is not a
    pair of equally shaped 2-D float32 arrays instead of reinterpreting it"""
    if not (_is_dev(mapx) and _is_dev(mapy)):
        raise TypeError('device source needs device maps')
    if mapx.dtype != np.float32 or mapy.dtype != np.float32:
        raise TypeError('mapx/mapy must be float32 (got %s / %s)' % (mapx.dtype, mapy.dtype))
    if mapx.ndim != 2 or tuple(mapx.shape) != tuple(mapy.shape):
        raise ValueError('mapx/mapy must be 2-D and of equal shape (got %s / %s)'
                         % (mapx.shape, mapy.shape))


# ----------------------------------------------------------------- remap --
def remap(src, mapx, mapy, interpolation='linear', border_mode='constant', border_value=0.0,
          out_dtype=None, out=None, ctx=None, map_roi=None):
    """cv2.remap(src, mapx, mapy, ...).  map_roi=(x, y, w, h) (device maps only)
    evaluates just that window of the maps — the keepSize=False crop of
    LensDistortion.correct without computing the discarded border."""
    interp, border = interp_id(interpolation), border_id(border_mode)
    if _is_dev(src):
        ctx = _ctx_of(src, mapx, mapy, ctx=ctx)
        _check_dev_maps(mapx, mapy)
        n, sh, sw = as_frames(src)
        mh, mw = mapx.shape
        dh, dw, moff = mh, mw, 0
        if map_roi is not None:
            rx, ry, rw, rh = [int(v) for v in map_roi]
            if not (0 <= rx and 0 <= ry and rw > 0 and rh > 0 and rx + rw <= mw and ry + rh <= mh):
                raise ValueError('map_roi %s outside the %dx%d maps' % (map_roi, mh, mw))
            dh, dw, moff = rh, rw, (ry * mw + rx) * 4
        odt = _out_dtype(src.dtype, out_dtype)
        oshape = (dh, dw) if src.ndim == 2 else (n, dh, dw)
        dst = _dev_out(ctx, out, oshape, odt)
        ctx._check(ctx._lib.ipa_remap_dev(ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw,
                                          C.c_void_p(mapx.ptr.value + moff),
                                          C.c_void_p(mapy.ptr.value + moff), mw, dst.ptr,
                                          dtype_id(odt), dh, dw, dw, n, sh * sw, dh * dw, interp,
                                          border, float(border_value)), 'remap')
        return dst
    if map_roi is not None:
        raise ValueError('map_roi needs device arrays')
    ctx = ctx or default_context()
    src = _host(src)
    mapx, mapy = _host(mapx, np.float32), _host(mapy, np.float32)
    if mapx.shape != mapy.shape or mapx.ndim != 2:
        raise ValueError('mapx/mapy must be 2-D and of equal shape')
    n, sh, sw = as_frames(src)
    dh, dw = mapx.shape
    odt = _out_dtype(src.dtype, out_dtype)
    dst = np.empty((dh, dw) if src.ndim == 2 else (n, dh, dw), odt)
    ctx._check(ctx._lib.ipa_remap(ctx.handle, _p(src), dtype_id(src.dtype), sh, sw, _p(mapx),
                                  _p(mapy), _p(dst), dtype_id(odt), dh, dw, n, interp, border,
                                  float(border_value)), 'remap')
    return dst


def undistort(src, K, dist5, newK=None, interpolation='linear', border_mode='constant',
              border_value=0.0, out_dtype=None, out_shape=None, out=None, ctx=None):
    """LensDistortion.correct without map arrays (analytic coordinates)"""
    interp, border = interp_id(interpolation), border_id(border_mode)
    if newK is None:
        newK = K
    Kc, dc, nKc = L.dbl(np.ravel(K), 9), L.dbl(np.ravel(dist5)[:5], 5), L.dbl(np.ravel(newK), 9)
    dev = _is_dev(src)
    ctx = _ctx_of(src, ctx=ctx)
    if not dev:
        src = _host(src)
    n, sh, sw = as_frames(src)
    dh, dw = out_shape or (sh, sw)
    odt = _out_dtype(src.dtype, out_dtype)
    oshape = (dh, dw) if len(src.shape) == 2 else (n, dh, dw)
    if dev:
        dst = _dev_out(ctx, out, oshape, odt)
        ctx._check(ctx._lib.ipa_undistort_dev(ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw,
                                              Kc, dc, nKc, dst.ptr, dtype_id(odt), dh, dw, dw, n,
                                              sh * sw, dh * dw, interp, border,
                                              float(border_value)), 'undistort')
        return dst
    dst = np.empty(oshape, odt)
    ctx._check(ctx._lib.ipa_undistort(ctx.handle, _p(src), dtype_id(src.dtype), sh, sw, Kc, dc,
                                      nKc, _p(dst), dtype_id(odt), dh, dw, n, interp, border,
                                      float(border_value)), 'undistort')
    return dst


def warp_perspective(src, M_dst2src, out_shape, interpolation='linear', border_mode='constant',
                     border_value=0.0, out_dtype=None, out=None, ctx=None):
    """cv2.warpPerspective with M the DESTINATION->SOURCE matrix (pass inv(H)
    for a plain call, H for WARP_INVERSE_MAP)"""
    interp, border = interp_id(interpolation), border_id(border_mode)
    M = L.dbl(np.ravel(np.asarray(M_dst2src, dtype=np.float64)), 9)
    dev = _is_dev(src)
    ctx = _ctx_of(src, ctx=ctx)
    if not dev:
        src = _host(src)
    n, sh, sw = as_frames(src)
    dh, dw = int(out_shape[0]), int(out_shape[1])
    odt = _out_dtype(src.dtype, out_dtype)
    oshape = (dh, dw) if len(src.shape) == 2 else (n, dh, dw)
    if dev:
        dst = _dev_out(ctx, out, oshape, odt)
        ctx._check(ctx._lib.ipa_warp_perspective_dev(ctx.handle, src.ptr, dtype_id(src.dtype), sh,
                                                     sw, sw, M, dst.ptr, dtype_id(odt), dh, dw, dw,
                                                     n, sh * sw, dh * dw, interp, border,
                                                     float(border_value)), 'warp_perspective')
        return dst
    dst = np.empty(oshape, odt)
    ctx._check(ctx._lib.ipa_warp_perspective(ctx.handle, _p(src), dtype_id(src.dtype), sh, sw, M,
                                             _p(dst), dtype_id(odt), dh, dw, n, interp, border,
                                             float(border_value)), 'warp_perspective')
    return dst


# --------------------------------------------------------------- filters --
def _float_img(img):
    """filters run on float32/float64 (the reference's toFloatArray rule for ints)"""
    if _is_dev(img):
        if img.dtype not in (np.float32, np.float64):
            raise TypeError('device filters need float32/float64 arrays')
        return img
    img = np.asarray(img)
    if img.dtype not in (np.float32, np.float64):
        img = img.astype(np.float32 if img.dtype.itemsize <= 2 else np.float64)
    return np.ascontiguousarray(img)


def conv2d(img, kernel, mode='reflect', cval=0.0, mask=None, mode_y=None, out=None, ctx=None):
    """centred correlation == scipy.ndimage.correlate(img, kernel, mode=mode, cval=cval);
    mode applies to x (columns), mode_y (default: same) to y (rows)"""
    bx = border_id(mode)
    by = border_id(mode_y) if mode_y is not None else bx
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    if k.ndim != 2:
        raise ValueError('kernel must be 2-D')
    kp = k.ctypes.data_as(C.POINTER(C.c_double))
    img = _float_img(img)
    ctx = _ctx_of(img, mask, ctx=ctx)
    n, h, w = as_frames(img)
    if _is_dev(img):
        if mask is not None:
            if not _is_dev(mask):
                raise TypeError('device image needs a device mask')
            if mask.dtype != np.uint8 or tuple(mask.shape) != (h, w):
                raise ValueError('device mask must be uint8 of shape %s (got %s %s)'
                                 % ((h, w), mask.dtype, mask.shape))
        dst = _dev_out(ctx, out, img.shape, img.dtype)
        ctx._check(ctx._lib.ipa_conv2d_dev(ctx.handle, img.ptr, dtype_id(img.dtype), h, w, w, kp,
                                           k.shape[0], k.shape[1],
                                           mask.ptr if mask is not None else None, w, dst.ptr, w,
                                           n, h * w, h * w, bx, by, float(cval)), 'conv2d')
        return dst
    m = None
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != (h, w):
            raise ValueError('mask shape %s != image shape %s' % (m.shape, (h, w)))
    dst = np.empty_like(img)
    ctx._check(ctx._lib.ipa_conv2d(ctx.handle, _p(img), dtype_id(img.dtype), h, w, kp, k.shape[0],
                                   k.shape[1], _p(m) if m is not None else None, _p(dst), n, bx,
                                   by, float(cval)), 'conv2d')
    return dst


def sepconv2d(img, ky, kx, mode='reflect', cval=0.0, out=None, ctx=None):
    """separable correlation, axis 0 (ky) then axis 1 (kx); None skips an axis"""
    b = border_id(mode)
    ky = np.zeros(0) if ky is None else np.ascontiguousarray(ky, dtype=np.float64).ravel()
    kx = np.zeros(0) if kx is None else np.ascontiguousarray(kx, dtype=np.float64).ravel()
    dp = C.POINTER(C.c_double)
    img = _float_img(img)
    ctx = _ctx_of(img, ctx=ctx)
    n, h, w = as_frames(img)
    if _is_dev(img):
        dst = _dev_out(ctx, out, img.shape, img.dtype)
        ctx._check(ctx._lib.ipa_sepconv2d_dev(ctx.handle, img.ptr, dtype_id(img.dtype), h, w, w,
                                              ky.ctypes.data_as(dp), ky.size, kx.ctypes.data_as(dp),
                                              kx.size, dst.ptr, w, n, h * w, h * w, b, b,
                                              float(cval)), 'sepconv2d')
        return dst
    dst = np.empty_like(img)
    ctx._check(ctx._lib.ipa_sepconv2d(ctx.handle, _p(img), dtype_id(img.dtype), h, w,
                                      ky.ctypes.data_as(dp), ky.size, kx.ctypes.data_as(dp),
                                      kx.size, _p(dst), n, b, b, float(cval)), 'sepconv2d')
    return dst


def gaussian_kernel1d(sigma, radius=None, truncate=4.0):
    """the taps scipy.ndimage.gaussian_filter uses: radius=int(truncate*sigma+0.5)"""
    sigma = float(sigma)
    if radius is None:
        radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    k = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return k / k.sum()


def gaussian_filter(img, sigma, mode='reflect', cval=0.0, truncate=4.0, out=None, ctx=None):
    """scipy.ndimage.gaussian_filter(img, sigma, mode=..., truncate=...) for 2-D frames"""
    if np.isscalar(sigma):
        sigma = (sigma, sigma)
    ks = [gaussian_kernel1d(s, truncate=truncate) if s > 1e-15 else None for s in sigma]
    return sepconv2d(img, ks[0], ks[1], mode, cval, out=out, ctx=ctx)


def extend_array(arr, kernelXY, modex='reflect', modey='reflect', ctx=None):
    """filters/_extendArrayForConvolution.py:5-97 on the device"""
    kx, ky = int(kernelXY[0]), int(kernelXY[1])
    dev = _is_dev(arr)
    ctx = _ctx_of(arr, ctx=ctx)
    d_in = arr if dev else ctx.to_device(_host(arr))
    if d_in.ndim != 2:
        raise ValueError('extend_array works on 2-D arrays')
    h, w = d_in.shape
    oshape = (h + 2 * (ky // 2), w + 2 * (kx // 2))
    d_out = DeviceArray(ctx, oshape, d_in.dtype)
    ctx._check(ctx._lib.ipa_extend_array_dev(ctx.handle, d_in.ptr, dtype_id(d_in.dtype), h, w, w,
                                             kx, ky, border_id(modex), border_id(modey), d_out.ptr,
                                             oshape[1]), 'extend_array')
    return d_out if dev else d_out.get()


def conv_ydep(arr, kernels, modex='wrap', modey='reflect', ctx=None):
    """row-dependent k0 x k1 correlation, NaN-skipping
    (filters/varYSizeGaussianFilter.py:53-68); kernels: (H, k0, k1) float64"""
    dev = _is_dev(arr)
    ctx = _ctx_of(arr, ctx=ctx)
    d_in = arr if dev else ctx.to_device(_float_img(arr))
    if d_in.ndim != 2:
        raise ValueError('conv_ydep works on 2-D arrays')
    h, w = d_in.shape
    k = np.ascontiguousarray(kernels, dtype=np.float64)
    if k.ndim != 3 or k.shape[0] != h:
        raise ValueError('kernels must be (H, k0, k1)')
    d_k = ctx.to_device(k)
    d_out = DeviceArray(ctx, (h, w), d_in.dtype)
    ctx._check(ctx._lib.ipa_conv_ydep_dev(ctx.handle, d_in.ptr, dtype_id(d_in.dtype), h, w, w,
                                          d_k.ptr, k.shape[1], k.shape[2], border_id(modex),
                                          border_id(modey), d_out.ptr, w), 'conv_ydep')
    if dev:
        d_out._keepalive = d_k  # the table must outlive the asynchronous launch
        return d_out
    return d_out.get()


def var_y_gauss(arr, sig_min, sig_max, ky, rowk, modex='wrap', modey='reflect', ctx=None):
    """filters/varYSizeGaussianFilter.py:9-68 in one launch pair: per-row Gaussian tables built
    on the device (stdys = linspace(sig_min, sig_max, H)), `rowk` = the kx x-responses"""
    dev = _is_dev(arr)
    ctx = _ctx_of(arr, ctx=ctx)
    d_in = arr if dev else ctx.to_device(_float_img(arr))
    if d_in.ndim != 2:
        raise ValueError('var_y_gauss works on 2-D arrays')
    h, w = d_in.shape
    rk = np.ascontiguousarray(rowk, dtype=np.float64).ravel()
    d_out = DeviceArray(ctx, (h, w), d_in.dtype)
    ctx._check(ctx._lib.ipa_var_y_gauss_dev(
        ctx.handle, d_in.ptr, dtype_id(d_in.dtype), h, w, w, float(sig_min), float(sig_max),
        int(ky), rk.ctypes.data_as(C.POINTER(C.c_double)), rk.size, border_id(modex),
        border_id(modey), d_out.ptr, w), 'var_y_gauss')
    return d_out if dev else d_out.get()


def local_std(img, blurred, ksize, ctx=None):
    """filters/standardDeviation.py:34-70 (_calc) for ksize=(kx, ky)"""
    dev = _is_dev(img)
    ctx = _ctx_of(img, blurred, ctx=ctx)
    d_img = img if dev else ctx.to_device(_float_img(img))
    d_bl = blurred if _is_dev(blurred) else ctx.to_device(
        np.ascontiguousarray(blurred, dtype=d_img.dtype))
    if d_img.ndim != 2 or d_bl.shape != d_img.shape or d_bl.dtype != d_img.dtype:
        raise ValueError('img and blurred must be 2-D arrays of equal shape and dtype')
    h, w = d_img.shape
    d_out = DeviceArray(ctx, (h, w), d_img.dtype)
    ctx._check(ctx._lib.ipa_local_std_dev(ctx.handle, d_img.ptr, d_bl.ptr, dtype_id(d_img.dtype),
                                          h, w, w, w, int(ksize[0]), int(ksize[1]), d_out.ptr, w),
               'local_std')
    return d_out if dev else d_out.get()


def masked_mean(arr, mask, ksize, fill_mask=True, ctx=None, fn='median'):
    """filters/maskedFilter.py:12-102 (_calcMean / _calcMedian by ``fn``; default and every value
    other than 'mean': median, as in the reference).  fill_mask=True fills ``arr`` IN PLACE (host
    arrays are copied back into ``arr``); fill_mask=False returns a new NaN-padded array."""
    fn = 'mean' if fn == 'mean' else 'median'
    dev = _is_dev(arr)
    ctx = _ctx_of(arr, mask, ctx=ctx)
    if dev:
        if not _is_dev(mask):
            raise TypeError('device array needs a device mask')
        d_arr, d_mask = arr, mask
        if d_arr.dtype not in (np.float32, np.float64):
            raise TypeError('masked_mean needs float32/float64 arrays')
    else:
        if not (isinstance(arr, np.ndarray) and arr.dtype in (np.float32, np.float64)):
            raise TypeError('masked_mean needs a float32/float64 ndarray')
        d_arr = ctx.to_device(np.ascontiguousarray(arr))
        d_mask = ctx.to_device(np.ascontiguousarray(mask, dtype=np.uint8))
    if d_arr.ndim != 2 or tuple(d_mask.shape) != tuple(d_arr.shape) or d_mask.dtype != np.uint8:
        raise ValueError('arr and mask must be 2-D arrays of equal shape (mask uint8/bool)')
    h, w = d_arr.shape
    d_out = d_arr if fill_mask else DeviceArray(ctx, (h, w), d_arr.dtype)
    f = ctx._lib.ipa_masked_mean_dev if fn == 'mean' else ctx._lib.ipa_masked_median_dev
    ctx._check(f(ctx.handle, d_arr.ptr, dtype_id(d_arr.dtype), d_mask.ptr, h, w, w, w, int(ksize),
                 int(bool(fill_mask)), d_out.ptr, w), 'masked_' + fn)
    if dev:
        return d_out
    if fill_mask:
        arr[...] = d_out.get()
        return arr
    return d_out.get()


def nan_max(arr, ksize, ctx=None):
    """filters/nan_maximum_filter.py:17-37"""
    dev = _is_dev(arr)
    ctx = _ctx_of(arr, ctx=ctx)
    d_arr = arr if dev else ctx.to_device(_float_img(arr))
    if d_arr.ndim != 2 or d_arr.dtype not in (np.float32, np.float64):
        raise TypeError('nan_max needs a 2-D float32/float64 array')
    h, w = d_arr.shape
    d_out = DeviceArray(ctx, (h, w), d_arr.dtype)
    ctx._check(ctx._lib.ipa_nan_max_dev(ctx.handle, d_arr.ptr, dtype_id(d_arr.dtype), h, w, w,
                                        int(ksize), d_out.ptr, w), 'nan_max')
    return d_out if dev else d_out.get()


def closest_distance(arr, ksize=30, dtype=np.uint16, ctx=None):
    """render/closestDirectDistance.py:17-41: distance to the closest non-zero pixel"""
    dtype = np.dtype(dtype)
    if dtype not in (np.uint16, np.float64):
        raise NotImplementedError('closest_distance writes uint16 or float64')
    dev = _is_dev(arr)
    ctx = _ctx_of(arr, ctx=ctx)
    if dev:
        if arr.dtype != np.uint8:
            raise TypeError('device input must be uint8 (non-zero = set)')
        d_arr = arr
    else:
        d_arr = ctx.to_device(np.ascontiguousarray(np.asarray(arr) != 0, dtype=np.uint8))
    if d_arr.ndim != 2:
        raise ValueError('closest_distance works on 2-D arrays')
    h, w = d_arr.shape
    d_out = DeviceArray(ctx, (h, w), dtype)
    ctx._check(ctx._lib.ipa_closest_distance_dev(ctx.handle, d_arr.ptr, h, w, w, int(ksize),
                                                 d_out.ptr, dtype_id(dtype), w), 'closest_distance')
    return d_out if dev else d_out.get()


def pos_intensity_unc(image, sx, sy, ksize, ctx=None):
    """uncertainty/positionToIntensityUncertainty.py:7-49; ksize = half window.  sx / sy are
    both scalars or both (H, W) maps; returns float64"""
    maps = isinstance(sx, (np.ndarray, DeviceArray))
    dev = _is_dev(image)
    ctx = _ctx_of(image, sx if _is_dev(sx) else None, sy if _is_dev(sy) else None, ctx=ctx)
    d_img = image if dev else ctx.to_device(_float_img(image))
    if d_img.ndim != 2 or d_img.dtype not in (np.float32, np.float64):
        raise TypeError('pos_intensity_unc needs a 2-D float32/float64 image')
    h, w = d_img.shape
    d_sx = d_sy = None
    if maps:
        d_sx = sx if _is_dev(sx) else ctx.to_device(np.ascontiguousarray(sx, dtype=np.float64))
        d_sy = sy if _is_dev(sy) else ctx.to_device(np.ascontiguousarray(sy, dtype=np.float64))
        if tuple(d_sx.shape) != (h, w) or tuple(d_sy.shape) != (h, w) or \
                d_sx.dtype != np.float64 or d_sy.dtype != np.float64:
            raise ValueError('sigma maps must be float64 arrays of the image shape')
    d_out = DeviceArray(ctx, (h, w), np.float64)
    ctx._check(ctx._lib.ipa_pos_intensity_unc_dev(
        ctx.handle, d_img.ptr, dtype_id(d_img.dtype), h, w, w,
        d_sx.ptr if maps else None, d_sy.ptr if maps else None, w,
        0.0 if maps else float(sx), 0.0 if maps else float(sy), int(ksize), d_out.ptr, w),
        'pos_intensity_unc')
    return d_out if dev else d_out.get()


def median_threshold(img, threshold=0.1, condition='>', want_indices=True, ctx=None, size=3):
    """filters/medianThreshold.py:7-30: -> (out, indices) new arrays"""
    if condition not in ('>', '<'):
        raise ValueError("condition must be '>' or '<'")
    dev = _is_dev(img)
    ctx = _ctx_of(img, ctx=ctx)
    d_img = img if dev else ctx.to_device(_float_img(img))
    if d_img.ndim != 2 or d_img.dtype not in (np.float32, np.float64):
        raise TypeError('median_threshold needs a 2-D float32/float64 array')
    h, w = d_img.shape
    d_out = DeviceArray(ctx, (h, w), d_img.dtype)
    d_idx = DeviceArray(ctx, (h, w), np.uint8) if want_indices else None
    size = int(size)
    if size < 1:
        raise ValueError('size must be >= 1')
    ctx._check(ctx._lib.ipa_median_threshold_size_dev(
        ctx.handle, d_img.ptr, dtype_id(d_img.dtype), h, w, w, size, float(threshold),
        int(condition == '<'), d_out.ptr, w, d_idx.ptr if want_indices else None, w),
        'median_threshold')
    if dev:
        return d_out, d_idx
    return d_out.get(), (d_idx.get().astype(bool) if want_indices else None)


def calib_prefilter(img, bg=None, ff=None, threshold=0.1, out=None, ctx=None):
    """stages 2-4 of CameraCalibration.correct (camera/CameraCalibration.py:416-437) in one
    kernel: dark current, flat field, nan_to_num + thresholded 3x3 median.  bg / ff are cast to
    the image dtype; returns a new array (DeviceArray for device input)."""
    dev = _is_dev(img)
    ctx = _ctx_of(img, bg, ff, ctx=ctx)
    d_img = img if dev else ctx.to_device(_float_img(img))
    if d_img.ndim != 2 or d_img.dtype not in (np.float32, np.float64):
        raise TypeError('calib_prefilter needs a 2-D float32/float64 array')
    h, w = d_img.shape

    def side(a, name):
        if a is None:
            return None
        if _is_dev(a):
            if a.dtype != d_img.dtype or tuple(a.shape) != (h, w):
                raise ValueError('%s must match the image shape and dtype' % name)
            return a
        a = np.asarray(a)
        if a.shape != (h, w):
            a = np.broadcast_to(a, (h, w))
        return ctx.to_device(np.ascontiguousarray(a, dtype=d_img.dtype))

    d_bg, d_ff = side(bg, 'bg'), side(ff, 'ff')
    d_out = _dev_out(ctx, out, (h, w), d_img.dtype) if dev else DeviceArray(ctx, (h, w), d_img.dtype)
    ctx._check(ctx._lib.ipa_calib_prefilter_dev(
        ctx.handle, d_img.ptr, dtype_id(d_img.dtype), d_bg.ptr if d_bg is not None else None,
        d_ff.ptr if d_ff is not None else None, h, w, w, w, w, float(threshold), d_out.ptr, w),
        'calib_prefilter')
    return d_out if dev else d_out.get()


# ------------------------------------------------- fused remap -> filter --
def _fused_out(ctx, src, out, dh, dw, n):
    odt = np.float64 if src.dtype == np.float64 else np.float32
    return _dev_out(ctx, out, (dh, dw) if src.ndim == 2 else (n, dh, dw), odt), odt


def remap_conv2d(src, mapx, mapy, kernel, interpolation='linear', border_mode='constant',
                 border_value=0.0, conv_mode='reflect', out=None, ctx=None):
    """remap followed by a K x K centred correlation, one kernel, device arrays only"""
    ctx = _ctx_of(src, mapx, mapy, ctx=ctx)
    if not (_is_dev(src) and _is_dev(mapx) and _is_dev(mapy)):
        raise TypeError('remap_conv2d works on DeviceArrays (use Context.to_device)')
    _check_dev_maps(mapx, mapy)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    n, sh, sw = as_frames(src)
    dh, dw = mapx.shape
    dst, odt = _fused_out(ctx, src, out, dh, dw, n)
    cb = border_id(conv_mode)
    ctx._check(ctx._lib.ipa_remap_conv2d_dev(
        ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw, mapx.ptr, mapy.ptr, dw,
        k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], k.shape[1], dst.ptr, dtype_id(odt),
        dh, dw, dw, n, sh * sw, dh * dw, interp_id(interpolation), border_id(border_mode),
        float(border_value), cb, cb), 'remap_conv2d')
    return dst


def undistort_conv2d(src, K, dist5, newK, kernel, interpolation='linear', border_mode='constant',
                     border_value=0.0, conv_mode='reflect', out_shape=None, out=None, ctx=None):
    ctx = _ctx_of(src, ctx=ctx)
    if not _is_dev(src):
        raise TypeError('undistort_conv2d works on DeviceArrays (use Context.to_device)')
    if newK is None:
        newK = K
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    n, sh, sw = as_frames(src)
    dh, dw = out_shape or (sh, sw)
    dst, odt = _fused_out(ctx, src, out, dh, dw, n)
    cb = border_id(conv_mode)
    ctx._check(ctx._lib.ipa_undistort_conv2d_dev(
        ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw, L.dbl(np.ravel(K), 9),
        L.dbl(np.ravel(dist5)[:5], 5), L.dbl(np.ravel(newK), 9),
        k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], k.shape[1], dst.ptr, dtype_id(odt),
        dh, dw, dw, n, sh * sw, dh * dw, interp_id(interpolation), border_id(border_mode),
        float(border_value), cb, cb), 'undistort_conv2d')
    return dst


def warp_perspective_conv2d(src, M_dst2src, out_shape, kernel, interpolation='linear',
                            border_mode='constant', border_value=0.0, conv_mode='reflect',
                            out=None, ctx=None):
    ctx = _ctx_of(src, ctx=ctx)
    if not _is_dev(src):
        raise TypeError('warp_perspective_conv2d works on DeviceArrays (use Context.to_device)')
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    n, sh, sw = as_frames(src)
    dh, dw = int(out_shape[0]), int(out_shape[1])
    dst, odt = _fused_out(ctx, src, out, dh, dw, n)
    cb = border_id(conv_mode)
    ctx._check(ctx._lib.ipa_warp_perspective_conv2d_dev(
        ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw,
        L.dbl(np.ravel(np.asarray(M_dst2src, dtype=np.float64)), 9),
        k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], k.shape[1], dst.ptr, dtype_id(odt),
        dh, dw, dw, n, sh * sw, dh * dw, interp_id(interpolation), border_id(border_mode),
        float(border_value), cb, cb), 'warp_perspective_conv2d')
    return dst


def _sep_taps(ky, kx):
    ky = np.ascontiguousarray(ky, dtype=np.float64).ravel()
    kx = np.ascontiguousarray(kx, dtype=np.float64).ravel()
    dp = C.POINTER(C.c_double)
    return ky, kx, ky.ctypes.data_as(dp), kx.ctypes.data_as(dp)


def remap_sepconv2d(src, mapx, mapy, ky, kx, interpolation='linear', border_mode='constant',
                    border_value=0.0, conv_mode='reflect', out=None, ctx=None):
    """remap followed by a separable correlation (axis 0 with ky, then axis 1 with kx; the
    scipy.ndimage.gaussian_filter order) — one kernel where built, device arrays only"""
    ctx = _ctx_of(src, mapx, mapy, ctx=ctx)
    if not (_is_dev(src) and _is_dev(mapx) and _is_dev(mapy)):
        raise TypeError('remap_sepconv2d works on DeviceArrays (use Context.to_device)')
    _check_dev_maps(mapx, mapy)
    ky, kx, pky, pkx = _sep_taps(ky, kx)
    n, sh, sw = as_frames(src)
    dh, dw = mapx.shape
    dst = _dev_out(ctx, out, (dh, dw) if src.ndim == 2 else (n, dh, dw), np.float32)
    cb = border_id(conv_mode)
    ctx._check(ctx._lib.ipa_remap_sepconv2d_dev(
        ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw, mapx.ptr, mapy.ptr, dw, pky, ky.size,
        pkx, kx.size, dst.ptr, dtype_id(np.float32), dh, dw, dw, n, sh * sw, dh * dw,
        interp_id(interpolation), border_id(border_mode), float(border_value), cb, cb),
        'remap_sepconv2d')
    return dst


def undistort_sepconv2d(src, K, dist5, newK, ky, kx, interpolation='linear',
                        border_mode='constant', border_value=0.0, conv_mode='reflect',
                        out_shape=None, out=None, ctx=None):
    ctx = _ctx_of(src, ctx=ctx)
    if not _is_dev(src):
        raise TypeError('undistort_sepconv2d works on DeviceArrays (use Context.to_device)')
    if newK is None:
        newK = K
    ky, kx, pky, pkx = _sep_taps(ky, kx)
    n, sh, sw = as_frames(src)
    dh, dw = out_shape or (sh, sw)
    dst = _dev_out(ctx, out, (dh, dw) if src.ndim == 2 else (n, dh, dw), np.float32)
    cb = border_id(conv_mode)
    ctx._check(ctx._lib.ipa_undistort_sepconv2d_dev(
        ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw, L.dbl(np.ravel(K), 9),
        L.dbl(np.ravel(dist5)[:5], 5), L.dbl(np.ravel(newK), 9), pky, ky.size, pkx, kx.size,
        dst.ptr, dtype_id(np.float32), dh, dw, dw, n, sh * sw, dh * dw, interp_id(interpolation),
        border_id(border_mode), float(border_value), cb, cb), 'undistort_sepconv2d')
    return dst


def warp_perspective_sepconv2d(src, M_dst2src, out_shape, ky, kx, interpolation='linear',
                               border_mode='constant', border_value=0.0, conv_mode='reflect',
                               out=None, ctx=None):
    ctx = _ctx_of(src, ctx=ctx)
    if not _is_dev(src):
        raise TypeError('warp_perspective_sepconv2d works on DeviceArrays (use Context.to_device)')
    ky, kx, pky, pkx = _sep_taps(ky, kx)
    n, sh, sw = as_frames(src)
    dh, dw = int(out_shape[0]), int(out_shape[1])
    dst = _dev_out(ctx, out, (dh, dw) if src.ndim == 2 else (n, dh, dw), np.float32)
    cb = border_id(conv_mode)
    ctx._check(ctx._lib.ipa_warp_perspective_sepconv2d_dev(
        ctx.handle, src.ptr, dtype_id(src.dtype), sh, sw, sw,
        L.dbl(np.ravel(np.asarray(M_dst2src, dtype=np.float64)), 9), pky, ky.size, pkx, kx.size,
        dst.ptr, dtype_id(np.float32), dh, dw, dw, n, sh * sw, dh * dw, interp_id(interpolation),
        border_id(border_mode), float(border_value), cb, cb), 'warp_perspective_sepconv2d')
    return dst


# ------------------------------------------------------------------- IDW --
def idw_fill(grid, mask, ksize, weights, ctx=None):
    """in-place IDW hole filling (interpolate2dStructuredIDW._calc)"""
    w = np.ascontiguousarray(weights, dtype=np.float64)
    if w.shape != (2 * ksize + 1, 2 * ksize + 1):
        raise ValueError('weights must be (2*ksize+1)^2')
    wp = w.ctypes.data_as(C.POINTER(C.c_double))
    ctx = _ctx_of(grid, mask, ctx=ctx)
    if _is_dev(grid):
        if not _is_dev(mask):
            raise TypeError('device grid needs a device mask')
        h, wd = grid.shape
        ctx._check(ctx._lib.ipa_idw_fill_dev(ctx.handle, grid.ptr, dtype_id(grid.dtype), mask.ptr,
                                             h, wd, wd, int(ksize), wp), 'idw_fill')
        return grid
    if not (isinstance(grid, np.ndarray) and grid.flags.c_contiguous and grid.ndim == 2):
        raise ValueError('grid must be a C-contiguous 2-D ndarray (it is modified in place)')
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    h, wd = grid.shape
    ctx._check(ctx._lib.ipa_idw_fill(ctx.handle, _p(grid), dtype_id(grid.dtype), _p(m), h, wd,
                                     int(ksize), wp), 'idw_fill')
    return grid


def fast_idw_fill(grid, mask, offsets, weights, minnvals, ctx=None):
    """in-place growing-distance IDW (interpolate2dStructuredFastIDW._calc)"""
    offs = np.ascontiguousarray(offsets, dtype=np.int32)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    n = int(w.size)
    if offs.shape != (n, 2):
        raise ValueError('offsets must be (n,2) matching weights')
    wp = w.ctypes.data_as(C.POINTER(C.c_double))
    ctx = _ctx_of(grid, mask, ctx=ctx)
    if _is_dev(grid):
        if not _is_dev(mask):
            raise TypeError('device grid needs a device mask')
        h, wd = grid.shape
        ctx._check(ctx._lib.ipa_fast_idw_fill_dev(ctx.handle, grid.ptr, dtype_id(grid.dtype),
                                                  mask.ptr, h, wd, wd, _p(offs), wp, n,
                                                  int(minnvals)), 'fast_idw_fill')
        return grid
    if not (isinstance(grid, np.ndarray) and grid.flags.c_contiguous and grid.ndim == 2):
        raise ValueError('grid must be a C-contiguous 2-D ndarray (it is modified in place)')
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    h, wd = grid.shape
    ctx._check(ctx._lib.ipa_fast_idw_fill(ctx.handle, _p(grid), dtype_id(grid.dtype), _p(m), h,
                                          wd, _p(offs), wp, n, int(minnvals)), 'fast_idw_fill')
    return grid


def _fill_grid(grid):
    if not (isinstance(grid, np.ndarray) and grid.flags.c_contiguous and grid.ndim == 2):
        raise ValueError('grid must be a C-contiguous 2-D ndarray (it is modified in place)')
    if grid.dtype not in (np.float32, np.float64):
        raise TypeError('grid must be float32 or float64')
    return grid.shape


def unstructured_idw(x, y, v, grid, power=2, ctx=None):
    """every pixel of `grid` from n scattered points (interpolate2dUnstructuredIDW), in place"""
    xs, ys, vs = (np.ascontiguousarray(np.ravel(a), dtype=np.float64) for a in (x, y, v))
    if not (xs.size == ys.size == vs.size and vs.size >= 1):
        raise ValueError('x, y, v must be 1-D, of equal length >= 1')
    dp = C.POINTER(C.c_double)
    px, py, pv = (a.ctypes.data_as(dp) for a in (xs, ys, vs))
    ctx = _ctx_of(grid, ctx=ctx)
    if _is_dev(grid):
        h, w = grid.shape
        ctx._check(ctx._lib.ipa_unstructured_idw_dev(ctx.handle, grid.ptr, dtype_id(grid.dtype), h,
                                                     w, w, px, py, pv, int(vs.size), float(power)),
                   'unstructured_idw')
        return grid
    h, w = _fill_grid(grid)
    ctx._check(ctx._lib.ipa_unstructured_idw(ctx.handle, _p(grid), dtype_id(grid.dtype), h, w, px,
                                             py, pv, int(vs.size), float(power)),
               'unstructured_idw')
    return grid


def circular_idw_fill(grid, mask, ksize, power=2, fr=1, fphi=1, cx=0, cy=0, ctx=None):
    """in-place IDW hole filling with polar distances (interpolateCircular2dStructuredIDW)"""
    ctx = _ctx_of(grid, mask, ctx=ctx)
    args = (int(ksize), float(power), float(fr), float(fphi), float(cx), float(cy))
    if _is_dev(grid):
        if not _is_dev(mask):
            raise TypeError('device grid needs a device mask')
        h, w = grid.shape
        ctx._check(ctx._lib.ipa_circular_idw_fill_dev(ctx.handle, grid.ptr, dtype_id(grid.dtype),
                                                      mask.ptr, h, w, w, *args),
                   'circular_idw_fill')
        return grid
    h, w = _fill_grid(grid)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if m.shape != grid.shape:
        raise ValueError('mask and grid differ in shape')
    ctx._check(ctx._lib.ipa_circular_idw_fill(ctx.handle, _p(grid), dtype_id(grid.dtype), _p(m), h,
                                              w, *args), 'circular_idw_fill')
    return grid


def cross_avg_fill(grid, mask, ksize, power=2, ctx=None):
    """in-place fill of large holes from the four axis directions
    (interpolate2dStructuredCrossAvg)"""
    ctx = _ctx_of(grid, mask, ctx=ctx)
    if _is_dev(grid):
        if not _is_dev(mask):
            raise TypeError('device grid needs a device mask')
        h, w = grid.shape
        ctx._check(ctx._lib.ipa_cross_avg_fill_dev(ctx.handle, grid.ptr, dtype_id(grid.dtype),
                                                   mask.ptr, h, w, w, int(ksize), float(power)),
                   'cross_avg_fill')
        return grid
    h, w = _fill_grid(grid)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if m.shape != grid.shape:
        raise ValueError('mask and grid differ in shape')
    ctx._check(ctx._lib.ipa_cross_avg_fill(ctx.handle, _p(grid), dtype_id(grid.dtype), _p(m), h, w,
                                           int(ksize), float(power)), 'cross_avg_fill')
    return grid


def point_spread_idw(grid, mask, ksize, power=2, max_iter=1e5, ctx=None):
    """in-place point-spread IDW fill (interpolate2dStructuredPointSpreadIDW): grid AND mask are
    modified - filled pixels are unmasked"""
    ctx = _ctx_of(grid, mask, ctx=ctx)
    it = int(min(max_iter, 2 ** 62))
    if _is_dev(grid):
        if not _is_dev(mask):
            raise TypeError('device grid needs a device mask')
        h, w = grid.shape
        ctx._check(ctx._lib.ipa_point_spread_idw_dev(ctx.handle, grid.ptr, dtype_id(grid.dtype),
                                                     mask.ptr, h, w, w, int(ksize), float(power), it),
                   'point_spread_idw')
        return grid
    h, w = _fill_grid(grid)
    if not (isinstance(mask, np.ndarray) and mask.flags.c_contiguous and mask.shape == grid.shape and
            mask.dtype in (np.bool_, np.uint8)):
        raise ValueError('mask must be a C-contiguous bool / uint8 array of the grid\'s shape '
                         '(it is modified in place)')
    ctx._check(ctx._lib.ipa_point_spread_idw(ctx.handle, _p(grid), dtype_id(grid.dtype),
                                             _p(mask.view(np.uint8)), h, w, int(ksize), float(power),
                                             it), 'point_spread_idw')
    return grid


# ------------------------------------------------------- fastFilter / fastMean --
RESIZE_INTERP = {'linear': 1, 'cubic': 2, 'area': 3, 'lanczos4': 4, 1: 1, 2: 2, 3: 3, 4: 4}
FAST_FILTER_FN = {'median': 0, 'nanmedian': 1, 'mean': 2, 'nanmean': 3}


def resize(img, dsize_hw, interpolation='linear', out=None, ctx=None, src_shape=None):
    """cv2.resize(img, (w, h), interpolation=...) for 2-D float32 / float64 images;
    dsize_hw = (rows, columns) of the result.  ``src_shape`` (device arrays): resize only the
    top-left (rows, columns) of ``img`` - fastFilter's cropped grid, without a copy"""
    if interpolation not in RESIZE_INTERP:
        raise ValueError('resize: interpolation %r is not built (linear / 1, cubic / 2, area / 3, '
                         'lanczos4 / 4; cv2.INTER_NEAREST = 0 and the exact variants are not)'
                         % (interpolation,))
    interp = RESIZE_INTERP[interpolation]
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    if _is_dev(img):
        ctx = _ctx_of(img, ctx=ctx)
        if img.ndim != 2:
            raise ValueError('resize takes one 2-D image')
        sh, sw = img.shape
        pitch = sw
        if src_shape is not None:
            if not (0 < int(src_shape[0]) <= sh and 0 < int(src_shape[1]) <= sw):
                raise ValueError('src_shape %r outside the %r array' % (tuple(src_shape), img.shape))
            sh, sw = int(src_shape[0]), int(src_shape[1])
        dst = _dev_out(ctx, out, (dh, dw), img.dtype)
        ctx._check(ctx._lib.ipa_resize_dev(ctx.handle, img.ptr, dtype_id(img.dtype), sh, sw, pitch,
                                           dst.ptr, dh, dw, dw, interp), 'resize')
        return dst
    if src_shape is not None:
        img = np.asarray(img)[:int(src_shape[0]), :int(src_shape[1])]
    ctx = ctx or default_context()
    img = _float_img(img)
    if img.ndim != 2:
        raise ValueError('resize takes one 2-D image')
    sh, sw = img.shape
    dst = np.empty((dh, dw), img.dtype)
    ctx._check(ctx._lib.ipa_resize(ctx.handle, _p(img), dtype_id(img.dtype), sh, sw, _p(dst), dh,
                                   dw, interp), 'resize')
    return dst


def fast_filter_stat(arr, ksize, every, fn='median', ctx=None):
    """the strided window statistics of fastFilter (filters/fastFilter.py:52-122): float64 array of
    ceil(h / every) x ceil(w / every) cells"""
    f = FAST_FILTER_FN[fn]
    every, ksize = int(every), int(ksize)
    if _is_dev(arr):
        ctx = _ctx_of(arr, ctx=ctx)
        h, w = arr.shape
        out = ctx.empty((-(-h // every), -(-w // every)), np.float64)
        ctx._check(ctx._lib.ipa_fast_filter_stat_dev(ctx.handle, arr.ptr, dtype_id(arr.dtype), h, w,
                                                     w, ksize, every, f, out.ptr),
                   'fast_filter_stat')
        return out
    ctx = ctx or default_context()
    arr = _float_img(arr)
    if arr.ndim != 2:
        raise ValueError('fast_filter_stat takes one 2-D array')
    h, w = arr.shape
    out = np.empty((-(-h // every), -(-w // every)), np.float64)
    ctx._check(ctx._lib.ipa_fast_filter_stat(ctx.handle, _p(arr), dtype_id(arr.dtype), h, w, ksize,
                                             every, f, _p(out)), 'fast_filter_stat')
    return out
