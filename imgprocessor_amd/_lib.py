"""ctypes binding of libimgproc_hip.so (C ABI: include/imgproc_hip.h).

The library is the product: if it is missing or cannot be loaded this module
raises — there is no CPU or numpy fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMGPROC_HIP_LIB selects another build of the SAME library (tuning A/B variants)
LIB_PATH = os.environ.get('IMGPROC_HIP_LIB') or os.path.join(_HERE, 'libimgproc_hip.so')

# enums of include/imgproc_hip.h
U8, U16, F32, F64 = 0, 1, 2, 3
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC_CV, INTER_LANCZOS4, INTER_CUBIC_KEYS = 0, 1, 2, 4, 5
INTER_Q5 = 0x100
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT101 = 0, 1, 2, 3, 4

OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_OOM, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5

_vp, _i, _l, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_size_t
_dp = C.POINTER(C.c_double)

# name -> argtypes; every function returns int (ipa_status) unless noted
PROTOTYPES = {
    'ipa_version': [],
    'ipa_device_count': [C.POINTER(_i)],
    'ipa_device_pci_bus_id': [_i, C.c_char_p, C.c_size_t],
    'ipa_ctx_create': [_i, C.POINTER(_vp)],
    'ipa_ctx_destroy': [_vp],
    'ipa_ctx_synchronize': [_vp],
    'ipa_ctx_set_tuning': [_vp, C.c_char_p, _i],
    'ipa_ctx_get_tuning': [_vp, C.c_char_p, C.POINTER(_i)],
    'ipa_ctx_device_info': [_vp, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(_sz)],
    'ipa_mem_info': [_vp, C.POINTER(_sz), C.POINTER(_sz)],
    'ipa_malloc': [_vp, _sz, C.POINTER(_vp)],
    'ipa_free': [_vp, _vp],
    'ipa_host_alloc': [_vp, _sz, C.POINTER(_vp)],
    'ipa_host_free': [_vp, _vp],
    'ipa_memcpy_h2d': [_vp, _vp, _vp, _sz],
    'ipa_memcpy_d2h': [_vp, _vp, _vp, _sz],
    'ipa_memcpy_d2d': [_vp, _vp, _vp, _sz],
    'ipa_memset': [_vp, _vp, _i, _sz],
    'ipa_event_create': [_vp, C.POINTER(_vp)],
    'ipa_event_destroy': [_vp, _vp],
    'ipa_event_record': [_vp, _vp],
    'ipa_event_elapsed_ms': [_vp, _vp, _vp, C.POINTER(C.c_float)],
    'ipa_build_undistort_map_dev': [_vp, _dp, _dp, _dp, _i, _i, _vp, _vp, _l],
    'ipa_build_undistort_map': [_vp, _dp, _dp, _dp, _i, _i, _vp, _vp],
    'ipa_remap_dev': [_vp, _vp, _i, _i, _i, _l, _vp, _vp, _l, _vp, _i, _i, _i, _l, _i, _l, _l,
                      _i, _i, _d],
    'ipa_remap': [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d],
    'ipa_undistort_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _dp, _dp, _vp, _i, _i, _i, _l, _i, _l,
                          _l, _i, _i, _d],
    'ipa_undistort': [_vp, _vp, _i, _i, _i, _dp, _dp, _dp, _vp, _i, _i, _i, _i, _i, _i, _d],
    'ipa_warp_perspective_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _vp, _i, _i, _i, _l, _i, _l, _l,
                                 _i, _i, _d],
    'ipa_warp_perspective': [_vp, _vp, _i, _i, _i, _dp, _vp, _i, _i, _i, _i, _i, _i, _d],
    'ipa_conv2d_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _i, _i, _vp, _l, _vp, _l, _i, _l, _l, _i,
                       _i, _d],
    'ipa_conv2d': [_vp, _vp, _i, _i, _i, _dp, _i, _i, _vp, _vp, _i, _i, _i, _d],
    'ipa_sepconv2d_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _i, _dp, _i, _vp, _l, _i, _l, _l, _i, _i,
                          _d],
    'ipa_sepconv2d': [_vp, _vp, _i, _i, _i, _dp, _i, _dp, _i, _vp, _i, _i, _i, _d],
    'ipa_extend_array_dev': [_vp, _vp, _i, _i, _i, _l, _i, _i, _i, _i, _vp, _l],
    'ipa_deinterleave_dev': [_vp, _vp, _i, _i, _i, _i, _l, _vp, _l, _l],
    'ipa_interleave_dev': [_vp, _vp, _i, _i, _i, _i, _l, _l, _vp, _l],
    'ipa_conv_ydep_dev': [_vp, _vp, _i, _i, _i, _l, _vp, _i, _i, _i, _i, _vp, _l],
    'ipa_var_y_gauss_dev': [_vp, _vp, _i, _i, _i, _l, C.c_double, C.c_double, _i,
                            C.POINTER(C.c_double), _i, _i, _i, _vp, _l],
    'ipa_local_std_dev': [_vp, _vp, _vp, _i, _i, _i, _l, _l, _i, _i, _vp, _l],
    'ipa_masked_mean_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _l, _i, _i, _vp, _l],
    'ipa_masked_median_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _l, _i, _i, _vp, _l],
    'ipa_nan_max_dev': [_vp, _vp, _i, _i, _i, _l, _i, _vp, _l],
    'ipa_closest_distance_dev': [_vp, _vp, _i, _i, _l, _i, _vp, _i, _l],
    'ipa_pos_intensity_unc_dev': [_vp, _vp, _i, _i, _i, _l, _vp, _vp, _l, _d, _d, _i, _vp, _l],
    'ipa_median_threshold_dev': [_vp, _vp, _i, _i, _i, _l, _d, _i, _vp, _l, _vp, _l],
    'ipa_median_threshold_size_dev': [_vp, _vp, _i, _i, _i, _l, _i, _d, _i, _vp, _l, _vp, _l],
    'ipa_calib_prefilter_dev': [_vp, _vp, _i, _vp, _vp, _i, _i, _l, _l, _l, _d, _vp, _l],
    'ipa_remap_conv2d_dev': [_vp, _vp, _i, _i, _i, _l, _vp, _vp, _l, _dp, _i, _i, _vp, _i, _i,
                             _i, _l, _i, _l, _l, _i, _i, _d, _i, _i],
    'ipa_undistort_conv2d_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _dp, _dp, _dp, _i, _i, _vp, _i,
                                 _i, _i, _l, _i, _l, _l, _i, _i, _d, _i, _i],
    'ipa_warp_perspective_conv2d_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _dp, _i, _i, _vp, _i, _i,
                                        _i, _l, _i, _l, _l, _i, _i, _d, _i, _i],
    'ipa_remap_sepconv2d_dev': [_vp, _vp, _i, _i, _i, _l, _vp, _vp, _l, _dp, _i, _dp, _i, _vp, _i,
                                _i, _i, _l, _i, _l, _l, _i, _i, _d, _i, _i],
    'ipa_undistort_sepconv2d_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _dp, _dp, _dp, _i, _dp, _i, _vp,
                                    _i, _i, _i, _l, _i, _l, _l, _i, _i, _d, _i, _i],
    'ipa_warp_perspective_sepconv2d_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _dp, _i, _dp, _i, _vp, _i,
                                           _i, _i, _l, _i, _l, _l, _i, _i, _d, _i, _i],
    'ipa_idw_fill_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _i, _dp],
    'ipa_idw_fill': [_vp, _vp, _i, _vp, _i, _i, _i, _dp],
    'ipa_fast_idw_fill_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _vp, _dp, _i, _i],
    'ipa_fast_idw_fill': [_vp, _vp, _i, _vp, _i, _i, _vp, _dp, _i, _i],
    'ipa_unstructured_idw_dev': [_vp, _vp, _i, _i, _i, _l, _dp, _dp, _dp, _i, _d],
    'ipa_unstructured_idw': [_vp, _vp, _i, _i, _i, _dp, _dp, _dp, _i, _d],
    'ipa_circular_idw_fill_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _i, _d, _d, _d, _d, _d],
    'ipa_circular_idw_fill': [_vp, _vp, _i, _vp, _i, _i, _i, _d, _d, _d, _d, _d],
    'ipa_cross_avg_fill_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _i, _d],
    'ipa_cross_avg_fill': [_vp, _vp, _i, _vp, _i, _i, _i, _d],
    'ipa_point_spread_idw_dev': [_vp, _vp, _i, _vp, _i, _i, _l, _i, _d, _l],
    'ipa_point_spread_idw': [_vp, _vp, _i, _vp, _i, _i, _i, _d, _l],
    'ipa_resize_dev': [_vp, _vp, _i, _i, _i, _l, _vp, _i, _i, _l, _i],
    'ipa_resize': [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i],
    'ipa_fast_filter_stat_dev': [_vp, _vp, _i, _i, _i, _l, _i, _i, _i, _vp],
    'ipa_fast_filter_stat': [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
}
_CHARP = {'ipa_status_string': [_i], 'ipa_last_error': [_vp]}

_lib = None


class ImgProcHipError(RuntimeError):
    """a call into libimgproc_hip.so returned a non-zero status"""

    def __init__(self, status, message):
        RuntimeError.__init__(self, message)
        self.status = status


def lib():
    """load (once) and return the ctypes handle; raises if the extension is absent"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'imgprocessor_amd: %s is missing. Build it with '
                '`make -C imgprocessor_amd/csrc -j8` (or __graft_entry__.build()). '
                'There is no CPU fallback.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, args in PROTOTYPES.items():
            f = getattr(l, name)
            f.argtypes = args
            f.restype = _i
        for name, args in _CHARP.items():
            f = getattr(l, name)
            f.argtypes = args
            f.restype = C.c_char_p
        _lib = l
    return _lib


def check(status, ctx_handle=None, what=''):
    if status == OK:
        return
    l = lib()
    msg = l.ipa_last_error(ctx_handle) if ctx_handle else l.ipa_last_error(None)
    text = '%s: %s (%s)' % (what or 'libimgproc_hip', (msg or b'').decode() or '?',
                            l.ipa_status_string(status).decode())
    if status == ERR_BAD_ARG:
        raise ValueError(text)
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    if status == ERR_OOM:
        raise MemoryError(text)
    raise ImgProcHipError(status, text)


def dbl(values, n=None):
    """host double array for the small matrix / kernel arguments"""
    vals = [float(v) for v in values]
    if n is not None and len(vals) != n:
        raise ValueError('expected %d values, got %d' % (n, len(vals)))
    return (C.c_double * len(vals))(*vals)
