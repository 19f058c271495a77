"""``varYSizeGaussianFilter`` — reference: imgProcessor/filters/varYSizeGaussianFilter.py:9-68.

A Gaussian whose sigma along y changes from row to row.  The per-row k0 x k1
coefficient tables are built on the host the way the reference builds them
(``gaussian_filter`` of a centred delta, :40-46 — a few thousand tiny
filters); the O(H·W·k0·k1) NaN-skipping correlation (:53-68) runs as a HIP
kernel with the borders resolved on the fly (defaults modex='wrap',
modey='reflect', like the reference's padding).

``stdyrange`` as an ndarray raises UnboundLocalError in the reference (``mx``
is never set on that branch); only the int / (mn, mx) forms are defined.
"""
import numpy as np

from .. import ops


def _correlate1d_reflect(a, w, axis):
    r = len(w) // 2
    pad = [(0, 0)] * a.ndim
    pad[axis] = (r, r)
    p = np.pad(a, pad, mode='symmetric')
    out = np.zeros_like(a, dtype=np.float64)
    n = a.shape[axis]
    for i, wi in enumerate(w):
        sl = [slice(None)] * a.ndim
        sl[axis] = slice(i, i + n)
        out += wi * p[tuple(sl)]
    return out


def _gaussian_of_delta(shape, sigmas, truncate=4.0):
    """scipy.ndimage.gaussian_filter(delta, sigmas) for a tiny centred delta image"""
    out = np.zeros(shape)
    out[shape[0] // 2, shape[1] // 2] = 1
    for axis, s in enumerate(sigmas):
        if s > 1e-15:
            out = _correlate1d_reflect(out, ops.gaussian_kernel1d(s, truncate=truncate), axis)
    return out


def varYSizeGaussianFilter(arr, stdyrange, stdx=0, modex='wrap', modey='reflect', ctx=None):
    assert len(arr.shape) == 2, 'only works on 2d arrays at the moment'
    if isinstance(stdyrange, np.ndarray):
        raise UnboundLocalError("local variable 'mx' referenced before assignment "
                                "(the reference's ndarray branch is broken: pass an int or (mn, mx))")
    s0 = arr.shape[0]
    if type(stdyrange) not in (list, tuple):
        stdyrange = (0, stdyrange)
    mn, mx = stdyrange
    stdys = np.linspace(mn, mx, s0)
    kx = int(stdx * 2.5)
    kx += 1 - kx % 2
    ky = int(mx * 2.5)
    ky += 1 - ky % 2
    if modey != 'reflect':
        raise Exception('modey not supported')
    if modex not in ('reflect', 'wrap'):
        raise Exception('modex not supported')
    kernels = np.empty((s0, ky, kx))
    for i in range(s0):
        kernels[i] = _gaussian_of_delta((ky, kx), (stdys[i], stdx))
    return ops.conv_ydep(arr, kernels, modex=modex, modey=modey, ctx=ctx)
