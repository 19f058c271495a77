"""``varYSizeGaussianFilter`` — reference: imgProcessor/filters/varYSizeGaussianFilter.py:9-68.

A Gaussian whose sigma along y changes from row to row.  The per-row k0 x k1
coefficient tables are what the reference builds with one ``gaussian_filter``
call per row on a centred delta (:40-46); the delta is separable, so a table is
the outer product of two 1-D responses (same weights, truncation and
reflection; equal to the reference's tables to rounding).  The y responses of
all rows are computed ON THE DEVICE (``ipa_var_y_gauss_dev``), and the
O(H·W·k0·k1) NaN-skipping correlation (:53-68) forms its coefficients on the fly
with the borders resolved while staging (defaults modex='wrap',
modey='reflect', like the reference's padding).  ``_row_kernels`` (host tables)
remains for ``ops.conv_ydep`` callers with arbitrary tables and for the tests.

``stdyrange`` as an ndarray raises UnboundLocalError in the reference (``mx``
is never set on that branch); only the int / (mn, mx) forms are defined.
"""
import numpy as np

from .. import ops


def _fold_symmetric(k, n):
    """index of np.pad(mode='symmetric') / scipy 'reflect' for any integer position"""
    k = np.mod(k, 2 * n)
    return np.where(k < n, k, 2 * n - 1 - k)


def _delta_response(n, sigmas, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d(delta_n, sigma, mode='reflect') for EVERY sigma at once:
    (len(sigmas), n).  delta_n is 1 at n // 2; sigma == 0 leaves it untouched.  The weights are
    scipy's (exp(-x^2 / 2 sigma^2) over |x| <= int(truncate * sigma + 0.5), normalised); the
    reflected delta can be hit by several taps, which are summed."""
    sigmas = np.asarray(sigmas, dtype=np.float64)
    radius = (truncate * sigmas + 0.5).astype(np.int64)
    rmax = int(radius.max()) if sigmas.size else 0
    m = np.arange(-rmax, rmax + 1)
    with np.errstate(divide='ignore', invalid='ignore'):
        w = np.exp(-0.5 * (m[None, :] / sigmas[:, None]) ** 2)
    w[np.abs(m)[None, :] > radius[:, None]] = 0.0
    w /= w.sum(axis=1, keepdims=True)
    ident = ~(sigmas > 1e-15)
    w[ident] = 0.0
    w[ident, rmax] = 1.0
    hits = (_fold_symmetric(np.arange(n)[:, None] + m[None, :], n) == n // 2)  # (n, taps)
    return w @ hits.T.astype(np.float64)


def _row_kernels(stdys, stdx, ky, kx):
    """kernels[i] = gaussian_filter(delta_(ky,kx), (stdys[i], stdx)) of the reference (:40-46):
    the delta is separable, so every table is the outer product of two 1-D responses."""
    cols = _delta_response(ky, stdys)            # (s0, ky)
    row = _delta_response(kx, [stdx])[0]         # (kx,)
    return cols[:, :, None] * row[None, None, :]


def varYSizeGaussianFilter(arr, stdyrange, stdx=0, modex='wrap', modey='reflect', ctx=None):
    assert len(arr.shape) == 2, 'only works on 2d arrays at the moment'
    if isinstance(stdyrange, np.ndarray):
        raise UnboundLocalError("local variable 'mx' referenced before assignment "
                                "(the reference's ndarray branch is broken: pass an int or (mn, mx))")
    s0 = arr.shape[0]
    if type(stdyrange) not in (list, tuple):
        stdyrange = (0, stdyrange)
    mn, mx = stdyrange
    kx = int(stdx * 2.5)
    kx += 1 - kx % 2
    ky = int(mx * 2.5)
    ky += 1 - ky % 2
    if modey != 'reflect':
        raise Exception('modey not supported')
    if modex not in ('reflect', 'wrap'):
        raise Exception('modex not supported')
    # the per-row tables are separable (delta input): the y responses of all rows are built on
    # the device from (mn, mx); only the kx x-responses are computed here
    rowk = _delta_response(kx, [stdx])[0]
    return ops.var_y_gauss(arr, mn, mx, ky, rowk, modex=modex, modey=modey, ctx=ctx)
