"""``filter(img, kernel)`` — the dense K x K filter of the hot path.

The reference has no generic entry point of this name: it reaches
dense filtering through ``maskedConvolve`` (filters/maskedConvolve.py:13-43),
``scipy.ndimage.convolve`` / ``gaussian_filter`` call sites
(features/hog.py:62-63, filters/standardDeviation.py:23, filters/fastFilter.py:42)
and ``cv2.blur`` (camera/lens/estimateSystematicErrorLensCorrection.py:206-207).
``filter`` is the defined counterpart:

    filter(img, kernel, mode) == scipy.ndimage.correlate(img, kernel, mode=mode)

and ``maskedConvolve(arr, k, mask) == where(mask, filter(arr, fftshift(k)), 0)``.
"""
import numpy as np

from .. import ops


def filter(img, kernel, mode='reflect', cval=0.0, mask=None, ctx=None):  # noqa: A001
    """centred correlation of a (H,W) image / (N,H,W) batch (ndarray or DeviceArray)
    with a 2-D kernel.  mode: 'reflect' (edge pixel repeated, the reference's
    padding), 'wrap', 'nearest', 'mirror', 'constant'."""
    return ops.conv2d(img, kernel, mode=mode, cval=cval, mask=mask, ctx=ctx)


def gaussian_filter(img, sigma, mode='reflect', cval=0.0, truncate=4.0, ctx=None):
    """scipy.ndimage.gaussian_filter(img, sigma) as the reference calls it: separable,
    radius int(truncate*sigma+0.5), y pass then x pass, single pass over HBM"""
    return ops.gaussian_filter(img, sigma, mode=mode, cval=cval, truncate=truncate, ctx=ctx)


def box_filter(img, ksize=3, mode='reflect101', ctx=None):
    """cv2.blur(img, (k,k)) (default border BORDER_REFLECT_101) —
    camera/lens/estimateSystematicErrorLensCorrection.py:206-207"""
    k = np.full(ksize, 1.0 / ksize)
    return ops.sepconv2d(img, k, k, mode=mode, ctx=ctx)
