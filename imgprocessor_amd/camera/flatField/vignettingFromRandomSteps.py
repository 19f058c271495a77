"""The three resampling steps of the reference's
imgProcessor/camera/flatField/vignettingFromRandomSteps.py (ObjectVignettingSeparation):

  :245  ``cv2.warpPerspective(self.flatField, h, (f.shape[1], f.shape[0]))``
        the current flat field seen from the position of one fitted image
  :260  ``cv2.warpPerspective(div, h, (sh[1], sh[0]))`` on ``div = f / object`` with the
        background set to NaN, then ``np.nan_to_num``: the ratio image brought back onto the
        flat-field grid (NaN sources stay out of the interpolation: a footprint that touches a
        NaN gives NaN, and NaN becomes 0 = "no value" for the masked average that follows)
  :309  ``cv2.warpPerspective(img, H_inv, (s[1], s[0]))`` in ``_fitImg``: an input image
        fitted onto the object grid

All three are plain calls (no WARP_INVERSE_MAP): OpenCV inverts the 3x3 matrix and samples
bilinearly with coordinates rounded to 1/32 px, constant border 0.  Here that is
``ops.warp_perspective(..., 'linear_cv_q5')`` with inv(H) - float32 and float64 (the flat
field and the ratio images are float64 in the reference) run in their own precision.
Feature matching, the homography search and the masked moving averages around these calls
are host-side work outside the hot path.
"""
import numpy as np

from ... import ops
from ...device import DeviceArray


def _warp(img, H, shape, ctx=None):
    Minv = np.linalg.inv(np.asarray(H, dtype=np.float64))
    dev = isinstance(img, DeviceArray)
    if not dev:
        img = np.asarray(img)
        if img.dtype not in (np.uint8, np.uint16, np.float32, np.float64):
            img = img.astype(np.float64)
    return ops.warp_perspective(img, Minv, (int(shape[0]), int(shape[1])), 'linear_cv_q5',
                                'constant', 0.0, ctx=ctx)


def warpFlatField(flatField, Hinv, fit_shape, ctx=None):
    """:245 - the flat field warped onto a fitted image's grid (`Hinv` is the matrix the
    reference passes: the inverse homography of that image); fit_shape = f.shape"""
    return _warp(flatField, Hinv, fit_shape, ctx)


def warpRatioToFlatField(fit, obj, fit_mask, H, ff_shape, ctx=None):
    """:255-262 - ``div = fit / obj; div[fit_mask] = nan; warp(div, H); nan_to_num``"""
    with np.errstate(divide='ignore', invalid='ignore'):
        div = np.asarray(fit) / np.asarray(obj)
    div[np.asarray(fit_mask, dtype=bool)] = np.nan
    out = _warp(div, H, ff_shape, ctx)
    return np.nan_to_num(out)


def fitToObject(img, H_inv, obj_shape, ctx=None):
    """:309 - an input image warped onto the object grid"""
    return _warp(img, H_inv, obj_shape, ctx)
