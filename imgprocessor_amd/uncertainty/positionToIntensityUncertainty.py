"""``positionToIntensityUncertainty`` — reference:
imgProcessor/uncertainty/positionToIntensityUncertainty.py:52-89 (+ the two numba loops :7-49).

Standard deviation of the intensity a pixel would show if it were displaced
within a Gaussian point spread function with std. dev. ``sx``, ``sy`` (single
values or per-pixel maps).  Kept as written in the reference:

  * the Gaussian is ``numbaGaussian2d(psf, sx, sy)`` whose parameters are named
    ``(sy, sx)`` — the FIRST value acts on the row axis (:14, :39);
  * pixels closer than ``kernelSize // 2`` to the frame and NaN pixels stay 0;
  * unsigned images are widened first (:74-75); the result is float64.

``kernelSize=None`` derives ``max(3, 4*std + 1)`` (:91-92); the reference then
fails for non-integer results (a float is used as an array shape), here the
value is truncated to int.
"""
import numpy as np

from .. import ops
from ..device import DeviceArray


def _kSizeFromStd(std):
    return max(3, 4 * std + 1)


def positionToIntensityUncertainty(image, sx, sy, kernelSize=None, ctx=None):
    psf_is_const = not isinstance(sx, (np.ndarray, DeviceArray))
    if not psf_is_const:
        assert tuple(image.shape) == tuple(sx.shape) == tuple(sy.shape), \
            'Image and position uncertainty maps need to have same size'
        if kernelSize is None:
            if isinstance(sx, DeviceArray):
                raise ValueError('give kernelSize with device-resident sigma maps')
            kernelSize = _kSizeFromStd(max(sx.max(), sy.max()))
    else:
        assert type(sx) in (int, float) and type(sy) in (int, float), \
            'Image and position uncertainty values need to be int OR float'
        if kernelSize is None:
            kernelSize = _kSizeFromStd(max(sx, sy))
    if not isinstance(image, DeviceArray):
        image = np.asarray(image)
        if image.dtype.kind in 'ui':
            image = image.astype(np.float64)
    size = int(kernelSize) // 2
    if size < 1:
        size = 1
    return ops.pos_intensity_unc(image, sx, sy, size, ctx=ctx)
