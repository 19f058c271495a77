"""``medianThreshold`` — reference: imgProcessor/filters/medianThreshold.py:7-30.

Set every pixel to its 3x3 median where the relative deviation
``|(img - median) / median|`` exceeds (``condition='>'``) or stays below
(``'<'``) ``threshold``.  Returns ``(img, indices)`` like the reference:
``copy=False`` writes the result back into ``img``; ``threshold <= 0`` returns
``(img, None)`` untouched.

The median is scipy's ``median_filter(size=3)`` (edge pixels repeated).  Only
``size=3`` has a HIP kernel (the selection network is fixed-size); other sizes
raise NotImplementedError — there is no CPU fallback.  NaN pixels inside a
window make scipy's own median order-dependent; that case is outside parity.
"""
import numpy as np

from .. import ops
from ..device import DeviceArray


def medianThreshold(img, threshold=0.1, size=3, condition='>', copy=True, ctx=None):
    if not threshold > 0:
        return img, None
    if size != 3:
        raise NotImplementedError('medianThreshold: only size=3 has a HIP kernel')
    out, indices = ops.median_threshold(img, threshold, condition, ctx=ctx)
    if isinstance(img, DeviceArray):
        if copy:
            return out, indices
        img.copy_from(out)
        return img, indices
    if copy or not isinstance(img, np.ndarray):
        return out, indices
    if img.dtype != out.dtype:
        raise TypeError('copy=False needs a float32/float64 array')
    img[...] = out
    return img, indices
