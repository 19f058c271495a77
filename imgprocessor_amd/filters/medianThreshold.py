"""``medianThreshold`` — reference: imgProcessor/filters/medianThreshold.py:7-30.

Set every pixel to its ``size`` x ``size`` median where the relative deviation
``|(img - median) / median|`` exceeds (``condition='>'``) or stays below
(``'<'``) ``threshold``.  Returns ``(img, indices)`` like the reference:
``copy=False`` writes the result back into ``img``; ``threshold <= 0`` returns
``(img, None)`` untouched.

The median is scipy's ``median_filter(img, size=size)``: the element of rank
``size*size // 2`` of the window at offsets ``-size//2 .. size-1-size//2``, edge
pixels repeated.  ``size=3`` runs a 19-exchange selection network on the nine
LDS values; any other size finds the rank element by counting over the block's
LDS tile (round 4).  NaN pixels inside a window make scipy's own median
order-dependent; that case is outside parity.
"""
import numpy as np

from .. import ops
from ..device import DeviceArray


def medianThreshold(img, threshold=0.1, size=3, condition='>', copy=True, ctx=None):
    if not threshold > 0:
        return img, None
    if not (isinstance(size, (int, np.integer)) and size >= 1):
        raise NotImplementedError('medianThreshold: one integer window size for both axes')
    out, indices = ops.median_threshold(img, threshold, condition, ctx=ctx, size=int(size))
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
