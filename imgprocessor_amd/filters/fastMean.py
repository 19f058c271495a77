"""``fastMean`` — reference: imgProcessor/filters/fastMean.py:5-19.

"For bigger ksizes it is often faster to resize an image rather than blur it": the image is
shrunk to ``round(shape / f)`` with ``cv2.resize(..., INTER_AREA)`` and enlarged back with
``INTER_LINEAR``.  Both resizes are OpenCV's published algorithm on the GPU (`ops.resize`,
cv2-unpinned) for float32 / float64 images.  Integer images (the reference's demo feeds a uint8
one) keep their dtype like ``cv2.resize`` does: they are resized in float32 (``toFloatArray``'s
rule for uint8 / uint16; cv2's 8-bit fixed-point arithmetic differs between OpenCV versions and is
not restated) and BOTH results - ``cv2.resize`` hands an integer image back each time: the small
one before it is enlarged, and the enlarged one - are rounded half-to-even and saturated to the
dtype's range (``cv::saturate_cast``), for the returned array and for the in-place write alike.
``inplace=True`` writes the result into ``img``.
"""
import numpy as np

from .. import ops


def fastMean(img, f=10, inplace=False, ctx=None):
    if ops._is_dev(img):   # device-resident image (float32 / float64): stays on the device
        s0, s1 = img.shape
        small = ops.resize(img, (int(round(s0 / f)), int(round(s1 / f))), 'area', ctx=ctx)
        return ops.resize(small, (s0, s1), 'linear', out=img if inplace else None, ctx=ctx)
    src = np.asarray(img)
    s0, s1 = src.shape[:2]
    ss0 = int(round(s0 / f))
    ss1 = int(round(s1 / f))
    small = ops.resize(src, (ss0, ss1), 'area', ctx=ctx)
    if src.dtype.kind in 'ui':
        info = np.iinfo(src.dtype)
        small = np.clip(np.rint(small), info.min, info.max).astype(small.dtype)   # cv2's integer `small`
    big = ops.resize(small, (s0, s1), 'linear', ctx=ctx)
    if src.dtype.kind in 'ui':
        big = np.clip(np.rint(big), info.min, info.max).astype(src.dtype)
    if inplace:
        img[...] = big
        return img
    return big
