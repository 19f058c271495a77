"""dtype rules around the hot path — reference: imgProcessor/transformations.py
(toFloatArray :78-87, toUIntArray :11-75).

``toFloatArray``'s rule (uint8/uint16 -> float32, wider ints -> float64) is
what the HIP kernels apply when asked for a float32 destination from an
integer source (``out_dtype=np.float32``: the cast is fused into the gather, a
uint16 frame is read at 2 B/px).  The host helpers below are plain dtype
casts for callers of the reference API.
"""
import numpy as np


def toFloatArray(img):
    img = np.asarray(img)
    if img.dtype.kind == 'f':
        return img
    return img.astype({1: np.float32, 2: np.float32}.get(img.dtype.itemsize, np.float64))


def toUIntArray(img, dtype=None, cutNegative=True, cutHigh=True, range=None, copy=True):  # noqa: A002
    """clip then TRUNCATE toward zero (astype), as the reference does (:71).
    (cv2.remap on integer images rounds to nearest instead; the HIP integer
    outputs follow cv2: round-half-even + saturate.)"""
    img = np.array(img, copy=copy)
    mn, mx = float(np.min(img)), float(np.max(img))
    if dtype is None:
        span = mx if cutNegative else mx - min(mn, 0)
        dtype = np.uint8 if span <= 255 else np.uint16 if span <= 65535 else np.uint32
    dtype = np.dtype(dtype)
    info = np.iinfo(dtype)
    if range is not None:
        img = (img - range[0]) * (info.max / float(range[1] - range[0]))
    if cutNegative:
        img = np.where(img < 0, 0, img)
    elif mn < 0:
        img = img - mn
    if cutHigh:
        img = np.where(img > info.max, info.max, img)
    return img.astype(dtype)
