"""``standardDeviation2d`` — reference: imgProcessor/filters/standardDeviation.py:9-70.

Local standard deviation around a Gaussian-blurred mean.  Reference quirks
kept: ``ksize`` is always expanded to ``(ksize, ksize)`` (the ``not in (list,
tuple)`` test compares a value with types, :19-20) and is passed to
``gaussian_filter`` as SIGMA (:23); the window is [i-h, i+h) clipped to the
image and the divisor is (rows-1)*(cols-1) (:64-69).  Both the blur (separable
filter) and the window reduction run on the GPU.
"""
from .. import ops
from ..device import DeviceArray


def standardDeviation2d(img, ksize=5, blurred=None, ctx=None):
    ksize = (ksize, ksize)
    if blurred is None:
        blurred = ops.gaussian_filter(img, ksize, ctx=ctx)
    else:
        assert tuple(blurred.shape) == tuple(img.shape)
    if isinstance(img, DeviceArray) and not isinstance(blurred, DeviceArray):
        blurred = img.ctx.to_device(blurred)
    return ops.local_std(img, blurred, ksize, ctx=ctx)
