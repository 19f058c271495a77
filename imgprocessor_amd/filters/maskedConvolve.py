"""``maskedConvolve`` — reference: imgProcessor/filters/maskedConvolve.py:13-43.

The reference pads with extendArrayForConvolution and then indexes
``kernel[ii, jj]`` with ii, jj in [-h, h]: NEGATIVE indices wrap, so what it
computes is a centred correlation with ``np.fft.fftshift(kernel)`` on
mask==True pixels and 0 elsewhere.  That is reproduced here: the (tiny)
kernel is rolled on the host, the stencil runs in the HIP conv kernel with
the border resolved while staging (no padded copy).
"""
from __future__ import print_function

import numpy as np

from .. import ops
from ..device import DeviceArray


def maskedConvolve(arr, kernel, mask, mode='reflect', ctx=None):
    kernel = np.asarray(kernel, dtype=np.float64)
    if kernel.ndim != 2 or kernel.shape[0] != kernel.shape[1] or kernel.shape[0] % 2 == 0:
        # non-square kernels index out of bounds in the reference's _calc
        raise ValueError('maskedConvolve needs a square kernel of odd size')
    if mode != 'reflect':
        # the reference forwards mode to modex AND modey; modey='wrap' raises there
        # (_extendArrayForConvolution.py:57), so 'reflect' is the only runnable value
        raise Exception('modey not supported')
    h = kernel.shape[0] // 2
    shape = arr.shape
    print((shape[0] + 2 * h, shape[1] + 2 * h))  # the reference prints the padded shape (:19)
    k = np.fft.fftshift(kernel)
    if isinstance(arr, DeviceArray) and not isinstance(mask, DeviceArray):
        mask = arr.ctx.to_device(np.ascontiguousarray(mask, dtype=np.uint8))
    return ops.conv2d(arr, k, mode='reflect', mask=mask, ctx=ctx)
