"""``interpolate2dStructuredIDW`` — reference:
imgProcessor/interpolate/interpolate2dStructuredIDW.py:8-65.

Every pixel flagged in ``mask`` is replaced IN PLACE by the inverse-distance-
weighted mean of the unmasked pixels within +-kernel; the weight table is
built on the host exactly as the reference does (:16-21), the stencil runs as
a wave-cooperative HIP kernel (idw.hip).  The reference clamps the window to
``gx`` instead of ``gx-1`` and so reads one element out of bounds for masked
pixels within ``kernel`` of the bottom/right edge; here the window is clamped
to the array.
"""
import numpy as np

from .. import ops


def idw_weights(kernel, power=2, fx=1, fy=1):
    # scalar float arithmetic like the reference loop (:17-21): numpy's array pow may
    # differ from libm pow in the last bit
    k = int(kernel)
    w = np.zeros((2 * k + 1, 2 * k + 1))
    for xi in range(-k, k + 1):
        for yi in range(-k, k + 1):
            dist = ((fx * xi) ** 2 + (fy * yi) ** 2)
            if dist:
                w[xi + k, yi + k] = 1 / dist ** (0.5 * power)
    return w


def interpolate2dStructuredIDW(grid, mask, kernel=15, power=2, fx=1, fy=1, ctx=None):
    return ops.idw_fill(grid, mask, int(kernel), idw_weights(kernel, power, fx, fy), ctx=ctx)
