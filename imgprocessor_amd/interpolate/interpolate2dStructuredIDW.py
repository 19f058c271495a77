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
    k = int(kernel)
    xi = np.arange(-k, k + 1, dtype=np.float64)
    dist = (fx * xi[:, None]) ** 2 + (fy * xi[None, :]) ** 2
    w = np.zeros_like(dist)
    nz = dist != 0
    w[nz] = 1.0 / dist[nz] ** (0.5 * power)
    return w


def interpolate2dStructuredIDW(grid, mask, kernel=15, power=2, fx=1, fy=1, ctx=None):
    return ops.idw_fill(grid, mask, int(kernel), idw_weights(kernel, power, fx, fy), ctx=ctx)
