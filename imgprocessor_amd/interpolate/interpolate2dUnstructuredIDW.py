"""``interpolate2dUnstructuredIDW`` — reference:
imgProcessor/interpolate/interpolate2dUnstructuredIDW.py:7-38.

``grid[i, j]`` becomes the inverse-distance-weighted mean of the n scattered values ``v`` at
positions ``(x, y)`` - ``x`` runs along axis 0 (rows), ``y`` along axis 1, as the reference
indexes ``grid[i, j]`` with ``i`` against ``x``; a pixel that coincides with a point takes the
value of the first such point.  IN PLACE on ``grid`` (float32 / float64), which is returned.
One lane per pixel, the points read through the scalar unit, float64 sums in point order
(interp_more.hip).
"""
from .. import ops


def interpolate2dUnstructuredIDW(x, y, v, grid, power=2, ctx=None):
    return ops.unstructured_idw(x, y, v, grid, power, ctx=ctx)
