"""``interpolateCircular2dStructuredIDW`` — reference:
imgProcessor/interpolate/interpolateCircular2dStructuredIDW.py:7-69.

Same as ``interpolate2dStructuredIDW`` but the distance to a neighbour is measured in polar
coordinates about ``(cx, cy)``: ``dr`` = difference of the radii, ``dphi`` = the smaller angle
between both points times their mean radius, weighted by ``fr`` / ``fphi``.  Kept as written:

* rows AND columns run to ``grid.shape[0]`` (:16-17) - a grid with more columns than rows is
  filled only in its first ``shape[0]`` columns, one with fewer columns indexes out of bounds in
  the reference and raises ``ValueError`` here;
* the window is ``[i-kernel, min(i+kernel, gx))`` - upper end exclusive (:24-37, :47-48);
* the distance is the SQUARE of ``(fr dr)^2 + (fphi dphi)^2`` (:56), so ``power=2`` weights
  with the fourth power of the polar distance.

IN PLACE on ``grid`` (float32 / float64), which is returned.  One wave per masked pixel, its
64 lanes over the window, float64 arithmetic (interp_more.hip).
"""
from .. import ops


def interpolateCircular2dStructuredIDW(grid, mask, kernel=15, power=2, fr=1, fphi=1, cx=0, cy=0,
                                       ctx=None):
    return ops.circular_idw_fill(grid, mask, int(kernel), power, fr, fphi, cx, cy, ctx=ctx)
