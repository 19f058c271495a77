"""``interpolate2dStructuredPointSpreadIDW`` — reference:
imgProcessor/interpolate/interpolate2dStructuredPointSpreadIDW.py:7-141 ("same as
interpolate2dStructuredIDW but using the point spread method ... faster if there are bigger
connected masked areas and the border length is smaller").

The masked areas are eaten from their rims: ``_createBorder`` (:31-63) marks the masked pixels
next to unmasked ones, a sweep (:75-135) fills every marked pixel - in raster order - with the
inverse-distance-weighted mean of the unmasked pixels within ``[i-kernel, i+kernel) x
[j-kernel, j+kernel)`` and unmasks it AT ONCE, so the pixels that follow in the same sweep already
see it; border pass and sweep repeat until no transition is left (or ``maxIter`` sweeps).  The
source is reproduced as written:

* both scans of ``_createBorder`` carry their "previous value" across the ends of rows / columns,
  and an unmasked pixel that follows a masked one marks index ``j - 1`` (``i - 1``) of its own row
  (column) - for ``j = 0`` (``i = 0``) numpy's ``-1``, the LAST pixel of that row (column), which
  is then recomputed from its neighbours although it was never masked;
* the window's upper ends are exclusive, and the column limit is clamped to ``gy`` only when it
  exceeds the ROW count ``gx`` (:89-90) - with more columns than rows the window then runs to the
  end of the row; where it would leave the array (fewer columns than rows) it is clamped to the
  array here;
* border flags are only cleared by a successful fill.

``copy=True`` (default) works on copies of ``grid`` and ``mask``; ``copy=False`` modifies both, as
there.  Host side: the wrapper's three statements; device side (csrc/interp_more.hip): a border
launch per sweep (every pixel applies both scan rules by itself) and ONE workgroup of 16 waves for
the sweep, rows dealt to the waves in order, a pixel's window shared by the 64 lanes of its wave,
the raster-order dependence kept through per-row progress words in LDS.  float64 sums over lanes
instead of raster order: equal to the reference within a few ulps.
"""
import numpy as np

from .. import ops
from ..device import DeviceArray


def interpolate2dStructuredPointSpreadIDW(grid, mask, kernel=15, power=2, maxIter=1e5, copy=True,
                                          ctx=None):
    assert grid.shape == mask.shape, 'grid and mask shape are different'
    if isinstance(grid, DeviceArray):
        if copy:
            g2, m2 = grid.ctx.empty(grid.shape, grid.dtype), mask.ctx.empty(mask.shape, mask.dtype)
            g2.copy_from(grid)
            m2.copy_from(mask)
            grid, mask = g2, m2
        return ops.point_spread_idw(grid, mask, int(kernel), power, maxIter, ctx=ctx)
    if copy:
        # (the mask as well: it is modified)
        mask = np.array(mask, dtype=bool, order='C')
        grid = np.array(grid, order='C')
    return ops.point_spread_idw(grid, mask, int(kernel), power, maxIter, ctx=ctx)
