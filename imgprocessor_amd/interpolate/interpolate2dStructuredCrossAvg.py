"""``interpolate2dStructuredCrossAvg`` — reference:
imgProcessor/interpolate/interpolate2dStructuredCrossAvg.py:7-115 ("useful if large empty areas
need to be filled").

Every masked pixel searches along its column and row for the nearest unmasked pixel in each of
the four directions, takes the local average there (``_localAvg`` :21-44: the unmasked pixels
within +-kernel) and blends the averages with weights ``1 / distance^(power/2)``.  The source
is reproduced as written:

* the search towards the last row stores its value in slot 1 but raises ``valid[2]`` (:79-81),
  so that value never enters the blend, and slot 2 - the search towards column 0 - counts as
  valid when either of the two succeeded; when only the former did, slot 2 still holds what the
  last earlier masked pixel (raster order) with an unmasked pixel to its left stored there.
  Before any such pixel the slot is ``np.empty`` garbage in the reference; it is left out here;
* the search towards the last column runs only when ``i < gy - 1`` - the ROW index against the
  column count (:96);
* distances are uint16, the weights float32 and normalised in float32, the local averages are
  rounded to the grid's dtype before blending (the ``vals`` array has it);
* ``_localAvg`` clamps its window to ``gx`` / ``gy`` instead of ``gx-1`` / ``gy-1`` and reads
  that index (out of bounds); the window is clamped to the array here.

IN PLACE on ``grid`` (float32 / float64), which is returned.  Four launches: the rows' stale
slot sources, a prefix over rows, the local averages at every unmasked pixel next to a masked
one, and the fill (one wave per masked pixel, 64 positions per search step) - interp_more.hip.
"""
from .. import ops


def interpolate2dStructuredCrossAvg(grid, mask, kernel=15, power=2, ctx=None):
    return ops.cross_avg_fill(grid, mask, int(kernel), power, ctx=ctx)
