"""``interpolate2dStructuredFastIDW`` — reference:
imgProcessor/interpolate/interpolate2dStructuredFastIDW.py:9-63 with
utils/growPositions.py:5-31.

Neighbours are visited in growing-distance order and the walk stops after
``minnvals`` unmasked neighbours were found (or when it runs far outside the
image), per masked pixel; weights are 1/dist**(power/2) with dist the
Euclidean distance, as the reference computes them.
"""
import numpy as np

from .. import ops


def growPositions(ksize):
    """offsets (dy,dx) of the (2k+1)^2 window sorted by distance, centre dropped
    (utils/growPositions.py); ties keep numpy's argsort order like the reference"""
    i = ksize * 2 + 1
    dist = np.fromfunction(lambda x, y: ((x - ksize) ** 2 + (y - ksize) ** 2) ** 0.5, (i, i))
    pos = np.dstack(np.unravel_index(np.argsort(dist.ravel()), (i, i)))[0, 1:]
    return pos - ksize, dist[pos[:, 0], pos[:, 1]]


def interpolate2dStructuredFastIDW(grid, mask, kernel=15, power=2, minnvals=5, ctx=None):
    indices, dist = growPositions(int(kernel))
    weights = 1 / dist ** (0.5 * power)
    return ops.fast_idw_fill(grid, mask, indices, weights, int(minnvals) - 1, ctx=ctx)
