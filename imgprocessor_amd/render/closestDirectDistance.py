"""``closestDirectDistance`` — reference: imgProcessor/render/closestDirectDistance.py:6-41.

For every zero pixel of ``arr`` the distance to the closest non-zero pixel within
a +-``ksize`` window (2*ksize when there is none), 0 on the non-zero pixels.
The default ``dtype=uint16`` stores the float distance by truncation, as numba
does in the reference; ``float64`` keeps it.
"""
import numpy as np

from .. import ops


def closestDirectDistance(arr, ksize=30, dtype=np.uint16, ctx=None):
    return ops.closest_distance(arr, ksize, dtype, ctx=ctx)
