"""``nan_maximum_filter`` — reference: imgProcessor/filters/nan_maximum_filter.py:6-37.

NaN-ignoring maximum over the window [i-ksize//2, min(i+ksize//2, n)) on both
axes; NaN where the whole window is NaN.
"""
from .. import ops


def nan_maximum_filter(arr, ksize, ctx=None):
    return ops.nan_max(arr, ksize, ctx=ctx)
