"""``maskedFilter`` — reference: imgProcessor/filters/maskedFilter.py:12-102.

Median (``fn='median'``, the default) or mean (``fn='mean'``) of the UNMASKED neighbours in
a ksize window, either written into the masked pixels of ``arr`` in place
(``fill_mask=True``) or computed for the unmasked pixels into a new NaN-padded
array (``fill_mask=False``).  The window is [i-ksize//2, min(i+ksize//2, n)) on
both axes, as in the reference.

Both run as wave-cooperative HIP kernels for the (usually sparse) fill; the
median is a radix selection over the window values, bit-identical to sorting.
"""
from .. import ops


def maskedFilter(arr, mask, ksize=30, fill_mask=True, fn='median', ctx=None):
    """default ``fn='median'``; like the reference (:30-35) every value other than 'mean'
    selects the median"""
    return ops.masked_mean(arr, mask, ksize, fill_mask=fill_mask, ctx=ctx,
                           fn='mean' if fn == 'mean' else 'median')
