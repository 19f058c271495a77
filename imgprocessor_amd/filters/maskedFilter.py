"""``maskedFilter`` — reference: imgProcessor/filters/maskedFilter.py:12-37.

Mean of the UNMASKED neighbours in a ksize window, either written into the
masked pixels of ``arr`` in place (``fill_mask=True``) or computed for the
unmasked pixels into a new NaN-padded array (``fill_mask=False``).  The
window is [i-ksize//2, min(i+ksize//2, n)) on both axes, as in the reference.

Only ``fn='mean'`` runs on the GPU; ``fn='median'`` (window selection over up
to ksize**2 values per pixel, :75-108) is not part of the hot path and raises
NotImplementedError — there is no CPU fallback.
"""
from .. import ops


def maskedFilter(arr, mask, ksize=30, fill_mask=True, fn='mean', ctx=None):
    if fn == 'mean':
        return ops.masked_mean(arr, mask, ksize, fill_mask=fill_mask, ctx=ctx)
    if fn == 'median':
        raise NotImplementedError("maskedFilter(fn='median') has no HIP kernel")
    raise ValueError("fn must be 'mean' or 'median'")
