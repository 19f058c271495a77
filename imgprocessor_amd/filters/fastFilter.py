"""``fastFilter`` — reference: imgProcessor/filters/fastFilter.py:9-122.

"A fast 2d filter for large kernel sizes that also works with nans": only every ``every``-th
pixel of the image is a window centre and only every ``every``-th pixel of its +-``ksize``
window enters the statistic (median / mean, plain or NaN-ignoring); the coarse grid of
statistics is optionally smoothed (``scipy.ndimage.gaussian_filter``) and brought back to image
size with ``cv2.resize(..., interpolation=INTER_LANCZOS4)``.

As written in the reference: ``every`` is re-derived from the row count (``every = s0 // (s0 //
every)``, :27-29), and the loops return their LAST indices, which are then used as sizes
(:36-37) - the last row and the last column of window centres are dropped before the resize.

The statistics run one wave per window on the GPU (values compacted into LDS, the median by rank
counting, `csrc/resize.hip`), the resize is OpenCV's published algorithm on the float64 grid
(cv2-unpinned, see `ops.resize`).  ``borderMode`` is accepted and unused, as in the reference.
"""
import numpy as np

from .. import ops

INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 1, 2, 3, 4   # cv2's numbers
BORDER_REFLECT = 2


def fastFilter(arr, ksize=30, every=None, resize=True, fn='median', interpolation=INTER_LANCZOS4,
               smoothksize=0, borderMode=BORDER_REFLECT, ctx=None):
    if every is None:
        every = max(ksize // 3, 1)
    else:
        assert ksize >= 3 * every
    dev = ops._is_dev(arr)   # device-resident array: statistics and enlargement stay on the device
    if not dev:
        arr = np.asarray(arr)
    s0, s1 = arr.shape[:2]
    ss0 = s0 // every
    every = s0 // ss0
    grid = ops.fast_filter_stat(arr, ksize, every, fn, ctx=ctx)
    if dev:
        if resize and not smoothksize:
            return ops.resize(grid, (s0, s1), interpolation, ctx=ctx,
                              src_shape=(grid.shape[0] - 1, grid.shape[1] - 1))
        grid = grid.get()
    out = np.ascontiguousarray(grid[:grid.shape[0] - 1, :grid.shape[1] - 1])
    if smoothksize:
        out = ops.gaussian_filter(out, smoothksize, ctx=ctx)
    if not resize:
        return out
    return ops.resize(out, (s0, s1), interpolation, ctx=ctx)
