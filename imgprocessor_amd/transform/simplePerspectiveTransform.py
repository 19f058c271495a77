"""Quad -> rectangle perspective warp on the GPU.

Reference call surface: imgProcessor/transform/simplePerspectiveTransform.py:6-31.
The four corners are ordered (top-left, top-right, bottom-right, bottom-left),
the target rectangle is either ``shape`` or the mean lengths of opposite quad
edges, and the warp is ``ops.warp_perspective`` with the homography of
``utils.geometry.getPerspectiveTransform`` (8x8 solve, like cv2's).  With
``inverse=True`` the rectangle is first scaled to the image size and the
mapping direction is swapped, as in the reference.
"""
import numpy as np

from .. import ops
from ..utils.geometry import sortCorners, getPerspectiveTransform


def _edge(a, b):
    return np.linalg.norm(a - b)  # float32, like the reference's edge lengths


def simplePerspectiveTransform(img, quad, shape=None, interpolation='linear', inverse=False,
                               ctx=None):
    tl, tr, br, bl = corners = sortCorners(quad).astype(np.float32)
    if shape is None:
        out_w = int(round((_edge(tl, tr) + _edge(bl, br)) / 2))
        out_h = int(round((_edge(tr, br) + _edge(tl, bl)) / 2))
    else:
        out_h, out_w = shape
    rect = np.array([(0, 0), (out_w, 0), (out_w, out_h), (0, out_h)], dtype=np.float32)
    if inverse:
        rows, cols = img.shape[:2]
        rect /= (out_w / cols, out_h / rows)
        src_pts, dst_pts = rect, corners
    else:
        src_pts, dst_pts = corners, rect
    forward = getPerspectiveTransform(src_pts, dst_pts)
    return ops.warp_perspective(img, np.linalg.inv(forward), (out_h, out_w), interpolation,
                                ctx=ctx)
