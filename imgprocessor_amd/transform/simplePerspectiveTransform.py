"""``simplePerspectiveTransform`` — reference:
imgProcessor/transform/simplePerspectiveTransform.py:6-31."""
import numpy as np

from .. import ops
from ..utils.geometry import sortCorners, getPerspectiveTransform


def simplePerspectiveTransform(img, quad, shape=None, interpolation='linear', inverse=False,
                               ctx=None):
    p = sortCorners(quad).astype(np.float32)
    if shape is not None:
        height, width = shape
    else:
        # output size from the average quad edge lengths
        width = int(round(0.5 * (np.linalg.norm(p[0] - p[1]) + np.linalg.norm(p[3] - p[2]))))
        height = int(round(0.5 * (np.linalg.norm(p[1] - p[2]) + np.linalg.norm(p[0] - p[3]))))
    dst = np.float32([[0, 0], [width, 0], [width, height], [0, height]])
    if inverse:
        s0, s1 = img.shape[:2]
        dst /= ((width / s1), (height / s0))
        H = getPerspectiveTransform(dst, p)
    else:
        H = getPerspectiveTransform(p, dst)
    return ops.warp_perspective(img, np.linalg.inv(H), (height, width), interpolation, ctx=ctx)
