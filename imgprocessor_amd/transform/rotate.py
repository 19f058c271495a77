"""``rotate`` — reference: imgProcessor/transform/rotate.py:8-19
(cv2.getRotationMatrix2D + cv2.warpAffine, INTER_CUBIC, BORDER_REFLECT).

Reproduced as written, including its conventions for non-square images: the
centre is ((s0-1)/2, (s1-1)/2) taken as (x, y) and ``image.shape`` is passed as
cv2's dsize, i.e. (width, height) = (s0, s1).
"""
import numpy as np

from .. import ops


def rotation_matrix_2d(center, angle, scale=1.0):
    """cv2.getRotationMatrix2D"""
    a = np.deg2rad(angle)
    al, be = scale * np.cos(a), scale * np.sin(a)
    cx, cy = center
    return np.array([[al, be, (1 - al) * cx - be * cy],
                     [-be, al, be * cx + (1 - al) * cy]])


def rotate(image, angle, interpolation='cubic_cv_q5', borderMode='reflect', borderValue=0,
           ctx=None):
    """angle [deg]"""
    s0, s1 = image.shape
    image_center = (s0 - 1) / 2., (s1 - 1) / 2.
    M = np.vstack([rotation_matrix_2d(image_center, angle, 1.0), [0, 0, 1.0]])
    # warpAffine without WARP_INVERSE_MAP inverts M; dsize=image.shape -> (width, height)=(s0, s1)
    return ops.warp_perspective(image, np.linalg.inv(M), (s1, s0), interpolation, borderMode,
                                borderValue, ctx=ctx)
