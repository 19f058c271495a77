"""Cartesian <-> polar resampling on the GPU remap kernel.

Reference call surface: imgProcessor/transform/polarTransform.py:26-105
(``linearToPolar``, ``polarToLinear`` and their ``*Maps`` builders).  Only the
coordinate grids are made on the host (two small outer products); the
resampling is ``ops.remap`` with BORDER_REFLECT.  The reference's default
``INTER_AREA`` is treated by ``cv2.remap`` as bilinear, which is the default
here.  Returned map pairs keep the reference's order: the first array is what
the reference hands to ``cv2.remap`` as ``map1`` (source x), the second ``map2``.
"""
import numpy as np

from .. import ops

_TWO_PI = 2.0 * np.pi


def _f32(*arrays):
    return tuple(np.asarray(a, dtype=np.float32) for a in arrays)


def linearToPolarMaps(shape, center=None, final_radius=None, initial_radius=None,
                      phase_width=None):
    """sampling grid of the unrolled image: rows = radius, columns = angle"""
    rows, cols = shape
    c0, c1 = ((rows - 1) / 2, (cols - 1) / 2) if center is None else center
    r_out = ((0.5 * rows) ** 2 + (0.5 * cols) ** 2) ** 0.5 if final_radius is None else final_radius
    r_in = 0 if initial_radius is None else initial_radius
    n_phi = _TWO_PI * r_out if phase_width is None else phase_width
    # sample counts are truncated to int (the reference passes the floats to np.linspace)
    radii = np.linspace(r_in, r_out, int(r_out - r_in))
    angles = np.linspace(0.75 * _TWO_PI, -0.25 * _TWO_PI, int(n_phi))
    along_c0 = np.multiply.outer(radii, np.cos(angles)) + c0
    along_c1 = np.multiply.outer(radii, np.sin(angles)) + c1
    return _f32(along_c1, along_c0)


def linearToPolar(img, center=None, final_radius=None, initial_radius=None, phase_width=None,
                  interpolation='linear', maps=None, borderValue=0, borderMode='reflect',
                  ctx=None):
    first, second = maps if maps is not None else linearToPolarMaps(
        img.shape[:2], center, final_radius, initial_radius, phase_width)
    return ops.remap(img, first, second, interpolation, borderMode, borderValue, ctx=ctx)


def polarToLinearMaps(orig_shape, out_shape=None, center=None):
    """sampling grid that rolls an unrolled (radius, angle) image back up"""
    n_r, n_phi = orig_shape
    if out_shape is None:
        side = 2 * n_r / np.sqrt(2.0)
        out_shape = (int(round(side)) - (1 - n_r % 2),
                     int(round(2 * n_phi / _TWO_PI / np.sqrt(2.0))))
    out_rows, out_cols = out_shape
    c0, c1 = (out_cols // 2, out_rows // 2) if center is None else center
    dx = np.arange(out_cols, dtype=np.float64)[None, :] - c0
    dy = np.arange(out_rows, dtype=np.float64)[:, None] - c1
    radius = np.hypot(dx, dy)
    angle = np.arctan2(dy, dx)
    column = (angle + np.pi) / _TWO_PI * (n_phi - 2)  # -pi..pi -> columns of the unrolled image
    return _f32(column, radius)


def polarToLinear(img, shape=None, center=None, maps=None, interpolation='linear', borderValue=0,
                  borderMode='reflect', ctx=None):
    first, second = maps if maps is not None else polarToLinearMaps(img.shape[:2], shape, center)
    return ops.remap(img, first, second, interpolation, borderMode, borderValue, ctx=ctx)
