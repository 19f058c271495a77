"""Cartesian <-> polar maps — reference: imgProcessor/transform/polarTransform.py:26-105.
The maps are built on the host exactly as the reference builds them; the remap runs on the
GPU (the reference's default INTER_AREA degrades to bilinear inside cv2.remap; BORDER_REFLECT)."""
import numpy as np

from .. import ops


def _polar2cart(r, phi, center):
    return r * np.cos(phi) + center[0], r * np.sin(phi) + center[1]


def _cart2polar(x, y, center):
    xx, yy = x - center[0], y - center[1]
    return np.hypot(xx, yy), np.arctan2(yy, xx)


def linearToPolarMaps(shape, center=None, final_radius=None, initial_radius=None,
                      phase_width=None):
    s0, s1 = shape
    if center is None:
        center = (s0 - 1) / 2, (s1 - 1) / 2
    if final_radius is None:
        final_radius = ((0.5 * s0) ** 2 + (0.5 * s1) ** 2) ** 0.5
    if initial_radius is None:
        initial_radius = 0
    if phase_width is None:
        phase_width = 2 * np.pi * final_radius
    # (np.linspace wants integer sample counts on current numpy; the reference passes floats)
    phi, R = np.meshgrid(np.linspace(1.5 * np.pi, -0.5 * np.pi, int(phase_width)),
                         np.linspace(initial_radius, final_radius,
                                     int(final_radius - initial_radius)))
    mapX, mapY = _polar2cart(R, phi, center)
    return mapY.astype(np.float32), mapX.astype(np.float32)


def linearToPolar(img, center=None, final_radius=None, initial_radius=None, phase_width=None,
                  interpolation='linear', maps=None, borderValue=0, borderMode='reflect',
                  ctx=None):
    if maps is None:
        mapY, mapX = linearToPolarMaps(img.shape[:2], center, final_radius, initial_radius,
                                       phase_width)
    else:
        mapY, mapX = maps
    # the reference calls cv2.remap(img, mapY, mapX): its "mapY" is cv2's map1 (x coordinates)
    return ops.remap(img, mapY, mapX, interpolation, borderMode, borderValue, ctx=ctx)


def polarToLinearMaps(orig_shape, out_shape=None, center=None):
    s0, s1 = orig_shape
    if out_shape is None:
        out_shape = (int(round(2 * s0 / 2 ** 0.5)) - (1 - s0 % 2),
                     int(round(2 * s1 / (2 * np.pi) / 2 ** 0.5)))
    ss0, ss1 = out_shape
    if center is None:
        center = ss1 // 2, ss0 // 2
    yy, xx = np.mgrid[0:ss0:1., 0:ss1:1.]
    r, phi = _cart2polar(xx, yy, center)
    phi = (phi + np.pi) / (2 * np.pi) * (s1 - 2)  # -pi..pi -> 0..s1
    return phi.astype(np.float32), r.astype(np.float32)


def polarToLinear(img, shape=None, center=None, maps=None, interpolation='linear', borderValue=0,
                  borderMode='reflect', ctx=None):
    if maps is None:
        mapY, mapX = polarToLinearMaps(img.shape[:2], shape, center)
    else:
        mapY, mapX = maps
    return ops.remap(img, mapY, mapX, interpolation, borderMode, borderValue, ctx=ctx)
