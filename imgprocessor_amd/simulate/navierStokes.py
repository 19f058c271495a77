"""``shiftImage`` of the reference's imgProcessor/simulate/navierStokes.py (:52-62): remap an
image along a velocity field.  The flow solver around it (``navierStokes2d``) is a host-side
numba stencil iteration outside the hot path (SURVEY section 8 f1 lists only the remap).
"""
import numpy as np

from .. import ops
from ..device import DeviceArray


def shiftImage(u, v, t, img, interpolation='lanczos4'):
    """remap `img` by the displacement (u, v) * t:
    cv2.remap(img.astype(float32), x + u t, y + v t, INTER_LANCZOS4) - cv2's default border
    (constant 0).  `interpolation` takes the names of ops.INTERPOLATIONS."""
    u = np.asarray(u)
    v = np.asarray(v)
    ny, nx = u.shape
    sy, sx = np.mgrid[:float(ny):1, :float(nx):1]
    sx += u * t
    sy += v * t
    if isinstance(img, DeviceArray):
        ctx = img.ctx
        return ops.remap(img, ctx.to_device(sx.astype(np.float32)),
                         ctx.to_device(sy.astype(np.float32)), interpolation, 'constant', 0.0,
                         out_dtype=np.float32)
    return ops.remap(np.asarray(img).astype(np.float32), sx.astype(np.float32),
                     sy.astype(np.float32), interpolation, 'constant', 0.0)
