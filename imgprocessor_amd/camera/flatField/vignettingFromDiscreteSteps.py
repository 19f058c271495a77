"""The resampling step of the reference's
imgProcessor/camera/flatField/vignettingFromDiscreteSteps.py (:309-314): the flat field fitted
on the coarse object grid is brought to image resolution with
``cv2.remap(ff, xx, yy, interpolation=INTER_LANCZOS4, borderMode=BORDER_REFLECT)``.
The statistics that produce `ff` (object detection, fits, repair functions) are host-side work
outside the hot path.
"""
import numpy as np

from ... import ops


def rescaleToGrid(ff, xx, yy):
    """ff sampled at the (sub-pixel) grid positions xx, yy - Lanczos4, reflecting border
    (fedcba|abcdef, cv2.BORDER_REFLECT)"""
    return ops.remap(np.asarray(ff), np.asarray(xx).astype(np.float32),
                     np.asarray(yy).astype(np.float32), 'lanczos4', 'reflect', 0.0)
