"""Other callers of the remap kernels (reference: imgProcessor/transform/) — thin host
wrappers: matrices / maps are built on the host, the per-pixel work runs in the HIP kernels."""
from .rotate import rotate  # noqa: F401
from .simplePerspectiveTransform import simplePerspectiveTransform  # noqa: F401
from .polarTransform import (linearToPolar, polarToLinear, linearToPolarMaps,  # noqa: F401
                             polarToLinearMaps)
