"""IDW hole-filling stencils of the hot path (reference: imgProcessor/interpolate/)."""
from .interpolate2dStructuredIDW import interpolate2dStructuredIDW  # noqa: F401
from .interpolate2dStructuredFastIDW import interpolate2dStructuredFastIDW  # noqa: F401
