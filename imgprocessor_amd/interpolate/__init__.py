"""Hole-filling / scattered-point stencils of the hot path (reference: imgProcessor/interpolate/)."""
from .interpolate2dStructuredIDW import interpolate2dStructuredIDW  # noqa: F401
from .interpolate2dStructuredFastIDW import interpolate2dStructuredFastIDW  # noqa: F401
from .interpolate2dUnstructuredIDW import interpolate2dUnstructuredIDW  # noqa: F401
from .interpolateCircular2dStructuredIDW import interpolateCircular2dStructuredIDW  # noqa: F401
from .interpolate2dStructuredCrossAvg import interpolate2dStructuredCrossAvg  # noqa: F401
from .interpolate2dStructuredPointSpreadIDW import interpolate2dStructuredPointSpreadIDW  # noqa: F401
