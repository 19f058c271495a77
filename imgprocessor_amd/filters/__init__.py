"""K x K filters of the hot path (reference: imgProcessor/filters/)."""
from .filter import filter, gaussian_filter, box_filter  # noqa: F401,A001
from .maskedConvolve import maskedConvolve  # noqa: F401
from ._extendArrayForConvolution import extendArrayForConvolution  # noqa: F401
from .varYSizeGaussianFilter import varYSizeGaussianFilter  # noqa: F401
from .standardDeviation import standardDeviation2d  # noqa: F401
from .maskedFilter import maskedFilter  # noqa: F401
from .nan_maximum_filter import nan_maximum_filter  # noqa: F401
from .medianThreshold import medianThreshold  # noqa: F401
from .fastFilter import fastFilter  # noqa: F401
from .fastMean import fastMean  # noqa: F401
