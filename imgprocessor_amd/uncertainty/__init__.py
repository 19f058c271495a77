"""uncertainty/ stencils of the reference that share the hot path's kernels."""
from .positionToIntensityUncertainty import positionToIntensityUncertainty  # noqa: F401
