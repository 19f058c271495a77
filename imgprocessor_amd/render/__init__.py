"""render/ stencils of the reference that share the hot path's kernels."""
from .closestDirectDistance import closestDirectDistance  # noqa: F401
