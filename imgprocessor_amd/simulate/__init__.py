from .navierStokes import shiftImage  # noqa: F401
