from .vignettingFromDiscreteSteps import rescaleToGrid  # noqa: F401
