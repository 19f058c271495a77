from .geometry import (genericCameraMatrix, sortCorners, getPerspectiveTransform,  # noqa: F401
                       getOptimalNewCameraMatrix, perspectiveTransform)
