"""camera/ classes of the hot path: LensDistortion, PerspectiveCorrection, CameraCalibration."""
