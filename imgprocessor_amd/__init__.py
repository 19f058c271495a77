"""imgprocessor_amd — MI355X (gfx950) implementation of imgProcessor's per-pixel
hot path: lens-undistort / perspective remap, K x K filters, IDW stencils.

Module layout mirrors the reference package (radjkarl/imgProcessor) for the
parts that are on the path:

    imgprocessor_amd.camera.LensDistortion.LensDistortion
    imgprocessor_amd.camera.PerspectiveCorrection.PerspectiveCorrection
    imgprocessor_amd.filters.{filter, maskedConvolve, gaussian_filter, ...}
    imgprocessor_amd.interpolate.{interpolate2dStructuredIDW, ...FastIDW}

All arithmetic runs in hand-written HIP kernels reached through the ctypes
C ABI of libimgproc_hip.so (include/imgproc_hip.h); there is no CPU fallback.
Arrays are indexed array[y, x] like the reference (imgProcessor/__init__.py:9).
"""
__version__ = '0.1.0'

from .device import Context, DeviceArray, default_context, device_count  # noqa: F401
from . import ops  # noqa: F401
