"""MI355X-native differentiable Gaussian rasterizer.

``diff_gaussian_rasterization`` is the drop-in package (same public surface as
the reference's); ``gaussian_renderer.render`` is the caller-side contract;
``csrc/`` holds the HIP kernels and the C ABI (``include/ngs_rasterizer.h``).
"""
from . import diff_gaussian_rasterization  # noqa: F401
from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__version__ = "0.1.0"
