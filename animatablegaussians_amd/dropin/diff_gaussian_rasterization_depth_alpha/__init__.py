"""Package name the reference's ``gaussians/gaussian_renderer.py:14`` imports; backed by libag_hip.so."""
from animatablegaussians_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _RasterizeGaussians,
    rasterize_gaussians,
)
