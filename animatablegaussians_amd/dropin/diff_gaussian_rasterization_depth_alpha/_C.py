"""Stand-in for the reference's pybind module ``diff_gaussian_rasterization_depth_alpha._C`` (``ext.cpp:15-19``)."""
from animatablegaussians_amd.rasterizer import native_mark_visible as mark_visible  # noqa: F401
from animatablegaussians_amd.rasterizer import native_rasterize_gaussians as rasterize_gaussians  # noqa: F401
from animatablegaussians_amd.rasterizer import native_rasterize_gaussians_backward as rasterize_gaussians_backward  # noqa: F401
