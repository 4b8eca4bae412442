"""Drop-in shim: the reference's alternate renderer does
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(reference: src/pointrix/renderer/base_splatting.py:17).  With this repository on PYTHONPATH that import resolves to
the adapter over the MI355X-native operators."""
from splatter_a_video_amd.diff_rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
