"""``from simple_knn._C import distCUDA2`` (reference: src/pointrix/utils/gaussian_points/gaussian_utils.py:5; the call
``distCUDA2(position.cuda())`` at :70-71 -- mean squared distance to the three nearest neighbours, float32 [N]) -> this
library's exact grid KNN (HIP; no CPU fallback)."""
from splatter_a_video_amd.knn import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
