"""``from pytorch3d.ops import knn_points`` (reference: src/geometry_utils.py:3; the call
``knn_points(points[None], points[None], None, None, K=K+1)`` at :15, result fields ``.dists`` / ``.idx`` read at :17) ->
the exact grid KNN of this library (HIP kernels behind ``splat_knn_*``; no CPU fallback)."""
from splatter_a_video_amd.knn import knn_points  # noqa: F401

__all__ = ["knn_points"]
