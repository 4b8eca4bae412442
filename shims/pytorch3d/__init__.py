"""Opt-in import-path shim: the three ``pytorch3d`` entry points the reference touches (``ops.knn_points``,
``renderer.look_at_rotation``, ``transforms.quaternion_to_matrix`` / ``matrix_to_quaternion``), for platforms where
pytorch3d has no build (ROCm).  NOT pytorch3d: see shims/README.md."""
from . import ops, renderer, transforms  # noqa: F401

__version__ = "0.0.0+splatter_a_video_amd.shim"
