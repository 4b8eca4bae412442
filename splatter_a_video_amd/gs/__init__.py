"""MI355X-native implementation of the ``dptr.gs`` operator surface
(reference: src/submodules/dptr/dptr/gs/__init__.py:3-26).  ``import dptr.gs as gs`` resolves here
through the top-level ``dptr`` shim package of this repository."""
from .point_ops import (compute_cov3d, compute_sh, compute_sh_free, ewa_project, ewa_project_ortho, project_point,
                        project_point_ortho)
from .fused_ops import compute_sh_into, preprocess_ortho, preprocess_persp
from .raster_ops import (SortStatus, alpha_blending, alpha_blending_shared, alpha_blending_enhanced, alpha_blending_with_bias, rasterization,
                         rasterization_ortho, sort_gaussian, sort_gaussian_capped)

__all__ = [
    "project_point",
    "compute_cov3d",
    "ewa_project",
    "sort_gaussian",
    "compute_sh",
    "compute_sh_free",
    "alpha_blending",
    "rasterization",
    "alpha_blending_enhanced",
    "alpha_blending_with_bias",
    # extensions (orthographic camera ops the reference renderer does in eager torch)
    "project_point_ortho",
    "ewa_project_ortho",
    # fused per-frame operators of the MI355X renderer
    "preprocess_ortho",
    "preprocess_persp",
    "compute_sh_into",
    "alpha_blending_shared",
    "sort_gaussian_capped",
    "SortStatus",
    "rasterization_ortho",
]
