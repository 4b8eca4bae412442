"""``diff_gaussian_rasterization`` call shape over the native operators.

The reference's alternate renderer builds a ``GaussianRasterizationSettings`` and calls a ``GaussianRasterizer``
(reference: src/pointrix/renderer/base_splatting.py:17,123-174); that third-party CUDA package is not in the reference
tree and has no version pin (``requirements.txt`` does not list it), so this adapter is anchored on the call site:
same class names, field names, keyword arguments, return tuple and error behaviour, computed by the perspective
operator chain of ``splatter_a_video_amd.gs`` (project_point -> compute_cov3d -> ewa_project -> compute_sh ->
sort_gaussian -> alpha_blending).  **Parity with the third-party binary is unpinned.**

Conventions taken from the call site and the package's published interface:
  * ``viewmatrix`` / ``projmatrix`` are the TRANSPOSED 4x4 matrices the reference's camera stores
    (``world_view_transform``, ``full_proj_transform``): a row vector p maps to ``p @ viewmatrix``;
  * focal lengths and principal point come from the projection the two matrices imply
    (``P = projmatrix @ viewmatrix^-1``), which for the symmetric frustum of the reference's camera is
    ``fx = W / (2 tanfovx)``, ``cx = W / 2``;
  * ``means2D`` is the zero [P,3] tensor whose ``.grad`` receives the screen-space gradient: its first two columns
    get ``dL_duv * [W/2, H/2]`` (the same tap the ``ndc`` argument of ``alpha_blending`` feeds), the third stays 0;
  * ``bg`` is one background value per colour channel;
  * returns ``(color[3,H,W], radii[P] int32)``, ``radii`` 0 for culled Gaussians.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import Tensor

from . import gs

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


def _camera(s: GaussianRasterizationSettings, device):
    """(intr[4] = fx, fy, cx, cy ; extr[3,4]) as device tensors, no host synchronisation"""
    W, H = int(s.image_width), int(s.image_height)
    view_t = s.viewmatrix.to(device=device, dtype=torch.float32)
    full_t = s.projmatrix.to(device=device, dtype=torch.float32)
    if view_t.shape != (4, 4) or full_t.shape != (4, 4):
        raise ValueError("viewmatrix and projmatrix must be 4x4")
    extr = view_t.t()[:3, :].contiguous()
    # full_t = view_t @ proj_t  ->  proj_t = view_t^-1 @ full_t ; proj = proj_t^T maps camera space to clip space
    proj_t = torch.linalg.solve(view_t, full_t)
    fx = proj_t[0, 0] * (0.5 * W)
    fy = proj_t[1, 1] * (0.5 * H)
    cx = (1.0 + proj_t[2, 0]) * (0.5 * W)
    cy = (1.0 + proj_t[2, 1]) * (0.5 * H)
    return torch.stack([fx, fy, cx, cy]).contiguous(), extr


class GaussianRasterizer(torch.nn.Module):
    #: near plane and screen-extent culling of the projection (the operator defaults of ``gs.project_point``)
    nearest = 0.2
    extent = 1.3

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: Tensor) -> Tensor:
        """bool[P]: Gaussians that survive the projection's culling"""
        s = self.raster_settings
        with torch.no_grad():
            intr, extr = _camera(s, positions.device)
            _, depth = gs.project_point(positions, intr, extr, int(s.image_width), int(s.image_height), self.nearest,
                                        self.extent)
        return depth[:, 0] != 0

    def forward(self, means3D: Tensor, means2D: Optional[Tensor], opacities: Tensor, shs: Optional[Tensor] = None,
                colors_precomp: Optional[Tensor] = None, scales: Optional[Tensor] = None,
                rotations: Optional[Tensor] = None, cov3D_precomp: Optional[Tensor] = None):
        s = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        W, H = int(s.image_width), int(s.image_height)
        dev = means3D.device
        intr, extr = _camera(s, dev)

        uv, depth = gs.project_point(means3D, intr, extr, W, H, self.nearest, self.extent)
        visible = depth != 0
        if cov3D_precomp is None:
            scl = scales if float(s.scale_modifier) == 1.0 else scales * float(s.scale_modifier)
            cov3d = gs.compute_cov3d(scl, rotations, visible)
        else:
            cov3d = cov3D_precomp
        conic, radius, tiles = gs.ewa_project(means3D, cov3d, intr, extr, uv, W, H, visible)

        if shs is not None:
            dirs = torch.nn.functional.normalize(means3D - s.campos.to(device=dev, dtype=torch.float32).reshape(1, 3), dim=1)
            rgb = gs.compute_sh(shs, int(s.sh_degree), dirs, visible)
        else:
            rgb = colors_precomp
        if rgb.dim() != 2 or rgb.shape[1] != 3:
            raise ValueError("colours must be [P,3]")

        idx_sorted, tile_range, _ = gs.sort_gaussian_capped(uv, depth, W, H, radius, None, conic.detach(), opacities.detach())
        ndc = None
        if means2D is not None and means2D.requires_grad:
            if means2D.dim() != 2 or means2D.shape[0] != means3D.shape[0] or means2D.shape[1] < 2:
                raise ValueError("means2D must be [P,3] (or [P,2])")
            ndc = means2D[:, :2]
        # one background per channel: sum_k w_k c_k + T bg == sum_k w_k (c_k - bg) + bg, because sum_k w_k + T == 1 holds
        # term by term for the applied contributors; the operator keeps its scalar-background contract
        bg = s.bg.to(device=dev, dtype=torch.float32).reshape(-1)
        if bg.numel() == 1:
            bg = bg.expand(3)
        if bg.numel() != 3:
            raise ValueError("bg must hold one value per colour channel")
        color = gs.alpha_blending(uv, conic, opacities, rgb - bg[None, :], idx_sorted, tile_range, 0.0, W, H, ndc)
        color = color + bg[:, None, None]
        return color, radius
