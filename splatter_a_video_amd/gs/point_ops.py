"""Per-Gaussian operators of the ``dptr.gs`` surface: project_point, compute_cov3d, ewa_project,
compute_sh, compute_sh_free (+ orthographic variants, an extension the reference implements in
eager torch inside its renderer).

Signatures, defaults and the autograd contract follow the reference operator files
(reference: src/submodules/dptr/dptr/gs/project_point.py:8-98, compute_cov3d.py:7-64,
ewa_project.py:8-94, compute_sh.py:8-71, compute_sh_free.py:8-71); the native layer is
libsplat_hip.so through ctypes.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import _lib as L


def _extr12(extr: Tensor) -> Tensor:
    extr = L.need(extr, "extr")
    if extr.numel() < 12:
        raise ValueError("extr must hold at least 3x4 floats (row-major [R|T])")
    return extr


def _intr4(intr: Tensor) -> Tensor:
    intr = L.need(intr, "intr")
    if intr.numel() < 4:
        raise ValueError("intr must be [fx, fy, cx, cy]")
    return intr


def _points(t: Tensor, name: str, width: int) -> Tensor:
    t = L.need(t, name)
    if t.dim() != 2 or t.shape[1] != width:
        raise ValueError(f"{name} must have shape [P, {width}], got {tuple(t.shape)}")
    return t


_ALL_VISIBLE = {}


def _visible(visible: Optional[Tensor], P: int, device) -> Tensor:
    if visible is None:   # read-only all-ones mask, created once per (P, device) instead of one fill kernel per call
        key = (int(P), str(device))
        v = _ALL_VISIBLE.get(key)
        if v is None:
            if len(_ALL_VISIBLE) > 8:
                _ALL_VISIBLE.clear()
            v = torch.ones(P, dtype=torch.uint8, device=device)
            _ALL_VISIBLE[key] = v
        return v
    v = L.need(visible, "visible", torch.uint8)
    if v.numel() != P:
        raise ValueError("visible must have P elements")
    return v


# ------------------------------------------------------------------ project_point
class _ProjectPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, intr, extr, W, H, nearest, extent, ortho):
        xyz = _points(xyz, "xyz", 3)
        extr_c = _extr12(extr)
        intr_c = None if ortho else _intr4(intr)
        P = xyz.shape[0]
        uv = torch.empty(P, 2, dtype=torch.float32, device=xyz.device)      # kernels write every element
        depth = torch.empty(P, 1, dtype=torch.float32, device=xyz.device)   # (zeros for culled points)
        L.check(L.lib().splat_project_point_forward(
            L.ci(P), L.ptr(xyz), L.ptr(intr_c), L.ptr(extr_c), L.ci(W), L.ci(H), L.cf(nearest), L.cf(extent),
            L.ci(1 if ortho else 0), L.ptr(uv), L.ptr(depth), L.stream()))
        ctx.meta = (int(W), int(H), bool(ortho))
        ctx.save_for_backward(xyz, intr_c if intr_c is not None else extr_c, extr_c, depth)
        ctx.cam_grad = (False if ortho or not isinstance(intr, Tensor) else intr.requires_grad, extr.requires_grad)
        ctx.extr_shape = extr.shape
        return uv, depth

    @staticmethod
    def backward(ctx, dL_duv, dL_ddepth):
        W, H, ortho = ctx.meta
        xyz, intr_c, extr_c, depth = ctx.saved_tensors
        P = xyz.shape[0]
        dL_duv = L.need(dL_duv, "dL_duv")
        dL_ddepth = L.need(dL_ddepth, "dL_ddepth")
        dxyz = torch.empty_like(xyz)
        g_intr, g_extr = ctx.cam_grad
        dintr = torch.zeros(4, dtype=torch.float32, device=xyz.device) if g_intr else None
        dextr = torch.zeros(12, dtype=torch.float32, device=xyz.device) if g_extr else None
        L.check(L.lib().splat_project_point_backward(
            L.ci(P), L.ptr(xyz), L.ptr(None if ortho else intr_c), L.ptr(extr_c), L.ci(W), L.ci(H),
            L.ci(1 if ortho else 0), L.ptr(depth), L.ptr(dL_duv), L.ptr(dL_ddepth), L.ptr(dxyz), L.ptr(dintr),
            L.ptr(dextr), L.stream()))
        if dextr is not None:
            full = torch.zeros(ctx.extr_shape, dtype=torch.float32, device=xyz.device)
            full.view(-1)[:12] = dextr
            dextr = full
        return dxyz, dintr, dextr, None, None, None, None, None


def project_point(xyz: Tensor, intr: Tensor, extr: Tensor, W: int, H: int, nearest: float = 0.2,
                  extent: float = 1.3) -> Tuple[Tensor, Tensor]:
    """Perspective projection + frustum culling -> (uv[P,2], depth[P,1]); culled points stay 0."""
    return _ProjectPoint.apply(xyz, intr, extr, W, H, nearest, extent, False)


def project_point_ortho(xyz: Tensor, extr: Tensor, W: int, H: int, nearest: float = 0.2,
                        extent: float = 1.3) -> Tuple[Tensor, Tensor]:
    """Orthographic twin (reference: src/pointrix/renderer/dptr_ortho_enhanced.py:145-202)."""
    return _ProjectPoint.apply(xyz, None, extr, W, H, nearest, extent, True)


# ------------------------------------------------------------------ compute_cov3d
class _ComputeCov3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scales, uquats, visible):
        scales = _points(scales, "scales", 3)
        uquats = _points(uquats, "uquats", 4)
        P = scales.shape[0]
        vis = _visible(visible, P, scales.device)
        cov3d = torch.empty(P, 6, dtype=torch.float32, device=scales.device)
        L.check(L.lib().splat_compute_cov3d_forward(L.ci(P), L.ptr(scales), L.ptr(uquats), L.ptr(vis), L.ptr(cov3d),
                                                    L.stream()))
        ctx.save_for_backward(scales, uquats, vis)
        return cov3d

    @staticmethod
    def backward(ctx, dL_dcov3d):
        scales, uquats, vis = ctx.saved_tensors
        P = scales.shape[0]
        g = L.need(dL_dcov3d, "dL_dcov3d")
        ds = torch.empty_like(scales)
        dq = torch.empty_like(uquats)
        L.check(L.lib().splat_compute_cov3d_backward(L.ci(P), L.ptr(scales), L.ptr(uquats), L.ptr(vis), L.ptr(g),
                                                     L.ptr(ds), L.ptr(dq), L.stream()))
        return ds, dq, None


def compute_cov3d(scales: Tensor, uquats: Tensor, visible: Optional[Tensor] = None) -> Tensor:
    """Sigma = R S S^T R^T from scales and UNIT quaternions (r,x,y,z) -> upper triangle [P,6]."""
    return _ComputeCov3D.apply(scales, uquats, visible)


# ------------------------------------------------------------------ ewa_project
class _EWAProject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, cov3d, intr, extr, uv, W, H, visible, ortho):
        xyz = _points(xyz, "xyz", 3)
        cov3d = _points(cov3d, "cov3d", 6)
        uv = _points(uv, "uv", 2)
        extr_c = _extr12(extr)
        intr_c = None if ortho else _intr4(intr)
        P = xyz.shape[0]
        vis = _visible(visible, P, xyz.device)
        conic = torch.empty(P, 3, dtype=torch.float32, device=xyz.device)
        radius = torch.empty(P, dtype=torch.int32, device=xyz.device)
        tiles = torch.empty(P, dtype=torch.int32, device=xyz.device)
        L.check(L.lib().splat_ewa_project_forward(
            L.ci(P), L.ptr(xyz), L.ptr(cov3d), L.ptr(intr_c), L.ptr(extr_c), L.ptr(uv), L.ci(W), L.ci(H), L.ptr(vis),
            L.ci(1 if ortho else 0), L.ptr(conic), L.ptr(radius), L.ptr(tiles), L.stream()))
        ctx.meta = (int(W), int(H), bool(ortho))
        ctx.save_for_backward(xyz, cov3d, intr_c if intr_c is not None else extr_c, extr_c, radius)
        ctx.cam_grad = (False if ortho or not isinstance(intr, Tensor) else intr.requires_grad, extr.requires_grad)
        ctx.extr_shape = extr.shape
        ctx.mark_non_differentiable(radius, tiles)
        return conic, radius, tiles

    @staticmethod
    def backward(ctx, dL_dconic, _dr, _dt):
        W, H, ortho = ctx.meta
        xyz, cov3d, intr_c, extr_c, radius = ctx.saved_tensors
        P = xyz.shape[0]
        g = L.need(dL_dconic, "dL_dconic")
        dxyz = torch.empty_like(xyz)
        dcov = torch.empty_like(cov3d)
        g_intr, g_extr = ctx.cam_grad
        dintr = torch.zeros(4, dtype=torch.float32, device=xyz.device) if g_intr else None
        dextr = torch.zeros(12, dtype=torch.float32, device=xyz.device) if g_extr else None
        L.check(L.lib().splat_ewa_project_backward(
            L.ci(P), L.ptr(xyz), L.ptr(cov3d), L.ptr(None if ortho else intr_c), L.ptr(extr_c), L.ci(W), L.ci(H),
            L.ci(1 if ortho else 0), L.ptr(radius), L.ptr(g), L.ptr(dxyz), L.ptr(dcov), L.ptr(dintr), L.ptr(dextr),
            L.stream()))
        if dextr is not None:
            full = torch.zeros(ctx.extr_shape, dtype=torch.float32, device=xyz.device)
            full.view(-1)[:12] = dextr
            dextr = full
        return dxyz, dcov, dintr, dextr, None, None, None, None, None


def ewa_project(xyz: Tensor, cov3d: Tensor, intr: Tensor, extr: Tensor, uv: Tensor, W: int, H: int,
                visible: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Perspective EWA splatting -> (conic[P,3], radius[P] int32, tiles[P] int32)."""
    return _EWAProject.apply(xyz, cov3d, intr, extr, uv, W, H, visible, False)


def ewa_project_ortho(xyz: Tensor, cov3d: Tensor, extr: Tensor, uv: Tensor, W: int, H: int,
                      visible: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Orthographic twin, J = diag(W/2, H/2) (reference: dptr_ortho_enhanced.py:18-111)."""
    return _EWAProject.apply(xyz, cov3d, None, extr, uv, W, H, visible, True)


# ------------------------------------------------------------------ compute_sh / compute_sh_free
class _ComputeSH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shs, degree, view_dirs, visible, free, sink=None):
        shs = L.need(shs, "shs")
        degree = int(degree)
        if degree < 0 or degree > 3:
            raise ValueError("SH degree must be in 0..3")
        if shs.dim() != 3 or shs.shape[2] != 3:
            raise ValueError("shs must have shape [P, D, 3]")
        P = shs.shape[0]
        if shs.shape[1] < (degree + 1) ** 2:
            raise ValueError(f"shs holds {shs.shape[1]} coefficients per point, degree {degree} needs {(degree + 1) ** 2}")
        dirs = _points(view_dirs, "view_dirs", 3)
        vis = _visible(visible, P, shs.device)
        colors = torch.empty(P, 3, dtype=torch.float32, device=shs.device)
        clamped = None if free else torch.empty(P, 3, dtype=torch.uint8, device=shs.device)
        L.check(L.lib().splat_compute_sh_forward(L.ci(P), L.ptr(shs), L.ci(degree), L.ptr(dirs), L.ptr(vis),
                                                 L.ci(1 if free else 0), L.ptr(colors), L.ptr(clamped), L.stream()))
        ctx.meta = (degree, bool(free))
        ctx.sink = sink
        if free:
            ctx.save_for_backward(shs, dirs, vis)
        else:
            ctx.save_for_backward(shs, dirs, vis, clamped)
        return colors

    @staticmethod
    def backward(ctx, dL_dcolor):
        degree, free = ctx.meta
        saved = ctx.saved_tensors
        shs, dirs, vis = saved[0], saved[1], saved[2]
        clamped = None if free else saved[3]
        P = shs.shape[0]
        g = L.need(dL_dcolor, "dL_dcolor")
        sink = ctx.sink            # caller-owned gradient buffer: the kernel adds into it, autograd sees no gradient
        if sink is not None:
            dshs = sink
        else:
            dshs = torch.zeros_like(shs) if shs.shape[1] != (degree + 1) ** 2 else torch.empty_like(shs)
        ddirs = torch.empty_like(dirs) if ctx.needs_input_grad[2] else None
        L.check(L.lib().splat_compute_sh_backward(L.ci(P), L.ptr(shs), L.ci(degree), L.ptr(dirs), L.ptr(vis),
                                                  L.ptr(clamped), L.ci(1 if free else 0), L.ptr(g),
                                                  L.ci(1 if sink is not None else 0), L.ptr(dshs), L.ptr(ddirs),
                                                  L.stream()))
        return (None if sink is not None else dshs), None, ddirs, None, None, None


def compute_sh(shs: Tensor, degree: int, view_dirs: Tensor, visible: Optional[Tensor] = None) -> Tensor:
    """RGB from real SH of degree <= 3 (shs [P,D,3]), +0.5 and clamped at 0."""
    return _ComputeSH.apply(shs, degree, view_dirs, visible, False)


def compute_sh_free(shs: Tensor, degree: int, view_dirs: Tensor, visible: Optional[Tensor] = None) -> Tensor:
    """As compute_sh without the +0.5 offset and the clamp."""
    return _ComputeSH.apply(shs, degree, view_dirs, visible, True)
