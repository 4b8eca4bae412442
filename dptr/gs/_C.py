"""``dptr.gs._C`` over the C ABI of libsplat_hip.so: the 18 functions the reference binds with pybind11
(reference: src/submodules/dptr/dptr/gs/src/ext.cpp:14-33), with the reference's argument tuples and return tuples
(torch tensors in, freshly allocated torch tensors out, zero-filled where the reference's ``torch::zeros`` is semantic).
With this module the reference's OWN operator files (``dptr/gs/project_point.py`` ... ``alpha_blending_with_bias.py``,
which do ``import dptr.gs._C as _C``) run unchanged on the HIP kernels -- the lowest drop-in level; the package one level
up (``dptr.gs``, native autograd Functions, atomic-free backward, fused operators) is the one to use.

No autograd here (the reference's ``_C`` has none either) and no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
from torch import Tensor

from splatter_a_video_amd import _lib as L
from splatter_a_video_amd.gs.point_ops import _extr12, _intr4, _points, _visible


def _tiles(W: int, H: int) -> int:
    return ((W + 15) // 16) * ((H + 15) // 16)


def _vis(visible: Optional[Tensor], P: int, device) -> Tensor:
    if visible is not None and visible.dim() > 1:
        visible = visible.reshape(-1)
    return _visible(visible, P, device)


# ------------------------------------------------------------------ project_point (project_point.cu:147-245)
def project_point_forward(xyz, intr, extr, W, H, nearest, extent) -> Tuple[Tensor, Tensor]:
    xyz = _points(xyz, "xyz", 3)
    P, dev = xyz.shape[0], xyz.device
    uv = torch.empty(P, 2, dtype=torch.float32, device=dev)
    depth = torch.empty(P, 1, dtype=torch.float32, device=dev)
    L.check(L.lib().splat_project_point_forward(L.ci(P), L.ptr(xyz), L.ptr(_intr4(intr)), L.ptr(_extr12(extr)), L.ci(W), L.ci(H),
                                                L.cf(nearest), L.cf(extent), L.ci(0), L.ptr(uv), L.ptr(depth), L.stream()))
    return uv, depth


def project_point_backward(xyz, intr, extr, W, H, uv, depth, dL_duv, dL_ddepth) -> Tuple[Tensor, Tensor, Tensor]:
    xyz = _points(xyz, "xyz", 3)
    P, dev = xyz.shape[0], xyz.device
    dxyz = torch.empty_like(xyz)
    dintr = torch.zeros(4, dtype=torch.float32, device=dev)
    dextr12 = torch.zeros(12, dtype=torch.float32, device=dev)
    L.check(L.lib().splat_project_point_backward(
        L.ci(P), L.ptr(xyz), L.ptr(_intr4(intr)), L.ptr(_extr12(extr)), L.ci(W), L.ci(H), L.ci(0), L.ptr(L.need(depth, "depth")),
        L.ptr(L.need(dL_duv, "dL_duv")), L.ptr(L.need(dL_ddepth, "dL_ddepth")), L.ptr(dxyz), L.ptr(dintr), L.ptr(dextr12), L.stream()))
    dextr = torch.zeros(extr.shape, dtype=torch.float32, device=dev)
    dextr.view(-1)[:12] = dextr12
    return dxyz, dintr, dextr


# ------------------------------------------------------------------ compute_cov3d (compute_cov3d.cu:119-200)
def compute_cov3d_forward(scales, uquats, visible) -> Tensor:
    scales = _points(scales, "scales", 3)
    uquats = _points(uquats, "uquats", 4)
    P = scales.shape[0]
    cov3d = torch.empty(P, 6, dtype=torch.float32, device=scales.device)
    L.check(L.lib().splat_compute_cov3d_forward(L.ci(P), L.ptr(scales), L.ptr(uquats), L.ptr(_vis(visible, P, scales.device)),
                                                L.ptr(cov3d), L.stream()))
    return cov3d


def compute_cov3d_backward(scales, uquats, visible, dL_dcov3Ds) -> Tuple[Tensor, Tensor]:
    scales = _points(scales, "scales", 3)
    uquats = _points(uquats, "uquats", 4)
    P = scales.shape[0]
    ds, dq = torch.empty_like(scales), torch.empty_like(uquats)
    L.check(L.lib().splat_compute_cov3d_backward(L.ci(P), L.ptr(scales), L.ptr(uquats), L.ptr(_vis(visible, P, scales.device)),
                                                 L.ptr(L.need(dL_dcov3Ds, "dL_dcov3Ds")), L.ptr(ds), L.ptr(dq), L.stream()))
    return ds, dq


# ------------------------------------------------------------------ ewa_project (ewa_project.cu:254-345)
def ewa_project_forward(xyz, cov3d, intr, extr, uv, W, H, visible) -> Tuple[Tensor, Tensor, Tensor]:
    xyz = _points(xyz, "xyz", 3)
    P, dev = xyz.shape[0], xyz.device
    conic = torch.empty(P, 3, dtype=torch.float32, device=dev)
    radius = torch.empty(P, dtype=torch.int32, device=dev)
    tiles = torch.empty(P, dtype=torch.int32, device=dev)
    L.check(L.lib().splat_ewa_project_forward(
        L.ci(P), L.ptr(xyz), L.ptr(_points(cov3d, "cov3d", 6)), L.ptr(_intr4(intr)), L.ptr(_extr12(extr)), L.ptr(_points(uv, "uv", 2)),
        L.ci(W), L.ci(H), L.ptr(_vis(visible, P, dev)), L.ci(0), L.ptr(conic), L.ptr(radius), L.ptr(tiles), L.stream()))
    return conic, radius, tiles


def ewa_project_backward(xyz, cov3d, intr, extr, radius, dL_dconic) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    xyz = _points(xyz, "xyz", 3)
    cov3d = _points(cov3d, "cov3d", 6)
    P, dev = xyz.shape[0], xyz.device
    dxyz, dcov = torch.empty_like(xyz), torch.empty_like(cov3d)
    dintr = torch.zeros(4, dtype=torch.float32, device=dev)
    dextr12 = torch.zeros(12, dtype=torch.float32, device=dev)
    L.check(L.lib().splat_ewa_project_backward(
        L.ci(P), L.ptr(xyz), L.ptr(cov3d), L.ptr(_intr4(intr)), L.ptr(_extr12(extr)), L.ci(0), L.ci(0), L.ci(0),
        L.ptr(L.need(radius, "radius", torch.int32)), L.ptr(L.need(dL_dconic, "dL_dconic")), L.ptr(dxyz), L.ptr(dcov), L.ptr(dintr),
        L.ptr(dextr12), L.stream()))
    dextr = torch.zeros(extr.shape, dtype=torch.float32, device=dev)
    dextr.view(-1)[:12] = dextr12
    return dxyz, dcov, dintr, dextr


# ------------------------------------------------------------------ sort helpers (sort_gaussian.cu:72-146)
def compute_gaussian_key(uv, depth, W, H, radius, tiles) -> Tuple[Tensor, Tensor]:
    """``tiles`` is the inclusive cumsum of the per-Gaussian tile counts (sort_gaussian.py:42); two host syncs as in
    the reference (the pair count sizes the outputs)"""
    uv = _points(uv, "uv", 2)
    P, dev = uv.shape[0], uv.device
    cum = L.need(tiles.reshape(-1), "tiles", torch.int32)
    M = int(cum[-1].item()) if P > 0 else 0
    key = torch.zeros(M, dtype=torch.int64, device=dev)
    gidx = torch.zeros(M, dtype=torch.int32, device=dev)
    L.check(L.lib().splat_compute_gaussian_key(L.ci(P), L.ptr(uv), L.ptr(L.need(depth, "depth")),
                                               L.ptr(L.need(radius.reshape(-1), "radius", torch.int32)), L.ptr(cum), L.ci(W), L.ci(H),
                                               L.ptr(key), L.ptr(gidx), L.stream()))
    return key, gidx


def compute_tile_gaussian_range(W, H, tiles, key_sorted) -> Tensor:
    tr = torch.zeros(_tiles(W, H), 2, dtype=torch.int32, device=key_sorted.device)
    ks = L.need(key_sorted, "key_sorted", torch.int64)
    L.check(L.lib().splat_compute_tile_gaussian_range(ctypes.c_int64(ks.numel()), L.ptr(ks), L.ptr(tr), L.stream()))
    return tr


# ------------------------------------------------------------------ compute_sh / compute_sh_free (compute_sh.cu:197-296)
def _sh_fwd(shs, degree, view_dirs, visible, free):
    shs = L.need(shs, "shs")
    P, dev = shs.shape[0], shs.device
    colors = torch.empty(P, 3, dtype=torch.float32, device=dev)
    clamped = None if free else torch.empty(P, 3, dtype=torch.uint8, device=dev)
    L.check(L.lib().splat_compute_sh_forward(L.ci(P), L.ptr(shs), L.ci(degree), L.ptr(_points(view_dirs, "view_dirs", 3)),
                                             L.ptr(_vis(visible, P, dev)), L.ci(1 if free else 0), L.ptr(colors), L.ptr(clamped),
                                             L.stream()))
    return colors, clamped


def _sh_bwd(shs, degree, view_dirs, visible, clamped, dL_dcolor, free):
    shs = L.need(shs, "shs")
    P, dev = shs.shape[0], shs.device
    dirs = _points(view_dirs, "view_dirs", 3)
    dshs = torch.zeros_like(shs)
    ddirs = torch.empty_like(dirs)
    L.check(L.lib().splat_compute_sh_backward(L.ci(P), L.ptr(shs), L.ci(degree), L.ptr(dirs), L.ptr(_vis(visible, P, dev)),
                                              L.ptr(clamped), L.ci(1 if free else 0), L.ptr(L.need(dL_dcolor, "dL_dcolor")),
                                              L.ci(0), L.ptr(dshs), L.ptr(ddirs), L.stream()))
    return dshs, ddirs


def compute_sh_forward(shs, degree, view_dirs, visible) -> Tuple[Tensor, Tensor]:
    colors, clamped = _sh_fwd(shs, degree, view_dirs, visible, False)
    return colors, clamped.view(torch.bool)


def compute_sh_backward(shs, degree, view_dirs, visible, clamped, dL_dcolor) -> Tuple[Tensor, Tensor]:
    return _sh_bwd(shs, degree, view_dirs, visible, L.need(clamped, "clamped", torch.uint8), dL_dcolor, False)


def compute_sh_free_forward(shs, degree, view_dirs, visible) -> Tensor:
    return _sh_fwd(shs, degree, view_dirs, visible, True)[0]


def compute_sh_free_backward(shs, degree, view_dirs, visible, dL_dcolor) -> Tuple[Tensor, Tensor]:
    return _sh_bwd(shs, degree, view_dirs, visible, None, dL_dcolor, True)


# ------------------------------------------------------------------ alpha blending (alpha_blending.cu:251-583 and variants)
def _blend_fwd(uv, conic, opacity, feature, bias, idx_sorted, tile_range, bg, W, H, K, trunc):
    uv = _points(uv, "uv", 2)
    feature = L.need(feature, "feature")
    P, C = feature.shape
    dev = feature.device
    out = torch.empty(C, H, W, dtype=torch.float32, device=dev)
    final_T = torch.empty(H, W, dtype=torch.float32, device=dev)
    ncontrib = torch.empty(H, W, dtype=torch.int32, device=dev)
    gs_idx = torch.empty(H, W, K, dtype=torch.int32, device=dev) if K > 0 else None
    pack = torch.empty(max(P, 1) * L.lib().splat_blend_pack_floats(C), dtype=torch.float32, device=dev)
    L.check(L.lib().splat_alpha_blending_forward(
        L.ci(P), L.ci(C), L.ptr(uv), L.ptr(L.need(conic, "conic")), L.ptr(L.need(opacity, "opacity")), L.ptr(feature),
        L.ptr(None if bias is None else L.need(bias, "opacity_bias")), L.ptr(L.need(idx_sorted, "idx_sorted", torch.int32)),
        L.ptr(L.need(tile_range, "tile_range", torch.int32)), L.cf(bg), L.ptr(None), L.ci(W), L.ci(H), L.ci(K), L.ci(1 if trunc else 0),
        L.ptr(out), L.ptr(final_T), L.ptr(ncontrib), L.ptr(gs_idx), L.ptr(pack), L.stream()))
    return out, final_T, ncontrib, gs_idx


def _blend_bwd(uv, conic, opacity, feature, bias, idx_sorted, tile_range, bg, W, H, final_T, ncontrib, dL_drendered):
    """the reference's backward: zero-initialised outputs, float atomics (here: one per wave, splat and component)"""
    uv = _points(uv, "uv", 2)
    feature = L.need(feature, "feature")
    P, C = feature.shape
    dev = feature.device
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    duv, dabs, dconic, dop, dfeat = z(P, 2), z(P, 2), z(P, 3), z(opacity.shape), z(P, C)
    dbias = z(bias.shape) if bias is not None else None
    if idx_sorted.numel() > 0:
        pack = torch.empty(max(P, 1) * L.lib().splat_blend_pack_floats(C), dtype=torch.float32, device=dev)
        L.check(L.lib().splat_alpha_blending_backward(
            L.ci(P), L.ci(C), L.ptr(uv), L.ptr(L.need(conic, "conic")), L.ptr(L.need(opacity, "opacity")), L.ptr(feature),
            L.ptr(None if bias is None else L.need(bias, "opacity_bias")), L.ptr(L.need(idx_sorted, "idx_sorted", torch.int32)),
            L.ptr(L.need(tile_range, "tile_range", torch.int32)), L.cf(bg), L.ci(W), L.ci(H), L.ptr(L.need(final_T, "final_T")),
            L.ptr(L.need(ncontrib, "ncontrib", torch.int32)), L.ptr(L.need(dL_drendered, "dL_drendered")), L.ptr(duv), L.ptr(dabs),
            L.ptr(dconic), L.ptr(dop), L.ptr(dfeat), L.ptr(dbias), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None),
            L.ptr(pack), L.ci(0), L.ptr(None), L.stream()))
    return duv, dconic, dop, dfeat, dbias, dabs


def alpha_blending_forward(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H):
    return _blend_fwd(uv, conic, opacity, feature, None, idx_sorted, tile_range, bg, W, H, 0, False)[:3]


def alpha_blending_backward(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, final_T, ncontrib, dL_drendered):
    duv, dconic, dop, dfeat, _, dabs = _blend_bwd(uv, conic, opacity, feature, None, idx_sorted, tile_range, bg, W, H, final_T,
                                                  ncontrib, dL_drendered)
    return duv, dconic, dop, dfeat, dabs


def alpha_blending_forward_enhanced(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, K, enable_truncation):
    return _blend_fwd(uv, conic, opacity, feature, None, idx_sorted, tile_range, bg, W, H, int(K), bool(enable_truncation))


def alpha_blending_backward_enhanced(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, final_T, ncontrib, dL_drendered):
    return alpha_blending_backward(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, final_T, ncontrib, dL_drendered)


def alpha_blending_forward_with_bias(uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, W, H):
    return _blend_fwd(uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, W, H, 0, False)[:3]


def alpha_blending_backward_with_bias(uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, W, H, final_T, ncontrib,
                                      dL_drendered):
    duv, dconic, dop, dfeat, dbias, dabs = _blend_bwd(uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, W, H,
                                                      final_T, ncontrib, dL_drendered)
    return duv, dconic, dop, dfeat, dbias, dabs


__all__ = [
    "project_point_forward", "project_point_backward", "compute_cov3d_forward", "compute_cov3d_backward",
    "ewa_project_forward", "ewa_project_backward", "compute_gaussian_key", "compute_tile_gaussian_range",
    "compute_sh_forward", "compute_sh_backward", "alpha_blending_forward", "alpha_blending_backward",
    "alpha_blending_forward_enhanced", "alpha_blending_backward_enhanced", "compute_sh_free_forward",
    "compute_sh_free_backward", "alpha_blending_forward_with_bias", "alpha_blending_backward_with_bias",
]
