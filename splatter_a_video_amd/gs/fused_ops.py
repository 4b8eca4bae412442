"""Fused per-frame operators of the MI355X renderer (extensions of the ``dptr.gs`` surface).

``preprocess_ortho`` does in one pass over the Gaussians what the reference renderer spreads over an eager-torch
orthographic projection, ``gs.compute_cov3d`` and an eager-torch EWA projection
(reference: src/pointrix/renderer/dptr_ortho_enhanced.py:282-310): same outputs as
``project_point_ortho`` -> ``compute_cov3d`` -> ``ewa_project_ortho``, without the visible mask / cov3d round trips.

Gradient sinks: a backward can ADD its parameter gradients straight into caller-owned buffers (for instance the
views of a ``parallel.FlatGradBucket``) instead of returning fresh tensors for autograd to accumulate; autograd then
sees no gradient for those inputs.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from .point_ops import _ComputeSH, _extr12, _intr4, _points


def check_sink(sink: Optional[Dict[str, Tensor]], shapes: Dict[str, Tensor]) -> Optional[Dict[str, Tensor]]:
    """a sink maps input names to contiguous float32 device buffers with the input's number of elements"""
    if not sink:
        return None
    for k, buf in sink.items():
        if k not in shapes:
            raise ValueError(f"grad_sink has no input called {k!r} (inputs: {sorted(shapes)})")
        L.need(buf, f"grad_sink[{k}]")
        if not buf.is_contiguous() or buf.numel() != shapes[k].numel():
            raise ValueError(f"grad_sink[{k}] must be contiguous with {shapes[k].numel()} elements")
    return sink


class _PreprocessOrtho(torch.autograd.Function):
    """one pass per direction; ``intr`` None: the orthographic camera, else the pinhole camera (fx, fy, cx, cy)"""

    @staticmethod
    def forward(ctx, xyz, scales, uquats, extr, W, H, nearest, extent, offset, sink, intr=None):
        xyz = _points(xyz, "xyz", 3)
        scales = _points(scales, "scales", 3)
        uquats = _points(uquats, "uquats", 4)
        extr_c = _extr12(extr)
        intr_c = _intr4(intr) if intr is not None else None
        P = xyz.shape[0]
        if scales.shape[0] != P or uquats.shape[0] != P:
            raise ValueError("xyz, scales and uquats must describe the same number of Gaussians")
        off = _points(offset, "offset", 3) if offset is not None else None
        if off is not None and off.shape[0] != P:
            raise ValueError("offset must be [P, 3]")
        dev = xyz.device
        uv = torch.empty(P, 2, dtype=torch.float32, device=dev)      # kernels write every element
        depth = torch.empty(P, 1, dtype=torch.float32, device=dev)
        conic = torch.empty(P, 3, dtype=torch.float32, device=dev)
        radius = torch.empty(P, dtype=torch.int32, device=dev)
        tiles = torch.empty(P, dtype=torch.int32, device=dev)
        if intr_c is None:
            L.check(L.lib().splat_preprocess_ortho_forward(
                L.ci(P), L.ptr(xyz), L.ptr(off), L.ptr(scales), L.ptr(uquats), L.ptr(extr_c), L.ci(W), L.ci(H),
                L.cf(nearest), L.cf(extent), L.ptr(uv), L.ptr(depth), L.ptr(conic), L.ptr(radius), L.ptr(tiles), L.stream()))
        else:
            L.check(L.lib().splat_preprocess_persp_forward(
                L.ci(P), L.ptr(xyz), L.ptr(off), L.ptr(scales), L.ptr(uquats), L.ptr(intr_c), L.ptr(extr_c), L.ci(W), L.ci(H),
                L.cf(nearest), L.cf(extent), L.ptr(uv), L.ptr(depth), L.ptr(conic), L.ptr(radius), L.ptr(tiles), L.stream()))
        ctx.meta = (int(W), int(H))
        ctx.sink = sink
        ctx.intr = intr_c
        ctx.save_for_backward(xyz, off, scales, uquats, extr_c, depth, radius)
        ctx.mark_non_differentiable(radius, tiles)
        ctx.set_materialize_grads(False)
        return uv, depth, conic, radius, tiles

    @staticmethod
    def backward(ctx, dL_duv, dL_ddepth, dL_dconic, _r, _t):
        W, H = ctx.meta
        xyz, off, scales, uquats, extr_c, depth, radius = ctx.saved_tensors
        P = xyz.shape[0]
        sink = ctx.sink or {}
        need_xyz = ctx.needs_input_grad[0] or (off is not None and ctx.needs_input_grad[8])
        need_s, need_q = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g_uv = L.need(dL_duv, "dL_duv") if dL_duv is not None else None
        g_d = L.need(dL_ddepth, "dL_ddepth") if dL_ddepth is not None else None
        g_c = L.need(dL_dconic, "dL_dconic") if dL_dconic is not None else None
        if g_uv is None and g_d is not None:
            g_uv = torch.zeros(P, 2, dtype=torch.float32, device=xyz.device)

        def out(name, wanted, like, have_grad):
            """(buffer for the kernel, tensor returned to autograd)"""
            if not wanted or not have_grad:
                return None, None
            if name in sink:
                return sink[name], None
            buf = torch.empty_like(like)
            return buf, buf

        persp = ctx.intr is not None       # (the perspective EWA Jacobian carries a position gradient of its own)
        b_xyz, r_xyz = out("xyz", need_xyz, xyz, g_uv is not None or (persp and g_c is not None))
        b_s, r_s = out("scales", need_s, scales, g_c is not None)
        b_q, r_q = out("uquats", need_q, uquats, g_c is not None)
        sinked = [b is not None and r is None for b, r in ((b_xyz, r_xyz), (b_s, r_s), (b_q, r_q))]
        fresh = [r is not None for r in (r_xyz, r_s, r_q)]
        if any(sinked) and any(fresh):
            raise ValueError("grad_sink must cover every input of preprocess_ortho that requires grad (xyz, scales, uquats)")
        if b_xyz is not None or b_s is not None or b_q is not None:
            if persp:
                L.check(L.lib().splat_preprocess_persp_backward(
                    L.ci(P), L.ptr(xyz), L.ptr(off), L.ptr(scales), L.ptr(uquats), L.ptr(ctx.intr), L.ptr(extr_c), L.ci(W),
                    L.ci(H), L.ptr(depth), L.ptr(radius), L.ptr(g_uv), L.ptr(g_d), L.ptr(g_c), L.ci(1 if any(sinked) else 0),
                    L.ptr(b_xyz), L.ptr(b_s), L.ptr(b_q), L.stream()))
            else:
                L.check(L.lib().splat_preprocess_ortho_backward(
                    L.ci(P), L.ptr(xyz), L.ptr(off), L.ptr(scales), L.ptr(uquats), L.ptr(extr_c), L.ci(W), L.ci(H),
                    L.ptr(depth), L.ptr(radius), L.ptr(g_uv), L.ptr(g_d), L.ptr(g_c), L.ci(1 if any(sinked) else 0),
                    L.ptr(b_xyz), L.ptr(b_s), L.ptr(b_q), L.stream()))
        r_off = r_xyz if (off is not None and ctx.needs_input_grad[8]) else None
        return (r_xyz if ctx.needs_input_grad[0] else None), r_s, r_q, None, None, None, None, None, r_off, None, None


def preprocess_ortho(xyz: Tensor, scales: Tensor, uquats: Tensor, extr: Tensor, W: int, H: int, nearest: float = 0.2,
                     extent: float = 1.3, offset: Optional[Tensor] = None,
                     grad_sink: Optional[Dict[str, Tensor]] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(uv[P,2], depth[P,1], conic[P,3], radius[P] i32, tiles[P] i32) of the orthographic camera ``extr``;
    ``offset`` [P,3] is added to ``xyz`` inside the kernel (per-frame displacement).  ``grad_sink`` may hold buffers
    for "xyz", "scales", "uquats"."""
    sink = check_sink(grad_sink, {"xyz": xyz, "scales": scales, "uquats": uquats})
    return _PreprocessOrtho.apply(xyz, scales, uquats, extr, W, H, nearest, extent, offset, sink)


def preprocess_persp(xyz: Tensor, scales: Tensor, uquats: Tensor, intr: Tensor, extr: Tensor, W: int, H: int,
                     nearest: float = 0.2, extent: float = 1.3, offset: Optional[Tensor] = None,
                     grad_sink: Optional[Dict[str, Tensor]] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """``preprocess_ortho`` for the pinhole camera (``intr`` = fx, fy, cx, cy): the first three operators of
    ``gs.rasterization`` / ``DPTRRender.render_iter`` (project_point -> compute_cov3d -> ewa_project; reference
    src/submodules/dptr/dptr/gs/__init__.py:55-77, src/pointrix/renderer/dptr.py:107-147) in one pass per direction, the
    position gradient through the projection and the EWA Jacobian included.  Camera gradients are not produced: when ``intr``
    or ``extr`` require grad use the separate operators."""
    if (isinstance(intr, Tensor) and intr.requires_grad) or (isinstance(extr, Tensor) and extr.requires_grad):
        raise ValueError("preprocess_persp does not differentiate the camera: use project_point / ewa_project")
    sink = check_sink(grad_sink, {"xyz": xyz, "scales": scales, "uquats": uquats})
    return _PreprocessOrtho.apply(xyz, scales, uquats, extr, W, H, nearest, extent, offset, sink, intr)


def compute_sh_into(shs: Tensor, degree: int, view_dirs: Tensor, visible: Optional[Tensor], shs_grad: Tensor) -> Tensor:
    """``compute_sh`` whose backward ADDS dL/dshs into ``shs_grad`` (same shape as ``shs``, e.g. a FlatGradBucket view)
    instead of returning it: no 58 MB temporary and no accumulate pass per frame at 300k Gaussians."""
    if shs.shape[1] != (int(degree) + 1) ** 2:
        raise ValueError("compute_sh_into needs shs with exactly (degree + 1)^2 coefficients per point")
    check_sink({"shs": shs_grad}, {"shs": shs})
    return _ComputeSH.apply(shs, degree, view_dirs, visible, False, shs_grad)
