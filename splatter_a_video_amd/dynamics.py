"""Per-frame evaluation of the dynamic Gaussians on the native path.

Mirrors what the reference's point-cloud class computes per rendered frame
(reference: src/dynamic_gaussian_with_base_point_cloud.py:171-198 ``get_opacity`` / ``get_scaling`` /
``get_rotation(time)``, :236-250 ``get_position(time)``; knot construction :64-66) -- one fused HIP launch
for all four outputs instead of ~25 eager torch kernels, one fused launch for the backward.

* ``FrameClock``      host-side per-frame scalars (segment index, offset inside the segment, time bases).
* ``evaluate(...)``   functional form (autograd-aware), any subset of outputs.
* ``DynamicGaussians`` parameter holder with the reference's attribute and getter names.

There is no CPU fallback: the tensors must live on the GPU and libsplat_hip.so must be built.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib as L

_F12 = ctypes.c_float * 12

# layouts of the spline table (include/splat_hip.h)
GAUSSIAN_MAJOR = 0      # [N, 4*I*3] == [N,4,I,3]: the reference's parameter (:73-75)
SEGMENT_MAJOR = 1       # [I, N, 4, 3]: native layout, one contiguous 48-byte record per Gaussian and frame


def to_segment_major(pos_cubic_node: Tensor, interval_num: int) -> Tensor:
    """reference layout [N, 4*I*3] -> native [I, N, 4, 3] (checkpoint import)"""
    N = pos_cubic_node.shape[0]
    return pos_cubic_node.reshape(N, 4, interval_num, 3).permute(2, 0, 1, 3).contiguous()


def to_gaussian_major(cubic_seg: Tensor) -> Tensor:
    """native [I, N, 4, 3] -> reference layout [N, 4*I*3] (checkpoint export)"""
    I, N = cubic_seg.shape[0], cubic_seg.shape[1]
    return cubic_seg.permute(1, 2, 0, 3).reshape(N, 4 * I * 3).contiguous()


class FrameClock:
    """Time scalars of one clip.  ``intervals`` are the float32 spline knots in [0, 1]; when omitted they are
    built like the reference does (:64-66): one segment per 5 frames, knots at truncated linspace frame indices."""

    def __init__(self, num_frames: int, intervals: Optional[Sequence[float]] = None, start_frame_id: int = 0,
                 time_len: Optional[int] = None):
        if num_frames < 2:
            raise ValueError("a clip needs at least 2 frames")
        self.num_frames = int(num_frames)
        if intervals is None:
            n_seg = math.ceil(num_frames / 5)
            # exactly the reference's construction (float32 torch.linspace, then .long(): :64-66) -- numpy's linspace
            # rounds differently when the step is not representable and can land one frame off
            idx = torch.linspace(0, num_frames - 1, n_seg + 1).long().numpy()
            intervals = idx.astype(np.float32) / np.float32(num_frames - 1)
        self.intervals = np.ascontiguousarray(np.asarray(intervals, dtype=np.float32))
        if self.intervals.ndim != 1 or self.intervals.size < 2:
            raise ValueError("intervals must hold at least two knots")
        self.interval_num = int(self.intervals.size - 1)
        self.start_frame_id = int(start_frame_id)
        self.time_len = int(time_len if time_len is not None else num_frames - 1)
        self._cache: Dict[float, Tuple[int, float, "ctypes.Array"]] = {}

    def scalars(self, time) -> Tuple[int, float, "ctypes.Array"]:
        """(segment, d, basis[12]) for frame ``time`` in the reference's float32 arithmetic: the segment is
        searchsorted(knots, t - 1e-7) - 1 clamped at 0 (:243-245), d = t - knot[segment]; basis = t'^0..3,
        cos(t' k pi), sin(t' k pi), k = 1..4 with t' = (time - start_frame_id) / time_len (:186-193)."""
        if isinstance(time, Tensor):
            time = time.item()
        key = float(time)
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        nt = key / float(self.num_frames - 1)
        seg = int(np.searchsorted(self.intervals, np.float32(nt - 1e-7), side="left")) - 1
        seg = min(max(seg, 0), self.interval_num - 1)
        d = float(np.float32(nt) - self.intervals[seg])
        rt = np.float32((key - float(self.start_frame_id)) / float(self.time_len))
        k = np.arange(4, dtype=np.float32)
        arg = (rt * (k + np.float32(1.0)) * np.float32(np.pi)).astype(np.float32)
        basis = np.concatenate([np.power(rt, k), np.cos(arg), np.sin(arg)]).astype(np.float32)
        out = (seg, d, _F12(*basis.tolist()))
        self._cache[key] = out
        return out


def _opt(t: Optional[Tensor], name: str, rows: int, width: int) -> Optional[Tensor]:
    if t is None:
        return None
    t = L.need(t, name)
    if t.numel() != rows * width:
        raise ValueError(f"{name} must hold {rows} x {width} floats, got shape {tuple(t.shape)}")
    return t


class _DynamicEval(torch.autograd.Function):
    """inputs: position[N,3] cubic[N,4*I*3] rotation[N,4] rot_poly[N,4,4] rot_fourier[N,8,4] opacity[N,1] scaling[N,3]
    outputs: pos_t[N,3] rot_t[N,4] opa_t[N,1] scl_t[N,3] (a zero-size tensor for a group whose inputs are None)."""

    @staticmethod
    def forward(ctx, position, cubic, rotation, rot_poly, rot_fourier, opacity, scaling, seg, d, basis, I, sink, layout):
        ctx.set_materialize_grads(False)             # an unused output arrives as None, not as a zero tensor
        ref = next(t for t in (position, rotation, opacity, scaling) if t is not None)
        N, dev = ref.shape[0], ref.device
        want_pos = position is not None and cubic is not None
        want_rot = rotation is not None
        position = _opt(position, "position", N, 3) if want_pos else None
        cubic = _opt(cubic, "pos_cubic_node", N, 4 * I * 3) if want_pos else None
        rotation = _opt(rotation, "rotation", N, 4)
        rot_poly = _opt(rot_poly, "rot_poly_feat", N, 16) if want_rot else None
        rot_fourier = _opt(rot_fourier, "rot_fourier_feat", N, 32) if want_rot else None
        if want_rot and (rot_poly is None or rot_fourier is None):
            raise ValueError("rotation needs rot_poly_feat [N,4,4] and rot_fourier_feat [N,8,4]")
        opacity = _opt(opacity, "opacity", N, 1)
        scaling = _opt(scaling, "scaling", N, 3)
        new = lambda w, on: torch.empty((N, w) if on else (0,), dtype=torch.float32, device=dev)
        pos_t, rot_t = new(3, want_pos), new(4, want_rot)
        opa_t, scl_t = new(1, opacity is not None), new(3, scaling is not None)
        on = lambda t, flag: L.ptr(t if flag else None)
        L.check(L.lib().splat_dynamic_eval_forward(
            L.ci(N), L.ci(I), L.ci(seg), L.cf(d), basis, L.ptr(position), L.ptr(cubic), L.ci(layout),
            L.ptr(rotation), L.ptr(rot_poly), L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), on(pos_t, want_pos),
            on(rot_t, want_rot), on(opa_t, opacity is not None), on(scl_t, scaling is not None), L.stream()))
        ctx.meta = (N, int(I), int(seg), float(d), basis, want_pos, cubic.shape if want_pos else None, int(layout))
        ctx.sink = sink
        ctx.save_for_backward(rotation, rot_poly, rot_fourier, opacity, scaling)
        ctx.mark_non_differentiable(*[t for t, f in ((pos_t, want_pos), (rot_t, want_rot), (opa_t, opacity is not None),
                                                    (scl_t, scaling is not None)) if not f])
        return pos_t, rot_t, opa_t, scl_t

    @staticmethod
    def backward(ctx, g_pos, g_rot, g_opa, g_scl):
        N, I, seg, d, basis, want_pos, cubic_shape, layout = ctx.meta
        rotation, rot_poly, rot_fourier, opacity, scaling = ctx.saved_tensors
        need = ctx.needs_input_grad
        sink = ctx.sink or {}
        dev = next(t for t in (g_pos, g_rot, g_opa, g_scl) if t is not None).device

        def grad_of(idx, name, g, shape, zero=False):
            """(buffer the kernel writes, tensor handed back to autograd)"""
            if g is None or not need[idx]:
                return None, None
            if name in sink:                        # accumulate straight into the caller's gradient bucket
                return sink[name], None
            buf = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=dev)
            return buf, buf

        g_pos = L.need(g_pos, "g_pos") if (g_pos is not None and want_pos and (need[0] or need[1])) else None
        g_rot = L.need(g_rot, "g_rot") if (g_rot is not None and rotation is not None and need[2]) else None
        g_opa = L.need(g_opa, "g_opa") if (g_opa is not None and opacity is not None and need[5]) else None
        g_scl = L.need(g_scl, "g_scl") if (g_scl is not None and scaling is not None and need[6]) else None
        b_pos, r_pos = grad_of(0, "position", g_pos, (N, 3))
        b_cub, r_cub = grad_of(1, "pos_cubic_node", g_pos, cubic_shape, zero=True)   # dense like the reference's autograd
        b_rot, r_rot = grad_of(2, "rotation", g_rot, (N, 4))
        b_opa, r_opa = grad_of(5, "opacity", g_opa, (N, 1))
        b_scl, r_scl = grad_of(6, "scaling", g_scl, (N, 3))
        acc = 1 if sink else 0
        if sink and any(r is not None for r in (r_pos, r_cub, r_rot, r_opa, r_scl)):
            raise ValueError("grad_sink must cover every parameter that requires grad: "
                             "position, pos_cubic_node, rotation, opacity, scaling")
        L.check(L.lib().splat_dynamic_eval_backward(
            L.ci(N), L.ci(I), L.ci(seg), L.cf(d), basis, L.ptr(rotation), L.ptr(rot_poly), L.ptr(rot_fourier),
            L.ptr(opacity), L.ptr(scaling), L.ptr(g_pos), L.ptr(g_rot), L.ptr(g_opa), L.ptr(g_scl), L.ci(acc),
            L.ci(layout), L.ptr(b_pos), L.ptr(b_cub), L.ptr(b_rot), L.ptr(b_opa), L.ptr(b_scl), L.stream()))
        # rot_poly / rot_fourier: the reference detaches both sums (:195-197) -> no gradient
        return r_pos, r_cub, r_rot, None, None, r_opa, r_scl, None, None, None, None, None, None


def evaluate(clock: FrameClock, time, *, position: Optional[Tensor] = None, pos_cubic_node: Optional[Tensor] = None,
             rotation: Optional[Tensor] = None, rot_poly_feat: Optional[Tensor] = None,
             rot_fourier_feat: Optional[Tensor] = None, opacity: Optional[Tensor] = None,
             scaling: Optional[Tensor] = None, grad_sink: Optional[Dict[str, Tensor]] = None,
             cubic_layout: int = GAUSSIAN_MAJOR):
    """One fused launch: returns (pos_t, rot_t, opa_t, scl_t); an entry is None when its inputs were not given.

    ``grad_sink`` (optional) maps parameter names to float32 buffers of the parameter's shape: the backward then
    ADDS the gradients into those buffers (e.g. views of a ``FlatGradBucket``) and autograd sees no gradient for
    them -- no dense zero-filled spline gradient, no AccumulateGrad kernels.
    ``cubic_layout``: GAUSSIAN_MAJOR (reference, [N,4*I*3]) or SEGMENT_MAJOR (native, [I,N,4,3]) for
    ``pos_cubic_node`` and its gradient.
    """
    if cubic_layout not in (GAUSSIAN_MAJOR, SEGMENT_MAJOR):
        raise ValueError("cubic_layout must be GAUSSIAN_MAJOR or SEGMENT_MAJOR")
    seg, d, basis = clock.scalars(time)
    if grad_sink:
        for k, v in grad_sink.items():
            L.need(v, f"grad_sink[{k}]")
            if not v.is_contiguous():
                raise ValueError(f"grad_sink[{k}] must be contiguous")
    pos_t, rot_t, opa_t, scl_t = _DynamicEval.apply(position, pos_cubic_node, rotation, rot_poly_feat, rot_fourier_feat,
                                                    opacity, scaling, seg, d, basis, clock.interval_num, grad_sink,
                                                    cubic_layout)
    pick = lambda t: t if t.numel() or (t.dim() == 2) else None
    return pick(pos_t), pick(rot_t), pick(opa_t), pick(scl_t)


class _FramePreprocess(torch.autograd.Function):
    """dynamic parameters + time -> (uv, depth, conic, radius, tiles, opacity) of the orthographic camera, one launch
    each way (splat_frame_preprocess_*)."""

    @staticmethod
    def forward(ctx, position, cubic, rotation, rot_poly, rot_fourier, opacity, scaling, extr, seg, d, basis, I, W, H,
                nearest, extent, sink, layout):
        N = position.shape[0]
        position = _opt(position, "position", N, 3)
        cubic = _opt(cubic, "pos_cubic_node", N, 4 * I * 3)
        rotation = _opt(rotation, "rotation", N, 4)
        rot_poly = _opt(rot_poly, "rot_poly_feat", N, 16)
        rot_fourier = _opt(rot_fourier, "rot_fourier_feat", N, 32)
        opacity = _opt(opacity, "opacity", N, 1)
        scaling = _opt(scaling, "scaling", N, 3)
        extr_c = L.need(extr, "extr")
        if extr_c.numel() < 12:
            raise ValueError("extr must hold at least 3x4 floats (row-major [R|T])")
        dev = position.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        uv, depth, conic, opa_t = f32(N, 2), f32(N, 1), f32(N, 3), f32(N, 1)
        radius = torch.empty(N, dtype=torch.int32, device=dev)
        tiles = torch.empty(N, dtype=torch.int32, device=dev)
        L.check(L.lib().splat_frame_preprocess_forward(
            L.ci(N), L.ci(I), L.ci(seg), L.cf(d), basis, L.ptr(position), L.ptr(cubic), L.ci(layout), L.ptr(rotation),
            L.ptr(rot_poly), L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c), L.ci(W), L.ci(H),
            L.cf(nearest), L.cf(extent), L.ptr(uv), L.ptr(depth), L.ptr(conic), L.ptr(radius), L.ptr(tiles),
            L.ptr(opa_t), L.stream()))
        ctx.meta = (N, int(I), int(seg), float(d), basis, int(W), int(H), int(layout), cubic.shape)
        ctx.sink = sink
        ctx.save_for_backward(position, cubic, rotation, rot_poly, rot_fourier, opacity, scaling, extr_c, depth, radius)
        ctx.mark_non_differentiable(radius, tiles)
        ctx.set_materialize_grads(False)
        return uv, depth, conic, radius, tiles, opa_t

    @staticmethod
    def backward(ctx, g_uv, g_depth, g_conic, _r, _t, g_opa):
        N, I, seg, d, basis, W, H, layout, cubic_shape = ctx.meta
        position, cubic, rotation, rot_poly, rot_fourier, opacity, scaling, extr_c, depth, radius = ctx.saved_tensors
        need = ctx.needs_input_grad
        sink = ctx.sink or {}
        dev = position.device
        g_uv = L.need(g_uv, "dL_duv") if g_uv is not None else None
        g_depth = L.need(g_depth, "dL_ddepth") if g_depth is not None else None
        g_conic = L.need(g_conic, "dL_dconic") if g_conic is not None else None
        g_opa = L.need(g_opa, "dL_dopacity") if g_opa is not None else None
        if g_uv is None and g_depth is not None:
            g_uv = torch.zeros(N, 2, dtype=torch.float32, device=dev)

        def out(idx, name, shape, zero=False):
            if not need[idx]:
                return None, None
            if name in sink:
                return sink[name], None
            buf = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=dev)
            return buf, buf

        b_pos, r_pos = out(0, "position", (N, 3))
        b_cub, r_cub = out(1, "pos_cubic_node", cubic_shape, zero=True)   # dense like the reference's autograd
        b_rot, r_rot = out(2, "rotation", (N, 4))
        b_opa, r_opa = out(5, "opacity", (N, 1))
        b_scl, r_scl = out(6, "scaling", (N, 3))
        rets = (r_pos, r_cub, r_rot, r_opa, r_scl)
        bufs = (b_pos, b_cub, b_rot, b_opa, b_scl)
        sinked = any(b is not None and r is None for b, r in zip(bufs, rets))
        if sinked and any(r is not None for r in rets):
            raise ValueError("grad_sink must cover every parameter that requires grad: "
                             "position, pos_cubic_node, rotation, opacity, scaling")
        if any(b is not None for b in bufs):
            L.check(L.lib().splat_frame_preprocess_backward(
                L.ci(N), L.ci(I), L.ci(seg), L.cf(d), basis, L.ptr(position), L.ptr(cubic), L.ci(layout),
                L.ptr(rotation), L.ptr(rot_poly), L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c),
                L.ci(W), L.ci(H), L.ptr(depth), L.ptr(radius), L.ptr(g_uv), L.ptr(g_depth), L.ptr(g_conic), L.ptr(g_opa),
                L.ci(1 if sinked else 0), L.ptr(b_pos), L.ptr(b_cub), L.ptr(b_rot), L.ptr(b_opa), L.ptr(b_scl),
                L.stream()))
        # rot_poly / rot_fourier: detached in the reference (:195-197) -> no gradient
        return (r_pos, r_cub, r_rot, None, None, r_opa, r_scl) + (None,) * 11


class _PositionPolyFourier(torch.autograd.Function):
    @staticmethod
    def forward(ctx, position, pos_poly_feat, pos_fourier_feat, basis):
        position = L.need(position, "position")
        N = position.shape[0]
        poly = _opt(pos_poly_feat, "pos_poly_feat", N, 12)
        four = _opt(pos_fourier_feat, "pos_fourier_feat", N, 24)
        out = torch.empty(N, 3, dtype=torch.float32, device=position.device)
        L.check(L.lib().splat_position_poly_fourier_forward(L.ci(N), basis, L.ptr(position), L.ptr(poly), L.ptr(four), L.ptr(out),
                                                            L.stream()))
        ctx.basis, ctx.shapes = basis, (position.shape, pos_poly_feat.shape, pos_fourier_feat.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        g = L.need(g, "dL_dposition_t")
        N = g.shape[0]
        dev = g.device
        need = ctx.needs_input_grad
        d_pos = torch.empty(ctx.shapes[0], dtype=torch.float32, device=dev) if need[0] else None
        d_poly = torch.empty(ctx.shapes[1], dtype=torch.float32, device=dev) if need[1] else None
        d_four = torch.empty(ctx.shapes[2], dtype=torch.float32, device=dev) if need[2] else None
        L.check(L.lib().splat_position_poly_fourier_backward(L.ci(N), ctx.basis, L.ptr(g), L.ci(0), L.ptr(d_pos), L.ptr(d_poly),
                                                             L.ptr(d_four), L.stream()))
        return d_pos, d_poly, d_four, None


def position_poly_fourier(clock: FrameClock, time, position: Tensor, pos_poly_feat: Tensor, pos_fourier_feat: Tensor,
                          detach_pos: bool = False) -> Tensor:
    """``get_position(time, detach_pos)`` of the reference's polynomial / Fourier point cloud
    (src/dynamic_gaussian_points.py:169-186): position + sum_k pos_poly_feat[:, k] t'^k + sum_m pos_fourier_feat[:, m]
    basis_m(t'); one native launch each way, gradients for all three inputs (none for ``position`` with ``detach_pos``)."""
    _, _, basis = clock.scalars(time)
    return _PositionPolyFourier.apply(position.detach() if detach_pos else position, pos_poly_feat, pos_fourier_feat, basis)


def frame_preprocess(clock: FrameClock, time, extr: Tensor, W: int, H: int, *, position: Tensor, pos_cubic_node: Tensor,
                     rotation: Tensor, rot_poly_feat: Tensor, rot_fourier_feat: Tensor, opacity: Tensor,
                     scaling: Tensor, nearest: float = 0.2, extent: float = 1.3,
                     grad_sink: Optional[Dict[str, Tensor]] = None, cubic_layout: int = GAUSSIAN_MAJOR):
    """Per-frame evaluation of the dynamic Gaussians fused with the orthographic preprocess:
    returns (uv[N,2], depth[N,1], conic[N,3], radius[N] i32, tiles[N] i32, opacity[N,1] = sigmoid) -- what
    ``evaluate`` followed by ``gs.preprocess_ortho`` returns, without the per-frame position / rotation / scale tensors.
    ``grad_sink`` as in ``evaluate``."""
    if cubic_layout not in (GAUSSIAN_MAJOR, SEGMENT_MAJOR):
        raise ValueError("cubic_layout must be GAUSSIAN_MAJOR or SEGMENT_MAJOR")
    seg, d, basis = clock.scalars(time)
    if grad_sink:
        for k, v in grad_sink.items():
            L.need(v, f"grad_sink[{k}]")
            if not v.is_contiguous():
                raise ValueError(f"grad_sink[{k}] must be contiguous")
    return _FramePreprocess.apply(position, pos_cubic_node, rotation, rot_poly_feat, rot_fourier_feat, opacity, scaling,
                                  extr, seg, d, basis, clock.interval_num, W, H, nearest, extent, grad_sink, cubic_layout)


def frame_table(clock: FrameClock, times, device) -> Tensor:
    """device table of per-frame scalars (segment, offset inside it, the 12 time-basis values; 64 bytes per frame) of the frame
    times ``times`` -- the table of the batched entry points (splat_frame_preprocess_forward_batch, splat_dynamic_positions_*)"""
    host = np.zeros((len(times), 16), np.float32)
    for f, t in enumerate(times):
        seg, d, basis = clock.scalars(t)
        host[f, 0] = np.array([seg], np.int32).view(np.float32)[0]
        host[f, 1] = d
        host[f, 2:14] = np.frombuffer(basis, dtype=np.float32, count=12)
    _walk_order(host)
    return _upload(host, device)


def _walk_order(host: np.ndarray) -> None:
    """column 14 of a frame table: (as int32) 1 + the frame the positions' backward visits at that step of its walk -- the
    frames grouped by spline segment (stable), so that a thread flushes a segment's coefficient gradients once"""
    segs = host[:, 0].copy().view(np.int32)
    host[:, 14] = (np.argsort(segs, kind="stable").astype(np.int32) + 1).view(np.float32)


def _upload(host: np.ndarray, device) -> Tensor:
    """a small host table to the device without blocking the host: pinned staging + asynchronous copy on the current stream
    (a pageable ``.to(device)`` is a synchronous copy on the step's critical path)"""
    t = torch.from_numpy(host)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def positions_batch_forward(tab: Tensor, position: Tensor, pos_cubic_node: Tensor, interval_num: int,
                            cubic_layout: int = GAUSSIAN_MAJOR, out: Optional[Tensor] = None) -> Tensor:
    """``get_position(t)`` of every Gaussian for ALL frame times of ``tab`` (``frame_table``) in one launch: [F, N, 3].  The pair
    frames of a training batch -- track_gs = position(ids2) and the node sequence of the ARAP term
    (src/trainer_fragGS.py:486-507,671-675).  Raw operator (no autograd; ``positions_batch`` is the autograd form)."""
    position = L.need(position, "position")
    cubic = L.need(pos_cubic_node, "pos_cubic_node")
    N, F = position.shape[0], tab.shape[0]
    if cubic.numel() != N * 4 * interval_num * 3:
        raise ValueError("pos_cubic_node must hold N * 4 * interval_num * 3 floats")
    if out is None:
        out = torch.empty(F, N, 3, dtype=torch.float32, device=position.device)
    elif tuple(out.shape) != (F, N, 3) or out.dtype != torch.float32 or not out.is_cuda or (N and (out.stride(1), out.stride(2)) != (3, 1)):
        raise ValueError(f"out must be a float32 GPU buffer [F={F}, N={N}, 3] with dense frames (any frame stride)")
    L.check(L.lib().splat_dynamic_positions_batch_forward(
        L.ci(F), L.ci(N), L.ci(interval_num), L.ptr(tab), L.ptr(position), L.ptr(cubic), L.ci(cubic_layout), L.ptr(out),
        ctypes.c_int64(out.stride(0) if F > 1 else N * 3), L.stream()))
    return out


def positions_batch_backward(tab: Tensor, g: Tensor, interval_num: int, cubic_layout: int, d_position: Optional[Tensor],
                             d_pos_cubic_node: Optional[Tensor]) -> None:
    """``g`` [F, N, 3] = dL/dposition(t_f) is ADDED into ``d_position`` [N, 3] and into the coefficient rows of every frame's
    segment of ``d_pos_cubic_node`` (either may be None); one launch, no atomics"""
    F, N = g.shape[0], g.shape[1]
    if not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 3 and g.shape[2] == 3 and (N == 0 or (g.stride(1), g.stride(2)) == (3, 1))):
        g = L.need(g, "g")      # (a strided view with dense frames -- g_pairs[:, 0] -- is read in place)
    if tab.shape[0] != F:
        raise ValueError("one table entry per frame of g")
    L.check(L.lib().splat_dynamic_positions_batch_backward(
        L.ci(F), L.ci(N), L.ci(interval_num), L.ptr(tab), L.ptr(g), ctypes.c_int64(g.stride(0) if F > 1 else N * 3), L.ci(cubic_layout),
        L.ptr(d_position), L.ptr(d_pos_cubic_node), L.stream()))


class _PositionsBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, position, cubic, tab, I, layout, sink):
        out = positions_batch_forward(tab, position, cubic, I, layout)
        ctx.meta, ctx.sink, ctx.shapes = (int(I), int(layout)), sink, (position.shape, cubic.shape)
        ctx.save_for_backward(tab)
        return out

    @staticmethod
    def backward(ctx, g):
        (tab,) = ctx.saved_tensors
        I, layout = ctx.meta
        sink = ctx.sink or {}
        need_p, need_c = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dev = g.device
        b_pos = sink.get("position") if need_p else None
        b_cub = sink.get("pos_cubic_node") if need_c else None
        r_pos = r_cub = None
        if need_p and b_pos is None:
            b_pos = r_pos = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=dev)
        if need_c and b_cub is None:
            b_cub = r_cub = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dev)
        positions_batch_backward(tab, g, I, layout, b_pos, b_cub)
        return r_pos, r_cub, None, None, None, None


def positions_batch(clock: FrameClock, times, position: Tensor, pos_cubic_node: Tensor, cubic_layout: int = GAUSSIAN_MAJOR,
                    grad_sink: Optional[Dict[str, Tensor]] = None, tab: Optional[Tensor] = None) -> Tensor:
    """[F, N, 3]: ``get_position(time)`` (reference: src/dynamic_gaussian_with_base_point_cloud.py:236-250) for every time of
    ``times`` in ONE launch each way; differentiable w.r.t. ``position`` and ``pos_cubic_node`` (``grad_sink`` as in ``evaluate``)."""
    if tab is None:
        tab = frame_table(clock, times, position.device)
    return _PositionsBatch.apply(position, pos_cubic_node, tab, clock.interval_num, int(cubic_layout), grad_sink)


class DynamicGaussians(torch.nn.Module):
    """Parameter holder with the reference's attribute / getter names (position, pos_cubic_node, rotation,
    rot_poly_feat, rot_fourier_feat, opacity, scaling; get_position(time), get_rotation(time), get_opacity,
    get_scaling) plus ``frame(time)``, the fused form the MI355X renderer uses."""

    def __init__(self, clock: FrameClock, position: Tensor, pos_cubic_node: Tensor, rotation: Tensor, opacity: Tensor,
                 scaling: Tensor, rot_poly_feat: Optional[Tensor] = None, rot_fourier_feat: Optional[Tensor] = None,
                 cubic_layout: int = GAUSSIAN_MAJOR):
        """``pos_cubic_node`` is given in the reference layout [N, 4*I*3]; with ``cubic_layout=SEGMENT_MAJOR`` it is
        STORED (and optimised) as [I,N,4,3] and ``reference_pos_cubic_node()`` converts back for checkpoints."""
        super().__init__()
        N = position.shape[0]
        self.clock = clock
        self.cubic_layout = int(cubic_layout)
        P = torch.nn.Parameter
        self.position = P(position.contiguous(), requires_grad=False)            # :90 position is not optimised
        if pos_cubic_node.numel() != N * 4 * clock.interval_num * 3:
            raise ValueError("pos_cubic_node must be [N, 4 * interval_num * 3]")
        pos_cubic_node = pos_cubic_node.reshape(N, -1)
        if self.cubic_layout == SEGMENT_MAJOR:
            pos_cubic_node = to_segment_major(pos_cubic_node, clock.interval_num)
        self.pos_cubic_node = P(pos_cubic_node.contiguous())
        self.rotation = P(rotation.contiguous())
        self.opacity = P(opacity.reshape(N, 1).contiguous())
        self.scaling = P(scaling.contiguous())
        z = lambda k: torch.zeros(N, k, 4, dtype=torch.float32, device=position.device)
        self.rot_poly_feat = P(rot_poly_feat.contiguous() if rot_poly_feat is not None else z(4))
        self.rot_fourier_feat = P(rot_fourier_feat.contiguous() if rot_fourier_feat is not None else z(8))

    def reference_pos_cubic_node(self) -> Tensor:
        """the spline table in the reference's [N, 4*I*3] layout (checkpoint export)"""
        t = self.pos_cubic_node.detach()
        return to_gaussian_major(t) if self.cubic_layout == SEGMENT_MAJOR else t

    def frame(self, time, grad_sink: Optional[Dict[str, Tensor]] = None):
        return evaluate(self.clock, time, position=self.position, pos_cubic_node=self.pos_cubic_node,
                        rotation=self.rotation, rot_poly_feat=self.rot_poly_feat,
                        rot_fourier_feat=self.rot_fourier_feat, opacity=self.opacity, scaling=self.scaling,
                        grad_sink=grad_sink, cubic_layout=self.cubic_layout)

    def preprocess(self, time, extr: Tensor, W: int, H: int, nearest: float = 0.2, extent: float = 1.3,
                   grad_sink: Optional[Dict[str, Tensor]] = None):
        """``frame(time)`` + orthographic projection / cov3d / EWA in one launch"""
        return frame_preprocess(self.clock, time, extr, W, H, position=self.position, pos_cubic_node=self.pos_cubic_node,
                                rotation=self.rotation, rot_poly_feat=self.rot_poly_feat,
                                rot_fourier_feat=self.rot_fourier_feat, opacity=self.opacity, scaling=self.scaling,
                                nearest=nearest, extent=extent, grad_sink=grad_sink, cubic_layout=self.cubic_layout)

    def get_position(self, time, detach_pos: bool = False) -> Tensor:
        return evaluate(self.clock, time, position=self.position, pos_cubic_node=self.pos_cubic_node,
                        cubic_layout=self.cubic_layout)[0]

    def get_rotation(self, time) -> Tensor:
        return evaluate(self.clock, time, rotation=self.rotation, rot_poly_feat=self.rot_poly_feat,
                        rot_fourier_feat=self.rot_fourier_feat)[1]

    @property
    def get_opacity(self) -> Tensor:
        return evaluate(self.clock, 0, opacity=self.opacity)[2]

    @property
    def get_scaling(self) -> Tensor:
        return evaluate(self.clock, 0, scaling=self.scaling)[3]
