"""Frame-batched rendering: F frames of one Gaussian set through ONE set of kernel launches (SURVEY 7 stage 6).

The reference renders the frames of a batch one after the other (``render_batch``,
src/pointrix/renderer/dptr_ortho_enhanced.py:385-433: ~13 native launches and ~80 eager kernels per frame).  Here every
kernel of the per-frame path -- orthographic preprocess, tile binning, per-tile sort, record packing, compositing
forward, compositing backward -- takes the frame as a grid dimension, and the Gaussian-side backward (pair reduce +
preprocess backward) runs once per batch: a batch costs the launches of one frame, the short kernels fill the chip, and
the compositing kernels see F x T tiles in one launch (no half-empty last round of workgroups).  Same images and
gradients as F calls of the per-frame operators (tests/test_gpu_frames.py).

``FrameBatch`` owns the batch's device buffers (allocated once, ~140 MB per 480p frame of 300k Gaussians); ``render`` is
autograd-aware.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib as L
from .gs.fused_ops import check_sink
from .gs.point_ops import _extr12, _points


def _tiles(W: int, H: int) -> int:
    return ((W + 15) // 16) * ((H + 15) // 16)


class FrameBatch:
    """Buffers and launch sequence of a batch of ``F`` frames of ``P`` Gaussians at ``W`` x ``H`` with ``C`` composited
    feature channels (C <= 32).  ``capacity`` = tile-Gaussian pairs reserved per frame; None: measured on the first call
    (one host sync), afterwards the batch runs without any host synchronisation and ``check()`` (call it whenever the
    host synchronises anyway, e.g. once per optimiser step) raises if a frame outgrew it."""

    def __init__(self, F: int, P: int, W: int, H: int, C: int, device, capacity: Optional[int] = None,
                 want_abs: bool = False, slack: float = 1.25):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError("FrameBatch lives on the GPU (there is no CPU path)")
        if not (1 <= C <= 32):
            raise ValueError("1 <= C <= 32 feature channels per batch")
        self.F, self.P, self.W, self.H, self.C = int(F), int(P), int(W), int(H), int(C)
        self.T = _tiles(W, H)
        self.dev, self.want_abs, self.slack = dev, bool(want_abs), float(slack)
        lib = L.lib()
        f32, i32 = torch.float32, torch.int32
        F_, P_, T_ = self.F, self.P, self.T
        self.uv = torch.empty(F_, P_, 2, dtype=f32, device=dev)
        self.depth = torch.empty(F_, P_, 1, dtype=f32, device=dev)
        self.conic = torch.empty(F_, P_, 3, dtype=f32, device=dev)
        self.radius = torch.empty(F_, P_, dtype=i32, device=dev)
        self.tile_range = torch.empty(F_, T_, 2, dtype=i32, device=dev)
        self.pairs = torch.zeros(F_, dtype=i32, device=dev)          # M of every frame (device)
        self.overflow = torch.zeros(1, dtype=i32, device=dev)
        self.bin_bytes = int(lib.splat_bin_scratch_bytes(P_, W, H))
        self.bin_scratch = torch.empty(F_ * self.bin_bytes, dtype=torch.uint8, device=dev)
        self.goff = torch.empty(F_, P_, dtype=i32, device=dev)
        self.pack = torch.empty(F_ * P_ * int(lib.splat_blend_pack_floats(C)), dtype=f32, device=dev)
        self.out = None
        self.final_T = torch.empty(F_, H, W, dtype=f32, device=dev)
        self.ncontrib = torch.empty(F_, H, W, dtype=i32, device=dev)
        self.ncp = int(lib.splat_blend_pair_stride(C, 1 if want_abs else 0, 0))
        self.tap = torch.zeros(P_, 2, dtype=f32, device=dev)         # densification taps of the last backward
        self.abs_tap = torch.zeros(P_, 2, dtype=f32, device=dev) if want_abs else None
        self.radii_max = torch.zeros(P_, dtype=i32, device=dev)
        self.capacity = None
        self.keys = self.owner = self.idx_sorted = self.slot_sorted = self.pair_records = None
        if capacity is not None:
            self._reserve(int(capacity))

    # ------------------------------------------------------------------ buffers that depend on the pair capacity
    def _reserve(self, capacity: int) -> None:
        dev, F_ = self.dev, self.F
        self.capacity = int(capacity)
        self.keys = torch.empty(F_, capacity, dtype=torch.int64, device=dev)
        self.owner = torch.empty(F_, capacity, dtype=torch.int32, device=dev)
        self.idx_sorted = torch.empty(F_, capacity, dtype=torch.int32, device=dev)
        self.slot_sorted = torch.empty(F_, capacity, dtype=torch.int32, device=dev)
        self.pair_records = torch.empty(F_ * capacity * self.ncp, dtype=torch.float32, device=dev)

    def check(self) -> int:
        """host sync; raises when a frame's pairs exceeded the capacity (its surplus pairs were dropped), else returns the
        largest per-frame pair count"""
        m = int(self.pairs.max().item())
        if self.capacity is not None and m > self.capacity:
            raise L.SplatError(f"FrameBatch: {m} tile-Gaussian pairs in one frame exceed the capacity {self.capacity}")
        return m

    def memory_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in vars(self).values() if isinstance(t, Tensor))

    # ------------------------------------------------------------------ forward / backward launch sequences
    def _forward(self, xyz, scales, uquats, opacity, feature, offsets, extr, bg, nearest, extent):
        lib, st = L.lib(), L.stream()
        F_, P_, W, H, C = self.F, self.P, self.W, self.H, self.C
        L.check(lib.splat_preprocess_ortho_forward_batch(
            L.ci(F_), L.ci(P_), L.ptr(xyz), L.ptr(offsets), L.ptr(scales), L.ptr(uquats), L.ptr(extr), L.ci(W), L.ci(H),
            L.cf(nearest), L.cf(extent), L.ptr(self.uv), L.ptr(self.depth), L.ptr(self.conic), L.ptr(self.radius), st))
        L.check(lib.splat_bin_count_batch(L.ci(F_), L.ci(P_), L.ptr(self.uv), L.ptr(self.radius), L.ci(W), L.ci(H),
                                          L.ptr(self.bin_scratch), L.ptr(self.tile_range), L.ptr(self.pairs), st))
        if self.capacity is None:      # first batch: size the pair buffers (the only host sync of the object's life)
            self._reserve(int(int(self.pairs.max().item()) * self.slack) + 1024)
        cap = self.capacity
        L.check(lib.splat_bin_sort_batch(
            L.ci(F_), L.ci(P_), L.ptr(self.uv), L.ptr(self.depth), L.ptr(self.radius), L.ci(W), L.ci(H),
            L.ptr(self.bin_scratch), L.ptr(self.tile_range), ctypes.c_int64(cap), L.ptr(self.keys), L.ptr(self.idx_sorted),
            L.ptr(self.overflow), L.ptr(self.goff), L.ptr(self.owner), L.ptr(self.slot_sorted), st))
        out = torch.empty(F_, C, H, W, dtype=torch.float32, device=self.dev)
        op_fs = 0 if opacity.numel() == P_ else P_
        ft_fs = 0 if feature.numel() == P_ * C else P_ * C
        L.check(lib.splat_alpha_blending_forward_batch(
            L.ci(F_), L.ci(P_), L.ci(C), L.ptr(self.uv), L.ptr(self.conic), L.ptr(opacity), ctypes.c_int64(op_fs),
            L.ptr(feature), ctypes.c_int64(ft_fs), L.ptr(self.idx_sorted), L.ptr(self.tile_range), ctypes.c_int64(cap),
            L.cf(bg), L.ptr(None), L.ci(W), L.ci(H), L.ci(0), L.ci(0), L.ptr(out), L.ptr(self.final_T),
            L.ptr(self.ncontrib), L.ptr(None), L.ptr(self.pack), st))
        return out

    def _backward(self, dL_dout, xyz, scales, uquats, extr, bg, bufs, accumulate, dbg=None):
        lib, st = L.lib(), L.stream()
        F_, P_, W, H, C = self.F, self.P, self.W, self.H, self.C
        cap = self.capacity
        L.check(lib.splat_alpha_blending_backward_batch(
            L.ci(F_), L.ci(P_), L.ci(C), L.ptr(self.idx_sorted), L.ptr(self.tile_range), ctypes.c_int64(cap), L.cf(bg),
            L.ci(W), L.ci(H), L.ptr(self.final_T), L.ptr(self.ncontrib), L.ptr(dL_dout), L.ci(1 if self.want_abs else 0),
            L.ptr(self.slot_sorted), L.ptr(self.pair_records), L.ptr(self.pack), L.ptr(dbg), st))
        L.check(lib.splat_frames_gauss_backward_static(
            L.ci(F_), L.ci(P_), L.ci(C), L.ci(W), L.ci(H), ctypes.c_int64(cap), L.ci(1 if self.want_abs else 0),
            L.ptr(self.pair_records), L.ptr(self.goff), L.ptr(self.radius), L.ptr(xyz), L.ptr(scales), L.ptr(uquats),
            L.ptr(extr), L.ci(1 if accumulate else 0), L.ptr(bufs["xyz"]), L.ptr(bufs["scales"]), L.ptr(bufs["uquats"]),
            L.ptr(bufs["opacity"]), L.ptr(bufs["feature"]), L.ptr(self.tap), L.ptr(self.abs_tap), L.ptr(self.radii_max), st))

    # ------------------------------------------------------------------ public
    def render(self, xyz: Tensor, scales: Tensor, uquats: Tensor, opacity: Tensor, feature: Tensor, offsets: Optional[Tensor],
               extr: Tensor, bg: float = 0.0, nearest: float = 0.01, extent: float = 1.3,
               grad_sink: Optional[Dict[str, Tensor]] = None) -> Tensor:
        """images [F,C,H,W] of the F frames ``xyz + offsets[f]`` (static scale / rotation / opacity / feature [P,C]) under
        the orthographic camera ``extr``.  Differentiable w.r.t. xyz, scales, uquats, opacity, feature; the backward ADDS
        the gradients of the names found in ``grad_sink`` into those buffers instead (e.g. FlatGradBucket views; autograd
        then sees no gradient for them), the others are returned to autograd.  After
        the backward, ``tap`` / ``abs_tap`` hold the batch's summed densification taps and ``radii_max`` the largest
        screen radius of every Gaussian over the frames."""
        sink = check_sink(grad_sink, {"xyz": xyz, "scales": scales, "uquats": uquats, "opacity": opacity, "feature": feature})
        return _RenderFrames.apply(xyz, scales, uquats, opacity, feature, offsets, extr, self, float(bg), float(nearest),
                                   float(extent), sink)


class _RenderFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scales, uquats, opacity, feature, offsets, extr, fb: FrameBatch, bg, nearest, extent, sink):
        xyz = _points(xyz, "xyz", 3)
        scales = _points(scales, "scales", 3)
        uquats = _points(uquats, "uquats", 4)
        opacity = L.need(opacity, "opacity")
        feature = _points(feature, "feature", fb.C)
        extr_c = _extr12(extr)
        P, F = fb.P, fb.F
        if xyz.shape[0] != P or scales.shape[0] != P or uquats.shape[0] != P or opacity.numel() != P or feature.shape[0] != P:
            raise ValueError(f"the batch was built for {P} Gaussians")
        off = None
        if offsets is not None:
            off = L.need(offsets, "offsets")
            if tuple(off.shape) != (F, P, 3):
                raise ValueError(f"offsets must be [F={F}, P={P}, 3]")
        elif F > 1:
            raise ValueError("several frames of static Gaussians need per-frame offsets")
        out = fb._forward(xyz, scales, uquats, opacity, feature, off, extr_c, bg, nearest, extent)
        ctx.fb, ctx.bg, ctx.sink = fb, bg, sink
        ctx.save_for_backward(xyz, scales, uquats, opacity, feature, extr_c)
        return out

    @staticmethod
    def backward(ctx, dL_dout):
        fb: FrameBatch = ctx.fb
        xyz, scales, uquats, opacity, feature, extr_c = ctx.saved_tensors
        g = L.need(dL_dout, "dL_dout")
        sink = ctx.sink or {}
        like = {"xyz": xyz, "scales": scales, "uquats": uquats, "opacity": opacity, "feature": feature}
        # one accumulate flag for the launch: with a sink in play the buffers autograd receives start from zero
        fresh = torch.zeros_like if sink else torch.empty_like
        bufs = {k: (sink[k] if k in sink else fresh(v)) for k, v in like.items()}
        ret = tuple(None if k in sink else bufs[k] for k in ("xyz", "scales", "uquats", "opacity", "feature"))
        from .gs.raster_ops import _debug_T_front
        dbg = _debug_T_front(fb.F * fb.H, fb.W, g.device)
        fb._backward(g, xyz, scales, uquats, extr_c, ctx.bg, bufs, accumulate=bool(sink), dbg=dbg)
        return ret + (None,) * 7
