"""Densification statistics and structure updates on the device.

Mirrors the state and the methods of the reference's optimizer wrapper
(reference: src/pointrix/optimizer/atlas_gs_optimizer.py -- ``max_radii2D``, ``pos_gradient_accum``, ``denom``;
``accumulate_viewspace_grad`` :414-433, ``update_structure`` :93-121, ``generate_clone_mask`` / ``generate_split_mask``
:199-251, the prune filter :363-375, ``prune_postprocess`` :381-391) and the batch reduction of its renderer
(src/pointrix/renderer/dptr_ortho_enhanced.py:425-431), as a handful of streaming HIP kernels:

* ``accumulate_frame``  after each frame's backward: one pass instead of ``viewspace_grad += x.grad`` / ``cat().any()`` /
  ``cat().max()`` over the batch
* ``update``            once per optimiser step
* ``masks``             clone / split / prune decisions
* ``compact``           one prefix sum of the keep mask, then every per-Gaussian tensor (parameters, Adam moments,
  statistics) is compacted with it -- the reference runs a boolean-index pass with its own ``nonzero`` per tensor

There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L


class DensifyState:
    def __init__(self, num_points: int, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError("DensifyState lives on the GPU")
        self.device = dev
        self.reset(num_points)

    # ---------------------------------------------------------------- state (names as in the reference)
    def reset(self, num_points: int) -> None:
        """reset_densification_state (:393-401) + an empty batch"""
        N, dev = int(num_points), self.device
        self.num_points = N
        self.max_radii2D = torch.zeros(N, dtype=torch.float32, device=dev)
        self.pos_gradient_accum = torch.zeros(N, 1, dtype=torch.float32, device=dev)
        self.denom = torch.zeros(N, 1, dtype=torch.float32, device=dev)
        self.viewspace_grad = torch.zeros(N, 2, dtype=torch.float32, device=dev)
        self.visibility = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.radii = torch.zeros(N, dtype=torch.int32, device=dev)

    def begin_batch(self) -> None:
        self.viewspace_grad.zero_()
        self.visibility.zero_()
        self.radii.zero_()

    # ---------------------------------------------------------------- per frame / per step
    def accumulate_frame(self, radius: Tensor, tap_grad: Optional[Tensor], scale: Tuple[float, float] = (1.0, 1.0)) -> None:
        """``radius`` int32 [N] of the frame; ``tap_grad`` [N,2] = the frame's ndc (or abs_ndc) gradient -- or dL_duv
        with ``scale = (W/2, H/2)``, which is what the tap receives (alpha_blending.py:140-147)."""
        radius = L.need(radius, "radius", torch.int32)
        if radius.numel() != self.num_points:
            raise ValueError("radius must have one entry per Gaussian")
        tap = None
        if tap_grad is not None:
            tap = L.need(tap_grad, "tap_grad")
            if tap.numel() != 2 * self.num_points:
                raise ValueError("tap_grad must be [N, 2]")
        L.check(L.lib().splat_densify_accumulate(
            L.ci(self.num_points), L.ptr(radius), L.ptr(tap), L.cf(scale[0]), L.cf(scale[1]),
            L.ptr(self.viewspace_grad if tap is not None else None), L.ptr(self.visibility), L.ptr(self.radii), L.stream()))

    def update(self) -> None:
        """statistics part of update_structure for the batch accumulated since begin_batch()"""
        L.check(L.lib().splat_densify_update(
            L.ci(self.num_points), L.ptr(self.visibility), L.ptr(self.viewspace_grad), L.ptr(self.radii),
            L.ptr(self.max_radii2D), L.ptr(self.pos_gradient_accum), L.ptr(self.denom), L.stream()))

    # ---------------------------------------------------------------- decisions
    def masks(self, scaling_raw: Tensor, opacity_raw: Tensor, densify_grad_threshold: float, percent_dense: float,
              cameras_extent: float, min_opacity: float, size_threshold: float = 20.0) -> Tuple[Tensor, Tensor, Tensor]:
        """(clone, split, prune) boolean masks; ``scaling_raw`` / ``opacity_raw`` are the raw parameters (exp / sigmoid
        are applied inside, as get_scaling / get_opacity do)."""
        N = self.num_points
        sc = L.need(scaling_raw, "scaling_raw")
        op = L.need(opacity_raw, "opacity_raw")
        if sc.numel() != 3 * N or op.numel() != N:
            raise ValueError("scaling_raw must be [N,3] and opacity_raw [N,1]")
        out = [torch.empty(N, dtype=torch.uint8, device=self.device) for _ in range(3)]
        L.check(L.lib().splat_densify_masks(
            L.ci(N), L.ptr(self.pos_gradient_accum), L.ptr(self.denom), L.ptr(self.max_radii2D), L.ptr(sc), L.ptr(op),
            L.cf(densify_grad_threshold), L.cf(percent_dense), L.cf(cameras_extent), L.cf(min_opacity),
            L.cf(size_threshold), L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.stream()))
        return tuple(o.view(torch.bool) for o in out)

    def prune_postprocess(self, valid_points_mask: Tensor) -> None:
        """keep the statistics of the surviving Gaussians (:381-391)"""
        kept = compact(valid_points_mask, {"a": self.pos_gradient_accum, "d": self.denom, "m": self.max_radii2D})
        self.pos_gradient_accum, self.denom, self.max_radii2D = kept["a"], kept["d"], kept["m"]
        n = self.max_radii2D.shape[0]
        self.num_points = n
        self.viewspace_grad = torch.zeros(n, 2, dtype=torch.float32, device=self.device)
        self.visibility = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.radii = torch.zeros(n, dtype=torch.int32, device=self.device)


def compact(mask: Tensor, tensors: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """``{name: t[mask]}`` for per-Gaussian tensors ``t`` of shape [N, ...] (32-bit element types) with one prefix sum
    of the mask shared by all of them.  One host synchronisation (the number of rows kept, to size the outputs)."""
    mask_u8 = L.need(mask, "mask", torch.uint8)
    N = mask_u8.numel()
    dev = mask_u8.device
    lib = L.lib()
    index = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(int(lib.splat_compact_scratch_bytes(N)), 4), dtype=torch.uint8, device=dev)
    L.check(lib.splat_compact_scan(L.ci(N), L.ptr(mask_u8), L.ptr(index), L.ptr(count), L.ptr(scratch), L.stream()))
    n = int(count.item())
    out = {}
    for name, t in tensors.items():
        if not isinstance(t, Tensor) or not t.is_cuda:
            raise ValueError(f"{name} must be a GPU tensor")
        if t.shape[0] != N:
            raise ValueError(f"{name} must have {N} rows")
        if t.element_size() != 4:
            raise ValueError(f"{name}: only 32-bit element types are supported")
        src = t.detach().contiguous()
        row_words = src.numel() // N if N else 1
        dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
        if N and n:
            L.check(lib.splat_compact_rows(L.ci(N), L.ptr(mask_u8), L.ptr(index), L.ci(row_words),
                                           ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), L.stream()))
        out[name] = dst
    return out
