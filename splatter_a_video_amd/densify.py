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


# ---------------------------------------------------------------------------------------------------------------------
# structure surgery: clone / split / prune of every per-Gaussian tensor together with its Adam moments
# ---------------------------------------------------------------------------------------------------------------------
def _scan(mask: Tensor):
    """(mask as uint8, exclusive prefix index[N] int32, number selected) -- one host synchronisation"""
    mask_u8 = L.need(mask, "mask", torch.uint8)
    N, dev = mask_u8.numel(), mask_u8.device
    lib = L.lib()
    index = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(int(lib.splat_compact_scratch_bytes(N)), 4), dtype=torch.uint8, device=dev)
    L.check(lib.splat_compact_scan(L.ci(N), L.ptr(mask_u8), L.ptr(index), L.ptr(count), L.ptr(scratch), L.stream()))
    return mask_u8, index, int(count.item())


def _append(t: Tensor, mask_u8: Tensor, index: Tensor, n_sel: int, repeat: int, fill: Optional[Tensor] = None,
            zeros: bool = False) -> Tensor:
    """cat(t, rows) where rows = t[mask].repeat(repeat, 1, ..) -- or ``fill`` ([repeat * n_sel, ...], e.g. the children's new
    positions), or zeros (fresh Adam moments, points.py:332-360)"""
    N = t.shape[0]
    if t.element_size() != 4:
        raise ValueError("only 32-bit element types are supported")
    src = t.detach().contiguous()
    out = torch.empty((N + repeat * n_sel,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out[:N].copy_(src)
    if n_sel == 0:
        return out
    tail = out[N:]
    if zeros:
        tail.zero_()
    elif fill is not None:
        tail.copy_(fill.reshape(tail.shape))
    else:
        row_words = src.numel() // N
        L.check(L.lib().splat_gather_rows_repeat(L.ci(N), L.ptr(mask_u8), L.ptr(index), L.ci(n_sel), L.ci(repeat), L.ci(row_words),
                                                 ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(tail.data_ptr()), L.stream()))
    return out


Moments = Dict[str, Tuple[Tensor, Tensor]]


def densify_clone(params: Dict[str, Tensor], moments: Optional[Moments], mask: Tensor):
    """``densify_clone`` of the reference (atlas_gs_optimizer.py:289-304 + points.py:225-249,332-375): the selected
    Gaussians are appended once more; the new rows' Adam moments (``moments[name] = (exp_avg, exp_avg_sq)``) are zero.
    One prefix sum of the mask serves every tensor.  -> (params, moments, number cloned)"""
    mask_u8, index, n = _scan(mask)
    new_p = {k: _append(v, mask_u8, index, n, 1) for k, v in params.items()}
    new_m = None
    if moments is not None:
        new_m = {k: tuple(_append(m, mask_u8, index, n, 1, zeros=True) for m in mv) for k, mv in moments.items()}
    return new_p, new_m, n


def _need_seed(seed, unit_normals) -> int:
    """The Philox counter is (Gaussian index, replica) only: the SAME seed draws the SAME unit normals for a Gaussian index at
    every densification round, so the caller passes a seed that changes from round to round (e.g. base_seed + iteration; every
    data-parallel rank the same one).  No default: a forgotten seed would correlate the children of successive rounds."""
    if unit_normals is not None:
        return 0 if seed is None else int(seed)
    if seed is None:
        raise ValueError("densify split: pass `seed` (a value that differs from one densification round to the next, identical "
                         "on every rank, e.g. base_seed + iteration) or explicit `unit_normals`")
    return int(seed)


def split_children(position: Tensor, scaling_raw: Tensor, rotation_raw: Tensor, mask: Tensor, split_num: int = 2,
                   seed: Optional[int] = None, unit_normals: Optional[Tensor] = None):
    """``new_pos_scale`` (atlas_gs_optimizer.py:255-287): positions [split_num * n, 3] and log-scales of the children of
    the selected Gaussians.  The normal draws come from a counter-based generator keyed by ``seed`` and addressed by
    (Gaussian id, replica): every data-parallel rank that passes the same seed gets the same children, with no generator
    state to keep in step.  ``unit_normals`` replaces the draws (parity tests)."""
    seed = _need_seed(seed, unit_normals)
    mask_u8, index, n = _scan(mask)
    return _split_children(position, scaling_raw, rotation_raw, mask_u8, index, n, split_num, seed, unit_normals) + (n,)


def _split_children(position, scaling_raw, rotation_raw, mask_u8, index, n, split_num, seed, unit_normals):
    N, dev = position.shape[0], position.device
    new_pos = torch.empty(split_num * n, 3, dtype=torch.float32, device=dev)
    new_scl = torch.empty(split_num * n, 3, dtype=torch.float32, device=dev)
    zn = None
    if unit_normals is not None:
        zn = L.need(unit_normals, "unit_normals")
        if zn.numel() != split_num * n * 3:
            raise ValueError("unit_normals must be [split_num * n_selected, 3]")
    L.check(L.lib().splat_densify_split_sample(
        L.ci(N), L.ptr(mask_u8), L.ptr(index), L.ci(n), L.ci(split_num), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
        L.ptr(L.need(position.detach(), "position")), L.ptr(L.need(scaling_raw.detach(), "scaling")),
        L.ptr(L.need(rotation_raw.detach(), "rotation")), L.ptr(zn), L.ptr(new_pos), L.ptr(new_scl), L.stream()))
    return new_pos, new_scl


def densify_split(params: Dict[str, Tensor], moments: Optional[Moments], mask: Tensor, split_num: int = 2,
                  seed: Optional[int] = None, unit_normals: Optional[Tensor] = None, position: str = "position",
                  scaling: str = "scaling", rotation: str = "rotation"):
    """``densify_split`` of the reference (atlas_gs_optimizer.py:306-349): every selected Gaussian is replaced by
    ``split_num`` children (sampled positions, shrunk scales, every other attribute repeated, fresh Adam moments)
    appended at the end, then the parents are removed (parameters and moments compacted with one shared prefix sum).
    ``seed``: required unless ``unit_normals`` is given -- see ``_need_seed``.
    -> (params, moments, valid_points_mask over the extended cloud [for prune_postprocess], number split)"""
    seed = _need_seed(seed, unit_normals)
    mask_u8, index, n = _scan(mask)
    new_pos, new_scl = _split_children(params[position], params[scaling], params[rotation], mask_u8, index, n, split_num, seed,
                                       unit_normals)
    fills = {position: new_pos, scaling: new_scl}
    ext_p = {k: _append(v, mask_u8, index, n, split_num, fill=fills.get(k)) for k, v in params.items()}
    ext_m = None
    if moments is not None:
        ext_m = {k: tuple(_append(m, mask_u8, index, n, split_num, zeros=True) for m in mv) for k, mv in moments.items()}
    valid = torch.cat([~mask.bool(), torch.ones(split_num * n, dtype=torch.bool, device=mask.device)])
    flat = dict(ext_p)
    if ext_m is not None:
        for k, (a, b) in ext_m.items():
            flat["\x00a" + k], flat["\x00b" + k] = a, b
    kept = compact(valid, flat)
    out_p = {k: kept[k] for k in ext_p}
    out_m = None if ext_m is None else {k: (kept["\x00a" + k], kept["\x00b" + k]) for k in ext_m}
    return out_p, out_m, valid, n


def prune_points(params: Dict[str, Tensor], moments: Optional[Moments], valid_points_mask: Tensor):
    """``remove_points`` (points.py:251-277,279-330): keep the rows of ``valid_points_mask`` in every parameter and moment"""
    flat = dict(params)
    if moments is not None:
        for k, (a, b) in moments.items():
            flat["\x00a" + k], flat["\x00b" + k] = a, b
    kept = compact(valid_points_mask, flat)
    out_p = {k: kept[k] for k in params}
    out_m = None if moments is None else {k: (kept["\x00a" + k], kept["\x00b" + k]) for k in moments}
    return out_p, out_m


def spatial_order(uv: Tensor, W: int, H: int) -> Tensor:
    """Permutation (int64 [P]) that puts the Gaussians in Morton (Z-curve) order of their screen positions ``uv`` [P,2]
    (pixels; any frame of the clip serves -- the Gaussians move by a fraction of a tile).  The binning kernels walk the
    Gaussians in index order and scatter their (Gaussian, tile) pairs into per-tile segments; with image neighbours next to
    each other in memory those writes and the later gathers stay inside a few tiles' worth of cache lines (BASELINE
    configs[1]: bin_scatter 17.8 -> 9.4 us, tile_sort 17.1 -> 12.3 us per frame).  Apply it with ``reorder_points`` when the
    per-Gaussian arrays are rebuilt anyway (after initialisation and after every densification).  The reference keeps
    whatever order its point cloud has; rendering results do not depend on it."""
    uv = L.need(uv.detach(), "uv")
    P = uv.shape[0]
    if uv.dim() != 2 or uv.shape[1] != 2:
        raise ValueError("uv must be [P, 2]")
    keys = torch.empty(P, dtype=torch.int32, device=uv.device)
    L.check(L.lib().splat_morton_keys(L.ci(P), L.ptr(uv), L.ci(W), L.ci(H), L.ptr(keys), L.stream()))
    return torch.sort(keys, stable=True).indices


def reorder_points(params: Dict[str, Tensor], moments: Optional[Moments], perm: Tensor):
    """Rows of every parameter (and of the two Adam moments of each) in the order ``perm`` -- the structural counterpart of
    ``prune_points`` / ``densify_clone`` for a pure permutation (``spatial_order``)."""
    out_p = {k: v.index_select(0, perm).contiguous() for k, v in params.items()}
    out_m = None if moments is None else {k: (a.index_select(0, perm).contiguous(), b.index_select(0, perm).contiguous())
                                          for k, (a, b) in moments.items()}
    return out_p, out_m
