"""Tile operators of the ``dptr.gs`` surface: sort_gaussian, alpha_blending,
alpha_blending_enhanced, alpha_blending_with_bias and the rasterization() convenience chain.

Signatures / defaults / autograd contract follow the reference
(reference: src/submodules/dptr/dptr/gs/sort_gaussian.py:8-54, alpha_blending.py:7-147,
alpha_blending_enhanced.py:7-160, alpha_blending_with_bias.py, __init__.py:28-100).
"""
from __future__ import annotations

import ctypes
import os
import warnings
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from .point_ops import compute_cov3d, ewa_project, project_point


def _num_tiles(W: int, H: int) -> int:
    return ((W + 15) // 16) * ((H + 15) // 16)


_HALF_WH = {}


def _half_wh(W: int, H: int, device) -> Tensor:
    """[W/2, H/2] on the device, created once per (W, H, device): building it in every backward (as the
    reference does, alpha_blending.py:113-115) is a blocking host-to-device copy on the critical path."""
    key = (int(W), int(H), str(device))
    t = _HALF_WH.get(key)
    if t is None:
        t = torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32, device=device)
        _HALF_WH[key] = t
    return t


# ------------------------------------------------------------------ pair map: the sort's companion for the atomic-free backward
class PairMap:
    """What ``sort_gaussian`` knows beyond the reference's two outputs: ``goff`` (inclusive prefix of tiles per Gaussian)
    and ``slot_sorted`` (Gaussian-major pair slot of every sorted entry).  ``alpha_blending*``'s backward uses it to run
    without global atomics.  The map is bound to the exact ``idx_sorted`` / ``tile_range`` tensors it was made with
    (storage address, size and autograd version counter): ``match`` raises when either was modified in place since."""

    def __init__(self, goff: Tensor, slot_sorted: Tensor, idx_sorted: Tensor, tile_range: Tensor):
        self.goff, self.slot_sorted = goff, slot_sorted
        self.P, self.M = int(goff.numel()), int(idx_sorted.numel())
        self._idx = (idx_sorted.data_ptr(), idx_sorted._version)
        self._tr = (tile_range.data_ptr(), tile_range.numel(), tile_range._version)

    def match(self, idx_sorted: Tensor, tile_range: Tensor, P: int) -> None:
        if idx_sorted.data_ptr() != self._idx[0] or idx_sorted.numel() != self.M:
            raise L.SplatError("pair map belongs to a different idx_sorted tensor")
        if idx_sorted._version != self._idx[1]:
            raise L.SplatError("idx_sorted was modified in place after sort_gaussian: its pair map is stale")
        if (tile_range.data_ptr(), tile_range.numel()) != self._tr[:2]:
            raise L.SplatError("alpha blending got idx_sorted and tile_range of two different sort_gaussian calls")
        if tile_range._version != self._tr[2]:
            raise L.SplatError("tile_range was modified in place after sort_gaussian: the pair map is stale")
        if P != self.P:
            raise L.SplatError(f"the sort covered {self.P} Gaussians, the blend is called with {P}")


_PAIRMAPS = {}     # idx_sorted storage address -> PairMap (entry dropped when the tensor dies)
_warned_foreign = False


def _register_pairmap(idx_sorted: Tensor, pm: PairMap) -> None:
    key = idx_sorted.data_ptr()
    _PAIRMAPS[key] = pm
    weakref.finalize(idx_sorted, lambda k=key, m=pm: _PAIRMAPS.pop(k, None) if _PAIRMAPS.get(k) is m else None)


def _find_pairmap(idx_sorted: Tensor, tile_range: Tensor, P: int, given: Optional[PairMap] = None) -> Optional[PairMap]:
    """the pair map of this ``idx_sorted`` (validated), or None for an index list this package did not produce (a
    foreign or copied tensor: the backward then takes the wave-reduced atomic kernel; warned about once)"""
    global _warned_foreign
    pm = given if given is not None else _PAIRMAPS.get(idx_sorted.data_ptr())
    if pm is None:
        if idx_sorted.numel() > 0 and not _warned_foreign:
            _warned_foreign = True
            warnings.warn("alpha_blending: idx_sorted does not come from this package's sort_gaussian (or is a copy of its "
                          "result): the backward falls back to the atomic kernel", RuntimeWarning, stacklevel=3)
        return None
    pm.match(idx_sorted, tile_range, P)
    return pm


# ------------------------------------------------------------------ sort_gaussian
_OVERFLOW_SINK = {}


def _overflow_sink(device) -> Tensor:
    t = _OVERFLOW_SINK.get(str(device))
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _OVERFLOW_SINK[str(device)] = t
    return t


class SortStatus:
    """Device-side outcome of a capacity-bounded sort: ``pairs`` (int32[1], the true number of tile-Gaussian
    pairs M) and ``capacity``; ``overflow`` (bool[1], M exceeded the capacity and pairs were dropped) derives from them."""

    def __init__(self, pairs: Tensor, capacity: int, pairmap: Optional[PairMap] = None):
        self.pairs, self.capacity, self.pairmap = pairs, int(capacity), pairmap

    @property
    def overflow(self) -> Tensor:
        return self.pairs > self.capacity

    def check(self) -> int:
        """host sync; raises when the capacity was too small, else returns M"""
        m = int(self.pairs.item())
        if m > self.capacity:
            raise L.SplatError(f"sort_gaussian_capped: {m} tile-Gaussian pairs exceed the capacity {self.capacity}")
        return m


def _sort(uv: Tensor, depth: Tensor, W: int, H: int, radius: Tensor, tiles: Optional[Tensor], capacity: Optional[int],
          conic: Optional[Tensor] = None, opacity: Optional[Tensor] = None):
    """``conic`` + ``opacity``: reach masks (splat_bin_*_batch_reach with F = 1) -- only the pairs whose tile the splat can reach
    with alpha >= 1/255 are created; not for the reference-shaped ``sort_gaussian``, whose lists are its results"""
    uv = L.need(uv, "uv")
    depth = L.need(depth, "depth")
    radius = L.need(radius, "radius", torch.int32)
    if tiles is not None and tiles.shape[0] != uv.shape[0]:
        raise ValueError("tiles must have P elements")
    P = uv.shape[0]
    if depth.numel() != P or radius.numel() != P:
        raise ValueError("uv, depth, radius must agree on P")
    dev = uv.device
    T = _num_tiles(W, H)
    tile_range = torch.empty(T, 2, dtype=torch.int32, device=dev)
    if P == 0:
        z = torch.zeros(1, dtype=torch.int32, device=dev)
        return torch.empty(0, dtype=torch.int32, device=dev), tile_range.zero_(), SortStatus(z, 0)
    lib = L.lib()
    scratch = torch.empty(lib.splat_bin_scratch_bytes(P, W, H), dtype=torch.uint8, device=dev)
    m_dev = torch.empty(1, dtype=torch.int32, device=dev)
    reach = None
    if conic is not None:
        conic, opacity = L.need(conic, "conic"), L.need(opacity, "opacity")
        if conic.numel() != 3 * P or opacity.numel() != P:
            raise ValueError("conic [P,3] and opacity [P] must agree with uv on P")
        reach = torch.empty(P, dtype=torch.int32, device=dev)
        L.check(lib.splat_bin_count_batch_reach(L.ci(1), L.ci(P), L.ptr(uv), L.ptr(radius), L.ptr(conic), L.ptr(opacity),
                                                ctypes.c_int64(0), L.ci(W), L.ci(H), L.ptr(scratch), L.ptr(tile_range), L.ptr(m_dev),
                                                L.ptr(None), L.ptr(reach), L.stream()))
    else:
        L.check(lib.splat_bin_count(L.ci(P), L.ptr(uv), L.ptr(radius), L.ci(W), L.ci(H), L.ptr(scratch),
                                    L.ptr(tile_range), L.ptr(m_dev), L.ptr(None), L.stream()))
    M = int(m_dev.item()) if capacity is None else int(capacity)   # the only host sync (none with a capacity)
    idx_sorted = torch.empty(M, dtype=torch.int32, device=dev)
    overflow = _overflow_sink(dev)   # the kernels only ever store 1 here; pairs > capacity carries the same fact
    pm = None
    if M > 0:
        keys = torch.empty(M, dtype=torch.int64, device=dev)
        owner = torch.empty(M, dtype=torch.int32, device=dev)
        slot_sorted = torch.empty(M, dtype=torch.int32, device=dev)
        goff = torch.empty(P, dtype=torch.int32, device=dev)   # inclusive tiles-per-Gaussian prefix, written by the sort
        if reach is not None:
            L.check(lib.splat_bin_sort_batch_reach(L.ci(1), L.ci(P), L.ptr(uv), L.ptr(depth), L.ptr(radius), L.ptr(reach),
                                                   L.ci(W), L.ci(H), L.ptr(scratch),
                                                   L.ptr(tile_range), ctypes.c_int64(M), L.ptr(keys), L.ptr(idx_sorted),
                                                   L.ptr(overflow), L.ptr(goff), L.ptr(owner), L.ptr(slot_sorted), L.stream()))
        else:
            L.check(lib.splat_bin_sort(L.ci(P), L.ptr(uv), L.ptr(depth), L.ptr(radius), L.ci(W), L.ci(H), L.ptr(scratch),
                                       L.ptr(tile_range), ctypes.c_int64(M), L.ptr(keys), L.ptr(idx_sorted),
                                       L.ptr(overflow), L.ptr(goff), L.ptr(owner), L.ptr(slot_sorted), L.stream()))
        # companion of idx_sorted (the reference's signature has no room for it): lets alpha_blending's backward run
        # without global atomics; found again through the tensor's storage address and validated there
        pm = PairMap(goff, slot_sorted, idx_sorted, tile_range)
        _register_pairmap(idx_sorted, pm)
    return idx_sorted, tile_range, SortStatus(m_dev, M, pm)


def sort_gaussian(uv: Tensor, depth: Tensor, W: int, H: int, radius: Tensor, tiles: Tensor) -> Tuple[Tensor, Tensor]:
    """(idx_sorted[M] int32, tile_range[T,2] int32): Gaussian ids ordered by (tile, depth) and each
    tile's [start, end) slice.  Ties (same tile, bit-equal depth) are ordered by ascending id, i.e.
    the result of a STABLE sort of the reference's 64-bit keys (the reference's torch.sort is
    unstable).  ``tiles`` is accepted for signature parity; the per-tile counts are re-derived from
    uv/radius on the device.  One host sync (to size idx_sorted; the reference needs two)."""
    idx_sorted, tile_range, _ = _sort(uv, depth, W, H, radius, tiles, None)
    return idx_sorted, tile_range


def sort_gaussian_capped(uv: Tensor, depth: Tensor, W: int, H: int, radius: Tensor, capacity: Optional[int],
                         conic: Optional[Tensor] = None, opacity: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, SortStatus]:
    """``sort_gaussian`` without the host synchronisation: ``idx_sorted`` is allocated with ``capacity`` entries (only
    the first M are meaningful, tile_range never points past them) and the caller checks ``status`` whenever it next
    synchronises anyway (e.g. once per gradient step).  On overflow the surplus pairs are dropped and flagged
    (``status.check()`` raises); every range, slot and prefix the sort leaves behind is clamped to the capacity, so
    blending such a result is memory-safe (and meaningless).  ``capacity`` None: sized exactly, with the one host sync.
    With ``conic`` [P,3] and ``opacity`` [P] the lists hold only the (Gaussian, tile) pairs whose tile the splat can reach with
    alpha >= 1/255 (reach masks, include/splat_hip.h): a third fewer entries on the bench scene for every kernel behind the sort;
    ``alpha_blending*`` composites the same images, ids and gradients from them (list positions -- ncontrib -- differ from the
    reference's, which is why ``sort_gaussian`` itself never drops a pair)."""
    if capacity is not None and capacity < 0:
        raise ValueError("capacity must be >= 0")
    if (conic is None) != (opacity is None):
        raise ValueError("conic and opacity go together")
    return _sort(uv, depth, W, H, radius, None, capacity, conic, opacity)


# ------------------------------------------------------------------ debug: does the backward replay the forward's decisions?
_T_FRONT_CAPTURE = None


class capture_T_front:
    """``with capture_T_front() as cap: loss.backward()`` -- every blend backward inside stores, per pixel, the
    transmittance its back-to-front replay (``T /= 1 - alpha``, src/alpha_blending.cu:196-214) arrives at in front of
    the first splat into ``cap.maps`` ([H,W] tensors).  It equals 1 up to rounding iff the backward made exactly the
    forward's inclusion decisions for that pixel; one flipped decision moves it by a factor >= 1/(1 - 1/255)."""

    def __init__(self):
        self.maps = []

    def __enter__(self):
        global _T_FRONT_CAPTURE
        _T_FRONT_CAPTURE = self
        return self

    def __exit__(self, *exc):
        global _T_FRONT_CAPTURE
        _T_FRONT_CAPTURE = None
        return False


def _debug_T_front(H: int, W: int, device) -> Optional[Tensor]:
    if _T_FRONT_CAPTURE is None:
        return None
    t = torch.full((H, W), float("nan"), dtype=torch.float32, device=device)
    _T_FRONT_CAPTURE.maps.append(t)
    return t


# ------------------------------------------------------------------ alpha blending (3 variants, one Function)
class _AlphaBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, conic, opacity, feature, bias, idx_sorted, tile_range, bg, W, H, ndc, abs_ndc, K, trunc):
        uv = L.need(uv, "uv")
        conic = L.need(conic, "conic")
        opacity = L.need(opacity, "opacity")
        feature = L.need(feature, "feature")
        idx_sorted = L.need(idx_sorted, "idx_sorted", torch.int32)
        tile_range = L.need(tile_range, "tile_range", torch.int32)
        bias_c = None if bias is None else L.need(bias, "opacity_bias")
        if feature.dim() != 2:
            raise ValueError("feature must have shape [P, C]")
        P, C = feature.shape
        if C < 1:
            raise ValueError("feature needs at least one channel")
        if uv.shape[0] != P or conic.shape[0] != P or opacity.numel() != P:
            raise ValueError("uv / conic / opacity / feature must agree on P")
        if tile_range.numel() != 2 * _num_tiles(W, H):
            raise ValueError("tile_range must have shape [ceil(W/16)*ceil(H/16), 2]")
        dev = feature.device
        out = torch.empty(C, H, W, dtype=torch.float32, device=dev)
        final_T = torch.empty(H, W, dtype=torch.float32, device=dev)
        ncontrib = torch.empty(H, W, dtype=torch.int32, device=dev)
        gs_idx = torch.empty(H, W, K, dtype=torch.int32, device=dev) if K > 0 else None  # kernel pads with -1
        pack = torch.empty(max(P, 1) * L.lib().splat_blend_pack_floats(C), dtype=torch.float32, device=dev)
        ctx.pairmap = _find_pairmap(idx_sorted, tile_range, P)
        # the cull's keep words of every sorted entry: with them the pair-mode backward walks quarter lists instead of
        # culling again (kept only when a backward can follow)
        flags = None
        if ctx.pairmap is not None and idx_sorted.numel() > 0 and any(ctx.needs_input_grad[:5]):
            flags = torch.empty(idx_sorted.numel(), dtype=torch.int32, device=dev)
        L.check(L.lib().splat_alpha_blending_forward_flags(
            L.ci(P), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opacity), L.ptr(feature), L.ptr(bias_c),
            L.ptr(idx_sorted), L.ptr(tile_range), L.cf(bg), L.ptr(None), L.ci(W), L.ci(H), L.ci(K),
            L.ci(1 if trunc else 0), L.ptr(out), L.ptr(final_T), L.ptr(ncontrib), L.ptr(gs_idx), L.ptr(pack), L.ptr(flags),
            L.stream()))
        ctx.flags = flags
        ctx.meta = (float(bg), int(W), int(H), bias is not None, ndc is not None, abs_ndc is not None)
        ctx.pack = pack          # packed records: reused by the backward when C <= 32 (one channel chunk)
        saved = [uv, conic, opacity, feature, idx_sorted, tile_range, final_T, ncontrib]
        if bias_c is not None:
            saved.append(bias_c)
        ctx.save_for_backward(*saved)
        if K > 0:
            ctx.mark_non_differentiable(ncontrib, gs_idx)
            return out, ncontrib, gs_idx
        return out

    @staticmethod
    def backward(ctx, dL_dout, *_unused):
        bg, W, H, has_bias, has_ndc, has_abs = ctx.meta
        saved = ctx.saved_tensors
        uv, conic, opacity, feature, idx_sorted, tile_range, final_T, ncontrib = saved[:8]
        bias = saved[8] if has_bias else None
        P, C = feature.shape
        g = L.need(dL_dout, "dL_dout")
        dev = feature.device
        pm = ctx.pairmap
        M = idx_sorted.numel()
        if M == 0:   # no Gaussian touches any tile: zero gradients (the reference's atomics add nothing)
            z = torch.zeros
            return (z(P, 2, device=dev), z(P, 3, device=dev), z(opacity.shape, device=dev), z(P, C, device=dev),
                    z(bias.shape, device=dev) if has_bias else None, None, None, None, None, None,
                    z(P, 2, device=dev) if has_ndc else None, z(P, 2, device=dev) if has_abs else None, None, None)
        if pm is not None:
            # pair mode: every gradient element is written by the reduce kernel -> no zero fill
            alloc = torch.empty
            goff, slot_sorted = pm.goff, pm.slot_sorted
            scratch = torch.empty(M * L.lib().splat_blend_pair_floats(C, 1 if has_bias else 0), dtype=torch.float32,
                                  device=dev)
        else:
            alloc = torch.zeros
            goff = slot_sorted = scratch = None
        duv = alloc(P, 2, dtype=torch.float32, device=dev)
        dabs = alloc(P, 2, dtype=torch.float32, device=dev) if has_abs else None
        dconic = alloc(P, 3, dtype=torch.float32, device=dev)
        dop = alloc(opacity.shape, dtype=torch.float32, device=dev)
        dfeat = alloc(P, C, dtype=torch.float32, device=dev)
        dbias = alloc(bias.shape, dtype=torch.float32, device=dev) if has_bias else None
        pack = ctx.pack
        # gradient taps used by densification (reference: alpha_blending.py:112-120): the pair-mode reduce kernel writes
        # them next to dL_duv; the atomic-mode fallback scales afterwards
        in_kernel = scratch is not None
        dndc = alloc(P, 2, dtype=torch.float32, device=dev) if (has_ndc and in_kernel) else None
        dabs_ndc = alloc(P, 2, dtype=torch.float32, device=dev) if (has_abs and in_kernel) else None
        L.check(L.lib().splat_alpha_blending_backward_flags(
            L.ci(P), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opacity), L.ptr(feature), L.ptr(bias), L.ptr(idx_sorted),
            L.ptr(tile_range), L.cf(bg), L.ci(W), L.ci(H), L.ptr(final_T), L.ptr(ncontrib), L.ptr(g), L.ptr(duv),
            L.ptr(dabs), L.ptr(dconic), L.ptr(dop), L.ptr(dfeat), L.ptr(dbias), L.ptr(dndc), L.ptr(dabs_ndc), L.ptr(goff),
            L.ptr(slot_sorted), L.ptr(scratch), L.ptr(pack), L.ci(1), L.ptr(_debug_T_front(H, W, dev)),
            L.ptr(ctx.flags if pm is not None else None), L.stream()))
        if (has_ndc or has_abs) and not in_kernel:
            half = _half_wh(W, H, dev)
            if has_ndc:
                dndc = duv * half[None, :]
            if has_abs:
                dabs_ndc = dabs * half[None, :]
        return duv, dconic, dop, dfeat, dbias, None, None, None, None, None, dndc, dabs_ndc, None, None


def alpha_blending(uv: Tensor, conic: Tensor, opacity: Tensor, feature: Tensor, idx_sorted: Tensor,
                   title_bins: Tensor, bg: float, W: int, H: int, ndc: Optional[Tensor] = None,
                   abs_ndc: Optional[Tensor] = None) -> Tensor:
    """Front-to-back compositing of feature[P,C] -> [C,H,W]."""
    return _AlphaBlend.apply(uv, conic, opacity, feature, None, idx_sorted, title_bins, bg, W, H, ndc, abs_ndc, 0,
                             False)


def alpha_blending_enhanced(uv: Tensor, conic: Tensor, opacity: Tensor, feature: Tensor, idx_sorted: Tensor,
                            title_bins: Tensor, bg: float, W: int, H: int, ndc: Optional[Tensor] = None,
                            abs_ndc: Optional[Tensor] = None, K: int = 10, enable_truncation: bool = False):
    """alpha_blending that also returns ncontrib[H,W] and the first K contributing ids gs_idx[H,W,K]
    (-1 padded); with enable_truncation a pixel stops after K contributors."""
    K = int(K)
    if K < 1:
        raise ValueError("K must be >= 1")
    return _AlphaBlend.apply(uv, conic, opacity, feature, None, idx_sorted, title_bins, bg, W, H, ndc, abs_ndc, K,
                             bool(enable_truncation))


def alpha_blending_with_bias(uv: Tensor, conic: Tensor, opacity: Tensor, feature: Tensor, opacity_bias: Tensor,
                             idx_sorted: Tensor, title_bins: Tensor, bg: float, W: int, H: int,
                             ndc: Optional[Tensor] = None, abs_ndc: Optional[Tensor] = None) -> Tensor:
    """alpha = min(0.99, opacity * G + opacity_bias[P,1])."""
    if opacity_bias is None:
        raise ValueError("opacity_bias is required")
    return _AlphaBlend.apply(uv, conic, opacity, feature, opacity_bias, idx_sorted, title_bins, bg, W, H, ndc,
                             abs_ndc, 0, False)


_BG_CHANNELS = {}


def _bg_channels(widths, bgs, device) -> Tensor:
    """per-channel background of a row of feature sets (read-only, cached: three fill kernels and a cat per call otherwise)"""
    key = (widths, bgs, str(device))
    t = _BG_CHANNELS.get(key)
    if t is None:
        if len(_BG_CHANNELS) > 64:
            _BG_CHANNELS.clear()
        t = torch.cat([torch.full((w,), b, dtype=torch.float32, device=device) for w, b in zip(widths, bgs)])
        _BG_CHANNELS[key] = t
    return t


# ------------------------------------------------------------------ several feature sets, one forward pass
class _BlendShared(torch.autograd.Function):
    """inputs: uv, conic, opacity, idx_sorted, tile_range, W, H, K, ndc, abs_ndc, bgs, detach_opacity, taps, *features"""

    @staticmethod
    def forward(ctx, uv, conic, opacity, idx_sorted, tile_range, W, H, K, ndc, abs_ndc, bgs, detach_opacity, taps, *features):
        uv = L.need(uv, "uv")
        conic = L.need(conic, "conic")
        opacity = L.need(opacity, "opacity")
        idx_sorted = L.need(idx_sorted, "idx_sorted", torch.int32)
        tile_range = L.need(tile_range, "tile_range", torch.int32)
        feats = [L.need(f, f"features[{i}]") for i, f in enumerate(features)]
        P = uv.shape[0]
        if any(f.dim() != 2 or f.shape[0] != P for f in feats):
            raise ValueError("every feature set must be [P, C_i]")
        widths = [int(f.shape[1]) for f in feats]
        C = sum(widths)
        dev = uv.device
        allf = torch.cat(feats, dim=1) if len(feats) > 1 else feats[0]
        bgc = _bg_channels(tuple(widths), tuple(float(b) for b in bgs), dev)
        out = torch.empty(C, H, W, dtype=torch.float32, device=dev)
        final_T = torch.empty(H, W, dtype=torch.float32, device=dev)
        ncontrib = torch.empty(H, W, dtype=torch.int32, device=dev)
        gs_idx = torch.empty(H, W, K, dtype=torch.int32, device=dev) if K > 0 else None
        pack = torch.empty(max(P, 1) * L.lib().splat_blend_pack_floats(C), dtype=torch.float32, device=dev)
        ctx.pairmap = _find_pairmap(idx_sorted, tile_range, P)
        flags = None   # the cull's keep words for the sets' backward passes (see _AlphaBlend)
        if ctx.pairmap is not None and idx_sorted.numel() > 0 and any(ctx.needs_input_grad):
            flags = torch.empty(idx_sorted.numel(), dtype=torch.int32, device=dev)
        L.check(L.lib().splat_alpha_blending_forward_flags(
            L.ci(P), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opacity), L.ptr(allf), L.ptr(None), L.ptr(idx_sorted),
            L.ptr(tile_range), L.cf(0.0), L.ptr(bgc), L.ci(W), L.ci(H), L.ci(K), L.ci(0), L.ptr(out), L.ptr(final_T),
            L.ptr(ncontrib), L.ptr(gs_idx), L.ptr(pack), L.ptr(flags), L.stream()))
        ctx.flags = flags
        # the forward's packed records stay alive until the backward only when the one-pass backward will stage them (the
        # renderer's own plan: splat_blend_sets_uses_forward_pack); 128 B per Gaussian otherwise freed right here
        ctx.fwd_pack = pack if (flags is not None and _uses_forward_pack(widths, detach_opacity, taps)) else None
        ctx.meta = (int(W), int(H), tuple(float(b) for b in bgs), tuple(bool(d) for d in detach_opacity),
                    tuple(bool(t) for t in taps), ndc is not None, abs_ndc is not None, widths)
        ctx.save_for_backward(uv, conic, opacity, idx_sorted, tile_range, final_T, ncontrib, *feats)
        ctx.set_materialize_grads(False)
        imgs = tuple(out[o:o + w] for o, w in zip(_offsets(widths), widths))
        extra = (ncontrib,) + ((gs_idx,) if gs_idx is not None else ())
        ctx.mark_non_differentiable(*extra)
        return imgs + extra

    @staticmethod
    def backward(ctx, *grads):
        W, H, bgs, detach, taps, has_ndc, has_abs, widths = ctx.meta
        uv, conic, opacity, idx_sorted, tile_range, final_T, ncontrib = ctx.saved_tensors[:7]
        feats = ctx.saved_tensors[7:]
        P, dev = uv.shape[0], uv.device
        pm = ctx.pairmap
        M = idx_sorted.numel()
        lib = L.lib()
        duv_t = dconic_t = dop_t = dndc = dabs_t = None
        dfeats = [None] * len(feats)
        one = _BlendShared._one_pass(ctx, grads)
        if one is not None:
            return one
        for s, (f, g) in enumerate(zip(feats, grads[:len(feats)])):
            if g is None:
                continue
            C = f.shape[1]
            g = L.need(g, "dL_dout")
            want_abs = has_abs and taps[s]
            if M == 0:
                dfeats[s] = torch.zeros(P, C, dtype=torch.float32, device=dev)
                continue
            if pm is not None:
                alloc = torch.empty
                goff, slot_sorted = pm.goff, pm.slot_sorted
                scratch = torch.empty(M * lib.splat_blend_pair_floats(C, 0), dtype=torch.float32, device=dev)
            else:
                alloc = torch.zeros
                goff = slot_sorted = scratch = None
            duv = alloc(P, 2, dtype=torch.float32, device=dev)
            dabs = alloc(P, 2, dtype=torch.float32, device=dev) if want_abs else None
            dconic = alloc(P, 3, dtype=torch.float32, device=dev)
            dop = alloc(opacity.shape, dtype=torch.float32, device=dev)
            dfeat = alloc(P, C, dtype=torch.float32, device=dev)
            pack = torch.empty(max(P, 1) * lib.splat_blend_pack_floats(C), dtype=torch.float32, device=dev)
            L.check(lib.splat_alpha_blending_backward_flags(
                L.ci(P), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opacity), L.ptr(f), L.ptr(None), L.ptr(idx_sorted),
                L.ptr(tile_range), L.cf(bgs[s]), L.ci(W), L.ci(H), L.ptr(final_T), L.ptr(ncontrib), L.ptr(g), L.ptr(duv),
                L.ptr(dabs), L.ptr(dconic), L.ptr(dop), L.ptr(dfeat), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(goff),
                L.ptr(slot_sorted), L.ptr(scratch), L.ptr(pack), L.ci(0), L.ptr(_debug_T_front(H, W, dev)),
                L.ptr(ctx.flags if pm is not None else None), L.stream()))
            dfeats[s] = dfeat
            duv_t = duv if duv_t is None else duv_t + duv
            dconic_t = dconic if dconic_t is None else dconic_t + dconic
            if not detach[s]:
                dop_t = dop if dop_t is None else dop_t + dop
            if taps[s]:
                half = _half_wh(W, H, dev)
                if has_ndc:
                    t = duv * half[None, :]
                    dndc = t if dndc is None else dndc + t
                if want_abs:
                    t = dabs * half[None, :]
                    dabs_t = t if dabs_t is None else dabs_t + t
        if M == 0:
            z = torch.zeros
            duv_t, dconic_t = z(P, 2, device=dev), z(P, 3, device=dev)
            dop_t = z(opacity.shape, device=dev)
            dndc = z(P, 2, device=dev) if has_ndc else None
            dabs_t = z(P, 2, device=dev) if has_abs else None
        return (duv_t, dconic_t, dop_t, None, None, None, None, None, dndc, dabs_t, None, None, None) + tuple(dfeats)


def _blend_shared_one_pass(ctx, grads):
    """The sets' backward in ONE pass of the tile kernels (splat_alpha_blending_backward_batch_sets at F = 1: one replay of the
    alpha / transmittance chain for the renderer's three blends, dL/dalpha routed per set) + one per-Gaussian sum of the SETS
    records (splat_pair_records_segment_sum), when the sets fit the one-pass plan (one set per routing group -- taps / live
    opacity / detached opacity -- of at most 4 / 4 / 20 channels), every set has a gradient and the sort is this package's
    (pair map + the forward's cull words).  Returns the backward's result tuple, or None: one native backward per set."""
    import ctypes

    from ..frames import _one_pass_plan, _set_groups
    W, H, bgs, detach, taps, has_ndc, has_abs, widths = ctx.meta
    uv, conic, opacity, idx_sorted, tile_range, final_T, ncontrib = ctx.saved_tensors[:7]
    feats = ctx.saved_tensors[7:]
    n = len(feats)
    M = idx_sorted.numel()
    if ctx.pairmap is None or ctx.flags is None or M == 0 or any(g is None for g in grads[:n]):
        return None
    if not OPTIONS["shared_one_pass"]:
        return None
    C = sum(widths)
    meta = tuple((w, bg, d, t) for w, bg, d, t in zip(widths, bgs, detach, taps))
    plan = _one_pass_plan(meta, list(widths), C)
    if plan is None:
        return None
    c0s, cns, pbg, _, tap_set = plan
    P, dev = uv.shape[0], uv.device
    lib = L.lib()
    groups = _set_groups(meta)
    fptr, dptr = [0, 0, 0], [0, 0, 0]
    keep = []
    for f, g, grp, w in zip(feats, grads[:n], groups, widths):
        g = L.need(g, "dL_dout")
        if tuple(g.shape) != (w, H, W):
            raise ValueError("image gradient of a set must be [c, H, W]")
        keep.append(g)
        fptr[grp], dptr[grp] = f.data_ptr(), g.data_ptr()
    want_abs = 1 if (has_abs and tap_set is not None) else 0
    ncp = int(lib.splat_blend_sets_pair_stride(C))
    NG = 12                                        # csrc/common.h SETS_NG: [ux uy ca cb | cc o ax ay | tx ty 0 0 | row channels]
    assert ncp == (NG + C + 3) // 4 * 4
    rec = torch.empty(M * ncp, dtype=torch.float32, device=dev)
    i3_ = ctypes.c_int32 * 3
    std = ctx.fwd_pack is not None           # decided at forward time (the library option is not re-queried here)
    pack = None if std else torch.empty(max(P, 1) * int(lib.splat_blend_sets_pack_floats()), dtype=torch.float32, device=dev)
    i3, f3, p3, l3 = ctypes.c_int32 * 3, ctypes.c_float * 3, ctypes.c_void_p * 3, ctypes.c_int64 * 3
    pm = ctx.pairmap
    L.check(lib.splat_alpha_blending_backward_batch_sets_packed(
        L.ci(1), L.ci(P), L.ci(C), i3(*c0s), i3(*cns), f3(*pbg), L.ptr(uv), L.ptr(conic), L.ptr(opacity), ctypes.c_int64(0),
        L.ptr(None), ctypes.c_int64(0), p3(*fptr), l3(0, 0, 0), L.ptr(idx_sorted), L.ptr(tile_range), ctypes.c_int64(M), L.ci(W),
        L.ci(H), L.ptr(final_T), L.ptr(ncontrib), L.ptr(None), p3(*dptr), L.ci(want_abs), L.ptr(pm.slot_sorted), L.ptr(rec),
        L.ptr(pack), L.ptr(ctx.flags), L.ptr(_debug_T_front(H, W, dev)), L.ptr(ctx.fwd_pack if std else None), L.stream()))
    red = torch.empty(P, ncp, dtype=torch.float32, device=dev)
    L.check(lib.splat_pair_records_segment_sum(L.ci(P), L.ci(ncp), L.ptr(rec), L.ptr(pm.goff), L.ptr(red), L.stream()))
    half = _half_wh(W, H, dev)
    duv, dconic = red[:, 0:2].contiguous(), red[:, 2:5].contiguous()
    dop = red[:, 5].contiguous().reshape(opacity.shape)
    dndc = (red[:, 8:10] * half[None, :]) if (has_ndc and tap_set is not None) else (torch.zeros(P, 2, device=dev) if has_ndc else None)
    dabs = (red[:, 6:8] * half[None, :]) if want_abs else (torch.zeros(P, 2, device=dev) if has_abs else None)
    dfeats = []
    for grp, w in zip(groups, widths):
        c0 = NG + c0s[grp]
        dfeats.append(red[:, c0:c0 + w].contiguous())
    return (duv, dconic, dop, None, None, None, None, None, dndc, dabs, None, None, None) + tuple(dfeats)


_BlendShared._one_pass = staticmethod(_blend_shared_one_pass)

# python-level switches of the shared blend (tests flip them in code -- no environment variable is read; the library's own
# options: L.set_option)
OPTIONS = {"shared_one_pass": True, "sets_fwdrec": True, "rasterization_one_call": True}


def _uses_forward_pack(widths, detach, taps) -> bool:
    """will the one-pass backward of these sets stage the forward's packed records?  (the C library owns the condition)"""
    import ctypes

    from ..frames import _one_pass_plan
    if not (OPTIONS["shared_one_pass"] and OPTIONS["sets_fwdrec"]):
        return False
    C = sum(widths)
    meta = tuple((w, 0.0, bool(d), bool(t)) for w, d, t in zip(widths, detach, taps))
    plan = _one_pass_plan(meta, list(widths), C)
    if plan is None:
        return False
    i3 = ctypes.c_int32 * 3
    return bool(L.lib().splat_blend_sets_uses_forward_pack(L.ci(C), i3(*plan[0]), i3(*plan[1]), L.ci(1)))


def _offsets(widths):
    o, out = 0, []
    for w in widths:
        out.append(o)
        o += w
    return out


def alpha_blending_shared(uv: Tensor, conic: Tensor, opacity: Tensor, features, idx_sorted: Tensor, title_bins: Tensor,
                          bgs, W: int, H: int, ndc: Optional[Tensor] = None, abs_ndc: Optional[Tensor] = None, K: int = 0,
                          detach_opacity=None, taps=None):
    """Several feature sets of ONE geometry composited in a single forward pass (extension).

    Equivalent to ``alpha_blending(uv, conic, opacity[.detach()], features[i], idx_sorted, title_bins, bgs[i], W, H,
    ndc or ndc.detach())`` for every i (``detach_opacity[i]``: that set does not feed the opacity gradient;
    ``taps[i]``: that set feeds the ``ndc`` / ``abs_ndc`` gradient taps), but the transmittance chain, the culling and
    the record gathers are paid once: this is how the reference's renderer uses its three blends
    (dptr_ortho_enhanced.py:331-375).  Backward: one native backward per set.  Returns
    ``(images..., ncontrib)`` plus ``gs_idx[H,W,K]`` when ``K > 0`` (``alpha_blending_enhanced``'s extra outputs)."""
    features = list(features)
    n = len(features)
    if n < 1 or len(bgs) != n:
        raise ValueError("need one background value per feature set")
    detach_opacity = [False] * n if detach_opacity is None else list(detach_opacity)
    taps = [i == 0 for i in range(n)] if taps is None else list(taps)
    if len(detach_opacity) != n or len(taps) != n:
        raise ValueError("detach_opacity / taps need one entry per feature set")
    return _BlendShared.apply(uv, conic, opacity, idx_sorted, title_bins, int(W), int(H), int(K), ndc, abs_ndc, tuple(bgs),
                              tuple(detach_opacity), tuple(taps), *features)


# ------------------------------------------------------------------ rasterization (5-op chain)
def rasterization(xyz: Tensor, scale: Tensor, rotate: Tensor, opacity: Tensor, feature: Tensor, intr: Tensor,
                  extr: Tensor, W: int, H: int, bg: float, ndc: Optional[Tensor] = None) -> Tensor:
    """project_point -> compute_cov3d -> ewa_project -> sort_gaussian -> alpha_blending (reference:
    src/submodules/dptr/dptr/gs/__init__.py:28-100).  The three per-Gaussian operators run as ONE fused pass per direction
    (``preprocess_persp``: same arithmetic, no visible mask / cov3d round trips) unless the camera itself requires grad."""
    cam_grad = (isinstance(intr, Tensor) and intr.requires_grad) or (isinstance(extr, Tensor) and extr.requires_grad)
    if not cam_grad and ndc is None and OPTIONS["rasterization_one_call"] and feature.dim() == 2 and 1 <= feature.shape[1] <= 32:
        # the whole chain behind ONE call of the C ABI per direction (a pooled one-frame batch: frames.frame_rasterization) -- half the
        # host time of the operator chain; with a tap tensor (`ndc`) or a differentiable camera the operators below run
        from ..frames import frame_rasterization
        return frame_rasterization(xyz, scale, rotate, opacity, feature, extr, W, H, bg, intr=intr, nearest=0.2, extent=1.3)
    if cam_grad:
        uv, depth = project_point(xyz, intr, extr, W, H)
        visible = depth != 0
        cov3d = compute_cov3d(scale, rotate, visible)
        conic, radius, tiles = ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    else:
        from .fused_ops import preprocess_persp
        uv, depth, conic, radius, tiles = preprocess_persp(xyz, scale, rotate, intr, extr, W, H)
    # the lists stay inside: only the pairs whose tile the splat can reach with alpha >= 1/255 (reach masks)
    idx_sorted, tile_range, _ = sort_gaussian_capped(uv, depth, W, H, radius, None, conic.detach(), opacity.detach())
    return alpha_blending(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, ndc)


def rasterization_ortho(xyz: Tensor, scale: Tensor, rotate: Tensor, opacity: Tensor, feature: Tensor, extr: Tensor, W: int, H: int,
                        bg: float, offset: Optional[Tensor] = None, nearest: float = 0.01, extent: float = 1.3, grad_sink=None) -> Tensor:
    """``rasterization`` for the ORTHOGRAPHIC camera of the reference's video renderer (dptr_ortho_enhanced.py:282-349: project_point,
    compute_cov3d, the EWA projection, sort_gaussian, alpha_blending), one frame, behind one call of the C ABI per direction
    (``frames.frame_rasterization``: a pooled one-frame batch).  ``offset`` [P, 3] is added to ``xyz`` inside the kernel;
    ``grad_sink`` may hold buffers for "xyz", "scales", "uquats", "opacity", "feature"."""
    from ..frames import frame_rasterization
    return frame_rasterization(xyz, scale, rotate, opacity, feature, extr, W, H, bg, intr=None, offset=offset, nearest=nearest,
                               extent=extent, grad_sink=grad_sink)
