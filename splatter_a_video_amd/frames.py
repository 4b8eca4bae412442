"""Frame-batched rendering: F frames of one Gaussian set through ONE set of kernel launches (SURVEY 7 stage 6).

The reference renders the frames of a batch one after the other (``render_batch``,
src/pointrix/renderer/dptr_ortho_enhanced.py:385-433: ~13 native launches and ~80 eager kernels per frame).  Here every
kernel of the per-frame path -- orthographic preprocess, tile binning, per-tile sort, record packing, compositing
forward, compositing backward -- takes the frame as a grid dimension, and the Gaussian-side backward (pair reduce +
preprocess backward) runs once per batch: a batch costs the launches of one frame, the short kernels fill the chip, and
the compositing kernels see F x T tiles in one launch (no half-empty last round of workgroups).  Same images and
gradients as F calls of the per-frame operators (tests/test_gpu_frames.py).

``FrameBatch`` owns the batch's device buffers (allocated once, ~140 MB per 480p frame of 300k Gaussians); ``render`` is
autograd-aware and crosses the C ABI ONCE per direction (``splat_frames_forward`` / ``splat_frames_backward``: the struct
``splat_frames_t`` carries every pointer); ``render_sets`` / ``render_dynamic`` compose the ``*_batch`` entry points.
No CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib as L
from .gs.fused_ops import check_sink
from .gs.point_ops import _extr12, _points


_FRAMES_FIELDS = ([("struct_bytes", ctypes.c_size_t)]
                  + [(n, ctypes.c_int32) for n in ("F", "P", "W", "H", "C", "want_abs", "accumulate")]
                  + [("capacity", ctypes.c_int64)] + [(n, ctypes.c_float) for n in ("nearest", "extent", "bg")]
                  + [("stream", ctypes.c_void_p)]
                  + [(n, ctypes.c_void_p) for n in (
                      "xyz", "offsets", "scales", "uquats", "opacity", "feature", "extr", "uv", "depth", "conic", "radius",
                      "bin_scratch", "tile_range", "pairs", "overflow", "goff_incl", "owner", "idx_sorted", "slot_sorted", "keys",
                      "pack", "out", "final_T", "ncontrib", "dL_dout", "pair_records", "d_xyz", "d_scales", "d_uquats",
                      "d_opacity", "d_feature", "tap", "abs_tap", "radii_max", "dbg_T_front", "cull_flags")]
                  + [("extr_frame_stride", ctypes.c_int64), ("intr_frame_stride", ctypes.c_int64), ("intr", ctypes.c_void_p),
                     ("perspective", ctypes.c_int32), ("reach", ctypes.c_void_p)])


class _SplatFrames(ctypes.Structure):
    """mirror of splat_frames_t (include/splat_hip.h): every pointer of a batch, for the one-call entry points (the
    library checks ``struct_bytes`` against its own sizeof)"""
    _fields_ = _FRAMES_FIELDS


class _SplatCamera(ctypes.Structure):
    """mirror of splat_camera_t (include/splat_hip.h)"""
    _fields_ = [("perspective", ctypes.c_int32), ("intr", ctypes.c_void_p), ("intr_frame_stride", ctypes.c_int64),
                ("extr", ctypes.c_void_p), ("extr_frame_stride", ctypes.c_int64), ("offsets", ctypes.c_void_p)]


def _versions(*tensors):
    """(tensor, autograd version) of the camera / offset tensors a backward re-reads WITHOUT save_for_backward (they are
    passed as raw pointers): an in-place edit between forward and backward would silently change the gradients"""
    return tuple((t, t._version) for t in tensors if t is not None)


def _check_versions(saved, what: str) -> None:
    for t, v in saved:
        if t._version != v:
            raise RuntimeError(f"FrameBatch: {what} was modified in place between the forward and its backward (the backward "
                               "re-projects under the forward's cameras / offsets); clone it before editing")


class _Camera:
    """Camera of a batch: ``extr`` one world-to-camera matrix ([4,4] / [3,4]) or one per frame ([F,4,4] / [F,3,4] -- the
    reference's render_batch gives every batch element its own, dptr_ortho_enhanced.py:409-411); ``intr`` None: the
    orthographic camera of the video renderer, else the pinhole camera's (fx, fy, cx, cy), [4] or [F,4]."""

    def __init__(self, extr: Tensor, intr: Optional[Tensor], F: int):
        extr = L.need(extr, "extr")
        if extr.dim() == 3:
            if extr.shape[0] != F or tuple(extr.shape[1:]) not in ((4, 4), (3, 4)):
                raise ValueError(f"per-frame extr must be [F={F}, 4, 4] or [F, 3, 4]")
            self.extr_fs = int(extr.shape[1] * extr.shape[2])
        else:
            if extr.numel() < 12:
                raise ValueError("extr must hold at least 3x4 floats (row-major [R|T])")
            self.extr_fs = 0
        self.extr = extr
        self.intr, self.intr_fs = None, 0
        if intr is not None:
            intr = L.need(intr, "intr")
            if intr.dim() == 2:
                if tuple(intr.shape) != (F, 4):
                    raise ValueError(f"per-frame intr must be [F={F}, 4]")
                self.intr_fs = 4
            elif intr.numel() < 4:
                raise ValueError("intr must be [fx, fy, cx, cy]")
            self.intr = intr
        self.perspective = intr is not None

    def struct(self, offsets: Optional[Tensor] = None) -> _SplatCamera:
        c = _SplatCamera()
        c.perspective = 1 if self.perspective else 0
        c.intr = None if self.intr is None else self.intr.data_ptr()
        c.intr_frame_stride = self.intr_fs
        c.extr = self.extr.data_ptr()
        c.extr_frame_stride = self.extr_fs
        c.offsets = None if offsets is None else offsets.data_ptr()
        return c

    def tensors(self):
        return (self.extr,) if self.intr is None else (self.extr, self.intr)


# python-level switch of the multi-set paths (tests flip it in code -- no environment variable is read; the library's own
# options: L.set_option)
# "reach": create only the (Gaussian, tile) pairs whose tile the splat can reach with alpha >= 1/255 (splat_bin_*_batch_reach:
# a third fewer pairs on the bench scene, same images / ids / gradients); False: the reference's bounding-square pairs
OPTIONS = {"sets_one_pass": True, "reach": True}


def _tiles(W: int, H: int) -> int:
    return ((W + 15) // 16) * ((H + 15) // 16)


class FrameBatch:
    """Buffers and launch sequence of a batch of ``F`` frames of ``P`` Gaussians at ``W`` x ``H`` with ``C`` composited
    feature channels (C <= 32).  ``capacity`` = tile-Gaussian pairs reserved per frame; None: measured on the first call
    (one host sync), afterwards the batch runs without any host synchronisation and ``check()`` (call it whenever the
    host synchronises anyway, e.g. once per optimiser step) raises if a frame outgrew it; without ``check()`` the next
    forward raises, a step or a few late (the flag travels to pinned host memory behind the binning kernels; the error is
    raised once and the flag cleared, so batches that fit keep running afterwards).  An overflow of the LAST batches of a run
    is only seen by ``check()``: call it after the final step."""

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
        self.reach = torch.empty(F_, P_, dtype=i32, device=dev)      # reach words of the count step for the sort step
        self.pack = torch.empty(F_ * P_ * int(lib.splat_blend_pack_floats(C)), dtype=f32, device=dev)
        self.out = None
        self.final_T = torch.empty(F_, H, W, dtype=f32, device=dev)
        self.ncontrib = torch.empty(F_, H, W, dtype=i32, device=dev)
        self.ncp = int(lib.splat_blend_pair_stride(C, 1 if want_abs else 0, 0))
        self.tap = torch.zeros(P_, 2, dtype=f32, device=dev)         # densification taps of the last backward
        self.abs_tap = torch.zeros(P_, 2, dtype=f32, device=dev) if want_abs else None
        self.radii_max = torch.zeros(P_, dtype=i32, device=dev)
        self.capacity = None
        self.keys = self.owner = self.idx_sorted = self.slot_sorted = self.pair_records = self.cull_flags = None
        # one forward, one backward: every render* call overwrites the batch's buffers (sorted lists, packed records, final_T,
        # ncontrib, cull flags ...) that its own backward reads.  `generation` counts the forwards; a backward whose forward
        # is not the latest one raises instead of silently using another call's lists.
        self.generation = 0
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
        self.cull_flags = torch.empty(F_, capacity, dtype=torch.int32, device=dev)   # the forward's cull, for the backward

    def _set_buffer(self, key, numel: int) -> Tensor:
        """scratch of the multi-set backward (pair records / packed records per feature set), kept across steps"""
        cache = self.__dict__.setdefault("_set_buffers", {})
        t = cache.get(key)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, dtype=torch.float32, device=self.dev)
            cache[key] = t
        return t

    def check(self) -> int:
        """host sync; raises when a frame's pairs exceeded the capacity (its surplus pairs were dropped), else returns the
        largest per-frame pair count"""
        m = int(self.pairs.max().item())
        if self.capacity is not None and m > self.capacity:
            self.overflow.zero_()                   # reported here: the sticky flag must not fail the next batch again
            self._ovf_event = None
            raise L.SplatError(f"FrameBatch: {m} tile-Gaussian pairs in one frame exceed the capacity {self.capacity}")
        return m

    def _begin_forward(self) -> int:
        self._poll_overflow()
        self.generation += 1
        return self.generation

    def _note_overflow(self) -> None:
        """behind the binning kernels of a forward: the overflow flag travels to pinned host memory asynchronously.  The
        device flag is sticky (bin_scatter only ever sets it), so ONE copy in flight is enough: while an earlier copy has not
        completed no new one is enqueued -- a host that runs ahead of the GPU still polls an event that WILL complete (replacing
        the pending event on every forward left it polling events that never had)."""
        if torch.cuda.is_current_stream_capturing():
            return                                  # (inside a HIP-graph capture nothing may be polled: call check() after a replay)
        ev = getattr(self, "_ovf_event", None)
        if ev is not None and not ev.query():
            return                                  # the copy in flight is older than this batch; the next one sees its flag
        if ev is not None:
            self._read_overflow()                   # a completed copy nobody polled yet
        if getattr(self, "_ovf_host", None) is None:
            self._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._ovf_host.copy_(self.overflow, non_blocking=True)
        self._ovf_event = torch.cuda.Event()
        self._ovf_event.record()

    def _read_overflow(self) -> None:
        """the completed host copy of the flag: raise once, then clear the device flag so that later batches that fit run
        again (a caller may catch the error, e.g. to rebuild with more slack)"""
        self._ovf_event = None
        if int(self._ovf_host[0]) != 0:
            self._ovf_host[0] = 0
            self.overflow.zero_()                   # stream-ordered behind the kernels that set it
            raise L.SplatError(f"FrameBatch: a batch held more tile-Gaussian pairs in one frame than the capacity "
                               f"{self.capacity} (its surplus pairs were dropped: images and gradients of that batch -- an earlier "
                               "one, possibly also the one just enqueued: the flag is sticky -- were incomplete); build the batch "
                               "with a larger `capacity` / `slack`")

    def _poll_overflow(self) -> None:
        """at the next forward, without a host synchronisation: raises if an earlier batch outgrew the capacity -- a caller
        that never calls check() still learns of it a few steps late instead of never"""
        if torch.cuda.is_current_stream_capturing():
            return
        ev = getattr(self, "_ovf_event", None)
        if ev is not None and ev.query():
            self._read_overflow()

    def _check_generation(self, gen: int) -> None:
        if gen != self.generation:
            raise L.SplatError(
                f"FrameBatch: backward of render call #{gen} after render call #{self.generation} on the same batch -- the later "
                "forward has overwritten the buffers this backward reads (sorted lists, packed records, final_T, ncontrib).  "
                "A FrameBatch holds ONE forward at a time: run the backward before the next render* (also under no_grad), or "
                "use one FrameBatch per micro-batch.")

    def memory_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in vars(self).values() if isinstance(t, Tensor))

    # ------------------------------------------------------------------ forward / backward launch sequences
    def _geometry(self, xyz, scales, uquats, offsets, cam: "_Camera", nearest, extent, opacity=None, op_fs=0):
        """fused preprocess (projection + cov3d + EWA under the batch's camera) + tile binning + per-tile depth sort of all frames"""
        lib, st = L.lib(), L.stream()
        F_, P_, W, H = self.F, self.P, self.W, self.H
        L.check(lib.splat_preprocess_forward_batch_cam(
            L.ci(F_), L.ci(P_), L.ptr(xyz), L.ptr(offsets), L.ptr(scales), L.ptr(uquats), ctypes.byref(cam.struct()), L.ci(W),
            L.ci(H), L.cf(nearest), L.cf(extent), L.ptr(self.uv), L.ptr(self.depth), L.ptr(self.conic), L.ptr(self.radius), st))
        self._bin_and_sort(opacity, op_fs)

    def _bin_and_sort(self, opacity: Optional[Tensor] = None, op_fs: int = 0):
        """tile binning + per-tile depth sort of all frames.  With ``opacity`` ([P], or per frame [F,P] with ``op_fs`` = P) and
        OPTIONS["reach"]: only the pairs whose tile the splat can reach with alpha >= 1/255 are created."""
        lib, st = L.lib(), L.stream()
        F_, P_, W, H = self.F, self.P, self.W, self.H
        reach = opacity is not None and OPTIONS["reach"]
        if reach:
            L.check(lib.splat_bin_count_batch_reach(
                L.ci(F_), L.ci(P_), L.ptr(self.uv), L.ptr(self.radius), L.ptr(self.conic), L.ptr(opacity), ctypes.c_int64(op_fs),
                L.ci(W), L.ci(H), L.ptr(self.bin_scratch), L.ptr(self.tile_range), L.ptr(self.pairs), L.ptr(None),
                L.ptr(self.reach), st))
        else:
            L.check(lib.splat_bin_count_batch(L.ci(F_), L.ci(P_), L.ptr(self.uv), L.ptr(self.radius), L.ci(W), L.ci(H),
                                              L.ptr(self.bin_scratch), L.ptr(self.tile_range), L.ptr(self.pairs), st))
        if self.capacity is None:      # first batch: size the pair buffers (the only host sync of the object's life)
            self._reserve(int(int(self.pairs.max().item()) * self.slack) + 1024)
        cap = self.capacity
        if reach:
            L.check(lib.splat_bin_sort_batch_reach(
                L.ci(F_), L.ci(P_), L.ptr(self.uv), L.ptr(self.depth), L.ptr(self.radius), L.ptr(self.reach), L.ci(W), L.ci(H),
                L.ptr(self.bin_scratch), L.ptr(self.tile_range),
                ctypes.c_int64(cap), L.ptr(self.keys), L.ptr(self.idx_sorted), L.ptr(self.overflow), L.ptr(self.goff),
                L.ptr(self.owner), L.ptr(self.slot_sorted), st))
        else:
            L.check(lib.splat_bin_sort_batch(
                L.ci(F_), L.ci(P_), L.ptr(self.uv), L.ptr(self.depth), L.ptr(self.radius), L.ci(W), L.ci(H),
                L.ptr(self.bin_scratch), L.ptr(self.tile_range), ctypes.c_int64(cap), L.ptr(self.keys), L.ptr(self.idx_sorted),
                L.ptr(self.overflow), L.ptr(self.goff), L.ptr(self.owner), L.ptr(self.slot_sorted), st))
        self._note_overflow()

    def _struct(self, xyz, scales, uquats, opacity, feature, offsets, cam, bg, nearest=0.01, extent=1.3) -> _SplatFrames:
        b = _SplatFrames()
        extr = cam.extr
        b.extr_frame_stride, b.intr_frame_stride = cam.extr_fs, cam.intr_fs
        b.intr = None if cam.intr is None else cam.intr.data_ptr()
        b.perspective = 1 if cam.perspective else 0
        b.struct_bytes = ctypes.sizeof(_SplatFrames)
        b.F, b.P, b.W, b.H, b.C = self.F, self.P, self.W, self.H, self.C
        b.want_abs = 1 if self.want_abs else 0
        b.capacity = self.capacity or 0
        b.nearest, b.extent, b.bg = nearest, extent, bg
        b.stream = L.stream()
        dp = lambda t: None if t is None else t.data_ptr()
        for name, t in dict(xyz=xyz, offsets=offsets, scales=scales, uquats=uquats, opacity=opacity, feature=feature, extr=extr,
                            uv=self.uv, depth=self.depth, conic=self.conic, radius=self.radius, bin_scratch=self.bin_scratch,
                            tile_range=self.tile_range, pairs=self.pairs, overflow=self.overflow, goff_incl=self.goff,
                            owner=self.owner, idx_sorted=self.idx_sorted, slot_sorted=self.slot_sorted, keys=self.keys,
                            pack=self.pack, final_T=self.final_T, ncontrib=self.ncontrib, pair_records=self.pair_records,
                            tap=self.tap, abs_tap=self.abs_tap, radii_max=self.radii_max,
                            cull_flags=self.cull_flags,
                            reach=self.reach if (opacity is not None and OPTIONS["reach"]) else None).items():
            setattr(b, name, dp(t))
        return b

    def _forward_onecall(self, xyz, scales, uquats, opacity, feature, offsets, cam, bg, nearest, extent):
        """the whole forward of the batch behind ONE crossing of the C ABI (splat_frames_forward)"""
        lib = L.lib()
        if self.capacity is None:      # first batch: size the pair buffers (the only host sync of the object's life)
            L.check(lib.splat_frames_count(ctypes.byref(self._struct(xyz, scales, uquats, opacity, feature, offsets, cam, bg,
                                                                     nearest, extent))))
            self._reserve(int(int(self.pairs.max().item()) * self.slack) + 1024)
        out = torch.empty(self.F, self.C, self.H, self.W, dtype=torch.float32, device=self.dev)
        b = self._struct(xyz, scales, uquats, opacity, feature, offsets, cam, bg, nearest, extent)
        b.out = out.data_ptr()
        L.check(lib.splat_frames_forward(ctypes.byref(b)))
        self._note_overflow()
        return out

    def _backward_onecall(self, dL_dout, xyz, scales, uquats, offsets, cam, bg, bufs, accumulate, dbg=None):
        b = self._struct(xyz, scales, uquats, None, None, offsets, cam, bg)
        b.accumulate = 1 if accumulate else 0
        b.dL_dout = dL_dout.data_ptr()
        b.d_xyz, b.d_scales, b.d_uquats = bufs["xyz"].data_ptr(), bufs["scales"].data_ptr(), bufs["uquats"].data_ptr()
        b.d_opacity, b.d_feature = bufs["opacity"].data_ptr(), bufs["feature"].data_ptr()
        b.dbg_T_front = None if dbg is None else dbg.data_ptr()
        L.check(L.lib().splat_frames_backward(ctypes.byref(b)))

    # ------------------------------------------------------------------ public
    def render(self, xyz: Tensor, scales: Tensor, uquats: Tensor, opacity: Tensor, feature: Tensor, offsets: Optional[Tensor],
               extr: Tensor, bg: float = 0.0, nearest: float = 0.01, extent: float = 1.3,
               grad_sink: Optional[Dict[str, Tensor]] = None, intr: Optional[Tensor] = None) -> Tensor:
        """images [F,C,H,W] of the F frames ``xyz + offsets[f]`` (static scale / rotation / opacity / feature [P,C]) under
        the orthographic camera ``extr`` -- or, with ``intr`` (fx, fy, cx, cy), the pinhole camera of gs.rasterization.
        ``extr`` / ``intr`` may hold one camera per frame ([F,4,4] / [F,4]: render_batch's per-element cameras,
        dptr_ortho_enhanced.py:409-411); ``offsets`` may then be None.  With one orthographic camera the Gaussian-side
        backward runs its projection chain once per batch, otherwise once per frame (cameras are not differentiated).  Differentiable w.r.t. xyz, scales, uquats, opacity, feature; the backward ADDS
        the gradients of the names found in ``grad_sink`` into those buffers instead (e.g. FlatGradBucket views; autograd
        then sees no gradient for them), the others are returned to autograd.  After
        the backward, ``tap`` / ``abs_tap`` hold the batch's summed densification taps and ``radii_max`` the largest
        screen radius of every Gaussian over the frames."""
        sink = check_sink(grad_sink, {"xyz": xyz, "scales": scales, "uquats": uquats, "opacity": opacity, "feature": feature})
        return _RenderFrames.apply(xyz, scales, uquats, opacity, feature, offsets, extr, self, float(bg), float(nearest),
                                   float(extent), sink, intr)


    def fuse_l1(self, targets, weights, sums: Tensor) -> None:
        """arm the LOSS-FUSED backward for the next backward pass of a ``render_sets`` / ``render_dynamic_sets`` forward: the
        image gradients of ``sum_s weights[s] * mean |out[s] - targets[s]|`` (the reference's l1_loss terms, src/trainer_fragGS.py:
        573-600) are derived inside the tile kernel from the forward's output row and the target images ``targets[s]``
        [F, c_s, H, W]; ``sums`` [F, tiles, 3] receives sum |out - target| per frame, tile and routing group (tap set, second
        set, detached set; every entry is written: add them up).  Call ``torch.autograd.backward(outputs, fb.l1_placeholders())`` afterwards: the gradient tensors handed to
        the backward are placeholders (no gradient image is written or read)."""
        self._l1 = dict(targets=list(targets), weights=list(weights), sums=sums)

    def l1_placeholders(self, widths=None):
        """zero-stride gradient tensors of the sets' shapes for a loss-fused backward (no memory behind them)"""
        ws = widths or [3, 1, self.C - 4]
        z = torch.zeros(1, dtype=torch.float32, device=self.dev)
        return [z.expand(self.F, w, self.H, self.W) for w in ws]

    # ------------------------------------------------------------------ dynamic Gaussians (rows a15 + f1)
    def frame_table(self, clock, times) -> Tensor:
        """device table of the per-frame scalars (segment, offset inside it, time bases) of ``times`` (cached)"""
        key = (id(clock), tuple(float(t) for t in times))
        cache = self.__dict__.setdefault("_tables", {})
        tab = cache.get(key)
        if tab is None:
            import numpy as np
            if len(times) != self.F:
                raise ValueError(f"the batch holds {self.F} frames")
            host = np.zeros((self.F, 16), np.float32)
            for f, t in enumerate(times):
                seg, d, basis = clock.scalars(t)
                host[f, 0] = np.array([seg], np.int32).view(np.float32)[0]
                host[f, 1] = d
                host[f, 2:14] = np.frombuffer(basis, dtype=np.float32, count=12)
            from .dynamics import _upload, _walk_order
            _walk_order(host)
            tab = _upload(host, self.dev)
            cache[key] = tab
        return tab

    def render_dynamic(self, clock, times, extr: Tensor, feature: Tensor, *, position: Tensor, pos_cubic_node: Tensor,
                       rotation: Tensor, rot_poly_feat: Tensor, rot_fourier_feat: Tensor, opacity: Tensor, scaling: Tensor,
                       cubic_layout: int = 1, bg: float = 0.0, nearest: float = 0.01, extent: float = 1.3,
                       grad_sink: Optional[Dict[str, Tensor]] = None) -> Tensor:
        """images [F,C,H,W] of the reference's dynamic Gaussians (spline position, normalised rotation with the detached
        polynomial / Fourier sums, sigmoid opacity, exp scale: src/dynamic_gaussian_with_base_point_cloud.py:171-250) at the
        frame times ``times`` of ``clock``, evaluated inside the batched preprocess.  Differentiable w.r.t. position,
        pos_cubic_node, rotation, opacity, scaling and feature; ``grad_sink`` as in ``render``."""
        tab = self.frame_table(clock, times)
        sink = check_sink(grad_sink, {"position": position, "pos_cubic_node": pos_cubic_node, "rotation": rotation,
                                      "opacity": opacity, "scaling": scaling, "feature": feature})
        return _RenderDynamic.apply(position, pos_cubic_node, rotation, opacity, scaling, feature, rot_poly_feat, rot_fourier_feat,
                                    extr, tab, self, int(clock.interval_num), int(cubic_layout), float(bg), float(nearest),
                                    float(extent), sink)

    def render_dynamic_sets(self, clock, times, extr: Tensor, sets, *, position: Tensor, pos_cubic_node: Tensor,
                            rotation: Tensor, rot_poly_feat: Tensor, rot_fourier_feat: Tensor, opacity: Tensor, scaling: Tensor,
                            cubic_layout: int = 1, K: int = 0, nearest: float = 0.01, extent: float = 1.3,
                            grad_sink: Optional[Dict[str, Tensor]] = None):
        """The reference's real training frame, for all frames of the batch: its dynamic Gaussians (``render_dynamic``)
        through the three blends of ``render_iter`` (``render_sets``: same ``sets`` list, same return value).  The sets must
        fit the one-pass backward (one set per routing group, <= 4 / 4 / 20 channels).

        A set's ``feature`` may also be a LIST of tensors -- the set is their concatenation along the channels, as the
        reference's renderer concatenates its attributes (RenderFeatures.combine; ``["track_gs"] + render_attributes``,
        src/trainer_fragGS.py:511) -- and a tensor of a list may be per frame, ``[F, P, c]`` (track_gs = position(ids2) differs
        from frame to frame; a strided view with dense rows is read in place): no ``[F, P, C]`` row is materialised, a shared
        tensor's gradient is the sum over the frames, a per-frame tensor gets every frame's own.  ``grad_sink["feature:k"]``
        (k = position of the tensor in the flattened list of all sets' tensors) receives that gradient by ADDITION instead of
        autograd.  Any one-pass plan takes lists / per-frame tensors: the renderer's own plan (rgb 3 | depth 1 | 19 attribute
        channels) stages the forward's records in the tile kernel, other plans repack them in one launch."""
        tab = self.frame_table(clock, times)
        meta, parts, feats = _parse_sets(sets, self.F, self.P)
        widths = [1 if m[0] == "depth" else m[0] for m in meta]
        if sum(widths) != self.C:
            raise ValueError(f"the sets hold {sum(widths)} channels, the batch was built for C = {self.C}")
        plan = _one_pass_plan(meta, widths, self.C)
        if plan is None:
            raise ValueError("render_dynamic_sets needs sets that fit the one-pass backward: one set per routing group "
                             "(taps / live opacity / detached opacity) of at most 4 / 4 / 20 channels")
        fsink = {k: v for k, v in (grad_sink or {}).items() if k.startswith("feature:")}
        psink = {k: v for k, v in (grad_sink or {}).items() if not k.startswith("feature:")}
        sink = check_sink(psink, {"position": position, "pos_cubic_node": pos_cubic_node, "rotation": rotation,
                                  "opacity": opacity, "scaling": scaling})
        for k, buf in fsink.items():
            i = int(k.split(":")[1])
            if not (0 <= i < len(feats)) or tuple(buf.shape) != tuple(feats[i].shape) or buf.dtype != torch.float32 \
                    or not buf.is_cuda or buf.stride(-1) != 1 or buf.stride(-2) != buf.shape[-1]:
                raise ValueError(f"grad_sink[{k!r}] must be a float32 GPU buffer of the feature tensor's shape with dense rows")
        return _RenderDynamicSets.apply(position, pos_cubic_node, rotation, opacity, scaling, rot_poly_feat, rot_fourier_feat, extr,
                                        tab, self, int(clock.interval_num), int(cubic_layout), meta, int(K), float(nearest),
                                        float(extent), sink, parts, fsink or None, *feats)

    # ------------------------------------------------------------------ several feature sets of one geometry (row a1)
    def render_sets(self, xyz: Tensor, scales: Tensor, uquats: Tensor, opacity: Tensor, sets, offsets: Optional[Tensor],
                    extr: Tensor, K: int = 0, nearest: float = 0.01, extent: float = 1.3,
                    grad_sink: Optional[Dict[str, Tensor]] = None, intr: Optional[Tensor] = None):
        """The reference renderer's blends of ONE geometry over all frames of the batch (render_iter,
        src/pointrix/renderer/dptr_ortho_enhanced.py:331-375: rgb through alpha_blending_enhanced with the taps; depth with
        bg = 1; the extra attributes with opacity.detach()).  ``sets``: a list of dicts ``feature`` ([P,c] tensor shared by
        the frames, or the string "depth" for the per-frame depth of the projection), ``bg``, ``detach_opacity``, ``taps``.
        The widths must add up to the batch's ``C``.  One forward pass composites the concatenated row; per set one
        backward pass of the tile kernels and one Gaussian-side reduction.  Returns ``(images per set ..., gs_idx)`` with
        images [F,c,H,W] and gs_idx [F,H,W,K] (None for K = 0).  Cameras (``extr`` per frame, ``intr``) as in ``render``."""
        def one(f):
            # a list of shared tensors = their concatenation along the channels, as the reference's renderer concatenates its
            # attributes (RenderFeatures.combine); per-frame tensors belong to the dynamic renderer (render_dynamic_sets)
            if isinstance(f, (list, tuple)):
                if any(t.dim() != 2 for t in f):
                    raise ValueError("render_sets: the tensors of a feature list are [P, c] (per-frame tensors: render_dynamic_sets)")
                return f[0] if len(f) == 1 else torch.cat(list(f), dim=1)
            return f
        sets = [dict(s_, feature=one(s_["feature"])) for s_ in sets]
        feats = [s_["feature"] for s_ in sets if not isinstance(s_["feature"], str)]
        meta = tuple((("depth" if isinstance(s_["feature"], str) else int(s_["feature"].shape[1])), float(s_.get("bg", 0.0)),
                      bool(s_.get("detach_opacity", False)), bool(s_.get("taps", False))) for s_ in sets)
        width = sum(1 if m[0] == "depth" else m[0] for m in meta)
        if width != self.C:
            raise ValueError(f"the sets hold {width} channels, the batch was built for C = {self.C}")
        if sum(1 for m in meta if m[3]) > 1:
            raise ValueError("at most one set feeds the densification taps")
        sink = check_sink(grad_sink, {"xyz": xyz, "scales": scales, "uquats": uquats, "opacity": opacity})
        res = _RenderSets.apply(xyz, scales, uquats, opacity, offsets, extr, self, meta, int(K), float(nearest), float(extent),
                                sink, intr, *feats)
        return res


class _RenderDynamic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, position, cubic, rotation, opacity, scaling, feature, rot_poly, rot_fourier, extr, tab, fb, I, layout, bg,
                nearest, extent, sink):
        P, F = fb.P, fb.F
        position = _points(position, "position", 3)
        rotation = _points(rotation, "rotation", 4)
        scaling = _points(scaling, "scaling", 3)
        opacity = L.need(opacity, "opacity")
        cubic = L.need(cubic, "pos_cubic_node")
        feature = _points(feature, "feature", fb.C)
        rot_poly, rot_fourier = L.need(rot_poly, "rot_poly_feat"), L.need(rot_fourier, "rot_fourier_feat")
        if cubic.numel() != P * 4 * I * 3 or rot_poly.numel() != P * 16 or rot_fourier.numel() != P * 32 or opacity.numel() != P:
            raise ValueError("parameter shapes do not match the batch (P Gaussians, I spline segments)")
        if position.shape[0] != P or rotation.shape[0] != P or scaling.shape[0] != P or feature.shape[0] != P:
            raise ValueError(f"the batch was built for {P} Gaussians")
        extr_c = _extr12(extr)
        lib, st = L.lib(), L.stream()
        W, H, C = fb.W, fb.H, fb.C
        ctx.gen = fb._begin_forward()
        opa_t = torch.empty(P, 1, dtype=torch.float32, device=fb.dev)
        L.check(lib.splat_frame_preprocess_forward_batch(
            L.ci(F), L.ci(P), L.ci(I), L.ptr(tab), L.ptr(position), L.ptr(cubic), L.ci(layout), L.ptr(rotation), L.ptr(rot_poly),
            L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c), L.ci(W), L.ci(H), L.cf(nearest), L.cf(extent),
            L.ptr(fb.uv), L.ptr(fb.depth), L.ptr(fb.conic), L.ptr(fb.radius), L.ptr(opa_t), st))
        fb._bin_and_sort(opa_t)
        cap = fb.capacity
        out = torch.empty(F, C, H, W, dtype=torch.float32, device=fb.dev)
        L.check(lib.splat_alpha_blending_forward_batch(
            L.ci(F), L.ci(P), L.ci(C), L.ptr(fb.uv), L.ptr(fb.conic), L.ptr(opa_t), ctypes.c_int64(0), L.ptr(feature),
            ctypes.c_int64(0), L.ptr(fb.idx_sorted), L.ptr(fb.tile_range), ctypes.c_int64(cap), L.cf(bg), L.ptr(None), L.ci(W),
            L.ci(H), L.ci(0), L.ci(0), L.ptr(out), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(None), L.ptr(fb.pack),
            L.ptr(fb.cull_flags), st))
        ctx.fb, ctx.meta, ctx.sink = fb, (I, layout, bg), sink
        ctx.save_for_backward(position, cubic, rotation, opacity, scaling, feature, rot_poly, rot_fourier, extr_c, tab)
        return out

    @staticmethod
    def backward(ctx, dL_dout):
        fb: FrameBatch = ctx.fb
        fb._check_generation(ctx.gen)
        position, cubic, rotation, opacity, scaling, feature, rot_poly, rot_fourier, extr_c, tab = ctx.saved_tensors
        I, layout, bg = ctx.meta
        sink = ctx.sink or {}
        g = L.need(dL_dout, "dL_dout")
        lib, st = L.lib(), L.stream()
        F, P, W, H, C, cap = fb.F, fb.P, fb.W, fb.H, fb.C, fb.capacity
        from .gs.raster_ops import _debug_T_front
        L.check(lib.splat_alpha_blending_backward_batch(
            L.ci(F), L.ci(P), L.ci(C), L.ptr(fb.idx_sorted), L.ptr(fb.tile_range), ctypes.c_int64(cap), L.cf(bg), L.ci(W), L.ci(H),
            L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(g), L.ci(1 if fb.want_abs else 0), L.ptr(fb.slot_sorted),
            L.ptr(fb.pair_records), L.ptr(fb.pack), L.ptr(fb.cull_flags), L.ptr(_debug_T_front(F * H, W, g.device)), st))
        like = {"position": position, "pos_cubic_node": cubic, "rotation": rotation, "opacity": opacity, "scaling": scaling,
                "feature": feature}
        need = dict(zip(like, ctx.needs_input_grad[:6]))
        bufs = {k: (sink[k] if k in sink else (torch.zeros_like(v) if need[k] else None)) for k, v in like.items()}
        L.check(lib.splat_frames_gauss_backward_dynamic(
            L.ci(F), L.ci(P), L.ci(I), L.ci(C), L.ci(W), L.ci(H), ctypes.c_int64(cap), L.ci(1 if fb.want_abs else 0),
            L.ptr(fb.pair_records), L.ptr(fb.goff), L.ptr(fb.radius), L.ptr(tab), L.ptr(position), L.ptr(cubic), L.ci(layout),
            L.ptr(rotation), L.ptr(rot_poly), L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c),
            L.ptr(bufs["position"]), L.ptr(bufs["pos_cubic_node"]), L.ptr(bufs["rotation"]), L.ptr(bufs["opacity"]),
            L.ptr(bufs["scaling"]), L.ptr(bufs["feature"]), L.ptr(fb.tap), L.ptr(fb.abs_tap), L.ptr(fb.radii_max), st))
        ret = tuple(None if (k in sink or bufs[k] is None) else bufs[k] for k in like)
        return ret + (None,) * 11


class _RenderDynamicSets(torch.autograd.Function):
    N_FIXED = 19      # arguments in front of the feature tensors

    @staticmethod
    def forward(ctx, position, cubic, rotation, opacity, scaling, rot_poly, rot_fourier, extr, tab, fb, I, layout, meta, K,
                nearest, extent, sink, parts, fsink, *feats):
        P, F = fb.P, fb.F
        position = _points(position, "position", 3)
        rotation = _points(rotation, "rotation", 4)
        scaling = _points(scaling, "scaling", 3)
        opacity = L.need(opacity, "opacity")
        cubic = L.need(cubic, "pos_cubic_node")
        rot_poly, rot_fourier = L.need(rot_poly, "rot_poly_feat"), L.need(rot_fourier, "rot_fourier_feat")
        if cubic.numel() != P * 4 * I * 3 or rot_poly.numel() != P * 16 or rot_fourier.numel() != P * 32 or opacity.numel() != P:
            raise ValueError("parameter shapes do not match the batch (P Gaussians, I spline segments)")
        if position.shape[0] != P or rotation.shape[0] != P or scaling.shape[0] != P:
            raise ValueError(f"the batch was built for {P} Gaussians")
        sources = _has_sources(parts)
        if not sources:
            _check_set_features(meta, feats, P)
        extr_c = _extr12(extr)
        lib, st = L.lib(), L.stream()
        W, H, C = fb.W, fb.H, fb.C
        ctx.gen = fb._begin_forward()
        opa_t = torch.empty(P, 1, dtype=torch.float32, device=fb.dev)
        L.check(lib.splat_frame_preprocess_forward_batch(
            L.ci(F), L.ci(P), L.ci(I), L.ptr(tab), L.ptr(position), L.ptr(cubic), L.ci(layout), L.ptr(rotation), L.ptr(rot_poly),
            L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c), L.ci(W), L.ci(H), L.cf(nearest), L.cf(extent),
            L.ptr(fb.uv), L.ptr(fb.depth), L.ptr(fb.conic), L.ptr(fb.radius), L.ptr(opa_t), st))
        fb._bin_and_sort(opa_t)
        if sources:
            feats = tuple(_source_tensor(t, pf, F, P) for t, (_, pf) in zip(feats, [p_ for pp in parts for p_ in pp]))
            out, gs_idx, ctx.blend = _blend_sources_forward(fb, meta, parts, feats, opa_t, K)
        else:
            out, gs_idx, ctx.blend = _blend_sets_forward(fb, meta, feats, opa_t, 0, K)
        if ctx.blend["plan"] is None:
            raise ValueError("render_dynamic_sets needs sets that fit the one-pass backward")
        ctx.fb, ctx.meta, ctx.sink, ctx.geo = fb, meta, sink, (I, layout)
        ctx.parts, ctx.fsink, ctx.sources = parts, (fsink or {}), sources
        ctx.opa_t = opa_t
        ctx.save_for_backward(position, cubic, rotation, opacity, scaling, rot_poly, rot_fourier, extr_c, tab, *feats)
        ctx.set_materialize_grads(False)
        imgs, c0 = [], 0
        for w, _, _, _ in meta:
            w = 1 if w == "depth" else w
            imgs.append(out[:, c0:c0 + w])
            c0 += w
        if gs_idx is not None:
            ctx.mark_non_differentiable(gs_idx)
        return tuple(imgs) + (gs_idx,)

    @staticmethod
    def backward(ctx, *grads):
        fb: FrameBatch = ctx.fb
        fb._check_generation(ctx.gen)
        position, cubic, rotation, opacity, scaling, rot_poly, rot_fourier, extr_c, tab = ctx.saved_tensors[:9]
        feats = ctx.saved_tensors[9:]
        meta, sink = ctx.meta, (ctx.sink or {})
        I, layout = ctx.geo
        lib, st = L.lib(), L.stream()
        F, P, W, H, C, cap = fb.F, fb.P, fb.W, fb.H, fb.C, fb.capacity
        dev = fb.dev
        widths = ctx.blend["widths"]
        like = {"position": position, "pos_cubic_node": cubic, "rotation": rotation, "opacity": opacity, "scaling": scaling}
        bufs = {k: (sink[k] if k in sink else torch.zeros_like(v)) for k, v in like.items()}
        c0s, cns, bgs, depth_ch, tap_set = _one_pass_plan(meta, widths, C)
        want_abs = 1 if (tap_set is not None and fb.want_abs) else 0
        NF = _RenderDynamicSets.N_FIXED
        has_tap = tap_set is not None
        if ctx.sources:
            # the row by SOURCES: a shared tensor's gradient is the sum over the frames, a per-frame tensor gets every frame's own
            rec = _blend_sets_backward_one_pass(fb, meta, ctx.blend, grads[:len(meta)], ctx.opa_t, 0, want_abs)
            table = (L.FeatureSource * L.MAX_SOURCES)()
            dfe, n, fi, c0 = [], 0, 0, 0
            for si, ((w, _, _, _), pp) in enumerate(zip(meta, ctx.parts)):
                if w == "depth":
                    c0 += 1
                    continue
                for cn, per_frame in pp:
                    t = feats[fi]
                    key = f"feature:{fi}"
                    need = grads[si] is not None and (ctx.needs_input_grad[NF + fi] or key in ctx.fsink)
                    buf = ctx.fsink.get(key)
                    ret = None
                    if need and buf is None:
                        buf = ret = torch.zeros(t.shape, dtype=torch.float32, device=dev)
                    dfe.append(ret)
                    if need:
                        table[n].c0, table[n].cn, table[n].feature = c0, cn, t.data_ptr()
                        table[n].d_feature = buf.data_ptr()
                        table[n].frame_stride = int(buf.stride(0)) if per_frame else 0
                        n += 1
                    c0 += cn
                    fi += 1
            L.check(lib.splat_frames_gauss_backward_dynamic_sources(
                L.ci(F), L.ci(P), L.ci(I), L.ci(C), L.ci(W), L.ci(H), ctypes.c_int64(cap), L.ptr(rec), L.ptr(fb.goff),
                L.ptr(fb.radius), L.ptr(tab), L.ptr(position), L.ptr(cubic), L.ci(layout), L.ptr(rotation), L.ptr(rot_poly),
                L.ptr(rot_fourier), L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c), L.ptr(bufs["position"]),
                L.ptr(bufs["pos_cubic_node"]), L.ptr(bufs["rotation"]), L.ptr(bufs["opacity"]), L.ptr(bufs["scaling"]), L.ci(n),
                table, L.ci(depth_ch), L.ptr(fb.tap if has_tap else None),
                L.ptr(fb.abs_tap if (has_tap and want_abs) else None), L.ptr(fb.radii_max if has_tap else None), st))
            ret = tuple(None if k in sink else bufs[k] for k in like)
            return ret + (None,) * (NF - 5) + tuple(dfe)
        group_of = _set_groups(meta)
        fi, dfe, dfs, strides = 0, [], [None, None, None], [0, 0, 0]
        for si, (w, _, _, _) in enumerate(meta):
            if w == "depth":
                continue
            need = grads[si] is not None and ctx.needs_input_grad[NF + fi]
            dfeat = torch.zeros_like(feats[fi]) if need else None
            dfe.append(dfeat)
            dfs[group_of[si]], strides[group_of[si]] = dfeat, int(feats[fi].shape[1])
            fi += 1
        rec = _blend_sets_backward_one_pass(fb, meta, ctx.blend, grads[:len(meta)], ctx.opa_t, 0, want_abs)
        i3 = ctypes.c_int32 * 3
        p3 = (ctypes.c_void_p * 3)(*[0 if d is None else d.data_ptr() for d in dfs])
        L.check(lib.splat_frames_gauss_backward_dynamic_sets(
            L.ci(F), L.ci(P), L.ci(I), L.ci(C), L.ci(W), L.ci(H), ctypes.c_int64(cap), L.ptr(rec), L.ptr(fb.goff), L.ptr(fb.radius),
            L.ptr(tab), L.ptr(position), L.ptr(cubic), L.ci(layout), L.ptr(rotation), L.ptr(rot_poly), L.ptr(rot_fourier),
            L.ptr(opacity), L.ptr(scaling), L.ptr(extr_c), L.ptr(bufs["position"]), L.ptr(bufs["pos_cubic_node"]),
            L.ptr(bufs["rotation"]), L.ptr(bufs["opacity"]), L.ptr(bufs["scaling"]), i3(*c0s), i3(*cns), p3, i3(*strides),
            L.ci(depth_ch), L.ptr(fb.tap if has_tap else None), L.ptr(fb.abs_tap if (has_tap and want_abs) else None),
            L.ptr(fb.radii_max if has_tap else None), st))
        ret = tuple(None if k in sink else bufs[k] for k in like)
        return ret + (None,) * (NF - 5) + tuple(dfe)


def _parse_sets(sets, F, P):
    """``sets`` -> (meta: per set (width | "depth", bg, detach_opacity, taps); parts: per set the (channels, per_frame) of every
    tensor its feature is made of -- () for the depth; the flat list of those tensors)"""
    meta, parts, feats = [], [], []
    for s_ in sets:
        f = s_["feature"]
        common = (float(s_.get("bg", 0.0)), bool(s_.get("detach_opacity", False)), bool(s_.get("taps", False)))
        if isinstance(f, str):
            if f != "depth":
                raise ValueError('a set\'s feature is a tensor, a list of tensors or the string "depth"')
            meta.append(("depth",) + common)
            parts.append(())
            continue
        pp = []
        for t in (list(f) if isinstance(f, (list, tuple)) else [f]):
            if not isinstance(t, Tensor) or t.dim() not in (2, 3):
                raise ValueError("a feature tensor is [P, c] (shared by the frames) or [F, P, c] (one row set per frame)")
            if t.dim() == 3 and (t.shape[0] != F or t.shape[1] != P):
                raise ValueError(f"a per-frame feature tensor must be [F={F}, P={P}, c], got {tuple(t.shape)}")
            if t.dim() == 2 and t.shape[0] != P:
                raise ValueError(f"a feature tensor must be [P={P}, c], got {tuple(t.shape)}")
            pp.append((int(t.shape[-1]), t.dim() == 3))
            feats.append(t)
        meta.append((sum(c for c, _ in pp),) + common)
        parts.append(tuple(pp))
    return tuple(meta), tuple(parts), feats


def _has_sources(parts) -> bool:
    """does any set consist of several tensors or of a per-frame tensor? (then the row is described by sources)"""
    return any(len(pp) > 1 or (len(pp) == 1 and pp[0][1]) for pp in parts)


def _source_tensor(t: Tensor, per_frame: bool, F: int, P: int) -> Tensor:
    """a source the kernels read in place: float32 on the device, dense rows; a per-frame tensor may be a strided view
    (frame stride >= P * c), anything else is made contiguous"""
    if not per_frame:
        return L.need(t, "feature")
    cn = t.shape[2]
    if t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) == cn and t.stride(0) >= P * cn:
        L.need(t[0], "feature")      # (device gate; a frame's rows are contiguous)
        return t
    return L.need(t, "feature")


def _blend_sources_forward(fb, meta, parts, feats, opacity, K):
    """Forward compositing of a row described by SOURCES (splat_alpha_blending_forward_batch_sources): every tensor of every set
    is read in place -- shared rows [P, c], per-frame rows [F, P, c] (any frame stride), the per-frame depth."""
    lib, st = L.lib(), L.stream()
    F, P, W, H, C, cap = fb.F, fb.P, fb.W, fb.H, fb.C, fb.capacity
    widths = [1 if m[0] == "depth" else m[0] for m in meta]
    plan = _one_pass_plan(meta, widths, C)
    cache = fb.__dict__.setdefault("_bgc", {})
    bgc = cache.get(meta)
    if bgc is None:
        bgc = torch.tensor([bg for (w, bg, _, _), n in zip(meta, widths) for _ in range(n)], dtype=torch.float32, device=fb.dev)
        cache[meta] = bgc
    nsrc = sum(1 if w == "depth" else len(pp) for (w, _, _, _), pp in zip(meta, parts))    # the depth counts as a source
    if nsrc > L.MAX_SOURCES:
        raise ValueError(f"at most {L.MAX_SOURCES} feature sources per row (the depth counts as one), got {nsrc}")
    table = (L.FeatureSource * L.MAX_SOURCES)()
    n, c0, it = 0, 0, iter(feats)
    for (w, _, _, _), pp in zip(meta, parts):
        if w == "depth":
            table[n].c0, table[n].cn, table[n].feature, table[n].frame_stride = c0, 1, fb.depth.data_ptr(), P
            n += 1
            c0 += 1
            continue
        for cn, per_frame in pp:
            t = next(it)
            table[n].c0, table[n].cn, table[n].feature = c0, cn, t.data_ptr()
            table[n].frame_stride = int(t.stride(0)) if per_frame else 0
            n += 1
            c0 += cn
    out = torch.empty(F, C, H, W, dtype=torch.float32, device=fb.dev)
    gs_idx = torch.empty(F, H, W, K, dtype=torch.int32, device=fb.dev) if K > 0 else None
    L.check(lib.splat_alpha_blending_forward_batch_sources(
        L.ci(F), L.ci(P), L.ci(C), L.ci(n), table, L.ptr(fb.uv), L.ptr(fb.conic), L.ptr(opacity), ctypes.c_int64(0),
        L.ptr(fb.idx_sorted), L.ptr(fb.tile_range), ctypes.c_int64(cap), L.ptr(bgc), L.ci(W), L.ci(H), L.ci(K), L.ci(0),
        L.ptr(out), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(gs_idx), L.ptr(fb.pack), L.ptr(fb.cull_flags), st))
    # (the forward-pack decision is taken HERE and kept in the autograd context: an option flipped between this forward and its
    #  backward must not change what the backward stages)
    return out, gs_idx, dict(plan=plan, tens=None, row=None, widths=widths, std=_uses_forward_pack(plan, C), out_row=out)


def _blend_sets_forward(fb, meta, feats, opacity, op_fs, K):
    """Forward compositing of the sets' row for all frames.  When the sets fit the one-pass backward every set is read from its
    own tensor (shared features [P,c], the per-frame depth [F,P,1]) -- no [F,P,C] row is materialised; otherwise the row is
    concatenated and the per-set passes use it.  Returns (out [F,C,H,W], gs_idx, state for _blend_sets_backward)."""
    lib, st = L.lib(), L.stream()
    F, P, W, H, C, cap = fb.F, fb.P, fb.W, fb.H, fb.C, fb.capacity
    widths = [1 if m[0] == "depth" else m[0] for m in meta]
    one_pass_wanted = OPTIONS["sets_one_pass"]
    plan = _one_pass_plan(meta, widths, C) if one_pass_wanted else None
    if plan is None and one_pass_wanted and not fb.__dict__.get("_warned_per_set"):
        fb.__dict__["_warned_per_set"] = True   # (once per batch object)
        warnings.warn("FrameBatch.render_sets: these feature sets do not fit the one-pass backward (one set per routing group -- "
                      "taps / live opacity / detached opacity -- of at most 4 / 4 / 20 channels): the backward runs one pass of "
                      "the tile kernels per set", RuntimeWarning, stacklevel=3)
    tens, it = [], iter(feats)
    for w, _, _, _ in meta:
        tens.append(fb.depth if w == "depth" else _points(next(it), "feature", w))
    cache = fb.__dict__.setdefault("_bgc", {})
    bgc = cache.get(meta)
    if bgc is None:
        bgc = torch.tensor([bg for (w, bg, _, _), n in zip(meta, widths) for _ in range(n)], dtype=torch.float32, device=fb.dev)
        cache[meta] = bgc
    out = torch.empty(F, C, H, W, dtype=torch.float32, device=fb.dev)
    gs_idx = torch.empty(F, H, W, K, dtype=torch.int32, device=fb.dev) if K > 0 else None
    row = None
    if plan is None:
        row = torch.cat([t if m[0] == "depth" else t.unsqueeze(0).expand(F, P, t.shape[1]) for t, m in zip(tens, meta)],
                        dim=2).contiguous()
        L.check(lib.splat_alpha_blending_forward_batch(
            L.ci(F), L.ci(P), L.ci(C), L.ptr(fb.uv), L.ptr(fb.conic), L.ptr(opacity), ctypes.c_int64(op_fs), L.ptr(row),
            ctypes.c_int64(P * C), L.ptr(fb.idx_sorted), L.ptr(fb.tile_range), ctypes.c_int64(cap), L.cf(0.0), L.ptr(bgc),
            L.ci(W), L.ci(H), L.ci(K), L.ci(0), L.ptr(out), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(gs_idx),
            L.ptr(fb.pack), L.ptr(fb.cull_flags), st))
    else:
        tabs = _set_tables(meta, plan, tens, P)
        L.check(lib.splat_alpha_blending_forward_batch_sets(
            L.ci(F), L.ci(P), L.ci(C), tabs["c0"], tabs["cn"], tabs["feat"], tabs["fs"], L.ptr(fb.uv), L.ptr(fb.conic),
            L.ptr(opacity), ctypes.c_int64(op_fs), L.ptr(fb.idx_sorted), L.ptr(fb.tile_range), ctypes.c_int64(cap), L.ptr(bgc),
            L.ci(W), L.ci(H), L.ci(K), L.ci(0), L.ptr(out), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(gs_idx),
            L.ptr(fb.pack), L.ptr(fb.cull_flags), st))
    return out, gs_idx, dict(plan=plan, tens=tens, row=row, widths=widths, std=_uses_forward_pack(plan, C), out_row=out)


def _set_tables(meta, plan, tens, P):
    """ctypes tables (three routing groups) of a one-pass plan: first row channel, width, feature pointer, frame stride"""
    c0s, cns, bgs, _, _ = plan
    groups = _set_groups(meta)
    ptr, fs = [0, 0, 0], [0, 0, 0]
    for (w, _, _, _), t, g in zip(meta, tens or [None] * len(meta), groups):
        ptr[g] = 0 if t is None else t.data_ptr()   # (None: a row by sources -- its backward stages the forward's records)
        fs[g] = P if w == "depth" else 0        # the depth [F,P,1] is per frame, feature rows are shared
    i3, f3 = ctypes.c_int32 * 3, ctypes.c_float * 3
    return dict(c0=i3(*c0s), cn=i3(*cns), bg=f3(*bgs), feat=(ctypes.c_void_p * 3)(*ptr), fs=(ctypes.c_int64 * 3)(*fs))


def _blend_sets_backward_one_pass(fb, meta, state, grads, opacity, op_fs, want_abs):
    """ONE pass of the tile kernels for the three sets (splat_alpha_blending_backward_batch_sets): every set's image gradient
    from its own tensor.  Returns the pair-record buffer.  With ``fb.fuse_l1(...)`` armed the incoming gradient tensors are
    placeholders: the kernel derives the L1 loss' gradient from the forward's output row and the target images."""
    lib, st = L.lib(), L.stream()
    F, P, W, H, C, cap = fb.F, fb.P, fb.W, fb.H, fb.C, fb.capacity
    plan, tens = state["plan"], state["tens"]
    tabs = _set_tables(meta, plan, tens, P)
    groups = _set_groups(meta)
    dl = [0, 0, 0]
    keep = []
    l1 = fb.__dict__.pop("_l1", None)
    if l1 is not None:
        from .gs.raster_ops import _debug_T_front
        row = state.get("out_row")
        if row is None or tuple(row.shape) != (F, C, H, W) or not L.get_option("bwd_quarters"):
            raise ValueError("the loss-fused backward needs the forward's output row and the quarter-list kernels")
        scale = [0.0, 0.0, 0.0]
        for tgt, wgt, w, grp in zip(l1["targets"], l1["weights"], state["widths"], groups):
            t = L.need(tgt, "target image")
            if tuple(t.shape) != (F, w, H, W):
                raise ValueError(f"a set's target image must be [F={F}, {w}, H, W]")
            keep.append(t)
            dl[grp] = t.data_ptr()
            scale[grp] = float(wgt) / (F * w * H * W)          # d (weight * mean |pred - target|) / d pred
        sums = L.need(l1["sums"], "l1 sums")
        if sums.numel() != 3 * F * fb.T:
            raise ValueError(f"l1 sums must be [F={F}, tiles={fb.T}, 3]")
        rec = fb._set_buffer(("rec", "sets"), F * cap * int(lib.splat_blend_sets_pair_stride(C)))
        std = state["std"] if "std" in state else _uses_forward_pack(plan, C)
        pack = None if std else fb._set_buffer(("pack", "sets"), F * P * int(lib.splat_blend_sets_pack_floats()))
        L.check(lib.splat_alpha_blending_backward_batch_sets_l1(
            L.ci(F), L.ci(P), L.ci(C), tabs["c0"], tabs["cn"], tabs["bg"], L.ptr(fb.uv), L.ptr(fb.conic), L.ptr(opacity),
            ctypes.c_int64(op_fs), tabs["feat"], tabs["fs"], L.ptr(fb.idx_sorted), L.ptr(fb.tile_range), ctypes.c_int64(cap),
            L.ci(W), L.ci(H), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(row), (ctypes.c_void_p * 3)(*dl),
            (ctypes.c_float * 3)(*scale), L.ptr(sums), L.ci(want_abs), L.ptr(fb.slot_sorted), L.ptr(rec), L.ptr(pack),
            L.ptr(fb.cull_flags), L.ptr(_debug_T_front(F * H, W, fb.dev)), L.ptr(fb.pack if (std or tens is None) else None), st))
        # the slab order of the three groups: (tap set, second set, detached set) -> set index of the caller's list
        l1["group_of_set"] = list(groups)
        return rec
    for g_, w, grp in zip(grads, state["widths"], groups):
        t = L.need(g_, "dL_dout") if g_ is not None else torch.zeros(F, w, H, W, dtype=torch.float32, device=fb.dev)
        if tuple(t.shape) != (F, w, H, W):
            raise ValueError("image gradient of a set must be [F, c, H, W]")
        keep.append(t)
        dl[grp] = t.data_ptr()
    from .gs.raster_ops import _debug_T_front
    rec = fb._set_buffer(("rec", "sets"), F * cap * int(lib.splat_blend_sets_pair_stride(C)))
    # the renderer's own plan (rgb 0-2 with the taps | depth 3 | 19 detached attributes 4-22): the tile kernel stages the records
    # the FORWARD packed (fb.pack) -- no packing launch, no second record array; other plans pack their own (a row described by
    # sources -- tens is None -- out of the forward's records, which hold the row's channels)
    std = state["std"] if "std" in state else _uses_forward_pack(plan, C)     # decided at forward time
    pack = None if std else fb._set_buffer(("pack", "sets"), F * P * int(lib.splat_blend_sets_pack_floats()))
    L.check(lib.splat_alpha_blending_backward_batch_sets_packed(
        L.ci(F), L.ci(P), L.ci(C), tabs["c0"], tabs["cn"], tabs["bg"], L.ptr(fb.uv), L.ptr(fb.conic), L.ptr(opacity),
        ctypes.c_int64(op_fs), L.ptr(None), ctypes.c_int64(0), tabs["feat"], tabs["fs"], L.ptr(fb.idx_sorted),
        L.ptr(fb.tile_range), ctypes.c_int64(cap), L.ci(W), L.ci(H), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(None),
        (ctypes.c_void_p * 3)(*dl), L.ci(want_abs), L.ptr(fb.slot_sorted), L.ptr(rec), L.ptr(pack), L.ptr(fb.cull_flags),
        L.ptr(_debug_T_front(F * H, W, fb.dev)), L.ptr(fb.pack if (std or tens is None) else None), st))
    return rec


def _uses_forward_pack(plan, C) -> bool:
    """does the one-pass backward of this plan stage the records the FORWARD packed (no packing launch, no pack scratch)?  The C
    library owns the condition (splat_blend_sets_uses_forward_pack: the renderer's own plan with the cull words)."""
    from .gs.raster_ops import OPTIONS as RO
    if plan is None or not RO["sets_fwdrec"]:
        return False
    i3 = ctypes.c_int32 * 3
    return bool(L.lib().splat_blend_sets_uses_forward_pack(L.ci(C), i3(*plan[0]), i3(*plan[1]), L.ci(1)))


def _check_set_features(meta, feats, P):
    """every shared feature tensor of the sets is [P, c] with the width its set declares (the kernels index rows 0 .. P-1)"""
    it = iter(feats)
    for w, _, _, _ in meta:
        if w == "depth":
            continue
        t = next(it)
        if t.dim() != 2 or t.shape[0] != P or t.shape[1] != w:
            raise ValueError(f"a set's feature must be [P={P}, {w}], got {tuple(t.shape)}")


def _set_groups(meta):
    """Routing group of every set: 0 = feeds the taps, 1 = live opacity without taps, 2 = opacity.detach()."""
    return [0 if taps else (2 if detach else 1) for (_, _, detach, taps) in meta]


def _one_pass_plan(meta, widths, C):
    """Can ONE pass of splat_alpha_blending_backward_batch_sets serve these sets?  At most one set per routing group, the
    tap set blended with the live opacity, widths <= (4, 4, 20), C <= 28.  Returns (c0[3], cn[3], bg[3], depth row channel
    or -1, index of the tap set or None), or None: the per-set passes run."""
    if C > 28:
        return None
    groups = _set_groups(meta)
    if len(set(groups)) != len(groups):
        return None
    c0s, cns, bgs, depth_ch, tap_set = [0, 0, 0], [0, 0, 0], [0.0, 0.0, 0.0], -1, None
    c0 = 0
    for si, ((w, bg, detach, taps), cn, g) in enumerate(zip(meta, widths, groups)):
        if (taps and detach) or cn > (4, 4, 20)[g]:
            return None
        c0s[g], cns[g], bgs[g] = c0, cn, bg
        if w == "depth":
            depth_ch = c0
        if taps:
            tap_set = si
        c0 += cn
    return c0s, cns, bgs, depth_ch, tap_set


class _RenderSets(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scales, uquats, opacity, offsets, extr, fb, meta, K, nearest, extent, sink, intr, *feats):
        xyz = _points(xyz, "xyz", 3)
        scales = _points(scales, "scales", 3)
        uquats = _points(uquats, "uquats", 4)
        opacity = L.need(opacity, "opacity")
        P, F = fb.P, fb.F
        cam = _Camera(extr, intr, F)
        if xyz.shape[0] != P or scales.shape[0] != P or uquats.shape[0] != P:
            raise ValueError(f"the batch was built for {P} Gaussians")
        if opacity.numel() != P:
            # (the Gaussian-side backward sums the opacity gradient over the frames: a per-frame opacity [F,P,1] has no entry point)
            raise ValueError(f"opacity must hold one value per Gaussian ({P}), shared by the frames; got {tuple(opacity.shape)}")
        _check_set_features(meta, feats, P)
        off = L.need(offsets, "offsets") if offsets is not None else None
        if off is not None and tuple(off.shape) != (F, P, 3):
            raise ValueError(f"offsets must be [F={F}, P={P}, 3]")
        if off is None and F > 1 and cam.extr_fs == 0:
            raise ValueError("several frames of static Gaussians need per-frame offsets or per-frame cameras")
        ctx.gen = fb._begin_forward()
        fb._geometry(xyz, scales, uquats, off, cam, nearest, extent, opacity)
        op_fs = 0
        C = fb.C
        out, gs_idx, ctx.blend = _blend_sets_forward(fb, meta, feats, opacity, op_fs, K)
        ctx.fb, ctx.meta, ctx.sink, ctx.K = fb, meta, sink, K
        ctx.cam, ctx.off = cam, off
        ctx.cam_versions = _versions(cam.extr, cam.intr, off)
        ctx.save_for_backward(xyz, scales, uquats, opacity, *feats)
        ctx.set_materialize_grads(False)
        imgs, c0 = [], 0
        for w, _, _, _ in meta:
            w = 1 if w == "depth" else w
            imgs.append(out[:, c0:c0 + w])
            c0 += w
        if gs_idx is not None:
            ctx.mark_non_differentiable(gs_idx)
            return tuple(imgs) + (gs_idx,)
        return tuple(imgs) + (None,)

    @staticmethod
    def backward(ctx, *grads):
        fb: FrameBatch = ctx.fb
        fb._check_generation(ctx.gen)
        xyz, scales, uquats, opacity = ctx.saved_tensors[:4]
        feats = ctx.saved_tensors[4:]
        _check_versions(ctx.cam_versions, "a camera / offset tensor")
        camc = ctx.cam.struct(ctx.off)
        meta, sink = ctx.meta, (ctx.sink or {})
        lib, st = L.lib(), L.stream()
        F, P, W, H, C, cap = fb.F, fb.P, fb.W, fb.H, fb.C, fb.capacity
        dev = fb.dev
        widths = ctx.blend["widths"]
        like = {"xyz": xyz, "scales": scales, "uquats": uquats, "opacity": opacity}
        bufs = {k: (sink[k] if k in sink else torch.zeros_like(v)) for k, v in like.items()}   # every set accumulates
        dfe = []
        op_fs = 0
        from .gs.raster_ops import _debug_T_front
        plan = ctx.blend["plan"]
        if plan is not None:
            # the sets share the alpha / transmittance replay: ONE pass of the tile kernels and ONE Gaussian-side reduction
            # route dL/dalpha of every set where the reference's three autograd nodes would
            c0s, cns, bgs, depth_ch, tap_set = plan
            want_abs = 1 if (tap_set is not None and fb.want_abs) else 0
            fi, dfs, strides = 0, [None, None, None], [0, 0, 0]
            group_of = _set_groups(meta)
            for si, (w, _, _, _) in enumerate(meta):
                if w == "depth":
                    continue
                need = grads[si] is not None and ctx.needs_input_grad[13 + fi]
                dfeat = torch.zeros_like(feats[fi]) if need else None
                dfe.append(dfeat)
                dfs[group_of[si]], strides[group_of[si]] = dfeat, int(feats[fi].shape[1])
                fi += 1
            rec = _blend_sets_backward_one_pass(fb, meta, ctx.blend, grads[:len(meta)], opacity, op_fs, want_abs)
            i3 = ctypes.c_int32 * 3
            p3 = (ctypes.c_void_p * 3)(*[0 if d is None else d.data_ptr() for d in dfs])
            has_tap = tap_set is not None
            L.check(lib.splat_frames_gauss_backward_static_sets_cam(
                L.ci(F), L.ci(P), L.ci(C), L.ci(W), L.ci(H), ctypes.c_int64(cap), L.ptr(rec), L.ptr(fb.goff), L.ptr(fb.radius),
                L.ptr(xyz), L.ptr(scales), L.ptr(uquats), ctypes.byref(camc), L.ci(1), L.ptr(bufs["xyz"]), L.ptr(bufs["scales"]),
                L.ptr(bufs["uquats"]), L.ptr(bufs["opacity"]), i3(*c0s), i3(*cns), p3, i3(*strides), L.ci(depth_ch),
                L.ptr(fb.tap if has_tap else None), L.ptr(fb.abs_tap if (has_tap and want_abs) else None),
                L.ptr(fb.radii_max if has_tap else None), st))
            ret = tuple(None if k in sink else bufs[k] for k in ("xyz", "scales", "uquats", "opacity"))
            return ret + (None,) * 9 + tuple(dfe)
        row = ctx.blend["row"]
        dL = torch.cat([(g if g is not None else torch.zeros(F, w, H, W, dtype=torch.float32, device=dev))
                        for g, w in zip(grads[:len(meta)], widths)], dim=1).contiguous()
        c0, fi = 0, 0
        for si, ((w, bg, detach, taps), g) in enumerate(zip(meta, grads[:len(meta)])):
            cn = widths[si]
            is_depth = w == "depth"
            dfeat = None
            if not is_depth:
                dfeat = torch.zeros_like(feats[fi]) if g is not None and ctx.needs_input_grad[13 + fi] else None
                dfe.append(dfeat)
                fi += 1
            if g is not None:
                want_abs = 1 if (taps and fb.want_abs) else 0
                ncp = int(lib.splat_blend_pair_stride(cn, want_abs, 0))
                rec = fb._set_buffer(("rec", si), F * cap * ncp)
                pack = fb._set_buffer(("pack", si), F * P * int(lib.splat_blend_pack_floats(cn)))
                L.check(lib.splat_alpha_blending_backward_batch_set(
                    L.ci(F), L.ci(P), L.ci(C), L.ci(c0), L.ci(cn), L.ptr(fb.uv), L.ptr(fb.conic), L.ptr(opacity),
                    ctypes.c_int64(op_fs), L.ptr(row), ctypes.c_int64(P * C), L.ptr(fb.idx_sorted), L.ptr(fb.tile_range),
                    ctypes.c_int64(cap), L.cf(bg), L.ci(W), L.ci(H), L.ptr(fb.final_T), L.ptr(fb.ncontrib), L.ptr(dL),
                    L.ci(want_abs), L.ptr(fb.slot_sorted), L.ptr(rec), L.ptr(pack), L.ptr(_debug_T_front(F * H, W, dev)), st))
                L.check(lib.splat_frames_gauss_backward_static_cam(
                    L.ci(F), L.ci(P), L.ci(cn), L.ci(W), L.ci(H), ctypes.c_int64(cap), L.ci(want_abs), L.ptr(rec), L.ptr(fb.goff),
                    L.ptr(fb.radius), L.ptr(xyz), L.ptr(scales), L.ptr(uquats), ctypes.byref(camc), L.ci(1), L.ptr(bufs["xyz"]),
                    L.ptr(bufs["scales"]), L.ptr(bufs["uquats"]), L.ptr(bufs["opacity"]), L.ptr(dfeat), L.ci(cn),
                    L.ci(1 if detach else 0), L.ci(0 if is_depth else -1), L.ptr(fb.tap if taps else None),
                    L.ptr(fb.abs_tap if (taps and want_abs) else None), L.ptr(fb.radii_max if taps else None), st))
            c0 += cn
        ret = tuple(None if k in sink else bufs[k] for k in ("xyz", "scales", "uquats", "opacity"))
        return ret + (None,) * 9 + tuple(dfe)


class _RenderFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scales, uquats, opacity, feature, offsets, extr, fb: FrameBatch, bg, nearest, extent, sink, intr=None):
        xyz = _points(xyz, "xyz", 3)
        scales = _points(scales, "scales", 3)
        uquats = _points(uquats, "uquats", 4)
        opacity = L.need(opacity, "opacity")
        feature = _points(feature, "feature", fb.C)
        P, F = fb.P, fb.F
        cam = _Camera(extr, intr, F)
        if xyz.shape[0] != P or scales.shape[0] != P or uquats.shape[0] != P or opacity.numel() != P or feature.shape[0] != P:
            raise ValueError(f"the batch was built for {P} Gaussians")
        off = None
        if offsets is not None:
            off = L.need(offsets, "offsets")
            if tuple(off.shape) != (F, P, 3):
                raise ValueError(f"offsets must be [F={F}, P={P}, 3]")
        elif F > 1 and cam.extr_fs == 0:
            raise ValueError("several frames of static Gaussians need per-frame offsets or per-frame cameras")
        ctx.gen = fb._begin_forward()
        out = fb._forward_onecall(xyz, scales, uquats, opacity, feature, off, cam, bg, nearest, extent)
        ctx.fb, ctx.bg, ctx.sink = fb, bg, sink
        ctx.cam, ctx.off = cam, off
        ctx.cam_versions = _versions(cam.extr, cam.intr, off)
        ctx.save_for_backward(xyz, scales, uquats, opacity, feature)
        return out

    @staticmethod
    def backward(ctx, dL_dout):
        fb: FrameBatch = ctx.fb
        fb._check_generation(ctx.gen)
        xyz, scales, uquats, opacity, feature = ctx.saved_tensors
        _check_versions(ctx.cam_versions, "a camera / offset tensor")
        g = L.need(dL_dout, "dL_dout")
        sink = ctx.sink or {}
        like = {"xyz": xyz, "scales": scales, "uquats": uquats, "opacity": opacity, "feature": feature}
        # one accumulate flag for the launch: with a sink in play the buffers autograd receives start from zero
        fresh = torch.zeros_like if sink else torch.empty_like
        bufs = {k: (sink[k] if k in sink else fresh(v)) for k, v in like.items()}
        ret = tuple(None if k in sink else bufs[k] for k in ("xyz", "scales", "uquats", "opacity", "feature"))
        from .gs.raster_ops import _debug_T_front
        dbg = _debug_T_front(fb.F * fb.H, fb.W, g.device)
        fb._backward_onecall(g, xyz, scales, uquats, ctx.off, ctx.cam, ctx.bg, bufs, accumulate=bool(sink), dbg=dbg)
        fb._pending = False        # (frame_rasterization's pool: the batch's forward has had its backward)
        return ret + (None,) * 8


# ------------------------------------------------------------------ ONE frame behind one call per direction
# gs.rasterization's chain -- project_point -> compute_cov3d -> ewa_project -> sort_gaussian -> alpha_blending (reference:
# src/submodules/dptr/dptr/gs/__init__.py:28-100) -- for a caller that wants the image of ONE frame: a FrameBatch of one frame
# does it in one crossing of the C ABI per direction (splat_frames_forward / _backward: fused preprocess, binning with reach
# masks, sort, compositing | tile backward, Gaussian-side backward with the projection chain), one autograd node, a dozen tensor
# allocations less than the operator chain -- about 100 us of host time per frame instead of 220, which keeps a frame-by-frame
# loop GPU-bound on a busy host.  The batches are pooled by shape: a batch whose forward still waits for its backward is not
# reused (two views rendered before one backward get two batches; at most POOL_MAX per shape, then the oldest is reused and ITS
# late backward raises).
_FRAME_POOL: Dict[tuple, list] = {}
POOL_MAX = 4


def _pooled_batch(P: int, W: int, H: int, C: int, dev, needs_grad: bool) -> FrameBatch:
    key = (str(dev), int(P), int(W), int(H), int(C))
    pool = _FRAME_POOL.setdefault(key, [])
    fb = next((b for b in pool if not getattr(b, "_pending", False)), None)
    if fb is None:
        if len(pool) >= POOL_MAX:
            fb = pool.pop(0)
        else:
            fb = FrameBatch(1, P, W, H, C, dev)
        pool.append(fb)
    fb._pending = bool(needs_grad)
    return fb


def frame_rasterization(xyz: Tensor, scale: Tensor, rotate: Tensor, opacity: Tensor, feature: Tensor, extr: Tensor, W: int, H: int,
                        bg: float = 0.0, intr: Optional[Tensor] = None, offset: Optional[Tensor] = None, nearest: float = 0.01,
                        extent: float = 1.3, grad_sink: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """image [C, H, W] of one frame: the orthographic camera ``extr`` -- or, with ``intr`` (fx, fy, cx, cy), the pinhole camera
    of ``gs.rasterization`` -- over ``xyz (+ offset)``; differentiable w.r.t. xyz, scale, rotate, opacity, feature (cameras are
    not differentiated).  ``grad_sink`` as in ``FrameBatch.render`` (names xyz, scales, uquats, opacity, feature).  The batch that
    rendered the frame is ``frame_rasterization.last`` (its ``tap`` / ``radii_max`` after the backward: the densification taps)."""
    P, C = feature.shape
    needs = torch.is_grad_enabled() and any(t.requires_grad for t in (xyz, scale, rotate, opacity, feature))
    fb = _pooled_batch(P, W, H, C, xyz.device, needs)
    off = None if offset is None else offset.reshape(1, P, 3)
    out = fb.render(xyz, scale, rotate, opacity, feature, off, extr, bg=bg, nearest=nearest, extent=extent, grad_sink=grad_sink,
                    intr=intr)
    frame_rasterization.last = fb
    return out[0]

