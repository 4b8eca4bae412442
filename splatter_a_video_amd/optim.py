"""Adam on the flat parameter buffer of the frame-sharded data-parallel renderer.

After the gradient all-reduce every rank holds the same flat gradient; ``FlatAdam.step`` applies the update rule of
``torch.optim.Adam`` (what the reference's optimizer wrapper steps, src/pointrix/optimizer/optimizer.py:70-83, built with
eps = 1e-15 and one learning rate per parameter group) to the whole ``FlatGradBucket`` in ONE native launch
(``splat_adam_step``), with the groups' learning rates looked up by segment.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Union

import torch

from . import _lib as L
from .parallel import FlatGradBucket

MAX_SEGMENTS = 16


class PatternLR:
    """Two parameter groups interleaved inside ONE tensor: of every ``period`` consecutive elements the first ``head`` take
    ``head_lr``, the others ``lr``.  The reference keeps the SH coefficients as two parameters -- ``features`` (the DC triplet,
    lr 0.0025) and ``features_rest`` (lr 0.000125; src/configs/frag_gs_v10.yaml:44-47) -- and concatenates them per forward; here
    the block stays one [N, 16, 3] tensor: ``PatternLR(1.25e-4, head_lr=2.5e-3, period=48, head=3)``."""

    def __init__(self, lr: float, head_lr: float, period: int, head: int):
        if not (0 < head <= period):
            raise ValueError("0 < head <= period")
        self.lr, self.head_lr, self.period, self.head = float(lr), float(head_lr), int(period), int(head)

    def __eq__(self, o):
        return isinstance(o, PatternLR) and (self.lr, self.head_lr, self.period, self.head) == (o.lr, o.head_lr, o.period, o.head)

    def __repr__(self):
        return f"PatternLR({self.lr}, head_lr={self.head_lr}, period={self.period}, head={self.head})"


def lr_segments(slices: Dict[str, tuple], lrs: Dict[str, float], patterns: bool = False):
    """(segment ends, segment learning rates) of the flat buffer: neighbouring groups with equal rates share a segment (the
    kernel looks a segment up per element).  With ``patterns``: also (period, head, head rate) per segment -- a ``PatternLR``
    group keeps a segment of its own.  Pure host logic."""
    ends, seg_lr, pat = [], [], []
    for n, (_, e) in slices.items():
        r = lrs[n]
        if isinstance(r, PatternLR):
            ends.append(e); seg_lr.append(r.lr); pat.append((r.period, r.head, r.head_lr))
            continue
        r = float(r)
        if seg_lr and seg_lr[-1] == r and pat[-1] is None:
            ends[-1] = e
        else:
            ends.append(e); seg_lr.append(r); pat.append(None)
    if len(ends) > MAX_SEGMENTS:
        raise ValueError(f"at most {MAX_SEGMENTS} learning-rate segments")
    if patterns:
        return ends, seg_lr, [p or (0, 0, 0.0) for p in pat]
    if any(p is not None for p in pat):
        raise ValueError("a PatternLR group needs the pattern-aware caller (FlatAdam)")
    return ends, seg_lr


class FlatAdam:
    def __init__(self, bucket: FlatGradBucket, lr: Union[float, Dict[str, float]], betas=(0.9, 0.999), eps: float = 1e-15):
        if not bucket.flat_param.is_cuda:
            raise ValueError("FlatAdam steps GPU buffers (there is no CPU path)")
        self.bucket = bucket
        rate = lambda r: r if isinstance(r, PatternLR) else float(r)
        self.lr = {n: rate(lr[n] if isinstance(lr, dict) else lr) for n in bucket.slices}
        self._build_segments()
        self.beta1, self.beta2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.exp_avg = torch.zeros_like(bucket.flat_param)
        self.exp_avg_sq = torch.zeros_like(bucket.flat_param)
        self.t = 0

    def _build_segments(self) -> None:
        ends, seg_lr, pat = lr_segments(self.bucket.slices, self.lr, patterns=True)
        self.nseg = len(ends)
        self.seg_end = (ctypes.c_int64 * self.nseg)(*ends)
        self.seg_lr = (ctypes.c_float * self.nseg)(*seg_lr)
        self.has_pattern = any(p[0] for p in pat)
        self.seg_period = (ctypes.c_int32 * self.nseg)(*[p[0] for p in pat])
        self.seg_head = (ctypes.c_int32 * self.nseg)(*[p[1] for p in pat])
        self.seg_head_lr = (ctypes.c_float * self.nseg)(*[p[2] for p in pat])

    def set_lr(self, lr: Union[float, Dict[str, float]]) -> None:
        """New learning rate(s) -- one number for every group or ``{group name: rate}`` for some of them -- from the next
        ``step`` on; moments and step count are kept.  What the reference's scheduler does to the position group every
        iteration (src/pointrix/optimizer/scheduler.py: the exponential position-lr decay): call it before each step.
        Segments are re-derived, so groups whose rates diverge split and groups that meet merge again."""
        if isinstance(lr, dict):
            unknown = [n for n in lr if n not in self.lr]
            if unknown:
                raise KeyError(f"no parameter group(s) {unknown}; groups: {list(self.lr)}")
            for n, r in lr.items():
                self.lr[n] = r if isinstance(r, PatternLR) else float(r)
        else:
            for n in self.lr:
                self.lr[n] = float(lr)
        self._build_segments()

    def full_moments(self):
        """(exp_avg, exp_avg_sq) laid out like the bucket (``OwnerShardedAdam`` has to gather them)"""
        return self.exp_avg, self.exp_avg_sq

    def load_moments(self, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor) -> None:
        self.exp_avg.copy_(exp_avg)
        self.exp_avg_sq.copy_(exp_avg_sq)

    def zero_moments(self, name: str) -> None:
        a, b = self.bucket.slices[name]
        self.exp_avg[a:b].zero_()
        self.exp_avg_sq[a:b].zero_()

    def step(self, grad: Optional[torch.Tensor] = None, grad_scale: float = 1.0) -> None:
        """one Adam step with the bucket's active gradient buffer (or ``grad``), scaled by ``grad_scale`` first"""
        g = self.bucket.flat_grad if grad is None else grad
        p = self.bucket.flat_param
        if g.numel() != p.numel() or not g.is_cuda or g.dtype != torch.float32:
            raise ValueError("grad must be a float32 GPU tensor with one entry per parameter")
        self.t += 1
        with torch.no_grad():
            if self.has_pattern:
                L.check(L.lib().splat_adam_step_pattern(
                    ctypes.c_int64(p.numel()), L.ptr(p), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                    L.ci(self.nseg), self.seg_end, self.seg_lr, self.seg_period, self.seg_head, self.seg_head_lr,
                    L.cf(self.beta1), L.cf(self.beta2), L.cf(self.eps), L.ci(self.t), L.cf(grad_scale), L.stream()))
            else:
                L.check(L.lib().splat_adam_step(
                    ctypes.c_int64(p.numel()), L.ptr(p), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                    L.ci(self.nseg), self.seg_end, self.seg_lr, L.cf(self.beta1), L.cf(self.beta2), L.cf(self.eps),
                    L.ci(self.t), L.cf(grad_scale), L.stream()))


class OwnerShardedAdam:
    """Adam for ``parallel.owner_sharded_step``: the REPLICATED slice [b, total) of the flat buffer is stepped on every rank,
    of the owned parameter [a, b) only this rank's block [lo, hi) -- the moments of the other blocks do not exist here (the
    spline table of the reference's model at 200 frames: 662 MB of parameters, 2 x 662 MB of moments; 1/8 of the moments and of
    the optimiser's streaming per rank at 8 GPUs).  Same update rule and learning-rate groups as ``FlatAdam`` (two launches of
    ``splat_adam_step`` / ``splat_adam_step_pattern``)."""

    def __init__(self, bucket: FlatGradBucket, shards, lr: Union[float, Dict[str, float]], betas=(0.9, 0.999), eps: float = 1e-15):
        if not bucket.flat_param.is_cuda:
            raise ValueError("OwnerShardedAdam steps GPU buffers (there is no CPU path)")
        self.bucket, self.shards = bucket, shards
        rate = lambda r: r if isinstance(r, PatternLR) else float(r)
        self.lr = {n: rate(lr[n] if isinstance(lr, dict) else lr) for n in bucket.slices}
        self.beta1, self.beta2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        lo, hi = shards.own
        if lo % 4 or hi % 4 or shards.b % 4:
            raise ValueError("block boundaries must be multiples of 4 floats (16-byte aligned sub-buffers)")
        self._check_owned_rate()
        dev = bucket.flat_param.device
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self.m_own, self.v_own = z(hi - lo), z(hi - lo)
        self.m_rep, self.v_rep = z(shards.total - shards.b), z(shards.total - shards.b)
        self.t = 0

    def _check_owned_rate(self) -> None:
        """an ``OwnerShards`` parameter is cut in whole UNITS (segments of the spline table), not in periods of a pattern: a
        PatternLR there is refused on every rank alike (``Zero1Shards`` has no owned parameter: its blocks handle the phase)"""
        n = getattr(self.shards, "name", None)
        if n is not None and isinstance(self.lr[n], PatternLR):
            raise ValueError(f"the owned parameter {n!r} cannot take a PatternLR")

    def set_lr(self, lr: Union[float, Dict[str, float]]) -> None:
        """as ``FlatAdam.set_lr``"""
        if isinstance(lr, dict):
            unknown = [n for n in lr if n not in self.lr]
            if unknown:
                raise KeyError(f"no parameter group(s) {unknown}; groups: {list(self.lr)}")
            for n, r in lr.items():
                self.lr[n] = r if isinstance(r, PatternLR) else float(r)
        else:
            for n in self.lr:
                self.lr[n] = float(lr)
        self._check_owned_rate()

    def _segments(self, lo: int, hi: int):
        """learning-rate segments (+ patterns) of the sub-buffer [lo, hi), relative to lo.  A block that starts INSIDE a
        ``PatternLR`` group (ZeRO-1 cuts the buffer anywhere) starts mid-period: the rest of that period becomes a segment of its
        own (what is left of the head keeps the head rate), the whole periods behind it keep the pattern."""
        sl, lrs = {}, {}
        for n, (a, b) in self.bucket.slices.items():
            s0, s1 = max(a, lo), min(b, hi)
            if s1 <= s0:
                continue
            r = self.lr[n]
            if isinstance(r, PatternLR) and (s0 - a) % r.period:
                ph = (s0 - a) % r.period
                cut = min(s0 + r.period - ph, s1)
                sl[n + "#pre"] = (s0 - lo, cut - lo)
                lrs[n + "#pre"] = PatternLR(r.lr, r.head_lr, r.period, r.head - ph) if ph < r.head else r.lr
                if s1 > cut:
                    sl[n], lrs[n] = (cut - lo, s1 - lo), r
            else:
                sl[n], lrs[n] = (s0 - lo, s1 - lo), r
        return lr_segments(sl, lrs, patterns=True)

    def _step(self, lo: int, hi: int, m, v, grad_scale: float) -> None:
        if hi <= lo:
            return
        p, g = self.bucket.flat_param[lo:hi], self.bucket.flat_grad[lo:hi]
        ends, rates, pat = self._segments(lo, hi)
        n = len(ends)
        c_ends, c_rates = (ctypes.c_int64 * n)(*ends), (ctypes.c_float * n)(*rates)
        if any(q[0] for q in pat):
            L.check(L.lib().splat_adam_step_pattern(
                ctypes.c_int64(hi - lo), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ci(n), c_ends, c_rates,
                (ctypes.c_int32 * n)(*[q[0] for q in pat]), (ctypes.c_int32 * n)(*[q[1] for q in pat]),
                (ctypes.c_float * n)(*[q[2] for q in pat]), L.cf(self.beta1), L.cf(self.beta2), L.cf(self.eps), L.ci(self.t),
                L.cf(grad_scale), L.stream()))
        else:
            L.check(L.lib().splat_adam_step(ctypes.c_int64(hi - lo), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ci(n), c_ends, c_rates,
                                            L.cf(self.beta1), L.cf(self.beta2), L.cf(self.eps), L.ci(self.t), L.cf(grad_scale), L.stream()))

    def step(self, grad_scale: float = 1.0) -> None:
        self.t += 1
        with torch.no_grad():
            lo, hi = self.shards.own
            self._step(lo, hi, self.m_own, self.v_own, grad_scale)
            self._step(self.shards.b, self.shards.total, self.m_rep, self.v_rep, grad_scale)

    # ---- the moments as whole flat buffers (structure changes: clone / split / prune move them with their Gaussians)
    def full_moments(self):
        """(exp_avg, exp_avg_sq) laid out like the bucket, every owner's block gathered (a collective: all ranks call it)"""
        from .parallel import owner_gather
        sh = self.shards
        lo, hi = sh.own
        out = []
        for own, rep in ((self.m_own, self.m_rep), (self.v_own, self.v_rep)):
            full = torch.zeros_like(self.bucket.flat_param)
            full[lo:hi].copy_(own)
            full[sh.b:sh.total].copy_(rep)
            owner_gather(self.bucket, sh, flat=full)
            out.append(full)
        return out[0], out[1]

    def load_moments(self, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor) -> None:
        """this rank's share of whole flat moment buffers"""
        sh = self.shards
        lo, hi = sh.own
        self.m_own.copy_(exp_avg[lo:hi]); self.v_own.copy_(exp_avg_sq[lo:hi])
        self.m_rep.copy_(exp_avg[sh.b:sh.total]); self.v_rep.copy_(exp_avg_sq[sh.b:sh.total])

    def zero_moments(self, name: str) -> None:
        """the moments of parameter ``name`` := 0: its part of the replicated slice, or what of it lies in this rank's block"""
        a, b = self.bucket.slices[name]
        sh = self.shards
        if a >= sh.b:
            for t in (self.m_rep, self.v_rep):
                t[a - sh.b:b - sh.b].zero_()
            return
        if b > sh.b:
            raise ValueError(f"parameter {name!r} straddles the owned / replicated boundary")
        lo, hi = sh.own
        s0, s1 = max(a, lo), min(b, hi)
        if s1 > s0:
            for t in (self.m_own, self.v_own):
                t[s0 - lo:s1 - lo].zero_()
