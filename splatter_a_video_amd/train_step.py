"""One training step of the reference's trainer, composed from the native pieces of SURVEY.md section 8.

What ``Trainer.train_one_step`` runs per step (reference: src/trainer_fragGS.py:736-790 with ``compute_all_losses``
:470-735): two model forwards (``ids1`` and a random ``ids2``; src/loaders/gs_data2.py:57-88), ``render_iter`` of frame
``ids1`` with ``track_gs = position(ids2)`` in front of the render attributes (:506-512, ``num_idx = 20``), image losses,
K = 5 nearest neighbours + the ARAP energy of the pair (:671-675; src/geometry_utils.py:7-38,90-123), backward, Adam, and every
``interval`` steps clone / split / prune (src/pointrix/optimizer/atlas_gs_optimizer.py:93-184).  Here a step takes ``F`` such
(ids1, ids2) pairs per rank at once -- the frame batch -- and is a fixed sequence of native launches:

    SH colours (once)                                   gs.compute_sh_into                       row a7
    position(ids1), position(ids2) of all pairs         splat_dynamic_positions_batch_forward    row a15
    neighbours of the sampled vertices + ARAP           splat_knn_brute_batch, splat_arap_energy_batch   row f3
    dynamic preprocess + binning + sort + 3 blends      FrameBatch.render_dynamic_sets           rows a15/f1, a8, a9-a11
        the attribute set = [track_gs (per frame) | attributes (shared)] read in place (feature sources)
    L1 losses and their gradients                       fused into the tile backward (splat_alpha_blending_backward_batch_sets_l1;
                                                        fused_l1=False: splat_l1_loss_grad + gradient images)
    three-set tile backward + Gaussian-side walk        (autograd of render_dynamic_sets)        row a10
        track_gs' per-frame gradient lands next to the ARAP gradient of position(ids2)
    both position gradients -> spline segments          splat_dynamic_positions_batch_backward
    all-reduce of the flat bucket (N > 1), Adam          parallel.FlatGradBucket, optim.FlatAdam  row 8e
        (owner_sharded=True: the spline table reduced to the owners of its time blocks, sharded moments, blocks gathered)
    densification statistics, reduced over the ranks    densify.DensifyState, parallel.reduce_densify_batch   row f2
    every `interval` steps: clone / split / prune with the Adam moments, Morton reorder, buffers rebuilt at the new N

No trainer class, data loader or loss library of the reference is rebuilt: the ground-truth frames are handed in as tensors.
There is no CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib as L
from . import densify as D
from .arap import pair_arap, pair_connectivity
from .dynamics import (GAUSSIAN_MAJOR, SEGMENT_MAJOR, FrameClock, frame_table, positions_batch_backward,
                       positions_batch_forward, to_gaussian_major, to_segment_major)
from .frames import FrameBatch
from .gs.fused_ops import compute_sh_into
from .gs.point_ops import project_point_ortho
from .optim import FlatAdam, OwnerShardedAdam, PatternLR
from .parallel import (FlatGradBucket, OwnerShards, PositionExchangePlan, Zero1Shards, exchange_frames, gather_times, owner_gather,
                       owner_reduce, reduce_densify_batch)

TRAINABLE = ("pos_cubic_node", "rotation", "opacity", "scaling", "shs", "attrs")
FROZEN = ("position", "rot_poly_feat", "rot_fourier_feat")          # :90 position is not optimised; :195-197 detached tables
# learning rates of the reference's configuration (src/configs/frag_gs_v10.yaml:40-66).  The SH block is ONE [N,16,3] tensor here;
# its two parameter groups -- features (the DC triplet of every Gaussian) and features_rest -- keep their own rates through a
# pattern inside the Adam kernel's segment (optim.PatternLR: of every 48 floats the first 3)
REFERENCE_LR = {"pos_cubic_node": 6e-5, "rotation": 1e-3, "opacity": 5e-2, "scaling": 5e-3,
                "shs": PatternLR(1.25e-4, head_lr=2.5e-3, period=48, head=3), "attrs": 1e-3}


@dataclass
class DensifyConfig:
    """src/configs/frag_gs_v10.yaml:26-38 (extra_cfg of AtlasGaussianSplattingOptimizer)"""
    interval: int = 100                 # duplicate_interval = prune_interval
    start_iter: int = 500
    stop_iter: int = 100000
    grad_threshold: float = 2e-4
    percent_dense: float = 1e-3
    cameras_extent: float = 1.0
    min_opacity: float = 0.05
    split_num: int = 2
    size_threshold: float = 20.0
    seed: int = 0


@dataclass
class LossWeights:
    rgb: float = 1.0
    depth: float = 1.0
    attr: float = 1.0
    arap: float = 1e-3                  # rigid_error = cal_arap_error(...) / 1000 (src/trainer_fragGS.py:674)


class _Phases:
    """optional per-phase GPU time of a step (events on the current stream; read with ``ms()`` after a synchronise)"""

    def __init__(self, on: bool):
        self.on, self.marks = on, []

    def mark(self, name: str) -> None:
        if self.on:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def ms(self) -> Dict[str, float]:
        out: Dict[str, float] = {}
        for (_, a), (name, b) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        return out


class TrainingStep:
    """Parameters (flat bucket + Adam), frame batch and densification state of one video's Gaussians, and the step.

    ``params``: per-Gaussian tensors on the GPU -- position [N,3], pos_cubic_node [N, 4*I*3] (the reference's layout; stored
    segment-major), rotation [N,4], rot_poly_feat [N,4,4], rot_fourier_feat [N,8,4], opacity [N,1] (logit), scaling [N,3]
    (log), shs [N,16,3], attrs [N,A] (the render attributes behind track_gs; 3 + 1 + 3 + A = the composited row, A = 16 for
    the renderer's own 23-channel plan).  At set-up (and after every densification) the Gaussians are put in Morton order of their
    screen positions (``spatial_order=False``: kept as given); ``initial_order`` maps the rows here to the caller's."""

    def __init__(self, params: Dict[str, Tensor], clock: FrameClock, W: int, H: int, frames_per_step: int, extr: Tensor,
                 lr: Optional[Dict[str, float]] = None, weights: Optional[LossWeights] = None,
                 densify: Optional[DensifyConfig] = None, K: int = 20, knn_K: int = 5, arap_samples: int = 512,
                 bg: float = 0.0, sample_seed: Optional[int] = None, timing: bool = False, owner_sharded: bool = False,
                 spatial_order: bool = True, zero1: bool = False, fused_l1: bool = True, exchange_positions: bool = False):
        self.clock, self.W, self.H, self.F = clock, int(W), int(H), int(frames_per_step)
        self.extr = extr
        self.dev = params["position"].device
        self.lr = dict(REFERENCE_LR if lr is None else lr)
        self.w = weights or LossWeights()
        self.cfg = densify or DensifyConfig()
        self.K, self.knn_K, self.S, self.bg = int(K), int(knn_K), int(arap_samples), float(bg)
        self.timing = timing
        self.fused_l1 = bool(fused_l1)      # the L1 losses' gradient inside the tile backward (no gradient images)
        self.iteration = 0
        self.history: List[int] = []            # Gaussian count after every structure change
        self.last: Dict[str, Tensor] = {}
        self.phase_ms: Dict[str, float] = {}
        on = dist.is_available() and dist.is_initialized()
        self.world, self.rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
        # the ARAP samples are drawn on the device (no host copy on the step's critical path); by default every rank draws its own
        # (the ranks render different pairs: the seed folds the rank in)
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(sample_seed) if sample_seed is not None else 7919 * self.rank)
        # the spline table's gradient reduced to the owners of its time blocks, their Adam moments sharded
        self.owner_sharded = bool(owner_sharded)
        # ZeRO-1: the WHOLE flat buffer in `world` equal blocks -- reduce-scatter, Adam on 1 / world, all-gather
        self.zero1 = bool(zero1)
        if self.zero1 and self.owner_sharded:
            raise ValueError("owner_sharded and zero1 are two schedules of the same step: pick one")
        # POSITION EXCHANGE (with owner_sharded): position(ids2) of a pair frame in another rank's time block comes from
        # its owner and its gradient goes back there -- the spline table, its gradient and its moments never leave their owner.
        # The frames ids1 a rank renders must lie in ITS block (OwnerShards.frames_of_rank deals them so).
        self.exchange = bool(exchange_positions)
        if self.exchange and not self.owner_sharded:
            raise ValueError("exchange_positions belongs to the owner-sharded step (owner_sharded=True)")
        p0 = {k: params[k] for k in TRAINABLE + FROZEN}
        # setup, as after every densification: the Gaussians in Morton order of their screen positions at the clip's first frame
        # (the binning kernels' locality and the neighbour search's bound want space neighbours at neighbouring
        # indices; results do not depend on the order).  ``initial_order[i]`` = the caller's row of Gaussian i here.
        self.initial_order = None
        if spatial_order:
            self.initial_order = self._morton(p0)
            p0, _ = D.reorder_points(p0, None, self.initial_order)
        self._build(p0, None, 0)

    # ------------------------------------------------------------------ buffers at the current Gaussian count
    def _build(self, p: Dict[str, Tensor], moments, adam_t: int) -> None:
        """flat bucket, Adam, frame batch and statistics for the per-Gaussian tensors ``p`` (pos_cubic_node in the reference's
        [N, 4*I*3] layout); ``moments``: {name: (exp_avg, exp_avg_sq)} in the same layouts, or None (fresh)"""
        I = self.clock.interval_num
        N = p["position"].shape[0]
        self.N = N
        seg = lambda t: to_segment_major(t.reshape(N, -1), I)
        train = {k: (seg(p[k]) if k == "pos_cubic_node" else p[k].contiguous()) for k in TRAINABLE}
        self.bucket = FlatGradBucket(train, pad_to=4 * self.world if self.zero1 else 1)
        self.p = self.bucket.params
        self.frozen = {k: p[k].contiguous() for k in FROZEN}
        if self.zero1:
            self.shards = Zero1Shards(self.bucket, self.world, self.rank)
            self.opt = OwnerShardedAdam(self.bucket, self.shards, self.lr, eps=1e-15)
        elif self.owner_sharded:
            self.shards = OwnerShards(self.bucket, "pos_cubic_node", self.world, self.rank)
            self.opt = OwnerShardedAdam(self.bucket, self.shards, self.lr, eps=1e-15)
        else:
            self.shards = None
            self.opt = FlatAdam(self.bucket, self.lr, eps=1e-15)
        self.opt.t = adam_t
        if moments is not None:
            full = [torch.zeros_like(self.bucket.flat_param) for _ in range(2)]
            for k in TRAINABLE:
                a, b = self.bucket.slices[k]
                for dst, src in zip(full, moments[k]):
                    src = seg(src) if k == "pos_cubic_node" else src
                    dst[a:b].copy_(src.reshape(-1))
            self.opt.load_moments(full[0], full[1])
        A = self.p["attrs"].shape[1]
        self.C = 3 + 1 + 3 + A
        self.fb = FrameBatch(self.F, N, self.W, self.H, self.C, self.dev, want_abs=False)
        self.dstate = D.DensifyState(N, self.dev)
        self.dirs = torch.zeros(N, 3, device=self.dev)
        self.dirs[:, 2] = 1.0
        self.pairs = torch.empty(self.F, 2, N, 3, dtype=torch.float32, device=self.dev)      # position(ids1), position(ids2)
        self.g_pairs = torch.empty_like(self.pairs)
        self._tabs: Dict[tuple, tuple] = {}
        self.history.append(N)

    def _tables(self, times1, times2):
        key = (tuple(float(t) for t in times1), tuple(float(t) for t in times2))
        hit = self._tabs.get(key)
        if hit is None:
            inter = [t for pair in zip(times1, times2) for t in pair]      # (t1_0, t2_0, t1_1, t2_1, ...): pairs[f] = [pos1, pos2]
            hit = (frame_table(self.clock, inter, self.dev),)
            if len(self._tabs) > 64:
                self._tabs.clear()
            self._tabs[key] = hit
        return hit[0]

    @staticmethod
    def _fill(t: Tensor, v: float = 0.0) -> None:
        import ctypes
        L.check(L.lib().splat_fill_f32(L.ptr(t), ctypes.c_size_t(t.numel()), L.cf(v), L.stream()))

    def _l1(self, pred: Tensor, target: Tensor, weight: float, loss_slot: Tensor) -> Tensor:
        """gradient image of weight * mean |pred - target| over the step's GLOBAL batch share of this rank (mean over the local
        frames; the ranks' means are averaged by the optimiser's 1 / world), the sum of |.| added to ``loss_slot``"""
        import ctypes
        F, c, H, W = pred.shape
        inner = c * H * W
        if pred.stride(1) != H * W or pred.stride(2) != W or pred.stride(3) != 1:
            pred = pred.contiguous()
        target = L.need(target, "ground truth")
        if tuple(target.shape) != (F, c, H, W):
            raise ValueError(f"ground truth must be [{F}, {c}, {H}, {W}]")
        g = torch.empty(F, c, H, W, dtype=torch.float32, device=pred.device)
        L.check(L.lib().splat_l1_loss_grad(L.ci(F), ctypes.c_int64(inner), L.ptr(pred), ctypes.c_int64(pred.stride(0)), L.ptr(target),
                                           L.cf(weight / (F * inner)), L.ptr(g), L.ptr(loss_slot), L.stream()))
        return g

    # ------------------------------------------------------------------ the step
    def step(self, times1: Sequence[float], times2: Sequence[float], gt: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """one gradient step on the pairs (times1[f], times2[f]); ``gt``: rgb [F,3,H,W], depth [F,1,H,W], attr [F,3+A,H,W] of
        the frames times1 (attr: track_gs of the pair in its first three channels).  Returns device scalars (no host sync):
        the L1 sums of the three images, the ARAP energies.  The order of the pairs inside a step does not change the result;
        sorted by times1 the Gaussian-side backward shares its projection / EWA chain between consecutive frames of one spline
        segment."""
        if len(times1) != self.F or len(times2) != self.F:
            raise ValueError(f"the step takes {self.F} frame pairs")
        ph = _Phases(self.timing)
        ph.mark("start")
        F, N, I = self.F, self.N, self.clock.interval_num
        p, fz, bk = self.p, self.frozen, self.bucket
        bk.zero_grad()
        g = {k: bk.grad(k) for k in p}
        # ---- model evaluation outside the renderer: SH colours (constant view direction) and both positions of every pair
        rgb = compute_sh_into(p["shs"], 3, self.dirs, None, g["shs"])
        tab12 = self._tables(times1, times2)
        if self.exchange:
            self.clock_times1 = [float(t) for t in times1]
            plan = self._positions_by_exchange(times1, times2)
        else:
            positions_batch_forward(tab12, fz["position"], p["pos_cubic_node"], I, SEGMENT_MAJOR, out=self.pairs.view(2 * F, N, 3))
        self._fill(self.g_pairs)
        ph.mark("model_eval")
        # ---- rigidity of the pair: neighbours of the sampled vertices in frame ids1, ARAP energy + gradient of both frames
        S = min(self.S, N)
        # (ascending: the Gaussians are in Morton order, so a wave of the neighbour search holds space neighbours; the energy is
        #  a sum over the samples, their order does not matter)
        if N > S:     # with replacement, as np.random.choice(Nv, sample_num) of cal_arap_error (src/geometry_utils.py:103)
            sample = torch.sort(torch.randint(N, (F, S), generator=self.gen, device=self.dev), dim=1).values
        else:
            sample = torch.arange(N, device=self.dev).expand(F, N).contiguous()
        nbr = pair_connectivity(self.pairs[:, 0], sample, K=self.knn_K)
        arap = pair_arap(self.pairs, sample, nbr, d_pairs=self.g_pairs, grad_scale=self.w.arap / F)
        ph.mark("knn_arap")
        # ---- the frame: dynamic Gaussians through the three blends, track_gs = position(ids2) read in place
        times = list(times1)
        tab1 = self.fb.frame_table(self.clock, times)
        sets = [dict(feature=rgb, bg=self.bg, taps=True), dict(feature="depth", bg=1.0),
                dict(feature=[self.pairs[:, 1], p["attrs"]], bg=0.0, detach_opacity=True)]
        sink = {k: g[k] for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
        sink["feature:1"] = self.g_pairs[:, 1]          # track_gs' gradient: next to the ARAP gradient of position(ids2)
        sink["feature:2"] = g["attrs"]
        out = self.fb.render_dynamic_sets(self.clock, times, self.extr, sets, position=fz["position"],
                                          pos_cubic_node=p["pos_cubic_node"], rotation=p["rotation"],
                                          rot_poly_feat=fz["rot_poly_feat"], rot_fourier_feat=fz["rot_fourier_feat"],
                                          opacity=p["opacity"], scaling=p["scaling"], cubic_layout=SEGMENT_MAJOR, K=self.K,
                                          grad_sink=sink)
        ph.mark("render_forward")
        if self.fused_l1 and L.get_option("bwd_quarters"):
            # the L1 terms' gradient images are never materialised: the tile kernel derives them from the forward's output row and
            # the ground-truth frames where it hoists the image gradient (splat_alpha_blending_backward_batch_sets_l1)
            fsums = torch.empty(F, self.fb.T, 3, dtype=torch.float32, device=self.dev)      # per tile: every entry is written
            self.fb.fuse_l1([gt["rgb"], gt["depth"], gt["attr"]], [self.w.rgb, self.w.depth, self.w.attr], fsums)
            ph.mark("loss")
            torch.autograd.backward(list(out[:3]), self.fb.l1_placeholders([3, 1, self.C - 4]))
            sums = fsums.sum((0, 1))
        else:
            sums = torch.zeros(3, dtype=torch.float32, device=self.dev)
            grads = [self._l1(out[0], gt["rgb"], self.w.rgb, sums[0:1]), self._l1(out[1], gt["depth"], self.w.depth, sums[1:2]),
                     self._l1(out[2], gt["attr"], self.w.attr, sums[2:3])]
            ph.mark("loss")
            torch.autograd.backward(list(out[:3]), grads)
        # ---- both position gradients of every pair (ARAP on ids1 and ids2, track_gs on ids2) reach the spline segments
        if self.exchange:
            self._position_gradients_by_exchange(plan, g["pos_cubic_node"])
        else:
            positions_batch_backward(tab12, self.g_pairs.view(2 * F, N, 3), I, SEGMENT_MAJOR, None, g["pos_cubic_node"])
        ph.mark("render_backward")
        # ---- data parallelism: one all-reduce of the flat bucket, identical Adam on every rank -- or (owner_sharded) the
        #      spline table's gradient reduced to the owners of its time blocks, their blocks stepped there and gathered
        if self.exchange:
            # the table's gradient is complete on its owner (own frames + the position gradients it received): nothing of it is
            # reduced, nothing of the table gathered; the other ranks' blocks of the local copy go stale and are never read
            owner_reduce(bk, self.shards, reduce_owned=False)
            self.opt.step(grad_scale=1.0 / self.world)
        elif self.owner_sharded or self.zero1:
            owner_reduce(bk, self.shards)
            self.opt.step(grad_scale=1.0 / self.world)
            owner_gather(bk, self.shards)
        else:
            bk.all_reduce()
            self.opt.step(grad_scale=1.0 / self.world)
        ph.mark("allreduce_adam")
        # ---- densification statistics of the batch (reduced over the ranks: identical decisions everywhere)
        st = self.dstate
        st.begin_batch()
        # the taps are those of THIS rank's loss share: the mean over its F frames (`_l1` divides by the local F).  The step's
        # loss is the mean over all world * F frames (the optimiser divides the summed gradient by `world`), so the statistic a
        # single process with the whole batch would accumulate (render_batch sums the batch's taps: frag_model.py:326-343) is the
        # ranks' sum / world -- the clone / split thresholds then do not depend on the number of GPUs
        st.accumulate_frame(self.fb.radii_max, self.fb.tap, scale=(1.0 / self.world, 1.0 / self.world))
        reduce_densify_batch(st.viewspace_grad, st.visibility, st.radii)
        st.update()
        ph.mark("densify_stats")
        self.iteration += 1
        hw = self.W * self.H
        self.last = {"l1_rgb": sums[0] / (F * 3 * hw), "l1_depth": sums[1] / (F * hw),
                     "l1_attr": sums[2] / (F * (self.C - 4) * hw), "arap": arap.mean()}
        if self.timing:
            self._marks = ph
        return self.last

    # ------------------------------------------------------------------ position exchange (owner-sharded table)
    def _owner_of_time(self, t: float) -> int:
        seg = int(self.clock.scalars(t)[0])
        ub = self.shards.unit_bounds
        return next(r for r in range(self.world) if ub[r] <= seg < ub[r + 1])

    def _positions_by_exchange(self, times1, times2) -> PositionExchangePlan:
        """pairs[:, 0] = position(ids1) from this rank's own block; pairs[:, 1] = position(ids2) from the owners of those frames
        (this rank evaluates what the others -- and itself -- requested of its block and sends it)"""
        F, N, I = self.F, self.N, self.clock.interval_num
        fz, p = self.frozen, self.p
        for t in times1:
            if self._owner_of_time(t) != self.rank:
                raise ValueError(f"position exchange: frame {t} of ids1 is not in rank {self.rank}'s time block "
                                 "(deal the frames with OwnerShards.frames_of_rank)")
        positions_batch_forward(frame_table(self.clock, list(times1), self.dev), fz["position"], p["pos_cubic_node"], I, SEGMENT_MAJOR,
                                out=self.pairs[:, 0])
        plan = PositionExchangePlan(gather_times(times2, self.world, self.rank), self._owner_of_time, self.rank)
        plan.tab = frame_table(self.clock, [t for _, _, t in plan.serve], self.dev) if plan.serve else None
        plan.buf = torch.empty(len(plan.serve), N, 3, dtype=torch.float32, device=self.dev)
        if plan.serve:
            positions_batch_forward(plan.tab, fz["position"], p["pos_cubic_node"], I, SEGMENT_MAJOR, out=plan.buf)
        sends, recvs = [], []
        for j, (r, k, _) in enumerate(plan.serve):
            if r == self.rank:
                self.pairs[k, 1].copy_(plan.buf[j])
            else:
                sends.append((r, plan.buf[j]))
        for k, o in enumerate(plan.mine):
            if o != self.rank:
                recvs.append((o, self.pairs[k, 1]))
        if self.world > 1:
            exchange_frames(sends, recvs)
        return plan

    def _position_gradients_by_exchange(self, plan: PositionExchangePlan, d_table: Tensor) -> None:
        """dL/dposition(ids1) into this rank's block; dL/dposition(ids2) to the owners, whose positions' backward adds what they
        receive (and their own requests) into THEIR block"""
        F, N, I = self.F, self.N, self.clock.interval_num
        tab1 = frame_table(self.clock, [float(self.clock_times1[k]) for k in range(F)], self.dev)
        positions_batch_backward(tab1, self.g_pairs[:, 0], I, SEGMENT_MAJOR, None, d_table)
        gbuf = torch.empty_like(plan.buf)
        sends, recvs = [], []
        for k, o in enumerate(plan.mine):
            if o != self.rank:
                sends.append((o, self.g_pairs[k, 1]))
        for j, (r, k, _) in enumerate(plan.serve):
            if r == self.rank:
                gbuf[j].copy_(self.g_pairs[k, 1])
            else:
                recvs.append((r, gbuf[j]))
        if self.world > 1:
            exchange_frames(sends, recvs)
        if plan.serve:
            positions_batch_backward(plan.tab, gbuf, I, SEGMENT_MAJOR, None, d_table)

    def sync_table(self) -> None:
        """position exchange: every owner's block of the spline table to every rank (a collective; before anything reads the whole
        table: a structure change, a checkpoint)"""
        if self.exchange:
            owner_gather(self.bucket, self.shards)

    def phases(self) -> Dict[str, float]:
        """GPU milliseconds of the last step's phases (``timing=True``; synchronises)"""
        torch.cuda.synchronize()
        return self._marks.ms() if getattr(self, "_marks", None) is not None else {}

    def loss(self) -> float:
        """weighted loss of the last step (host sync)"""
        l = self.last
        return float(self.w.rgb * l["l1_rgb"] + self.w.depth * l["l1_depth"] + self.w.attr * l["l1_attr"] + self.w.arap * l["arap"])

    # ------------------------------------------------------------------ structure
    def _gather(self):
        """the per-Gaussian tensors and Adam moments in row layout (pos_cubic_node back in the reference's [N, 4*I*3])"""
        self.sync_table()
        N = self.N
        p = {k: v.detach() for k, v in self.p.items()}
        p["pos_cubic_node"] = to_gaussian_major(p["pos_cubic_node"].reshape(self.clock.interval_num, N, 4, 3))
        p.update(self.frozen)
        m = {}
        exp_avg, exp_avg_sq = self.opt.full_moments()        # (owner_sharded: gathered from the owners -- a collective)
        for k in TRAINABLE:
            a, b = self.bucket.slices[k]
            ea, es = exp_avg[a:b].view(self.p[k].shape), exp_avg_sq[a:b].view(self.p[k].shape)
            if k == "pos_cubic_node":
                I = self.clock.interval_num
                ea, es = to_gaussian_major(ea.reshape(I, N, 4, 3)), to_gaussian_major(es.reshape(I, N, 4, 3))
            m[k] = (ea, es)
        return p, m

    def set_lr(self, lr: Dict[str, float]) -> None:
        """new learning rate(s) from the next step on (the reference's ExponLRScheduler on the spline table,
        src/configs/frag_gs_v10.yaml:68-76: call it before each step); they survive a rebuild"""
        self.lr.update({k: (v if isinstance(v, PatternLR) else float(v)) for k, v in lr.items()})
        self.opt.set_lr(lr)

    def reset_opacity(self, ceiling: float = 0.01) -> None:
        """``reset_opacity`` of the reference (atlas_gs_optimizer.py:178-190, every ``opacity_reset_interval`` steps): opacities
        above ``ceiling`` fall back to it (logit space), and -- as ``replace_tensor_to_optimizer`` does -- the Adam moments of the
        opacity group start from zero"""
        with torch.no_grad():
            p = self.p["opacity"]
            p.copy_(torch.minimum(p, torch.full_like(p, math.log(ceiling / (1.0 - ceiling)))))
            self.opt.zero_moments("opacity")

    def _morton(self, p: Dict[str, Tensor]) -> Tensor:
        """permutation into Morton order of the screen positions at the clip's first frame; ``p`` in row layout"""
        tab0 = frame_table(self.clock, [0], self.dev)
        pos0 = positions_batch_forward(tab0, p["position"].contiguous(), p["pos_cubic_node"].reshape(p["position"].shape[0], -1).contiguous(),
                                       self.clock.interval_num, GAUSSIAN_MAJOR)[0]
        with torch.no_grad():
            uv, _ = project_point_ortho(pos0, self.extr, self.W, self.H, nearest=0.01)
        return D.spatial_order(uv, self.W, self.H)

    def maybe_densify(self) -> bool:
        """clone / split / prune at the reference's cadence (atlas_gs_optimizer.py:120-121,166-176); True when N changed"""
        c = self.cfg
        it = self.iteration
        if not (c.start_iter < it < c.stop_iter) or it % c.interval != 0:
            return False
        self.densify()
        return True

    def densify(self) -> None:
        """``densification`` of the reference (:166-176) on the device: clone, split (children from the counter-based generator,
        same seed on every rank), prune -- each with the Adam moments -- then the Gaussians in Morton order of their screen
        positions and every buffer rebuilt at the new count."""
        c, st = self.cfg, self.dstate
        p, m = self._gather()
        clone, split, _ = st.masks(p["scaling"], p["opacity"], c.grad_threshold, c.percent_dense, c.cameras_extent, c.min_opacity,
                                   c.size_threshold)
        p, m, n_clone = D.densify_clone(p, m, clone)
        # the reference splits on the SAME gradients, zero for the clones just appended (generate_split_mask :236-240)
        split = torch.cat([split, torch.zeros(n_clone, dtype=torch.bool, device=self.dev)])
        p, m, _, n_split = D.densify_split(p, m, split, c.split_num, seed=c.seed + self.iteration)
        # prune after reset_densification_state (:304,:343): opacity and world-size criteria on fresh statistics
        fresh = D.DensifyState(p["position"].shape[0], self.dev)
        _, _, prune = fresh.masks(p["scaling"], p["opacity"], c.grad_threshold, c.percent_dense, c.cameras_extent, c.min_opacity,
                                  c.size_threshold)
        p, m = D.prune_points(p, m, ~prune)
        N2 = p["position"].shape[0]
        p, m = D.reorder_points(p, m, self._morton(p))
        self.last_change = dict(cloned=n_clone, split=n_split, pruned=int(prune.sum()), N=N2)
        self._build(p, m, self.opt.t)


def synthetic_video_params(sc, clock: FrameClock, device, attrs: int = 16, seed: int = 99, cubic_sigma: float = 0.002,
                           rot_sigma: float = 0.01) -> Dict[str, Tensor]:
    """the synthetic scene of SURVEY 8d parameterised as the reference's dynamic Gaussians (row a15): canonical position + cubic
    spline per segment, raw rotation + frozen polynomial / Fourier tables, logit opacity, log scale, SH, render attributes"""
    N, I = sc.N, clock.interval_num
    rng = np.random.default_rng(seed)
    op = np.clip(sc.opacity, 1e-4, 1 - 1e-4)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=device)
    return {"position": t(sc.xyz), "pos_cubic_node": t(cubic_sigma * rng.normal(size=(N, 4 * I * 3))), "rotation": t(sc.rotate),
            "rot_poly_feat": t(rot_sigma * rng.normal(size=(N, 4, 4))), "rot_fourier_feat": t(rot_sigma * rng.normal(size=(N, 8, 4))),
            "opacity": t(np.log(op / (1 - op))), "scaling": t(np.log(sc.scale)), "shs": t(sc.shs),
            "attrs": t(rng.uniform(-1, 1, size=(N, attrs)))}


@torch.no_grad()
def render_ground_truth(params: Dict[str, Tensor], clock: FrameClock, W: int, H: int, extr: Tensor, times1, times2,
                        bg: float = 0.0, K: int = 0) -> Dict[str, Tensor]:
    """the frames ``times1`` of a Gaussian set (track_gs = its positions at ``times2``) rendered with the step's own forward:
    ground truth of a synthetic clip -- rgb [F,3,H,W], depth [F,1,H,W], attr [F,3+A,H,W]"""
    from .gs.point_ops import compute_sh
    dev = params["position"].device
    F, N, I = len(times1), params["position"].shape[0], clock.interval_num
    A = params["attrs"].shape[1]
    fb = FrameBatch(F, N, W, H, 3 + 1 + 3 + A, dev)
    dirs = torch.zeros(N, 3, device=dev)
    dirs[:, 2] = 1.0
    rgb = compute_sh(params["shs"], 3, dirs)
    cubic = params["pos_cubic_node"].reshape(N, -1)
    pos2 = positions_batch_forward(frame_table(clock, list(times2), dev), params["position"], cubic, I, GAUSSIAN_MAJOR)
    sets = [dict(feature=rgb, bg=bg, taps=True), dict(feature="depth", bg=1.0),
            dict(feature=[pos2, params["attrs"]], bg=0.0, detach_opacity=True)]
    out = fb.render_dynamic_sets(clock, list(times1), extr, sets, position=params["position"], pos_cubic_node=cubic,
                                 rotation=params["rotation"], rot_poly_feat=params["rot_poly_feat"],
                                 rot_fourier_feat=params["rot_fourier_feat"], opacity=params["opacity"], scaling=params["scaling"],
                                 cubic_layout=GAUSSIAN_MAJOR, K=K)
    return {"rgb": out[0].clone(), "depth": out[1].clone(), "attr": out[2].clone()}
