#!/usr/bin/env python
"""bench.py -- rendered frames/s, forward+backward, 300k dynamic Gaussians @ 854x480 (BASELINE.json
configs[1]) through the MI355X-native rasterizer behind the dptr.gs operator surface.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run the process is one of the N ranks; started
    plainly -- no RANK in the environment -- it launches the N ranks itself through torch.distributed.run on 127.0.0.1 and
    relays rank 0's JSON line.  Either way world size == --gpus is asserted and the line carries `ranks_seen`, an
    all-reduce of ones.  `--launch-check` stops after that rendezvous: the launcher's CPU test, gloo.)

A "step" is one SYNCHRONOUS gradient step of the frame-sharded data-parallel renderer: every rank renders
`--frames` frames of the clip forward+backward (SH colour -> ortho projection -> cov3d -> EWA -> tile sort
-> alpha blending, and the whole backward chain), the Gaussian gradients accumulate in one flat bucket,
ONE all-reduce of that bucket (RCCL; skipped at N=1), then one Adam step on the flat parameter buffer
(every rank applies the same update) -- only then does the next step's forward start.  Weak scaling:
frames per rank fixed, so N ranks render N*frames frames per step (N=8, frames=25 -> the 200-frame
configs[2]).  value = frames rendered by all ranks / wall time (max over ranks).

Paths (same images and gradients, tests/test_gpu_frames.py, test_gpu_fused.py):
  default      frame batch: every kernel takes the frame as a grid dimension (splatter_a_video_amd.frames),
               the Gaussian-side backward and the SH kernels run once per step
  --per-frame  the fused per-frame operators of round 1 (13 launches per frame)
  --ops        the reference's operator sequence through autograd (dptr_ortho_enhanced.py:282-349)

Inputs are synthetic (SURVEY.md 8d generator) and resident in HBM before the timed region.
The oracle (oracle/) is only used for the `cpu_baseline` leg (rank 0, N=1, bounded samples).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dptr.gs as gs  # noqa: E402
import splatter_a_video_amd._lib as L  # noqa: E402
from splatter_a_video_amd.frames import FrameBatch  # noqa: E402
from splatter_a_video_amd.optim import FlatAdam  # noqa: E402
from splatter_a_video_amd.parallel import FlatGradBucket  # noqa: E402
from splatter_a_video_amd.synth import make_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
HBM_ACHIEVABLE_GBS = 6300.0
# HBM/fabric bytes per launch from PMC counters and the compositing kernels' issue counters, measured offline on the
# same command line (tools/round_profile.sh: separate rocprofv3 --pmc passes; read requests sized by
# TCC_EA0_RDREQ_{32B,64B,128B}, WRITE_SIZE in KiB); stamped with the profile they come from and only reported when the
# bench runs the configuration they were taken at.
PMC_TAG = os.environ.get("SPLAT_PMC_TAG", "r06")
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", f"{PMC_TAG}_pmc_traffic.json")
PMC_COUNTER_FILE = os.path.join(ROOT, "profiles", f"{PMC_TAG}_pmc_blend_counters.json")
PMC_KERNEL_NAMES = {"blend_bwd": ("blend_bwd_quarter_kernel", "blend_bwd_mfma_kernel"), "blend_fwd": "blend_fwd_kernel", "tile_sort": "tile_sort_kernel",
                    "gauss_bwd": "frames_gauss_bwd_static_kernel", "pair_reduce": "pair_reduce_kernel",
                    "sh_fwd": "sh_fwd_kernel", "sh_bwd": "sh_bwd_kernel", "bin_scatter": "bin_scatter_kernel"}


def pmc_stamp_ok(rec):
    """an offline PMC record may only be quoted by the build it was taken from: its `build_id` (tools/pmc_*.py stamp it with
    splat_build_id() of the library that ran, next to `git_head`) must be the running library's"""
    return rec.get("build_id") is not None and rec.get("build_id") == L.build_id()


def pmc_traffic(kernel, tag_cfg):
    """(bytes per launch, source note) of a kernel from the offline PMC record, if it was taken at this configuration BY THIS
    BUILD of the library"""
    if not os.path.exists(PMC_TRAFFIC_FILE):
        return None, None
    try:
        rec = json.load(open(PMC_TRAFFIC_FILE))
        if rec.get("config") != tag_cfg:
            return None, None
        if not pmc_stamp_ok(rec):
            return None, (f"profiles/{PMC_TAG}_pmc_traffic.json is from build {rec.get('build_id')}, the running library is "
                          f"{L.build_id()}: not quoted")
        names = PMC_KERNEL_NAMES.get(kernel, "")
        k = None
        for nm in (names if isinstance(names, tuple) else (names,)):   # (the kernel the launch actually ran: first name on record)
            k = k or rec["kernels"].get(nm, None)
        if k is None:
            return None, None
        return k["read_bytes"] + k["write_bytes"], (f"profiles/{PMC_TAG}_pmc_traffic.json (build {rec.get('build_id')}, git "
                                                    f"{str(rec.get('git_head'))[:12]}; {rec.get('source', 'rocprofv3 --pmc, offline')})")
    except Exception:
        return None, None


def pmc_issue(kernel, tag_cfg):
    if not os.path.exists(PMC_COUNTER_FILE):
        return None
    try:
        rec = json.load(open(PMC_COUNTER_FILE))
        if rec.get("config") not in (None, tag_cfg) or not pmc_stamp_ok(rec):
            return None
        k = rec["kernels"].get(kernel, None)
        return None if k is None else dict(k["derived"], kernel_name=k.get("kernel_name"),
                                           source=f"profiles/{PMC_TAG}_pmc_blend_counters.json (build {rec.get('build_id')}, git "
                                                  f"{str(rec.get('git_head'))[:12]}; rocprofv3 --pmc, offline)")
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=25, help="frames per rank per gradient step")
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--clip", type=int, default=50, help="frames in the clip (per 8 ranks: 200)")
    ap.add_argument("--channels", type=int, default=0, help="extra feature channels (configs[4]: 32 -> no SH)")
    ap.add_argument("--per-frame", action="store_true", help="fused per-frame operators (round-1 path) instead of the frame batch")
    ap.add_argument("--per-frame-fused", action="store_true",
                    help="frame by frame through gs.rasterization_ortho: the whole frame behind one call of the C ABI per direction "
                         "(a pooled one-frame batch) -- half the host time of the operator path")
    ap.add_argument("--ops", action="store_true", help="per-operator chain through autograd")
    ap.add_argument("--dynamic", action="store_true",
                    help="parameterise the scene as the reference's dynamic Gaussians and run their per-frame evaluation "
                         "inside the fused preprocess (row a15 on the path; frame batch, or per-frame operators with --per-frame)")
    ap.add_argument("--render-iter", action="store_true",
                    help="the reference's real frame (row a1, dptr_ortho_enhanced.py:205-383): rgb through alpha_blending_enhanced "
                         "(K = 20, ndc + abs_ndc taps), depth (bg = 1) and 19 attribute channels (opacity detached) per frame, "
                         "through the native OrthoEnhancedRenderer (per-frame operators, shared forward pass)")
    ap.add_argument("--attr-channels", type=int, default=19,
                    help="--render-iter (frame batch): width of the attribute set (the trainer's plans: 19 = track_gs + its render "
                         "attributes; 1 / 3 / 4 = ['mask_attribute'] / ['dino_attribute'] / both, src/trainer_fragGS.py:657,1214,1272)")
    ap.add_argument("--train-step", action="store_true",
                    help="the reference's whole training step composed from the native pieces (splatter_a_video_amd/train_step.py; "
                         "src/trainer_fragGS.py:736-790): two dynamic evaluations, the training frame with track_gs = "
                         "position(ids2), L1 losses, K = 5 neighbours + ARAP per pair, backward, all-reduce, Adam, densification "
                         "statistics (+ one clone / split / prune / rebuild, amortised over its interval); --frames pairs per rank "
                         "and step")
    ap.add_argument("--owner-sharded", action="store_true",
                    help="--train-step at N > 1: the spline table's gradient reduced to the owners of its time blocks, their Adam "
                         "moments sharded, updated blocks gathered (parallel.owner_reduce / owner_gather, optim.OwnerShardedAdam) "
                         "instead of one all-reduce of the flat bucket + replicated Adam")
    ap.add_argument("--exchange-positions", action="store_true",
                    help="--train-step --owner-sharded at N > 1: position(ids2) of a pair frame in another rank's time block is "
                         "evaluated by its owner and sent, its gradient sent back (TrainingStep(exchange_positions=True)): the spline "
                         "table is neither reduced nor gathered in a step; the ranks render contiguous time blocks of the clip")
    ap.add_argument("--stale-overlap", action="store_true",
                    help="stale-1 mode: double-buffered gradient bucket, the all-reduce of step s overlaps step s+1's frames "
                         "and no optimiser runs (NOT synchronous data parallelism; for comparison only)")
    ap.add_argument("--overlap", action="store_true",
                    help="EXACT overlap: the rank's frames run as two half-batches with a gradient buffer each; the first half's "
                         "all-reduce runs under the second half's forward + backward, the sums are added, one Adam step "
                         "(parallel.overlapped_halves_step: same parameters as the synchronous step to fp32 summation order)")
    ap.add_argument("--no-comm-analysis", action="store_true",
                    help="N > 1: skip the extra measurements of the line (all-reduce alone, step without the collective, the "
                         "--overlap variant)")
    ap.add_argument("--ref-flow", action="store_true",
                    help="the reference's LITERAL per-frame call sequence (dptr_ortho_enhanced.py:272-376): eager-torch orthographic "
                         "projection + EWA on the GPU, the synchronising sort_gaussian, three separate blends through autograd")
    ap.add_argument("--no-optimizer", action="store_true", help="skip the Adam step (forward+backward+all-reduce only)")
    ap.add_argument("--no-spatial-order", action="store_true",
                    help="keep the synthetic scene's random Gaussian order (default: Morton order of the screen positions, as "
                         "densify.spatial_order / reorder_points maintain it)")
    ap.add_argument("--scene", choices=["uniform", "clustered"], default="uniform",
                    help="clustered: 70 %% of the Gaussians inside blobs that cover 10 %% of the image (foreground objects; "
                         "tile-list imbalance stress, SURVEY 7 hard part ii)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run at N = 1: skip the compact lines of the other configurations (c4, c5, render_iter, per-frame "
                         "operators, --ref-flow) in `other_configs`")
    ap.add_argument("--no-extra-lines", action="store_true",
                    help="skip the second workload of the line (N = 1: the reference's training frame, --render-iter --dynamic)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="--gpus 1 without a launcher: initialise the RCCL process group anyway (world size 1 on 127.0.0.1), so that "
                         "`ranks_seen` comes from a real RCCL all-reduce and the step's collective executes on the one GPU of the box")
    ap.add_argument("--zero1", action="store_true",
                    help="ZeRO-1 schedule of the step: reduce-scatter of the flat gradient buffer, Adam on this rank's 1 / N block "
                         "(moments sharded), all-gather of the updated parameters (parallel.Zero1Shards); the default line at N > 1 "
                         "times it beside `synchronous` and `overlap_exact`")
    ap.add_argument("--unfused-l1", action="store_true",
                    help="--train-step: the L1 losses as three splat_l1_loss_grad launches + gradient images (A/B of the loss-fused backward)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: start / join the N ranks, all-reduce ones, print {n_gpus, ranks_seen} and exit")
    return ap.parse_args()


def self_launch(a) -> int:
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks here (one per GPU, rendezvous on
    127.0.0.1) and hand their output through.  Reference wiring: src/train.py:19-31,210-213 (one process per GPU, NCCL
    process group, DistributedSampler shards the frames)."""
    import socket
    import subprocess
    backend = os.environ.get("SPLAT_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU over RCCL; "
                         "SPLAT_BENCH_BACKEND=gloo lets ranks share a device for control-flow tests)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SPLAT_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class FrameRenderer:
    """Frame-sharded DP unit: parameters replicated, gradients of all local frames accumulate into
    one flat bucket (views), one all-reduce + one Adam step per gradient step."""

    def __init__(self, sc, device, frames, C_extra=0, mode="batch", dynamic=False, stale_overlap=False, optimizer=True,
                 halves=False, attr_channels=19, zero1=False):
        self.sc = sc
        self.mode = mode
        self.dynamic = dynamic
        self.capacity = None      # pair capacity of the sync-free sort (learned on the first frame)
        self.sort_status = []
        self.dev = device
        self.F = len(frames)
        N = sc.N
        self.W, self.H = sc.W, sc.H
        self.use_sh = C_extra == 0
        src = dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity)
        if dynamic:
            # the same Gaussians parameterised as the reference's dynamic model (row a15): canonical position + cubic
            # spline per 5-frame segment (native segment-major table), raw rotation + frozen polynomial / Fourier
            # tables, logit opacity, log scale
            from splatter_a_video_amd.dynamics import FrameClock, to_segment_major
            self.clock = FrameClock(sc.F)
            I = self.clock.interval_num
            rng = np.random.default_rng(99)
            cub = (0.002 * rng.normal(size=(N, 4 * I * 3))).astype(np.float32)
            op = np.clip(sc.opacity, 1e-4, 1 - 1e-4)
            src = dict(pos_cubic_node=to_segment_major(torch.as_tensor(cub), I).numpy(), rotation=sc.rotate,
                       opacity=np.log(op / (1 - op)).astype(np.float32), scaling=np.log(sc.scale).astype(np.float32))
            self.position = torch.as_tensor(sc.xyz, device=device)
            self.rot_poly = torch.as_tensor((0.01 * rng.normal(size=(N, 4, 4))).astype(np.float32), device=device)
            self.rot_fourier = torch.as_tensor((0.01 * rng.normal(size=(N, 8, 4))).astype(np.float32), device=device)
        if self.use_sh:
            src["shs"] = sc.shs
        else:
            src["feature"] = sc.feature
        # the reference's training frame (render_iter of its dynamic Gaussians): the first three of the 19 attribute channels are
        # track_gs = position(ids2) of a pair frame (src/trainer_fragGS.py:506-511), read per frame as a feature source; the
        # other 16 are the model's attributes (mask 1 + pos_poly_feat 12 + dino 3, src/configs/frag_gs_v10.yaml:115-118)
        self.track = dynamic and mode == "render_iter"
        if mode in ("render_iter", "render_iter_frame", "ref_flow"):
            self.A = 19 if (self.track or mode != "render_iter") else int(attr_channels)
            attrs = np.random.default_rng(7).uniform(-1, 1, size=(N, 16 if self.track else self.A)).astype(np.float32)
            if mode == "render_iter_frame":
                # the per-frame renderer takes its attributes by name, as the reference's model holds them: one parameter
                # tensor each (views of ONE [N, 19] parameter made autograd copy both slices per frame in either direction)
                src["mask_attribute"] = np.ascontiguousarray(attrs[:, :1])
                src["dino_attribute"] = np.ascontiguousarray(attrs[:, 1:])
            else:
                src["attrs"] = attrs
        # stale-1 mode only: two gradient buffers, the all-reduce of step s runs on RCCL's stream while step s+1 fills the other
        self.overlap = bool(stale_overlap) and dist.is_available() and dist.is_initialized()
        # --overlap (exact): two half-batches, a gradient buffer each (parallel.overlapped_halves_step)
        self.halves = bool(halves) and mode in ("batch", "render_iter") and len(frames) >= 2 and not self.overlap
        # --zero1: the flat buffer in `world` equal blocks (parallel.Zero1Shards): reduce-scatter -> Adam on 1 / world -> all-gather
        self.zero1 = bool(zero1) and optimizer and not self.overlap and not self.halves
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.bucket = FlatGradBucket({k: torch.as_tensor(v, device=device) for k, v in src.items()},
                                     buffers=2 if (self.overlap or self.halves) else 1, pad_to=4 * world if self.zero1 else 1)
        self.p = self.bucket.params
        # Adam on the flat buffer (the reference's optimiser, eps 1e-15).  The learning rate is kept small so that the
        # synthetic scene's statistics (pairs per frame, list lengths) stay put over the run's steps: the cost of the
        # update does not depend on it.
        self.opt = FlatAdam(self.bucket, 1e-6, eps=1e-15) if (optimizer and not self.overlap) else None
        if self.zero1:
            from splatter_a_video_amd.optim import OwnerShardedAdam
            from splatter_a_video_amd.parallel import Zero1Shards
            self.shards = Zero1Shards(self.bucket, world, dist.get_rank() if world > 1 else 0)
            self.opt = OwnerShardedAdam(self.bucket, self.shards, 1e-6, eps=1e-15)
        self.extr = torch.tensor(sc.extr, device=device)
        self.phase = torch.tensor(sc.phase, device=device)
        self.dirs = torch.zeros(N, 3, device=device)
        self.dirs[:, 2] = 1.0
        self.C = 3 if self.use_sh else C_extra
        g = torch.Generator(device="cpu").manual_seed(4321)
        self.dL_dout = torch.randn(self.C, self.H, self.W, generator=g).to(device)
        self.frames = list(frames)
        if dynamic:
            self.offs = self.frames
        else:
            self.offs = [self.offsets(f) for f in self.frames]
        if self.mode in ("render_iter", "render_iter_frame", "ref_flow"):
            from splatter_a_video_amd.renderer import OrthoEnhancedRenderer
            self.renderer = OrthoEnhancedRenderer(densify_abs_grad_enable=True)
            self.dL_depth = torch.randn(1, self.H, self.W, generator=g).to(device)
            self.dL_attr = torch.randn(self.A, self.H, self.W, generator=g).to(device)
        # the step's frames as ONE frame batch, or (--overlap) as two half-batches
        self.parts = []
        if self.mode in ("batch", "render_iter"):
            if self.mode == "batch" and self.C > 32:
                raise SystemExit("the frame batch composites at most 32 channels per call")
            cut = (self.F + 1) // 2 if self.halves else self.F
            for lo, hi in ((0, cut), (cut, self.F)):
                if hi > lo:
                    self.parts.append(self._part(lo, hi, N, device, g))
            self.batch = self.parts[0].batch
        self.last = {}
        self._feat_in = None

    @property
    def flat_grad(self):
        """the bucket's ACTIVE gradient buffer (after a step: the one the optimiser consumed)"""
        return self.bucket.flat_grad

    def _part(self, lo, hi, N, device, g):
        """frames [lo, hi) of the step as one FrameBatch with its offsets and image gradients"""
        class Part:
            pass
        pt = Part()
        pt.frames = self.frames[lo:hi]
        n = hi - lo
        rep = lambda t: t.unsqueeze(0).repeat(n, 1, 1, 1).contiguous()
        pt.off_all = None if self.dynamic else torch.stack(self.offs[lo:hi]).contiguous()
        if self.mode == "render_iter":
            pt.batch = FrameBatch(n, N, self.W, self.H, 3 + 1 + self.A, device, want_abs=True)
            pt.dL_sets = [rep(self.dL_dout), rep(self.dL_depth), rep(self.dL_attr)]
            if self.track:   # pair frames of the part's frames, their positions and the gradient those receive
                from splatter_a_video_amd.dynamics import frame_table
                pair = [int((17 * t + 11) % self.sc.F) for t in pt.frames]
                pt.tab2 = frame_table(self.clock, [t if t != u else (t + 1) % self.sc.F for t, u in zip(pair, pt.frames)], device)
                pt.pos2 = torch.empty(n, N, 3, device=device)
                pt.g_pos2 = torch.empty(n, N, 3, device=device)
        else:
            pt.batch = FrameBatch(n, N, self.W, self.H, self.C, device)
            pt.dL_all = rep(self.dL_dout)
        return pt

    def offsets(self, f):
        d = 0.05 * torch.sin(2.0 * np.pi * (f / float(self.sc.F)) + self.phase)
        off = torch.zeros(self.sc.N, 3, device=self.dev)
        off[:, 0] = d
        off[:, 1] = d
        return off

    # ------------------------------------------------------------------ all local frames of a step, one launch sequence
    def frames_batched(self, part=None):
        part = part or self.parts[0]
        p = self.p
        g = {k: self.bucket.grad(k) for k in p}
        feat = gs.compute_sh_into(p["shs"], 3, self.dirs, None, g["shs"]) if self.use_sh else p["feature"]
        if self.dynamic:
            from splatter_a_video_amd.dynamics import SEGMENT_MAJOR
            sink = {k: g[k] for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
            if not self.use_sh:
                sink["feature"] = g["feature"]
            out = part.batch.render_dynamic(self.clock, part.frames, self.extr, feat, position=self.position,
                                            pos_cubic_node=p["pos_cubic_node"], rotation=p["rotation"],
                                            rot_poly_feat=self.rot_poly, rot_fourier_feat=self.rot_fourier, opacity=p["opacity"],
                                            scaling=p["scaling"], cubic_layout=SEGMENT_MAJOR, bg=self.sc.bg, nearest=0.01,
                                            grad_sink=sink)
            out.backward(part.dL_all)
            self.last = dict(M=self.last.get("M", 0), T=self.batch.T)
            return
        sink = {"xyz": g["xyz"], "scales": g["scale"], "uquats": g["rotate"], "opacity": g["opacity"]}
        if not self.use_sh:
            sink["feature"] = g["feature"]
        out = part.batch.render(p["xyz"], p["scale"], p["rotate"], p["opacity"], feat, part.off_all, self.extr,
                                bg=self.sc.bg, nearest=0.01, grad_sink=sink)
        out.backward(part.dL_all)
        self.last = dict(M=self.last.get("M", 0), T=self.batch.T)

    # ------------------------------------------------------------------ the reference's real frame (row a1), frame batch
    def frames_render_iter_batched(self, part=None):
        part = part or self.parts[0]
        p = self.p
        g = {k: self.bucket.grad(k) for k in p}
        rgb = gs.compute_sh_into(p["shs"], 3, self.dirs, None, g["shs"])      # once per step (constant view direction)
        sets = [dict(feature=rgb, bg=self.sc.bg, taps=True), dict(feature="depth", bg=1.0),
                dict(feature=p["attrs"], bg=0.0, detach_opacity=True)]
        if self.dynamic:   # the reference's real training frame: its dynamic Gaussians through the three blends, track_gs included
            from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, positions_batch_backward, positions_batch_forward
            I = self.clock.interval_num
            positions_batch_forward(part.tab2, self.position, p["pos_cubic_node"], I, SEGMENT_MAJOR, out=part.pos2)
            L.check(L.lib().splat_fill_f32(L.ptr(part.g_pos2), ctypes.c_size_t(part.g_pos2.numel()), L.cf(0.0), L.stream()))
            sets[2] = dict(feature=[part.pos2, p["attrs"]], bg=0.0, detach_opacity=True)
            sink = {k: g[k] for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
            sink.update({"feature:1": part.g_pos2, "feature:2": g["attrs"]})
            out = part.batch.render_dynamic_sets(
                self.clock, part.frames, self.extr, sets, position=self.position, pos_cubic_node=p["pos_cubic_node"],
                rotation=p["rotation"], rot_poly_feat=self.rot_poly, rot_fourier_feat=self.rot_fourier, opacity=p["opacity"],
                scaling=p["scaling"], cubic_layout=SEGMENT_MAJOR, K=20, grad_sink=sink)
            torch.autograd.backward(list(out[:3]), part.dL_sets)
            positions_batch_backward(part.tab2, part.g_pos2, I, SEGMENT_MAJOR, None, g["pos_cubic_node"])
            self.last = dict(M=self.last.get("M", 0), T=self.batch.T)
            return
        out = part.batch.render_sets(p["xyz"], p["scale"], p["rotate"], p["opacity"], sets, part.off_all, self.extr, K=20,
                                     grad_sink={"xyz": g["xyz"], "scales": g["scale"], "uquats": g["rotate"], "opacity": g["opacity"]})
        torch.autograd.backward(list(out[:3]), part.dL_sets)
        self.last = dict(M=self.last.get("M", 0), T=self.batch.T)

    # ------------------------------------------------------------------ the reference's real frame (row a1), frame by frame
    def frames_render_iter(self):
        p = self.p
        rgb = self.renderer.colors(p["shs"])          # once per batch: the view direction is constant (render_batch)
        rgb_in = rgb.detach().requires_grad_(True)    # the frames' colour gradients are summed before the one SH backward
        attrs = {"mask_attribute": p["mask_attribute"], "dino_attribute": p["dino_attribute"]}
        row = torch.cat(list(attrs.values()), dim=-1)  # likewise the attribute row (render_batch: one concatenation per batch)
        row_in = row.detach().requires_grad_(True)
        for off in self.offs:
            r = self.renderer.render_iter(self.H, self.W, self.extr, p["xyz"] + off, p["opacity"], p["scale"], p["rotate"], None,
                                          num_idx=20, rgb=rgb_in, render_attributes=attrs, attribute_row=row_in)
            f = r["rendered_features_split"]
            torch.autograd.backward([f["rgb"], f["depth"], f["mask_attribute"], f["dino_attribute"]],
                                    [self.dL_dout, self.dL_depth, self.dL_attr[:1], self.dL_attr[1:]])
            self._radii = r["radii"]
        rgb.backward(rgb_in.grad)
        row.backward(row_in.grad)
        self.last = dict(M=self.last.get("M", 0), T=((self.W + 15) // 16) * ((self.H + 15) // 16))

    # ------------------------------------------------------------------ the reference's literal call sequence, frame by frame
    def frames_ref_flow(self, backward=True):
        """render_iter exactly as src/pointrix/renderer/dptr_ortho_enhanced.py:270-376 issues it, frame by frame: SH colours per
        frame, EAGER torch orthographic projection and EWA (tools/eager_ortho.py -- the reference keeps these two steps in
        torch), gs.compute_cov3d, the SYNCHRONISING gs.sort_gaussian, then three separate blends through autograd
        (alpha_blending_enhanced K = 20 with the ndc / abs_ndc taps, the depth with bg = 1, the attributes with
        opacity.detach()).  What an unmodified copy of the reference's renderer file gets from this library."""
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import eager_ortho as eo
        p = self.p
        W, H = self.W, self.H
        for off in self.offs:
            pos = p["xyz"] + off
            rgb = gs.compute_sh(p["shs"], 3, self.dirs)
            uv, depth = eo.project_point_ortho(pos, self.extr, W, H, nearest=0.01)
            visible = depth != 0
            cov3d = gs.compute_cov3d(p["scale"], p["rotate"], visible)
            conic, radius, tiles = eo.ewa_project_ortho(pos, cov3d, self.extr, uv, W, H, visible.squeeze(-1))
            idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
            ndc = torch.zeros_like(uv, requires_grad=True)
            abs_ndc = torch.zeros_like(uv, requires_grad=True)
            img, ncontrib, gs_idx = gs.alpha_blending_enhanced(uv, conic, p["opacity"], rgb, idx, tr, self.sc.bg, W, H, ndc, abs_ndc, K=20)
            dep = gs.alpha_blending(uv, conic, p["opacity"], depth, idx, tr, 1.0, W, H, ndc.detach())
            att = gs.alpha_blending(uv, conic, p["opacity"].detach(), p["attrs"], idx, tr, 0.0, W, H, ndc.detach())
            if backward:
                torch.autograd.backward([img, dep, att], [self.dL_dout, self.dL_depth, self.dL_attr])
            self.last = dict(M=idx.numel(), T=tr.shape[0])

    def eager_steps_only(self):
        """the two eager-torch steps of the literal flow alone (projection, cov3d, EWA forward + backward through autograd) for
        all local frames: what of the --ref-flow frame is framework time rather than this library's kernels"""
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import eager_ortho as eo
        p = self.p
        W, H = self.W, self.H
        for off in self.offs:
            pos = p["xyz"] + off
            uv, depth = eo.project_point_ortho(pos, self.extr, W, H, nearest=0.01)
            visible = depth != 0
            cov3d = gs.compute_cov3d(p["scale"], p["rotate"], visible)
            conic, radius, tiles = eo.ewa_project_ortho(pos, cov3d, self.extr, uv, W, H, visible.squeeze(-1))
            (uv.sum() + depth.sum() + conic.sum()).backward()

    # ------------------------------------------------------------------ one frame (round-1 paths)
    def frame(self, off):
        """one frame, forward + backward.  ``frame`` mode: fused per-frame operators whose backward adds the parameter
        gradients straight into the flat bucket, sort without a host sync; ``ops`` mode: the reference's operator
        sequence (dptr_ortho_enhanced.py:282-349) through autograd.  Same images, same gradients."""
        p = self.p
        W, H = self.W, self.H
        opacity = p["opacity"]
        fused = self.mode == "frame"
        if self.mode == "frame_fused":
            # the whole frame behind ONE call of the C ABI per direction (gs.rasterization_ortho: a pooled one-frame batch)
            g = {k: self.bucket.grad(k) for k in self.p}
            feat = self._feat_in if self._feat_in is not None else p["feature"]
            img = gs.rasterization_ortho(p["xyz"], p["scale"], p["rotate"], p["opacity"], feat, self.extr, W, H, self.sc.bg, offset=off,
                                         nearest=0.01, grad_sink={"xyz": g["xyz"], "scales": g["scale"], "uquats": g["rotate"],
                                                                  "opacity": g["opacity"]})
            img.backward(self.dL_dout)
            from splatter_a_video_amd.frames import frame_rasterization
            self.last = dict(M=int(frame_rasterization.last.pairs.max().item()) if "M" not in self.last else self.last["M"],
                             T=frame_rasterization.last.T)
            return img
        if self.dynamic:
            from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, frame_preprocess
            g = {k: self.bucket.grad(k) for k in self.p}
            feat = (self._feat_in if self._feat_in is not None else
                    gs.compute_sh_into(p["shs"], 3, self.dirs, None, g["shs"]) if self.use_sh else p["feature"])
            uv, depth, conic, radius, tiles, opacity = frame_preprocess(
                self.clock, off, self.extr, W, H, position=self.position, pos_cubic_node=p["pos_cubic_node"],
                rotation=p["rotation"], rot_poly_feat=self.rot_poly, rot_fourier_feat=self.rot_fourier,
                opacity=p["opacity"], scaling=p["scaling"], nearest=0.01, cubic_layout=SEGMENT_MAJOR,
                grad_sink={k: g[k] for k in ("pos_cubic_node", "rotation", "opacity", "scaling")})
        elif fused:
            g = {k: self.bucket.grad(k) for k in self.p}
            feat = (self._feat_in if self._feat_in is not None else
                    gs.compute_sh_into(p["shs"], 3, self.dirs, None, g["shs"]) if self.use_sh else p["feature"])
            uv, depth, conic, radius, tiles = gs.preprocess_ortho(
                p["xyz"], p["scale"], p["rotate"], self.extr, W, H, nearest=0.01, offset=off,
                grad_sink={"xyz": g["xyz"], "scales": g["scale"], "uquats": g["rotate"]})
        else:
            pos = p["xyz"] + off
            feat = gs.compute_sh(p["shs"], 3, self.dirs) if self.use_sh else p["feature"]
            uv, depth = gs.project_point_ortho(pos, self.extr, W, H, nearest=0.01)
            visible = depth != 0
            cov3d = gs.compute_cov3d(p["scale"], p["rotate"], visible)
            conic, radius, tiles = gs.ewa_project_ortho(pos, cov3d, self.extr, uv, W, H, visible)
        if not fused and not self.dynamic:
            idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        elif self.capacity is None:   # first frame: learn the pair count with the synchronising sort
            idx, tr, _ = gs.sort_gaussian_capped(uv, depth, W, H, radius, None, conic.detach(), opacity.detach())
            self.capacity = int(idx.numel() * 1.25) + 1024
        else:   # (conic + opacity: only the pairs whose tile the splat can reach -- the reach masks of the frame batch)
            idx, tr, st = gs.sort_gaussian_capped(uv, depth, W, H, radius, self.capacity, conic.detach(), opacity.detach())
            self.sort_status.append(st)
        # densification tap, as the reference renderers create it (dptr.py:151-162)
        ndc = torch.zeros_like(uv, requires_grad=True)
        img = gs.alpha_blending(uv, conic, opacity, feat, idx, tr, self.sc.bg, W, H, ndc)
        img.backward(self.dL_dout)
        self.last = dict(M=idx.numel() if self.mode == "ops" else self.last.get("M", idx.numel()), T=tr.shape[0])
        return img

    def forward_only(self):
        """the forward pass of all local frames alone (SH colours -> preprocess -> binning -> sort -> compositing), no graph"""
        p = self.p
        if self.mode == "ref_flow":
            with torch.no_grad():
                return self.frames_ref_flow(backward=False)
        with torch.no_grad():
            if self.halves:      # (the two half-batches one after the other)
                return [self._forward_only_part(pt) for pt in self.parts]
            return self._forward_only_part(self.parts[0]) if self.parts else None

    def _forward_only_part(self, part):
        p = self.p
        with torch.no_grad():
            if self.mode == "batch":
                feat = gs.compute_sh(p["shs"], 3, self.dirs) if self.use_sh else p["feature"]
                if self.dynamic:
                    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR
                    return part.batch.render_dynamic(self.clock, part.frames, self.extr, feat, position=self.position,
                                                     pos_cubic_node=p["pos_cubic_node"], rotation=p["rotation"],
                                                     rot_poly_feat=self.rot_poly, rot_fourier_feat=self.rot_fourier,
                                                     opacity=p["opacity"], scaling=p["scaling"], cubic_layout=SEGMENT_MAJOR,
                                                     bg=self.sc.bg, nearest=0.01)
                return part.batch.render(p["xyz"], p["scale"], p["rotate"], p["opacity"], feat, part.off_all, self.extr,
                                         bg=self.sc.bg, nearest=0.01)
            if self.mode == "render_iter":
                rgb = gs.compute_sh(p["shs"], 3, self.dirs)
                sets = [dict(feature=rgb, bg=self.sc.bg, taps=True), dict(feature="depth", bg=1.0),
                        dict(feature=p["attrs"], bg=0.0, detach_opacity=True)]
                if self.dynamic:
                    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, positions_batch_forward
                    positions_batch_forward(part.tab2, self.position, p["pos_cubic_node"], self.clock.interval_num, SEGMENT_MAJOR, out=part.pos2)
                    sets[2] = dict(feature=[part.pos2, p["attrs"]], bg=0.0, detach_opacity=True)
                    return part.batch.render_dynamic_sets(
                        self.clock, part.frames, self.extr, sets, position=self.position, pos_cubic_node=p["pos_cubic_node"],
                        rotation=p["rotation"], rot_poly_feat=self.rot_poly, rot_fourier_feat=self.rot_fourier,
                        opacity=p["opacity"], scaling=p["scaling"], cubic_layout=SEGMENT_MAJOR, K=20)
                return part.batch.render_sets(p["xyz"], p["scale"], p["rotate"], p["opacity"], sets, part.off_all, self.extr, K=20)
        return None

    def scene_stats(self):
        """list statistics of the last rendered batch (host sync): pairs, mean tile-list length, mean / max last-contributor
        index per pixel, list entries the pixels walk (sum of ncontrib: what the reference's per-pixel loops visit up to the
        last contributor)"""
        if self.mode not in ("batch", "render_iter"):
            return None
        B = self.batch
        pairs = B.pairs.double()
        nc = B.ncontrib.double()
        tr = B.tile_range.long()
        ln = (tr[..., 1] - tr[..., 0]).double()
        return {"pairs_per_frame": float(pairs.mean()), "tiles": B.T, "mean_list_length": float(ln.mean()),
                "max_list_length": int(ln.max()), "list_length_p99": float(torch.quantile(ln.flatten()[:1 << 24].float(), 0.99)),
                "mean_ncontrib": float(nc.mean()), "max_ncontrib": int(nc.max()),
                "walked_entries_per_frame": float(nc.sum() / B.F), "mean_final_T": float(B.final_T.double().mean())}

    def reference_list_stats(self):
        """pairs and walked list entries of the REFERENCE's lists for the same frames (host sync, after the timed region): one
        forward of the first part on a scratch batch with the reach masks off -- every tile of a splat's bounding square gets its
        pair (include/utils.h:17-37), as the reference's sort_gaussian creates them.  SURVEY 8d prices the path per pair of THAT
        rule, so `roofline` and the per-kernel algorithmic bytes use these counts; the lists the kernels actually walk
        (`tile_pairs_M`, `scene_stats`) hold only the pairs whose tile the splat can reach."""
        if self.mode not in ("batch", "render_iter") or not self.parts:
            return None
        from splatter_a_video_amd import frames as FR
        part = self.parts[0]
        b0 = part.batch
        old = FR.OPTIONS["reach"]
        FR.OPTIONS["reach"] = False
        try:
            part.batch = FrameBatch(b0.F, b0.P, b0.W, b0.H, b0.C, b0.dev, want_abs=b0.want_abs)
            with torch.no_grad():
                self._forward_only_part(part)
            fb = part.batch
            return {"pairs_per_frame": float(fb.pairs.double().mean()), "max_pairs_per_frame": int(fb.pairs.max()),
                    "walked_entries_per_frame": float(fb.ncontrib.double().sum() / fb.F),
                    "rule": "every tile of the splat's bounding square (the reference's sort_gaussian)"}
        finally:
            part.batch = b0
            FR.OPTIONS["reach"] = old

    def step(self, collective=True):
        """one gradient step: local frames forward+backward, ONE all-reduce of the flat bucket (skipped when no process
        group exists, and in rank 0's private kernel-timing pass), one Adam step; returns when everything is enqueued"""
        if self.halves:
            from splatter_a_video_amd.parallel import overlapped_halves_step
            run = self.frames_batched if self.mode == "batch" else self.frames_render_iter_batched
            if collective:
                overlapped_halves_step(self.bucket, lambda: run(self.parts[0]), lambda: run(self.parts[1]), self.opt)
            else:                 # rank 0's private kernel-timing pass: same launches, no collective
                for k, pt in enumerate(self.parts):
                    self.bucket.activate(k)
                    self.bucket.zero_grad()
                    run(pt)
                self.bucket.fold(0, 1)
                if self.opt is not None:
                    self.opt.step()
            return
        self.bucket.swap()        # stale-1 mode: waits for the collective that last used the buffer we switch to
        self.bucket.zero_grad()
        if self.mode == "batch":
            self.frames_batched()
        elif self.mode == "render_iter":
            self.frames_render_iter_batched()
        elif self.mode == "render_iter_frame":
            self.frames_render_iter()
        elif self.mode == "ref_flow":
            self.frames_ref_flow()
        else:
            # fused per-frame operators: the SH colours once per step (the view direction is constant, as in the frame batch and
            # in the native renderer), the frames' colour gradients summed before the one SH backward; the reference's operator
            # chain (--ops) evaluates them per frame as its renderer does
            feat = None
            if self.mode in ("frame", "frame_fused") and self.use_sh:
                feat = gs.compute_sh_into(self.p["shs"], 3, self.dirs, None, self.bucket.grad("shs"))
                self._feat_in = feat.detach().requires_grad_(True)
            for off in self.offs:
                self.frame(off)
            if feat is not None:
                feat.backward(self._feat_in.grad)
                self._feat_in = None
        if self.zero1:
            from splatter_a_video_amd.parallel import owner_gather, owner_reduce
            if collective:
                owner_reduce(self.bucket, self.shards)
            self.opt.step()
            if collective:
                owner_gather(self.bucket, self.shards)
            return
        if collective and dist.is_available() and dist.is_initialized():
            self.bucket.all_reduce(async_op=self.overlap)     # synchronous unless --stale-overlap
        if self.opt is not None:
            self.opt.step()

    def finish(self):
        """all outstanding gradient collectives have completed (end of the timed region / of training)"""
        self.bucket.wait()

    def check_sorts(self):
        """after the timed region: every capacity-bounded sort of the run fitted (host sync)"""
        m = 0
        for pt in self.parts:
            m = max(m, pt.batch.check())
        for st in self.sort_status:
            m = max(m, st.check())
        self.sort_status.clear()
        if self.mode == "frame_fused":
            from splatter_a_video_amd.frames import frame_rasterization
            m = max(m, frame_rasterization.last.check())
        if self.mode == "render_iter_frame" and getattr(self, "_radii", None) is not None:
            # pairs of the last frame, re-derived from its screen-space geometry (the renderer's synchronising sort does
            # not expose its count)
            with torch.no_grad():
                uv, depth, conic, radius, tiles = gs.preprocess_ortho(self.p["xyz"] + self.offs[-1], self.p["scale"], self.p["rotate"],
                                                                      self.extr, self.W, self.H, nearest=0.01)
                m = int(tiles.sum().item())
        if m:
            self.last["M"] = m


def kernel_bytes(name, N, M, HW, C, T, use_sh, fpl=1.0, sets=False, nparam=None, sh_acc=True, dyn=False):
    """Algorithmic HBM bytes of ONE LAUNCH that covers ``fpl`` frames (SURVEY.md 8d bookkeeping, per kernel): what the frames
    share (parameters, SH coefficients, gradient accumulators, Adam state) is counted ONCE per launch, per-frame arrays
    (screen-space geometry, pair lists, records, images) once per frame.  ``sets``: the renderer's three feature sets
    (3 + 1 + 19 channels) in one pass."""
    F_in = 192 if use_sh else 0
    if nparam is None:       # floats of the flat parameter buffer the optimiser streams (callers pass the bucket's real size)
        nparam = N * (3 + 3 + 4 + 1 + (48 if use_sh else C))
    if sets:
        C = 23
    rec_g = 10 if sets else 8                # gradient floats of a pair record in front of the feature gradients
    shared = {
        "sh_fwd": N * (F_in + 12 + 1 + 12 + 3),
        # dirs, visible, clamped, dL_dcolors in; dL_dshs out (+ read when it accumulates into the bucket).  The coefficients
        # themselves are only read for a direction gradient, which no caller on the path requests.
        "sh_bwd": N * (12 + 1 + 3 + 12 + F_in * (2 if sh_acc else 1)),
        # static parameters in: xyz, scale, quat
        "preprocess_fwd": N * (12 + 12 + 16),
        # dynamic parameters in (position, rotation + 192 B of tables, opacity, scaling), activated opacity out
        "frame_preprocess_fwd": N * (12 + 16 + 192 + 4 + 12 + 4),
        "blend_pack": N * 4 * C,                               # the feature rows are shared by the frames
        # Gaussian-side backward: parameters in, read-modify-write of their gradients (+ feature gradients), taps out
        "gauss_bwd": N * (40 + 2 * (44 + 4 * C)),
        "adam_step": 7 * 4 * nparam,
    }
    per_frame = {
        "project_point_fwd": N * (12 + 8 + 4),
        "project_point_bwd": N * (4 + 8 + 4 + 12 + 12),
        "preprocess_fwd": N * (12 + 8 + 4 + 12 + 4 + 4),       # offset in; uv, depth, conic, radius, tiles out
        "preprocess_bwd": N * (12 + 12 + 12 + 16 + 4 + 4 + 8 + 4 + 12 + 2 * (12 + 12 + 16)),
        "frame_preprocess_fwd": N * (48 + 8 + 4 + 12 + 4 + 4),  # 48-byte spline segment in; uv, depth, conic, radius, tiles out
        "frame_preprocess_bwd": N * (12 + 48 + 16 + 192 + 4 + 12 + 4 + 4 + 8 + 4 + 12 + 4 + 2 * (48 + 16 + 4 + 12)),
        "cov3d_fwd": N * (12 + 16 + 1 + 24),
        "cov3d_bwd": N * (12 + 16 + 1 + 24 + 12 + 16),
        "ewa_fwd": N * (12 + 24 + 8 + 1 + 12 + 4 + 4),
        "ewa_bwd": N * (12 + 24 + 4 + 12 + 24),
        "bin_count": N * 12,
        "bin_colscan": 0,
        "bin_tilescan": T * 12,
        "bin_scatter": N * 16 + M * 8,
        "tile_sort": M * (8 + 4),
        "blend_pack": N * 28 + N * ((8 + C + 15) // 16 * 64),
        "blend_fwd": M * (28 + 4 * C) + HW * (4 * C + 8),
        # gather again + one record per pair (plain store) + dL_dout / final_T / ncontrib
        "blend_bwd": M * (28 + 4 * C) + M * (4 * rec_g + 4 * C) + HW * (4 * C + 8),
        # reads the records through the inverse pair map, writes the per-Gaussian gradients
        "pair_reduce": M * (4 + 4 * rec_g + 4 * C) + N * (4 + 4 * rec_g + 4 * C),
        # the records of the frame (+ dynamic Gaussians: the frame's 48-byte spline segment in, its gradient out: DESIGN 4)
        "gauss_bwd": M * (4 * rec_g + 4 * C) + N * 8 + (N * 88 if dyn else 0),
    }
    return shared.get(name, 0) + fpl * per_frame.get(name, 0)


def compute_rates(stats, kernels):
    """pixel-Gaussian evaluations: list entries the pixels walk up to their last contributor (sum of ncontrib; the reference's
    loops visit at least these) at the reference's arithmetic per visit -- 16 flops forward (alpha_blending.cu:78-100: dx dy
    power exp alpha T), 14 flops recompute backward (:196-203) -- as a LOWER bound of the useful work (contributing visits add
    2C resp. ~32 + 7C more); FP32 vector / matrix peak 157.3 TFLOP/s"""
    ev = stats["walked_entries_per_frame"]
    comp = {"evaluations_per_frame": ev, "peak_TFLOPs": 157.3, "unit": "TFLOP/s",
            "definition": "sum over pixels of ncontrib in the REFERENCE's lists (entries its per-pixel loops walk up to the last "
                          "contributor) x 16 flops (forward) / 14 flops (backward recompute): lower bound of the useful arithmetic"}
    for kn, fl in (("blend_fwd", 16.0), ("blend_bwd", 14.0)):
        if kn in kernels and kernels[kn]["us_per_frame"] > 0:
            rate = ev / (kernels[kn]["us_per_frame"] * 1e-6)
            comp[kn] = {"G_evaluations_per_s": round(rate / 1e9, 2), "TFLOPs_lower_bound": round(rate * fl / 1e12, 3),
                        "frac_of_fp32_peak": round(rate * fl / 157.3e12, 4)}
    return comp


def cpu_baseline_torch(sc, C_extra, budget_s=12.0):
    """the PyTorch-eager restatement of the reference's semantics (oracle/torch_eager.py), float32,
    one frame forward+backward.  configs[0] (10k Gaussians, 256x256): median of 5 runs after one warm-up.  The bench's
    own workload: ONE bounded sample -- the compositing loop (tile groups, longest lists first) is cut after `budget_s`
    seconds and the frame time extrapolated from the fraction of the tile-list entries done (stated in `sample`).
    Eager torch on CPU does not scale to all host cores for these op sizes (intra-op thread fan-out dominates), so the
    thread count is capped at 32 and reported in `cores`."""
    from oracle import torch_eager as te
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    C = C_extra if C_extra else 3
    use_sh = not C_extra
    g1 = np.random.default_rng(4321).normal(size=(C, 256, 256)).astype(np.float32)
    c1 = make_scene(10000, 256, 256, F=sc.F, C=C_extra, seed=1234)
    t_c1, M1 = [], None
    for _ in range(6):
        r = te.frame_forward_backward(c1, 0, g1, use_sh=use_sh, budget_s=5.0)
        t_c1.append(r["seconds"]); M1 = r["M"]
        if sum(t_c1) > 30.0:
            break
    m1 = statistics.median(t_c1[1:]) if len(t_c1) > 1 else t_c1[0]
    g2 = np.random.default_rng(4321).normal(size=(C, sc.H, sc.W)).astype(np.float32)
    r2 = te.frame_forward_backward(sc, 0, g2, use_sh=use_sh, budget_s=budget_s)
    return {"value": 1.0 / r2["seconds"], "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 frame fwd+bwd of the same workload ({sc.N} Gaussians, {sc.W}x{sc.H}, M={r2['M']}), PyTorch-eager float32 "
                      f"restatement (oracle/torch_eager.py): compositing cut after {budget_s:.0f} s = {100 * r2['fraction']:.1f} % of "
                      f"the tile-list entries, frame time extrapolated to {r2['seconds']:.1f} s; configs[0] (10k Gaussians, "
                      f"256x256, M={M1}): median of {max(len(t_c1) - 1, 1)} runs after 1 warm-up {m1 * 1e3:.0f} ms = "
                      f"{1.0 / m1:.2f} frames/s"}


def cpu_baseline_c(sc, C_extra):
    """Second, stronger baseline: one frame forward+backward with the C oracle, OpenMP over tiles on all host cores."""
    import oracle
    threads = os.cpu_count() or 1
    if not oracle.has_openmp():
        threads = 1
    oracle.set_threads(threads)
    xyz = sc.positions(0)
    feat = sc.feature if C_extra else None
    shs = None if C_extra else sc.shs
    C = C_extra if C_extra else 3
    g = np.random.default_rng(4321).normal(size=(C, sc.H, sc.W)).astype(np.float32)
    t0 = time.perf_counter()
    (out, fT, nc), saved = oracle.render_forward(xyz, sc.scale, sc.rotate, sc.opacity, feat, sc.intr, sc.extr, sc.W, sc.H,
                                                 sc.bg, ortho=True, shs=shs)
    oracle.render_backward(xyz, sc.scale, sc.rotate, sc.opacity, sc.intr, sc.extr, sc.W, sc.H, sc.bg, saved, g,
                           ortho=True, shs=shs)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 frame fwd+bwd of the same workload (M={saved['idx_sorted'].size}), C oracle with OpenMP over "
                      f"tiles, {dt:.2f} s wall"}


def train_step_line(a, sc, dev, frames, timed, world, launched):
    """The reference's training step (src/trainer_fragGS.py:736-790) at this workload through train_step.TrainingStep: ms per step,
    frames per second and the per-phase split; densification (clone / split / prune with the Adam moments + Morton reorder +
    rebuild of every buffer at the new count) timed once and amortised over the reference's interval of 100 steps.  The
    ground-truth frames are rendered from the scene's own parameters; the trained copy starts from perturbed ones."""
    from splatter_a_video_amd import train_step as TS
    from splatter_a_video_amd.dynamics import FrameClock
    clock = FrameClock(sc.F)
    # the detached set = track_gs (3 channels) + the model's attributes: --attr-channels 19 (default) = 16 attributes, the renderer's
    # own plan; --attr-channels 7 / 4 = the trainer's ['dino_attribute'] / ['mask_attribute'] rows behind track_gs (4 / 1 attributes)
    n_attrs = max(1, int(a.attr_channels) - 3)
    truth = TS.synthetic_video_params(sc, clock, dev, attrs=n_attrs)
    extr = torch.tensor(sc.extr, device=dev)
    t1 = list(frames)
    t2 = [int((17 * t + 11) % sc.F) for t in t1]
    t2 = [t if t != u else (t + 1) % sc.F for t, u in zip(t2, t1)]
    gt = TS.render_ground_truth(truth, clock, sc.W, sc.H, extr, t1, t2)
    gen = torch.Generator(device=dev).manual_seed(7)
    start = {k: v.clone() for k, v in truth.items()}
    for k, sg in (("shs", 0.1), ("attrs", 0.2), ("opacity", 0.3), ("scaling", 0.05)):
        start[k] = start[k] + sg * torch.randn(start[k].shape, device=dev, generator=gen)
    start["pos_cubic_node"] = torch.zeros_like(start["pos_cubic_node"])
    cfg = TS.DensifyConfig(cameras_extent=5.0)
    lr = {k: 1e-6 for k in TS.REFERENCE_LR}      # as everywhere in this file: small rates keep the scene's statistics put over the run
    st = TS.TrainingStep(start, clock, sc.W, sc.H, len(t1), extr, lr=lr, densify=cfg, K=20, owner_sharded=a.owner_sharded, zero1=a.zero1,
                         fused_l1=not a.unfused_l1, exchange_positions=a.exchange_positions)
    del truth
    if not (world > 1 and not a.zero1 and not a.owner_sharded):
        start = None
    dt = timed(lambda: st.step(t1, t2, gt))
    st.fb.check()
    loss = st.loss()
    ms_step = dt / a.steps * 1e3
    # per-phase split: one more step with events between the phases
    st.timing = True
    st.step(t1, t2, gt)
    phases = {k: round(v, 3) for k, v in st.phases().items()}
    st.timing = False
    # per-kernel HIP-event times of one more step (rank 0's own pass at N > 1 would enter collectives: N = 1 only)
    kern = None
    if world == 1 and not a.no_kernel_timing:
        torch.cuda.synchronize()
        L.profile_reset()
        L.profile_enable(True)
        st.step(t1, t2, gt)
        torch.cuda.synchronize()
        L.profile_enable(False)
        kern = {}
        for n in ("sh_fwd", "dynamic_positions_fwd", "knn_brute_bound", "knn_brute_merge", "knn_brute", "arap_energy", "frame_preprocess_fwd",
                  "bin_count", "bin_colscan", "bin_tilescan", "bin_scatter", "tile_sort", "blend_pack", "blend_fwd", "l1_loss_grad",
                  "blend_bwd", "gauss_bwd", "sh_bwd", "dynamic_positions_bwd", "fill", "adam_step", "densify"):
            ms, cnt = L.profile_read(n)
            if cnt:
                kern[n] = {"us_per_step": round(ms * 1e3, 1), "launches": cnt}
        for n in ("knn_brute",):      # (prefix match: the bound and merge launches are listed on their own)
            if n in kern:
                for sub in ("knn_brute_bound", "knn_brute_merge"):
                    if sub in kern:
                        kern[n]["us_per_step"] = round(kern[n]["us_per_step"] - kern[sub]["us_per_step"], 1)
                        kern[n]["launches"] -= kern[sub]["launches"]
        L.profile_reset()
    # densification once (every rank: the statistics were reduced, the decisions are identical), timed on its own
    n0 = st.N
    dens = []
    for _ in range(2):                            # twice: the first rebuild also grows the allocator's pools
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.densify()
        torch.cuda.synchronize()
        dens.append((time.perf_counter() - t0) * 1e3)
        if not dens[1:]:
            ch = dict(st.last_change)
        st.step(t1, t2, gt)                       # the step runs at the new count (buffers rebuilt, capacity re-measured)
        torch.cuda.synchronize()
        st.fb.check()
    dens_ms = min(dens)
    F = len(t1)
    amort = dens_ms / cfg.interval
    sharded = a.owner_sharded or a.zero1
    opt_desc = ("ZeRO-1: reduce-scatter of the flat gradient, Adam on this rank's 1 / N block, all-gather of the parameters" if a.zero1 else
                "owner-sharded spline table with POSITION EXCHANGE (position(ids2) from its owner, its gradient back: the table is "
                "neither reduced nor gathered)" if a.exchange_positions else
                "owner-sharded spline table (reduce to owner, sharded moments, gather)" if a.owner_sharded
                else "all-reduce of the flat bucket + replicated Adam")
    moments_MB = round(sum(t.numel() for t in ((st.opt.m_own, st.opt.v_own, st.opt.m_rep, st.opt.v_rep)
                                               if sharded else (st.opt.exp_avg, st.opt.exp_avg_sq))) * 4 / 1e6, 1)
    bucket_MB = round(st.bucket.flat_grad.numel() * 4 / 1e6, 1)
    pairs_M = int(st.fb.pairs.max().item())
    zero1 = exchange = None
    if start is not None:      # N > 1, default schedule: the same step under ZeRO-1 beside it (every rank takes part)
        try:
            del st
            torch.cuda.empty_cache()
            st = TS.TrainingStep(start, clock, sc.W, sc.H, len(t1), extr, lr=lr, densify=cfg, K=20, zero1=True)
            dtz = timed(lambda: st.step(t1, t2, gt))
            st.fb.check()
            zero1 = {"train_step_ms": round(dtz / a.steps * 1e3, 3), "value": round(F * a.steps * world / dtz, 2), "unit": "frames/s",
                     "adam_moments_MB_per_rank": round((st.opt.m_own.numel() + st.opt.v_own.numel()) * 4 / 1e6, 1),
                     "what": "the same step with zero1=True: reduce-scatter of the flat gradient, Adam on 1 / N, all-gather"}
        except Exception as e:   # noqa: BLE001
            zero1 = {"error": repr(e)[:300]}
        # ... and with the POSITION EXCHANGE (owner-sharded table; the ranks render contiguous time blocks of the clip, so these are
        # other pairs of the same count: position(ids2) from its owner, its gradient back -- no reduce / gather of the table)
        try:
            del st
            torch.cuda.empty_cache()
            tb1 = [(dist.get_rank() * (sc.F // world) + i) % sc.F for i in range(len(t1))]
            tb2 = [int((17 * t + 11) % sc.F) for t in tb1]
            tb2 = [t if t != u else (t + 1) % sc.F for t, u in zip(tb2, tb1)]
            truth_b = TS.synthetic_video_params(sc, clock, dev, attrs=n_attrs)
            gtb = TS.render_ground_truth(truth_b, clock, sc.W, sc.H, extr, tb1, tb2)
            del truth_b
            st = TS.TrainingStep(start, clock, sc.W, sc.H, len(t1), extr, lr=lr, densify=cfg, K=20, owner_sharded=True,
                                 exchange_positions=True)
            dtx = timed(lambda: st.step(tb1, tb2, gtb))
            st.fb.check()
            exchange = {"train_step_ms": round(dtx / a.steps * 1e3, 3), "value": round(F * a.steps * world / dtx, 2), "unit": "frames/s",
                        "adam_moments_MB_per_rank": round(sum(t.numel() for t in (st.opt.m_own, st.opt.v_own, st.opt.m_rep, st.opt.v_rep)) * 4 / 1e6, 1),
                        "what": "the same step count with owner_sharded=True, exchange_positions=True on contiguous time blocks: "
                                "position(ids2) evaluated by its owner and sent, its gradient sent back; the spline table is neither "
                                "reduced nor gathered"}
        except Exception as e:   # noqa: BLE001
            exchange = {"error": repr(e)[:300]}
        del start
    return {
        "metric": "training steps of the reference's trainer composed from the native pieces (src/trainer_fragGS.py:736-790), "
                  f"{F} (ids1, ids2) pairs per rank and step @480p, 300k Gaussians",
        "value": round(F * a.steps * world / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "train_step_ms": round(ms_step, 3), "ms_per_pair": round(ms_step / F, 4),
        "train_step_ms_with_densification_amortised": round(ms_step + amort, 3),
        "phases_ms": phases, "kernels_us_per_step": kern,
        "densify": {"ms_once": round(dens_ms, 2), "ms_first_and_second": [round(x, 2) for x in dens], "interval_steps": cfg.interval, "ms_per_step_amortised": round(amort, 3),
                    "gaussians_before": n0, **ch,
                    "what": "masks, clone, split (counter-based children), prune -- parameters and Adam moments --, Morton reorder, "
                            "flat bucket / Adam / frame batch rebuilt at the new count (host-synchronising: a handful of counts)"},
        "loss": round(loss, 6),
        "reference_context": "the reference publishes 15-20 min for 20 000 steps on an RTX 3090 = 45-60 ms per step of ONE pair "
                             "(SURVEY 6; paper App. A.1) -- other hardware, losses with SSIM / depth / flow terms: context only",
        "config": {"workload": f"{sc.N} dynamic Gaussians of the reference's model, {F} frame pairs/rank/step of a {sc.F}-frame "
                               f"{sc.W}x{sc.H} clip: SH deg 3 colours, position(ids1) + position(ids2), K = 5 neighbours of 512 "
                               "sampled vertices + ARAP per pair, render_iter's three blends (rgb enhanced K=20 with taps | depth | "
                               f"track_gs + {n_attrs} attribute channels, opacity detached), L1 on the three images, backward, "
                               "all-reduce, Adam on the flat buffer, densification statistics",
                   "equivalent_flags": "--train-step" + (" --owner-sharded" if a.owner_sharded else "") + (" --zero1" if a.zero1 else "")
                                       + (" --exchange-positions" if a.exchange_positions else "")
                                       + ("" if a.attr_channels == 19 else f" --attr-channels {a.attr_channels}"),
                   "tile_pairs_M": pairs_M, "grad_bucket_MB": bucket_MB, "optimizer": opt_desc,
                   "adam_moments_MB_per_rank": moments_MB},
        "zero1": zero1, "position_exchange": exchange}


def main():
    a = parse()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run (any N)
    if a.gpus > 1 and not launched:
        sys.exit(self_launch(a))
    if a.force_process_group and not launched and a.gpus == 1:
        # a world of one rank, rendezvous on 127.0.0.1: everything below runs as under the launcher (RCCL communicator, the
        # `ranks_seen` all-reduce, the step's collective on RCCL's stream)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        launched = True
        import splatter_a_video_amd.parallel as _par
        _par.MIN_WORLD = 1      # the owner / ZeRO-1 collectives (reduce-scatter, all-gather) run in a world of one rank too
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} runs inside a world of {world} ranks: launch it with --nproc-per-node {a.gpus} "
                         "(or without torchrun: it starts its own ranks)")
    backend = os.environ.get("SPLAT_BENCH_BACKEND", "nccl")
    have_gpu = torch.cuda.is_available()
    ranks_seen = 1
    if launched:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SPLAT_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a 1-GPU box (ranks share cuda:0)
        if have_gpu:
            if backend == "nccl" and torch.cuda.device_count() < world:
                raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, {torch.cuda.device_count()} visible")
            local = local % max(1, torch.cuda.device_count())    # rank -> GPU (gloo: ranks may share a device)
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        ones = torch.ones(1, device=torch.device("cuda", local) if (have_gpu and backend == "nccl") else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world == a.gpus, (ranks_seen, world, a.gpus)
    if a.launch_check:
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": ranks_seen, "backend": backend if launched else None,
                              "self_launched": os.environ.get("SPLAT_BENCH_SELF_LAUNCHED") == "1"}))
        if launched:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert have_gpu, "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    mode = ("ref_flow" if a.ref_flow else ("render_iter_frame" if a.per_frame else "render_iter") if a.render_iter else "ops" if a.ops
            else "frame_fused" if a.per_frame_fused else "frame" if a.per_frame else "batch")
    clip = max(a.clip, 25 * world)
    def build_scene(gaussians, width, height, channels):
        sc_ = make_scene(gaussians, width, height, F=clip, C=channels, seed=1234, clustered=0.7 if a.scene == "clustered" else 0.0)
        if not a.no_spatial_order:
            # setup, as a trainer does after initialisation and after every densification: Gaussians in Morton order of their
            # screen positions (densify.spatial_order; results do not depend on the order, the binning kernels' locality does)
            from splatter_a_video_amd.densify import spatial_order
            uv0, _ = gs.project_point_ortho(torch.tensor(sc_.positions(0), device=dev), torch.tensor(sc_.extr, device=dev), width,
                                            height, nearest=0.01)
            order = spatial_order(uv0, width, height).cpu().numpy()
            for k in ("xyz", "phase", "scale", "rotate", "opacity", "shs", "feature"):
                v = getattr(sc_, k)
                if v is not None:
                    setattr(sc_, k, np.ascontiguousarray(v[order]))
        return sc_

    sc = build_scene(a.gaussians, a.width, a.height, a.channels)
    # rank r renders frames {f : f mod world == r} of the step's frame batch
    frames = [((i * world + rank) % clip) for i in range(a.frames)]
    if a.exchange_positions:     # contiguous time blocks: a rank renders the frames whose spline segments it owns
        a.owner_sharded = True
        frames = [(rank * (clip // world) + i) % clip for i in range(a.frames)]

    def sync():
        torch.cuda.synchronize()
        if launched:
            dist.barrier()
            torch.cuda.synchronize()

    # setup, not workload: one step of a 512-Gaussian 64x64 scene through the same operators loads the library's code
    # objects and initialises the allocator, so that `--warmup 0` does not time HIP module loading
    tiny = make_scene(512, 64, 64, F=clip, C=a.channels, seed=1)
    Rt = FrameRenderer(tiny, dev, frames[:2], a.channels, mode=mode, dynamic=a.dynamic, optimizer=not a.no_optimizer, attr_channels=a.attr_channels)
    Rt.step(collective=False)
    torch.cuda.synchronize()
    del Rt, tiny
    if launched:   # communicator / channel set-up is setup as well (first collective of the process group)
        prime = torch.zeros(1 << 20, device=dev)
        dist.all_reduce(prime)
        torch.cuda.synchronize()
        del prime

    def timed(fn, finish=None):
        """W untimed + exactly K timed calls of fn between barrier + synchronize on both sides; max over ranks"""
        for _ in range(a.warmup):
            fn()
        if finish:
            finish()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        if finish:
            finish()          # stale-1 mode: the last step's all-reduce is inside the timed region
        sync()
        d = time.perf_counter() - t0
        if launched:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d

    if a.train_step:   # the composed training step as the line's workload (extra_lines of the default run carries it as well)
        line = train_step_line(a, sc, dev, frames, timed, world, launched)
        line.update({"ms_per_step": line["train_step_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                     "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "build_id": L.build_id()})
        if rank == 0:
            print(json.dumps(line))
        if launched:
            dist.barrier()
            dist.destroy_process_group()
        return

    R = FrameRenderer(sc, dev, frames, a.channels, mode=mode, dynamic=a.dynamic, stale_overlap=a.stale_overlap,
                      optimizer=not a.no_optimizer, halves=a.overlap, attr_channels=a.attr_channels, zero1=a.zero1)
    dt = timed(R.step, R.finish)
    R.check_sorts()   # the sync-free sorts of the timed steps all fitted their capacity (raises otherwise)

    # N > 1: what the collective costs (every rank takes part; rank 0 reports).  The step is synchronous, so the all-reduce of
    # the flat bucket is exposed by construction; these figures say how much of the step it is, and what the exact half-batch
    # overlap (--overlap) makes of it.  Failures here must not cost the line its headline: they are reported in `comm.error`.
    def comm_analysis(R_, dt_, mode_, dynamic_, with_overlap=True):
        if not (launched and world > 1) or a.no_comm_analysis or a.stale_overlap:
            return None
        comm = {"bucket_MB": round(R_.flat_grad.numel() * 4 / 1e6, 1), "backend": backend}
        try:
            buf = torch.zeros_like(R_.flat_grad)
            for _ in range(2):
                dist.all_reduce(buf)
            reps = 10
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                dist.all_reduce(buf)
            sync()
            tt = torch.tensor([(time.perf_counter() - t0) / reps], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ar = float(tt.item())
            del buf
            dt_nc = timed(lambda: R_.step(collective=False))
            step_ms, nc_ms = dt_ / a.steps * 1e3, dt_nc / a.steps * 1e3
            nbytes = R_.flat_grad.numel() * 4
            comm.update({"allreduce_ms": round(ar * 1e3, 4), "allreduce_reps": reps,
                         "allreduce_algbw_GBps": round(nbytes / ar / 1e9, 1),
                         "allreduce_busbw_GBps": round(nbytes / ar / 1e9 * 2 * (world - 1) / world, 1),
                         "step_ms": round(step_ms, 3), "step_ms_without_collective": round(nc_ms, 3),
                         "exposed_comm_frac": round(max(0.0, step_ms - nc_ms) / step_ms, 4),
                         "mode": "exact half-batch overlap" if R_.halves else "synchronous"})
        except Exception as e:   # noqa: BLE001
            comm["error"] = repr(e)[:300]
        if with_overlap and not a.overlap and mode_ in ("batch", "render_iter") and a.frames >= 2:
            try:
                R3 = FrameRenderer(sc, dev, frames, a.channels if mode_ == mode else 0, mode=mode_, dynamic=dynamic_,
                                   optimizer=not a.no_optimizer, halves=True)
                dt3 = timed(R3.step, R3.finish)
                R3.check_sorts()
                comm["overlap_exact"] = {"value": round(a.frames * a.steps * world / dt3, 2), "unit": "frames/s",
                                         "ms_per_step": round(dt3 / a.steps * 1e3, 3),
                                         "what": "same step with --overlap: two half-batches, the first half's all-reduce under "
                                                 "the second half's forward + backward, sums added, one Adam step (exact)"}
                del R3
            except Exception as e:   # noqa: BLE001
                comm["overlap_exact"] = {"error": repr(e)[:300]}
        if with_overlap and not a.zero1 and not a.overlap and not a.no_optimizer:
            try:
                R4 = FrameRenderer(sc, dev, frames, a.channels if mode_ == mode else 0, mode=mode_, dynamic=dynamic_, zero1=True)
                dt4 = timed(R4.step, R4.finish)
                R4.check_sorts()
                comm["zero1"] = {"value": round(a.frames * a.steps * world / dt4, 2), "unit": "frames/s",
                                 "ms_per_step": round(dt4 / a.steps * 1e3, 3),
                                 "adam_moments_MB_per_rank": round(2 * (R4.shards.own[1] - R4.shards.own[0]) * 4 / 1e6, 1),
                                 "what": "same step with --zero1: reduce-scatter of the flat gradient buffer, Adam on this rank's "
                                         "1 / N block (moments sharded), all-gather of the updated parameters (exact)"}
                del R4
            except Exception as e:   # noqa: BLE001
                comm["zero1"] = {"error": repr(e)[:300]}
        return comm

    comm = comm_analysis(R, dt, mode, a.dynamic)
    frames_total = a.frames * a.steps * world

    def faster_exact_schedule(dt_, comm_):
        """N > 1: the step was timed under BOTH exact schedules (synchronous: all-reduce -> Adam; half-batch overlap: the first
        half's all-reduce under the second half's frames -- same parameters to fp32 summation order).  The line reports the
        faster one and says which; the other stays in `comm`."""
        ov = (comm_ or {}).get("overlap_exact") or {}
        z1 = (comm_ or {}).get("zero1") or {}
        if a.overlap or a.zero1 or ("value" not in ov and "value" not in z1):
            return dt_, None
        sync_fps = frames_total / dt_
        comm_["synchronous"] = {"value": round(sync_fps, 2), "unit": "frames/s", "ms_per_step": round(dt_ / a.steps * 1e3, 3)}
        best, name = sync_fps, "synchronous"
        for cand, label in ((ov, "exact half-batch overlap"), (z1, "zero1: reduce-scatter -> sharded Adam -> all-gather")):
            if cand.get("value", 0) > best:
                best, name = cand["value"], label
        return (dt_ if name == "synchronous" else frames_total / best), name

    dt, schedule = faster_exact_schedule(dt, comm)
    fps = frames_total / dt
    M, T = R.last["M"], R.last["T"]
    # SURVEY 8d's per-pair figures are the reference's: algorithmic bytes / flops are priced with the pairs and the walked list
    # entries of ITS rule (bounding squares); the kernels walk the shorter lists of the reach masks (`tile_pairs_M`)
    ref_lists = None
    M_walked = M
    HW = a.width * a.height
    tag_cfg = f"{a.gaussians}x{a.width}x{a.height}x{a.channels}:{mode}" + ("" if a.no_spatial_order else ":morton")

    roofline = None
    kernels = {}
    fwd_ms = bwd_ms = opt_ms = None
    if rank == 0 and not a.no_kernel_timing:
        # same step again with per-kernel HIP events recorded on the launch stream
        torch.cuda.synchronize()
        L.profile_reset()
        L.profile_enable(True)
        R.step(collective=False)  # rank-0 only: must not enter a collective
        torch.cuda.synchronize()
        L.profile_enable(False)
        names = ["sh_fwd", "frame_preprocess_fwd", "frame_preprocess_bwd", "preprocess_fwd", "project_point_fwd", "cov3d_fwd", "ewa_fwd", "bin_count", "bin_colscan", "bin_tilescan",
                 "bin_scatter", "tile_sort", "blend_pack", "blend_fwd", "blend_bwd", "pair_reduce", "gauss_bwd", "preprocess_bwd", "ewa_bwd", "project_point_bwd",
                 "cov3d_bwd", "sh_bwd", "adam_step"]
        raw = {n: L.profile_read(n) for n in names}
        # (behind the event-timed step, not in front of it: the scratch forward synchronises with the host, and kernels that
        #  follow an idle device run at ramping clocks)
        ref_lists = R.reference_list_stats()
        if ref_lists:
            M = ref_lists["max_pairs_per_frame"]
        for n in names:
            ms, cnt = raw[n]
            if cnt:
                avg = ms / cnt
                fpl = a.frames / cnt       # frames one launch covers (1 on the per-frame paths, F in the batch)
                if mode in ("render_iter_frame", "ref_flow") and n in ("blend_bwd", "blend_pack", "pair_reduce"):
                    # per-frame renderer: one backward launch per set -> the per-launch figure is the mean over the sets
                    b = sum(kernel_bytes(n, a.gaussians, M, HW, c, T, False, a.frames * 3.0 / cnt) for c in (3, 1, 19)) / 3.0
                else:
                    b = kernel_bytes(n, a.gaussians, M, HW, R.C, T, R.use_sh, fpl,
                                     sets=mode.startswith("render_iter") and n.startswith(("blend", "gauss_bwd", "pair_reduce")),
                                     nparam=R.flat_grad.numel(), sh_acc=mode in ("batch", "frame", "frame_fused", "render_iter"), dyn=R.dynamic)
                kernels[n] = {"avg_us": round(avg * 1e3, 2), "launches": cnt, "frames_per_launch": round(fpl, 2),
                              "us_per_frame": round(ms * 1e3 / a.frames, 2), "alg_MB_per_launch": round(b / 1e6, 2),
                              "GBps": round(b / (avg * 1e-3) / 1e9, 1) if avg > 0 else None}
                if kernels[n]["GBps"] is not None and kernels[n]["GBps"] > HBM_ACHIEVABLE_GBS:
                    # priced bytes / time above what HBM sustains: part of the kernel's reads hit the memory-side cache / L2 (its
                    # producer ran right before it) -- the figure is a rate of ALGORITHMIC bytes, not HBM traffic
                    kernels[n]["note"] = "cache-resident: algorithmic bytes / time exceeds the achievable HBM rate"
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["us_per_frame"])
            ach = kernels[dom]["GBps"]
            traffic, tsrc = pmc_traffic(dom, tag_cfg)
            # HBM pricing of the dominant kernel (SURVEY 8d's algorithmic bytes); for a compositing kernel -- bound by FP32
            # issue, not by bandwidth -- the line's primary roofline becomes the FP32 pipe once the scene statistics are known
            # (below), and this pricing stays beside it as `hbm`
            roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                        "issue": pmc_issue(dom, tag_cfg),
                        "avg_us": kernels[dom]["avg_us"], "frames_per_launch": kernels[dom]["frames_per_launch"],
                        "alg_bytes_per_launch": int(kernels[dom]["alg_MB_per_launch"] * 1e6),
                        "pairs_priced": ("the reference's (tile, Gaussian) pairs per frame (bounding squares: SURVEY 8d's unit), "
                                         f"{M}; the kernel walks {M_walked} (reach masks)") if ref_lists else None}
            is_bwd = lambda k: k.endswith("_bwd") or k == "pair_reduce"
            fwd_ms = sum(kernels[k]["us_per_frame"] for k in kernels if not is_bwd(k) and k != "adam_step") / 1e3
            bwd_ms = sum(kernels[k]["us_per_frame"] for k in kernels if is_bwd(k)) / 1e3
            opt_ms = kernels.get("adam_step", {}).get("us_per_frame", 0.0) / 1e3
        L.profile_reset()

    # forward only (the north star's render target: >= 149 frames/s at 480p / 300k), timed the same way
    forward_only = None
    if mode in ("batch", "render_iter", "ref_flow"):
        dtf = timed(R.forward_only)
        R.check_sorts()
        forward_only = {"value": round(a.frames * a.steps * world / dtf, 2), "unit": "frames/s",
                        "ms_per_frame": round(dtf / (a.frames * a.steps) * 1e3, 4),
                        "what": ("forward render of the reference's LITERAL call sequence under no_grad (eager-torch projection + EWA, "
                                 "synchronising sort, three blends), frame by frame: the north star's >= 149 FPS check on the unchanged "
                                 "renderer flow" if mode == "ref_flow" else
                                 "forward pass alone (SH -> preprocess -> binning -> sort -> compositing), same frames, timed like `value`")}
    stats = R.scene_stats() if rank == 0 else None
    eager_only = None
    if mode == "ref_flow":
        dte = timed(R.eager_steps_only)
        eager_only = {"ms_per_frame": round(dte / (a.frames * a.steps) * 1e3, 4),
                      "what": "the flow's two eager-torch steps alone (orthographic projection + EWA, with gs.compute_cov3d between "
                              "them), forward + backward, timed like `value`: framework kernels, not this library's"}

    if stats is not None:
        stats["scene"] = a.scene
        stats["reference_lists"] = ref_lists
    if roofline is not None and stats is not None:
        comp = compute_rates(dict(stats, walked_entries_per_frame=(ref_lists or stats)["walked_entries_per_frame"]), kernels)
        roofline["compute"] = comp
        dom = roofline["kernel"]
        if dom in comp:
            # SURVEY 8d's contract: `frac` = algorithmic bytes / time / HBM peak (what rounds 1-3 reported).  The compositing
            # kernels are bound by FP32 instruction ISSUE, not by bandwidth: their f32 MFMAs run on the same FP32 lanes as the VALU
            # (157.3 TFLOP/s is both the dense f32 MFMA peak and the packed-FP32 vector peak) -- that secondary ceiling is
            # reported beside it under its own name: the reference's arithmetic over the list entries the pixels walk (a LOWER bound
            # of the useful flops: 16 per visited (pixel, entry) pair forward, 14 backward; reference-equivalent work, not hardware
            # utilisation -- the kernel culls entries the reference walks) per launch / the launch's duration.
            roofline.update({"bound": "hbm", "limited_by": "fp32-issue", "frac_hbm": roofline["frac"],
                             "frac_fp32_issue": comp[dom]["frac_of_fp32_peak"],
                             "fp32_issue": {"achieved": comp[dom]["TFLOPs_lower_bound"], "peak": comp["peak_TFLOPs"], "unit": "TFLOP/s",
                                            "frac": comp[dom]["frac_of_fp32_peak"],
                                            "alg_flops_per_launch": int(comp["evaluations_per_frame"] * (16.0 if dom == "blend_fwd" else 14.0)
                                                                        * roofline["frames_per_launch"]),
                                            "what": "FP32 pipe (f32 MFMA + VALU share it): walked (pixel, list entry) pairs x the "
                                                    "reference's flops per visit, per launch / launch duration, against the dense f32 "
                                                    "MFMA peak = packed-FP32 vector peak"}})

    # second workload of the line (N = 1, default configuration only): the reference's REAL training frame -- its dynamic
    # Gaussians (time-varying position and rotation) through render_iter's three blends (rgb enhanced K = 20 + depth + 19
    # attribute channels) -- forward + backward + Adam, timed exactly like `value`
    extra_lines = []
    if mode == "batch" and not a.dynamic and a.channels == 0 and not a.no_extra_lines and not a.stale_overlap and not a.overlap:
        # (every rank takes part at N > 1: the line carries its own bucket / all-reduce / exposed-communication figures)
        R2 = FrameRenderer(sc, dev, frames, 0, mode="render_iter", dynamic=True, optimizer=not a.no_optimizer)
        dt2 = timed(R2.step, R2.finish)
        R2.check_sorts()
        comm2 = comm_analysis(R2, dt2, "render_iter", True)
        dt2, schedule2 = faster_exact_schedule(dt2, comm2)
        dtf2 = timed(R2.forward_only)
        extra_lines.append({
            "metric": "rendered frames/sec fwd+bwd @480p, 300k Gaussians (the reference's training frame: dynamic Gaussians, "
                      "render_iter's three blends, 23 channels)",
            "value": round(a.frames * a.steps * world / dt2, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt2 / a.steps * 1e3, 3), "ms_per_frame": round(dt2 / (a.frames * a.steps) * 1e3, 4),
            "forward_only": {"value": round(a.frames * a.steps * world / dtf2, 2), "unit": "frames/s"},
            "grad_bucket_MB": round(R2.flat_grad.numel() * 4 / 1e6, 1),
            "allreduce_ms": None if not comm2 else comm2.get("allreduce_ms"),
            "exposed_comm_frac": None if not comm2 else comm2.get("exposed_comm_frac"), "comm": comm2,
            "config": {"workload": f"{a.gaussians} dynamic Gaussians of the reference's model (spline position, time-varying "
                                   f"rotation; dynamic_gaussian_with_base_point_cloud.py:171-250), {a.frames} frames/step, "
                                   f"{a.width}x{a.height}, rgb (SH deg 3, enhanced K=20, taps) + depth + 19 attribute channels "
                                   "= track_gs (position of a pair frame, per frame; src/trainer_fragGS.py:506-511) + 16 model "
                                   "attributes, fwd+bwd (incl. track_gs' gradient back into the pair frames' spline segments) + Adam",
                       "equivalent_flags": "--render-iter --dynamic", "schedule": schedule2,
                       "tile_pairs_M": R2.last.get("M")}})
        del R2
        # third workload: the reference's WHOLE training step composed from the native pieces (train_step.py)
        try:
            extra_lines.append(train_step_line(a, sc, dev, frames, timed, world, launched))
        except Exception as e:   # noqa: BLE001  (must not cost the line its headline; identical on every rank)
            extra_lines.append({"metric": "training step (train_step.py)", "error": repr(e)[:400]})

    # the other configurations of BASELINE.json / BASELINE.md section 9 inside the SAME (driver-run) default line, N = 1: each timed
    # exactly like `value` (W + K steps between synchronisations), compact -- their full lines are `profiles/r05_bench_*.json`
    other_configs = []
    if (world == 1 and mode == "batch" and not a.dynamic and a.channels == 0 and not a.no_extra_lines and not a.stale_overlap
            and not a.overlap and not a.no_other_configs):
        specs = [("c5 (configs[4]): 32 feature channels", "--channels 32", dict(channels=32), "batch"),
                 ("render_iter, static Gaussians: the renderer's three blends, 23 channels", "--render-iter", {}, "render_iter"),
                 ("render_iter with the trainer's ['mask_attribute', 'dino_attribute'] plan: 3 | 1 | 4 channels, same one-pass kernels",
                  "--render-iter --attr-channels 4", dict(attr_channels=4), "render_iter"),
                 ("fused per-frame operators (the drop-in path, frame by frame)", "--per-frame", {}, "frame"),
                 ("frame by frame through gs.rasterization_ortho: one call of the C ABI per direction", "--per-frame-fused", {}, "frame_fused"),
                 ("the reference's renderer file's LITERAL call sequence on this library", "--ref-flow", {}, "ref_flow"),
                 ("c4 (configs[3]): 1M Gaussians, 1280x720", "--gaussians 1000000 --width 1280 --height 720",
                  dict(gaussians=1000000, width=1280, height=720), "batch")]
        for what, flags, over, m2 in specs:
            try:
                g2, w2, h2, c2 = (over.get("gaussians", a.gaussians), over.get("width", a.width), over.get("height", a.height),
                                  over.get("channels", 0))
                sc2 = sc if (g2, w2, h2, c2) == (a.gaussians, a.width, a.height, 0) else build_scene(g2, w2, h2, c2)
                Ro = FrameRenderer(sc2, dev, frames, c2, mode=m2, optimizer=not a.no_optimizer, attr_channels=over.get("attr_channels", 19))
                dto = timed(Ro.step, Ro.finish)
                Ro.check_sorts()
                ent = {"what": what, "equivalent_flags": flags, "value": round(a.frames * a.steps / dto, 2), "unit": "frames/s",
                       "ms_per_frame": round(dto / (a.frames * a.steps) * 1e3, 4), "tile_pairs_M": Ro.last.get("M")}
                if m2 in ("batch", "render_iter", "ref_flow"):
                    dtfo = timed(Ro.forward_only)
                    ent["forward_only"] = round(a.frames * a.steps / dtfo, 2)
                other_configs.append(ent)
                del Ro, sc2
            except Exception as e:   # noqa: BLE001  (must not cost the line its headline)
                other_configs.append({"what": what, "equivalent_flags": flags, "error": repr(e)[:300]})
            torch.cuda.empty_cache()
        # what the UNCHANGED trainer pays per step for its neighbour search (src/trainer_fragGS.py:672 ->
        # src/geometry_utils.py:15: knn_points over ALL N Gaussians, K = 5 + self), through the shim's import name, timed like `value`
        try:
            sys.path.insert(1, os.path.join(ROOT, "shims"))
            from pytorch3d.ops import knn_points
            pts = R.p["xyz"].detach() + R.offs[0] if not R.dynamic else R.position
            with torch.no_grad():
                dtk = timed(lambda: knn_points(pts[None], pts[None], None, None, K=6))
            other_configs.append({"what": "pytorch3d.ops.knn_points(points[None], points[None], None, None, K=6) over all "
                                          f"{a.gaussians} Gaussians (shims/: the unchanged trainer's neighbour search, once per step)",
                                  "equivalent_flags": "(knn_full)", "value": round(dtk / a.steps * 1e3, 4), "unit": "ms",
                                  "higher_is_better": False})
        except Exception as e:   # noqa: BLE001
            other_configs.append({"what": "knn_points over all Gaussians", "equivalent_flags": "(knn_full)", "error": repr(e)[:300]})

    cpu = cpu_c = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline_torch(sc, a.channels)
        cpu_c = cpu_baseline_c(sc, a.channels)

    if rank == 0:
        par = f"frame-sharded dp{world}, " + ("stale-1: all-reduce overlapped with the next step, no optimiser" if R.overlap else
                                              "exact overlap: two half-batches, all-reduce of half 1 under half 2, sum -> Adam -> next forward"
                                              if (R.halves or schedule == "exact half-batch overlap") else
                                              "zero1: reduce-scatter of the flat gradient, Adam on 1 / N, all-gather of the parameters"
                                              if (R.zero1 or (schedule or "").startswith("zero1")) else
                                              "synchronous: all-reduce -> Adam -> next forward" if R.opt is not None else
                                              "synchronous all-reduce, no optimiser")
        line = {
            "metric": "rendered frames/sec fwd+bwd @480p, 300k Gaussians" + (f" (render_iter: three blends, {4 + getattr(R, 'A', 19)} channels)"
                                                                             if mode.startswith("render_iter") or mode == "ref_flow" else ""),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ranks_seen": ranks_seen, "build_id": L.build_id(),
            "allreduce_ms": None if not comm else comm.get("allreduce_ms"),
            "exposed_comm_frac": None if not comm else comm.get("exposed_comm_frac"),
            "comm": comm,
            "config": {"workload": (f"{a.gaussians} dynamic Gaussians of the reference's model (spline position, time-varying rotation)"
                                    if R.dynamic else
                                    f"{a.gaussians} Gaussians moving by per-frame position offsets (SURVEY 8d generator: static "
                                    f"scale / rotation / opacity / colour, xy += 0.05 sin(2 pi f / F + phase))")
                                   + (", clustered scene (70 % of the Gaussians in blobs covering 10 % of the image)" if a.scene == "clustered" else "")
                                   + f", {a.frames} frames/rank/step of a {clip}-frame "
                                   f"{a.width}x{a.height} clip, fwd+bwd, ortho camera, "
                                   + ("SH deg 3 -> RGB" if R.use_sh else f"{a.channels} feature channels")
                                   + (", Adam step on the flat parameter buffer" if R.opt is not None else ""),
                       "gaussians": a.gaussians, "width": a.width, "height": a.height, "frames_per_rank_per_step": a.frames,
                       "tile_pairs_M": M_walked, "tile_pairs_M_reference": (ref_lists or {}).get("max_pairs_per_frame"),
                       "channels": R.C, "parallelism": par,
                       "schedule": schedule and (schedule + " (the fastest of the exact schedules timed in this run; the others: "
                                                 "comm.synchronous / comm.overlap_exact / comm.zero1)"),
                       "gaussian_order": "random" if a.no_spatial_order else "morton (densify.spatial_order at setup)",
                       "frame_inputs": "the frames' position offsets / frame tables are built once per clip at set-up; the timed step "
                                       "launches this library's kernels only",
                       "path": ("the reference's real training frame as a frame batch: its dynamic Gaussians (per-frame evaluation "
                                "inside the batched preprocess) through render_iter's three blends (rgb enhanced K=20 + depth + 19 "
                                "attribute channels), one forward over the 23-channel row, one backward pass for the three sets"
                                if (R.dynamic and mode == "render_iter") else
                                "frame batch of the reference's dynamic Gaussians: their per-frame evaluation inside the batched "
                                "preprocess, the Gaussian-side backward walks all frames" if (R.dynamic and mode == "batch") else
                                "dynamic-Gaussian evaluation fused into the per-frame preprocess + gradient sinks" if R.dynamic else
                                "render_iter of the reference's renderer (rgb enhanced K=20 + depth + 19 attribute channels per "
                                "frame), native OrthoEnhancedRenderer, per-frame operators" if mode == "render_iter_frame" else
                                "the reference's LITERAL render_iter call sequence (dptr_ortho_enhanced.py:270-376): eager-torch "
                                "projection + EWA, compute_cov3d, synchronising sort_gaussian, three separate blends through "
                                "autograd, frame by frame" if mode == "ref_flow" else
                                "render_iter of the reference's renderer (rgb enhanced K=20 + depth + 19 attribute channels per frame) "
                                "as a frame batch: one forward over the 23-channel row, one backward pass for the three feature sets"
                                if mode == "render_iter" else
                                "frame batch: the frame is a grid dimension of every kernel; SH and the Gaussian-side backward once per step"
                                if mode == "batch" else
                                "fused per-frame operators + gradient sinks" if mode == "frame" else "per-operator autograd chain"),
                       "grad_bucket_MB": round(R.flat_grad.numel() * 4 / 1e6, 1),
                       "batch_buffers_MB": round(R.batch.memory_bytes() / 1e6, 1) if mode in ("batch", "render_iter") else None},
            "ms_per_frame": round(dt / (a.frames * a.steps) * 1e3, 4),
            "gpu_kernel_ms_per_frame": {"forward": None if fwd_ms is None else round(fwd_ms, 4),
                                        "backward": None if bwd_ms is None else round(bwd_ms, 4),
                                        "optimizer": None if opt_ms is None else round(opt_ms, 4)},
            "forward_only": forward_only, "eager_steps_only": eager_only, "scene_stats": stats,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_c_oracle": cpu_c, "kernels": kernels,
            "extra_lines": extra_lines, "other_configs": other_configs,
        }
        # compact summary of the reference's real workloads, the LAST key of the line (the driver keeps the line's last 2000
        # characters): every figure is timed in THIS run like `value`
        oc = {x.get("equivalent_flags"): x for x in other_configs if "error" not in x}
        el = [x for x in extra_lines if "error" not in x]
        pick = lambda d, k: None if d is None else d.get(k)
        tf = next((x for x in el if "training frame" in x.get("metric", "")), None)
        ts = next((x for x in el if "train_step_ms" in x), None)
        line["summary"] = {"headline_fps": line["value"], "training_frame_fps": pick(tf, "value"), "train_step_ms": pick(ts, "train_step_ms"),
                           "c4_fps": pick(oc.get("--gaussians 1000000 --width 1280 --height 720"), "value"),
                           "c5_fps": pick(oc.get("--channels 32"), "value"),
                           "render_iter_fps": pick(oc.get("--render-iter"), "value"),
                           "render_iter_attr4_fps": pick(oc.get("--render-iter --attr-channels 4"), "value"),
                           "per_frame_fps": pick(oc.get("--per-frame"), "value"),
                           "per_frame_fused_fps": pick(oc.get("--per-frame-fused"), "value"),
                           "ref_flow_fwd_fps": pick(oc.get("--ref-flow"), "forward_only"),
                           "knn_full_ms": pick(oc.get("(knn_full)"), "value"),
                           "fwd_only_fps": pick(forward_only, "value"), "n_gpus": world, "ranks_seen": ranks_seen,
                           "build_id": L.build_id()}
        print(json.dumps(line))
    if launched:
        dist.barrier()  # rank 0 did extra (untimed) measurement work: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
