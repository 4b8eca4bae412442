"""World-size-2 gloo tests of the frame-sharded DP plumbing (splatter_a_video_amd/parallel.py):
the all-reduced flat bucket of two ranks rendering disjoint frame shards equals the gradient of a
single process rendering all frames.  A differentiable CPU stand-in renderer is enough here -- the
collective logic does not depend on the kernels (those are covered by the -m gpu parity tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from splatter_a_video_amd.parallel import (FlatGradBucket, frames_of_rank, overlapped_halves_step, reduce_densify_batch,
                                            reduce_visibility, sharded_step)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"xyz": torch.randn(50, 3, generator=g), "opacity": torch.rand(50, 1, generator=g),
            "shs": torch.randn(50, 16, 3, generator=g)}


def _toy_render(p, f):
    """any differentiable function of (params, frame) -- stands in for one frame's render + loss"""
    phase = 0.1 * f
    img = (torch.sin(p["xyz"] + phase).sum(1, keepdim=True) * p["opacity"]).sum() + (p["shs"][:, 0] * (f + 1)).sum()
    return img


def _worker(rank, world, port, frames, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = FlatGradBucket(_params())
        mine = frames_of_rank(frames, rank, world)
        sharded_step(b, mine, lambda f: _toy_render(b.params, f).backward())
        vis = torch.zeros(50, dtype=torch.bool); vis[rank::2] = True
        rad = torch.full((50,), rank + 1, dtype=torch.int32)
        v, r = reduce_visibility(vis, rad)
        if rank == 0:
            torch.save({"flat": b.flat_grad.clone(), "vis": v, "rad": r, "mine": mine}, out)
    finally:
        dist.destroy_process_group()


def test_frames_of_rank_partition():
    frames = list(range(7))
    parts = [frames_of_rank(frames, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == frames
    assert parts[0] == [0, 3, 6]


def test_flat_bucket_accumulates_in_place():
    b = FlatGradBucket(_params())
    ptr = b.flat_grad.data_ptr()
    for f in range(3):
        _toy_render(b.params, f).backward()
    assert b.flat_grad.data_ptr() == ptr
    for n, p in b.params.items():
        assert p.grad.data_ptr() == b.grad(n).data_ptr()          # still views of the bucket
    ref = _params()
    for t in ref.values():
        t.requires_grad_(True)
    sum(_toy_render(ref, f) for f in range(3)).backward()
    for n in ref:
        torch.testing.assert_close(b.grad(n), ref[n].grad)
    b.all_reduce()                                                  # single process: no-op


@pytest.mark.timeout(120)
def test_two_rank_allreduce_matches_single_process(tmp_path):
    frames = list(range(6))
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), frames, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["mine"] == [0, 2, 4]
    ref = _params()
    for t in ref.values():
        t.requires_grad_(True)
    sum(_toy_render(ref, f) for f in frames).backward()
    flat_ref = torch.cat([ref[n].grad.reshape(-1) for n in ref])
    torch.testing.assert_close(got["flat"], flat_ref, rtol=1e-5, atol=1e-5)
    assert bool(got["vis"].all()) and bool((got["rad"] == 2).all())


def _worker_double(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = FlatGradBucket(_params(), buffers=2)
        results = []
        for step in range(4):                       # the collective of step s runs while step s+1 fills the other buffer
            b.swap()
            b.zero_grad()
            for f in frames_of_rank(list(range(10 * step, 10 * step + 6)), rank, world):
                _toy_render(b.params, f).backward()
            b.all_reduce(async_op=True)
            results.append(b.active)
        b.wait()
        if rank == 0:
            torch.save({"g2": b.flat_grads[results[2]].clone(), "g3": b.flat_grads[results[3]].clone(), "order": results}, out)
    finally:
        dist.destroy_process_group()


def test_double_buffered_async_allreduce(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_double, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["order"] == [1, 0, 1, 0]
    for step, key in ((2, "g2"), (3, "g3")):
        ref = FlatGradBucket(_params())
        for f in range(10 * step, 10 * step + 6):
            _toy_render(ref.params, f).backward()
        assert torch.allclose(got[key], ref.flat_grad, rtol=1e-5, atol=1e-6)


def test_single_buffer_swap_is_a_noop():
    b = FlatGradBucket(_params())
    g = b.params["xyz"].grad
    b.swap()
    assert b.active == 0 and b.params["xyz"].grad.data_ptr() == g.data_ptr()
    b2 = FlatGradBucket(_params(), buffers=2)
    p0 = b2.params["shs"].grad.data_ptr()
    b2.swap()
    assert b2.active == 1 and b2.params["shs"].grad.data_ptr() != p0 and b2.grad("shs").data_ptr() == b2.params["shs"].grad.data_ptr()
    _toy_render(b2.params, 1).backward()
    assert b2.flat_grads[1].abs().sum() > 0 and b2.flat_grads[0].abs().sum() == 0


# ------------------------------------------------------------------ synchronous step: all-reduce -> optimiser -> next forward
class _TorchFlatAdam:
    """torch restatement of optim.FlatAdam's rule (the product's Adam is a HIP kernel, tests/test_gpu_optim.py checks it
    against torch.optim.Adam); here only the ORDER of a synchronous data-parallel step is under test"""

    def __init__(self, bucket, lr):
        self.b, self.lr, self.t = bucket, lr, 0
        self.m = torch.zeros_like(bucket.flat_param); self.v = torch.zeros_like(bucket.flat_param)

    def step(self, grad_scale=1.0):
        self.t += 1
        g = self.b.flat_grad * grad_scale
        self.m.mul_(0.9).add_(g, alpha=0.1)
        self.v.mul_(0.999).addcmul_(g, g, value=0.001)
        with torch.no_grad():
            self.b.flat_param -= (self.lr / (1 - 0.9 ** self.t)) * self.m / (self.v.sqrt() / (1 - 0.999 ** self.t) ** 0.5 + 1e-15)


def _worker_sync(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = FlatGradBucket(_params())
        opt = _TorchFlatAdam(b, 1e-2)
        for step in range(3):      # every step's forward sees the parameters the previous step's optimiser produced
            frames = frames_of_rank(list(range(6 * step, 6 * step + 6)), rank, world)
            sharded_step(b, frames, lambda f: _toy_render(b.params, f).backward(), optimizer=opt)
        torch.save({"param": b.flat_param.detach().clone()}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_synchronous_steps_match_single_process(tmp_path):
    out = str(tmp_path / "p")
    mp.spawn(_worker_sync, args=(2, _free_port(), out), nprocs=2, join=True)
    p0, p1 = torch.load(out + ".0")["param"], torch.load(out + ".1")["param"]
    assert torch.equal(p0, p1)                       # replicas stay bit-identical
    ref = FlatGradBucket(_params())
    opt = _TorchFlatAdam(ref, 1e-2)
    for step in range(3):
        sharded_step(ref, list(range(6 * step, 6 * step + 6)), lambda f: _toy_render(ref.params, f).backward(), optimizer=opt)
    torch.testing.assert_close(p0, ref.flat_param.detach(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ exact overlap: two half-batches, all-reduce of half 1 under half 2
def _worker_halves(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = FlatGradBucket(_params(), buffers=2)
        opt = _TorchFlatAdam(b, 1e-2)
        grads = []
        for step in range(3):
            mine = frames_of_rank(list(range(6 * step, 6 * step + 6)), rank, world)      # 3 frames: halves of 2 + 1
            render = lambda fs: [_toy_render(b.params, f).backward() for f in fs]
            overlapped_halves_step(b, lambda: render(mine[:2]), lambda: render(mine[2:]), optimizer=opt)
            assert b.pending == [None, None]            # nothing outstanding when the optimiser has stepped
            grads.append(b.flat_grad.clone())
        torch.save({"param": b.flat_param.detach().clone(), "grads": grads}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_overlapped_halves_equal_the_synchronous_step(tmp_path):
    """the exact overlap mode -- half 1's gradients are all-reduced (async, second flat buffer) under half
    2's forward + backward, the sums are added, Adam steps once -- leaves the parameters of the synchronous step (to fp32
    summation order: 2e-6 of the maximum), on both ranks bit-identically, over several steps"""
    out = str(tmp_path / "h")
    mp.spawn(_worker_halves, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["param"], r1["param"])
    ref = FlatGradBucket(_params())
    opt = _TorchFlatAdam(ref, 1e-2)
    for step in range(3):
        sharded_step(ref, list(range(6 * step, 6 * step + 6)), lambda f: _toy_render(ref.params, f).backward(), optimizer=opt)
        g = ref.flat_grad
        assert float((r0["grads"][step] - g).abs().max()) <= 2e-6 * float(g.abs().max())
    p = ref.flat_param.detach()
    assert float((r0["param"] - p).abs().max()) <= 2e-6 * float(p.abs().max())


def test_overlapped_halves_need_two_buffers_and_work_without_a_process_group():
    b1 = FlatGradBucket(_params())
    with pytest.raises(ValueError, match="buffers=2"):
        overlapped_halves_step(b1, lambda: None, lambda: None)
    b = FlatGradBucket(_params(), buffers=2)
    overlapped_halves_step(b, lambda: _toy_render(b.params, 0).backward(), lambda: _toy_render(b.params, 1).backward())
    ref = FlatGradBucket(_params())
    sharded_step(ref, [0, 1], lambda f: _toy_render(ref.params, f).backward())
    torch.testing.assert_close(b.flat_grad, ref.flat_grad)
    assert b.active == 1


# ------------------------------------------------------------------ densification decisions are rank-deterministic
def _densify_inputs(N=400, F=6, seed=3):
    rng = np.random.default_rng(seed)
    radius = (rng.integers(0, 30, size=(F, N)) * (rng.random((F, N)) < 0.6)).astype(np.int32)
    taps = (rng.normal(size=(F, N, 2)) * 2e-4 * (radius > 0)[..., None]).astype(np.float32)
    scaling_raw = np.log(rng.uniform(0.002, 0.05, size=(N, 3))).astype(np.float32)
    opacity_raw = rng.normal(0, 2, size=(N, 1)).astype(np.float32)
    return radius, taps, scaling_raw, opacity_raw


def _masks_after(frames, radius, taps, scaling_raw, opacity_raw, reduce):
    import oracle
    N = radius.shape[1]
    vg = np.zeros((N, 2), np.float32); vis = np.zeros(N, np.uint8); radii = np.zeros(N, np.int32)
    for f in frames:
        oracle.densify_accumulate(radius[f], taps[f], 1.0, 1.0, vg, vis, radii)
    if reduce:
        tv, tvis, tr = torch.from_numpy(vg), torch.from_numpy(vis), torch.from_numpy(radii)
        reduce_densify_batch(tv, tvis, tr)         # in place, shares memory with the numpy arrays
    max_r = np.zeros(N, np.float32); accum = np.zeros(N, np.float32); denom = np.zeros(N, np.float32)
    oracle.densify_update(vis, vg, radii, max_r, accum, denom)
    return oracle.densify_masks(accum, denom, max_r, scaling_raw, opacity_raw, 2e-4, 0.03, 1.0, 0.05, 20.0)


def _worker_densify(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        radius, taps, sr, orr = _densify_inputs()
        frames = frames_of_rank(list(range(radius.shape[0])), rank, world)
        masks = _masks_after(frames, radius, taps, sr, orr, reduce=True)
        np.savez(out + f".{rank}.npz", clone=masks[0], split=masks[1], prune=masks[2])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_take_identical_densification_decisions(tmp_path):
    """every rank sees only its frames' gradient taps / radii; after reduce_densify_batch the clone / split / prune masks
    are identical on both ranks and equal to a single process that rendered the whole batch"""
    out = str(tmp_path / "m")
    mp.spawn(_worker_densify, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    radius, taps, sr, orr = _densify_inputs()
    ref = _masks_after(range(radius.shape[0]), radius, taps, sr, orr, reduce=False)
    unreduced = _masks_after(frames_of_rank(list(range(radius.shape[0])), 0, 2), radius, taps, sr, orr, reduce=False)
    for k, r in zip(("clone", "split", "prune"), ref):
        np.testing.assert_array_equal(a[k], b[k])
        # float sums over 3 + 3 frames vs 6 frames in a row may differ in the last bit: a threshold comparison can flip
        assert (a[k] != r).sum() <= 1
    assert ref[0].sum() > 5 and ref[1].sum() > 0
    assert any((u != r).sum() > 5 for u, r in zip(unreduced, ref))     # without the reduction the ranks would diverge


def test_lr_segments_merge_and_split():
    """host logic of FlatAdam's per-segment learning rates (optim.lr_segments)"""
    from splatter_a_video_amd.optim import MAX_SEGMENTS, lr_segments
    sl = {"a": (0, 10), "b": (10, 25), "c": (25, 26), "d": (26, 100)}
    assert lr_segments(sl, dict(a=1e-3, b=1e-3, c=1e-3, d=1e-3)) == ([100], [1e-3])
    assert lr_segments(sl, dict(a=1e-3, b=2e-3, c=2e-3, d=1e-3)) == ([10, 26, 100], [1e-3, 2e-3, 1e-3])
    many = {f"p{i}": (i, i + 1) for i in range(MAX_SEGMENTS + 1)}
    with pytest.raises(ValueError):
        lr_segments(many, {k: float(i) for i, k in enumerate(many)})


def test_split_needs_a_round_dependent_seed():
    """densify.split_children / densify_split refuse to draw without a seed (the Philox counter is (Gaussian, replica) only:
    a constant seed would repeat the same normals at every densification round)"""
    import torch
    from splatter_a_video_amd import densify as D
    z = torch.zeros(4, 3)
    with pytest.raises(ValueError, match="seed"):
        D.split_children(z, z, torch.zeros(4, 4), torch.ones(4, dtype=torch.bool))
    with pytest.raises(ValueError, match="seed"):
        D.densify_split({"position": z, "scaling": z, "rotation": torch.zeros(4, 4)}, None, torch.ones(4, dtype=torch.bool))


# ------------------------------------------------------------------ owner-sharded step: reduce-scatter + sharded Adam + all-gather
class _TorchOwnerAdam:
    """torch restatement of optim.OwnerShardedAdam: the replicated slice everywhere, of the owned parameter only this rank's
    block -- moments for nothing else"""

    def __init__(self, bucket, shards, lr):
        self.b, self.s, self.lr, self.t = bucket, shards, lr, 0
        lo, hi = shards.own
        self.parts = [(lo, hi), (shards.b, shards.total)]
        self.m = [torch.zeros(h - l) for l, h in self.parts]
        self.v = [torch.zeros(h - l) for l, h in self.parts]

    def step(self, grad_scale=1.0):
        self.t += 1
        with torch.no_grad():
            for (l, h), m, v in zip(self.parts, self.m, self.v):
                g = self.b.flat_grad[l:h] * grad_scale
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                self.b.flat_param[l:h] -= (self.lr / (1 - 0.9 ** self.t)) * m / (v.sqrt() / (1 - 0.999 ** self.t) ** 0.5 + 1e-15)


def _seg_params(seed=0, I=5, N=30):
    g = torch.Generator().manual_seed(seed)
    # the owned parameter FIRST: a spline table stored segment-major [I, N, 4, 3]
    return {"cubic": 0.1 * torch.randn(I, N, 4, 3, generator=g), "xyz": torch.randn(N, 3, generator=g), "opacity": torch.rand(N, 1, generator=g)}


def _seg_render(p, f, I=5):
    """a frame reads ITS segment of the table (f // 5 ... here f % I) and, like track_gs, the segment of a pair frame"""
    s1, s2 = f % I, (3 * f + 1) % I
    pos = p["xyz"] + p["cubic"][s1, :, 3] + 0.3 * p["cubic"][s1, :, 2]
    pair = p["xyz"] + p["cubic"][s2, :, 3]
    return (torch.sin(pos).sum(1, keepdim=True) * p["opacity"]).sum() + 0.1 * (pair * pos).sum()


def _worker_owner(rank, world, port, out):
    from splatter_a_video_amd.parallel import OwnerShards, owner_sharded_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = FlatGradBucket(_seg_params())
        sh = OwnerShards(b, "cubic", world, rank)
        opt = _TorchOwnerAdam(b, sh, 1e-2)
        assert sh.bounds[0] == 0 and sh.bounds[-1] == b.slices["cubic"][1] and not sh.equal      # 5 segments on 2 ranks: 2 + 3
        unit_of_frame = [f % 5 for f in range(10)]
        mine_blocks = sh.frames_of_rank(unit_of_frame)
        for step in range(3):
            frames = frames_of_rank(list(range(6 * step, 6 * step + 6)), rank, world)
            owner_sharded_step(b, sh, frames, lambda f: _seg_render(b.params, f).backward(), opt)
        torch.save({"param": b.flat_param.detach().clone(), "blocks": mine_blocks, "own": sh.own}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_owner_sharded_step_equals_the_synchronous_step(tmp_path):
    """the spline table's gradient reduced to its owners only, Adam moments kept by the owner alone,
    updated blocks gathered -- the parameters of sharded_step (dense all-reduce + replicated Adam) to fp32 summation order,
    replicas bit-identical, over several steps with frames that touch OTHER ranks' segments (the pair frames)"""
    out = str(tmp_path / "o")
    mp.spawn(_worker_owner, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["param"], r1["param"])
    assert r0["own"][1] == r1["own"][0] and sorted(r0["blocks"] + r1["blocks"]) == list(range(10))
    assert r0["blocks"] == [0, 1, 5, 6] and r1["blocks"] == [2, 3, 4, 7, 8, 9]          # time blocks: segments 0-1 | 2-4
    ref = FlatGradBucket(_seg_params())
    opt = _TorchFlatAdam(ref, 1e-2)
    for step in range(3):
        sharded_step(ref, list(range(6 * step, 6 * step + 6)), lambda f: _seg_render(ref.params, f).backward(), optimizer=opt)
    p = ref.flat_param.detach()
    assert float((r0["param"] - p).abs().max()) <= 2e-6 * float(p.abs().max())


def test_owner_shards_need_the_owned_parameter_first():
    from splatter_a_video_amd.parallel import OwnerShards, owner_sharded_step
    b = FlatGradBucket(_params())
    with pytest.raises(ValueError, match="first tensor"):
        OwnerShards(b, "shs", 2, 0)
    # without a process group the step is the plain local step
    b2 = FlatGradBucket(_seg_params())
    sh = OwnerShards(b2, "cubic", 1, 0)
    opt = _TorchOwnerAdam(b2, sh, 1e-2)
    owner_sharded_step(b2, sh, [0, 1], lambda f: _seg_render(b2.params, f).backward(), opt)
    ref = FlatGradBucket(_seg_params())
    o2 = _TorchFlatAdam(ref, 1e-2)
    sharded_step(ref, [0, 1], lambda f: _seg_render(ref.params, f).backward(), optimizer=o2)
    torch.testing.assert_close(b2.flat_param.detach(), ref.flat_param.detach())


# ------------------------------------------------------------------ ZeRO-1 over the whole flat buffer
class _TorchZeroAdam:
    """torch restatement of optim.OwnerShardedAdam on parallel.Zero1Shards: moments for this rank's block only"""

    def __init__(self, bucket, shards, lr):
        self.b, self.lr, self.t = bucket, lr, 0
        self.lo, self.hi = shards.own
        self.m, self.v = torch.zeros(self.hi - self.lo), torch.zeros(self.hi - self.lo)

    def step(self, grad_scale=1.0):
        self.t += 1
        with torch.no_grad():
            g = self.b.flat_grad[self.lo:self.hi] * grad_scale
            self.m.mul_(0.9).add_(g, alpha=0.1)
            self.v.mul_(0.999).addcmul_(g, g, value=0.001)
            self.b.flat_param[self.lo:self.hi] -= (self.lr / (1 - 0.9 ** self.t)) * self.m / (self.v.sqrt() / (1 - 0.999 ** self.t) ** 0.5 + 1e-15)


def _zero_params():
    p = _seg_params(N=31)
    p["extra"] = torch.tensor([0.3, -0.2, 0.1])          # 31 * 64 + 3 floats: NOT two blocks of whole float4
    return p


def _zero_render(p, f):
    return _seg_render(p, f) + (p["extra"] * (f + 1.0)).sum()


def _worker_zero1(rank, world, port, out):
    from splatter_a_video_amd.parallel import Zero1Shards, zero1_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with pytest.raises(ValueError, match="pad_to"):
            Zero1Shards(FlatGradBucket(_zero_params()), world, rank)
        b = FlatGradBucket(_zero_params(), pad_to=4 * world)
        sh = Zero1Shards(b, world, rank)
        assert sh.equal and sh.bounds[-1] == b.flat_param.numel() and sh.own[0] % 4 == 0 and b.flat_param.numel() % (4 * world) == 0
        opt = _TorchZeroAdam(b, sh, 1e-2)
        for step in range(3):
            frames = frames_of_rank(list(range(6 * step, 6 * step + 6)), rank, world)
            zero1_step(b, sh, frames, lambda f: _zero_render(b.params, f).backward(), opt)
        torch.save({"param": b.flat_param.detach().clone(), "own": sh.own}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_zero1_step_equals_the_synchronous_step(tmp_path):
    """the whole flat buffer in `world` equal blocks: gradient reduced to the block's owner, Adam on the own block only, updated
    blocks gathered -- sharded_step's parameters to fp32 summation order, replicas bit-identical, padding tail untouched"""
    out = str(tmp_path / "z")
    mp.spawn(_worker_zero1, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["param"], r1["param"]) and r0["own"][1] == r1["own"][0]
    ref = FlatGradBucket(_zero_params())
    opt = _TorchFlatAdam(ref, 1e-2)
    for step in range(3):
        sharded_step(ref, list(range(6 * step, 6 * step + 6)), lambda f: _zero_render(ref.params, f).backward(), optimizer=opt)
    p = ref.flat_param.detach()
    n = p.numel()
    assert float((r0["param"][:n] - p).abs().max()) <= 2e-6 * float(p.abs().max())
    assert r0["param"].numel() > n and float(r0["param"][n:].abs().max()) == 0.0


def test_a_block_that_starts_inside_a_pattern_group_keeps_the_pattern():
    """host logic of OwnerShardedAdam._segments: ZeRO-1 cuts the buffer anywhere, e.g. inside the SH block whose DC triplets step
    with their own rate (PatternLR period 48, head 3).  The per-element rate of every block's segments must be the flat one."""
    from splatter_a_video_amd.optim import OwnerShardedAdam, PatternLR

    def rate_of(ends, rates, pats, i):
        start = 0
        for e, r, (per, head, hr) in zip(ends, rates, pats):
            if i < e:
                return hr if (per and (i - start) % per < head) else r
            start = e
        raise AssertionError

    class Stub:            # what _segments reads
        pass
    slices = {"opacity": (0, 37), "shs": (37, 37 + 48 * 11), "rotation": (37 + 48 * 11, 37 + 48 * 11 + 44)}
    lr = {"opacity": 5e-2, "shs": PatternLR(1.25e-4, head_lr=2.5e-3, period=48, head=3), "rotation": 1e-3}
    total = slices["rotation"][1]
    flat = [5e-2] * 37 + [2.5e-3 if k % 48 < 3 else 1.25e-4 for k in range(48 * 11)] + [1e-3] * 44
    for world in (1, 2, 3, 5, 8):
        per = -(-total // (4 * world)) * 4
        for r in range(world):
            lo, hi = r * per, min((r + 1) * per, total)
            o = Stub()
            o.bucket, o.lr = Stub(), lr
            o.bucket.slices = slices
            ends, rates, pats = OwnerShardedAdam._segments(o, lo, hi)
            assert ends[-1] == hi - lo
            got = [rate_of(ends, rates, pats, i) for i in range(hi - lo)]
            assert got == flat[lo:hi], (world, r)
    # a cut one float into the head (phase 1: two head floats left) and one past it (phase 3: none left)
    for lo in (37 + 48 * 2 + 1, 37 + 48 * 2 + 3, 37 + 48 * 2 + 47):
        o = Stub(); o.bucket, o.lr = Stub(), lr; o.bucket.slices = slices
        ends, rates, pats = OwnerShardedAdam._segments(o, lo, total)
        assert [rate_of(ends, rates, pats, i) for i in range(total - lo)] == flat[lo:]
