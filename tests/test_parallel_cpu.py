"""World-size-2 gloo tests of the frame-sharded DP plumbing (splatter_a_video_amd/parallel.py):
the all-reduced flat bucket of two ranks rendering disjoint frame shards equals the gradient of a
single process rendering all frames.  A differentiable CPU stand-in renderer is enough here -- the
collective logic does not depend on the kernels (those are covered by the -m gpu parity tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from splatter_a_video_amd.parallel import FlatGradBucket, frames_of_rank, reduce_visibility, sharded_step


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
