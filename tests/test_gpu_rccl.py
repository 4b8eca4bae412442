"""RCCL first contact on the one GPU of the box: a process group `backend="nccl"` (= RCCL on ROCm) of ONE
rank on cuda:0 executes every collective signature the 8-GPU run uses -- the flat bucket's all-reduce (synchronous, and
`async_op=True` + wait), the exact half-batch overlap, `reduce_scatter_tensor` / `all_gather_into_tensor` on bucket slices with
the asynchronous all-reduce of the replicated part beside them (`parallel.MIN_WORLD = 1` lifts the one-rank early-outs), the
MAX / SUM all-reduces of the densification statistics, and one composed training step under each optimiser schedule.  In a
world of one rank every collective is the identity: results must equal the no-process-group run bit for bit.  Reference wiring:
src/train.py:19-31,210-213 (init_process_group("nccl"), one process per GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_loaded():
    with open("/proc/self/maps") as f:
        return any("librccl" in line or "libnccl" in line for line in f)


def _train_steps(mode, steps=2):
    """`steps` composed training steps on a small clip; returns the flat parameters"""
    from splatter_a_video_amd import train_step as TS
    from test_gpu_train_step import _clip, _perturbed, _t
    Nn, Ww, Hh, T, F = 2500, 128, 96, 20, 3
    sc, clock, truth = _clip(Nn, Ww, Hh, T, seed=5)
    extr = _t(sc.extr)
    lr = dict(TS.REFERENCE_LR, pos_cubic_node=2e-3, shs=2e-2, attrs=2e-2, scaling=1e-2, rotation=5e-3)
    st = TS.TrainingStep(_perturbed(truth, 1), clock, Ww, Hh, F, extr, lr=lr, K=8, arap_samples=128, sample_seed=3,
                         owner_sharded=(mode in ("owner", "exchange")), zero1=(mode == "zero1"),
                         exchange_positions=(mode == "exchange"))
    t1, t2 = [0, 7, 13], [4, 2, 19]
    gt = TS.render_ground_truth(truth, clock, Ww, Hh, extr, t1, t2)
    for _ in range(steps):
        st.step(t1, t2, gt)
    torch.cuda.synchronize()
    n = sum(v.numel() for v in st.p.values())
    return st.bucket.flat_param.detach()[:n].clone(), st.dstate.pos_gradient_accum.clone()


def _worker(rank, port, out):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import splatter_a_video_amd.parallel as P
    from splatter_a_video_amd.optim import FlatAdam, OwnerShardedAdam
    res = {}
    # the references: the same work WITHOUT a process group
    ref = {m: _train_steps(m) for m in ("dense", "owner", "zero1", "exchange")}
    for m in ("owner", "zero1", "exchange"):      # (two runs of one step differ by ~1e-9 in a few spline coefficients: float atomics of the ARAP scatter)
        torch.testing.assert_close(ref[m][0], ref["dense"][0], rtol=0, atol=1e-6)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        res["backend"] = dist.get_backend()
        # ---- 1. the flat bucket's all-reduce, synchronous and asynchronous
        g = torch.Generator(device="cpu").manual_seed(0)
        tensors = {"cubic": torch.randn(8, 500, 4, 3, generator=g).to(dev), "xyz": torch.randn(500, 3, generator=g).to(dev),
                   "shs": torch.randn(500, 16, 3, generator=g).to(dev)}
        b = P.FlatGradBucket(tensors, buffers=2, pad_to=4)
        b.flat_grad.copy_(torch.randn(b.flat_grad.shape, generator=g))
        want = b.flat_grad.clone()
        b.all_reduce()
        b.all_reduce(async_op=True)
        assert b.pending[b.active] is not None
        b.wait()
        assert torch.equal(b.flat_grad, want)
        ones = torch.ones(3, device=dev)
        dist.all_reduce(ones)
        res["ranks_seen"] = int(ones[0].item())
        # ---- 2. the exact half-batch overlap: async all-reduce of half 1 under half 2's work, sync all-reduce, fold, Adam
        opt = FlatAdam(b, 1e-3)
        p0 = b.flat_param.detach().clone()
        P.overlapped_halves_step(b, lambda: b.grad("xyz").add_(1.0), lambda: b.grad("shs").add_(2.0), opt)
        torch.cuda.synchronize()
        moved = (b.flat_param.detach() - p0).abs()
        a_xyz, a_shs, a_cub = (b.slices[k] for k in ("xyz", "shs", "cubic"))
        assert moved[a_xyz[0]:a_xyz[1]].min() > 0 and moved[a_shs[0]:a_shs[1]].min() > 0 and moved[a_cub[0]:a_cub[1]].max() == 0
        # ---- 3. reduce_scatter_tensor / all_gather_into_tensor on bucket slices, the async all-reduce of the rest beside them
        P.MIN_WORLD = 1
        for shards in (P.OwnerShards(b, "cubic", 1, 0), P.Zero1Shards(b, 1, 0)):
            b.flat_grad.copy_(want)
            P.owner_reduce(b, shards)
            assert torch.equal(b.flat_grad, want)
            before = b.flat_param.detach().clone()
            P.owner_gather(b, shards)
            assert torch.equal(b.flat_param.detach(), before)
            oa = OwnerShardedAdam(b, shards, 1e-3)
            P.owner_sharded_step(b, shards, [0], lambda f: b.grad("cubic").add_(0.5), oa)
        # ---- 4. the densification statistics' collectives (SUM of float taps, MAX of the visibility bytes / int radii)
        vg, vis, rad = torch.rand(100, 2, device=dev), (torch.rand(100, device=dev) > 0.5).to(torch.uint8), torch.randint(0, 30, (100,), dtype=torch.int32, device=dev)
        keep = (vg.clone(), vis.clone(), rad.clone())
        P.reduce_densify_batch(vg, vis, rad)
        assert torch.equal(vg, keep[0]) and torch.equal(vis, keep[1]) and torch.equal(rad, keep[2])
        # ---- 4b. the position exchange's point-to-point batch (grouped isend / irecv; a world of one rank sends to itself) and its
        #          small all-gather of the requested frame times
        fa, fb_ = torch.randn(700, 3, device=dev), torch.empty(700, 3, device=dev)
        P.exchange_frames([(0, fa)], [(0, fb_)])
        torch.cuda.synchronize()
        assert torch.equal(fa, fb_)
        mine = torch.tensor([3.0, 7.0], dtype=torch.float64, device=dev)
        outl = [torch.empty_like(mine)]
        dist.all_gather(outl, mine)
        assert torch.equal(outl[0], mine)
        # ---- 5. one composed training step per schedule over RCCL (all-reduce | owner-sharded | ZeRO-1 | position exchange)
        for m in ("dense", "owner", "zero1", "exchange"):
            got = _train_steps(m)
            torch.testing.assert_close(got[0], ref[m][0], rtol=0, atol=1e-6)
            torch.testing.assert_close(got[1], ref[m][1], rtol=1e-5, atol=1e-9)
        res["rccl_loaded"] = _rccl_loaded()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    torch.save(res, out)


@pytest.mark.timeout(900)
def test_every_collective_of_the_multi_gpu_run_executes_on_rccl(tmp_path):
    out = str(tmp_path / "rccl")
    mp.spawn(_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = torch.load(out)
    assert res["backend"] == "nccl" and res["ranks_seen"] == 1 and res["rccl_loaded"]
