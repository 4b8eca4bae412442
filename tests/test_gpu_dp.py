"""Two data-parallel ranks on ONE device (gloo; RCCL needs one GPU per rank) through the bench's synchronous step --
local frames forward+backward on the frame batch, all-reduce of the flat bucket, Adam on the flat parameters -- against a
single process that renders all frames: same reduced gradient, same parameters after the step, and identical replicas."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, W, H, FRAMES = 6000, 128, 96, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(frames, steps=2, halves=False):
    import bench
    from splatter_a_video_amd.synth import make_scene
    sc = make_scene(N, W, H, F=12, seed=77)
    R = bench.FrameRenderer(sc, torch.device("cuda:0"), frames, mode="batch", halves=halves)
    assert R.halves == halves
    R.opt.set_lr(1e-3)              # a visible update: step 2's forward must see step 1's parameters
    grads = []
    for _ in range(steps):
        R.step()
        grads.append(R.flat_grad.clone())
    torch.cuda.synchronize()
    R.check_sorts()
    return grads, R.bucket.flat_param.detach().clone()


def _same_gradient(g2, g1, step):
    """two ranks vs one process: the first step's reduced gradient agrees element-wise (summation order only).  From the second
    step on the PARAMETERS already differ where Adam normalised a rounding-noise gradient to a full +-lr step with either sign (see
    the parameter check below), so a handful of Gaussians render a visibly different gradient: all but 1e-4 of the elements within
    the tolerance, none further off than 5x."""
    tol = 2e-4 * g1.abs() + 2e-6 * float(g1.abs().max())
    ratio = (g2 - g1).abs() / tol
    if step == 0:
        assert float(ratio.max()) <= 1.0, f"step {step}: {float(ratio.max())}"
    else:
        assert float((ratio > 1.0).float().mean()) <= 1e-4 and float(ratio.max()) <= 5.0, f"step {step}: {float(ratio.max())}"


def _worker(rank, world, port, out, halves=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grads, param = _run([f for f in range(FRAMES) if f % world == rank], halves=halves)
        torch.save({"grads": [g.cpu() for g in grads], "param": param.cpu()}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_device_match_single_process(tmp_path):
    out = str(tmp_path / "dp")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["param"], r1["param"])                    # replicas stay bit-identical
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    grads, param = _run(list(range(FRAMES)))
    for s, (g2, g1) in enumerate(zip(r0["grads"], grads)):
        _same_gradient(g2, g1.cpu(), s)
    # Adam normalises every element's step to ~lr whatever the gradient's size: where the gradient is rounding noise its
    # SIGN may differ between the two summation orders, so single elements differ by up to 2 lr per step -- but almost
    # all parameters agree closely
    d = (r0["param"] - param.cpu()).abs()
    assert float(d.max()) <= 2.1 * 1e-3 * 2
    assert float(d.median()) < 1e-6 and float((d > 1e-5).float().mean()) < 0.02
    assert float(g1.abs().max()) > 0


@pytest.mark.timeout(600)
def test_two_ranks_exact_overlap_equals_the_synchronous_step(tmp_path):
    """--overlap (parallel.overlapped_halves_step through the bench's step): each rank's 3 frames as half-batches of 2 + 1,
    half 1's all-reduce (async, second buffer) under half 2's forward + backward -- same reduced gradient and parameters as
    the single-process synchronous step, replicas bit-identical"""
    out = str(tmp_path / "ov")
    mp.spawn(_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["param"], r1["param"])
    grads, param = _run(list(range(FRAMES)))
    for s, (g2, g1) in enumerate(zip(r0["grads"], grads)):
        _same_gradient(g2, g1.cpu(), s)
    d = (r0["param"] - param.cpu()).abs()
    assert float(d.max()) <= 2.1 * 1e-3 * 2          # (Adam: sign of rounding-noise gradients, see above)
    assert float(d.median()) < 1e-6 and float((d > 1e-5).float().mean()) < 0.02


@pytest.mark.timeout(900)
def test_bench_gpus_2_launches_and_reports_two_ranks():
    """`python bench.py --gpus 2` with no torchrun environment starts two ranks itself (gloo: both share this box's GPU;
    over RCCL the same code path needs one GPU per rank) and its line says n_gpus 2, ranks_seen 2; N = 1 stays a plain run."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, SPLAT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    small = ["--gaussians", "6000", "--width", "128", "--height", "96", "--frames", "3", "--steps", "2", "--warmup", "1",
             "--no-cpu-baseline", "--no-kernel-timing", "--no-extra-lines"]
    lines = {}
    for n in (2, 1):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + small, env=env, capture_output=True,
                           text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[n] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert lines[2]["n_gpus"] == 2 and lines[2]["ranks_seen"] == 2
    assert lines[1]["n_gpus"] == 1 and lines[1]["ranks_seen"] == 1
    # the collective's cost is in the line: the bucket's all-reduce alone, the step without it, the
    # exposed fraction, and the exact half-batch overlap variant measured beside the synchronous step
    c = lines[2]["comm"]
    assert "error" not in c, c
    assert lines[2]["allreduce_ms"] == c["allreduce_ms"] > 0 and c["step_ms_without_collective"] > 0
    assert 0.0 <= lines[2]["exposed_comm_frac"] <= 1.0
    assert c["overlap_exact"]["value"] > 0, c
    assert lines[1]["comm"] is None and lines[1]["allreduce_ms"] is None
    assert len(lines[1]["build_id"]) == 16
    assert lines[2]["config"]["frames_per_rank_per_step"] == 3 and lines[2]["value"] > 0


# ---------------------------------------------------------------------------------------------------------------------
# the composed training step (train_step.py) under data parallelism
def _train_worker(rank, world, port, out, owner=False, steps=45, blocks=False, exchange=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from splatter_a_video_amd import train_step as TS
        from test_gpu_train_step import _clip, _perturbed, _t
        Nn, Ww, Hh, T, F = 3000, 128, 96, 20, 3
        sc, clock, truth = _clip(Nn, Ww, Hh, T, seed=5)
        extr = _t(sc.extr)
        cfg = TS.DensifyConfig(interval=15, start_iter=10, stop_iter=40, grad_threshold=5e-4, cameras_extent=60.0, min_opacity=0.02, seed=9)
        lr = dict(TS.REFERENCE_LR, pos_cubic_node=2e-3, shs=2e-2, attrs=2e-2, scaling=1e-2, rotation=5e-3)
        st = TS.TrainingStep(_perturbed(truth, 1), clock, Ww, Hh, F, extr, lr=lr, densify=cfg, K=8, arap_samples=128, sample_seed=rank,
                             owner_sharded=owner, exchange_positions=exchange)
        rng = np.random.default_rng(100 + rank)                     # every rank draws ITS OWN frame pairs
        counts, losses = [st.N], []
        # blocks: ids1 from the rank's own time block (what OwnerShards.frames_of_rank deals; the position exchange needs it)
        mine = [f for f in range(T) if int(clock.scalars(f)[0]) * world // clock.interval_num == rank] if blocks else list(range(T))
        for _ in range(steps):
            t1 = [int(t) for t in rng.choice(mine, F, replace=False)]
            t2 = [int((t + 1 + rng.integers(T - 1)) % T) for t in t1]
            st.step(t1, t2, TS.render_ground_truth(truth, clock, Ww, Hh, extr, t1, t2))
            losses.append(st.loss())
            if st.maybe_densify():
                counts.append(st.N)
        st.sync_table()         # (position exchange: the other owners' blocks of the table before anything reads all of it)
        torch.cuda.synchronize()
        torch.save({"param": st.bucket.flat_param.detach().cpu(), "m": st.opt.full_moments()[0].cpu(), "counts": counts, "losses": losses,
                    "slices": dict(st.bucket.slices), "frozen": st.frozen["position"].cpu()}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_training_step_replicas_stay_bit_identical_through_densification(tmp_path):
    """the composed training step on two ranks with DIFFERENT frame pairs: one all-reduce of the flat bucket per step, Adam with
    1 / world, densification statistics reduced over the ranks (SUM of the taps, MAX of radii / visibility) -- the replicas take
    the same clone / split / prune decisions (children from the counter-based generator), rebuild at the same count and end
    bit-identical: parameters, Adam moments, positions"""
    out = str(tmp_path / "ts")
    mp.spawn(_train_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["counts"] == r1["counts"] and len(set(r0["counts"])) >= 3, (r0["counts"], r1["counts"])
    assert torch.equal(r0["param"], r1["param"]) and torch.equal(r0["m"], r1["m"]) and torch.equal(r0["frozen"], r1["frozen"])
    assert np.isfinite(r0["losses"]).all() and np.mean(r0["losses"][-5:]) < np.mean(r0["losses"][:3])


@pytest.mark.timeout(900)
def test_two_ranks_owner_sharded_training_step_equals_the_all_reduce_step(tmp_path):
    """TrainingStep(owner_sharded=True): the spline table's gradient reduced to the owners of its time blocks only, each owner
    stepping its block with its own shard of the Adam moments, blocks gathered -- through a structure change (the moments are
    gathered, moved with their Gaussians and dealt out again at the new count) the replicas stay bit-identical, and parameters
    and assembled moments are the all-reduce step's.  (Two runs of the SAME step differ in a handful of spline coefficients by
    ~1e-9: the ARAP gradient is scattered with float atomics, as the reference's index_add is -- hence a tolerance between the
    runs, none between the ranks of a run.)"""
    out_o, out_d = str(tmp_path / "own"), str(tmp_path / "dense")
    mp.spawn(_train_worker, args=(2, _free_port(), out_o, True, 18), nprocs=2, join=True)
    mp.spawn(_train_worker, args=(2, _free_port(), out_d, False, 18), nprocs=2, join=True)
    o0, o1, d0 = torch.load(out_o + ".0"), torch.load(out_o + ".1"), torch.load(out_d + ".0")
    assert o0["counts"] == o1["counts"] == d0["counts"] and len(set(o0["counts"])) == 2
    assert torch.equal(o0["param"], o1["param"]) and torch.equal(o0["m"], o1["m"])
    torch.testing.assert_close(o0["param"], d0["param"], rtol=0, atol=1e-6)
    torch.testing.assert_close(o0["m"], d0["m"], rtol=0, atol=1e-6)


@pytest.mark.timeout(900)
def test_two_ranks_position_exchange_equals_the_all_reduce_step(tmp_path):
    """TrainingStep(owner_sharded=True, exchange_positions=True): position(ids2) of a pair frame in the other rank's time block is
    evaluated by its owner and sent, its gradient sent back and taken through the positions' backward THERE -- the spline table is
    neither reduced nor gathered in a step.  Three steps: every parameter and the assembled moments equal the all-reduce step's
    on the same pairs (1e-6: the owner sums the position gradients in one launch, the all-reduce the ranks' partial tables -- another
    summation order).  Eighteen steps through a structure change (the table is gathered for it once): the two ranks agree bit for
    bit, the Gaussian counts and the loss curve are the all-reduce run's (the parameters themselves part where Adam normalises a
    rounding-noise gradient to a full +-lr step of either sign, as between any two summation orders)."""
    out_x, out_d = str(tmp_path / "xch"), str(tmp_path / "dense")
    mp.spawn(_train_worker, args=(2, _free_port(), out_x, True, 3, True, True), nprocs=2, join=True)
    mp.spawn(_train_worker, args=(2, _free_port(), out_d, False, 3, True, False), nprocs=2, join=True)
    x0, x1, d0 = torch.load(out_x + ".0"), torch.load(out_x + ".1"), torch.load(out_d + ".0")
    assert torch.equal(x0["param"], x1["param"]) and torch.equal(x0["m"], x1["m"])
    torch.testing.assert_close(x0["param"], d0["param"], rtol=0, atol=1e-6)
    torch.testing.assert_close(x0["m"], d0["m"], rtol=0, atol=1e-6)
    mp.spawn(_train_worker, args=(2, _free_port(), out_x, True, 18, True, True), nprocs=2, join=True)
    mp.spawn(_train_worker, args=(2, _free_port(), out_d, False, 18, True, False), nprocs=2, join=True)
    x0, x1, d0 = torch.load(out_x + ".0"), torch.load(out_x + ".1"), torch.load(out_d + ".0")
    assert x0["counts"] == x1["counts"] == d0["counts"] and len(set(x0["counts"])) == 2
    assert torch.equal(x0["param"], x1["param"]) and torch.equal(x0["m"], x1["m"])
    np.testing.assert_allclose(x0["losses"], d0["losses"], rtol=2e-2)
    assert float((x0["param"] - d0["param"]).abs().mean()) < 2e-3


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n", [8, 4])
def test_bench_gpus_8_and_4_rehearsal_over_gloo(n):
    """what the driver's first multi-GPU run executes, rehearsed on one device: `python bench.py --gpus 8` (and 4) starts its ranks,
    the clip grows to 25 frames per rank (200 at 8 ranks: BASELINE configs[2]), the line says n_gpus / ranks_seen = N, carries
    the collective's figures and the exact-overlap variant, and the extra lines -- the reference's training frame and the
    composed training step -- run at N > 1 with their own bucket / all-reduce fields"""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, SPLAT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    small = ["--gaussians", "4000", "--width", "128", "--height", "96", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             "--no-kernel-timing"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + small, env=env, capture_output=True,
                       text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and line["ranks_seen"] == n
    assert line["config"]["frames_per_rank_per_step"] == 25 and f"of a {25 * n}-frame" in line["config"]["workload"]
    c = line["comm"]
    assert "error" not in c and c["allreduce_ms"] > 0 and c["overlap_exact"]["value"] > 0, c
    # both exact schedules were timed; the line's value is the faster one and says which
    best = max(c["synchronous"]["value"], c["overlap_exact"]["value"], (c.get("zero1") or {}).get("value", 0.0))
    assert c["synchronous"]["value"] > 0 and abs(line["value"] - best) < 0.02
    assert line["config"]["schedule"].startswith(("synchronous", "exact half-batch overlap", "zero1"))
    frame, step = line["extra_lines"]
    assert "error" not in frame and frame["n_gpus"] == n and frame["value"] > 0
    assert frame["comm"]["allreduce_ms"] > 0 and frame["grad_bucket_MB"] == frame["comm"]["bucket_MB"]
    assert "error" not in step, step
    assert step["n_gpus"] == n and step["train_step_ms"] > 0 and set(step["phases_ms"]) >= {"render_forward", "render_backward", "knn_arap"}
    # the same step under ZeRO-1 and with the position exchange (contiguous time blocks per rank) beside it
    assert "error" not in step["zero1"] and step["zero1"]["train_step_ms"] > 0, step["zero1"]
    assert "error" not in step["position_exchange"] and step["position_exchange"]["train_step_ms"] > 0, step["position_exchange"]


@pytest.mark.timeout(900)
def test_bench_train_step_owner_sharded_two_ranks_over_gloo():
    """`bench.py --gpus 2 --train-step --owner-sharded`: the composed step with the spline table's gradient reduced to its owners,
    through the launcher; the line says which optimiser ran and holds about half of the table's moments per rank"""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, SPLAT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    small = ["--gaussians", "4000", "--width", "128", "--height", "96", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             "--no-kernel-timing", "--train-step"]
    lines = {}
    for key, flag in ((False, []), (True, ["--owner-sharded"]), ("x", ["--exchange-positions", "--frames", "5"])):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + small + flag, env=env,
                           capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[key] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    dense, own, xch = lines[False], lines[True], lines["x"]
    # the position exchange through the launcher (two ranks, contiguous time blocks of the clip, point-to-point over gloo's host path)
    assert xch["n_gpus"] == 2 and xch["train_step_ms"] > 0 and "POSITION EXCHANGE" in xch["config"]["optimizer"]
    assert np.isfinite(xch["loss"])
    assert own["n_gpus"] == 2 and own["train_step_ms"] > 0 and own["config"]["optimizer"].startswith("owner-sharded")
    assert own["config"]["grad_bucket_MB"] == dense["config"]["grad_bucket_MB"]
    assert own["config"]["adam_moments_MB_per_rank"] < 0.85 * dense["config"]["adam_moments_MB_per_rank"]
    assert abs(own["loss"] - dense["loss"]) < 1e-4 * max(1.0, abs(dense["loss"]))


# ---------------------------------------------------------------------------------------------------------------------
# the densification statistic must not depend on the number of ranks
def _stat_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from splatter_a_video_amd import train_step as TS
        from test_gpu_train_step import _clip, _perturbed, _t
        Nn, Ww, Hh, T = 3000, 128, 96, 20
        sc, clock, truth = _clip(Nn, Ww, Hh, T, seed=5)
        extr = _t(sc.extr)
        t1_all, t2_all = [0, 3, 7, 11, 14, 18], [5, 9, 1, 16, 2, 12]
        t1, t2 = t1_all[rank::world], t2_all[rank::world]
        st = TS.TrainingStep(_perturbed(truth, 1), clock, Ww, Hh, len(t1), extr, K=8, arap_samples=128)
        st.step(t1, t2, TS.render_ground_truth(truth, clock, Ww, Hh, extr, t1, t2))
        torch.cuda.synchronize()
        p = {k: v.detach() for k, v in st.p.items()}
        masks = st.dstate.masks(p["scaling"], p["opacity"], 2e-5, 1e-3, 60.0, 0.02, 20.0)
        torch.save({"accum": st.dstate.pos_gradient_accum.cpu(), "denom": st.dstate.denom.cpu(), "radii": st.dstate.max_radii2D.cpu(),
                    "masks": [m.cpu() for m in masks]}, out + f".{world}.{rank}")
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_densification_statistic_is_world_size_invariant(tmp_path):
    """one rank with 2F frames and two ranks with F frames each accumulate the same `pos_gradient_accum` (the taps of a rank are
    those of its loss share = the mean over ITS frames; reduced over the ranks they are scaled by 1 / world, as the optimiser
    scales the gradient) and take the same clone / split decisions against the reference's per-frame threshold"""
    out = str(tmp_path / "st")
    mp.spawn(_stat_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    mp.spawn(_stat_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    one, two0, two1 = torch.load(out + ".1.0"), torch.load(out + ".2.0"), torch.load(out + ".2.1")
    assert torch.equal(two0["accum"], two1["accum"]) and torch.equal(two0["denom"], one["denom"]) and torch.equal(two0["radii"], one["radii"])
    a, b = one["accum"], two0["accum"]
    assert float(a.max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(a.max())              # fp32 summation order only -- not a factor `world`
    for m1, m2 in zip(one["masks"], two0["masks"]):
        assert int((m1 != m2).sum()) <= 2                                     # (a threshold tie may flip)
    assert int(one["masks"][0].sum()) + int(one["masks"][1].sum()) > 10       # the thresholds do select something
