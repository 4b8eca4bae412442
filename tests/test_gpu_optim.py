"""optim.FlatAdam (one HIP launch over the flat parameter buffer) against torch.optim.Adam -- the optimiser the reference
steps (src/pointrix/optimizer/optimizer.py:70-83; eps = 1e-15, one learning rate per parameter group)."""
import numpy as np
import pytest
import torch

from splatter_a_video_amd.optim import FlatAdam
from splatter_a_video_amd.parallel import FlatGradBucket

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sizes", [[(7, 3), (7, 1), (7, 16, 3)], [(100003, 3), (100003, 4), (100003, 1), (100003, 16, 3)]])
def test_flat_adam_equals_torch_adam(sizes):
    g = torch.Generator(device="cpu").manual_seed(0)
    names = [f"p{i}" for i in range(len(sizes))]
    lrs = {n: 10.0 ** -(2 + i % 3) for i, n in enumerate(names)}
    init = {n: torch.randn(*s, generator=g) for n, s in zip(names, sizes)}
    bucket = FlatGradBucket({n: t.cuda() for n, t in init.items()})
    opt = FlatAdam(bucket, lrs, eps=1e-15)
    ref_p = [init[n].clone().cuda().requires_grad_(True) for n in names]
    ref = torch.optim.Adam([{"params": [p], "lr": lrs[n]} for p, n in zip(ref_p, names)], lr=0.0, eps=1e-15)
    for step in range(5):
        grads = {n: torch.randn(*s, generator=g) * (0.1 + step) for n, s in zip(names, sizes)}
        if step == 3:
            grads[names[0]].zero_()          # zero gradient: m decays, v decays, p still moves
        for n, p in zip(names, ref_p):
            p.grad = grads[n].cuda()
            bucket.grad(n).copy_(grads[n].cuda())
        ref.step()
        opt.step()
        for n, p in zip(names, ref_p):
            torch.testing.assert_close(bucket.params[n].detach(), p.detach(), rtol=2e-5, atol=1e-7)


def test_flat_adam_grad_scale_and_validation():
    bucket = FlatGradBucket({"a": torch.ones(1001, device="cuda"), "b": torch.ones(13, 2, device="cuda")})
    twin = FlatGradBucket({"a": torch.ones(1001, device="cuda"), "b": torch.ones(13, 2, device="cuda")})
    o1, o2 = FlatAdam(bucket, 1e-2), FlatAdam(twin, 1e-2)
    bucket.flat_grad.fill_(4.0)
    twin.flat_grad.fill_(1.0)
    o1.step(grad_scale=0.25)
    o2.step()
    assert torch.equal(bucket.flat_param, twin.flat_param)
    with pytest.raises(ValueError):
        o1.step(grad=torch.zeros(3, device="cuda"))
    with pytest.raises(ValueError):
        FlatAdam(FlatGradBucket({"a": torch.ones(4)}), 1e-3)


def test_flat_adam_follows_a_learning_rate_schedule():
    """set_lr before every step = the reference's scheduler on the position group (src/pointrix/optimizer/scheduler.py):
    moments and step count survive, groups whose rates diverge split, groups that meet again merge"""
    g = torch.Generator(device="cpu").manual_seed(1)
    sizes = {"position": (5001, 3), "scaling": (5001, 3), "opacity": (5001, 1)}
    init = {n: torch.randn(*s, generator=g) for n, s in sizes.items()}
    bucket = FlatGradBucket({n: t.cuda() for n, t in init.items()})
    opt = FlatAdam(bucket, 1e-3, eps=1e-15)                 # one segment to start with
    assert opt.nseg == 1
    ref_p = {n: init[n].clone().cuda().requires_grad_(True) for n in sizes}
    ref = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": n} for n, p in ref_p.items()], lr=0.0, eps=1e-15)
    for step in range(6):
        lr_pos = 1e-3 * (0.5 ** step) if step < 4 else 1e-3   # decays, then meets the other groups again
        opt.set_lr({"position": lr_pos})
        for grp in ref.param_groups:
            if grp["name"] == "position":
                grp["lr"] = lr_pos
        assert opt.nseg == (1 if lr_pos == 1e-3 else 2)
        for n, p in ref_p.items():
            gr = torch.randn(*sizes[n], generator=g)
            p.grad = gr.cuda()
            bucket.grad(n).copy_(gr.cuda())
        ref.step()
        opt.step()
        for n, p in ref_p.items():
            torch.testing.assert_close(bucket.params[n].detach(), p.detach(), rtol=2e-5, atol=1e-7)
    with pytest.raises(KeyError):
        opt.set_lr({"nope": 1.0})


def test_owner_sharded_adam_steps_its_block_and_the_replicated_slice_like_flat_adam():
    """optim.OwnerShardedAdam (parallel.owner_sharded_step): rank 1 of 3 steps ITS block of the owned table and the whole
    replicated slice with the update rule of FlatAdam -- bit-identical there -- and leaves the other ranks' blocks alone"""
    from splatter_a_video_amd.optim import FlatAdam, OwnerShardedAdam
    from splatter_a_video_amd.parallel import FlatGradBucket, OwnerShards
    g = torch.Generator(device="cuda").manual_seed(3)
    I, N = 7, 1000
    mk = lambda: {"cubic": 0.1 * torch.randn(I, N, 4, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)),
                  "rotation": torch.randn(N, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)),
                  "opacity": torch.randn(N, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)),
                  "shs": torch.randn(N, 16, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))}
    from splatter_a_video_amd.optim import PatternLR
    lr = {"cubic": 1e-3, "rotation": 2e-3, "opacity": 5e-2, "shs": PatternLR(1.25e-4, head_lr=2.5e-3, period=48, head=3)}
    a, b = FlatGradBucket(mk()), FlatGradBucket(mk())
    sh = OwnerShards(b, "cubic", 3, 1)
    lo, hi = sh.own
    assert (lo, hi) == (2 * N * 12, 4 * N * 12)                    # segments 2, 3 of 7 (7 * r // 3)
    oa, ob = FlatAdam(a, lr), OwnerShardedAdam(b, sh, lr)
    assert ob.m_own.numel() == hi - lo and ob.m_rep.numel() == 53 * N
    before = b.flat_param.detach().clone()
    for _ in range(3):
        gr = torch.randn(a.flat_grad.numel(), device="cuda", generator=g)
        a.flat_grad.copy_(gr); b.flat_grad.copy_(gr)
        oa.step(grad_scale=0.5); ob.step(grad_scale=0.5)
    pa, pb = a.flat_param.detach(), b.flat_param.detach()
    assert torch.equal(pb[lo:hi], pa[lo:hi]) and torch.equal(pb[sh.b:], pa[sh.b:])
    assert torch.equal(pb[:lo], before[:lo]) and torch.equal(pb[hi:sh.b], before[hi:sh.b])
    assert not torch.equal(pb[lo:hi], before[lo:hi])
    # the moments as whole buffers (structure changes): the own block and the replicated slice of FlatAdam's, zeros elsewhere
    m, v = ob.full_moments()
    fm, fv = oa.full_moments()
    assert torch.equal(m[lo:hi], fm[lo:hi]) and torch.equal(v[sh.b:], fv[sh.b:]) and float(m[:lo].abs().max()) == 0.0
    oc = OwnerShardedAdam(b, OwnerShards(b, "cubic", 3, 2), lr)
    oc.load_moments(fm, fv)
    lo2, hi2 = oc.shards.own
    assert torch.equal(oc.m_own, fm[lo2:hi2]) and torch.equal(oc.v_rep, fv[sh.b:])
    ob.zero_moments("opacity")
    a0, b0 = b.slices["opacity"]
    m2, _ = ob.full_moments()
    assert float(m2[a0:b0].abs().max()) == 0.0 and torch.equal(m2[b0:], fm[b0:]) and torch.equal(m2[lo:hi], fm[lo:hi])
    ob.set_lr({"rotation": 1e-2})
    with pytest.raises(KeyError):
        ob.set_lr({"nope": 1.0})


def test_pattern_learning_rates_are_two_parameter_groups_inside_one_tensor():
    """optim.PatternLR: the SH block [N,16,3] as ONE tensor of the flat buffer whose DC triplets step with the rate of the
    reference's `features` group and the other 45 floats with `features_rest`'s (src/configs/frag_gs_v10.yaml:44-47) -- against
    torch.optim.Adam on the two tensors the reference keeps"""
    from splatter_a_video_amd.optim import FlatAdam, PatternLR
    from splatter_a_video_amd.parallel import FlatGradBucket
    g = torch.Generator(device="cpu").manual_seed(5)
    N = 3001
    shs0, op0 = torch.randn(N, 16, 3, generator=g), torch.randn(N, 1, generator=g)
    bucket = FlatGradBucket({"opacity": op0.cuda(), "shs": shs0.cuda(), "rotation": torch.randn(N, 4, generator=g).cuda()})
    opt = FlatAdam(bucket, {"opacity": 5e-2, "shs": PatternLR(1.25e-4, head_lr=2.5e-3, period=48, head=3), "rotation": 1e-3}, eps=1e-15)
    assert opt.has_pattern and opt.nseg == 3
    dc = shs0[:, :1].clone().cuda().requires_grad_(True)
    rest = shs0[:, 1:].clone().cuda().requires_grad_(True)
    op = op0.clone().cuda().requires_grad_(True)
    ref = torch.optim.Adam([{"params": [dc], "lr": 2.5e-3}, {"params": [rest], "lr": 1.25e-4}, {"params": [op], "lr": 5e-2}], eps=1e-15)
    for _ in range(4):
        gs_, go = torch.randn(N, 16, 3, generator=g).cuda(), torch.randn(N, 1, generator=g).cuda()
        dc.grad, rest.grad, op.grad = gs_[:, :1].clone(), gs_[:, 1:].clone(), go
        bucket.flat_grad.zero_()
        bucket.grad("shs").copy_(gs_); bucket.grad("opacity").copy_(go)
        ref.step()
        opt.step()
    got = bucket.params["shs"].detach()
    torch.testing.assert_close(got[:, :1], dc.detach(), rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(got[:, 1:], rest.detach(), rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(bucket.params["opacity"].detach(), op.detach(), rtol=2e-5, atol=1e-7)
