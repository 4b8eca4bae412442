"""Fused per-frame operators against the operator chain they replace (same HIP arithmetic, separate launches) and
against the C oracle."""
import numpy as np
import pytest
import torch

import dptr.gs as gs
import oracle
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _chain(xyz, off, scale, rotate, extr, W, H):
    pos = xyz + off if off is not None else xyz
    uv, depth = gs.project_point_ortho(pos, extr, W, H, nearest=0.01)
    vis = depth != 0
    cov = gs.compute_cov3d(scale, rotate, vis)
    conic, radius, tiles = gs.ewa_project_ortho(pos, cov, extr, uv, W, H, vis)
    return uv, depth, conic, radius, tiles


@pytest.mark.parametrize("N,W,H,with_offset", [(1, 16, 16, False), (5000, 160, 96, True), (120_001, 854, 480, True)])
def test_preprocess_ortho_equals_operator_chain(N, W, H, with_offset):
    sc = make_scene(N, W, H, C=3, seed=N)
    rng = np.random.default_rng(N)
    xyz = sc.positions(0).copy()
    xyz[: max(1, N // 50)] *= 40.0                                   # some points far outside: culled
    xyz[max(1, N // 50): max(2, N // 25), 2] = -5.0                   # some behind the near plane
    off = (0.02 * rng.normal(size=(N, 3))).astype(np.float32) if with_offset else None
    g_uv, g_d, g_c = (rng.normal(size=s).astype(np.float32) for s in ((N, 2), (N, 1), (N, 3)))
    outs, grads = [], []
    for fused in (False, True):
        p = dict(xyz=_t(xyz, True), scale=_t(sc.scale, True), rotate=_t(sc.rotate, True))
        o = _t(off, True) if off is not None else None
        extr = _t(sc.extr)
        if fused:
            r = gs.preprocess_ortho(p["xyz"], p["scale"], p["rotate"], extr, W, H, nearest=0.01, offset=o)
        else:
            r = _chain(p["xyz"], o, p["scale"], p["rotate"], extr, W, H)
        torch.autograd.backward([r[0], r[1], r[2]], [_t(g_uv), _t(g_d), _t(g_c)])
        outs.append([x.detach().cpu().numpy() for x in r])
        grads.append([p["xyz"].grad.cpu().numpy(), p["scale"].grad.cpu().numpy(), p["rotate"].grad.cpu().numpy()]
                     + ([o.grad.cpu().numpy()] if o is not None else []))
    (uv0, d0, c0, r0, t0), (uv1, d1, c1, r1, t1) = outs
    assert (d0 == 0).any() or N == 1
    np.testing.assert_array_equal(r0, r1)
    np.testing.assert_array_equal(t0, t1)
    np.testing.assert_allclose(uv1, uv0, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(d1, d0, rtol=1e-6, atol=1e-6)
    # off-diagonal conic entries cancel to ~0: absolute tolerance relative to the conic's scale (FMA contraction differs
    # between the fused and the separate kernels)
    np.testing.assert_allclose(c1, c0, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(c0).max())))
    for a, b in zip(grads[1], grads[0]):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b).max())))


def test_preprocess_ortho_matches_oracle():
    N, W, H = 3000, 128, 80
    sc = make_scene(N, W, H, C=3, seed=9)
    xyz = sc.positions(3)
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(_t(xyz), _t(sc.scale), _t(sc.rotate), _t(sc.extr), W, H, nearest=0.01)
    o_uv, o_d = oracle.project_point_ortho_forward(xyz, sc.extr, W, H, 0.01, 1.3)
    vis = (o_d != 0).reshape(-1)
    o_cov = oracle.compute_cov3d_forward(sc.scale, sc.rotate, vis)
    o_conic, o_r, o_t = oracle.ewa_project_forward(xyz, o_cov, None, sc.extr, o_uv, W, H, vis, ortho=True)
    np.testing.assert_allclose(uv.cpu().numpy(), o_uv, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), o_d.reshape(-1, 1), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(radius.cpu().numpy(), o_r)
    np.testing.assert_array_equal(tiles.cpu().numpy(), o_t)
    np.testing.assert_allclose(conic.cpu().numpy(), o_conic, rtol=5e-5, atol=2e-6 * max(1.0, float(np.abs(o_conic).max())))


def test_preprocess_ortho_grad_sink_accumulates():
    N, W, H = 20_000, 320, 200
    sc = make_scene(N, W, H, C=3, seed=4)
    rng = np.random.default_rng(0)
    extr = _t(sc.extr)
    frames = [0, 3]
    gs_ = [[_t(rng.normal(size=s).astype(np.float32)) for s in ((N, 2), (N, 1), (N, 3))] for _ in frames]
    ref = dict(xyz=_t(sc.positions(0), True), scale=_t(sc.scale, True), rotate=_t(sc.rotate, True))
    snk = {k: _t(v.detach().cpu().numpy(), True) for k, v in ref.items()}
    sink = {"xyz": torch.zeros(N, 3, device="cuda"), "scales": torch.zeros(N, 3, device="cuda"),
            "uquats": torch.zeros(N, 4, device="cuda")}
    for f, g in zip(frames, gs_):
        off = _t(sc.positions(f) - sc.positions(0))
        r = gs.preprocess_ortho(ref["xyz"], ref["scale"], ref["rotate"], extr, W, H, nearest=0.01, offset=off)
        torch.autograd.backward(list(r[:3]), g)
        r = gs.preprocess_ortho(snk["xyz"], snk["scale"], snk["rotate"], extr, W, H, nearest=0.01, offset=off, grad_sink=sink)
        torch.autograd.backward(list(r[:3]), g)
    for name, key in (("xyz", "xyz"), ("scales", "scale"), ("uquats", "rotate")):
        assert snk[key].grad is None
        b = ref[key].grad.cpu().numpy()
        np.testing.assert_allclose(sink[name].cpu().numpy(), b, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(b).max())))


def test_compute_sh_into_accumulates():
    N = 30_000
    rng = np.random.default_rng(2)
    shs = rng.normal(size=(N, 16, 3)).astype(np.float32) * 0.3
    dirs = rng.normal(size=(N, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    vis = rng.random(N) > 0.1
    gcol = [rng.normal(size=(N, 3)).astype(np.float32) for _ in range(3)]
    a = _t(shs, True)
    b = _t(shs, True)
    sink = torch.zeros(N, 16, 3, device="cuda")
    for g in gcol:
        gs.compute_sh(a, 3, _t(dirs), _t(vis)).backward(_t(g))
        rgb = gs.compute_sh_into(b, 3, _t(dirs), _t(vis), sink)
        rgb.backward(_t(g))
    assert b.grad is None
    ref = a.grad.cpu().numpy()
    np.testing.assert_allclose(sink.cpu().numpy(), ref, rtol=1e-5, atol=1e-6 * float(np.abs(ref).max()))


def test_sort_gaussian_capped_matches_sort_gaussian():
    N, W, H = 40_000, 320, 208
    sc = make_scene(N, W, H, C=3, seed=6)
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(_t(sc.positions(1)), _t(sc.scale), _t(sc.rotate), _t(sc.extr), W, H,
                                                          nearest=0.01)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    M = idx.numel()
    idx2, tr2, st = gs.sort_gaussian_capped(uv, depth, W, H, radius, capacity=M + 1000)
    assert st.check() == M and idx2.numel() == M + 1000
    assert torch.equal(tr, tr2) and torch.equal(idx, idx2[:M])
    # the capped result drives the blend exactly like the exact one (pair-mode backward included)
    feat = _t(sc.feature, True); op = _t(sc.opacity, True)
    g = torch.randn(3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    outs = []
    for i_, t_ in ((idx, tr), (idx2, tr2)):
        feat.grad = None; op.grad = None
        uvg = uv.detach().requires_grad_(); cg = conic.detach().requires_grad_()
        img = gs.alpha_blending(uvg, cg, op, feat, i_, t_, 0.0, W, H)
        img.backward(g)
        outs.append([img.detach(), uvg.grad, cg.grad, op.grad.clone(), feat.grad.clone()])
    assert torch.equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1:], outs[1][1:]):   # (survivors carried over a super-batch boundary add their part with float
        assert torch.allclose(x, y, rtol=2e-4, atol=2e-6 * float(y.abs().max()))   # atomics: not bit-reproducible)
    # too small: flagged, and check() raises
    _, _, st3 = gs.sort_gaussian_capped(uv, depth, W, H, radius, capacity=M // 2)
    with pytest.raises(Exception):
        st3.check()


def test_alpha_blending_shared_equals_separate_blends():
    """three feature sets in one forward pass == the reference renderer's three calls: images bit-identical, gradients
    and taps equal (same native backward per set, different summation order of the totals)"""
    N, W, H, K = 12_000, 208, 144, 12
    sc = make_scene(N, W, H, C=3, seed=21)
    rng = np.random.default_rng(3)
    attrs = rng.uniform(-1, 1, size=(N, 19)).astype(np.float32)
    uv0, depth0, conic0, radius, tiles = gs.preprocess_ortho(_t(sc.positions(1)), _t(sc.scale), _t(sc.rotate), _t(sc.extr), W, H,
                                                             nearest=0.01)
    idx, tr = gs.sort_gaussian(uv0, depth0, W, H, radius, tiles)
    g = [_t(rng.normal(size=(c, H, W)).astype(np.float32)) for c in (3, 1, 19)]
    res = []
    for shared in (False, True):
        uv = uv0.detach().requires_grad_(); conic = conic0.detach().requires_grad_(); dep = depth0.detach().requires_grad_()
        op = _t(sc.opacity, True); rgb = _t(sc.feature, True); at = _t(attrs, True)
        ndc = torch.zeros_like(uv, requires_grad=True); andc = torch.zeros_like(uv, requires_grad=True)
        if shared:
            i1, i2, i3, nc, gi = gs.alpha_blending_shared(uv, conic, op, [rgb, dep, at], idx, tr, [0.25, 1.0, 0.0], W, H, ndc, andc,
                                                          K=K, detach_opacity=[False, False, True], taps=[True, False, False])
        else:
            i1, nc, gi = gs.alpha_blending_enhanced(uv, conic, op, rgb, idx, tr, 0.25, W, H, ndc, andc, K=K)
            i2 = gs.alpha_blending(uv, conic, op, dep, idx, tr, 1.0, W, H, ndc.detach())
            i3 = gs.alpha_blending(uv, conic, op.detach(), at, idx, tr, 0.0, W, H, ndc.detach())
        ((i1 * g[0]).sum() + (i2 * g[1]).sum() + (i3 * g[2]).sum()).backward()
        res.append(dict(imgs=[i1.detach(), i2.detach(), i3.detach()], nc=nc, gi=gi,
                        grads=[uv.grad, conic.grad, op.grad, rgb.grad, dep.grad, at.grad, ndc.grad, andc.grad]))
    a, b = res
    for x, y in zip(a["imgs"], b["imgs"]):
        assert torch.equal(x, y)
    assert torch.equal(a["nc"], b["nc"]) and torch.equal(a["gi"], b["gi"])
    for x, y in zip(a["grads"], b["grads"]):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6 * float(x.abs().max()))
