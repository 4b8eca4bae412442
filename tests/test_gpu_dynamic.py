"""Row a15 on the GPU: the HIP dynamic-evaluation path (C ABI through splatter_a_video_amd.dynamics) against
(1) vectors produced by the reference's own methods, (2) the C oracle at a larger random size,
(3) the accumulate-into-bucket mode, (4) the reference-named getters."""
import os

import numpy as np
import pytest
import torch

import oracle
from splatter_a_video_amd.dynamics import DynamicGaussians, FrameClock, evaluate

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "dynamic_400x50.npz")
NAMES = ("position", "pos_cubic_node", "rotation", "rot_poly_feat", "rot_fourier_feat", "opacity", "scaling")


def _dev(a, grad=False):
    return torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=grad)


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(G))


@pytest.mark.parametrize("t", [0, 1, 5, 24, 25, 44, 45, 49])
def test_hip_matches_reference_vectors(gold, t):
    g = gold
    clock = FrameClock(int(g["T"]), g["intervals"], int(g["start_frame_id"]), int(g["time_len"]))
    p = {k: _dev(g[k], grad=True) for k in NAMES}
    pos, rot, opa, scl = evaluate(clock, t, **p)
    pre = f"t{t}_"
    tol = dict(rtol=3e-6, atol=3e-6)       # float32 path; FMA contraction and sum order differ from eager torch
    np.testing.assert_allclose(pos.detach().cpu().numpy(), g[pre + "pos"], **tol)
    np.testing.assert_allclose(rot.detach().cpu().numpy(), g[pre + "rot"], **tol)
    np.testing.assert_allclose(opa.detach().cpu().numpy(), g[pre + "opa"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(scl.detach().cpu().numpy(), g[pre + "scl"], rtol=3e-6, atol=1e-9)
    loss = (pos * _dev(g[pre + "g_pos"])).sum() + (rot * _dev(g[pre + "g_rot"])).sum() \
        + (opa * _dev(g[pre + "g_opa"])).sum() + (scl * _dev(g[pre + "g_scl"])).sum()
    loss.backward()
    np.testing.assert_allclose(p["position"].grad.cpu().numpy(), g[pre + "d_position"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(p["pos_cubic_node"].grad.cpu().numpy(), g[pre + "d_cubic"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(p["rotation"].grad.cpu().numpy(), g[pre + "d_rotation"], rtol=3e-5, atol=3e-6)
    np.testing.assert_allclose(p["opacity"].grad.cpu().numpy(), g[pre + "d_opacity"], rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(p["scaling"].grad.cpu().numpy(), g[pre + "d_scaling"], rtol=3e-6, atol=1e-9)
    assert p["rot_poly_feat"].grad is None and p["rot_fourier_feat"].grad is None     # detached in the reference


def _random_params(N, I, seed):
    rng = np.random.default_rng(seed)
    f = lambda *s, scale=1.0: rng.normal(0, scale, size=s).astype(np.float32)
    return dict(position=f(N, 3), pos_cubic_node=f(N, 4 * I * 3, scale=0.1), rotation=f(N, 4),
                rot_poly_feat=f(N, 4, 4, scale=0.05), rot_fourier_feat=f(N, 8, 4, scale=0.05),
                opacity=f(N, 1, scale=1.5), scaling=f(N, 3, scale=0.5) - 4.0), rng


@pytest.mark.parametrize("N,T", [(1, 6), (3, 11), (257, 23), (100_003, 250)])
def test_hip_matches_oracle(N, T):
    clock = FrameClock(T)
    I = clock.interval_num
    host, rng = _random_params(N, I, seed=N)
    for t in sorted({0, 1, T // 2, T - 1}):
        seg, d, basis = clock.scalars(t)
        b = np.array(list(basis), np.float32)
        o_pos, o_rot, o_opa, o_scl = oracle.dynamic_eval_forward(
            host["position"], host["pos_cubic_node"], host["rotation"], host["rot_poly_feat"], host["rot_fourier_feat"],
            host["opacity"], host["scaling"], seg, d, b[:4], b[4:])
        p = {k: _dev(v, grad=True) for k, v in host.items()}
        pos, rot, opa, scl = evaluate(clock, t, **p)
        np.testing.assert_allclose(pos.detach().cpu().numpy(), o_pos, rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(rot.detach().cpu().numpy(), o_rot, rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(opa.detach().cpu().numpy(), o_opa, rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(scl.detach().cpu().numpy(), o_scl, rtol=3e-6, atol=1e-9)
        g = dict(pos=rng.normal(size=(N, 3)), rot=rng.normal(size=(N, 4)), opa=rng.normal(size=(N, 1)),
                 scl=rng.normal(size=(N, 3)))
        g = {k: v.astype(np.float32) for k, v in g.items()}
        torch.autograd.backward([pos, rot, opa, scl], [_dev(g["pos"]), _dev(g["rot"]), _dev(g["opa"]), _dev(g["scl"])])
        o = oracle.dynamic_eval_backward((N, 4, I, 3), host["rotation"], host["rot_poly_feat"], host["rot_fourier_feat"],
                                         host["opacity"], host["scaling"], seg, d, b[:4], b[4:], g["pos"], g["rot"],
                                         g["opa"], g["scl"])
        np.testing.assert_allclose(p["position"].grad.cpu().numpy(), o[0], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(p["pos_cubic_node"].grad.cpu().numpy().reshape(N, 4, I, 3), o[1], rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(p["rotation"].grad.cpu().numpy(), o[2], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(p["opacity"].grad.cpu().numpy(), o[3], rtol=3e-5, atol=1e-7)
        np.testing.assert_allclose(p["scaling"].grad.cpu().numpy(), o[4], rtol=3e-6, atol=1e-9)


def test_grad_sink_accumulates_over_frames():
    """backward adds straight into caller-owned gradient buffers; three frames == the sum of three dense autograd runs"""
    N, T = 5000, 40
    clock = FrameClock(T)
    host, rng = _random_params(N, clock.interval_num, seed=11)
    frames = [3, 17, 18]
    gs = [{k: _dev(rng.normal(size=s)) for k, s in (("pos", (N, 3)), ("rot", (N, 4)), ("opa", (N, 1)), ("scl", (N, 3)))}
          for _ in frames]
    # dense autograd
    p = {k: _dev(v, grad=True) for k, v in host.items()}
    for t, g in zip(frames, gs):
        out = evaluate(clock, t, **p)
        torch.autograd.backward(list(out), [g["pos"], g["rot"], g["opa"], g["scl"]])
    # sink mode
    q = {k: _dev(v, grad=True) for k, v in host.items()}
    sink = {k: torch.zeros_like(q[k]) for k in ("position", "pos_cubic_node", "rotation", "opacity", "scaling")}
    for t, g in zip(frames, gs):
        out = evaluate(clock, t, grad_sink=sink, **q)
        torch.autograd.backward(list(out), [g["pos"], g["rot"], g["opa"], g["scl"]])
    for k in sink:
        assert q[k].grad is None
        np.testing.assert_allclose(sink[k].cpu().numpy(), p[k].grad.cpu().numpy(), rtol=2e-6, atol=1e-6)


def test_reference_named_getters_and_frozen_position():
    N, T = 777, 30
    clock = FrameClock(T)
    host, _ = _random_params(N, clock.interval_num, seed=5)
    m = DynamicGaussians(clock, **{k: _dev(v) for k, v in host.items()})
    pos, rot, opa, scl = m.frame(12)
    assert torch.equal(m.get_position(12), pos) and torch.equal(m.get_rotation(12), rot)
    assert torch.equal(m.get_opacity, opa) and torch.equal(m.get_scaling, scl)
    assert torch.allclose(rot.norm(dim=1), torch.ones(N, device="cuda"), atol=1e-6)
    (pos.sum() + rot[:, 0].sum() + opa.sum() + scl.sum()).backward()
    assert m.position.grad is None                       # position is not optimised in the reference (:90)
    assert m.pos_cubic_node.grad is not None and m.rotation.grad is not None
    seg = clock.scalars(12)[0]
    dense = m.pos_cubic_node.grad.reshape(N, 4, clock.interval_num, 3)
    other = [i for i in range(clock.interval_num) if i != seg]
    assert not dense[:, :, other].any() and dense[:, :, seg].abs().sum() > 0


def test_degenerate_quaternion_and_empty():
    clock = FrameClock(10)
    z = lambda *s: torch.zeros(*s, device="cuda")
    rot = evaluate(clock, 4, rotation=z(8, 4), rot_poly_feat=z(8, 4, 4), rot_fourier_feat=z(8, 8, 4))[1]
    assert torch.equal(rot, z(8, 4))                      # F.normalize of a zero vector is zero
    out = evaluate(clock, 4, opacity=z(0, 1), scaling=z(0, 3))
    assert out[2].shape == (0, 1) and out[3].shape == (0, 3)


def test_segment_major_layout_is_a_pure_permutation():
    """native [I,N,4,3] spline table: same values, same gradients (permuted), checkpoint export round-trips"""
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, to_gaussian_major, to_segment_major
    N, T = 4099, 60
    clock = FrameClock(T)
    I = clock.interval_num
    host, rng = _random_params(N, I, seed=3)
    g_pos = _dev(rng.normal(size=(N, 3)))
    for t in (0, 7, 30, 59):
        a = {k: _dev(v, grad=True) for k, v in host.items()}
        pos_a = evaluate(clock, t, position=a["position"], pos_cubic_node=a["pos_cubic_node"])[0]
        pos_a.backward(g_pos)
        seg_tab = to_segment_major(_dev(host["pos_cubic_node"]), I).requires_grad_()
        assert seg_tab.shape == (I, N, 4, 3)
        pos_b = evaluate(clock, t, position=_dev(host["position"]), pos_cubic_node=seg_tab, cubic_layout=SEGMENT_MAJOR)[0]
        pos_b.backward(g_pos)
        assert torch.equal(pos_a, pos_b)
        assert torch.equal(to_gaussian_major(seg_tab.grad), a["pos_cubic_node"].grad)
    assert torch.equal(to_gaussian_major(to_segment_major(_dev(host["pos_cubic_node"]), I)), _dev(host["pos_cubic_node"]))
    m = DynamicGaussians(clock, **{k: _dev(v) for k, v in host.items()}, cubic_layout=SEGMENT_MAJOR)
    assert m.pos_cubic_node.shape == (I, N, 4, 3)
    assert torch.equal(m.reference_pos_cubic_node(), _dev(host["pos_cubic_node"]))
    ref = DynamicGaussians(clock, **{k: _dev(v) for k, v in host.items()})
    for x, y in zip(m.frame(33), ref.frame(33)):
        assert torch.equal(x, y)


def _model(N, T, seed, W, H):
    """dynamic Gaussians that land inside a W x H ortho view: positions in [-1,1]^2 x [2,4], pixel-sized splats"""
    clock = FrameClock(T)
    host, rng = _random_params(N, clock.interval_num, seed=seed)
    host["position"] = np.concatenate([rng.uniform(-1.1, 1.1, size=(N, 2)), rng.uniform(2.0, 4.0, size=(N, 1))], 1).astype(np.float32)
    host["pos_cubic_node"] = (host["pos_cubic_node"] * 0.2).astype(np.float32)
    host["scaling"] = np.log(rng.uniform(0.004, 0.02, size=(N, 3))).astype(np.float32)
    extr = np.eye(4, dtype=np.float32)[:3]
    return clock, host, extr, rng


@pytest.mark.parametrize("N,T,W,H", [(1, 6, 32, 32), (4001, 30, 160, 96), (60_000, 250, 854, 480)])
def test_frame_preprocess_equals_evaluate_then_preprocess(N, T, W, H):
    import dptr.gs as gs
    from splatter_a_video_amd.dynamics import frame_preprocess
    clock, host, extr, rng = _model(N, T, 7 + N, W, H)
    g = [_dev(rng.normal(size=s)) for s in ((N, 2), (N, 1), (N, 3), (N, 1))]
    for t in sorted({0, T // 3, T - 1}):
        a = {k: _dev(v, grad=True) for k, v in host.items()}
        pos, rot, opa, scl = evaluate(clock, t, **a)
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(pos, scl, rot, _dev(extr), W, H, nearest=0.01)
        torch.autograd.backward([uv, depth, conic, opa], g)
        b = {k: _dev(v, grad=True) for k, v in host.items()}
        uv2, depth2, conic2, radius2, tiles2, opa2 = frame_preprocess(clock, t, _dev(extr), W, H, nearest=0.01, **b)
        torch.autograd.backward([uv2, depth2, conic2, opa2], g)
        assert N == 1 or (radius > 0).sum() > N // 4                    # the scene is actually on screen
        assert torch.equal(radius, radius2) and torch.equal(tiles, tiles2)
        np.testing.assert_allclose(uv2.detach().cpu().numpy(), uv.detach().cpu().numpy(), rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(depth2.detach().cpu().numpy(), depth.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
        c0 = conic.detach().cpu().numpy()
        np.testing.assert_allclose(conic2.detach().cpu().numpy(), c0, rtol=5e-5, atol=2e-6 * max(1.0, float(np.abs(c0).max())))
        np.testing.assert_allclose(opa2.detach().cpu().numpy(), opa.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
        for k in ("position", "pos_cubic_node", "rotation", "opacity", "scaling"):
            x, y = b[k].grad.cpu().numpy(), a[k].grad.cpu().numpy()
            np.testing.assert_allclose(x, y, rtol=3e-4, atol=2e-6 * max(1.0, float(np.abs(y).max())), err_msg=k)
        assert b["rot_poly_feat"].grad is None and b["rot_fourier_feat"].grad is None


def test_frame_preprocess_matches_oracle_chain():
    from splatter_a_video_amd.dynamics import frame_preprocess
    N, T, W, H = 3000, 40, 128, 96
    clock, host, extr, rng = _model(N, T, 21, W, H)
    t = 17
    seg, d, basis = clock.scalars(t)
    bb = np.array(list(basis), np.float32)
    o_pos, o_rot, o_opa, o_scl = oracle.dynamic_eval_forward(host["position"], host["pos_cubic_node"], host["rotation"],
                                                             host["rot_poly_feat"], host["rot_fourier_feat"], host["opacity"],
                                                             host["scaling"], seg, d, bb[:4], bb[4:])
    o_uv, o_d = oracle.project_point_ortho_forward(o_pos, extr, W, H, 0.01, 1.3)
    vis = (o_d != 0).reshape(-1)
    o_cov = oracle.compute_cov3d_forward(o_scl, o_rot, vis)
    o_conic, o_r, o_t = oracle.ewa_project_forward(o_pos, o_cov, None, extr, o_uv, W, H, vis, ortho=True)
    uv, depth, conic, radius, tiles, opa = frame_preprocess(clock, t, _dev(extr), W, H, nearest=0.01,
                                                           **{k: _dev(v) for k, v in host.items()})
    np.testing.assert_allclose(uv.cpu().numpy(), o_uv, rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), o_d.reshape(-1, 1), rtol=1e-5, atol=1e-5)
    same = radius.cpu().numpy() == o_r
    assert same.mean() > 0.999                                        # ceil(3 sqrt(lambda)) may flip on a rounding tie
    np.testing.assert_allclose(conic.cpu().numpy()[same], o_conic[same], rtol=2e-4, atol=2e-5 * float(np.abs(o_conic).max()))
    np.testing.assert_allclose(opa.cpu().numpy(), o_opa, rtol=3e-6, atol=1e-7)


def test_frame_preprocess_grad_sink_and_segment_major():
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, to_gaussian_major
    N, T, W, H = 9000, 60, 256, 144
    clock, host, extr, rng = _model(N, T, 33, W, H)
    ref = DynamicGaussians(clock, **{k: _dev(v) for k, v in host.items()})
    seg = DynamicGaussians(clock, **{k: _dev(v) for k, v in host.items()}, cubic_layout=SEGMENT_MAJOR)
    sink = {k: torch.zeros_like(getattr(seg, k)) for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
    for t in (5, 31, 32):
        g = [_dev(rng.normal(size=s)) for s in ((N, 2), (N, 1), (N, 3), (N, 1))]
        r = ref.preprocess(t, _dev(extr), W, H, nearest=0.01)
        torch.autograd.backward([r[0], r[1], r[2], r[5]], g)
        r = seg.preprocess(t, _dev(extr), W, H, nearest=0.01, grad_sink=sink)
        torch.autograd.backward([r[0], r[1], r[2], r[5]], g)
    assert seg.pos_cubic_node.grad is None and seg.rotation.grad is None
    for k in ("rotation", "opacity", "scaling"):
        y = getattr(ref, k).grad.cpu().numpy()
        np.testing.assert_allclose(sink[k].cpu().numpy(), y, rtol=1e-5, atol=2e-6 * max(1.0, float(np.abs(y).max())), err_msg=k)
    y = ref.pos_cubic_node.grad.cpu().numpy()
    np.testing.assert_allclose(to_gaussian_major(sink["pos_cubic_node"]).cpu().numpy(), y, rtol=1e-5,
                               atol=2e-6 * max(1.0, float(np.abs(y).max())))


def test_dynamic_pipeline_end_to_end_against_oracle():
    """The whole per-frame path with row a15 on it -- dynamic parameters -> fused preprocess -> SH -> sync-free sort ->
    blend -> backward into gradient sinks -- against the oracle chain (dynamic_eval -> render_forward / render_backward ->
    dynamic_eval_backward), two frames accumulated."""
    import dptr.gs as gs
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, frame_preprocess, to_gaussian_major, to_segment_major
    N, T, W, H = 6000, 30, 160, 112
    clock, host, extr, rng = _model(N, T, 91, W, H)
    host["scaling"] = np.log(rng.uniform(0.01, 0.035, size=(N, 3))).astype(np.float32)   # a few pixels wide
    shs = (rng.normal(size=(N, 16, 3)) * 0.3).astype(np.float32)
    I = clock.interval_num
    p = {k: _dev(v) for k, v in host.items()}
    p["pos_cubic_node"] = to_segment_major(p["pos_cubic_node"], I)
    shs_d = _dev(shs, grad=True)
    for k in ("pos_cubic_node", "rotation", "opacity", "scaling"):
        p[k].requires_grad_()
    sink = {k: torch.zeros_like(p[k]) for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
    shs_sink = torch.zeros_like(shs_d)
    dirs = torch.zeros(N, 3, device="cuda"); dirs[:, 2] = 1.0
    want = {k: np.zeros(host[k].shape, np.float64) for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
    want_shs = np.zeros(shs.shape, np.float64)
    for t in (4, 19):
        g = rng.normal(size=(3, H, W)).astype(np.float32)
        # ---- GPU
        feat = gs.compute_sh_into(shs_d, 3, dirs, None, shs_sink)
        uv, depth, conic, radius, tiles, opa = frame_preprocess(clock, t, _dev(extr), W, H, nearest=0.01, grad_sink=sink,
                                                               cubic_layout=SEGMENT_MAJOR, **p)
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        idx2, tr2, st = gs.sort_gaussian_capped(uv, depth, W, H, radius, idx.numel() + 64)
        img = gs.alpha_blending(uv, conic, opa, feat, idx2, tr2, 0.0, W, H, None)
        img.backward(_dev(g))
        assert st.check() == idx.numel()
        # ---- oracle
        seg, d, basis = clock.scalars(t)
        b = np.array(list(basis), np.float32)
        o_pos, o_rot, o_opa, o_scl = oracle.dynamic_eval_forward(host["position"], host["pos_cubic_node"], host["rotation"],
                                                                 host["rot_poly_feat"], host["rot_fourier_feat"],
                                                                 host["opacity"], host["scaling"], seg, d, b[:4], b[4:])
        (out, fT, nc), saved = oracle.render_forward(o_pos, o_scl, o_rot, o_opa, None, None, extr, W, H, 0.0, ortho=True,
                                                     shs=shs)
        bad = np.abs(img.detach().cpu().numpy() - out) > (1e-5 + 1e-4 * np.abs(out))
        assert bad.mean() < 2e-3
        gr = oracle.render_backward(o_pos, o_scl, o_rot, o_opa, None, extr, W, H, 0.0, saved, g, ortho=True, shs=shs)
        dpos, dcub, drot, dopa, dscl = oracle.dynamic_eval_backward(
            (N, 4, I, 3), host["rotation"], host["rot_poly_feat"], host["rot_fourier_feat"], host["opacity"], host["scaling"],
            seg, d, b[:4], b[4:], gr["xyz"], gr["rotate"], gr["opacity"], gr["scale"])
        want["pos_cubic_node"] += dcub.reshape(N, -1); want["rotation"] += drot; want["opacity"] += dopa; want["scaling"] += dscl
        want_shs += gr["shs"]
    got = {k: sink[k] for k in sink}
    got["pos_cubic_node"] = to_gaussian_major(sink["pos_cubic_node"])
    for k in want:
        a, bb = got[k].cpu().numpy().reshape(-1), want[k].reshape(-1)
        assert np.abs(bb).max() > 0
        err = np.abs(a - bb).max() / np.abs(bb).max()
        assert err < 5e-3, (k, err)                         # same bar as the static chain test (gradients: rel-to-max)
    err = np.abs(shs_sink.cpu().numpy() - want_shs).max() / np.abs(want_shs).max()
    assert err < 5e-3, err
    assert all(p[k].grad is None for k in sink) and shs_d.grad is None


def test_poly_fourier_position_matches_reference():
    """splat_position_poly_fourier_* (dynamics.position_poly_fourier) against the reference's own get_position
    (src/dynamic_gaussian_points.py:169-186; tests/golden/make_golden_polyfourier.py), values and gradients, with and
    without detach_pos; the rotation of that class is the same getter the spline class uses (checked alongside)."""
    from splatter_a_video_amd.dynamics import FrameClock, evaluate, position_poly_fourier
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "polyfourier_300x40.npz")))
    clock = FrameClock(int(g["T"]), start_frame_id=int(g["start_frame_id"]), time_len=int(g["time_len"]))
    dev = "cuda"
    for t in g["times"]:
        for det in (False, True):
            pre = f"t{t}_{'det_' if det else ''}"
            p = {k: torch.tensor(g[k], device=dev, requires_grad=True) for k in ("position", "pos_poly_feat", "pos_fourier_feat", "rotation")}
            pos = position_poly_fourier(clock, int(t), p["position"], p["pos_poly_feat"], p["pos_fourier_feat"], detach_pos=det)
            np.testing.assert_allclose(pos.detach().cpu().numpy(), g[pre + "pos"], rtol=2e-6, atol=2e-6)
            (pos * torch.tensor(g[pre + "g_pos"], device=dev)).sum().backward()
            if det:
                assert p["position"].grad is None
            else:
                np.testing.assert_allclose(p["position"].grad.cpu().numpy(), g[pre + "d_position"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(p["pos_poly_feat"].grad.cpu().numpy(), g[pre + "d_pos_poly"], rtol=2e-6, atol=1e-6)
            np.testing.assert_allclose(p["pos_fourier_feat"].grad.cpu().numpy(), g[pre + "d_pos_fourier"], rtol=2e-6, atol=1e-6)
        out = evaluate(clock, int(t), rotation=p["rotation"], rot_poly_feat=torch.tensor(g["rot_poly_feat"], device=dev),
                       rot_fourier_feat=torch.tensor(g["rot_fourier_feat"], device=dev))
        rot = out[1]
        np.testing.assert_allclose(rot.detach().cpu().numpy(), g[f"t{t}_rot"], rtol=2e-5, atol=2e-6)
