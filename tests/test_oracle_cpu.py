"""Pins the CPU oracle (oracle/splat_oracle.c):
  (1) against the golden vectors generated from the importable reference torch twins
      (tests/golden/make_golden.py): ortho project_point, ortho EWA, SH basis, cov3d formula;
  (2) against an independent float64 torch twin + autograd for every analytic backward;
  (3) through the mathematical identities the domain offers.
"""
import glob
import os

import numpy as np
import pytest
import torch

import torch_twin as tw
from splatter_a_video_amd.synth import make_scene

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ortho_*.npz")))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def T64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


# ------------------------------------------------------------------ (1) golden vectors
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_twins(oracle_mod, path):
    o = oracle_mod
    g = dict(np.load(path))
    W, H = int(g["W"]), int(g["H"])
    uv, depth = o.project_point_ortho_forward(g["xyz"], g["extr"], W, H, 0.01)
    # same culling decisions, coordinates to float32 rounding of a 3-term dot product
    assert ((depth != 0) == (g["depth"] != 0)).all()
    np.testing.assert_allclose(uv, g["uv"], rtol=1e-6, atol=3e-5)
    np.testing.assert_allclose(depth, g["depth"], rtol=1e-6, atol=1e-6)
    dx = o.project_point_ortho_backward(g["extr"], W, H, g["depth"], g["g_uv"], np.zeros_like(g["g_d"]))
    np.testing.assert_allclose(dx, g["dxyz_uv"], rtol=1e-5, atol=1e-4)
    dx = o.project_point_ortho_backward(g["extr"], W, H, g["depth"], np.zeros_like(g["g_uv"]), g["g_d"])
    np.testing.assert_allclose(dx, g["dxyz_d"], rtol=1e-6, atol=1e-6)

    vis = g["depth"].reshape(-1) != 0
    cov = o.compute_cov3d_forward(g["scale"], g["rotate"], vis)
    np.testing.assert_allclose(cov, g["cov3d"], rtol=1e-5, atol=1e-9)

    conic, radius, tiles = o.ewa_project_forward(g["xyz"], g["cov3d"], None, g["extr"], g["uv"], W, H, vis, ortho=True)
    assert (radius == g["radius"]).all()
    assert (tiles == g["tiles"]).all()
    np.testing.assert_allclose(conic, g["conic"], rtol=1e-5, atol=1e-7)
    _, dcov, _, _ = o.ewa_project_backward(g["xyz"], g["cov3d"], None, g["extr"], g["radius"], g["g_conic"], W, H,
                                           ortho=True)
    assert rel(dcov, g["dcov3d"]) < 1e-5

    for deg in range(4):
        nb = (deg + 1) ** 2
        c = o.compute_sh_forward(np.ascontiguousarray(g["shs"][:, :nb, :]), deg, g["dirs"], free=True)
        np.testing.assert_allclose(c, g[f"sh_deg{deg}"], rtol=1e-5, atol=1e-6)
        c2, clamped = o.compute_sh_forward(np.ascontiguousarray(g["shs"][:, :nb, :]), deg, g["dirs"])
        ref = g[f"sh_deg{deg}"] + 0.5
        np.testing.assert_allclose(c2, np.maximum(ref, 0), rtol=1e-5, atol=1e-6)
        assert (clamped == (ref < 0)).mean() > 0.999


# ------------------------------------------------------------------ (2) analytic backward vs autograd
@pytest.mark.parametrize("ortho", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_oracle_backward_matches_autograd(oracle_mod, ortho, bias):
    o = oracle_mod
    sc = make_scene(250, 64, 48, seed=5 + int(bias), ortho=ortho)
    rng = np.random.default_rng(3)
    xyz = sc.positions(2)
    if ortho:
        uv, depth = o.project_point_ortho_forward(xyz, sc.extr, sc.W, sc.H, 0.01)
    else:
        uv, depth = o.project_point_forward(xyz, sc.intr, sc.extr, sc.W, sc.H, 0.2)
    vis = depth.reshape(-1) != 0
    cov = o.compute_cov3d_forward(sc.scale, sc.rotate, vis)
    conic, radius, tiles = o.ewa_project_forward(xyz, cov, sc.intr, sc.extr, uv, sc.W, sc.H, vis, ortho=ortho)
    idx, tr = o.sort_gaussian(uv, depth, sc.W, sc.H, radius, tiles)
    C = 5
    bg = 0.3
    feat = rng.uniform(size=(sc.N, C)).astype(np.float32)
    ob = (rng.uniform(-0.05, 0.1, size=(sc.N, 1)).astype(np.float32)) if bias else None
    out, fT, nc = o.alpha_blending_forward(uv, conic, sc.opacity, feat, idx, tr, bg, sc.W, sc.H, opacity_bias=ob)
    g = rng.normal(size=out.shape).astype(np.float32)

    txyz = T64(xyz).requires_grad_(True); tscale = T64(sc.scale).requires_grad_(True)
    tq = T64(sc.rotate).requires_grad_(True); top = T64(sc.opacity).requires_grad_(True)
    tfeat = T64(feat).requires_grad_(True); tintr = T64(sc.intr).requires_grad_(True)
    textr = T64(sc.extr[:3, :4]).requires_grad_(True)
    tb = T64(ob).requires_grad_(True) if bias else None
    if ortho:
        tuv, td = tw.project_point_ortho(txyz, textr, sc.W, sc.H, 0.01)
    else:
        tuv, td = tw.project_point_persp(txyz, tintr, textr, sc.W, sc.H, 0.2)
    assert rel(uv, tuv.detach()) < 1e-6 and rel(depth, td.detach()) < 1e-6
    tcov = tw.cov3d(tscale, tq, torch.tensor(vis))
    assert rel(cov, tcov.detach()) < 1e-5
    tcov.retain_grad(); tuv.retain_grad()
    tconic = tw.ewa(txyz, tcov, tintr, textr, sc.W, sc.H, torch.tensor(radius > 0), ortho=ortho)
    tconic.retain_grad()
    assert rel(conic, tconic.detach()) < 1e-5
    tout, tfT, tnc, _ = tw.blend(tuv, tconic, top, tfeat, torch.tensor(idx), torch.tensor(tr), bg, sc.W, sc.H, bias=tb)
    assert rel(out, tout.detach()) < 1e-5
    assert rel(fT, tfT) < 1e-5
    assert (nc != tnc.numpy()).mean() < 1e-3
    (tout * T64(g)).sum().backward()

    res = o.alpha_blending_backward(uv, conic, sc.opacity, feat, idx, tr, bg, sc.W, sc.H, fT, nc, g, opacity_bias=ob)
    duv, dcon, dop, df, dabs = res[:5]
    assert rel(duv, tuv.grad) < 2e-5
    assert rel(dcon, tconic.grad) < 2e-5
    assert rel(dop, top.grad) < 2e-5
    assert rel(df, tfeat.grad) < 2e-5
    assert (dabs >= np.abs(duv) - 1e-6).all()
    if bias:
        assert rel(res[5], tb.grad) < 2e-5

    dxyz_e, dcov, dintr_e, dextr_e = o.ewa_project_backward(xyz, cov, sc.intr, sc.extr, radius, dcon, sc.W, sc.H,
                                                            ortho=ortho)
    assert rel(dcov, tcov.grad) < 2e-5
    if ortho:
        dxyz_p = o.project_point_ortho_backward(sc.extr, sc.W, sc.H, depth, duv, np.zeros_like(depth))
        assert np.abs(dxyz_e).max() == 0.0
    else:
        dxyz_p, dintr_p, dextr_p = o.project_point_backward(xyz, sc.intr, sc.extr, sc.W, sc.H, uv, depth, duv,
                                                            np.zeros_like(depth))
        assert rel(dintr_e + dintr_p, tintr.grad) < 5e-5
        assert rel(dextr_e + dextr_p, textr.grad) < 5e-5
    assert rel(dxyz_e + dxyz_p, txyz.grad) < 2e-5
    ds, dq = o.compute_cov3d_backward(sc.scale, sc.rotate, vis, dcov)
    assert rel(ds, tscale.grad) < 2e-5
    assert rel(dq, tq.grad) < 2e-5


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("free", [False, True])
def test_oracle_sh_backward_matches_autograd(oracle_mod, deg, free):
    o = oracle_mod
    rng = np.random.default_rng(deg)
    P, nb = 200, (deg + 1) ** 2
    shs = rng.normal(0, 0.5, size=(P, nb, 3)).astype(np.float32)
    dirs = rng.normal(size=(P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    g = rng.normal(size=(P, 3)).astype(np.float32)
    tsh = T64(shs).requires_grad_(True); td = T64(dirs).requires_grad_(True)
    tc = tw.sh_color(tsh, deg, td, free=free)
    (tc * T64(g)).sum().backward()
    if free:
        c = o.compute_sh_forward(shs, deg, dirs, free=True); clamped = None
    else:
        c, clamped = o.compute_sh_forward(shs, deg, dirs)
    assert rel(c, tc.detach()) < 1e-5
    dshs, ddirs = o.compute_sh_backward(shs, deg, dirs, None, clamped, g, free=free)
    assert rel(dshs, tsh.grad) < 2e-5
    if deg > 0:
        assert rel(ddirs, td.grad) < 2e-5


# ------------------------------------------------------------------ (3) identities / properties
def _pipeline(o, sc, f=0, C=3, K=0):
    feat = np.random.default_rng(9).uniform(size=(sc.N, C)).astype(np.float32)
    res, saved = o.render_forward(sc.positions(f), sc.scale, sc.rotate, sc.opacity, feat, sc.intr, sc.extr, sc.W,
                                  sc.H, sc.bg, ortho=sc.ortho, K=K)
    return res, saved, feat


def test_sort_invariants(oracle_mod):
    o = oracle_mod
    sc = make_scene(3000, 100, 60, seed=21)
    _, s, _ = _pipeline(o, sc)
    idx, tr, tiles, radius, depth = s["idx_sorted"], s["tile_range"], s["tiles"], s["radius"], s["depth"].reshape(-1)
    M = int(tiles.sum())
    assert idx.size == M
    # ranges tile the array exactly, in tile order, empty tiles are (0,0)
    pos = 0
    for t in range(tr.shape[0]):
        a, b = tr[t]
        if a == 0 and b == 0:
            continue
        assert a == pos and b > a
        d = depth[idx[a:b]]
        assert (np.diff(d) >= 0).all()                       # depth ascending inside a tile
        ties = np.diff(d) == 0
        assert (np.diff(idx[a:b])[ties] > 0).all()           # ties by ascending id (stable)
        pos = b
    assert pos == M
    # every Gaussian appears exactly `tiles` times
    assert (np.bincount(idx, minlength=sc.N) == tiles).all()
    assert ((radius > 0) == (tiles > 0)).all()


def test_blend_identities(oracle_mod):
    o = oracle_mod
    sc = make_scene(1500, 64, 48, seed=33)
    (out, fT, nc), s, feat = _pipeline(o, sc, C=2)
    ones = np.ones((sc.N, 1), np.float32)
    # sum of weights + final_T == 1  (feature == 1, bg == 0)
    o1, fT1, _ = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, ones, s["idx_sorted"], s["tile_range"], 0.0,
                                          sc.W, sc.H)
    np.testing.assert_allclose(o1[0] + fT1, 1.0, atol=2e-6)
    # bg enters as T*bg, same for every channel
    ob, _, _ = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, feat, s["idx_sorted"], s["tile_range"], 0.7,
                                        sc.W, sc.H)
    np.testing.assert_allclose(ob - out, np.broadcast_to(0.7 * fT[None], out.shape), atol=2e-6)
    # no Gaussians at all -> background image, T = 1, ncontrib = 0
    tr0 = np.zeros_like(s["tile_range"])
    oe, fe, ne = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, feat, np.zeros(0, np.int32), tr0, 0.25, sc.W, sc.H)
    assert (oe == 0.25).all() and (fe == 1).all() and (ne == 0).all()
    # channel chunking does not change the image (C = 40 > 32)
    f40 = np.random.default_rng(1).uniform(size=(sc.N, 40)).astype(np.float32)
    o40, _, _ = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, f40, s["idx_sorted"], s["tile_range"], 0.0, sc.W, sc.H)
    o8, _, _ = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, f40[:, 32:], s["idx_sorted"], s["tile_range"], 0.0, sc.W, sc.H)
    assert (o40[32:] == o8).all()


def test_single_isotropic_gaussian_centre_alpha(oracle_mod):
    o = oracle_mod
    W = H = 32
    uv = np.array([[10.0, 12.0]], np.float32)
    conic = np.array([[0.25, 0.0, 0.25]], np.float32)
    for op in (0.5, 0.999):
        out, fT, nc = o.alpha_blending_forward(uv, conic, np.array([[op]], np.float32), np.ones((1, 1), np.float32),
                                               np.zeros(1, np.int32), np.array([[0, 1]] + [[0, 0]] * 3, np.int32), 0.0, W, H)
        assert abs(out[0, 12, 10] - min(0.99, op)) < 1e-6
        assert nc[12, 10] == 1 and nc[15, 15] in (0, 1)
        assert out[0, 20, 20] == 0.0          # other tiles untouched


def test_enhanced_indices_and_truncation(oracle_mod):
    o = oracle_mod
    sc = make_scene(1200, 48, 48, seed=41)
    (out, fT, nc), s, feat = _pipeline(o, sc, C=3)
    K = 4
    oe, fTe, nce, gi = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, feat, s["idx_sorted"], s["tile_range"],
                                                0.0, sc.W, sc.H, K=K)
    assert (oe == out).all() and (nce == nc).all()
    # gs_idx == first K contributing ids of the float64 twin
    t = tw.blend(T64(s["uv"]), T64(s["conic"]), T64(sc.opacity), T64(feat), torch.tensor(s["idx_sorted"]),
                 torch.tensor(s["tile_range"]), 0.0, sc.W, sc.H, K=K)
    assert (gi != t[3].numpy()).mean() < 2e-3
    ot, fTt, nct, git = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, feat, s["idx_sorted"], s["tile_range"],
                                                 0.0, sc.W, sc.H, K=K, enable_truncation=True)
    assert (git == gi).all()
    full = (gi >= 0).all(-1)
    assert (fTt[full] >= fT[full] - 1e-7).all()   # truncated pixels keep more transmittance
    assert (ot[:, ~full] == out[:, ~full]).all()


def test_abs_grad_sums_per_channel_chunk(oracle_mod):
    """dL_dabs_uv adds |.| per <=32-channel launch (reference chunk loop, alpha_blending.cu:440-577)."""
    o = oracle_mod
    sc = make_scene(400, 32, 32, seed=51)
    (out, fT, nc), s, _ = _pipeline(o, sc, C=1)
    rng = np.random.default_rng(2)
    f40 = rng.uniform(size=(sc.N, 40)).astype(np.float32)
    o40, fT40, nc40 = o.alpha_blending_forward(s["uv"], s["conic"], sc.opacity, f40, s["idx_sorted"], s["tile_range"], 0.0, sc.W, sc.H)
    g = rng.normal(size=o40.shape).astype(np.float32)
    args = (s["uv"], s["conic"], sc.opacity)
    full = o.alpha_blending_backward(*args, f40, s["idx_sorted"], s["tile_range"], 0.0, sc.W, sc.H, fT40, nc40, g)
    a = o.alpha_blending_backward(*args, f40[:, :32], s["idx_sorted"], s["tile_range"], 0.0, sc.W, sc.H, fT40, nc40, g[:32])
    b = o.alpha_blending_backward(*args, f40[:, 32:], s["idx_sorted"], s["tile_range"], 0.0, sc.W, sc.H, fT40, nc40, g[32:])
    np.testing.assert_allclose(full[4], a[4] + b[4], rtol=1e-5, atol=1e-6)     # abs: sum of per-chunk |.|
    np.testing.assert_allclose(full[0], a[0] + b[0], rtol=1e-4, atol=1e-5)     # signed: linear


# ------------------------------------------------------------------ PyTorch-eager restatement (the bench's cpu_baseline leg)
def test_torch_eager_frame_matches_c_oracle(oracle_mod):
    """oracle/torch_eager.py (what bench.py times as the PyTorch-eager CPU baseline) renders the same frame and the same
    gradients as the C oracle -- two independently written restatements of the reference's semantics."""
    oracle = oracle_mod
    from oracle import torch_eager as te
    from splatter_a_video_amd.synth import make_scene
    sc = make_scene(1500, 100, 60, seed=21)
    sc.bg = 0.2
    g = np.random.default_rng(0).normal(size=(3, sc.H, sc.W)).astype(np.float32)
    img, p, M = te.frame_forward(sc, 0, use_sh=True, dL_dout=g)
    xyz = sc.positions(0)
    (out, fT, nc), saved = oracle.render_forward(xyz, sc.scale, sc.rotate, sc.opacity, None, sc.intr, sc.extr, sc.W, sc.H, sc.bg,
                                                 ortho=True, shs=sc.shs)
    assert M == saved["idx_sorted"].size
    bad = np.abs(img.detach().numpy() - out) > 1e-5 + 1e-4 * np.abs(out)
    assert bad.mean() < 1e-3
    gr = oracle.render_backward(xyz, sc.scale, sc.rotate, sc.opacity, sc.intr, sc.extr, sc.W, sc.H, sc.bg, saved, g, ortho=True,
                                shs=sc.shs)
    for name, key in (("xyz", "xyz"), ("scale", "scale"), ("rotate", "rotate"), ("opacity", "opacity"), ("shs", "shs")):
        a = p[name].grad.numpy().reshape(-1).astype(np.float64); b = gr[key].reshape(-1).astype(np.float64)
        mx = np.abs(b).max()
        assert (np.abs(a - b) > 2e-3 * np.abs(b) + 1e-4 * mx).mean() < 2e-3, name
