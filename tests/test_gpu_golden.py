"""HIP path directly against the golden vectors generated from the reference's importable torch twins
(tests/golden/make_golden.py): ortho projection fwd/bwd, ortho EWA fwd/bwd, SH colours, cov3d."""
import glob
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ortho_*.npz")))


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_matches_reference_twins(gpu, path):
    import dptr.gs as gs
    g = dict(np.load(path))
    W, H = int(g["W"]), int(g["H"])
    extr = dev(g["extr"], gpu)
    xyz = dev(g["xyz"], gpu).requires_grad_(True)
    uv, depth = gs.project_point_ortho(xyz, extr, W, H, nearest=0.01)
    assert ((depth.detach().cpu().numpy() != 0) == (g["depth"] != 0)).all()          # same culling decisions
    np.testing.assert_allclose(uv.detach().cpu().numpy(), g["uv"], rtol=1e-6, atol=3e-5)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), g["depth"], rtol=1e-6, atol=1e-6)
    (uv * dev(g["g_uv"], gpu)).sum().backward(retain_graph=True)
    np.testing.assert_allclose(xyz.grad.cpu().numpy(), g["dxyz_uv"], rtol=1e-5, atol=1e-4)
    xyz.grad = None
    (depth * dev(g["g_d"], gpu)).sum().backward()
    np.testing.assert_allclose(xyz.grad.cpu().numpy(), g["dxyz_d"], rtol=1e-6, atol=1e-6)

    vis = dev(g["depth"].reshape(-1) != 0, gpu)
    cov = gs.compute_cov3d(dev(g["scale"], gpu), dev(g["rotate"], gpu), vis)
    np.testing.assert_allclose(cov.cpu().numpy(), g["cov3d"], rtol=2e-5, atol=1e-6 * float(np.abs(g["cov3d"]).max()))

    t_cov = dev(g["cov3d"], gpu).requires_grad_(True)
    conic, radius, tiles = gs.ewa_project_ortho(dev(g["xyz"], gpu), t_cov, extr, dev(g["uv"], gpu), W, H, vis)
    assert (radius.cpu().numpy() != g["radius"]).mean() <= 1e-3
    assert (tiles.cpu().numpy() != g["tiles"]).mean() <= 1e-3
    same = radius.cpu().numpy() == g["radius"]
    np.testing.assert_allclose(conic.detach().cpu().numpy()[same], g["conic"][same], rtol=2e-5, atol=1e-7)
    (conic * dev(g["g_conic"], gpu)).sum().backward()
    if same.all():
        assert rel(t_cov.grad.cpu().numpy(), g["dcov3d"]) < 2e-5

    for deg in range(4):
        nb = (deg + 1) ** 2
        shs = dev(np.ascontiguousarray(g["shs"][:, :nb, :]), gpu)
        c = gs.compute_sh_free(shs, deg, dev(g["dirs"], gpu))
        np.testing.assert_allclose(c.cpu().numpy(), g[f"sh_deg{deg}"], rtol=1e-5, atol=2e-6)
        c2 = gs.compute_sh(shs, deg, dev(g["dirs"], gpu))
        np.testing.assert_allclose(c2.cpu().numpy(), np.maximum(g[f"sh_deg{deg}"] + 0.5, 0), rtol=1e-5, atol=2e-6)


def test_c1_compositing_on_reference_geometry(gpu, oracle_mod):
    """BASELINE configs[0] (10k Gaussians, one 256x256 frame): screen-space geometry PRODUCED BY THE REFERENCE's own torch
    functions (golden uv / depth / conic / radius / tiles) sorted and composited forward + backward on the GPU, against
    the oracle's restatement of the CUDA sort / blend on the same reference geometry."""
    import dptr.gs as gs
    from test_gpu_parity import assert_grad
    o = oracle_mod
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ortho_c1_10k_256x256.npz")))
    W, H, N = int(g["W"]), int(g["H"]), g["xyz"].shape[0]
    rng = np.random.default_rng(7)
    opacity = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, size=(N, 1))))).astype(np.float32)
    rgb = np.maximum(g["sh_deg3"] + 0.5, 0).astype(np.float32)            # the reference's eval_sh colours (+0.5, clamp)
    idx_r, tr_r = o.sort_gaussian(g["uv"], g["depth"], W, H, g["radius"], g["tiles"])
    out_r, fT_r, nc_r = o.alpha_blending_forward(g["uv"], g["conic"], opacity, rgb, idx_r, tr_r, 0.0, W, H)[:3]
    t = {k: dev(v, gpu).requires_grad_(True) for k, v in dict(uv=g["uv"], conic=g["conic"], opacity=opacity, rgb=rgb).items()}
    idx, tr = gs.sort_gaussian(t["uv"].detach(), dev(g["depth"], gpu), W, H, dev(g["radius"], gpu), dev(g["tiles"], gpu))
    assert (idx.cpu().numpy() == idx_r).all() and (tr.cpu().numpy() == tr_r).all()          # bit-exact sort
    assert idx.numel() == int(g["tiles"].sum())
    ndc = torch.zeros(N, 2, device=gpu, requires_grad=True)
    img = gs.alpha_blending(t["uv"], t["conic"], t["opacity"], t["rgb"], idx, tr, 0.0, W, H, ndc)
    bad = np.abs(img.detach().cpu().numpy() - out_r) > 1e-5 + 1e-4 * np.abs(out_r)
    assert bad.mean() < 1e-3
    gout = rng.normal(size=(3, H, W)).astype(np.float32)
    (img * dev(gout, gpu)).sum().backward()
    gr = o.alpha_blending_backward(g["uv"], g["conic"], opacity, rgb, idx_r, tr_r, 0.0, W, H, fT_r, nc_r, gout)
    assert_grad(t["uv"].grad, gr[0], "dL_duv")
    assert_grad(t["conic"].grad, gr[1], "dL_dconic")
    assert_grad(t["opacity"].grad, gr[2], "dL_dopacity")
    assert_grad(t["rgb"].grad, gr[3], "dL_dfeature")
    assert_grad(ndc.grad, gr[0] * np.array([[0.5 * W, 0.5 * H]], np.float32), "dL_dndc")
