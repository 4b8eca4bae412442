"""``dptr.gs._C`` (the reference's 18 pybind names, src/submodules/dptr/dptr/gs/src/ext.cpp:14-33, over the C ABI) driven
with the argument tuples the reference's own operator files pass (dptr/gs/*.py), against the CPU oracle."""
import numpy as np
import pytest
import torch

from splatter_a_video_amd.synth import make_scene
from test_gpu_parity import assert_grad, oracle_geometry

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def test_reference_operator_sequence_on_the_shim(gpu, oracle_mod):
    """the body of the reference's rasterization() (dptr/gs/__init__.py:28-100) and of its autograd Functions' forward /
    backward, written against _C exactly as those files do"""
    import dptr.gs._C as _C
    o = oracle_mod
    N, W, H = 3000, 96, 64
    sc = make_scene(N, W, H, seed=31, ortho=False)
    sc.xyz[:, 2] += np.linspace(0, 1e-3, N, dtype=np.float32)        # unique depths: torch.sort is not stable
    rng = np.random.default_rng(2)
    feat = rng.uniform(size=(N, 3)).astype(np.float32)
    xyz = sc.positions(0)
    intr, extr = _t(sc.intr), _t(sc.extr)
    # forward chain: project_point.py:59, compute_cov3d.py:39, ewa_project.py:59, sort_gaussian.py:42-52, alpha_blending.py:62
    uv, depth = _C.project_point_forward(_t(xyz), intr, extr, W, H, 0.2, 1.3)
    visible = depth != 0
    cov3d = _C.compute_cov3d_forward(_t(sc.scale), _t(sc.rotate), visible)
    conic, radius, tiles = _C.ewa_project_forward(_t(xyz), cov3d, intr, extr, uv, W, H, visible)
    cum = torch.cumsum(tiles, dim=0, dtype=torch.int32)
    key, gidx = _C.compute_gaussian_key(uv, depth, W, H, radius, cum)
    key_sorted, indices = torch.sort(key)
    idx_sorted = torch.gather(gidx, 0, indices)
    tile_range = _C.compute_tile_gaussian_range(W, H, cum, key_sorted)
    img, final_T, ncontrib = _C.alpha_blending_forward(uv, conic, _t(sc.opacity), _t(feat), idx_sorted, tile_range, 0.1, W, H)

    (out_r, fT_r, nc_r), saved = o.render_forward(xyz, sc.scale, sc.rotate, sc.opacity, feat, sc.intr, sc.extr, W, H, 0.1, ortho=False)
    assert (idx_sorted.cpu().numpy() == saved["idx_sorted"]).all() and (tile_range.cpu().numpy() == saved["tile_range"]).all()
    np.testing.assert_allclose(uv.cpu().numpy(), saved["uv"], rtol=1e-5, atol=1e-4)
    bad = np.abs(img.cpu().numpy() - out_r) > 1e-5 + 1e-4 * np.abs(out_r)
    assert bad.mean() < 1e-3
    assert (ncontrib.cpu().numpy() != nc_r).mean() < 1e-3

    # backward chain: alpha_blending.py:97, ewa_project.py:71, project_point.py:77, compute_cov3d.py:51
    g = rng.normal(size=(3, H, W)).astype(np.float32)
    duv, dconic, dop, dfeat, dabs = _C.alpha_blending_backward(uv, conic, _t(sc.opacity), _t(feat), idx_sorted, tile_range, 0.1, W, H,
                                                               final_T, ncontrib, _t(g))
    dxyz_e, dcov, dintr, dextr = _C.ewa_project_backward(_t(xyz), cov3d, intr, extr, radius, dconic)
    dxyz_p, _, _ = _C.project_point_backward(_t(xyz), intr, extr, W, H, uv, depth, duv, torch.zeros_like(depth))
    dscale, dquat = _C.compute_cov3d_backward(_t(sc.scale), _t(sc.rotate), visible, dcov)
    gr = o.render_backward(xyz, sc.scale, sc.rotate, sc.opacity, sc.intr, sc.extr, W, H, 0.1, saved, g, ortho=False)
    assert_grad(dxyz_e + dxyz_p, gr["xyz"], "xyz")
    assert_grad(dscale, gr["scale"], "scale")
    assert_grad(dquat, gr["rotate"], "rotate")
    assert_grad(dop, gr["opacity"], "opacity")
    assert_grad(dfeat, gr["feature"], "feature")
    assert_grad(dabs, gr["abs_uv"], "abs_uv")
    assert dextr.shape == extr.shape and dintr.shape == (4,)


def test_shim_sh_and_blend_variants(gpu, oracle_mod):
    import dptr.gs._C as _C
    o = oracle_mod
    N, W, H, K = 2000, 80, 48, 6
    sc = make_scene(N, W, H, seed=5)
    G = oracle_geometry(o, sc)
    rng = np.random.default_rng(3)
    dirs = rng.normal(size=(N, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    vis = torch.ones(N, dtype=torch.bool, device="cuda")
    col, clamped = _C.compute_sh_forward(_t(sc.shs), 3, _t(dirs), vis)
    col_r, cl_r = o.compute_sh_forward(sc.shs, 3, dirs)
    np.testing.assert_allclose(col.cpu().numpy(), col_r, rtol=1e-5, atol=1e-6)
    assert clamped.dtype == torch.bool
    gcol = rng.normal(size=(N, 3)).astype(np.float32)
    dshs, ddirs = _C.compute_sh_backward(_t(sc.shs), 3, _t(dirs), vis, clamped, _t(gcol))
    dshs_r, ddirs_r = o.compute_sh_backward(sc.shs, 3, dirs, None, cl_r, gcol)
    assert_grad(dshs, dshs_r, "dshs")
    assert_grad(ddirs, ddirs_r, "ddirs")
    free = _C.compute_sh_free_forward(_t(sc.shs), 3, _t(dirs), vis)
    free_r = o.compute_sh_forward(sc.shs, 3, dirs, free=True)
    np.testing.assert_allclose(free.cpu().numpy(), free_r, rtol=1e-5, atol=1e-6)
    dshs_f, _ = _C.compute_sh_free_backward(_t(sc.shs), 3, _t(dirs), vis, _t(gcol))
    assert_grad(dshs_f, o.compute_sh_backward(sc.shs, 3, dirs, None, None, gcol, free=True)[0], "dshs free")

    feat = rng.uniform(size=(N, 3)).astype(np.float32)
    ob = rng.uniform(-0.05, 0.1, size=(N, 1)).astype(np.float32)
    args = (_t(G["uv"]), _t(G["conic"]), _t(sc.opacity), _t(feat))
    tail = (_t(G["idx"]), _t(G["tr"]), 0.0, W, H)
    img, fT, nc, gi = _C.alpha_blending_forward_enhanced(*args, *tail, K, False)
    ref = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, W, H, K=K)
    assert (np.abs(img.cpu().numpy() - ref[0]) > 1e-5 + 1e-4 * np.abs(ref[0])).mean() < 1e-3
    assert (gi.cpu().numpy() != ref[3]).mean() < 1e-3 and gi.shape == (H, W, K)
    g = rng.normal(size=(3, H, W)).astype(np.float32)
    r5 = _C.alpha_blending_backward_enhanced(*args, *tail, fT, nc, _t(g))
    assert len(r5) == 5
    imgb, fTb, ncb = _C.alpha_blending_forward_with_bias(*args, _t(ob), *tail)
    refb = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, W, H, opacity_bias=ob)
    assert (np.abs(imgb.cpu().numpy() - refb[0]) > 1e-5 + 1e-4 * np.abs(refb[0])).mean() < 1e-3
    r6 = _C.alpha_blending_backward_with_bias(*args, _t(ob), *tail, fTb, ncb, _t(g))
    grb = o.alpha_blending_backward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, W, H, refb[1], refb[2], g, opacity_bias=ob)
    assert len(r6) == 6
    assert_grad(r6[4], grb[5], "dL_dopacity_bias")
    assert_grad(r6[0], grb[0], "dL_duv (bias)")
