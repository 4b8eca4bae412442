"""Parity of the HIP path (through the dptr.gs operator surface -> ctypes -> C ABI ->
libsplat_hip.so) against the CPU oracle on identical seeded inputs.

Tolerances (BASELINE.md section 2 / SURVEY.md 8c): images atol 1e-5 + rtol 1e-4; gradients 2e-3
relative to the tensor's max magnitude (float atomics / reduction order); integer outputs equal
except threshold ties on <= 0.01 % of the elements; the sort output is bit-exact.
"""
import numpy as np
import pytest
import torch

from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu

IMG_ATOL, IMG_RTOL = 1e-5, 1e-4
GRAD_RTOL = 2e-3
INT_MISMATCH = 1e-4


def dev(a, device, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=device)
    return t if dtype is None else t.to(dtype)


def relmax(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def assert_grad(a, b, what, tol=GRAD_RTOL, atol_frac=1e-4, budget=1e-4, outlier=3.0):
    """element-wise: |a - b| <= tol * |b| + atol_frac * max|b| on all but a fraction ``budget`` of the elements (those
    may sit on a discrete decision of one pixel); no element further off than ``outlier`` x tol of the tensor's maximum."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == b.shape or a.size == b.size, (what, a.shape, b.shape)
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    if a.size == 0:
        return
    mx = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b)
    bad = err > tol * np.abs(b) + atol_frac * mx
    allowed = max(int(np.ceil(budget * a.size)), 2)   # small tensors: two elements (one pixel on a threshold touches a few splats)
    assert bad.sum() <= allowed, (f"{what}: {int(bad.sum())} of {a.size} elements off by more than rtol {tol:g} + "
                                  f"{atol_frac:g} * max (worst {err.max() / mx:.3e} of max)")
    assert err.max() / mx < outlier * tol, f"{what}: outlier {err.max() / mx:.3e} of the tensor maximum (bound {outlier:g} x {tol:g})"


def oracle_geometry(o, sc, f=0):
    xyz = sc.positions(f)
    if sc.ortho:
        uv, depth = o.project_point_ortho_forward(xyz, sc.extr, sc.W, sc.H, 0.01)
    else:
        uv, depth = o.project_point_forward(xyz, sc.intr, sc.extr, sc.W, sc.H, 0.2)
    vis = depth.reshape(-1) != 0
    cov = o.compute_cov3d_forward(sc.scale, sc.rotate, vis)
    conic, radius, tiles = o.ewa_project_forward(xyz, cov, sc.intr, sc.extr, uv, sc.W, sc.H, vis, ortho=sc.ortho)
    idx, tr = o.sort_gaussian(uv, depth, sc.W, sc.H, radius, tiles)
    return dict(xyz=xyz, uv=uv, depth=depth, vis=vis, cov=cov, conic=conic, radius=radius, tiles=tiles, idx=idx, tr=tr)


# ------------------------------------------------------------------ per-Gaussian ops
@pytest.mark.parametrize("ortho", [False, True])
@pytest.mark.parametrize("N,W,H", [(1, 16, 16), (777, 100, 60), (20000, 256, 256)])
def test_pointwise_ops(gpu, oracle_mod, ortho, N, W, H):
    import dptr.gs as gs
    o = oracle_mod
    sc = make_scene(N, W, H, seed=7 + N, ortho=ortho)
    rng = np.random.default_rng(N)
    xyz = sc.positions(1)
    xyz[: max(1, N // 40), 2] = 0.001        # near-culled
    # ---- project
    t_xyz = dev(xyz, gpu).requires_grad_(True)
    t_intr = dev(sc.intr, gpu).requires_grad_(True)
    t_extr = dev(sc.extr, gpu).requires_grad_(True)
    if ortho:
        uv_r, d_r = o.project_point_ortho_forward(xyz, sc.extr, W, H, 0.01)
        uv, d = gs.project_point_ortho(t_xyz, t_extr.detach(), W, H, nearest=0.01)
    else:
        uv_r, d_r = o.project_point_forward(xyz, sc.intr, sc.extr, W, H, 0.2)
        uv, d = gs.project_point(t_xyz, t_intr, t_extr, W, H)
    assert ((d.detach().cpu().numpy() != 0) == (d_r != 0)).all()
    np.testing.assert_allclose(uv.detach().cpu().numpy(), uv_r, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(d.detach().cpu().numpy(), d_r, rtol=1e-6, atol=1e-6)
    g_uv = rng.normal(size=(N, 2)).astype(np.float32); g_d = rng.normal(size=(N, 1)).astype(np.float32)
    (uv * dev(g_uv, gpu)).sum().add((d * dev(g_d, gpu)).sum()).backward()
    if ortho:
        dx_r = o.project_point_ortho_backward(sc.extr, W, H, d_r, g_uv, g_d)
    else:
        dx_r, di_r, de_r = o.project_point_backward(xyz, sc.intr, sc.extr, W, H, uv_r, d_r, g_uv, g_d)
        assert_grad(t_intr.grad, di_r, "dL_dintr")
        assert_grad(t_extr.grad[:3, :4], de_r, "dL_dextr")
    assert_grad(t_xyz.grad, dx_r, "dL_dxyz", 1e-4)

    # ---- cov3d
    vis = d_r.reshape(-1) != 0
    t_s = dev(sc.scale, gpu).requires_grad_(True); t_q = dev(sc.rotate, gpu).requires_grad_(True)
    cov = gs.compute_cov3d(t_s, t_q, dev(vis, gpu).reshape(-1, 1))
    cov_r = o.compute_cov3d_forward(sc.scale, sc.rotate, vis)
    np.testing.assert_allclose(cov.detach().cpu().numpy(), cov_r, rtol=1e-5, atol=1e-6 * float(np.abs(cov_r).max()))
    g_c = rng.normal(size=(N, 6)).astype(np.float32)
    (cov * dev(g_c, gpu)).sum().backward()
    ds_r, dq_r = o.compute_cov3d_backward(sc.scale, sc.rotate, vis, g_c)
    assert_grad(t_s.grad, ds_r, "dL_dscale", 1e-4); assert_grad(t_q.grad, dq_r, "dL_dquat", 1e-4)

    # ---- ewa (same inputs as the oracle: its uv / cov3d)
    t_xyz2 = dev(xyz, gpu).requires_grad_(True); t_cov = dev(cov_r, gpu).requires_grad_(True)
    t_intr2 = dev(sc.intr, gpu).requires_grad_(True); t_extr2 = dev(sc.extr, gpu).requires_grad_(True)
    conic_r, rad_r, tiles_r = o.ewa_project_forward(xyz, cov_r, sc.intr, sc.extr, uv_r, W, H, vis, ortho=ortho)
    if ortho:
        conic, rad, tiles = gs.ewa_project_ortho(t_xyz2, t_cov, t_extr2.detach(), dev(uv_r, gpu), W, H, dev(vis, gpu))
    else:
        conic, rad, tiles = gs.ewa_project(t_xyz2, t_cov, t_intr2, t_extr2, dev(uv_r, gpu), W, H, dev(vis, gpu))
    assert (rad.cpu().numpy() != rad_r).mean() <= INT_MISMATCH
    assert (tiles.cpu().numpy() != tiles_r).mean() <= INT_MISMATCH
    same = rad.cpu().numpy() == rad_r
    np.testing.assert_allclose(conic.detach().cpu().numpy()[same], conic_r[same], rtol=2e-4, atol=1e-7)
    g_k = rng.normal(size=(N, 3)).astype(np.float32)
    (conic * dev(g_k, gpu)).sum().backward()
    dxe_r, dcov_r, dei_r, dee_r = o.ewa_project_backward(xyz, cov_r, sc.intr, sc.extr, rad_r, g_k, W, H, ortho=ortho)
    if same.all():
        assert_grad(t_cov.grad, dcov_r, "dL_dcov3d", 5e-4)
        if not ortho:
            assert_grad(t_xyz2.grad, dxe_r, "ewa dL_dxyz", 5e-4)
            assert_grad(t_intr2.grad[:2], dei_r[:2], "ewa dL_dintr")
            assert_grad(t_extr2.grad[:3, :4], dee_r, "ewa dL_dextr")


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("free", [False, True])
def test_compute_sh(gpu, oracle_mod, deg, free):
    import dptr.gs as gs
    o = oracle_mod
    rng = np.random.default_rng(100 + deg)
    P, nb = 5000, (deg + 1) ** 2
    shs = rng.normal(0, 0.5, size=(P, nb, 3)).astype(np.float32)
    dirs = rng.normal(size=(P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs[: P // 2] = np.array([0, 0, 1], np.float32)
    vis = rng.uniform(size=P) > 0.1
    g = rng.normal(size=(P, 3)).astype(np.float32)
    t_sh = dev(shs, gpu).requires_grad_(True); t_d = dev(dirs, gpu).requires_grad_(True)
    fn = gs.compute_sh_free if free else gs.compute_sh
    c = fn(t_sh, deg, t_d, dev(vis, gpu))
    if free:
        c_r = o.compute_sh_forward(shs, deg, dirs, vis, free=True); cl = None
    else:
        c_r, cl = o.compute_sh_forward(shs, deg, dirs, vis)
    np.testing.assert_allclose(c.detach().cpu().numpy(), c_r, rtol=1e-5, atol=2e-6)
    (c * dev(g, gpu)).sum().backward()
    dsh_r, dd_r = o.compute_sh_backward(shs, deg, dirs, vis, cl, g, free=free)
    # clamp decisions can flip for values within rounding of 0: compare where the oracle's mask holds
    assert_grad(t_sh.grad, dsh_r, "dL_dshs", 1e-3)
    if deg > 0:
        assert_grad(t_d.grad, dd_r, "dL_ddirs", 1e-3)


# ------------------------------------------------------------------ sort
@pytest.mark.parametrize("N,W,H,sigma", [(1, 16, 16, 2.0), (500, 33, 17, 2.0), (5000, 256, 256, 2.0),
                                         (60000, 854, 480, 2.0), (6000, 64, 64, 12.0), (19000, 64, 64, 12.0),
                                         (2300, 48, 48, 12.0)])
def test_sort_gaussian_bit_exact(gpu, oracle_mod, N, W, H, sigma):
    """idx_sorted / tile_range identical to the oracle (stable order); sigma=12 px on 64x64 makes
    tiles with > 2048 pairs (2300: one block and a tail; 6000: 4096 < n < 8192; 19000: three merge levels above the
    2048-key register blocks), the crowded-tile path of tile_sort."""
    import dptr.gs as gs
    o = oracle_mod
    sc = make_scene(N, W, H, seed=N + 1)
    if sigma != 2.0:
        sc.scale *= sigma / 2.0
    G = oracle_geometry(o, sc)
    idx, tr = gs.sort_gaussian(dev(G["uv"], gpu), dev(G["depth"], gpu), W, H, dev(G["radius"], gpu), dev(G["tiles"], gpu))
    assert idx.dtype == torch.int32 and tr.dtype == torch.int32
    assert idx.shape[0] == G["idx"].shape[0]
    assert (tr.cpu().numpy() == G["tr"]).all()
    assert (idx.cpu().numpy() == G["idx"]).all()
    if sigma != 2.0:
        assert (G["tr"][:, 1] - G["tr"][:, 0]).max() > 2048
    # the pair map the sort produces alongside (prefix fused into the binning kernels): goff = inclusive cumsum of
    # the tiles each Gaussian touches (reference: sort_gaussian.py:42), slot_sorted = a permutation of the pair slots
    # whose owner is the Gaussian sorted to that position
    from splatter_a_video_amd.gs.raster_ops import _find_pairmap
    pm = _find_pairmap(idx, tr, N)
    goff, slot_sorted = pm.goff, pm.slot_sorted
    want = np.cumsum(G["tiles"].astype(np.int64))
    assert (goff.cpu().numpy() == want).all()
    slots = slot_sorted.cpu().numpy()
    assert (np.sort(slots) == np.arange(slots.size)).all()
    assert (np.searchsorted(want, slots, side="right") == G["idx"]).all()


def test_sort_gaussian_ties_and_empty(gpu, oracle_mod):
    import dptr.gs as gs
    o = oracle_mod
    # many bit-equal depths -> order must be ascending id inside a tile
    sc = make_scene(4000, 64, 64, seed=3)
    sc.xyz[:, 2] = np.float32(0.5)
    G = oracle_geometry(o, sc)
    idx, tr = gs.sort_gaussian(dev(G["uv"], gpu), dev(G["depth"], gpu), 64, 64, dev(G["radius"], gpu), dev(G["tiles"], gpu))
    assert (idx.cpu().numpy() == G["idx"]).all() and (tr.cpu().numpy() == G["tr"]).all()
    # nothing visible
    z = torch.zeros(10, 2, device=gpu)
    idx, tr = gs.sort_gaussian(z, torch.zeros(10, 1, device=gpu), 64, 64, torch.zeros(10, dtype=torch.int32, device=gpu),
                               torch.zeros(10, dtype=torch.int32, device=gpu))
    assert idx.numel() == 0 and (tr == 0).all()
    # P == 0
    idx, tr = gs.sort_gaussian(torch.zeros(0, 2, device=gpu), torch.zeros(0, 1, device=gpu), 64, 64,
                               torch.zeros(0, dtype=torch.int32, device=gpu), torch.zeros(0, dtype=torch.int32, device=gpu))
    assert idx.numel() == 0 and tr.shape == (16, 2) and (tr == 0).all()


# ------------------------------------------------------------------ alpha blending
def _blend_case(gpu, o, N, W, H, C, bg, variant, seed=0, K=4, mode="atomic", strict=True):
    """mode "atomic": idx_sorted comes from the oracle (plain tensor) -> wave-reduced atomics backward;
    mode "pair": idx_sorted comes from gs.sort_gaussian (bit-identical, carries the pair map) ->
    atomic-free backward."""
    import dptr.gs as gs
    sc = make_scene(N, W, H, seed=seed + N + C)
    G = oracle_geometry(o, sc)
    rng = np.random.default_rng(seed + 17)
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    ob = rng.uniform(-0.05, 0.1, size=(N, 1)).astype(np.float32) if variant == "bias" else None
    kw = {}
    if variant in ("enh", "trunc"):
        kw = dict(K=K, enable_truncation=variant == "trunc")
    res_r = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], bg, W, H, opacity_bias=ob, **kw)
    t = {k: dev(v, gpu).requires_grad_(True) for k, v in dict(uv=G["uv"], conic=G["conic"], opacity=sc.opacity, feat=feat).items()}
    t_idx, t_tr = dev(G["idx"], gpu), dev(G["tr"], gpu)
    if mode == "pair":
        t_idx, t_tr = gs.sort_gaussian(dev(G["uv"], gpu), dev(G["depth"], gpu), W, H, dev(G["radius"], gpu), dev(G["tiles"], gpu))
        assert (t_idx.cpu().numpy() == G["idx"]).all()
    ndc = torch.zeros(N, 2, device=gpu, requires_grad=True)
    andc = torch.zeros(N, 2, device=gpu, requires_grad=True)
    t_ob = dev(ob, gpu).requires_grad_(True) if ob is not None else None
    if variant == "bias":
        out = gs.alpha_blending_with_bias(t["uv"], t["conic"], t["opacity"], t["feat"], t_ob, t_idx, t_tr, bg, W, H, ndc, andc)
    elif variant in ("enh", "trunc"):
        out, nc, gi = gs.alpha_blending_enhanced(t["uv"], t["conic"], t["opacity"], t["feat"], t_idx, t_tr, bg, W, H, ndc, andc, **kw)
        assert (nc.cpu().numpy() != res_r[2]).mean() <= INT_MISMATCH
        assert (gi.cpu().numpy() != res_r[3]).mean() <= 2 * INT_MISMATCH
    else:
        out = gs.alpha_blending(t["uv"], t["conic"], t["opacity"], t["feat"], t_idx, t_tr, bg, W, H, ndc, andc)
    assert out.shape == (C, H, W)
    if strict:
        np.testing.assert_allclose(out.detach().cpu().numpy(), res_r[0], rtol=IMG_RTOL, atol=IMG_ATOL)
    else:   # a pixel may sit on a discrete decision (alpha = 1/255, T = 1e-4) that expf rounding flips
        bad = np.abs(out.detach().cpu().numpy() - res_r[0]) > (IMG_ATOL + IMG_RTOL * np.abs(res_r[0]))
        assert bad.mean() < 1e-3, bad.mean()
    g = rng.normal(size=(C, H, W)).astype(np.float32)
    (out * dev(g, gpu)).sum().backward()
    gr = o.alpha_blending_backward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], bg, W, H, res_r[1], res_r[2], g,
                                   opacity_bias=ob)
    assert_grad(t["uv"].grad, gr[0], "dL_duv")
    assert_grad(t["conic"].grad, gr[1], "dL_dconic")
    assert_grad(t["opacity"].grad, gr[2], "dL_dopacity")
    assert_grad(t["feat"].grad, gr[3], "dL_dfeature")
    half = np.array([[0.5 * W, 0.5 * H]], np.float32)
    assert_grad(ndc.grad, gr[0] * half, "dL_dndc")
    assert_grad(andc.grad, gr[4] * half, "dL_dabs_ndc")
    if ob is not None:
        assert_grad(t_ob.grad, gr[5], "dL_dopacity_bias")


@pytest.mark.parametrize("mode", ["atomic", "pair"])
@pytest.mark.parametrize("C", [1, 3, 5, 19, 32, 40])
def test_alpha_blending_channels(gpu, oracle_mod, C, mode):
    _blend_case(gpu, oracle_mod, 4000, 100, 60, C, 0.3, "plain", mode=mode)


@pytest.mark.parametrize("mode", ["atomic", "pair"])
@pytest.mark.parametrize("variant", ["plain", "enh", "trunc", "bias"])
@pytest.mark.parametrize("N,W,H", [(1, 16, 16), (300, 33, 17), (10000, 256, 256)])
def test_alpha_blending_variants(gpu, oracle_mod, variant, N, W, H, mode):
    _blend_case(gpu, oracle_mod, N, W, H, 3, 1.0 if variant == "plain" else 0.0, variant, seed=5, mode=mode)


@pytest.mark.parametrize("variant", ["plain", "enh", "trunc"])
@pytest.mark.parametrize("C", [16, 20, 24, 32])
def test_alpha_blending_wide_rows_on_the_matrix_pipe(gpu, oracle_mod, C, variant):
    """rows of 16 .. 32 channels: the forward's channel sums run as MFMAs (quarter-major lanes, four survivors per trip, DESIGN
    4i) -- every width of that path with the id lists and the truncation of the enhanced variants, on an image that is no
    multiple of the tile"""
    _blend_case(gpu, oracle_mod, 5000, 150, 90, C, 0.37, variant, seed=40 + C, K=7, mode="pair")


def test_alpha_blending_pair_mode_big_splats(gpu, oracle_mod):
    """sigma = 12 px: long per-Gaussian pair lists and > 4096-entry tiles through the pair-mode backward."""
    sc_kw = dict()
    import dptr.gs as gs
    o = oracle_mod
    sc = make_scene(3000, 64, 64, seed=77)
    sc.scale *= 6.0
    G = oracle_geometry(o, sc)
    feat = np.random.default_rng(5).uniform(size=(sc.N, 3)).astype(np.float32)
    out_r, fT_r, nc_r = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, 64, 64)
    idx, tr = gs.sort_gaussian(dev(G["uv"], gpu), dev(G["depth"], gpu), 64, 64, dev(G["radius"], gpu), dev(G["tiles"], gpu))
    t = {k: dev(v, gpu).requires_grad_(True) for k, v in dict(uv=G["uv"], conic=G["conic"], opacity=sc.opacity, feat=feat).items()}
    out = gs.alpha_blending(t["uv"], t["conic"], t["opacity"], t["feat"], idx, tr, 0.0, 64, 64)
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_r, rtol=IMG_RTOL, atol=IMG_ATOL)
    g = np.random.default_rng(6).normal(size=out_r.shape).astype(np.float32)
    out.backward(dev(g, gpu))
    gr = o.alpha_blending_backward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, 64, 64, fT_r, nc_r, g)
    assert_grad(t["uv"].grad, gr[0], "dL_duv"); assert_grad(t["conic"].grad, gr[1], "dL_dconic")
    assert_grad(t["opacity"].grad, gr[2], "dL_dopacity"); assert_grad(t["feat"].grad, gr[3], "dL_dfeature")


def test_alpha_blending_dense_saturating(gpu, oracle_mod):
    """High opacity + many layers: exercises the T < 1e-4 stop and the ncontrib replay."""
    import dptr.gs as gs
    o = oracle_mod
    sc = make_scene(20000, 64, 64, seed=99)
    sc.opacity[:] = 0.95
    G = oracle_geometry(o, sc)
    feat = np.random.default_rng(1).uniform(size=(sc.N, 3)).astype(np.float32)
    out_r, fT_r, nc_r = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, 64, 64)
    t_op = dev(sc.opacity, gpu).requires_grad_(True)
    out = gs.alpha_blending(dev(G["uv"], gpu), dev(G["conic"], gpu), t_op, dev(feat, gpu), dev(G["idx"], gpu), dev(G["tr"], gpu), 0.0, 64, 64)
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_r, rtol=IMG_RTOL, atol=IMG_ATOL)
    assert (fT_r < 1e-3).mean() > 0.5
    g = np.ones((3, 64, 64), np.float32)
    out.backward(dev(g, gpu))
    gr = o.alpha_blending_backward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.0, 64, 64, fT_r, nc_r, g)
    assert_grad(t_op.grad, gr[2], "dL_dopacity")


def test_alpha_blending_empty(gpu):
    import dptr.gs as gs
    W, H = 40, 24
    tr = torch.zeros(3 * 2, 2, dtype=torch.int32, device=gpu)
    out = gs.alpha_blending(torch.zeros(0, 2, device=gpu), torch.zeros(0, 3, device=gpu), torch.zeros(0, 1, device=gpu),
                            torch.zeros(0, 4, device=gpu), torch.zeros(0, dtype=torch.int32, device=gpu), tr, 0.25, W, H)
    assert out.shape == (4, H, W) and (out == 0.25).all()


# ------------------------------------------------------------------ whole chain through the operator surface
@pytest.mark.parametrize("N,W,H", [(10000, 256, 256)])
def test_rasterization_chain_matches_oracle(gpu, oracle_mod, N, W, H):
    """BASELINE config 1 (10k Gaussians, 256x256): gs.rasterization forward + backward vs the oracle chain."""
    import dptr.gs as gs
    o = oracle_mod
    sc = make_scene(N, W, H, seed=1234, ortho=False)
    rng = np.random.default_rng(0)
    feat = rng.uniform(size=(N, 3)).astype(np.float32)
    xyz = sc.positions(0)
    (out_r, fT_r, nc_r), saved = o.render_forward(xyz, sc.scale, sc.rotate, sc.opacity, feat, sc.intr, sc.extr, W, H, 0.0,
                                                  ortho=False)
    names = dict(xyz=xyz, scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity, feat=feat)
    t = {k: dev(v, gpu).requires_grad_(True) for k, v in names.items()}
    ndc = torch.zeros(N, 2, device=gpu, requires_grad=True)
    out = gs.rasterization(t["xyz"], t["scale"], t["rotate"], t["opacity"], t["feat"], dev(sc.intr, gpu), dev(sc.extr, gpu),
                           W, H, 0.0, ndc)
    # radius ties may move a handful of pairs: compare images with a small mismatch budget
    diff = np.abs(out.detach().cpu().numpy() - out_r) > (IMG_ATOL + IMG_RTOL * np.abs(out_r))
    assert diff.mean() < 1e-3
    g = rng.normal(size=out_r.shape).astype(np.float32)
    (out * dev(g, gpu)).sum().backward()
    gr = o.render_backward(xyz, sc.scale, sc.rotate, sc.opacity, sc.intr, sc.extr, W, H, 0.0, saved, g, ortho=False)
    assert_grad(t["xyz"].grad, gr["xyz"], "xyz", 5e-3)
    assert_grad(t["scale"].grad, gr["scale"], "scale", 5e-3)
    assert_grad(t["rotate"].grad, gr["rotate"], "rotate", 5e-3)
    assert_grad(t["opacity"].grad, gr["opacity"], "opacity", 5e-3)
    assert_grad(t["feat"].grad, gr["feature"], "feature", 5e-3)


def test_full_size_properties(gpu):
    """BASELINE config 2 size (300k Gaussians, 854x480): properties that need no CPU oracle run --
    weights + final transmittance sum to 1, per-tile depth order, every pair accounted for."""
    import dptr.gs as gs
    N, W, H = 300000, 854, 480
    sc = make_scene(N, W, H, seed=1234)
    xyz = dev(sc.positions(0), gpu); extr = dev(sc.extr, gpu)
    uv, depth = gs.project_point_ortho(xyz, extr, W, H, nearest=0.01)
    vis = depth != 0
    cov = gs.compute_cov3d(dev(sc.scale, gpu), dev(sc.rotate, gpu), vis)
    conic, radius, tiles = gs.ewa_project_ortho(xyz, cov, extr, uv, W, H, vis)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    M = int(tiles.sum().item())
    assert idx.numel() == M and M > 3 * N
    assert torch.equal(torch.bincount(idx.long(), minlength=N).int(), tiles)
    d = depth.reshape(-1)[idx.long()]
    seg = torch.repeat_interleave(torch.arange(tr.shape[0], device=gpu), (tr[:, 1] - tr[:, 0]).long())
    assert seg.numel() == M
    same_tile = seg[1:] == seg[:-1]
    assert bool((d[1:] >= d[:-1])[same_tile].all())
    ones = torch.ones(N, 1, device=gpu)
    op = dev(sc.opacity, gpu)
    out, nc, gi = gs.alpha_blending_enhanced(uv, conic, op, ones, idx, tr, 0.0, W, H, K=2)
    out2 = gs.alpha_blending(uv, conic, op, ones, idx, tr, 1.0, W, H)     # bg = 1 -> sum w + T = 1
    assert float((out2 - 1.0).abs().max()) < 1e-5
    assert float(out.max()) <= 1.0 + 1e-6 and int(nc.max()) > 0
    assert bool(((gi[..., 0] >= 0) == (nc > 0)).all())


# ------------------------------------------------------------------ full-size, size-independent properties (no CPU oracle run)
def _gpu_chain(gpu, sc, feat_np, bg=0.0, f=0):
    import dptr.gs as gs
    W, H = sc.W, sc.H
    xyz = dev(sc.positions(f), gpu); extr = dev(sc.extr, gpu)
    uv, depth = gs.project_point_ortho(xyz, extr, W, H, nearest=0.01)
    vis = depth != 0
    cov = gs.compute_cov3d(dev(sc.scale, gpu), dev(sc.rotate, gpu), vis)
    conic, radius, tiles = gs.ewa_project_ortho(xyz, cov, extr, uv, W, H, vis)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    return dict(uv=uv, conic=conic, idx=idx, tr=tr, opacity=dev(sc.opacity, gpu), radius=radius)


@pytest.mark.parametrize("N,W,H,C", [(300000, 854, 480, 32), (1000000, 1280, 720, 3)])
def test_full_size_linearity_and_weight_sum(gpu, N, W, H, C):
    """BASELINE configs 5 (32 channels) and 4 (1M Gaussians, 720p):
    (1) the forward is linear in the features (bg = 0);
    (2) with dL_dout = 1 on one channel, sum_i dL_dfeature[i,c] == sum_pixels (1 - T_final): the backward
        redistributes exactly the weights the forward applied;
    (3) dL_dfeature does not depend on the feature values."""
    import dptr.gs as gs
    sc = make_scene(N, W, H, seed=4321)
    G = _gpu_chain(gpu, sc, None)
    g = torch.Generator(device="cpu").manual_seed(1)
    f1 = torch.rand(N, C, generator=g).to(gpu).requires_grad_(True)
    f2 = torch.rand(N, C, generator=g).to(gpu).requires_grad_(True)
    args = (G["uv"], G["conic"], G["opacity"])
    o1 = gs.alpha_blending(*args, f1, G["idx"], G["tr"], 0.0, W, H)
    o2 = gs.alpha_blending(*args, f2, G["idx"], G["tr"], 0.0, W, H)
    o12 = gs.alpha_blending(*args, (f1 + 2.0 * f2).detach(), G["idx"], G["tr"], 0.0, W, H)
    assert float((o12 - (o1 + 2.0 * o2)).detach().abs().max()) < 2e-4
    ones = torch.ones(N, 1, device=gpu)
    oT = gs.alpha_blending(*args, ones, G["idx"], G["tr"], 0.0, W, H)           # = 1 - T_final
    gout = torch.zeros(C, H, W, device=gpu); gout[C // 2] = 1.0
    o1.backward(gout)
    o2.backward(gout)
    total_w = float(oT.double().sum())
    got = float(f1.grad[:, C // 2].double().sum())
    assert abs(got - total_w) < 1e-4 * total_w
    # (the 32-channel backward sums its four waves' contributions with LDS float atomics: order-dependent rounding)
    assert float((f1.grad - f2.grad).abs().max()) <= 1e-5 * float(f1.grad.abs().max())
    assert float(f1.grad[:, :C // 2].abs().max()) == 0.0                          # untouched channels stay 0


def test_extreme_splats_and_thresholds(gpu, oracle_mod):
    """Edge cases the reference's loop distinguishes: needle-like splats (conic condition number > 1e4),
    a splat covering the whole image, opacities at / below the 1/255 threshold, alpha saturating at 0.99."""
    import dptr.gs as gs
    o = oracle_mod
    W, H = 96, 64
    sc = make_scene(600, W, H, seed=8)
    sc.scale[:150] *= np.array([[40.0, 0.05, 1.0]], np.float32)       # needles
    sc.scale[150:153] *= 60.0                                          # huge splats: every tile, long lists
    sc.opacity[153:250] = np.float32(1.0 / 255.0)                      # exactly at the threshold
    sc.opacity[250:300] = np.float32(0.0039)                           # just below
    sc.opacity[300:350] = np.float32(1.0)                              # clamps to 0.99
    G = oracle_geometry(o, sc)
    feat = np.random.default_rng(3).uniform(size=(sc.N, 3)).astype(np.float32)
    out_r, fT_r, nc_r = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.5, W, H)
    idx, tr = gs.sort_gaussian(dev(G["uv"], gpu), dev(G["depth"], gpu), W, H, dev(G["radius"], gpu), dev(G["tiles"], gpu))
    assert (idx.cpu().numpy() == G["idx"]).all()
    t = {k: dev(v, gpu).requires_grad_(True) for k, v in dict(uv=G["uv"], conic=G["conic"], opacity=sc.opacity, feat=feat).items()}
    out = gs.alpha_blending(t["uv"], t["conic"], t["opacity"], t["feat"], idx, tr, 0.5, W, H)
    bad = np.abs(out.detach().cpu().numpy() - out_r) > (IMG_ATOL + IMG_RTOL * np.abs(out_r))
    assert bad.mean() < 1e-3, bad.mean()
    g = np.random.default_rng(4).normal(size=out_r.shape).astype(np.float32)
    out.backward(dev(g, gpu))
    gr = o.alpha_blending_backward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.5, W, H, fT_r, nc_r, g)
    assert_grad(t["uv"].grad, gr[0], "dL_duv", 5e-3); assert_grad(t["conic"].grad, gr[1], "dL_dconic", 5e-3)
    assert_grad(t["opacity"].grad, gr[2], "dL_dopacity", 5e-3); assert_grad(t["feat"].grad, gr[3], "dL_dfeature", 5e-3)


@pytest.mark.parametrize("W,H", [(1, 1), (16, 1), (17, 33)])
def test_tiny_images(gpu, oracle_mod, W, H):
    import dptr.gs as gs
    o = oracle_mod
    sc = make_scene(50, max(W, 8), max(H, 8), seed=2)
    sc.W, sc.H = W, H
    G = oracle_geometry(o, sc)
    feat = np.random.default_rng(5).uniform(size=(sc.N, 2)).astype(np.float32)
    out_r, fT_r, nc_r = o.alpha_blending_forward(G["uv"], G["conic"], sc.opacity, feat, G["idx"], G["tr"], 0.1, W, H)
    idx, tr = gs.sort_gaussian(dev(G["uv"], gpu), dev(G["depth"], gpu), W, H, dev(G["radius"], gpu), dev(G["tiles"], gpu))
    assert (idx.cpu().numpy() == G["idx"]).all() and (tr.cpu().numpy() == G["tr"]).all()
    out = gs.alpha_blending(dev(G["uv"], gpu), dev(G["conic"], gpu), dev(sc.opacity, gpu), dev(feat, gpu), idx, tr, 0.1, W, H)
    np.testing.assert_allclose(out.cpu().numpy(), out_r, rtol=IMG_RTOL, atol=IMG_ATOL)


@pytest.mark.parametrize("seed", range(10))
def test_alpha_blending_random_configurations(gpu, oracle_mod, seed):
    """seeded sweep over image shapes (not multiples of the tile), Gaussian counts, channel counts that land in every
    kernel instantiation, background values and variants -- pair mode (matrix-core backward where it applies)"""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(17, 230)), int(rng.integers(17, 150))
    N = int(rng.integers(50, 6000))
    C = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 19, 20, 21, 33]))
    variant = str(rng.choice(["plain", "plain", "enh", "trunc", "bias"]))
    bg = float(rng.choice([0.0, 1.0, 0.37]))
    _blend_case(gpu, oracle_mod, N, W, H, C, bg, variant, seed=seed, K=int(rng.integers(1, 12)), mode="pair", strict=False)


def test_full_size_c2_sh_forward_backward_properties(gpu):
    """BASELINE configs[1] (300k Gaussians, 854x480, SH degree 3 -> RGB), the whole operator chain forward + backward
    through autograd, checked through properties that need no CPU run:
    (1) the backward replays exactly the forward's splats (replay transmittance = 1 at every pixel);
    (2) the backward is linear in dL_dout;
    (3) with unit features and dL_dout = 1 the feature gradients sum to the image's coverage, sum_k w_k = 1 - T_final;
    (4) the SH gradients are the colour gradients times the basis (direction (0,0,1): C0 for l = 0, C1 for (1, 0));
    (5) Gaussians that touch no tile receive exactly zero gradient."""
    import dptr.gs as gs
    from splatter_a_video_amd.gs.raster_ops import capture_T_front
    N, W, H = 300000, 854, 480
    sc = make_scene(N, W, H, seed=1234)
    rng = np.random.default_rng(5)
    extr = dev(sc.extr, gpu)
    dirs = torch.zeros(N, 3, device=gpu); dirs[:, 2] = 1.0
    g1 = dev(rng.normal(size=(3, H, W)).astype(np.float32), gpu)
    g2 = dev(rng.normal(size=(3, H, W)).astype(np.float32), gpu)

    def run(gout):
        p = {k: dev(v, gpu).requires_grad_(True) for k, v in dict(xyz=sc.positions(1), scale=sc.scale, rotate=sc.rotate,
                                                                  opacity=sc.opacity, shs=sc.shs).items()}
        rgb = gs.compute_sh(p["shs"], 3, dirs)
        rgb.retain_grad()
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(p["xyz"], p["scale"], p["rotate"], extr, W, H, nearest=0.01)
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        img = gs.alpha_blending(uv, conic, p["opacity"], rgb, idx, tr, 0.0, W, H)
        with capture_T_front() as cap:
            (img * gout).sum().backward()
        return p, rgb, cap.maps[0], tiles, dict(uv=uv.detach(), conic=conic.detach(), idx=idx, tr=tr)

    pa, rgba, Tfront, tiles, geo = run(g1)
    assert float((Tfront - 1).abs().max()) < 2e-4                                    # (1)
    pb, _, _, _, _ = run(g2)
    pc, _, _, _, _ = run(2.0 * g1 - 0.5 * g2)
    for k in pa:                                                                     # (2)
        want = 2.0 * pa[k].grad - 0.5 * pb[k].grad
        scale = float(want.abs().max())
        assert float((pc[k].grad - want).abs().max()) < 2e-4 * scale + 1e-12, k
    ones = torch.ones(N, 1, device=gpu, requires_grad=True)                          # (3)
    cov = gs.alpha_blending(geo["uv"], geo["conic"], pa["opacity"].detach(), ones, geo["idx"], geo["tr"], 0.0, W, H)
    cov.sum().backward()
    assert abs(float(ones.grad.double().sum()) / float(cov.double().sum()) - 1.0) < 1e-5
    assert float(cov.max()) <= 1.0 + 1e-6
    C0, C1 = 0.28209479177387814, 0.4886025119029199                                 # (4)
    live = (rgba.detach() > 0).float()
    assert torch.allclose(pa["shs"].grad[:, 0], C0 * rgba.grad * live, rtol=1e-5, atol=1e-9)
    assert torch.allclose(pa["shs"].grad[:, 2], C1 * rgba.grad * live, rtol=1e-5, atol=1e-9)
    assert float(pa["shs"].grad[:, 1].abs().max()) == 0.0 and float(pa["shs"].grad[:, 3].abs().max()) == 0.0
    untouched = tiles == 0                                                           # (5)
    assert int(untouched.sum()) > 0
    for k in ("xyz", "scale", "rotate", "opacity"):
        assert float(pa[k].grad[untouched].abs().max()) == 0.0, k
