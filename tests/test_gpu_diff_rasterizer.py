"""GaussianRasterizer adapter (call site: reference src/pointrix/renderer/base_splatting.py:123-174)
against the explicit perspective operator chain, which tests/test_gpu_parity.py pins to the oracle.  The third-party
package the reference imports is absent from its tree: parity with that binary is unpinned, these tests pin the adapter
to the operators and to the call site's conventions (transposed matrices, means2D gradient tap, per-channel bg)."""
import math

import numpy as np
import pytest
import torch

import dptr.gs as gs
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from splatter_a_video_amd.diff_rasterizer import _camera as adapter_camera
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a, dtype=np.float32), device="cuda", requires_grad=grad)


def _camera(sc, px_shift=0.0, znear=0.01, zfar=100.0):
    """matrices in the convention of the reference's camera (camera.py:110-118): transposed world->view, and
    transposed (view then projection) product; returns the settings fields and the equivalent intr / extr"""
    W, H = sc.W, sc.H
    fx, fy = float(sc.intr[0]), float(sc.intr[1])
    tanx, tany = W / (2.0 * fx), H / (2.0 * fy)
    w2c = np.eye(4, dtype=np.float64); w2c[:3, :4] = sc.extr[:3, :4]
    P = np.zeros((4, 4)); P[0, 0] = 1.0 / tanx; P[1, 1] = 1.0 / tany; P[0, 2] = px_shift
    P[3, 2] = 1.0; P[2, 2] = zfar / (zfar - znear); P[2, 3] = -(zfar * znear) / (zfar - znear)
    view_t = w2c.T
    full_t = view_t @ P.T
    campos = np.linalg.inv(view_t)[3, :3]
    intr = np.array([fx, fy, (1.0 + px_shift) * W / 2.0, H / 2.0], np.float32)
    return dict(tanfovx=tanx, tanfovy=tany, viewmatrix=_t(view_t), projmatrix=_t(full_t), campos=_t(campos)), _t(intr), _t(sc.extr)


def _settings(sc, cam, bg, deg=3, scale_modifier=1.0):
    return GaussianRasterizationSettings(image_height=sc.H, image_width=sc.W, bg=bg, scale_modifier=scale_modifier,
                                         sh_degree=deg, prefiltered=False, debug=False, **cam)


def _chain(sc, p, intr, extr, campos, bg_scalar, deg=3, rgb=None, scale_modifier=1.0):
    W, H = sc.W, sc.H
    uv, depth = gs.project_point(p["xyz"], intr, extr, W, H)
    vis = depth != 0
    cov = gs.compute_cov3d(p["scale"] * scale_modifier, p["rotate"], vis)
    conic, radius, tiles = gs.ewa_project(p["xyz"], cov, intr, extr, uv, W, H, vis)
    if rgb is None:
        dirs = torch.nn.functional.normalize(p["xyz"] - campos[None, :], dim=1)
        rgb = gs.compute_sh(p["shs"], deg, dirs, vis)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    ndc = torch.zeros_like(uv, requires_grad=True)
    img = gs.alpha_blending(uv, conic, p["opacity"], rgb, idx, tr, bg_scalar, W, H, ndc)
    return img, radius, ndc, (uv, conic, idx, tr, cov)


def _leaves(sc):
    return {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity, shs=sc.shs).items()}


def test_adapter_equals_operator_chain_sh():
    sc = make_scene(6000, 192, 128, seed=3, ortho=False)
    cam, intr, extr = _camera(sc)
    g = _t(np.random.default_rng(0).normal(size=(3, sc.H, sc.W)))
    # the camera the adapter derives from the transposed matrices is the one they were built from; the chain below
    # then uses the adapter's float32 values (a last-bit difference in fx moves alpha >= 1/255 decisions of single pixels)
    settings = _settings(sc, cam, torch.full((3,), 0.25, device="cuda"))
    intr_a, extr_a = adapter_camera(settings, torch.device("cuda"))
    assert torch.allclose(intr_a, intr, rtol=2e-6) and torch.allclose(extr_a, extr[:3], atol=1e-6)
    intr, extr = intr_a, extr_a
    # chain
    p = _leaves(sc)
    img_c, radius_c, ndc, _ = _chain(sc, p, intr, extr, cam["campos"], 0.25)
    (img_c * g).sum().backward()
    # adapter
    q = _leaves(sc)
    means2D = torch.zeros(sc.N, 3, device="cuda", requires_grad=True)
    rast = GaussianRasterizer(raster_settings=settings)
    img_a, radii = rast(means3D=q["xyz"], means2D=means2D, shs=q["shs"], colors_precomp=None, opacities=q["opacity"],
                        scales=q["scale"], rotations=q["rotate"], cov3D_precomp=None)
    (img_a * g).sum().backward()
    assert img_a.shape == (3, sc.H, sc.W) and radii.dtype == torch.int32
    assert torch.equal(radii, radius_c) and int((radii > 0).sum()) > sc.N // 2
    assert torch.allclose(img_a, img_c, rtol=1e-5, atol=2e-6)
    # (the adapter composites c - bg and adds bg back: same value, different rounding of the per-pixel sums)
    err = float((means2D.grad[:, :2] - ndc.grad).abs().max()) / float(ndc.grad.abs().max())
    assert err < 2e-5, err
    assert float(means2D.grad[:, 2].abs().max()) == 0.0
    for k in p:
        x, y = q[k].grad, p[k].grad
        err = float((x - y).abs().max()) / float(y.abs().max())
        assert err < 5e-5, (k, err)
    assert torch.equal(rast.markVisible(q["xyz"].detach()), gs.project_point(q["xyz"].detach(), intr, extr, sc.W, sc.H)[1][:, 0] != 0)


def test_adapter_background_per_channel():
    """out_c = sum_k w_k c_k + T bg_c with a different bg per channel == scalar-bg renders recombined"""
    sc = make_scene(4000, 160, 96, seed=8, ortho=False)
    cam, intr, extr = _camera(sc)
    bg = torch.tensor([0.2, 0.5, 0.9], device="cuda")
    q = _leaves(sc)
    rast = GaussianRasterizer(_settings(sc, cam, bg))
    img, _ = rast(q["xyz"], None, q["opacity"], shs=q["shs"], scales=q["scale"], rotations=q["rotate"])
    with torch.no_grad():
        img0, _, _, (uv, conic, idx, tr, _) = _chain(sc, q, intr, extr, cam["campos"], 0.0)
        acc = gs.alpha_blending(uv, conic, q["opacity"], torch.ones(sc.N, 1, device="cuda"), idx, tr, 0.0, sc.W, sc.H)
    want = img0 + (1.0 - acc) * bg[:, None, None]
    assert torch.allclose(img, want, rtol=1e-5, atol=3e-6)
    # the gradient of the background term reaches the opacities: d/d opacity of (img . 1) differs from the bg-free one
    img.sum().backward()
    assert torch.isfinite(q["opacity"].grad).all() and float(q["opacity"].grad.abs().max()) > 0


def test_adapter_precomputed_inputs_and_principal_point():
    sc = make_scene(3000, 160, 96, seed=11, ortho=False)
    cam, intr, extr = _camera(sc, px_shift=0.1)          # cx = 1.1 * W/2 recovered from the projection matrix
    rng = np.random.default_rng(4)
    colors = _t(rng.uniform(0, 1, size=(sc.N, 3)))
    p = _leaves(sc)
    with torch.no_grad():
        img_c, radius_c, _, (_, _, _, _, cov) = _chain(sc, p, intr, extr, cam["campos"], 0.0, rgb=colors, scale_modifier=1.5)
        rast = GaussianRasterizer(_settings(sc, cam, torch.zeros(3, device="cuda"), scale_modifier=1.5))
        img_a, radii = rast(p["xyz"], None, p["opacity"], colors_precomp=colors, scales=p["scale"], rotations=p["rotate"])
        img_b, radii_b = rast(p["xyz"], None, p["opacity"], colors_precomp=colors, cov3D_precomp=cov)
    assert torch.equal(radii, radius_c) and torch.equal(radii_b, radius_c)
    assert torch.allclose(img_a, img_c, rtol=1e-5, atol=2e-6) and torch.allclose(img_b, img_c, rtol=1e-5, atol=2e-6)


def test_adapter_argument_errors():
    sc = make_scene(64, 64, 64, seed=1, ortho=False)
    cam, _, _ = _camera(sc)
    rast = GaussianRasterizer(_settings(sc, cam, torch.zeros(3, device="cuda")))
    p = _leaves(sc)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(p["xyz"], None, p["opacity"], scales=p["scale"], rotations=p["rotate"])
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(p["xyz"], None, p["opacity"], shs=p["shs"], colors_precomp=p["xyz"], scales=p["scale"], rotations=p["rotate"])
    with pytest.raises(Exception, match="3D covariance"):
        rast(p["xyz"], None, p["opacity"], shs=p["shs"])
    with pytest.raises(Exception, match="3D covariance"):
        rast(p["xyz"], None, p["opacity"], shs=p["shs"], scales=p["scale"], rotations=p["rotate"],
             cov3D_precomp=torch.zeros(sc.N, 6, device="cuda"))


def test_adapter_against_the_cpu_oracle(oracle_mod):
    """What ``GaussianRasterizer`` must return at the reference's call site (base_splatting.py:123-174), stated with the CPU
    oracle and nothing of this package: SH colours for the directions from ``campos`` (+0.5, clamp), perspective
    projection / cov3d / EWA, (tile, depth) order, compositing with ONE BACKGROUND PER CHANNEL (one scalar-background oracle
    blend per channel), radii, and the gradients of every input incl. the ``means2D`` screen-space tap."""
    from test_gpu_parity import assert_grad
    o = oracle_mod
    sc = make_scene(2500, 112, 80, seed=14, ortho=False)
    cam, intr, extr = _camera(sc)
    bg = np.array([0.15, 0.4, 0.8], np.float32)
    rng = np.random.default_rng(2)
    g = rng.normal(size=(3, sc.H, sc.W)).astype(np.float32)
    q = _leaves(sc)
    means2D = torch.zeros(sc.N, 3, device="cuda", requires_grad=True)
    settings = _settings(sc, cam, _t(bg))
    img, radii = GaussianRasterizer(settings)(means3D=q["xyz"], means2D=means2D, shs=q["shs"], opacities=q["opacity"],
                                              scales=q["scale"], rotations=q["rotate"])
    (img * _t(g)).sum().backward()
    # ---- oracle, with the float32 camera the adapter derives from the transposed matrices
    intr_a, extr_a = adapter_camera(settings, torch.device("cuda"))
    intr_n, extr_n = intr_a.cpu().numpy(), np.vstack([extr_a.cpu().numpy(), [[0, 0, 0, 1]]]).astype(np.float32)
    W, H = sc.W, sc.H
    campos = cam["campos"].cpu().numpy()
    uv, depth = o.project_point_forward(sc.xyz, intr_n, extr_n, W, H, 0.2)
    vis = depth.reshape(-1) != 0
    cov = o.compute_cov3d_forward(sc.scale, sc.rotate, vis)
    conic, radius, tiles = o.ewa_project_forward(sc.xyz, cov, intr_n, extr_n, uv, W, H, vis)
    d = sc.xyz - campos[None, :]
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    dirs = (d / nrm).astype(np.float32)
    rgb, clamped = o.compute_sh_forward(sc.shs, 3, dirs, vis)
    idx, tr = o.sort_gaussian(uv, depth, W, H, radius, tiles)
    assert (radii.cpu().numpy() != radius).mean() < 1e-3
    out = np.zeros((3, H, W), np.float32)
    duv = np.zeros((sc.N, 2)); dcon = np.zeros((sc.N, 3)); dop = np.zeros((sc.N, 1)); drgb = np.zeros((sc.N, 3))
    for c in range(3):
        r = o.alpha_blending_forward(uv, conic, sc.opacity, rgb[:, c:c + 1], idx, tr, float(bg[c]), W, H)
        out[c] = r[0][0]
        gr = o.alpha_blending_backward(uv, conic, sc.opacity, rgb[:, c:c + 1], idx, tr, float(bg[c]), W, H, r[1], r[2], g[c:c + 1])
        duv += gr[0]; dcon += gr[1]; dop += gr[2].reshape(-1, 1); drgb[:, c] = gr[3][:, 0]
    bad = np.abs(img.detach().cpu().numpy() - out) > 1e-5 + 1e-4 * np.abs(out)
    assert bad.mean() < 1e-3
    half = np.array([[0.5 * W, 0.5 * H]])
    assert_grad(means2D.grad[:, :2], (duv * half).astype(np.float32), "means2D tap")
    assert_grad(q["opacity"].grad, dop.astype(np.float32), "opacity")
    dxyz_e, dcov, _, _ = o.ewa_project_backward(sc.xyz, cov, intr_n, extr_n, radius, dcon.astype(np.float32), W, H, need_intr=False,
                                                need_extr=False)
    dxyz_p, _, _ = o.project_point_backward(sc.xyz, intr_n, extr_n, W, H, uv, depth, duv.astype(np.float32),
                                            np.zeros((sc.N, 1), np.float32), need_intr=False, need_extr=False)
    dscale, dquat = o.compute_cov3d_backward(sc.scale, sc.rotate, vis, dcov)
    dshs, ddirs = o.compute_sh_backward(sc.shs, 3, dirs, vis, clamped, drgb.astype(np.float32))
    # the direction depends on the position: d dirs / d xyz = (I - dir dir^T) / |d|
    dxyz_d = (ddirs - dirs * (ddirs * dirs).sum(1, keepdims=True)) / nrm
    assert_grad(q["scale"].grad, dscale, "scale")
    assert_grad(q["rotate"].grad, dquat, "rotate")
    assert_grad(q["shs"].grad, dshs, "shs")
    assert_grad(q["xyz"].grad, (dxyz_e + dxyz_p + dxyz_d).astype(np.float32), "xyz")
