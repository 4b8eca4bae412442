"""The frame-batch entry points -- the path bench.py times -- at BASELINE.json's FULL sizes (missing #3):
configs[1] (300k Gaussians, 854x480, 3 channels), configs[3] (1M Gaussians, 1280x720; pair buffers sized so that byte AND
float offsets of the last frame's records lie beyond 2^32 / 2^31) and configs[4] (32 feature channels) through
``FrameBatch.render``, and the reference's training frame (dynamic Gaussians through render_iter's three blends,
dptr_ortho_enhanced.py:331-376) through ``FrameBatch.render_dynamic_sets`` at configs[1].

No CPU oracle runs at these sizes (it needs minutes per frame); the checks are the size-independent properties of the
domain plus equality with the per-frame operators, which tests/test_gpu_parity.py compares with the oracle:
  * replay transmittance: the backward arrives at T = 1 in front of every pixel's first splat (capture_T_front) -- it
    replayed exactly the forward's inclusion decisions;
  * sum of the weights + final transmittance = 1 (a constant-1 feature over bg = 1 renders 1);
  * the backward is linear in dL_dout;
  * frame f of the batch == the per-frame operators on frame f: images bit for bit, gradients to summation order;
  * FrameBatch.check() clean (no frame outgrew the capacity).
"""
import numpy as np
import pytest
import torch

import dptr.gs as gs
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.gs.raster_ops import capture_T_front
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _offsets(sc, F):
    return np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32)


def _close(a, b, what, frac=2e-5):
    """same arithmetic, other summation order: element-wise 2e-4 relative + 2e-6 of the maximum on all but a 2e-5 fraction
    of the elements (ill-conditioned, nearly isotropic Gaussians; the batch's lists hold only the pairs that reach their tile, the
    per-frame operators' the whole bounding squares: other super-batch boundaries, other partial sums), none further off than
    10x that"""
    d = (a - b).abs()
    mx = float(b.abs().max())
    bad = d > 2e-4 * b.abs() + 2e-6 * mx + 1e-12
    assert int(bad.sum()) <= max(2, int(a.numel() * frac)), (what, int(bad.sum()), float(d.max()), mx)
    assert bool((d <= 2e-3 * b.abs() + 2e-5 * mx + 1e-12).all()), (what, float(d.max()), mx)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,N,W,H,C,F,capacity", [
    ("c2", 300000, 854, 480, 3, 8, None),
    # capacity 36 M pairs per frame: frame 5's records start at float 5 * 36e6 * 12 = 2.16e9 > 2^31 (byte 8.6e9 > 2^32)
    ("c4", 1000000, 1280, 720, 3, 6, 36_000_000),
    ("c5", 300000, 854, 480, 32, 4, None),
])
def test_frame_batch_render_full_size(name, N, W, H, C, F, capacity):
    sc = make_scene(N, W, H, F=50, C=(C if C > 3 else 0), seed=1234)
    rng = np.random.default_rng(7)
    off = _t(_offsets(sc, F))
    featv = sc.feature if C > 3 else rng.uniform(size=(N, C)).astype(np.float32)
    extr = _t(sc.extr)
    gen = torch.Generator(device="cpu").manual_seed(99)
    g1 = torch.randn(F, C, H, W, generator=gen).cuda()
    g2 = torch.randn(F, C, H, W, generator=gen).cuda()
    bg = 0.25

    def params():
        d = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}
        d["feature"] = _t(featv, True)
        return d

    B = FrameBatch(F, N, W, H, C, "cuda", capacity=capacity)
    if capacity is not None:
        stride = B.ncp
        assert (F - 1) * capacity * stride > 2 ** 31 and F * capacity * stride * 4 > 2 ** 32

    def batch_backward(g):
        p = params()
        out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, extr, bg=bg)
        with capture_T_front() as cap:
            out.backward(g)
        torch.cuda.synchronize()
        return out.detach(), {k: v.grad for k, v in p.items()}, cap.maps[0], B.tap.clone(), B.radii_max.clone()

    out, gb1, Tf, tap1, rad = batch_backward(g1)
    M = B.check()
    assert M > 2 * N and (capacity is None or M <= capacity)      # (reach masks: 2.5 pairs per Gaussian, 3.8 in the bounding squares)
    # ---- the backward replayed the forward's decisions, on every pixel of every frame
    assert Tf.shape[0] == F * H and torch.isfinite(Tf).all()
    assert float((Tf - 1).abs().max()) < 2e-4
    for k, v in gb1.items():
        assert torch.isfinite(v).all(), k

    # ---- weights + final transmittance = 1
    with torch.no_grad():
        ones = torch.ones(N, C, device="cuda")
        o1 = B.render(_t(sc.xyz), _t(sc.scale), _t(sc.rotate), _t(sc.opacity), ones, off, extr, bg=1.0)
        assert float((o1 - 1).abs().max()) < 2e-5
        assert float(B.final_T.min()) >= 0.0 and float(B.final_T.max()) <= 1.0
    B.check()

    # ---- linearity of the backward in dL_dout
    _, gb2, _, tap2, _ = batch_backward(g2)
    _, gb12, _, tap12, _ = batch_backward(g1 + 2.0 * g2)
    for k in gb1:
        want = gb1[k] + 2.0 * gb2[k]
        scale = float(gb1[k].abs().max() + 2.0 * gb2[k].abs().max())
        assert float((gb12[k] - want).abs().max()) <= 2e-4 * scale, k
    assert float((tap12 - (tap1 + 2.0 * tap2)).abs().max()) <= 2e-4 * float(tap1.abs().max() + 2 * tap2.abs().max())

    # ---- frame f of the batch == the per-frame operators on frame f
    p = params()
    tap_ref = torch.zeros(N, 2, device="cuda")
    rad_ref = torch.zeros(N, dtype=torch.int32, device="cuda")
    for f in range(F):
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(p["xyz"], p["scales"], p["uquats"], extr, W, H, nearest=0.01,
                                                              offset=off[f])
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        ndc = torch.zeros_like(uv, requires_grad=True)
        img = gs.alpha_blending(uv, conic, p["opacity"], p["feature"], idx, tr, bg, W, H, ndc)
        assert torch.equal(img, out[f]), (name, f)                 # same kernels, same arithmetic: bit-identical images
        img.backward(g1[f])
        tap_ref += ndc.grad
        rad_ref = torch.maximum(rad_ref, radius)
        del uv, depth, conic, radius, tiles, idx, tr, img, ndc
    for k in gb1:
        _close(gb1[k], p[k].grad, f"{name}:{k}")
    _close(tap1, tap_ref, f"{name}:tap")
    assert torch.equal(rad, rad_ref)


@pytest.mark.timeout(900)
def test_render_dynamic_sets_full_size_c2():
    """the reference's real training frame at configs[1]: 300k dynamic Gaussians (spline position, time-varying rotation)
    through rgb (enhanced, K = 20, taps + abs taps) + depth (bg = 1) + 19 attribute channels (opacity detached), F = 4"""
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, FrameClock, frame_preprocess, to_segment_major
    N, W, H, T, K = 300000, 854, 480, 50, 20
    times = [0, 7, 24, 49]
    F = len(times)
    sc = make_scene(N, W, H, F=T, seed=1234)
    rng = np.random.default_rng(12)
    clock = FrameClock(T)
    I = clock.interval_num
    cub = to_segment_major(torch.as_tensor((0.002 * rng.normal(size=(N, 4 * I * 3))).astype(np.float32)), I).numpy()
    op = np.clip(sc.opacity, 1e-4, 1 - 1e-4)
    base = dict(position=sc.xyz, pos_cubic_node=cub, rotation=sc.rotate, opacity=np.log(op / (1 - op)).astype(np.float32),
                scaling=np.log(sc.scale).astype(np.float32), rgb=rng.uniform(size=(N, 3)).astype(np.float32),
                attrs=rng.uniform(-1, 1, size=(N, 19)).astype(np.float32))
    rot_poly = _t((0.01 * rng.normal(size=(N, 4, 4))).astype(np.float32))
    rot_four = _t((0.01 * rng.normal(size=(N, 8, 4))).astype(np.float32))
    gen = torch.Generator(device="cpu").manual_seed(5)
    g_rgb, g_dep, g_att = (torch.randn(F, c, H, W, generator=gen).cuda() for c in (3, 1, 19))
    extr = _t(sc.extr)

    pb = {k: _t(v, True) for k, v in base.items()}
    B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
    sets = [dict(feature=pb["rgb"], bg=0.2, taps=True), dict(feature="depth", bg=1.0),
            dict(feature=pb["attrs"], bg=0.0, detach_opacity=True)]
    o_rgb, o_dep, o_att, ids = B.render_dynamic_sets(
        clock, times, extr, sets, position=pb["position"], pos_cubic_node=pb["pos_cubic_node"], rotation=pb["rotation"],
        rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pb["opacity"], scaling=pb["scaling"],
        cubic_layout=SEGMENT_MAJOR, K=K)
    assert ids.shape == (F, H, W, K)
    with capture_T_front() as cap:
        torch.autograd.backward([o_rgb, o_dep, o_att], [g_rgb, g_dep, g_att])
    torch.cuda.synchronize()
    assert B.check() > 2 * N
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    # the first K contributors of a pixel are distinct, in list order, and end with -1 padding only
    valid = ids >= 0
    assert bool((valid[..., 1:] <= valid[..., :-1]).all())
    assert int(valid[..., 0].sum()) == int((B.ncontrib > 0).sum())

    # ---- per-frame operators through autograd (dynamics.frame_preprocess -> sort -> the three blends)
    pa = {k: _t(v, True) for k, v in base.items()}
    tap_ref, atap_ref = 0, 0
    for f, t in enumerate(times):
        uv, depth, conic, radius, tiles, opa = frame_preprocess(
            clock, t, extr, W, H, position=pa["position"], pos_cubic_node=pa["pos_cubic_node"], rotation=pa["rotation"],
            rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pa["opacity"], scaling=pa["scaling"], nearest=0.01,
            cubic_layout=SEGMENT_MAJOR)
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        ndc = torch.zeros_like(uv, requires_grad=True)
        andc = torch.zeros_like(uv, requires_grad=True)
        rgb_i, _, ids_i = gs.alpha_blending_enhanced(uv, conic, opa, pa["rgb"], idx, tr, 0.2, W, H, ndc, andc, K=K)
        dep_i = gs.alpha_blending(uv, conic, opa, depth, idx, tr, 1.0, W, H, ndc.detach())
        att_i = gs.alpha_blending(uv, conic, opa.detach(), pa["attrs"], idx, tr, 0.0, W, H, ndc.detach())
        # the batched preprocess contracts its FMAs differently from the per-frame one: last-bit geometry, and once in a while a
        # splat on the alpha = 1/255 threshold of a pixel is applied on one side only (<= 4e-3 there)
        for nm, got, want in (("rgb", o_rgb[f], rgb_i), ("depth", o_dep[f], dep_i), ("attrs", o_att[f], att_i)):
            d = (got - want).detach().abs()
            off_ = d > 1e-5 + 1e-4 * want.detach().abs()
            assert int(off_.any(0).sum()) <= 40 and float(d.max()) < 5e-3, (f, nm, int(off_.any(0).sum()), float(d.max()))
        assert int((ids[f] != ids_i).any(-1).sum()) <= 40
        torch.autograd.backward([rgb_i, dep_i, att_i], [g_rgb[f], g_dep[f], g_att[f]])
        tap_ref = tap_ref + ndc.grad
        atap_ref = atap_ref + andc.grad
    for k in pa:
        a, b = pb[k].grad, pa[k].grad
        bad = (a - b).abs() > 1e-3 * b.abs() + 1e-5 * float(b.abs().max())
        assert float(bad.float().mean()) < 2e-4, (k, int(bad.sum()), float((a - b).abs().max()))
    for nm, a, b in (("tap", B.tap, tap_ref), ("abs_tap", B.abs_tap, atap_ref)):
        bad = (a - b).abs() > 1e-3 * b.abs() + 1e-5 * float(b.abs().max())
        assert float(bad.float().mean()) < 2e-4, (nm, int(bad.sum()))
