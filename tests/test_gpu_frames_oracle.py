"""The frame-batch entry points -- the path bench.py times -- straight against the CPU oracle and the reference-made golden
fixtures (no intermediate HIP path in between): FrameBatch.render, render_sets (K = 20, 3 + 1 + 19), render_dynamic and
render_dynamic_sets at BASELINE configs[0] size (10k Gaussians, 256 x 256, F = 3), and the 32-channel width of configs[4].
Kernels under test: frame_preprocess_fwd_batch / preprocess_ortho_forward_batch, the batched binning + sort, pack /
pack_sets, blend_fwd, blend_bwd_mfma, blend_bwd_sets, frames_gauss_bwd_static(_sets), frames_gauss_bwd_dynamic(_sets).
Semantics: src/pointrix/renderer/dptr_ortho_enhanced.py:331-376,385-433; src/alpha_blending.cu:112-249 (reference)."""
import os

import numpy as np
import pytest
import torch

import oracle_chain as oc
from splatter_a_video_amd import frames as FR
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.gs.raster_ops import capture_T_front
from test_gpu_parity import GRAD_RTOL, IMG_ATOL, IMG_RTOL, assert_grad

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(a, grad=False):
    return torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda", requires_grad=grad)


def _c1_scene(F, seed=0):
    """geometry of the reference-made c1 fixture (10k Gaussians, 256 x 256) + seeded opacities and per-frame offsets
    (frame 0 unmoved: its projection is the fixture's)"""
    g = dict(np.load(os.path.join(GOLD, "ortho_c1_10k_256x256.npz")))
    rng = np.random.default_rng(seed)
    N = g["xyz"].shape[0]
    opacity = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, size=(N, 1))))).astype(np.float32)
    off = np.zeros((F, N, 3), np.float32)
    for f in range(1, F):
        d = 0.05 * np.sin(2.0 * np.pi * f / 7.0 + rng.uniform(0, 2 * np.pi, size=N))
        off[f, :, 0] = d; off[f, :, 1] = -0.5 * d
    return g, opacity, off, rng


def _check_images(got, per, c0, name):
    for f, r in enumerate(per):
        c = c0
        for img in r["imgs"]:
            a = got[f, c:c + img.shape[0]].detach().cpu().numpy()
            bad = np.abs(a - img) > (IMG_ATOL + IMG_RTOL * np.abs(img))
            assert bad.mean() < 1e-3, (name, f, float(bad.mean()))
            c += img.shape[0]


def _same_geometry(B, per):
    rad = B.radius.cpu().numpy()
    same = all((rad[f] == r["radius"]).all() for f, r in enumerate(per))
    for f, r in enumerate(per):
        assert (rad[f] != r["radius"]).mean() <= 1e-4        # ceil(3 sqrt(lambda)) may flip on a rounding tie
    return same


@pytest.mark.parametrize("reach", [True, False], ids=["reach", "full_lists"])
def test_frame_batch_render_against_oracle_and_reference_geometry(oracle_mod, reach):
    """FrameBatch.render (the default bench path): images, final_T, ncontrib, every parameter gradient and the taps of three
    frames against the oracle chain; frame 0's batched preprocess against the REFERENCE's own uv / conic / radius.  With reach
    masks (the default; FR.OPTIONS["reach"]) the lists hold fewer pairs than the reference's -- none of the dropped ones reaches
    a pixel: pair counts and list positions (ncontrib) are compared on the full lists, everything else in both modes."""
    old_reach = FR.OPTIONS["reach"]
    FR.OPTIONS["reach"] = reach
    try:
        _frame_batch_render_against_oracle(oracle_mod, reach)
    finally:
        FR.OPTIONS["reach"] = old_reach


def _frame_batch_render_against_oracle(oracle_mod, reach):
    F, C, bg = 3, 3, 0.3
    g, opacity, off, rng = _c1_scene(F)
    N, W, H = g["xyz"].shape[0], int(g["W"]), int(g["H"])
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    gimg = rng.normal(size=(F, C, H, W)).astype(np.float32)
    p = {k: _t(v, True) for k, v in dict(xyz=g["xyz"], scales=g["scale"], uquats=g["rotate"], opacity=opacity, feature=feat).items()}
    B = FrameBatch(F, N, W, H, C, "cuda", want_abs=True)
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], _t(off), _t(g["extr"]), bg=bg)
    # ---- frame 0 of the batched preprocess vs the reference's torch functions (tests/golden/make_golden.py)
    np.testing.assert_allclose(B.uv[0].cpu().numpy(), g["uv"], rtol=1e-5, atol=2e-4)
    same0 = B.radius[0].cpu().numpy() == g["radius"]
    assert same0.mean() > 0.9999
    np.testing.assert_allclose(B.conic[0].cpu().numpy()[same0], g["conic"][same0], rtol=2e-5, atol=2e-6 * float(np.abs(g["conic"]).max()))
    with capture_T_front() as cap:
        out.backward(_t(gimg))
    torch.cuda.synchronize()
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    # ---- oracle chain, frame by frame
    sets = [dict(feature=feat, bg=bg, taps=True)]
    per, tot = oc.static_frames(oracle_mod, g["xyz"], off, g["scale"], g["rotate"], opacity, g["extr"], W, H, sets, [gimg])
    same = _same_geometry(B, per)
    if reach:
        assert 0.4 * max(r["M"] for r in per) < B.check() < 0.9 * max(r["M"] for r in per)
    else:
        assert B.check() == max(r["M"] for r in per) or not same
    _check_images(out, per, 0, "render")
    for f, r in enumerate(per):
        nc = B.ncontrib[f].cpu().numpy()
        if reach:   # the last contributor sits at an earlier position of a shorter list; pixels nothing reaches agree
            assert (nc > r["ncontrib"]).mean() < 1e-3 and ((nc > 0) != (r["ncontrib"] > 0)).mean() < 1e-3
        else:
            assert (nc != r["ncontrib"]).mean() < 1e-3
        assert np.abs(B.final_T[f].cpu().numpy() - r["final_T"]).max() < 5e-3
    tol = GRAD_RTOL if same else 5e-3
    assert_grad(p["xyz"].grad, tot["xyz"], "xyz", tol)
    assert_grad(p["scales"].grad, tot["scale"], "scales", tol)
    assert_grad(p["uquats"].grad, tot["rotate"], "uquats", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity", tol)
    assert_grad(p["feature"].grad, tot["feats"][0], "feature", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)
    assert_grad(B.abs_tap, tot["abs_tap"], "abs_tap", tol)
    assert (B.radii_max.cpu().numpy() == np.max([r["radius"] for r in per], 0)).mean() > 0.9999


@pytest.mark.parametrize("std,width", [(1, 19), (0, 19), (1, 1), (1, 3), (1, 4), (1, 7), (1, 8), (1, 9)])
def test_render_sets_against_oracle(oracle_mod, std, width, lib_option):
    """render_sets = the reference's render_iter over a batch: rgb enhanced (K = 20 ids, ndc + abs_ndc taps), depth (bg 1),
    19 attribute channels with opacity.detach() -- one forward over the 23-channel row, then the ONE-pass three-set backward
    (blend_bwd_sets_quarter_kernel: the renderer's plan staging the forward's records, option "sets_std" = 1, or the generic slot
    -> channel routing on its own packed records, "sets_std" = 0) and frames_gauss_bwd_static_sets -- against three oracle
    blends per frame.  The third set's width also takes the trainer's other plans (3 | 1 | 1, 3 | 1 | 3, 3 | 1 | 4: the mask,
    a track or a dino feature alone), which the same kernel serves through the slot routing."""
    lib_option("sets_std", std)
    F, K = 3, 20
    g, opacity, off, rng = _c1_scene(F, seed=3)
    N, W, H = g["xyz"].shape[0], int(g["W"]), int(g["H"])
    rgb = rng.uniform(size=(N, 3)).astype(np.float32)
    attrs = rng.uniform(-1, 1, size=(N, width)).astype(np.float32)
    gs_ = [rng.normal(size=(F, c, H, W)).astype(np.float32) for c in (3, 1, width)]
    p = {k: _t(v, True) for k, v in dict(xyz=g["xyz"], scales=g["scale"], uquats=g["rotate"], opacity=opacity, rgb=rgb,
                                         attrs=attrs).items()}
    B = FrameBatch(F, N, W, H, 4 + width, "cuda", want_abs=True)
    sets = [dict(feature=p["rgb"], bg=0.2, taps=True), dict(feature="depth", bg=1.0),
            dict(feature=p["attrs"], bg=0.0, detach_opacity=True)]
    o_rgb, o_dep, o_att, ids = B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, _t(off), _t(g["extr"]), K=K)
    with capture_T_front() as cap:
        torch.autograd.backward([o_rgb, o_dep, o_att], [_t(x) for x in gs_])
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[-1] - 1).abs().max()) < 2e-4
    osets = [dict(feature=rgb, bg=0.2, taps=True), dict(feature="depth", bg=1.0), dict(feature=attrs, bg=0.0, detach_opacity=True)]
    per, tot = oc.static_frames(oracle_mod, g["xyz"], off, g["scale"], g["rotate"], opacity, g["extr"], W, H, osets, gs_, K=K)
    same = _same_geometry(B, per)
    _check_images(torch.cat([o_rgb, o_dep, o_att], 1), per, 0, "render_sets")
    for f, r in enumerate(per):
        assert (ids[f].cpu().numpy() != r["gs_idx"]).any(-1).mean() < 1e-3
    tol = GRAD_RTOL if same else 5e-3
    assert_grad(p["xyz"].grad, tot["xyz"], "xyz", tol)
    assert_grad(p["scales"].grad, tot["scale"], "scales", tol)
    assert_grad(p["uquats"].grad, tot["rotate"], "uquats", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity (rgb + depth sets only)", tol)
    assert_grad(p["rgb"].grad, tot["feats"][0], "rgb", tol)
    assert_grad(p["attrs"].grad, tot["feats"][2], "attributes", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)
    assert_grad(B.abs_tap, tot["abs_tap"], "abs_tap", tol)


def _dynamic_host(N, T, seed):
    from splatter_a_video_amd.dynamics import FrameClock
    clock = FrameClock(T)
    I = clock.interval_num
    rng = np.random.default_rng(seed)
    f = lambda *s, scale=1.0: rng.normal(0, scale, size=s).astype(np.float32)
    host = dict(position=np.concatenate([rng.uniform(-1.1, 1.1, size=(N, 2)), rng.uniform(2.0, 4.0, size=(N, 1))], 1).astype(np.float32),
                pos_cubic_node=f(N, 4 * I * 3, scale=0.02), rotation=f(N, 4), rot_poly_feat=f(N, 4, 4, scale=0.05),
                rot_fourier_feat=f(N, 8, 4, scale=0.05), opacity=f(N, 1, scale=1.5),
                scaling=np.log(rng.uniform(0.006, 0.03, size=(N, 3))).astype(np.float32))
    return clock, host, rng


def _dyn_check(p, tot, tol, to_gm, I):
    N = p["position"].shape[0]
    assert_grad(to_gm(p["pos_cubic_node"].grad).reshape(N, 4, I, 3), tot["pos_cubic_node"], "pos_cubic_node", tol)
    assert_grad(p["rotation"].grad, tot["rotation"], "rotation", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity", tol)
    assert_grad(p["scaling"].grad, tot["scaling"], "scaling", tol)
    if p["position"].grad is not None:
        assert_grad(p["position"].grad, tot["position"], "position", tol)


def test_render_dynamic_against_oracle(oracle_mod):
    """render_dynamic (rows a15 + f1 in a batch): frame_preprocess_fwd_batch + frames_gauss_bwd_dynamic against the oracle's
    dynamic_eval around its static chain, at c1 size"""
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, to_gaussian_major, to_segment_major
    N, W, H, T, C = 10000, 256, 256, 30, 3
    times = [0, 14, 15, 29]
    F = len(times)
    clock, host, rng = _dynamic_host(N, T, 5)
    I = clock.interval_num
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    gimg = rng.normal(size=(F, C, H, W)).astype(np.float32)
    extr = np.eye(4, dtype=np.float32)
    p = {k: _t(v, k not in ("rot_poly_feat", "rot_fourier_feat")) for k, v in host.items()}
    p["pos_cubic_node"] = to_segment_major(_t(host["pos_cubic_node"]), I).requires_grad_(True)
    ft = _t(feat, True)
    B = FrameBatch(F, N, W, H, C, "cuda")
    out = B.render_dynamic(clock, times, _t(extr), ft, position=p["position"], pos_cubic_node=p["pos_cubic_node"],
                           rotation=p["rotation"], rot_poly_feat=p["rot_poly_feat"], rot_fourier_feat=p["rot_fourier_feat"],
                           opacity=p["opacity"], scaling=p["scaling"], cubic_layout=SEGMENT_MAJOR, bg=0.1)
    with capture_T_front() as cap:
        out.backward(_t(gimg))
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    per, tot = oc.dynamic_frames(oracle_mod, clock, times, host, extr, W, H, [dict(feature=feat, bg=0.1, taps=True)], [gimg])
    same = _same_geometry(B, per)
    _check_images(out, per, 0, "render_dynamic")
    tol = GRAD_RTOL if same else 5e-3
    _dyn_check(p, tot, tol, lambda x: to_gaussian_major(x), I)
    assert_grad(ft.grad, tot["feats"][0], "feature", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)


@pytest.mark.parametrize("std", [1, 0])
def test_render_dynamic_sets_against_oracle_with_reference_parameters(oracle_mod, std, lib_option):
    """render_dynamic_sets = the reference's training frame over a batch, on the parameters of the reference-made dynamic fixture
    (400 Gaussians x 50 frames, tests/golden/make_golden_dynamic.py; its activations are pinned in test_gpu_dynamic.py) tiled
    to fill a 96 x 64 view: dynamic evaluation -> three blends -> one-pass backward -> frames_gauss_bwd_dynamic_sets."""
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, FrameClock, to_gaussian_major, to_segment_major
    lib_option("sets_std", std)
    g = dict(np.load(os.path.join(GOLD, "dynamic_400x50.npz")))
    clock = FrameClock(int(g["T"]), g["intervals"], int(g["start_frame_id"]), int(g["time_len"]))
    I = clock.interval_num
    names = ("position", "pos_cubic_node", "rotation", "rot_poly_feat", "rot_fourier_feat", "opacity", "scaling")
    host = {k: np.ascontiguousarray(g[k], np.float32) for k in names}
    N, W, H, K = host["position"].shape[0], 96, 64, 8
    times = [int(t) for t in g["times"]][:6]
    F = len(times)
    # the fixture's Gaussians are N(0,1) positions with scales around e^-4: a camera that maps them into the view
    extr = np.eye(4, dtype=np.float32)
    extr[0, 0] = extr[1, 1] = 0.3; extr[2, 2] = 0.1; extr[2, 3] = 2.0
    host["scaling"] = host["scaling"] + 2.0      # a few pixels wide under that camera (same raw-parameter chain)
    rng = np.random.default_rng(8)
    rgb = rng.uniform(size=(N, 3)).astype(np.float32)
    attrs = rng.uniform(-1, 1, size=(N, 19)).astype(np.float32)
    gs_ = [rng.normal(size=(F, c, H, W)).astype(np.float32) for c in (3, 1, 19)]
    p = {k: _t(v, k not in ("rot_poly_feat", "rot_fourier_feat", "position")) for k, v in host.items()}
    p["pos_cubic_node"] = to_segment_major(_t(host["pos_cubic_node"]), I).requires_grad_(True)
    t_rgb, t_att = _t(rgb, True), _t(attrs, True)
    B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
    sets = [dict(feature=t_rgb, bg=0.2, taps=True), dict(feature="depth", bg=1.0), dict(feature=t_att, bg=0.0, detach_opacity=True)]
    o_rgb, o_dep, o_att, ids = B.render_dynamic_sets(
        clock, times, _t(extr), sets, position=p["position"], pos_cubic_node=p["pos_cubic_node"], rotation=p["rotation"],
        rot_poly_feat=p["rot_poly_feat"], rot_fourier_feat=p["rot_fourier_feat"], opacity=p["opacity"], scaling=p["scaling"],
        cubic_layout=SEGMENT_MAJOR, K=K)
    with capture_T_front() as cap:
        torch.autograd.backward([o_rgb, o_dep, o_att], [_t(x) for x in gs_])
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[-1] - 1).abs().max()) < 2e-4
    osets = [dict(feature=rgb, bg=0.2, taps=True), dict(feature="depth", bg=1.0), dict(feature=attrs, bg=0.0, detach_opacity=True)]
    per, tot = oc.dynamic_frames(oracle_mod, clock, times, host, extr, W, H, osets, gs_, K=K)
    assert sum(r["M"] for r in per) > 2000      # the view is populated
    same = _same_geometry(B, per)
    _check_images(torch.cat([o_rgb, o_dep, o_att], 1), per, 0, "render_dynamic_sets")
    for f, r in enumerate(per):
        assert (ids[f].cpu().numpy() != r["gs_idx"]).any(-1).mean() < 2e-3
    tol = GRAD_RTOL if same else 5e-3
    _dyn_check(p, tot, tol, lambda x: to_gaussian_major(x), I)
    assert_grad(t_rgb.grad, tot["feats"][0], "rgb", tol)
    assert_grad(t_att.grad, tot["feats"][2], "attributes", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)
    assert_grad(B.abs_tap, tot["abs_tap"], "abs_tap", tol)


def test_wide_row_batch_against_oracle(oracle_mod):
    """configs[4] width: 32 feature channels (the shared-slab matrix-core backward) at 12k Gaussians, two frames, vs the oracle"""
    from splatter_a_video_amd.synth import make_scene
    N, W, H, F, C = 12000, 256, 160, 2, 32
    sc = make_scene(N, W, H, seed=321)
    rng = np.random.default_rng(11)
    off = np.stack([sc.positions(f) - sc.xyz for f in (0, 9)]).astype(np.float32)
    feat = rng.uniform(-1, 1, size=(N, C)).astype(np.float32)
    gimg = rng.normal(size=(F, C, H, W)).astype(np.float32)
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=feat).items()}
    B = FrameBatch(F, N, W, H, C, "cuda")
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], _t(off), _t(sc.extr), bg=0.0)
    with capture_T_front() as cap:
        out.backward(_t(gimg))
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    per, tot = oc.static_frames(oracle_mod, sc.xyz, off, sc.scale, sc.rotate, sc.opacity, sc.extr, W, H,
                                [dict(feature=feat, bg=0.0, taps=True)], [gimg])
    same = _same_geometry(B, per)
    _check_images(out, per, 0, "wide row")
    tol = GRAD_RTOL if same else 5e-3
    assert_grad(p["xyz"].grad, tot["xyz"], "xyz", tol)
    assert_grad(p["scales"].grad, tot["scale"], "scales", tol)
    assert_grad(p["uquats"].grad, tot["rotate"], "uquats", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity", tol)
    assert_grad(p["feature"].grad, tot["feats"][0], "feature", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)


def test_clustered_scene_against_oracle(oracle_mod):
    """SURVEY 7 hard part (ii): a clustered scene (70 % of the Gaussians in blobs that cover 10 % of the image -- foreground
    objects) gives tile lists from a few entries to thousands in one launch: long replays, many super-batches and overflow rounds
    next to empty tiles.  Images and gradients against the oracle."""
    from splatter_a_video_amd.synth import make_scene
    N, W, H, F, C = 15000, 256, 160, 2, 3
    sc = make_scene(N, W, H, seed=55, clustered=0.7)
    sc.opacity[:] = np.clip(sc.opacity * 0.35, 0.0, 0.97)          # faint splats: the long lists are walked, not cut by saturation
    rng = np.random.default_rng(4)
    off = np.stack([sc.positions(f) - sc.xyz for f in (0, 13)]).astype(np.float32)
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    gimg = rng.normal(size=(F, C, H, W)).astype(np.float32)
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=feat).items()}
    B = FrameBatch(F, N, W, H, C, "cuda", want_abs=True)
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], _t(off), _t(sc.extr), bg=0.1)
    with capture_T_front() as cap:
        out.backward(_t(gimg))
    torch.cuda.synchronize()
    B.check()
    tr = B.tile_range.long()
    ln = (tr[..., 1] - tr[..., 0]).float()
    assert float(ln.max()) > 4.0 * float(ln.mean()) and float(ln.max()) > 600      # strongly unbalanced lists
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    per, tot = oc.static_frames(oracle_mod, sc.xyz, off, sc.scale, sc.rotate, sc.opacity, sc.extr, W, H,
                                [dict(feature=feat, bg=0.1, taps=True)], [gimg])
    same = _same_geometry(B, per)
    _check_images(out, per, 0, "clustered")
    tol = GRAD_RTOL if same else 5e-3
    assert_grad(p["xyz"].grad, tot["xyz"], "xyz", tol)
    assert_grad(p["scales"].grad, tot["scale"], "scales", tol)
    assert_grad(p["uquats"].grad, tot["rotate"], "uquats", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity", tol)
    assert_grad(p["feature"].grad, tot["feats"][0], "feature", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)
    assert_grad(B.abs_tap, tot["abs_tap"], "abs_tap", tol)


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    R = np.eye(4, dtype=np.float32)
    R[0, 0], R[0, 1], R[1, 0], R[1, 1] = c, -s, s, c
    return R


@pytest.mark.parametrize("entry", ["render", "render_sets"])
def test_per_frame_cameras_against_oracle(oracle_mod, entry):
    """render_batch gives every batch element its own camera (reference: dptr_ortho_enhanced.py:409-411): F orthographic
    cameras [F,4,4] in one batch, no offsets -- batched preprocess with per-frame extr, Gaussian-side backward with the
    projection chain per frame (frames_gauss_bwd_static_kernel CAM = 1) -- against the oracle chain frame by frame."""
    from splatter_a_video_amd.synth import make_scene
    N, W, H, F = 9000, 192, 128, 3
    sc = make_scene(N, W, H, seed=17)
    rng = np.random.default_rng(2)
    extr = np.stack([_rot_z(0.0), _rot_z(0.07), _rot_z(-0.11)])
    extr[1, 0, 3], extr[1, 1, 3] = 0.05, -0.02
    extr[2, 0, 3], extr[2, 2, 3] = -0.04, 0.3
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}
    if entry == "render":
        C = 3
        feat = rng.uniform(size=(N, C)).astype(np.float32)
        gs_ = [rng.normal(size=(F, C, H, W)).astype(np.float32)]
        ft = _t(feat, True)
        B = FrameBatch(F, N, W, H, C, "cuda", want_abs=True)
        outs = [B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], ft, None, _t(extr), bg=0.2)]
        osets = [dict(feature=feat, bg=0.2, taps=True)]
        feats_t = [ft]
    else:
        rgb = rng.uniform(size=(N, 3)).astype(np.float32)
        attrs = rng.uniform(-1, 1, size=(N, 19)).astype(np.float32)
        gs_ = [rng.normal(size=(F, c, H, W)).astype(np.float32) for c in (3, 1, 19)]
        t_rgb, t_att = _t(rgb, True), _t(attrs, True)
        B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
        sets = [dict(feature=t_rgb, bg=0.2, taps=True), dict(feature="depth", bg=1.0), dict(feature=t_att, bg=0.0, detach_opacity=True)]
        outs = list(B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, None, _t(extr))[:3])
        osets = [dict(feature=rgb, bg=0.2, taps=True), dict(feature="depth", bg=1.0), dict(feature=attrs, bg=0.0, detach_opacity=True)]
        feats_t = [t_rgb, None, t_att]
    torch.autograd.backward(outs, [_t(x) for x in gs_])
    torch.cuda.synchronize()
    B.check()
    per, tot = oc.static_frames(oracle_mod, sc.xyz, None, sc.scale, sc.rotate, sc.opacity, extr, W, H, osets, gs_)
    same = _same_geometry(B, per)
    _check_images(torch.cat(outs, 1), per, 0, entry)
    tol = GRAD_RTOL if same else 5e-3
    assert_grad(p["xyz"].grad, tot["xyz"], "xyz", tol)
    assert_grad(p["scales"].grad, tot["scale"], "scales", tol)
    assert_grad(p["uquats"].grad, tot["rotate"], "uquats", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity", tol)
    for ft_, want in zip(feats_t, tot["feats"]):
        if ft_ is not None:
            assert_grad(ft_.grad, want, "feature", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)
    assert_grad(B.abs_tap, tot["abs_tap"], "abs_tap", tol)


def test_perspective_batch_against_oracle(oracle_mod):
    """the pinhole camera of gs.rasterization / DPTRRender as a frame batch: fused perspective preprocess per frame (per-frame
    extr AND offsets), Gaussian-side backward with the perspective chain per frame (position gradient through the projection
    and the EWA Jacobian; CAM = 2) -- against the oracle's perspective chain"""
    from splatter_a_video_amd.synth import make_scene
    N, W, H, F, C = 8000, 160, 112, 2, 3
    sc = make_scene(N, W, H, seed=29, ortho=False)
    rng = np.random.default_rng(6)
    extr = np.stack([sc.extr, sc.extr @ _rot_z(0.05)]).astype(np.float32)
    off = (0.03 * rng.normal(size=(F, N, 3))).astype(np.float32)
    off[0] = 0.0
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    gimg = rng.normal(size=(F, C, H, W)).astype(np.float32)
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=feat).items()}
    B = FrameBatch(F, N, W, H, C, "cuda")
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], _t(off), _t(extr), bg=0.1, nearest=0.2,
                   intr=_t(sc.intr))
    with capture_T_front() as cap:
        out.backward(_t(gimg))
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    per, tot = oc.static_frames(oracle_mod, sc.xyz, off, sc.scale, sc.rotate, sc.opacity, extr, W, H,
                                [dict(feature=feat, bg=0.1, taps=True)], [gimg], intr=sc.intr, nearest=0.2)
    assert min(r["M"] for r in per) > 5000
    same = _same_geometry(B, per)
    _check_images(out, per, 0, "perspective")
    tol = GRAD_RTOL if same else 5e-3
    assert_grad(p["xyz"].grad, tot["xyz"], "xyz", tol)
    assert_grad(p["scales"].grad, tot["scale"], "scales", tol)
    assert_grad(p["uquats"].grad, tot["rotate"], "uquats", tol)
    assert_grad(p["opacity"].grad, tot["opacity"], "opacity", tol)
    assert_grad(p["feature"].grad, tot["feats"][0], "feature", tol)
    assert_grad(B.tap, tot["tap"], "tap", tol)


def test_fused_perspective_preprocess_equals_the_operator_chain():
    """gs.preprocess_persp (what gs.rasterization runs now) = project_point -> compute_cov3d -> ewa_project: projection, radius and
    tile counts bit for bit (same one-Gaussian device functions), the conic up to the FMA contraction of a cov3d that stays in
    registers, and the gradients"""
    import dptr.gs as gs
    from splatter_a_video_amd.synth import make_scene
    N, W, H = 20000, 320, 200
    sc = make_scene(N, W, H, seed=3, ortho=False)
    rng = np.random.default_rng(1)
    g = [_t(rng.normal(size=s).astype(np.float32)) for s in ((N, 2), (N, 1), (N, 3))]
    intr, extr = _t(sc.intr), _t(sc.extr)
    a = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate).items()}
    uv, depth = gs.project_point(a["xyz"], intr, extr, W, H)
    vis = depth != 0
    cov = gs.compute_cov3d(a["scale"], a["rotate"], vis)
    conic, radius, tiles = gs.ewa_project(a["xyz"], cov, intr, extr, uv, W, H, vis)
    torch.autograd.backward([uv, depth, conic], g)
    b = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate).items()}
    uv2, depth2, conic2, radius2, tiles2 = gs.preprocess_persp(b["xyz"], b["scale"], b["rotate"], intr, extr, W, H)
    assert torch.equal(uv2, uv) and torch.equal(depth2, depth) and torch.equal(radius2, radius) and torch.equal(tiles2, tiles)
    assert torch.allclose(conic2, conic, rtol=2e-5, atol=2e-6 * float(conic.detach().abs().max()))
    torch.autograd.backward([uv2, depth2, conic2], g)
    for k in a:
        x, y = b[k].grad, a[k].grad
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6 * float(y.abs().max())), k
    assert int((radius > 0).sum()) > N // 2
