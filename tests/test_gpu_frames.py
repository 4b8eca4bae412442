"""Frame-batched rendering (splatter_a_video_amd.frames.FrameBatch: every kernel takes the frame as a grid dimension,
the Gaussian-side backward runs once per batch) against F calls of the per-frame operators."""
import numpy as np
import pytest
import torch

import dptr.gs as gs
from splatter_a_video_amd._lib import SplatError
from splatter_a_video_amd import frames as FR
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.gs.raster_ops import capture_T_front
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _offsets(sc, F):
    return np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32)


def _per_frame(sc, p, off, feat, g, W, H, bg, abs_tap=False):
    """the per-frame operator path (fused preprocess -> sort -> blend), frame by frame through autograd"""
    imgs, taps, atap, rad = [], [], [], []
    extr = _t(sc.extr)
    for f in range(off.shape[0]):
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(p["xyz"], p["scales"], p["uquats"], extr, W, H, nearest=0.01,
                                                              offset=off[f])
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        ndc = torch.zeros_like(uv, requires_grad=True)
        andc = torch.zeros_like(uv, requires_grad=True) if abs_tap else None
        img = gs.alpha_blending(uv, conic, p["opacity"], feat, idx, tr, bg, W, H, ndc, andc)
        with capture_T_front() as cap:
            (img * g[f]).sum().backward()
        # the per-frame backward (its own cull) replays exactly the forward's decisions as well: what is left between the two
        # paths is summation order
        assert not cap.maps or float((cap.maps[0] - 1).abs().max()) < 2e-4     # (a frame without pairs launches no backward)
        imgs.append(img.detach()); taps.append(ndc.grad if ndc.grad is not None else torch.zeros_like(uv)); rad.append(radius)
        if abs_tap:
            atap.append(andc.grad)
    return torch.stack(imgs), sum(taps), (sum(atap) if abs_tap else None), torch.stack(rad).max(0).values


@pytest.mark.parametrize("N,W,H,F,C,abs_tap", [(3000, 100, 60, 3, 3, False), (20000, 256, 192, 4, 3, True),
                                               (5000, 96, 64, 2, 19, False), (1, 16, 16, 2, 1, False),
                                               (60000, 854, 480, 2, 3, False),
                                               (700, 64, 48, 35, 3, False)])   # (more frames than the Gaussian-side walk keeps slot ranges for in LDS)
def test_batch_equals_per_frame_operators(N, W, H, F, C, abs_tap):
    sc = make_scene(N, W, H, seed=N + C)
    rng = np.random.default_rng(N)
    off = _t(_offsets(sc, F))
    featv = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    bg = 0.3

    def params():
        return {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}

    pa, fa = params(), _t(featv, True)
    ref_img, ref_tap, ref_atap, ref_rad = _per_frame(sc, pa, off, fa, g, W, H, bg, abs_tap)

    pb, fb_ = params(), _t(featv, True)
    B = FrameBatch(F, N, W, H, C, "cuda", want_abs=abs_tap)
    out = B.render(pb["xyz"], pb["scales"], pb["uquats"], pb["opacity"], fb_, off, _t(sc.extr), bg=bg)
    assert out.shape == (F, C, H, W)
    assert torch.equal(out, ref_img)                      # same kernels, same arithmetic: bit-identical images
    with capture_T_front() as cap:
        out.backward(g)
    torch.cuda.synchronize()
    assert B.check() > 0 or N == 1
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4      # every frame's backward replays its forward's decisions

    def close(a, b, what):
        """same arithmetic, other summation order (the batch sums a Gaussian's records over the frames before the projection
        chain, the per-frame path runs the chain per frame): element-wise 2e-4 / 2e-6 of the maximum -- but for a handful
        of ill-conditioned Gaussians (nearly isotropic: their scale / rotation gradients are differences of terms 1e4 times
        larger), which stay within 1e-3 relative"""
        d = (a - b).abs()
        bad = d > 2e-4 * b.abs() + 2e-6 * float(b.abs().max()) + 1e-12
        assert int(bad.sum()) <= max(2, a.numel() // 100000), (what, int(bad.sum()))
        assert bool((d <= 2e-3 * b.abs() + 2e-5 * float(b.abs().max()) + 1e-12).all()), what

    for k in pa:
        close(pb[k].grad, pa[k].grad, k)
    assert torch.allclose(fb_.grad, fa.grad, rtol=2e-4, atol=2e-6 * float(fa.grad.abs().max()) + 1e-12)
    assert torch.allclose(B.tap, ref_tap, rtol=2e-4, atol=2e-6 * float(ref_tap.abs().max()) + 1e-12)
    if abs_tap:
        assert torch.allclose(B.abs_tap, ref_atap, rtol=2e-4, atol=2e-6 * float(ref_atap.abs().max()) + 1e-12)
    assert torch.equal(B.radii_max, ref_rad)


@pytest.mark.parametrize("C,abs_tap", [(3, False), (3, True), (16, False)])
def test_batch_dense_saturating_scene(C, abs_tap):
    """Long tile lists of large, faint splats: pixels saturate a few hundred splats in, long before their list ends (the forward stops
    whole blocks -- their cull flags stay 0 for the backward), more than half of a super-batch survives a block's cull (the
    backward's slabs take a second round), survivors are carried across many super-batches."""
    N, W, H, F = 12000, 64, 48, 2
    sc = make_scene(N, W, H, seed=77)
    sc.scale[:] = sc.scale * 6.0
    sc.opacity[:] = np.clip(sc.opacity * 0.08 + 0.03, 0.0, 0.97)
    rng = np.random.default_rng(1)
    off = _t(_offsets(sc, F))
    featv = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))

    def params():
        return {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}

    pa, fa = params(), _t(featv, True)
    ref_img, ref_tap, ref_atap, _ = _per_frame(sc, pa, off, fa, g, W, H, 0.0, abs_tap)
    pb, fb_ = params(), _t(featv, True)
    B = FrameBatch(F, N, W, H, C, "cuda", want_abs=abs_tap)
    out = B.render(pb["xyz"], pb["scales"], pb["uquats"], pb["opacity"], fb_, off, _t(sc.extr), bg=0.0)
    assert torch.equal(out, ref_img)
    assert int(B.ncontrib.max()) > 200 and float(B.final_T.min()) < 1.1e-4     # long replays, saturated pixels
    with capture_T_front() as cap:
        out.backward(g)
    torch.cuda.synchronize()
    assert B.check() > 100 * B.T
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    for k in pa:
        a, b = pb[k].grad, pa[k].grad
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5 * float(b.abs().max()) + 1e-12), k
    assert torch.allclose(fb_.grad, fa.grad, rtol=1e-3, atol=1e-5 * float(fa.grad.abs().max()) + 1e-12)
    assert torch.allclose(B.tap, ref_tap, rtol=1e-3, atol=1e-5 * float(ref_tap.abs().max()) + 1e-12)
    if abs_tap:
        assert torch.allclose(B.abs_tap, ref_atap, rtol=1e-3, atol=1e-5 * float(ref_atap.abs().max()) + 1e-12)


def test_batch_gradient_sinks_and_reuse():
    """the backward adds into caller-owned buffers (FlatGradBucket views); a FrameBatch is reused step after step
    without a host sync"""
    N, W, H, F, C = 4000, 128, 96, 3, 3
    sc = make_scene(N, W, H, seed=4)
    rng = np.random.default_rng(0)
    off = _t(_offsets(sc, F))
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity,
                                         feature=rng.uniform(size=(N, C)).astype(np.float32)).items()}
    B = FrameBatch(F, N, W, H, C, "cuda")
    B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, _t(sc.extr)).backward(g)
    want = {k: v.grad.clone() for k, v in p.items()}
    sink = {k: torch.full_like(v, 0.5) for k, v in p.items()}
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    for _ in range(2):      # two batches accumulate twice
        B.render(q["xyz"], q["scales"], q["uquats"], q["opacity"], q["feature"], off, _t(sc.extr), grad_sink=sink).backward(g)
    for k in p:
        assert q[k].grad is None
        assert torch.allclose(sink[k], 0.5 + 2 * want[k], rtol=2e-4, atol=2e-6 * float(want[k].abs().max()) + 1e-7), k
    B.check()


def test_batch_capacity_overflow_is_flagged_and_safe():
    N, W, H, F, C = 8000, 128, 96, 2, 3
    sc = make_scene(N, W, H, seed=9)
    off = _t(_offsets(sc, F))
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity,
                                         feature=np.ones((N, C), np.float32)).items()}
    B = FrameBatch(F, N, W, H, C, "cuda", capacity=2000)
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, _t(sc.extr))
    out.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(p["xyz"].grad).all()
    with pytest.raises(SplatError):
        B.check()
    # check() reported it and cleared the sticky flag.  A caller that never calls check() learns of an overflow at a later
    # forward without a host synchronisation of its own -- also when the host runs ahead of the GPU (the pending
    # event used to be replaced on every forward, so a loop without host syncs polled events that had never completed)
    render = lambda q=p: B.render(q["xyz"], q["scales"], q["uquats"], q["opacity"], q["feature"], off, _t(sc.extr))
    with torch.no_grad():
        raised = False
        for _ in range(200):         # every one of these overflows again; no host sync in between: the copy taken behind the
            try:                     # first one completes while the host keeps enqueuing, and the forward that finds it raises
                render()
            except SplatError:
                raised = True
                break
        assert raised
        torch.cuda.synchronize()
        # ... raised once; the flag is cleared (behind everything enqueued so far): a batch that fits runs again on the same
        # object
        few = p["xyz"].detach().clone()
        few[300:, 2] = -1.0          # behind the near plane: 300 Gaussians (~1200 pairs) stay
        small = dict(p, xyz=few)
        out = render(small)
        torch.cuda.synchronize()
        render(small)                # (the forward that polls the copy taken behind the fitting batch)
        torch.cuda.synchronize()
        render(small)
        assert B.check() <= 2000 and torch.isfinite(out).all()


def test_backward_refuses_cameras_or_offsets_edited_in_place():
    """the Gaussian-side backward re-projects xyz + offsets[f] under extr[f] from raw pointers -- an in-place edit
    between forward and backward must raise like autograd's own version check, not silently change the gradients"""
    N, W, H, F, C = 3000, 96, 64, 2, 3
    sc = make_scene(N, W, H, seed=4)
    off = _t(_offsets(sc, F))
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity,
                                         feature=np.ones((N, C), np.float32)).items()}
    B = FrameBatch(F, N, W, H, C, "cuda")
    extr = _t(sc.extr).unsqueeze(0).repeat(F, 1, 1).contiguous()
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, extr)
    off.mul_(1.5)
    with pytest.raises(RuntimeError, match="modified in place"):
        out.sum().backward()
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, extr)
    extr[1, 0, 3] += 0.01
    with pytest.raises(RuntimeError, match="modified in place"):
        out.sum().backward()
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, extr)
    out.sum().backward()
    assert torch.isfinite(p["xyz"].grad).all()


@pytest.mark.parametrize("one_pass", ["1", "0"])
def test_render_sets_equals_the_native_renderer_frame_by_frame(one_pass, monkeypatch):
    """(one_pass: the three sets' backward in ONE pass of the tile kernels, or one pass per set.)
    row a1 in a frame batch: rgb (enhanced K ids, ndc + abs_ndc taps) + depth (bg = 1) + 19 attribute channels
    (opacity detached) of every frame in one set of launches, against OrthoEnhancedRenderer.render_iter frame by frame
    (which tests/test_gpu_renderer_native.py / test_gpu_renderer_flow.py tie to the reference's call sequence)."""
    from splatter_a_video_amd.renderer import OrthoEnhancedRenderer
    monkeypatch.setitem(FR.OPTIONS, "sets_one_pass", one_pass == "1")
    N, W, H, F, K = 9000, 192, 128, 3, 20
    sc = make_scene(N, W, H, seed=15)
    rng = np.random.default_rng(3)
    off = _t(_offsets(sc, F))
    attrs_np = rng.uniform(-1, 1, size=(N, 19)).astype(np.float32)
    rgb_np = rng.uniform(0, 1, size=(N, 3)).astype(np.float32)
    g_rgb, g_dep, g_att = (_t(rng.normal(size=(F, c, H, W)).astype(np.float32)) for c in (3, 1, 19))
    extr = _t(sc.extr)

    def params():
        return {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, rgb=rgb_np,
                                                attrs=attrs_np).items()}

    # ---- reference path: the native per-frame renderer
    pa = params()
    R = OrthoEnhancedRenderer(densify_abs_grad_enable=True)
    imgs, taps, gids, radii = [], [], [], []
    for f in range(F):
        r = R.render_iter(H, W, extr, pa["xyz"] + off[f], pa["opacity"], pa["scales"], pa["uquats"], None, num_idx=K, rgb=pa["rgb"],
                          render_attributes={"mask_attribute": pa["attrs"][:, :1], "dino_attribute": pa["attrs"][:, 1:]})
        fs = r["rendered_features_split"]
        att = torch.cat([fs["mask_attribute"], fs["dino_attribute"]], 0)
        torch.autograd.backward([fs["rgb"], fs["depth"], att], [g_rgb[f], g_dep[f], g_att[f]])
        imgs.append((fs["rgb"].detach(), fs["depth"].detach(), att.detach()))
        taps.append(r["viewspace_points"].grad.clone()); gids.append(r["gs_idx"]); radii.append(r["radii"])

    # ---- frame batch
    pb = params()
    B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
    sets = [dict(feature=pb["rgb"], bg=0.0, taps=True), dict(feature="depth", bg=1.0),
            dict(feature=pb["attrs"], bg=0.0, detach_opacity=True)]
    o_rgb, o_dep, o_att, gs_idx = B.render_sets(pb["xyz"], pb["scales"], pb["uquats"], pb["opacity"], sets, off, extr, K=K)
    assert o_rgb.shape == (F, 3, H, W) and o_dep.shape == (F, 1, H, W) and o_att.shape == (F, 19, H, W) and gs_idx.shape == (F, H, W, K)
    for f in range(F):
        assert torch.equal(o_rgb[f], imgs[f][0]) and torch.equal(o_dep[f], imgs[f][1]) and torch.equal(o_att[f], imgs[f][2])
        assert torch.equal(gs_idx[f], gids[f])
    torch.autograd.backward([o_rgb, o_dep, o_att], [g_rgb, g_dep, g_att])
    torch.cuda.synchronize()
    B.check()
    for k in pa:
        a, b = pb[k].grad, pa[k].grad
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5 * float(b.abs().max()) + 1e-12), k
    want_tap = sum(taps)             # abs_ndc taps (densify_abs_grad_enable)
    assert torch.allclose(B.abs_tap, want_tap, rtol=1e-3, atol=1e-5 * float(want_tap.abs().max()))
    assert torch.equal(B.radii_max, torch.stack(radii).max(0).values)


@pytest.mark.parametrize("case", ["detached_first", "two_live_sets", "tap_only", "odd_sizes_one_frame"])
def test_render_sets_one_pass_equals_the_per_set_passes(case, monkeypatch):
    """splat_alpha_blending_backward_batch_sets (one replay of the alpha / T chain, dL/dalpha routed per set) against the
    per-set passes, for set orders and routings other than the reference renderer's: parameter gradients, feature
    gradients, both taps and the replay check (the one-pass kernel reproduces every inclusion decision of the forward)."""
    N, W, H, F = (777, 50, 34, 1) if case == "odd_sizes_one_frame" else (6000, 160, 96, 2)
    sc = make_scene(N, W, H, seed=31)
    rng = np.random.default_rng(5)
    off, extr = _t(_offsets(sc, F)), _t(sc.extr)
    widths = {"detached_first": (8, 3), "two_live_sets": (4, 2), "tap_only": (3,), "odd_sizes_one_frame": (3, 5)}[case]
    feats_np = [rng.uniform(-1, 1, size=(N, w)).astype(np.float32) for w in widths]
    gw = (3, 1, 5) if case == "odd_sizes_one_frame" else widths        # image gradients per set (incl. the depth set)
    gs_ = [_t(rng.normal(size=(F, w, H, W)).astype(np.float32)) for w in gw]

    def run(flag):
        monkeypatch.setitem(FR.OPTIONS, "sets_one_pass", flag == "1")
        p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}
        fe = [_t(f, True) for f in feats_np]
        if case == "detached_first":
            sets = [dict(feature=fe[0], bg=0.3, detach_opacity=True), dict(feature=fe[1], bg=0.1, taps=True)]
        elif case == "two_live_sets":
            sets = [dict(feature=fe[0], bg=0.0, taps=True), dict(feature=fe[1], bg=0.5)]
        elif case == "odd_sizes_one_frame":   # P, W, H no multiples of the block sizes; the per-frame depth as the middle set
            sets = [dict(feature=fe[0], bg=0.1, taps=True), dict(feature="depth", bg=1.0),
                    dict(feature=fe[1], bg=0.0, detach_opacity=True)]
        else:
            sets = [dict(feature=fe[0], bg=0.2, taps=True)]
        B = FrameBatch(F, N, W, H, sum(gw), "cuda", want_abs=True)
        res = B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, off, extr)
        with capture_T_front() as cap:
            torch.autograd.backward(list(res[:-1]), gs_)
        torch.cuda.synchronize()
        B.check()
        assert float((cap.maps[-1] - 1).abs().max()) < 2e-4
        return [x.detach() for x in res[:-1]], {k: v.grad for k, v in p.items()}, [f.grad for f in fe], B.tap.clone(), B.abs_tap.clone()

    i1, g1, f1, t1, a1 = run("1")
    i0, g0, f0, t0, a0 = run("0")
    for x, y in zip(i1, i0):
        assert torch.equal(x, y)
    close = lambda a, b: torch.allclose(a, b, rtol=1e-3, atol=2e-5 * float(b.abs().max()) + 1e-12)
    for k in g0:
        assert close(g1[k], g0[k]), k
    for x, y in zip(f1, f0):
        assert close(x, y)
    assert close(t1, t0) and close(a1, a0)


def test_render_sets_skips_sets_without_gradient():
    N, W, H, F = 3000, 96, 64, 2
    sc = make_scene(N, W, H, seed=2)
    off = _t(_offsets(sc, F))
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}
    rgb = _t(np.random.default_rng(0).uniform(size=(N, 3)).astype(np.float32), True)
    B = FrameBatch(F, N, W, H, 4, "cuda")
    o_rgb, o_dep, gi = B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"],
                                     [dict(feature=rgb, taps=True), dict(feature="depth", bg=1.0)], off, _t(sc.extr))
    assert gi is None
    o_rgb.sum().backward()           # the depth image is unused: its set launches nothing
    ref = FrameBatch(F, N, W, H, 3, "cuda")
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    rgb2 = rgb.detach().clone().requires_grad_(True)
    ref.render(q["xyz"], q["scales"], q["uquats"], q["opacity"], rgb2, off, _t(sc.extr)).sum().backward()
    for k in p:
        assert torch.allclose(p[k].grad, q[k].grad, rtol=1e-4, atol=1e-6 * float(q[k].grad.abs().max()) + 1e-12), k
    assert torch.allclose(rgb.grad, rgb2.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("layout", ["segment_major", "gaussian_major"])
def test_dynamic_batch_equals_per_frame_dynamic_path(layout):
    """rows a15 + f1 in a frame batch: the reference's dynamic Gaussians evaluated inside the batched preprocess, the
    Gaussian-side backward walking all frames -- against the per-frame fused path (dynamics.frame_preprocess -> sort ->
    blend), which tests/test_gpu_dynamic.py pins to the oracle and to the reference's getters."""
    from splatter_a_video_amd.dynamics import GAUSSIAN_MAJOR, SEGMENT_MAJOR, FrameClock, frame_preprocess, to_segment_major
    N, W, H, T, C = 5000, 128, 96, 30, 3
    times = [0, 3, 4, 5, 17, 29]
    F = len(times)
    sc = make_scene(N, W, H, F=T, seed=23)
    rng = np.random.default_rng(9)
    clock = FrameClock(T)
    I = clock.interval_num
    lay = SEGMENT_MAJOR if layout == "segment_major" else GAUSSIAN_MAJOR
    cub = (0.01 * rng.normal(size=(N, 4 * I * 3))).astype(np.float32)
    cub_t = to_segment_major(torch.as_tensor(cub), I).numpy() if lay == SEGMENT_MAJOR else cub
    op = np.clip(sc.opacity, 1e-4, 1 - 1e-4)
    base = dict(position=sc.xyz, pos_cubic_node=cub_t, rotation=sc.rotate + 0.1 * rng.normal(size=(N, 4)).astype(np.float32),
                opacity=np.log(op / (1 - op)).astype(np.float32), scaling=np.log(sc.scale).astype(np.float32),
                feature=rng.uniform(size=(N, C)).astype(np.float32))
    rot_poly = _t((0.02 * rng.normal(size=(N, 4, 4))).astype(np.float32))
    rot_four = _t((0.02 * rng.normal(size=(N, 8, 4))).astype(np.float32))
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    extr = _t(sc.extr)

    pa = {k: _t(v, True) for k, v in base.items()}
    ref_imgs, ref_tap = [], 0
    for f, t in enumerate(times):
        uv, depth, conic, radius, tiles, opa = frame_preprocess(
            clock, t, extr, W, H, position=pa["position"], pos_cubic_node=pa["pos_cubic_node"], rotation=pa["rotation"],
            rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pa["opacity"], scaling=pa["scaling"], nearest=0.01,
            cubic_layout=lay)
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        ndc = torch.zeros_like(uv, requires_grad=True)
        img = gs.alpha_blending(uv, conic, opa, pa["feature"], idx, tr, 0.1, W, H, ndc)
        (img * g[f]).sum().backward()
        ref_imgs.append(img.detach()); ref_tap = ref_tap + ndc.grad

    pb = {k: _t(v, True) for k, v in base.items()}
    B = FrameBatch(F, N, W, H, C, "cuda")
    out = B.render_dynamic(clock, times, extr, pb["feature"], position=pb["position"], pos_cubic_node=pb["pos_cubic_node"],
                           rotation=pb["rotation"], rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pb["opacity"],
                           scaling=pb["scaling"], cubic_layout=lay, bg=0.1)
    # (the frame-loop kernels keep the table rows in registers: same formulas, different FMA contraction -> not bit-equal)
    assert torch.allclose(out, torch.stack(ref_imgs), rtol=1e-4, atol=1e-5)
    with capture_T_front() as cap:
        out.backward(g)
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    for k in pa:
        a, b = pb[k].grad, pa[k].grad
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5 * float(b.abs().max()) + 1e-12), k
    assert torch.allclose(B.tap, ref_tap, rtol=1e-3, atol=1e-5 * float(ref_tap.abs().max()))
    # gradient sinks: the same gradients land in caller-owned buffers
    pc = {k: _t(v, True) for k, v in base.items()}
    sink = {k: torch.zeros_like(v) for k, v in pc.items()}
    B.render_dynamic(clock, times, extr, pc["feature"], position=pc["position"], pos_cubic_node=pc["pos_cubic_node"],
                     rotation=pc["rotation"], rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pc["opacity"],
                     scaling=pc["scaling"], cubic_layout=lay, bg=0.1, grad_sink=sink).backward(g)
    for k in pc:
        assert pc[k].grad is None and torch.allclose(sink[k], pb[k].grad, rtol=2e-4, atol=2e-6 * float(pb[k].grad.abs().max()) + 1e-12), k


def test_dynamic_sets_batch_equals_per_frame_three_blends():
    """the reference's real training frame in a frame batch -- its dynamic Gaussians (rows a15 + f1) through render_iter's
    three blends (row a1: rgb enhanced with the taps, depth with bg = 1, attributes with opacity.detach()) -- against the
    per-frame operators (dynamics.frame_preprocess -> sort -> alpha_blending_enhanced / alpha_blending x 2) through autograd."""
    from splatter_a_video_amd.dynamics import SEGMENT_MAJOR, FrameClock, frame_preprocess, to_segment_major
    N, W, H, T, K = 5000, 128, 96, 30, 8
    times = [0, 3, 4, 5, 17, 29]
    F = len(times)
    sc = make_scene(N, W, H, F=T, seed=41)
    rng = np.random.default_rng(12)
    clock = FrameClock(T)
    I = clock.interval_num
    cub = to_segment_major(torch.as_tensor((0.01 * rng.normal(size=(N, 4 * I * 3))).astype(np.float32)), I).numpy()
    op = np.clip(sc.opacity, 1e-4, 1 - 1e-4)
    base = dict(position=sc.xyz, pos_cubic_node=cub, rotation=sc.rotate + 0.1 * rng.normal(size=(N, 4)).astype(np.float32),
                opacity=np.log(op / (1 - op)).astype(np.float32), scaling=np.log(sc.scale).astype(np.float32),
                rgb=rng.uniform(size=(N, 3)).astype(np.float32), attrs=rng.uniform(-1, 1, size=(N, 19)).astype(np.float32))
    rot_poly = _t((0.02 * rng.normal(size=(N, 4, 4))).astype(np.float32))
    rot_four = _t((0.02 * rng.normal(size=(N, 8, 4))).astype(np.float32))
    g_rgb, g_dep, g_att = (_t(rng.normal(size=(F, c, H, W)).astype(np.float32)) for c in (3, 1, 19))
    extr = _t(sc.extr)

    pa = {k: _t(v, True) for k, v in base.items()}
    ref, ref_tap, ref_atap, ref_ids = [], 0, 0, []
    for f, t in enumerate(times):
        uv, depth, conic, radius, tiles, opa = frame_preprocess(
            clock, t, extr, W, H, position=pa["position"], pos_cubic_node=pa["pos_cubic_node"], rotation=pa["rotation"],
            rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pa["opacity"], scaling=pa["scaling"], nearest=0.01,
            cubic_layout=SEGMENT_MAJOR)
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        ndc = torch.zeros_like(uv, requires_grad=True)
        andc = torch.zeros_like(uv, requires_grad=True)
        rgb_i, _, ids = gs.alpha_blending_enhanced(uv, conic, opa, pa["rgb"], idx, tr, 0.2, W, H, ndc, andc, K=K)
        dep_i = gs.alpha_blending(uv, conic, opa, depth, idx, tr, 1.0, W, H, ndc.detach())
        att_i = gs.alpha_blending(uv, conic, opa.detach(), pa["attrs"], idx, tr, 0.0, W, H, ndc.detach())
        torch.autograd.backward([rgb_i, dep_i, att_i], [g_rgb[f], g_dep[f], g_att[f]])
        ref.append((rgb_i.detach(), dep_i.detach(), att_i.detach()))
        ref_tap = ref_tap + ndc.grad; ref_atap = ref_atap + andc.grad; ref_ids.append(ids)

    pb = {k: _t(v, True) for k, v in base.items()}
    B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
    sets = [dict(feature=pb["rgb"], bg=0.2, taps=True), dict(feature="depth", bg=1.0),
            dict(feature=pb["attrs"], bg=0.0, detach_opacity=True)]
    o_rgb, o_dep, o_att, ids = B.render_dynamic_sets(
        clock, times, extr, sets, position=pb["position"], pos_cubic_node=pb["pos_cubic_node"], rotation=pb["rotation"],
        rot_poly_feat=rot_poly, rot_fourier_feat=rot_four, opacity=pb["opacity"], scaling=pb["scaling"],
        cubic_layout=SEGMENT_MAJOR, K=K)
    for f in range(F):
        # the batched preprocess contracts its FMAs differently from the per-frame one: last-bit geometry, and once in a
        # while a splat that sits on the alpha = 1/255 threshold of a pixel is applied on one side only (<= 4e-3 there)
        for name, got, want in zip(("rgb", "depth", "attrs"), (o_rgb[f], o_dep[f], o_att[f]), ref[f]):
            off_ = ((got - want).abs() > 1e-5 + 1e-4 * want.abs())
            assert int(off_.any(0).sum()) <= 3 and float((got - want).abs().max()) < 5e-3, (f, name)
        assert int((ids[f] != ref_ids[f]).any(-1).sum()) <= 3
    with capture_T_front() as cap:
        torch.autograd.backward([o_rgb, o_dep, o_att], [g_rgb, g_dep, g_att])
    torch.cuda.synchronize()
    B.check()
    assert float((cap.maps[0] - 1).abs().max()) < 2e-4
    for k in pa:
        a, b = pb[k].grad, pa[k].grad
        bad = (a - b).abs() > 1e-3 * b.abs() + 1e-5 * float(b.abs().max())
        # (the one or two splats of the flipped decisions: up to a row of 19 attribute gradients each)
        assert int(bad.sum()) <= 40 and float(bad.float().mean()) < 1e-3, (k, int(bad.sum()), float((a - b).abs().max()))
    for name, a, b in (("tap", B.tap, ref_tap), ("abs_tap", B.abs_tap, ref_atap)):
        bad = (a - b).abs() > 1e-3 * b.abs() + 1e-5 * float(b.abs().max())
        assert int(bad.sum()) <= 4, (name, int(bad.sum()), float((a - b).abs().max()))


def test_one_forward_one_backward_contract_is_enforced():
    """a second render* on the same FrameBatch overwrites the buffers the first call's backward reads: that backward raises
    instead of returning gradients of the wrong lists (gradient accumulation over micro-batches, a no_grad eval render in
    between); sequential forward -> backward pairs stay legal"""
    N, W, H, F, C = 3000, 96, 64, 2, 3
    sc = make_scene(N, W, H, seed=6)
    off = _t(_offsets(sc, F))
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity,
                                         feature=np.ones((N, C), np.float32)).items()}
    B = FrameBatch(F, N, W, H, C, "cuda")
    args = (p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, _t(sc.extr))
    first = B.render(*args)
    with torch.no_grad():
        B.render(*args)                       # e.g. an evaluation render between forward and backward
    with pytest.raises(SplatError, match="ONE forward at a time"):
        first.sum().backward()
    a = B.render(*args)
    b = B.render(*args)                       # two micro-batches, backward afterwards: the first graph is stale
    b.sum().backward()
    with pytest.raises(SplatError):
        a.sum().backward()
    B.render(*args).sum().backward()          # the normal sequence still works
    sets = [dict(feature=p["feature"], taps=True)]
    s1 = B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, off, _t(sc.extr))
    B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, off, _t(sc.extr))
    with pytest.raises(SplatError):
        s1[0].sum().backward()


def test_render_sets_validates_shapes():
    N, W, H, F = 2000, 64, 48, 2
    sc = make_scene(N, W, H, seed=8)
    off, extr = _t(_offsets(sc, F)), _t(sc.extr)
    p = {k: _t(v) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}
    rgb = _t(np.ones((N, 3), np.float32))
    B = FrameBatch(F, N, W, H, 3, "cuda")
    ok = [dict(feature=rgb, taps=True)]
    B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], ok, off, extr)
    with pytest.raises(ValueError):           # a parameter with the wrong number of rows
        B.render_sets(p["xyz"][:-1], p["scales"], p["uquats"], p["opacity"], ok, off, extr)
    with pytest.raises(ValueError):
        B.render_sets(p["xyz"], p["scales"][:-1], p["uquats"], p["opacity"], ok, off, extr)
    with pytest.raises(ValueError):           # per-frame opacity has no entry point in the Gaussian-side backward
        B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"].unsqueeze(0).repeat(F, 1, 1), ok, off, extr)
    with pytest.raises(ValueError):           # a set's feature rows
        B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], [dict(feature=rgb[:-1], taps=True)], off, extr)


@pytest.mark.parametrize("case", ["one_empty_frame", "all_empty", "single_frame"])
def test_batch_with_empty_frames_and_a_single_frame(case):
    """edge cases of the frame batch against the per-frame operators: a frame whose offsets carry every Gaussian out of view
    (no (tile, Gaussian) pair: its image is the background, it adds nothing to any gradient) between two ordinary frames; a batch
    of such frames only; a batch of ONE frame"""
    N, W, H, C = 2500, 100, 60, 3
    F = 1 if case == "single_frame" else 3
    sc = make_scene(N, W, H, seed=41)
    off = _offsets(sc, F)
    if case == "one_empty_frame":
        off[1, :, 0] += 50.0                      # far to the right of the view
    if case == "all_empty":
        off[:, :, 0] += 50.0
    off = _t(off)
    rng = np.random.default_rng(4)
    featv = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    bg = 0.3
    params = lambda: {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity).items()}
    pa, fa = params(), _t(featv, True)
    ref_img, ref_tap, _, ref_rad = _per_frame(sc, pa, off, fa, g, W, H, bg)
    pb, fb_ = params(), _t(featv, True)
    B = FrameBatch(F, N, W, H, C, "cuda")
    out = B.render(pb["xyz"], pb["scales"], pb["uquats"], pb["opacity"], fb_, off, _t(sc.extr), bg=bg)
    assert torch.equal(out, ref_img)
    if case != "single_frame":
        empty = 1 if case == "one_empty_frame" else 0
        assert torch.equal(out[empty], torch.full_like(out[empty], bg))
    out.backward(g)
    torch.cuda.synchronize()
    B.check()
    for k in pa:
        ga, gb = pa[k].grad, pb[k].grad
        assert torch.isfinite(gb).all()
        assert torch.allclose(gb, ga, rtol=2e-3, atol=2e-5 * float(ga.abs().max()) + 1e-12), k
    assert torch.allclose(fb_.grad, fa.grad, rtol=2e-4, atol=2e-6 * float(fa.grad.abs().max()) + 1e-12)
    assert torch.allclose(B.tap, ref_tap, rtol=2e-4, atol=2e-6 * float(ref_tap.abs().max()) + 1e-12)
    assert torch.equal(B.radii_max, ref_rad)
    if case == "all_empty":
        assert float(fb_.grad.abs().max()) == 0.0 and float(pb["xyz"].grad.abs().max()) == 0.0 and int(B.radii_max.max()) == 0


def test_render_sets_with_an_empty_frame_equals_one_frame_batches():
    """the three blends of a batch whose middle frame shows nothing (no pair: images = the sets' backgrounds, ids = -1) against
    the same frames rendered one batch each: identical images and ids, gradients to summation order"""
    N, W, H, F, K = 3000, 112, 80, 3, 6
    sc = make_scene(N, W, H, seed=23)
    rng = np.random.default_rng(6)
    off = _offsets(sc, F)
    off[1, :, 1] -= 40.0
    off = _t(off)
    rgb_np, att_np = rng.uniform(size=(N, 3)).astype(np.float32), rng.uniform(-1, 1, size=(N, 19)).astype(np.float32)
    gs_ = [_t(rng.normal(size=(F, c, H, W)).astype(np.float32)) for c in (3, 1, 19)]
    extr = _t(sc.extr)
    params = lambda: {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, rgb=rgb_np,
                                                      attrs=att_np).items()}

    def render(B, p, o):
        sets = [dict(feature=p["rgb"], bg=0.1, taps=True), dict(feature="depth", bg=1.0),
                dict(feature=p["attrs"], bg=0.0, detach_opacity=True)]
        return B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, o, extr, K=K)

    pa, pb = params(), params()
    B = FrameBatch(F, N, W, H, 23, "cuda")
    out = render(B, pb, off)
    torch.autograd.backward(list(out[:3]), gs_)
    torch.cuda.synchronize()
    B.check()
    assert torch.equal(out[0][1], torch.full_like(out[0][1], 0.1)) and torch.equal(out[1][1], torch.ones_like(out[1][1]))
    assert float(out[2][1].detach().abs().max()) == 0.0 and int(out[3][1].max()) == -1
    tap = torch.zeros_like(B.tap)
    for f in range(F):
        B1 = FrameBatch(1, N, W, H, 23, "cuda")
        o1 = render(B1, pa, off[f:f + 1])
        for a, b in zip(o1, out):
            assert torch.equal(a[0], b[f])
        torch.autograd.backward(list(o1[:3]), [g[f:f + 1] for g in gs_])
        torch.cuda.synchronize()
        tap += B1.tap
    for k in pa:
        a, b = pb[k].grad, pa[k].grad
        assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=2e-3, atol=2e-5 * float(b.abs().max()) + 1e-12), k
    assert torch.allclose(B.tap, tap, rtol=1e-3, atol=1e-5 * float(tap.abs().max()) + 1e-12)
