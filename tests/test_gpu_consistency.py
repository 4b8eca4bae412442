"""Forward / backward consistency of the compositing kernels and robustness of the sort's edge cases.

The reference's forward and backward kernels evaluate a splat's alpha with one shared expression
(src/alpha_blending.cu:78-87 vs :196-203), so the backward includes exactly the (pixel, splat) pairs the forward applied
and its ``T /= (1 - alpha)`` replay arrives at T = 1 in front of the first splat.  The HIP kernels evaluate the exponent
as one polynomial on two different pipes (VALU fma chain in the forward, f32 matrix cores in the backward): the replay
transmittance recorded by ``capture_T_front`` is 1 up to rounding iff not a single inclusion decision flipped
(a flip is off by a factor >= 1 / (1 - 1/255), i.e. by >= 3.9e-3).
"""
import numpy as np
import pytest
import torch

import dptr.gs as gs
from splatter_a_video_amd._lib import SplatError
from splatter_a_video_amd.gs.raster_ops import capture_T_front
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu

FLIP = 1.0 / (1.0 - 1.0 / 255.0) - 1.0      # relative jump of the replayed transmittance per flipped decision


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _geometry(sc, f=0, sigma_scale=1.0):
    xyz = _t(sc.positions(f))
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(xyz, _t(sc.scale * sigma_scale), _t(sc.rotate), _t(sc.extr), sc.W, sc.H,
                                                          nearest=0.01)
    return uv, depth, conic, radius, tiles


@pytest.mark.parametrize("mode", ["pair", "atomic"])
@pytest.mark.parametrize("N,W,H,C,variant,sig", [
    (4000, 100, 60, 3, "plain", 1.0),
    (4000, 100, 60, 3, "enh", 1.0),
    (4000, 100, 60, 3, "bias", 1.0),
    (3000, 64, 64, 3, "plain", 6.0),          # long lists, saturating pixels
    (20000, 256, 256, 19, "plain", 1.0),
    (20000, 256, 256, 32, "plain", 1.0),
    (300000, 854, 480, 3, "plain", 1.0),      # BASELINE configs[1] size
])
def test_backward_replays_forward_decisions(N, W, H, C, variant, sig, mode):
    sc = make_scene(N, W, H, seed=11 + N + C)
    rng = np.random.default_rng(5)
    uv, depth, conic, radius, tiles = _geometry(sc, sigma_scale=sig)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    if mode == "atomic":          # a copy is a foreign index list: wave-reduced atomic kernel
        idx = idx.clone()
    feat = _t(rng.uniform(size=(N, C)).astype(np.float32), True)
    op = _t(sc.opacity, True)
    uvg, cg = uv.detach().requires_grad_(True), conic.detach().requires_grad_(True)
    if variant == "bias":
        ob = _t(rng.uniform(-0.05, 0.1, size=(N, 1)).astype(np.float32), True)
        out = gs.alpha_blending_with_bias(uvg, cg, op, feat, ob, idx, tr, 0.2, W, H)
    elif variant == "enh":
        out, _, _ = gs.alpha_blending_enhanced(uvg, cg, op, feat, idx, tr, 0.2, W, H, K=5)
    else:
        out = gs.alpha_blending(uvg, cg, op, feat, idx, tr, 0.2, W, H)
    g = _t(rng.normal(size=(C, H, W)).astype(np.float32))
    with capture_T_front() as cap:
        out.backward(g)
    torch.cuda.synchronize()
    assert len(cap.maps) == (C + 31) // 32
    m = cap.maps[0]
    assert torch.isfinite(m).all()
    err = (m - 1.0).abs()
    flipped = int((err > 0.25 * FLIP).sum())
    assert flipped == 0, f"{flipped} pixels replay a different set of splats than the forward applied (max |T-1| = {float(err.max()):.3e})"
    assert float(err.max()) < 2e-4     # rounding of the product chain only


def test_blend_after_overflowing_capped_sort_is_memory_safe():
    """ranges, slots and prefixes of an overflowed sort are clamped to the capacity -- blending it (results
    meaningless, flagged by status.check()) must stay inside every buffer."""
    N, W, H = 20000, 128, 96
    sc = make_scene(N, W, H, seed=3)
    uv, depth, conic, radius, tiles = _geometry(sc)
    idx_full, _ = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    M = idx_full.numel()
    cap = M // 3
    idx, tr, st = gs.sort_gaussian_capped(uv, depth, W, H, radius, cap)
    assert idx.numel() == cap
    assert int(tr.max()) <= cap
    feat = _t(np.random.default_rng(0).uniform(size=(N, 3)).astype(np.float32), True)
    op = _t(sc.opacity, True)
    uvg, cg = uv.detach().requires_grad_(True), conic.detach().requires_grad_(True)
    out = gs.alpha_blending(uvg, cg, op, feat, idx, tr, 0.0, W, H)
    out.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(feat.grad).all() and torch.isfinite(uvg.grad).all()
    with pytest.raises(SplatError):
        st.check()


def test_no_gaussian_touches_a_tile_forward_and_backward():
    """P > 0, M == 0 -- the reference returns the background image and zero gradients."""
    N, W, H = 50, 64, 48
    sc = make_scene(N, W, H, seed=1)
    xyz = sc.positions(0) + np.array([[50.0, 50.0, 0.0]], np.float32)     # everything far outside the view
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(_t(xyz), _t(sc.scale), _t(sc.rotate), _t(sc.extr), W, H, nearest=0.01)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    assert idx.numel() == 0
    feat = _t(np.ones((N, 3), np.float32), True)
    op = _t(sc.opacity, True)
    uvg, cg = uv.detach().requires_grad_(True), conic.detach().requires_grad_(True)
    ndc = torch.zeros(N, 2, device="cuda", requires_grad=True)
    out = gs.alpha_blending(uvg, cg, op, feat, idx, tr, 0.25, W, H, ndc)
    assert torch.allclose(out, torch.full_like(out, 0.25))
    out.sum().backward()
    for t in (uvg, cg, op, feat, ndc):
        assert t.grad is not None and float(t.grad.abs().max()) == 0.0
    imgs = gs.alpha_blending_shared(uvg, cg, op, [feat, feat[:, :1]], idx, tr, [0.0, 1.0], W, H)
    (imgs[0].sum() + imgs[1].sum()).backward()


def test_stale_pair_map_raises():
    """the sort's pair map is bound to its idx_sorted / tile_range tensors; an in-place edit of
    either is an error, not a silently wrong gradient."""
    N, W, H = 2000, 64, 48
    sc = make_scene(N, W, H, seed=2)
    uv, depth, conic, radius, tiles = _geometry(sc)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    feat = _t(np.ones((N, 3), np.float32), True)
    idx[:10] = 0       # in-place edit -> version counter moves
    with pytest.raises(SplatError):
        gs.alpha_blending(uv, conic, _t(sc.opacity), feat, idx, tr, 0.0, W, H)
    idx2, tr2 = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    idx3, tr3 = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    with pytest.raises(SplatError):
        gs.alpha_blending(uv, conic, _t(sc.opacity), feat, idx2, tr3, 0.0, W, H)   # outputs of two different sorts
