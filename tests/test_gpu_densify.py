"""SURVEY 8(f) rank 2 on the GPU: the HIP densification kernels (through splatter_a_video_amd.densify) against the vectors
produced by the reference's own optimizer methods, the C oracle at a larger size, and torch boolean indexing."""
import os

import numpy as np
import pytest
import torch

import oracle
from splatter_a_video_amd.densify import DensifyState, compact

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "densify_5000.npz")


def _t(a):
    return torch.tensor(np.asarray(a), device="cuda")


def test_statistics_and_masks_match_reference_vectors():
    g = dict(np.load(G))
    N, F, STEPS = int(g["N"]), int(g["F"]), int(g["STEPS"])
    st = DensifyState(N, "cuda")
    for s in range(STEPS):
        st.begin_batch()
        for f in range(F):
            st.accumulate_frame(_t(g[f"s{s}_radius"][f]), _t(g[f"s{s}_taps"][f]))
        np.testing.assert_array_equal(st.visibility.cpu().numpy().astype(bool), g[f"s{s}_visibility"])
        np.testing.assert_array_equal(st.radii.cpu().numpy(), g[f"s{s}_radii"])
        np.testing.assert_array_equal(st.viewspace_grad.cpu().numpy(), g[f"s{s}_viewspace_grad"])   # same order: bit exact
        st.update()
        np.testing.assert_array_equal(st.max_radii2D.cpu().numpy(), g[f"s{s}_max_radii2D"])
        np.testing.assert_array_equal(st.denom.cpu().numpy(), g[f"s{s}_denom"])
        np.testing.assert_allclose(st.pos_gradient_accum.cpu().numpy(), g[f"s{s}_pos_gradient_accum"], rtol=2e-6, atol=0)
    clone, split, prune = st.masks(_t(g["scaling_raw"]), _t(g["opacity_raw"]), float(g["densify_grad_threshold"]),
                                   float(g["percent_dense"]), float(g["cameras_extent"]), float(g["min_opacity"]))
    # exp / sigmoid implementations differ in the last ulp: a comparison sitting on its threshold may flip
    assert (clone.cpu().numpy() != g["clone_mask"]).sum() <= 2
    assert (split.cpu().numpy() != g["split_mask"]).sum() <= 2
    assert (prune.cpu().numpy() != ~g["prune_valid_mask"]).sum() <= 2
    # prune_postprocess keeps exactly the statistics of the survivors
    valid = ~prune
    a0, d0, m0 = st.pos_gradient_accum.clone(), st.denom.clone(), st.max_radii2D.clone()
    st.prune_postprocess(valid)
    assert st.num_points == int(valid.sum())
    assert torch.equal(st.pos_gradient_accum, a0[valid]) and torch.equal(st.denom, d0[valid]) and torch.equal(st.max_radii2D, m0[valid])


def test_dl_duv_with_scale_equals_tap():
    """feeding dL_duv with (W/2, H/2) is what the ndc tap receives"""
    N, W, H = 70_001, 854, 480
    rng = np.random.default_rng(1)
    duv = rng.normal(size=(N, 2)).astype(np.float32)
    radius = rng.integers(0, 9, size=N).astype(np.int32)
    a, b = DensifyState(N, "cuda"), DensifyState(N, "cuda")
    for _ in range(3):
        a.accumulate_frame(_t(radius), _t(duv) * torch.tensor([0.5 * W, 0.5 * H], device="cuda"))
        b.accumulate_frame(_t(radius), _t(duv), scale=(0.5 * W, 0.5 * H))
    assert torch.equal(a.viewspace_grad, b.viewspace_grad) and torch.equal(a.radii, b.radii)
    vg = np.zeros((N, 2), np.float32); vis = np.zeros(N, np.uint8); rr = np.zeros(N, np.int32)
    for _ in range(3):
        oracle.densify_accumulate(radius, duv, 0.5 * W, 0.5 * H, vg, vis, rr)
    np.testing.assert_allclose(b.viewspace_grad.cpu().numpy(), vg, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(b.visibility.cpu().numpy(), vis)


@pytest.mark.parametrize("N", [0, 1, 255, 256, 257, 100_003, 1_300_000])
def test_compact_equals_boolean_indexing(N):
    gen = torch.Generator(device="cuda").manual_seed(N)
    mask = torch.rand(N, device="cuda", generator=gen) < 0.41
    tensors = {"xyz": torch.randn(N, 3, device="cuda", generator=gen), "shs": torch.randn(N, 16, 3, device="cuda", generator=gen),
               "flat": torch.randn(N, device="cuda", generator=gen),
               "ids": torch.randint(0, 1 << 30, (N, 2), device="cuda", generator=gen, dtype=torch.int32)}
    out = compact(mask, tensors)
    for k, t in tensors.items():
        assert torch.equal(out[k], t[mask]), k
    if N:
        for m in (torch.zeros(N, dtype=torch.bool, device="cuda"), torch.ones(N, dtype=torch.bool, device="cuda")):
            assert torch.equal(compact(m, {"x": tensors["xyz"]})["x"], tensors["xyz"][m])


def test_masks_match_oracle_at_scale():
    N = 400_000
    rng = np.random.default_rng(5)
    st = DensifyState(N, "cuda")
    acc = np.abs(rng.normal(size=N)).astype(np.float32) * 1e-3
    den = rng.integers(0, 5, size=N).astype(np.float32)
    mr = rng.integers(0, 40, size=N).astype(np.float32)
    st.pos_gradient_accum.copy_(_t(acc).view(N, 1)); st.denom.copy_(_t(den).view(N, 1)); st.max_radii2D.copy_(_t(mr))
    sc = rng.normal(-4, 1.5, size=(N, 3)).astype(np.float32); op = rng.normal(-2, 3, size=(N, 1)).astype(np.float32)
    got = st.masks(_t(sc), _t(op), 2e-4, 0.01, 3.0, 0.005)
    want = oracle.densify_masks(acc, den, mr, sc, op, 2e-4, 0.01, 3.0, 0.005, 20.0)
    for a, b in zip(got, want):
        assert (a.cpu().numpy() != b).sum() <= 8          # threshold ties under different exp implementations
