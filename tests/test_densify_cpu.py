"""SURVEY 8(f) rank 2 (densification statistics): the C oracle pinned to vectors produced by the reference's own
optimizer methods (tests/golden/make_golden_densify.py)."""
import os

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(__file__), "golden", "densify_5000.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(G))


def test_batch_reduction_and_statistics_match_reference(gold):
    g = gold
    N, F, STEPS = int(g["N"]), int(g["F"]), int(g["STEPS"])
    max_r = np.zeros(N, np.float32); accum = np.zeros(N, np.float32); denom = np.zeros(N, np.float32)
    for s in range(STEPS):
        vg = np.zeros((N, 2), np.float32); vis = np.zeros(N, np.uint8); radii = np.zeros(N, np.int32)
        for f in range(F):
            oracle.densify_accumulate(g[f"s{s}_radius"][f], g[f"s{s}_taps"][f], 1.0, 1.0, vg, vis, radii)
        np.testing.assert_array_equal(vis.astype(bool), g[f"s{s}_visibility"])
        np.testing.assert_array_equal(radii, g[f"s{s}_radii"])
        np.testing.assert_array_equal(vg, g[f"s{s}_viewspace_grad"])       # same summation order: bit exact
        oracle.densify_update(vis, vg, radii, max_r, accum, denom)
        np.testing.assert_array_equal(max_r, g[f"s{s}_max_radii2D"])
        np.testing.assert_array_equal(denom, g[f"s{s}_denom"].reshape(-1))
        np.testing.assert_allclose(accum, g[f"s{s}_pos_gradient_accum"].reshape(-1), rtol=2e-6, atol=0)


def test_masks_match_reference(gold):
    g = gold
    s = int(g["STEPS"]) - 1
    clone, split, prune = oracle.densify_masks(g[f"s{s}_pos_gradient_accum"], g[f"s{s}_denom"], g[f"s{s}_max_radii2D"],
                                               g["scaling_raw"], g["opacity_raw"], float(g["densify_grad_threshold"]),
                                               float(g["percent_dense"]), float(g["cameras_extent"]), float(g["min_opacity"]), 20.0)
    # exp / sigmoid of libm vs torch can flip a comparison that sits within an ulp of its threshold
    assert (clone != g["clone_mask"]).sum() <= 2 and (split != g["split_mask"]).sum() <= 2
    assert (prune != ~g["prune_valid_mask"]).sum() <= 2
    assert clone.sum() > 100 and split.sum() > 100 and 100 < prune.sum() < prune.size - 100
    assert not (clone & split).any()


def test_compact_rows_is_boolean_indexing():
    rng = np.random.default_rng(0)
    for shape in ((1000,), (777, 3), (500, 16, 3)):
        src = rng.normal(size=shape).astype(np.float32)
        mask = rng.random(shape[0]) < 0.37
        np.testing.assert_array_equal(oracle.compact_rows(mask, src), src[mask])
    idx = rng.integers(0, 1 << 30, size=(300, 2)).astype(np.int32)
    m = np.zeros(300, bool)
    assert oracle.compact_rows(m, idx).shape[0] == 0
    np.testing.assert_array_equal(oracle.compact_rows(~m, idx), idx)
