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


def test_structure_surgery_restatement_matches_reference():
    """clone / split with the optimiser state: the numpy restatement against the vectors the reference's own
    AtlasGaussianSplattingOptimizer / PointCloud methods produced (tests/golden/make_golden_structure.py)"""
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "structure_3000.npz")))
    names = ["position", "scaling", "rotation", "opacity", "features"]
    st = lambda tag: ({n: g[f"{tag}_{n}"] for n in names}, {n: (g[f"{tag}_{n}_exp_avg"], g[f"{tag}_{n}_exp_avg_sq"]) for n in names})
    p0, m0 = st("s0")
    p1, m1 = oracle.structure_clone(p0, m0, g["clone_mask"])
    for n in names:
        np.testing.assert_array_equal(p1[n], g[f"s1_{n}"])
        np.testing.assert_array_equal(m1[n][0], g[f"s1_{n}_exp_avg"])
        np.testing.assert_array_equal(m1[n][1], g[f"s1_{n}_exp_avg_sq"])
    p2, m2, new_pos, new_scl = oracle.structure_split(p1, m1, g["split_mask"], int(g["split_num"]), g["unit_normals"])
    np.testing.assert_allclose(new_pos, g["new_pos"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(new_scl, g["new_scaling"], rtol=2e-6, atol=2e-6)
    for n in names:
        assert p2[n].shape == g[f"s2_{n}"].shape
        np.testing.assert_allclose(p2[n], g[f"s2_{n}"], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(m2[n][0], g[f"s2_{n}_exp_avg"])
    assert int(g["clone_mask"].sum()) > 100 and int(g["split_mask"].sum()) > 100
