"""ARAP energy: the numpy restatement pinned to vectors produced by the reference's own
cal_arap_error / estimate_rotation (tests/golden/make_golden_arap.py)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden", "arap_2000.npz")


def _nbr(g):
    Nv, K = g["nodes"].shape[1], int(g["K"])
    nbr = np.full((Nv, K), -1, np.int32)
    nbr[g["ii"], g["nn"]] = g["jj"]
    return nbr


@pytest.mark.parametrize("tag", ["unit", "weighted"])
def test_arap_restatement_matches_reference(oracle_mod, tag):
    g = dict(np.load(G))
    w = None if tag == "unit" else g["weight"]
    e, grad, rots = oracle_mod.arap_energy(g["nodes"], _nbr(g), w, g[f"{tag}_sample_idx"])
    assert abs(float(e) - float(g[f"{tag}_error"])) < 2e-5 * abs(float(g[f"{tag}_error"]))
    np.testing.assert_allclose(rots[0], g[f"{tag}_rot1"], rtol=0, atol=2e-5)
    # frame 3: planar motion, z edges exactly unchanged -> the reference's shortcut (any axis unchanged over the K edges) gives R = I
    np.testing.assert_allclose(rots[2], g[f"{tag}_rot3"], rtol=0, atol=2e-5)
    assert np.abs(g[f"{tag}_rot3"] - np.eye(3)).max() < 1e-6
    np.testing.assert_allclose(grad, g[f"{tag}_grad"], rtol=2e-4, atol=2e-5 * float(np.abs(g[f"{tag}_grad"]).max()))
    assert (np.linalg.det(rots.astype(np.float64)) > 0.99).all()  # reflections fixed (frame 2 is mirrored)
