"""splat_arap_energy (splatter_a_video_amd.arap.cal_arap_error: one launch, 3x3 SVD by Jacobi rotations) against the
vectors of the reference's own cal_arap_error / estimate_rotation (tests/golden/make_golden_arap.py) and against the
numpy oracle on a larger case."""
import os

import numpy as np
import pytest
import torch

from splatter_a_video_amd.arap import arap_rotations, cal_arap_error, neighbour_table

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "arap_2000.npz")


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.mark.parametrize("tag", ["unit", "weighted"])
def test_arap_matches_reference(tag):
    g = dict(np.load(G))
    K = int(g["K"])
    x = _t(g["nodes"]).requires_grad_(True)
    w = None if tag == "unit" else _t(g["weight"])
    sidx = _t(g[f"{tag}_sample_idx"])
    err = cal_arap_error(x, _t(g["ii"]), _t(g["jj"]), _t(g["nn"]), K=K, weight=w, sample_idx=sidx)
    assert abs(float(err) - float(g[f"{tag}_error"])) < 5e-5 * abs(float(g[f"{tag}_error"]))
    err.backward()
    ref = g[f"{tag}_grad"]
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-3, atol=5e-5 * float(np.abs(ref).max()))
    nbr = neighbour_table(_t(g["ii"]), _t(g["jj"]), _t(g["nn"]), x.shape[1], K)
    wt = w if w is not None else (nbr >= 0).float()
    rot = arap_rotations(x, nbr, wt, sidx)
    np.testing.assert_allclose(rot[0].cpu().numpy(), g[f"{tag}_rot1"], rtol=0, atol=5e-5)
    # frame 3 moves in the xy plane only (z edges exactly unchanged): the reference zeroes S there -> R = I for every vertex
    np.testing.assert_allclose(rot[2].cpu().numpy(), g[f"{tag}_rot3"], rtol=0, atol=5e-5)
    assert np.abs(g[f"{tag}_rot3"] - np.eye(3)).max() < 1e-6
    assert float(torch.linalg.det(rot.double()).min()) > 0.99


def test_arap_default_sampling_and_large_case(oracle_mod):
    rng = np.random.default_rng(3)
    Nv, K, Nt = 30000, 6, 4
    base = rng.uniform(-1, 1, size=(Nv, 3)).astype(np.float32)
    nbr = rng.integers(0, Nv, size=(Nv, K)).astype(np.int32)
    nbr[rng.random((Nv, K)) < 0.2] = -1
    nodes = np.stack([base + 0.05 * t * rng.normal(size=(Nv, 3)).astype(np.float32) for t in range(Nt)]).astype(np.float32)
    ii, nn = np.nonzero(nbr >= 0)
    jj = nbr[ii, nn]
    sidx = rng.integers(0, Nv, size=700)
    x = _t(nodes).requires_grad_(True)
    err = cal_arap_error(x, _t(ii), _t(jj.astype(np.int64)), _t(nn), K=K, sample_idx=_t(sidx))
    err.backward()
    e, grad, _ = oracle_mod.arap_energy(nodes, nbr, None, sidx)
    assert abs(float(err) - float(e)) < 1e-4 * float(e)
    np.testing.assert_allclose(x.grad.cpu().numpy(), grad, rtol=2e-3, atol=1e-4 * float(np.abs(grad).max()))
    # reference-style call: the vertices are drawn with np.random.choice when there are more than sample_num
    np.random.seed(5)
    want_idx = np.random.choice(Nv, 512)
    np.random.seed(5)
    e1 = cal_arap_error(_t(nodes), _t(ii), _t(jj.astype(np.int64)), _t(nn), K=K)
    e2 = cal_arap_error(_t(nodes), _t(ii), _t(jj.astype(np.int64)), _t(nn), K=K, sample_idx=_t(want_idx))
    assert abs(float(e1) - float(e2)) < 1e-5 * float(e2)          # (float atomics: the order of the sum varies)
