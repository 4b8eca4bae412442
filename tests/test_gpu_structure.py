"""Clone / split with the optimiser state on the device (splatter_a_video_amd.densify) against the golden vectors the
reference's OWN methods produced (tests/golden/make_golden_structure.py: AtlasGaussianSplattingOptimizer.densify_clone /
densify_split / new_pos_scale and PointCloud.extend_optimizer / prune_optimizer driving a real torch.optim.Adam)."""
import os

import numpy as np
import pytest
import torch

from splatter_a_video_amd import densify as D

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "structure_3000.npz")
NAMES = ["position", "scaling", "rotation", "opacity", "features"]


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def _state(g, tag):
    p = {n: _t(g[f"{tag}_{n}"]) for n in NAMES}
    m = {n: (_t(g[f"{tag}_{n}_exp_avg"]), _t(g[f"{tag}_{n}_exp_avg_sq"])) for n in NAMES}
    return p, m


def _assert_state(p, m, g, tag, exact=True):
    for n in NAMES:
        for got, key in ((p[n], f"{tag}_{n}"), (m[n][0], f"{tag}_{n}_exp_avg"), (m[n][1], f"{tag}_{n}_exp_avg_sq")):
            want = g[key]
            assert tuple(got.shape) == want.shape, key
            if exact:
                assert np.array_equal(got.cpu().numpy(), want), key
            else:
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-6, err_msg=key)


def test_clone_and_split_match_the_reference_methods():
    g = dict(np.load(G))
    split_num = int(g["split_num"])
    p0, m0 = _state(g, "s0")
    p1, m1, n_clone = D.densify_clone(p0, m0, _t(g["clone_mask"]))
    assert n_clone == int(g["clone_mask"].sum())
    _assert_state(p1, m1, g, "s1")                                       # pure data movement: bit-exact
    # children of the split, fed with the reference's own normal draws
    new_pos, new_scl, n_split = D.split_children(p1["position"], p1["scaling"], p1["rotation"], _t(g["split_mask"]), split_num,
                                                 unit_normals=_t(g["unit_normals"]))
    assert n_split == int(g["split_mask"].sum())
    np.testing.assert_allclose(new_pos.cpu().numpy(), g["new_pos"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(new_scl.cpu().numpy(), g["new_scaling"], rtol=2e-6, atol=2e-6)
    p2, m2, valid, _ = D.densify_split(p1, m1, _t(g["split_mask"]), split_num, unit_normals=_t(g["unit_normals"]))
    _assert_state(p2, m2, g, "s2", exact=False)
    for n in ("rotation", "opacity", "features"):                        # repeated attributes and all moments: bit-exact
        assert np.array_equal(p2[n].cpu().numpy(), g[f"s2_{n}"])
        assert np.array_equal(m2[n][0].cpu().numpy(), g[f"s2_{n}_exp_avg"])
    assert int(valid.sum()) == g["s2_position"].shape[0]


def test_split_sampling_is_counter_based():
    """same seed -> same children whatever else ran before (what keeps data-parallel replicas identical); unit variance"""
    g = dict(np.load(G))
    p1, _ = _state(g, "s1")
    mask = _t(g["split_mask"])
    a = D.split_children(p1["position"], p1["scaling"], p1["rotation"], mask, 2, seed=77)
    torch.randn(1000, device="cuda")                                      # the global generator moving must not matter
    b = D.split_children(p1["position"], p1["scaling"], p1["rotation"], mask, 2, seed=77)
    c = D.split_children(p1["position"], p1["scaling"], p1["rotation"], mask, 2, seed=78)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and not torch.equal(a[0], c[0])
    # a large selection, identity rotation, unit scales: the offsets ARE the normals
    N = 200000
    pos = torch.zeros(N, 3, device="cuda"); scl = torch.zeros(N, 3, device="cuda")
    rot = torch.zeros(N, 4, device="cuda"); rot[:, 0] = 1.0
    z, s, n = D.split_children(pos, scl, rot, torch.ones(N, dtype=torch.bool, device="cuda"), 2, seed=5)
    assert n == N and z.shape == (2 * N, 3)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z[:, 0] * z[:, 1]).mean())) < 5e-3 and abs(float((z[:N] * z[N:]).mean())) < 5e-3
    assert torch.allclose(s, torch.full_like(s, float(np.log(1.0 / 1.6))))


def test_prune_points_keeps_rows_and_moments():
    g = dict(np.load(G))
    p0, m0 = _state(g, "s0")
    keep = torch.rand(p0["position"].shape[0], device="cuda") < 0.6
    p, m = D.prune_points(p0, m0, keep)
    for n in NAMES:
        assert torch.equal(p[n], p0[n][keep]) and torch.equal(m[n][0], m0[n][0][keep]) and torch.equal(m[n][1], m0[n][1][keep])
