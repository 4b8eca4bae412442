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


def _morton_np(uv, W, H):
    def spread(v):
        v = v.astype(np.uint32)
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    fx = np.clip(uv[:, 0].astype(np.float32) * np.float32(32768.0 / W), 0, 32767)
    fy = np.clip(uv[:, 1].astype(np.float32) * np.float32(32768.0 / H), 0, 32767)
    return spread(fx.astype(np.uint32)) | (spread(fy.astype(np.uint32)) << 1)


def test_spatial_order_is_the_stable_morton_argsort_and_rendering_does_not_depend_on_it():
    """densify.spatial_order = stable argsort of the Z-curve code of the screen positions (outside positions clamp to the
    border); reorder_points permutes parameters and Adam moments alike; a frame batch rendered from the reordered Gaussians
    gives the same images, and the gradients of the same Gaussians."""
    from splatter_a_video_amd.frames import FrameBatch
    from splatter_a_video_amd.synth import make_scene
    import dptr.gs as gs
    N, W, H, F = 7000, 160, 96, 2
    sc = make_scene(N, W, H, seed=4)
    rng = np.random.default_rng(0)
    extr = _t(sc.extr)
    uv, _ = gs.project_point_ortho(_t(sc.xyz), extr, W, H, nearest=0.01)
    uv[:5] = torch.tensor([[-50.0, 3.0], [1e9, 2.0], [5.0, -7.0], [3.0, 1e6], [W + 10.0, H + 10.0]], device="cuda")
    perm = D.spatial_order(uv, W, H)
    want = np.argsort(_morton_np(uv.cpu().numpy(), W, H), kind="stable")
    assert np.array_equal(perm.cpu().numpy(), want)
    assert sorted(perm.tolist()) == list(range(N))

    base = dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, rgb=rng.uniform(size=(N, 3)).astype(np.float32))
    params = {k: _t(v) for k, v in base.items()}
    moments = {k: (_t(rng.normal(size=v.shape).astype(np.float32)), _t(rng.uniform(size=v.shape).astype(np.float32))) for k, v in base.items()}
    perm = D.spatial_order(gs.project_point_ortho(params["xyz"], extr, W, H, nearest=0.01)[0], W, H)
    p2, m2 = D.reorder_points(params, moments, perm)
    for k in params:
        assert torch.equal(p2[k], params[k][perm]) and torch.equal(m2[k][0], moments[k][0][perm]) and torch.equal(m2[k][1], moments[k][1][perm])

    off = _t((0.01 * rng.normal(size=(F, N, 3))).astype(np.float32))
    g = _t(rng.normal(size=(F, 3, H, W)).astype(np.float32))

    def run(p, o):
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        B = FrameBatch(F, N, W, H, 3, "cuda")
        img = B.render(q["xyz"], q["scales"], q["uquats"], q["opacity"], q["rgb"], o, extr)
        img.backward(g)
        return img.detach(), {k: v.grad for k, v in q.items()}

    i1, g1 = run(params, off)
    i2, g2 = run(p2, off[:, perm].contiguous())
    assert torch.allclose(i1, i2, rtol=1e-5, atol=1e-6)      # (summation order inside a pixel is the depth order: unchanged)
    for k in g1:
        a, b = g2[k], g1[k][perm]
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5 * float(b.abs().max()) + 1e-12), k
