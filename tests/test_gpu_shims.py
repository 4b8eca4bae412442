"""the reference's LITERAL call lines through the third-party import names the shims serve, against brute
force.  `src/geometry_utils.py:3,15-17`: ``from pytorch3d.ops import knn_points``; ``knn_res = knn_points(points[None],
points[None], None, None, K=K+1)``; ``knn_res.dists[0, :, 1:]``, ``knn_res.idx[0, :, 1:]`` (then edited IN PLACE at :20-22).
`src/pointrix/utils/gaussian_points/gaussian_utils.py:5,70`: ``from simple_knn._C import distCUDA2``;
``distCUDA2(position.cuda())``."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
SHIMS = os.path.join(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")), "shims")


@pytest.fixture(autouse=True)
def shim_path():
    sys.path.insert(1, SHIMS)
    yield
    sys.path.remove(SHIMS)


@pytest.mark.parametrize("Nv,K", [(4000, 5), (4000, 10), (60_000, 5)])
def test_geometry_utils_call_lines(Nv, K):
    from pytorch3d.ops import knn_points
    rng = np.random.default_rng(Nv + K)
    pts = np.concatenate([rng.uniform(-1, 1, size=(Nv, 2)), rng.uniform(0.1, 1.0, size=(Nv, 1))], 1).astype(np.float32)
    points = torch.tensor(pts, device="cuda", requires_grad=True)       # the trainer passes render_dict["position"] (requires grad)
    radius, least_edge_num = 0.1, 3
    knn_res = knn_points(points[None], points[None], None, None, K=K+1)
    nn_dist, nn_idx = knn_res.dists[0, :, 1:], knn_res.idx[0, :, 1:]
    assert nn_idx.dtype == torch.int64 and nn_dist.shape == (Nv, K) and nn_dist.requires_grad
    sub = rng.choice(Nv, 1500, replace=False)
    od, oi = oracle.knn_points(pts[sub], pts, K + 1)
    np.testing.assert_allclose(nn_dist.detach().cpu().numpy()[sub], od[:, 1:], rtol=3e-6, atol=1e-12)
    assert (nn_idx.cpu().numpy()[sub] == oi[:, 1:]).mean() > 0.999
    assert (knn_res.idx[0, :, 0] == torch.arange(Nv, device="cuda")).all()
    # the in-place edits the reference applies to the two views (geometry_utils.py:20-22) must be legal on what we return
    nn_idx[:, least_edge_num:] = torch.where(nn_dist[:, least_edge_num:] < radius ** 2, nn_idx[:, least_edge_num:], - torch.ones_like(nn_idx[:, least_edge_num:]))
    nn_dist[:, least_edge_num:] = torch.where(nn_dist[:, least_edge_num:] < radius ** 2, nn_dist[:, least_edge_num:], torch.ones_like(nn_dist[:, least_edge_num:]) * torch.inf)
    # (the reference never differentiates the masked entries -- exp(-inf) has no finite gradient; the unmasked columns do carry one)
    weight = torch.exp(-nn_dist[:, :least_edge_num] / nn_dist[:, :least_edge_num].mean())
    weight.sum().backward()
    assert torch.isfinite(points.grad).all() and points.grad.abs().sum() > 0


def test_gaussian_utils_call_line():
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(25_000, 3)).astype(np.float32)
    position = torch.tensor(pts)
    avg_dist = torch.clamp_min(distCUDA2(position.cuda()), 0.0000001)[..., None].cpu()
    od, _ = oracle.knn_points(pts, pts, 4)
    np.testing.assert_allclose(avg_dist[:, 0].numpy(), od[:, 1:].mean(axis=1), rtol=3e-6, atol=1e-12)
    assert avg_dist.shape == (25_000, 1)


def test_full_n_knn_at_c2_size_matches_brute_force_on_a_sample():
    """what the unchanged trainer pays per step (trainer_fragGS.py:672): all 300k Gaussians, K = 5 (+ self)"""
    from pytorch3d.ops import knn_points
    from splatter_a_video_amd.synth import make_scene
    sc = make_scene(300_000, 854, 480, C=3, seed=1234)
    pts = sc.positions(3)
    points = torch.tensor(pts, device="cuda")
    knn_res = knn_points(points[None], points[None], None, None, K=6)
    sub = np.random.default_rng(0).choice(300_000, 2000, replace=False)
    od, oi = oracle.knn_points(pts[sub], pts, 6)
    np.testing.assert_allclose(knn_res.dists[0].cpu().numpy()[sub], od, rtol=3e-6, atol=1e-12)
    assert (knn_res.idx[0].cpu().numpy()[sub] == oi).mean() > 0.999
