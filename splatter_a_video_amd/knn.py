"""Exact K nearest neighbours on the GPU.

``knn_points`` mirrors the part of ``pytorch3d.ops.knn_points`` the reference uses
(reference: src/geometry_utils.py:17-19 -- ``knn_points(points[None], points[None], None, None, K=K+1)``; pytorch3d is an
un-vendored CUDA dependency without a ROCm build): batched point sets ``[B, N, 3]``, result fields ``dists`` (squared
Euclidean, ascending), ``idx`` (int64) and ``knn`` (None unless ``return_nn``).  Uniform-grid search, exact; ties between
equal distances resolve to the smaller index.  ``dists`` (and ``knn``) carry gradients to ``p1`` / ``p2`` when those
require grad, as pytorch3d's do.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from collections import namedtuple
from typing import Optional

import torch
from torch import Tensor

from . import _lib as L

KNN = namedtuple("KNN", "dists idx knn")


def _knn_single(query: Tensor, points: Tensor, K: int):
    lib = L.lib()
    dev = points.device
    N, M = query.shape[0], points.shape[0]
    dists = torch.empty(N, K, dtype=torch.float32, device=dev)
    idx = torch.empty(N, K, dtype=torch.int32, device=dev)
    if N == 0:
        return dists, idx.long()
    budget = int(lib.splat_knn_grid_cells(L.ci(M)))
    plan = torch.empty(int(lib.splat_knn_plan_bytes()), dtype=torch.uint8, device=dev)
    cell_of = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(budget + 1, dtype=torch.int32, device=dev)
    L.check(lib.splat_knn_build(L.ci(M), L.ptr(points), L.ci(budget), L.ptr(plan), L.ptr(cell_of), L.ptr(count), L.stream()))
    cell_start = (torch.cumsum(count, 0, dtype=torch.int32) - count).contiguous()      # exclusive; entry [budget] = M
    fill = torch.zeros(budget, dtype=torch.int32, device=dev)
    sorted_pts = torch.empty(max(M, 1), 4, dtype=torch.float32, device=dev)
    L.check(lib.splat_knn_scatter(L.ci(M), L.ptr(points), L.ptr(cell_of), L.ptr(cell_start), L.ptr(fill), L.ptr(sorted_pts),
                                  L.stream()))
    # walk the queries cell by cell (cache locality) when they are the point set itself
    order = None
    if query.data_ptr() == points.data_ptr() and N == M:
        order = sorted_pts[:, 3].contiguous().view(torch.int32)
    L.check(lib.splat_knn_search(L.ci(N), L.ptr(query), L.ptr(order), L.ci(M), L.ptr(sorted_pts), L.ptr(cell_start),
                                 L.ptr(plan), L.ci(K), L.ptr(dists), L.ptr(idx), L.stream()))
    return dists, idx.long()


def knn_points(p1: Tensor, p2: Tensor, lengths1: Optional[Tensor] = None, lengths2: Optional[Tensor] = None, K: int = 1,
               version: int = -1, return_nn: bool = False, return_sorted: bool = True) -> KNN:
    """K nearest neighbours in ``p2`` of every point of ``p1`` ([B,N,3], [B,M,3]); ``lengths*`` must be None (the
    reference passes None); results are always sorted."""
    if lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("ragged batches (lengths1 / lengths2) are not supported")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[2] != 3 or p2.shape[2] != 3 or p1.shape[0] != p2.shape[0]:
        raise ValueError("p1 and p2 must be [B, N, 3] and [B, M, 3]")
    if not 1 <= K <= 16:
        raise ValueError("K must be in 1..16")
    same = p1.data_ptr() == p2.data_ptr() and p1.shape == p2.shape
    p1c = L.need(p1.detach(), "p1")
    p2c = p1c if same else L.need(p2.detach(), "p2")
    d, i = [], []
    for b in range(p1c.shape[0]):
        db, ib = _knn_single(p1c[b], p2c[b], K)
        d.append(db); i.append(ib)
    dists, idx = torch.stack(d), torch.stack(i)
    if (p1.requires_grad or p2.requires_grad) and torch.is_grad_enabled():
        # pytorch3d's dists are differentiable w.r.t. both point sets (the neighbour choice is not): rebuild them from
        # the gathered neighbours through autograd; the search itself ran on detached coordinates
        gathered = torch.stack([p2[b][idx[b].clamp(min=0)] for b in range(p1.shape[0])])
        dists = ((p1[:, :, None, :] - gathered) ** 2).sum(-1)
    nn = None
    if return_nn:
        nn = torch.stack([p2[b][idx[b].clamp(min=0)] for b in range(p1.shape[0])])
    return KNN(dists=dists, idx=idx, knn=nn)


def knn_brute_batch(points: Tensor, query_idx: Tensor, K: int):
    """K <= 8 nearest points of a FEW query vertices in each of B point sets, brute force in two launches: ``points`` [B, N, 3]
    (any batch stride, dense rows), ``query_idx`` [B, S] int64 = indices of the query vertices in their own set; returns
    (dists [B, S, K], idx [B, S, K] int32) ascending, ties -> smaller index, the query itself included (distance 0) -- rows
    ``query_idx`` of ``knn_points(points, points, K=K)``.  What the ARAP term of a training batch needs: the neighbours of its
    512 sampled vertices (src/geometry_utils.py:17-19,98-101), for which a grid build per point set would cost more than the
    scan.  No gradient (the reference's ARAP takes none through the neighbour choice)."""
    if points.dim() != 3 or points.shape[2] != 3 or query_idx.dim() != 2 or query_idx.shape[0] != points.shape[0]:
        raise ValueError("points must be [B, N, 3] and query_idx [B, S]")
    if not 1 <= K <= 8:
        raise ValueError("K must be in 1..8")
    p = points.detach()
    if not (p.is_cuda and p.dtype == torch.float32 and p.stride(2) == 1 and p.stride(1) == 3):
        p = L.need(p, "points")
    else:
        L.need(p[0], "points")
    q = L.need(query_idx, "query_idx", torch.int64)
    B, N, S = p.shape[0], p.shape[1], q.shape[1]
    lib = L.lib()
    dev = p.device
    dists = torch.empty(B, S, K, dtype=torch.float32, device=dev)
    idx = torch.empty(B, S, K, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(int(lib.splat_knn_brute_scratch_bytes(B, N, S)), 4), dtype=torch.uint8, device=dev)
    L.check(lib.splat_knn_brute_batch(L.ci(B), L.ci(N), L.ci(S), L.ci(K), L.ptr(p), ctypes.c_int64(p.stride(0) if B > 1 else N * 3),
                                      L.ptr(q), L.ptr(dists), L.ptr(idx), L.ptr(scratch), L.stream()))
    return dists, idx


def distCUDA2(points: Tensor) -> Tensor:
    """Mean squared distance of every point to its three nearest neighbours, the quantity ``simple_knn._C.distCUDA2``
    returns (reference use: src/pointrix/utils/gaussian_points/gaussian_utils.py:5,68-73, initial Gaussian scales;
    simple_knn is an un-vendored CUDA dependency: parity unpinned, semantics from its published kernel
    ``simple_knn.cu`` -- exact 3-NN excluding the point itself, mean of the squared distances)."""
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [N, 3]")
    p = L.need(points.detach(), "points")
    d = knn_points(p[None], p[None], None, None, K=4).dists[0]      # column 0 is the point itself
    return d[:, 1:].mean(dim=1)
