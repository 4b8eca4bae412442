"""As-rigid-as-possible energy of a node sequence on the device (second half).

``cal_arap_error`` has the reference's signature and result (reference: src/geometry_utils.py:90-123, with
``estimate_rotation`` :50-87 and ``produce_edge_matrix_nfmt`` :41-48): per sampled vertex and frame t >= 1 the rotation that
best maps the source edges onto the target edges (covariance -> SVD -> W U^T with the reflection fix, estimated without
gradient), then ``sum_k w_k |e_tgt_k - R e_src_k|^2``, summed and divided by the number of frames.  The reference runs ~50
eager launches per frame pair (scatter, bmm, torch.svd, det, ...); here it is ONE launch (``splat_arap_energy``: a thread
per (vertex, frame), 3x3 SVD by Jacobi rotations), forward and gradient together.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib as L


def neighbour_table(ii: Tensor, jj: Tensor, nn: Tensor, Nv: int, K: int) -> Tensor:
    """[Nv,K] int32 table of neighbour ids (-1: no edge) from the reference's edge lists (vertex, neighbour, slot)"""
    nbr = torch.full((Nv, K), -1, dtype=torch.int32, device=ii.device)
    nbr[ii.long(), nn.long()] = jj.to(torch.int32)
    return nbr


class _Arap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nodes, nbr, weight, sample_idx):
        nodes_c = L.need(nodes, "nodes_sequence")
        Nt, Nv, _ = nodes_c.shape
        K, S = nbr.shape[1], sample_idx.numel()
        energy = torch.zeros(1, dtype=torch.float32, device=nodes_c.device)
        need_grad = ctx.needs_input_grad[0]
        d_nodes = torch.zeros_like(nodes_c) if need_grad else None
        L.check(L.lib().splat_arap_energy(L.ci(Nt), L.ci(Nv), L.ci(K), L.ci(S), L.ptr(nodes_c), L.ptr(nbr), L.ptr(weight),
                                          L.ptr(sample_idx), L.ptr(energy), L.ptr(d_nodes), L.ptr(None), L.stream()))
        ctx.Nt = Nt
        if need_grad:
            ctx.save_for_backward(d_nodes)
        return energy[0] / Nt

    @staticmethod
    def backward(ctx, g):
        (d_nodes,) = ctx.saved_tensors
        return d_nodes * (g / ctx.Nt), None, None, None


def cal_arap_error(nodes_sequence: Tensor, ii: Tensor, jj: Tensor, nn: Tensor, K: int = 10, weight: Optional[Tensor] = None,
                   sample_num: int = 512, sample_idx: Optional[Tensor] = None) -> Tensor:
    """``nodes_sequence`` [Nt,Nv,3]; ``ii, jj, nn`` the edge lists of ``cal_connectivity_from_points``; ``weight`` [Nv,K] or
    None (1 per edge).  ``sample_idx`` fixes the sampled vertices (default: ``np.random.choice(Nv, sample_num)`` when
    Nv > sample_num, as the reference draws them, else all vertices)."""
    if nodes_sequence.dim() != 3 or nodes_sequence.shape[2] != 3:
        raise ValueError("nodes_sequence must be [Nt, Nv, 3]")
    if not 1 <= K <= 16:
        raise ValueError("K must be in 1..16")
    Nt, Nv, _ = nodes_sequence.shape
    dev = nodes_sequence.device
    nbr = neighbour_table(ii, jj, nn, Nv, K)
    w = None
    if weight is not None:
        w = L.need(weight.detach(), "weight")
        if tuple(w.shape) != (Nv, K):
            raise ValueError("weight must be [Nv, K]")
    if sample_idx is None:
        sample_idx = (torch.from_numpy(np.random.choice(Nv, sample_num)).long().to(dev) if Nv > sample_num
                      else torch.arange(Nv, device=dev))
    sample_idx = L.need(sample_idx.long(), "sample_idx", torch.int64)
    return _Arap.apply(nodes_sequence, nbr, w, sample_idx)


def arap_rotations(nodes_sequence: Tensor, nbr: Tensor, weight: Optional[Tensor], sample_idx: Tensor) -> Tensor:
    """[Nt-1,S,3,3]: the rotation ``estimate_rotation`` returns for every frame t >= 1 and sampled vertex"""
    nodes_c = L.need(nodes_sequence.detach(), "nodes_sequence")
    Nt, Nv, _ = nodes_c.shape
    S = sample_idx.numel()
    rot = torch.empty(max(Nt - 1, 0), S, 3, 3, dtype=torch.float32, device=nodes_c.device)
    energy = torch.zeros(1, dtype=torch.float32, device=nodes_c.device)
    L.check(L.lib().splat_arap_energy(L.ci(Nt), L.ci(Nv), L.ci(nbr.shape[1]), L.ci(S), L.ptr(nodes_c), L.ptr(nbr), L.ptr(weight),
                                      L.ptr(L.need(sample_idx.long(), "sample_idx", torch.int64)), L.ptr(energy), L.ptr(None),
                                      L.ptr(rot), L.stream()))
    return rot


def pair_connectivity(nodes: Tensor, sample_idx: Tensor, K: int = 5, radius: float = 0.1, least_edge_num: int = 3) -> Tensor:
    """Neighbour rows of the SAMPLED vertices only: what ``cal_connectivity_from_points(points, K=K)`` (reference:
    src/geometry_utils.py:7-38 -- K + 1 nearest neighbours, the vertex itself dropped, neighbours past ``least_edge_num`` cut
    at ``radius``) gives for the rows ``sample_idx`` -- the only rows ``cal_arap_error`` reads.  ``nodes`` [B, Nv, 3] (the
    pairs' first frames), ``sample_idx`` [B, S]; returns nbr [B, S, K] int32 (vertex ids, -1 = no edge)."""
    from .knn import knn_brute_batch
    d, i = knn_brute_batch(nodes, sample_idx, K + 1)
    d, i = d[:, :, 1:], i[:, :, 1:]
    cut = d >= radius * radius
    cut[:, :, :least_edge_num] = False
    return torch.where(cut, torch.full_like(i, -1), i).contiguous()


def pair_arap(pairs: Tensor, sample_idx: Tensor, nbr: Tensor, K_table: int = 10, d_pairs: Optional[Tensor] = None,
              grad_scale: float = 1.0) -> Tensor:
    """``cal_arap_error`` of B node PAIRS in one launch: ``pairs`` [B, 2, Nv, 3] (position(ids1), position(ids2) of every pair,
    contiguous), ``sample_idx`` [B, S], ``nbr`` [B, S, K] from ``pair_connectivity`` (edge weights 1, the edge table padded to
    ``K_table`` slots like the reference's default ``K = 10`` -- empty slots contribute nothing).  Returns the B energies
    (each divided by Nt = 2, as the reference does); ``d_pairs`` (zero-init buffer of ``pairs``' shape) receives
    ``grad_scale`` x the gradient of the UNDIVIDED energy / Nt by ADDITION (raw operator: no autograd)."""
    if pairs.dim() != 4 or pairs.shape[1] != 2 or pairs.shape[3] != 3 or not pairs.is_contiguous():
        raise ValueError("pairs must be a contiguous [B, 2, Nv, 3] tensor")
    pairs = L.need(pairs, "pairs")
    B, _, Nv, _ = pairs.shape
    S, K = nbr.shape[1], nbr.shape[2]
    nbr = L.need(nbr, "nbr", torch.int32)
    sample_idx = L.need(sample_idx, "sample_idx", torch.int64)
    if tuple(sample_idx.shape) != (B, S) or nbr.shape[0] != B:
        raise ValueError("sample_idx must be [B, S] and nbr [B, S, K]")
    if d_pairs is not None and (tuple(d_pairs.shape) != tuple(pairs.shape) or not d_pairs.is_contiguous()):
        raise ValueError("d_pairs must be a contiguous buffer of pairs' shape")
    energy = torch.zeros(B, dtype=torch.float32, device=pairs.device)
    L.check(L.lib().splat_arap_energy_batch(L.ci(B), L.ci(2), L.ci(Nv), L.ci(K), L.ci(S), L.ptr(pairs), ctypes.c_int64(2 * Nv * 3),
                                            L.ptr(nbr), L.ptr(None), L.ptr(sample_idx), L.ptr(energy), L.ptr(d_pairs),
                                            L.cf(grad_scale / 2.0), L.stream()))
    return energy / 2.0
