"""Golden vectors for the ARAP energy from the reference's own ``cal_arap_error`` /
``estimate_rotation`` / ``produce_edge_matrix_nfmt`` (src/geometry_utils.py:41-123), run on CPU in the build container: the
module's ``pytorch3d`` import is stubbed (not used by these functions) and its hard-coded ``.cuda()`` / ``device="cuda"``
are mapped to the CPU; the vertices ``np.random.choice`` samples are recovered by re-seeding.  Values AND autograd
gradients.  Data only travels.

    python tests/golden/make_golden_arap.py    ->  tests/golden/arap_2000.npz
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    m = types.ModuleType("pytorch3d"); sys.modules["pytorch3d"] = m
    ops = types.ModuleType("pytorch3d.ops"); ops.knn_points = None; sys.modules["pytorch3d.ops"] = ops
    torch.Tensor.cuda = lambda self, *a, **k: self                      # CPU-only container
    spec = importlib.util.spec_from_file_location("ref_geometry_utils", os.path.join(REF, "geometry_utils.py"))
    gu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gu)
    _orig = gu.produce_edge_matrix_nfmt
    gu.produce_edge_matrix_nfmt = lambda verts, shape, ii, jj, nn, device="cpu": _orig(verts, shape, ii, jj, nn, device="cpu")

    rng = np.random.default_rng(11)
    Nv, K, Kc, Nt, SN = 2000, 10, 5, 4, 256
    base = rng.uniform(-1, 1, size=(Nv, 3)).astype(np.float32)
    # connectivity as cal_connectivity_from_points builds it (K = 5 neighbours in a K = 10 edge matrix; some slots cut)
    d2 = ((base[:, None, :] - base[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1)[:, 1:Kc + 1]
    keep = np.ones((Nv, Kc), bool)
    keep[:, 3:] = np.take_along_axis(d2, order, 1)[:, 3:] < 0.02
    ii = np.repeat(np.arange(Nv), Kc).reshape(Nv, Kc)[keep]
    nn = np.tile(np.arange(Kc), (Nv, 1))[keep]
    jj = order[keep]
    # frames: rigidly rotated + deformed copies; a block of vertices stays exactly in place in frame 1 (the S = 0 branch)
    def rot(ax, ang):
        c, s = np.cos(ang), np.sin(ang)
        R = np.eye(3, dtype=np.float32); a, b = [(1, 2), (0, 2), (0, 1)][ax]
        R[a, a], R[a, b], R[b, a], R[b, b] = c, -s, s, c
        return R
    f1 = base @ rot(2, 0.4).T + 0.02 * rng.normal(size=(Nv, 3)).astype(np.float32)
    f1[:150] = base[:150]
    f2 = (base * np.array([1.0, 1.0, -1.0], np.float32)) @ rot(0, 1.1).T + 0.05 * rng.normal(size=(Nv, 3)).astype(np.float32)
    # frame 3: planar motion -- rotation about z, every z coordinate (hence every edge's z component) exactly unchanged: the
    # reference's `(source_edge == target_edge).all(dim=1)` is true on that axis, so it zeroes S (R = I) for EVERY vertex
    f3 = (base @ rot(2, 0.7).T).astype(np.float32)
    f3[:, 2] = base[:, 2]
    nodes = np.stack([base, f1.astype(np.float32), f2.astype(np.float32), f3])
    weight = rng.uniform(0.2, 1.0, size=(Nv, K)).astype(np.float32)
    out = dict(nodes=nodes, ii=ii.astype(np.int64), jj=jj.astype(np.int64), nn=nn.astype(np.int64), K=np.int32(K), weight=weight,
               sample_num=np.int32(SN))
    T = lambda a: torch.tensor(a)
    for tag, w in (("unit", None), ("weighted", weight)):
        SEED = 77
        np.random.seed(SEED)
        sample_idx = np.random.choice(Nv, SN)
        np.random.seed(SEED)
        x = T(nodes).requires_grad_(True)
        err = gu.cal_arap_error(x, T(ii), T(jj), T(nn), K=K, weight=None if w is None else T(w), sample_num=SN)
        err.backward()
        with torch.no_grad():
            R1 = gu.estimate_rotation(x[0], x[1], T(ii), T(jj), T(nn), K=K,
                                      weight=(torch.zeros(Nv, K).index_put_((T(ii), T(nn)), torch.ones(len(ii))) if w is None else T(w))[sample_idx],
                                      sample_idx=T(sample_idx))
            R3 = gu.estimate_rotation(x[0], x[3], T(ii), T(jj), T(nn), K=K,
                                      weight=(torch.zeros(Nv, K).index_put_((T(ii), T(nn)), torch.ones(len(ii))) if w is None else T(w))[sample_idx],
                                      sample_idx=T(sample_idx))
        out.update({f"{tag}_sample_idx": sample_idx.astype(np.int64), f"{tag}_error": np.float32(err.item()),
                    f"{tag}_grad": x.grad.numpy(), f"{tag}_rot1": R1.numpy(), f"{tag}_rot3": R3.numpy()})
        print(tag, float(err), float(x.grad.abs().max()))
    np.savez_compressed(os.path.join(HERE, "arap_2000.npz"), **out)


if __name__ == "__main__":
    main()
