"""Eager-torch orthographic projection and EWA on the GPU -- MEASUREMENT ONLY (bench.py --ref-flow).

The reference renderer keeps these two steps in eager torch (src/pointrix/renderer/dptr_ortho_enhanced.py:145-202
``project_point``, :18-111 ``ewa_project_torch_impl``: ~15 + ~60 small kernels per frame and their autograd graph) and only
calls native code for cov3d, the sort and the blends.  ``bench.py --ref-flow`` times that LITERAL call sequence on this
library's operators; for the two eager steps it needs a torch restatement that runs on the GPU.  This is it, written from
the arithmetic in SURVEY.md Appendix A.1 / A.3 (same operations, same masks, same integer conversions), differentiable
through autograd like the reference's.  The product path never imports it (it has fused HIP kernels for both steps:
gs.project_point_ortho / gs.ewa_project_ortho / gs.preprocess_ortho); tests/test_gpu_ref_flow.py checks it against them.
"""
from __future__ import annotations

import torch


def project_point_ortho(xyz: torch.Tensor, extr: torch.Tensor, W: int, H: int, nearest: float = 0.01, extent: float = 1.3):
    """uv [P,2], depth [P,1]; culled points (near plane, outside the extended image) are zero with zero gradient"""
    R, T = extr[:3, :3], extr[:3, 3]
    cam = xyz @ R.transpose(0, 1) + T
    u = (cam[:, 0] + 1.0) * (W * 0.5) - 0.5
    v = (cam[:, 1] + 1.0) * (H * 0.5) - 0.5
    d = torch.nan_to_num(cam[:, 2])
    near = d <= nearest
    xl, xh = (1.0 - extent) * W * 0.5, (1.0 + extent) * W * 0.5
    yl, yh = (1.0 - extent) * H * 0.5, (1.0 + extent) * H * 0.5
    out = (u < xl) | (u > xh) | (v < yl) | (v > yh)
    culled = near | out
    uv = torch.stack([u, v], dim=-1).clone()
    depth = d.unsqueeze(-1).clone()
    uv[culled] = 0.0
    depth[culled] = 0.0
    return uv, depth


def ewa_project_ortho(xyz: torch.Tensor, cov3d: torch.Tensor, extr: torch.Tensor, uv: torch.Tensor, W: int, H: int,
                      visible: torch.Tensor):
    """conic [P,3], radius [P] int32, tiles [P] int32 from the packed covariance [P,6] (xx xy xz yy yz zz)"""
    P = xyz.shape[0]
    R = extr[:3, :3]
    J = torch.zeros(2, 3, dtype=xyz.dtype, device=xyz.device)
    J[0, 0] = W * 0.5
    J[1, 1] = H * 0.5
    Tm = J @ R                                                     # [2,3], the same for every point
    S = torch.stack([cov3d[:, 0], cov3d[:, 1], cov3d[:, 2],
                     cov3d[:, 1], cov3d[:, 3], cov3d[:, 4],
                     cov3d[:, 2], cov3d[:, 4], cov3d[:, 5]], dim=-1).reshape(P, 3, 3)
    c2 = Tm.unsqueeze(0) @ S @ Tm.transpose(0, 1).unsqueeze(0)    # [P,2,2]
    a = c2[:, 0, 0] + 0.3
    b = c2[:, 0, 1]
    d = c2[:, 1, 1] + 0.3
    det = a * d - b * b
    ok = visible.reshape(-1) & (det != 0)
    mid = 0.5 * (a + d)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r = radius.to(xyz.dtype)
    ud, vd = uv[:, 0].detach(), uv[:, 1].detach()
    x0 = ((ud - r) / 16.0).to(torch.int32).clamp(0, gx)
    x1 = ((ud + r + 15.0) / 16.0).to(torch.int32).clamp(0, gx)
    y0 = ((vd - r) / 16.0).to(torch.int32).clamp(0, gy)
    y1 = ((vd + r + 15.0) / 16.0).to(torch.int32).clamp(0, gy)
    tiles = (x1 - x0) * (y1 - y0)
    ok = ok & (tiles > 0)
    inv = 1.0 / torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([d * inv, -b * inv, a * inv], dim=-1)
    zero3 = torch.zeros_like(conic)
    conic = torch.where(ok.unsqueeze(-1), conic, zero3)
    zi = torch.zeros_like(radius)
    return conic, torch.where(ok, radius, zi), torch.where(ok, tiles, zi)
