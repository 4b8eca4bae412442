"""Independent float64 PyTorch twin of the rasterizer ops (test helper).

Written from the operator SEMANTICS, vectorised over the pixels of a tile,
so that torch.autograd provides gradients to cross-check the analytic backward passes of the C
oracle (oracle/splat_oracle.c) and of the HIP kernels.  Not a product path.
"""
from __future__ import annotations

import torch

TILE = 16


def project_point_persp(xyz, intr, extr, W, H, nearest=0.2, extent=1.3):
    R = extr[:3, :3]
    t = xyz @ R.T + extr[:3, 3]
    inv = 1.0 / (t[:, 2] + 1e-7)
    u = intr[0] * t[:, 0] * inv + intr[2] - 0.5
    v = intr[1] * t[:, 1] * inv + intr[3] - 0.5
    d = t[:, 2]
    cull = torch.zeros_like(d, dtype=torch.bool)
    if nearest > 0:
        cull |= d <= nearest
    if extent > 0:
        cull |= (u < (1 - extent) * W * 0.5) | (u > (1 + extent) * W * 0.5)
        cull |= (v < (1 - extent) * H * 0.5) | (v > (1 + extent) * H * 0.5)
    keep = (~cull).to(xyz.dtype)
    uv = torch.stack([u, v], -1) * keep[:, None]
    return uv, (d * keep)[:, None]


def project_point_ortho(xyz, extr, W, H, nearest=0.01, extent=1.3):
    R = extr[:3, :3]
    t = xyz @ R.T + extr[:3, 3]
    u = (t[:, 0] + 1.0) * W / 2 - 0.5
    v = (t[:, 1] + 1.0) * H / 2 - 0.5
    d = t[:, 2]
    cull = (d <= nearest)
    cull |= (u < (1 - extent) * W * 0.5) | (u > (1 + extent) * W * 0.5)
    cull |= (v < (1 - extent) * H * 0.5) | (v > (1 + extent) * H * 0.5)
    keep = (~cull).to(xyz.dtype)
    return torch.stack([u, v], -1) * keep[:, None], (d * keep)[:, None]


def quat_to_R(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def cov3d(scale, quat, visible=None):
    R = quat_to_R(quat)
    L = R * scale[:, None, :]
    S = L @ L.transpose(1, 2)
    out = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)
    if visible is not None:
        out = out * visible.reshape(-1, 1).to(out.dtype)
    return out


def _sym(c):
    return torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                       -1).reshape(-1, 3, 3)


def ewa(xyz, cov3, intr, extr, W, H, mask, ortho=False):
    """conic only (radius / tiles are not differentiable); mask = radius>0 from the oracle."""
    R = extr[:3, :3]
    t = xyz @ R.T + extr[:3, 3]
    P = xyz.shape[0]
    J = torch.zeros(P, 2, 3, dtype=xyz.dtype)
    if ortho:
        J[:, 0, 0] = W / 2
        J[:, 1, 1] = H / 2
    else:
        J[:, 0, 0] = intr[0] / t[:, 2]
        J[:, 1, 1] = intr[1] / t[:, 2]
        J[:, 0, 2] = -intr[0] * t[:, 0] / t[:, 2] ** 2
        J[:, 1, 2] = -intr[1] * t[:, 1] / t[:, 2] ** 2
    T = J @ R
    c = T @ _sym(cov3) @ T.transpose(1, 2)
    a = c[:, 0, 0] + 0.3
    b = c[:, 0, 1]
    d = c[:, 1, 1] + 0.3
    det = a * d - b * b
    conic = torch.stack([d / det, -b / det, a / det], -1)
    return conic * mask.reshape(-1, 1).to(conic.dtype)


def sh_color(shs, deg, dirs, free=False):
    C0 = 0.28209479177387814
    C1 = 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = C0 * shs[:, 0]
    if deg > 0:
        r = r - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6] + \
            C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8]
    if deg > 2:
        r = r + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10] + \
            C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] + \
            C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14] + \
            C3[6] * x * (xx - 3 * yy) * shs[:, 15]
    if free:
        return r
    return torch.clamp_min(r + 0.5, 0.0)


def blend(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, bias=None, K=0, truncate=False):
    """Front-to-back compositing (Appendix A.6). Returns out[C,H,W], final_T[H,W], ncontrib[H,W], gs_idx.
    The 0.99 clamp is straight-through (the reference does not mask it in the gradient)."""
    dt = uv.dtype
    C = feature.shape[1]
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    out = torch.zeros(C, H, W, dtype=dt)
    fT = torch.zeros(H, W, dtype=dt)
    nc = torch.zeros(H, W, dtype=torch.int32)
    gi = torch.full((H, W, max(K, 1)), -1, dtype=torch.int32)
    opacity = opacity.reshape(-1)
    for tile in range(gx * gy):
        tx, ty = tile % gx, tile // gx
        x0, y0 = tx * TILE, ty * TILE
        x1, y1 = min(W, x0 + TILE), min(H, y0 + TILE)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        px = xs.reshape(-1).to(dt)
        py = ys.reshape(-1).to(dt)
        n = px.numel()
        T = torch.ones(n, dtype=dt)
        F = torch.zeros(n, C, dtype=dt)
        done = torch.zeros(n, dtype=torch.bool)
        last = torch.zeros(n, dtype=torch.int32)
        layer = torch.zeros(n, dtype=torch.long)
        gsl = torch.full((n, max(K, 1)), -1, dtype=torch.int32)
        r0, r1 = int(tile_range[tile, 0]), int(tile_range[tile, 1])
        contributor = 0
        for s in range(r0, r1):
            contributor += 1
            if bool(done.all()):
                break
            g = int(idx_sorted[s])
            dx = uv[g, 0] - px
            dy = uv[g, 1] - py
            power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
            araw = opacity[g] * torch.exp(power)
            if bias is not None:
                araw = araw + bias.reshape(-1)[g]
            alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()
            ok = (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
            nT = T * (1 - alpha)
            sat = ok & (nT < 1e-4)
            done = done | sat
            app = ok & ~sat
            w = torch.where(app, alpha * T, torch.zeros_like(T))
            F = F + w[:, None] * feature[g][None, :]
            T = torch.where(app, nT, T)
            last = torch.where(app, torch.full_like(last, contributor), last)
            if K > 0:
                rec = app & (layer < K)
                if bool(rec.any()):
                    ridx = torch.nonzero(rec).reshape(-1)
                    gsl[ridx, layer[ridx]] = g
                    layer = layer + rec.long()
                if truncate:
                    done = done | (app & (layer >= K))
        out[:, y0:y1, x0:x1] = (F + T[:, None] * bg).T.reshape(C, y1 - y0, x1 - x0)
        fT[y0:y1, x0:x1] = T.detach().reshape(y1 - y0, x1 - x0)
        nc[y0:y1, x0:x1] = last.reshape(y1 - y0, x1 - x0)
        gi[y0:y1, x0:x1] = gsl.reshape(y1 - y0, x1 - x0, -1)
    return out, fT, nc, gi
