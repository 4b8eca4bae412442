"""PyTorch-eager (CPU, float32) restatement of one rendered frame, forward + backward -- TEST INFRASTRUCTURE / CPU BASELINE
ONLY (SURVEY.md 8d "CPU baseline"); nothing in the product path imports this.

The reference has no CPU compositor (sort and alpha blending exist only as CUDA); what it does in eager torch is the
orthographic projection and EWA (src/pointrix/renderer/dptr_ortho_enhanced.py:18-111,145-202).  This module restates the
whole frame the way that file would if it had to stay in torch: eager preprocess as there, SH colour for the constant
view direction (:270-272, src/submodules/dptr/dptr/gs/src/compute_sh.cu:43-79), key emission + a global sort
(sort_gaussian.py:42-52), and the per-pixel compositing recurrences of src/alpha_blending.cu:32-109 vectorised over the
pixels of a tile and the tile's list (cumulative products), tiles processed in groups of equal padded length; the
backward is autograd's.  It is checked against the C oracle in tests/test_oracle_cpu.py.
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2_2 = 0.31539156525252005
SH_C3_3 = 0.3731763325901154


def _sh_rgb_z(shs: torch.Tensor) -> torch.Tensor:
    """SH degree 3 -> RGB for direction (0, 0, 1): only the zonal terms survive (compute_sh.cu:43-79 with x = y = 0)"""
    rgb = SH_C0 * shs[:, 0] + SH_C1 * shs[:, 2] + SH_C2_2 * 2.0 * shs[:, 6] + SH_C3_3 * 2.0 * shs[:, 12] + 0.5
    return torch.clamp_min(rgb, 0.0)


def _rot(q: torch.Tensor) -> torch.Tensor:
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def preprocess(xyz, scale, quat, extr, W, H, nearest=0.01, extent=1.3):
    """ortho projection + cov3d + ortho EWA in eager torch (dptr_ortho_enhanced.py:18-111,177-202; compute_cov3d.cu:24-58)"""
    R, T = extr[:3, :3], extr[:3, 3]
    t = xyz @ R.T + T
    u = (t[:, 0] + 1.0) * (W / 2.0) - 0.5
    v = (t[:, 1] + 1.0) * (H / 2.0) - 0.5
    d = torch.nan_to_num(t[:, 2])
    cull = (d <= nearest) | (u < (1 - extent) * W / 2) | (u > (1 + extent) * W / 2) | (v < (1 - extent) * H / 2) | (v > (1 + extent) * H / 2)
    vis = ~cull
    Rq = _rot(quat)
    Mx = Rq * scale[:, None, :]
    Sigma = Mx @ Mx.transpose(1, 2)
    J = torch.tensor([[W / 2.0, 0.0, 0.0], [0.0, H / 2.0, 0.0]], dtype=xyz.dtype)
    Tm = J @ R
    cov = Tm @ Sigma @ Tm.T
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int32)
    conic = torch.stack([c / det, -b / det, a / det], -1)
    vis = vis & (det != 0)
    return u, v, d, conic, torch.where(vis, radius, torch.zeros_like(radius)), vis


def tile_lists(u, v, d, radius, W, H):
    """(tile, depth)-sorted (Gaussian, tile) pairs: utils.h:17-37 rectangles, sort_gaussian.cu:24-69 keys, global sort"""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    un, vn, rn, dn = (x.detach().numpy() for x in (u, v, radius, d))
    r = rn.astype(np.float32)
    x0 = np.clip(((un - r) / 16.0).astype(np.int32), 0, gx); x1 = np.clip(((un + r + 15.0) / 16.0).astype(np.int32), 0, gx)
    y0 = np.clip(((vn - r) / 16.0).astype(np.int32), 0, gy); y1 = np.clip(((vn + r + 15.0) / 16.0).astype(np.int32), 0, gy)
    x0[rn <= 0] = 0; x1[rn <= 0] = 0; y0[rn <= 0] = 0; y1[rn <= 0] = 0
    wx, wy = x1 - x0, y1 - y0
    cnt = wx * wy
    gid = np.repeat(np.arange(un.size), cnt)
    first = np.cumsum(cnt) - cnt
    k = np.arange(gid.size) - np.repeat(first, cnt)
    wxr = np.repeat(np.maximum(wx, 1), cnt)
    tid = (np.repeat(y0, cnt) + k // wxr) * gx + np.repeat(x0, cnt) + k % wxr
    order = np.lexsort((gid, dn[gid].view(np.uint32), tid))        # stable: ties by ascending id
    gid, tid = gid[order], tid[order]
    T = gx * gy
    start = np.searchsorted(tid, np.arange(T), side="left"); end = np.searchsorted(tid, np.arange(T), side="right")
    return gid, start, end, gx, gy


def composite(u, v, conic, opacity, feat, gid, start, end, gx, gy, W, H, bg, dL_dout=None, group=48, deadline=None, stats=None):
    """alpha_blending.cu:32-109, vectorised: per group of tiles (lists padded to the group's longest), pixel x entry
    matrices, transmittance by cumulative product, the T < 1e-4 stop as a mask (T is monotone).  With ``dL_dout`` the
    group's share of sum(image * dL_dout) is back-propagated right away (the inputs' .grad accumulate), so that only one
    group's intermediates are alive at a time.  ``deadline`` (time.perf_counter() value): stop after the group that
    crosses it (bounded CPU-baseline sample); ``stats["done"]`` = fraction of the tile-list entries composited."""
    C = feat.shape[1]
    out = torch.zeros(C, H, W)
    lens = end - start
    order = np.argsort(-lens, kind="stable")
    px = torch.arange(16, dtype=torch.float32)
    ly, lx = torch.meshgrid(px, px, indexing="ij")
    lx, ly = lx.reshape(-1), ly.reshape(-1)
    gid_t = torch.from_numpy(gid)
    gpad = None
    if dL_dout is not None:
        gpad = torch.zeros(C, gy * 16, gx * 16)
        gpad[:, :H, :W] = dL_dout
    total, done = float(max(int(lens.sum()), 1)), 0
    for g0 in range(0, order.size, group):
        if deadline is not None and time.perf_counter() > deadline:
            break
        tiles = order[g0:g0 + group]
        Lmax = int(lens[tiles].max())
        if Lmax == 0:
            break
        done += int(lens[tiles].sum())
        nt = tiles.size
        idx = torch.zeros(nt, Lmax, dtype=torch.int64)
        valid = torch.zeros(nt, Lmax, dtype=torch.bool)
        for k, t in enumerate(tiles):
            n = int(lens[t])
            idx[k, :n] = gid_t[start[t]:end[t]]
            valid[k, :n] = True
        ox = torch.tensor([(int(t) % gx) * 16.0 for t in tiles]); oy = torch.tensor([(int(t) // gx) * 16.0 for t in tiles])
        pxx = ox[:, None] + lx[None, :]; pyy = oy[:, None] + ly[None, :]                    # [nt,256]
        dx = u[idx][:, None, :] - pxx[:, :, None]; dy = v[idx][:, None, :] - pyy[:, :, None]   # [nt,256,L]
        cn = conic[idx]
        power = -0.5 * (cn[:, None, :, 0] * dx * dx + cn[:, None, :, 2] * dy * dy) - cn[:, None, :, 1] * dx * dy
        alpha = torch.clamp_max(opacity[idx][:, None, :] * torch.exp(power), 0.99)
        keep = valid[:, None, :] & (power <= 0) & (alpha >= 1.0 / 255.0)
        alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
        Tin = torch.cumprod(1.0 - alpha, dim=2)                      # transmittance after each entry
        Tex = torch.cat([torch.ones_like(Tin[:, :, :1]), Tin[:, :, :-1]], 2)
        live = Tin >= 1e-4                                           # an entry that would push T below 1e-4 ends the pixel
        wgt = torch.where(live, alpha * Tex, torch.zeros_like(alpha))
        Tfin = torch.where(live, Tin, torch.full_like(Tin, 2.0)).min(dim=2).values.clamp_max(1.0)
        img = torch.einsum("tpl,tlc->tpc", wgt, feat[idx]) + Tfin[:, :, None] * bg       # [nt,256,C]
        if gpad is not None:
            gt = torch.stack([gpad[:, (int(t) // gx) * 16:(int(t) // gx) * 16 + 16, (int(t) % gx) * 16:(int(t) % gx) * 16 + 16]
                              .permute(1, 2, 0).reshape(256, C) for t in tiles])
            (img * gt).sum().backward()
            img = img.detach()
        for k, t in enumerate(tiles):
            x0, y0 = (int(t) % gx) * 16, (int(t) // gx) * 16
            hh, ww = min(16, H - y0), min(16, W - x0)
            out[:, y0:y0 + hh, x0:x0 + ww] = img[k].reshape(16, 16, C)[:hh, :ww].permute(2, 0, 1)
    if stats is not None:
        stats["done"] = done / total
    # tiles without pairs keep the background
    for t in np.nonzero(lens == 0)[0]:
        x0, y0 = (int(t) % gx) * 16, (int(t) // gx) * 16
        out[:, y0:y0 + 16, x0:x0 + 16] = bg
    return out


def frame_forward(sc, f, use_sh=True, dL_dout=None, budget_s=None, stats=None):
    """-> (image [C,H,W], dict of parameter tensors, M); with ``dL_dout`` [C,H,W] the parameters' .grad hold the gradient
    of sum(image * dL_dout) afterwards (compositing back-propagated group by group into its per-Gaussian inputs, then
    one backward through the eager preprocess).  ``budget_s``: stop compositing after that many seconds (``stats``
    receives the fraction of the tile lists done and the seconds spent outside / inside the compositing loop)."""
    rg = dL_dout is not None
    t0 = time.perf_counter()
    p = {k: torch.tensor(v, requires_grad=rg) for k, v in
         dict(xyz=sc.positions(f), scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity.reshape(-1)).items()}
    if use_sh:
        p["shs"] = torch.tensor(sc.shs, requires_grad=rg)
        feat = _sh_rgb_z(p["shs"])
    else:
        p["feature"] = torch.tensor(sc.feature, requires_grad=rg)
        feat = p["feature"]
    extr = torch.tensor(sc.extr)
    u, v, d, conic, radius, vis = preprocess(p["xyz"], p["scale"], p["rotate"], extr, sc.W, sc.H)
    gid, start, end, gx, gy = tile_lists(u, v, d, radius, sc.W, sc.H)
    t1 = time.perf_counter()
    st = {} if stats is None else stats
    deadline = None if budget_s is None else t1 + budget_s
    if not rg:
        with torch.no_grad():
            img = composite(u, v, conic, p["opacity"], feat, gid, start, end, gx, gy, sc.W, sc.H, sc.bg, deadline=deadline, stats=st)
        return img, p, int(gid.size)
    mids = [u, v, conic, p["opacity"], feat]
    leaf = [m.detach().requires_grad_(True) for m in mids]
    img = composite(*leaf, gid, start, end, gx, gy, sc.W, sc.H, sc.bg, dL_dout=torch.as_tensor(dL_dout), deadline=deadline, stats=st)
    t2 = time.perf_counter()
    grads = [l.grad if l.grad is not None else torch.zeros_like(l) for l in leaf]
    torch.autograd.backward([m for m in mids if m.requires_grad], [g for m, g in zip(mids, grads) if m.requires_grad])
    t3 = time.perf_counter()
    st["outside_s"] = (t1 - t0) + (t3 - t2)
    st["composite_s"] = t2 - t1
    return img, p, int(gid.size)


def frame_forward_backward(sc, f, g, use_sh=True, budget_s=None) -> dict:
    """one frame forward + backward; -> {"M", "seconds" (extrapolated to the whole frame when the compositing loop was
    cut at ``budget_s``), "fraction" (of the tile-list entries actually composited)}"""
    st = {}
    M = frame_forward(sc, f, use_sh, dL_dout=g, budget_s=budget_s, stats=st)[2]
    frac = max(st.get("done", 1.0), 1e-9)
    return {"M": M, "fraction": frac, "seconds": st["outside_s"] + st["composite_s"] / frac}
