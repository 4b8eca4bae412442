"""The reference renderer's per-frame chain composed from the CPU oracle (test infrastructure): SH-free feature sets of one
geometry through project_point (ortho) -> cov3d -> EWA -> sort -> one blend per set, forward and backward, with the
routing of DPTROrthoEnhancedRender.render_iter (reference: src/pointrix/renderer/dptr_ortho_enhanced.py:282-376: the tap
set carries ndc / abs_ndc, the depth set blends the projection's own depth, a detached set sees opacity.detach()).
Used by the frame-batch oracle tests (tests/test_gpu_frames_oracle.py)."""
import numpy as np


def frame(o, xyz, scale, rotate, opacity, extr, W, H, sets, grads, K=0, nearest=0.01, intr=None):
    """One frame.  ``sets``: dicts(feature=[P,c] array or "depth", bg, detach_opacity, taps); ``grads``: image gradient
    [c,H,W] per set.  Returns dict(imgs, gs_idx, ncontrib, final_T, radius, M, d=dict of float64 gradients: xyz, scale,
    rotate, opacity, feats (per set, None for depth), tap, abs_tap)."""
    P = xyz.shape[0]
    ortho = intr is None          # intr (fx, fy, cx, cy): the pinhole camera of gs.rasterization instead
    if ortho:
        uv, depth = o.project_point_ortho_forward(xyz, extr, W, H, nearest)
    else:
        uv, depth = o.project_point_forward(xyz, intr, extr, W, H, nearest)
    vis = depth.reshape(-1) != 0
    cov = o.compute_cov3d_forward(scale, rotate, vis)
    conic, radius, tiles = o.ewa_project_forward(xyz, cov, intr, extr, uv, W, H, vis, ortho=ortho)
    idx, tr = o.sort_gaussian(uv, depth, W, H, radius, tiles)
    duv = np.zeros((P, 2), np.float64); dcon = np.zeros((P, 3), np.float64); dop = np.zeros((P, 1), np.float64)
    ddepth = np.zeros((P, 1), np.float64)
    half = np.array([[0.5 * W, 0.5 * H]], np.float64)
    imgs, dfe, gs_idx, tap, atap, nc0, fT0 = [], [], None, None, None, None, None
    for s, g in zip(sets, grads):
        f = depth if isinstance(s["feature"], str) else s["feature"]
        k = K if s.get("taps") else 0
        res = o.alpha_blending_forward(uv, conic, opacity, f, idx, tr, s.get("bg", 0.0), W, H, K=k)
        img, fT, nc = res[:3]
        if k:
            gs_idx = res[3]
        if nc0 is None:
            nc0, fT0 = nc, fT
        imgs.append(img)
        b = o.alpha_blending_backward(uv, conic, opacity, f, idx, tr, s.get("bg", 0.0), W, H, fT, nc, g)
        duv += b[0]; dcon += b[1]
        if not s.get("detach_opacity"):
            dop += b[2]
        if isinstance(s["feature"], str):
            ddepth += b[3]
            dfe.append(None)
        else:
            dfe.append(b[3].astype(np.float64))
        if s.get("taps"):
            tap, atap = b[0] * half, b[4] * half
    dxyz_e, dcov, _, _ = o.ewa_project_backward(xyz, cov, intr, extr, radius, dcon.astype(np.float32), W, H, ortho=ortho,
                                                need_intr=False, need_extr=False)
    if ortho:
        dxyz = o.project_point_ortho_backward(extr, W, H, depth, duv.astype(np.float32), ddepth.astype(np.float32))
    else:
        dxyz, _, _ = o.project_point_backward(xyz, intr, extr, W, H, uv, depth, duv.astype(np.float32), ddepth.astype(np.float32),
                                              need_intr=False, need_extr=False)
    dscale, dquat = o.compute_cov3d_backward(scale, rotate, vis, dcov)
    return dict(imgs=imgs, gs_idx=gs_idx, ncontrib=nc0, final_T=fT0, radius=radius, M=int(idx.size), uv=uv, conic=conic,
                d=dict(xyz=dxyz.astype(np.float64) + dxyz_e, scale=dscale.astype(np.float64), rotate=dquat.astype(np.float64),
                       opacity=dop, feats=dfe, tap=tap, abs_tap=atap))


def static_frames(o, xyz, offsets, scale, rotate, opacity, extr, W, H, sets, grads, K=0, intr=None, nearest=0.01):
    """F frames xyz + offsets[f] of static Gaussians; gradients summed over the frames (what one backward of the batch gives).
    ``extr`` [4,4] or one per frame [F,4,4]; ``intr`` None (orthographic), [4] or [F,4]; ``offsets`` None: none."""
    F = grads[0].shape[0]
    tot, per = None, []
    for f in range(F):
        e = extr[f] if np.ndim(extr) == 3 else extr
        k = None if intr is None else (intr[f] if np.ndim(intr) == 2 else intr)
        pos = xyz if offsets is None else (xyz + offsets[f]).astype(np.float32)
        r = frame(o, pos, scale, rotate, opacity, e, W, H, sets, [g[f] for g in grads], K, nearest=nearest, intr=k)
        per.append(r)
        tot = _add(tot, r["d"])
    return per, tot


def dynamic_frames(o, clock, times, host, extr, W, H, sets, grads, K=0):
    """F frames of the reference's dynamic Gaussians (oracle dynamic_eval_* around the static chain); gradients w.r.t. the raw
    parameters (position, pos_cubic_node [N,4,I,3], rotation, opacity, scaling) summed over the frames"""
    N = host["position"].shape[0]
    I = clock.interval_num
    tot = dict(position=0.0, pos_cubic_node=0.0, rotation=0.0, opacity=0.0, scaling=0.0, feats=None, tap=0.0, abs_tap=0.0)
    per = []
    for f, t in enumerate(times):
        seg, d, basis = clock.scalars(t)
        b = np.array(list(basis), np.float32)
        pos, rot, opa, scl = o.dynamic_eval_forward(host["position"], host["pos_cubic_node"], host["rotation"],
                                                    host["rot_poly_feat"], host["rot_fourier_feat"], host["opacity"],
                                                    host["scaling"], seg, d, b[:4], b[4:])
        r = frame(o, pos, scl, rot, opa, extr, W, H, sets, [g[f] for g in grads], K)
        per.append(r)
        g = r["d"]
        dpos, dcub, drot, dopa, dscl = o.dynamic_eval_backward(
            (N, 4, I, 3), host["rotation"], host["rot_poly_feat"], host["rot_fourier_feat"], host["opacity"], host["scaling"],
            seg, d, b[:4], b[4:], g["xyz"].astype(np.float32), g["rotate"].astype(np.float32), g["opacity"].astype(np.float32),
            g["scale"].astype(np.float32))
        tot["position"] = tot["position"] + dpos.astype(np.float64)
        tot["pos_cubic_node"] = tot["pos_cubic_node"] + dcub.astype(np.float64)
        tot["rotation"] = tot["rotation"] + drot.astype(np.float64)
        tot["opacity"] = tot["opacity"] + dopa.astype(np.float64)
        tot["scaling"] = tot["scaling"] + dscl.astype(np.float64)
        tot["feats"] = g["feats"] if tot["feats"] is None else [None if a is None else a + c for a, c in zip(tot["feats"], g["feats"])]
        if g["tap"] is not None:
            tot["tap"] = tot["tap"] + g["tap"]; tot["abs_tap"] = tot["abs_tap"] + g["abs_tap"]
    return per, tot


def _add(tot, d):
    if tot is None:
        return {k: (list(v) if k == "feats" else v) for k, v in d.items()}
    out = {}
    for k, v in d.items():
        if k == "feats":
            out[k] = [None if a is None else a + c for a, c in zip(tot[k], v)]
        else:
            out[k] = None if v is None else tot[k] + v
    return out
