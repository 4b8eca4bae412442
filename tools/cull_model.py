"""CPU model of the per-quarter cull tests on the bench scene (tools only; imports the oracle for the geometry).
Counts, per (tile, splat) pair of a sample of Gaussians: blocks kept (exact block test), 4x4 quarters kept by the bounding-box test,
by the TANGENT-PLANE test (concave quadratic <= its tangent plane at the quarter's centre) and exactly (max alpha over the quarter's
pixel centres >= 1/255)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle
from splatter_a_video_amd.synth import make_scene

N, W, H = 300000, 854, 480
sc = make_scene(N, W, H, seed=1234)
xyz = sc.positions(0)
uv, depth = oracle.project_point_ortho_forward(xyz, sc.extr, W, H, nearest=0.01)
cov = oracle.compute_cov3d_forward(sc.scale, sc.rotate)
conic, radius, tiles = oracle.ewa_project_forward(xyz, cov, sc.intr, sc.extr, uv, W, H, visible=None, ortho=True)
rng = np.random.default_rng(0)
sample = rng.choice(N, 5000, replace=False)
gx, gy = (W + 15) // 16, (H + 15) // 16
tot = dict(qtan_nogate=0, blk_nogate=0, zero=0, pairs=0, blocks=0, qbox=0, qtan=0, qexact=0, active=0, qtan2=0)
px = np.arange(16)
for i in sample:
    r = radius[i]
    if r <= 0: continue
    u, v = uv[i]; A, B, C = conic[i]; o = sc.opacity[i, 0]
    if 255 * o < 0.999: 
        x0 = min(gx, max(0, int((u - r) / 16))); x1 = min(gx, max(0, int((u + r + 15) / 16)))
        y0 = min(gy, max(0, int((v - r) / 16))); y1 = min(gy, max(0, int((v + r + 15) / 16)))
        tot["pairs"] += (x1 - x0) * (y1 - y0); tot["zero"] += (x1 - x0) * (y1 - y0); continue
    tau = 2 * np.log(255 * o)
    det = A * C - B * B
    hx, hy = np.sqrt(tau * C / det), np.sqrt(tau * A / det)
    x0 = min(gx, max(0, int((u - r) / 16))); x1 = min(gx, max(0, int((u + r + 15) / 16)))
    y0 = min(gy, max(0, int((v - r) / 16))); y1 = min(gy, max(0, int((v + r + 15) / 16)))
    for ty in range(y0, y1):
        for tx in range(x0, x1):
            tot["pairs"] += 1
            nb0 = tot["blocks"]
            X, Y = np.meshgrid(tx * 16 + px, ty * 16 + px)
            dx, dy = X - u, Y - v
            q = A * dx * dx + 2 * B * dx * dy + C * dy * dy
            act = q <= tau
            tot["active"] += act.sum()
            for b in range(4):
                bx, by = 8 * (b & 1), 8 * (b >> 1)
                if not act[by:by + 8, bx:bx + 8].any():
                    # exact block test may still keep (continuous min over rect vs pixel centres are the same set here: rect of centres)
                    # continuous minimum over the rectangle of pixel centres
                    pass
                # continuous test: min of q over rect [bx..bx+7]
                def rect_min(xa, xb, ya, yb):
                    cx_ = min(max(u, xa), xb); cy_ = min(max(v, ya), yb)
                    if cx_ == u and cy_ == v: return 0.0
                    best = np.inf
                    for xe in (xa, xb):
                        ys = min(max(v - B * (xe - u) / C, ya), yb); d1, d2 = xe - u, ys - v
                        best = min(best, A * d1 * d1 + 2 * B * d1 * d2 + C * d2 * d2)
                    for ye in (ya, yb):
                        xs = min(max(u - B * (ye - v) / A, xa), xb); d1, d2 = xs - u, ye - v
                        best = min(best, A * d1 * d1 + 2 * B * d1 * d2 + C * d2 * d2)
                    return best
                X0, Y0 = tx * 16 + bx, ty * 16 + by
                anyq = False
                for qd in range(4):
                    qx0, qy0 = X0 + 4 * (qd & 1), Y0 + 4 * (qd >> 1)
                    ax = max(qx0 - u, u - (qx0 + 3), 0); ay = max(qy0 - v, v - (qy0 + 3), 0)
                    if not (ax <= hx and ay <= hy): continue
                    cxq, cyq = qx0 + 1.5 - u, qy0 + 1.5 - v
                    t1, t2 = A * cxq + B * cyq, B * cxq + C * cyq
                    if cxq * t1 + cyq * t2 - 3 * (abs(t1) + abs(t2)) <= tau:
                        tot["qtan_nogate"] += 1; anyq = True
                tot["blk_nogate"] += anyq
                if rect_min(X0, X0 + 7, Y0, Y0 + 7) > tau: continue
                tot["blocks"] += 1
                for qd in range(4):
                    qx0, qy0 = X0 + 4 * (qd & 1), Y0 + 4 * (qd >> 1)
                    ax = max(qx0 - u, u - (qx0 + 3), 0); ay = max(qy0 - v, v - (qy0 + 3), 0)
                    if not (ax <= hx and ay <= hy): continue
                    tot["qbox"] += 1
                    cxq, cyq = qx0 + 1.5 - u, qy0 + 1.5 - v
                    t1, t2 = A * cxq + B * cyq, B * cxq + C * cyq
                    qc = cxq * t1 + cyq * t2
                    if qc - 2 * 1.5 * (abs(t1) + abs(t2)) <= tau: tot["qtan"] += 1
                    # tangent test combined with box test is what would run
                    if rect_min(qx0, qx0 + 3, qy0, qy0 + 3) <= tau: tot["qexact"] += 1
            if tot["blocks"] == nb0: tot["zero"] += 1
p = tot["pairs"]
print({k: round(v / p, 3) for k, v in tot.items()})
