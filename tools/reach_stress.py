"""Randomised check of the reach masks (GPU box): over random scenes -- sizes, splat sizes from sub-pixel to half the image, elongation,
opacities down to below 1/255, clustered layouts, 3 / 19 channels -- the frame batch with the masks must give the images, final
transmittances and contributor ids of the full lists BIT FOR BIT (a dropped pair that reached a pixel would show as a difference).
   python tools/reach_stress.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from splatter_a_video_amd import frames as FR
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.synth import make_scene

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t = lambda a: torch.tensor(np.asarray(a), device="cuda")
rng = np.random.default_rng(20261001)
bad = 0
for c in range(cases):
    N = int(rng.integers(300, 30000)); W = int(rng.integers(40, 700)); H = int(rng.integers(40, 500)); F = int(rng.integers(1, 4))
    sig = float(np.exp(rng.uniform(np.log(0.4), np.log(0.25 * min(W, H)))))
    sc = make_scene(N, W, H, seed=int(rng.integers(1 << 30)), sigma_px=sig, clustered=float(rng.choice([0.0, 0.7])))
    kind = rng.integers(0, 4)
    if kind == 1: sc.scale[:, rng.integers(0, 2)] *= float(rng.uniform(2, 12))
    if kind == 2: sc.opacity[:] = (10.0 ** rng.uniform(-3.2, 0.0, size=sc.opacity.shape)).astype(np.float32)
    if kind == 3: sc.xyz[:, :2] *= 1.6          # centres off the image
    C = int(rng.choice([3, 19]))
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    off = t(np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32))
    res = []
    for reach in (True, False):
        FR.OPTIONS["reach"] = reach
        B = FrameBatch(F, N, W, H, C, "cuda")
        with torch.no_grad():
            out = B.render(t(sc.xyz), t(sc.scale), t(sc.rotate), t(sc.opacity), t(feat), off, t(sc.extr), bg=0.1)
        torch.cuda.synchronize()
        res.append((out.clone(), B.final_T.clone(), B.ncontrib.clone() > 0, B.check()))
    FR.OPTIONS["reach"] = True
    ok = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    bad += not ok
    print(f"case {c}: N={N} {W}x{H} F={F} sigma={sig:.1f}px kind={kind} C={C} pairs {res[0][3]} / {res[1][3]} -> {'ok' if ok else 'DIFFERENT'}")
print("FAILED" if bad else "all equal", bad)
sys.exit(1 if bad else 0)
