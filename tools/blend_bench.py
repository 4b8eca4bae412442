#!/usr/bin/env python
"""Kernel-level experiment harness (GPU box): times blend fwd/bwd (and the other kernels) at a
BASELINE config with the library's hipEvent profiler, for a list of env-var variants.
usage: python tools/blend_bench.py [--gaussians N --width W --height H --channels C --reps R] VAR=VAL,... ..."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dptr.gs as gs  # noqa: E402
import splatter_a_video_amd._lib as L  # noqa: E402
from splatter_a_video_amd.synth import make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=300000)
ap.add_argument("--width", type=int, default=854)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--opacity", type=float, default=None)
ap.add_argument("variants", nargs="*", default=[""])
a = ap.parse_args()

dev = torch.device("cuda:0")
sc = make_scene(a.gaussians, a.width, a.height, C=a.channels, seed=1234)
if a.opacity is not None:
    sc.opacity[:] = a.opacity
W, H = a.width, a.height
t = lambda x: torch.tensor(x, device=dev)
xyz, extr = t(sc.positions(0)), t(sc.extr)
uv, depth = gs.project_point_ortho(xyz, extr, W, H, nearest=0.01)
vis = depth != 0
cov = gs.compute_cov3d(t(sc.scale), t(sc.rotate), vis)
conic, radius, tiles = gs.ewa_project_ortho(xyz, cov, extr, uv, W, H, vis)
idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
feat = t(sc.feature).requires_grad_(True)
op = t(sc.opacity).requires_grad_(True)
uvg = uv.detach().requires_grad_(True)
cng = conic.detach().requires_grad_(True)
torch.manual_seed(0)
g = torch.randn(a.channels, H, W, device=dev)
M = idx.numel()
print(f"N={a.gaussians} {W}x{H} C={a.channels} M={M} tiles={tr.shape[0]} mean_list={M / tr.shape[0]:.0f}")
ref = None
for var in a.variants:
    for kv in filter(None, var.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    out = None
    for rep in range(a.reps + 1):
        if rep == 1:
            torch.cuda.synchronize(); L.profile_reset(); L.profile_enable(True)
        for x in (feat, op, uvg, cng):
            x.grad = None
        out = gs.alpha_blending(uvg, cng, op, feat, idx, tr, 0.0, W, H)
        out.backward(g)
    torch.cuda.synchronize()
    L.profile_enable(False)
    f_ms, f_n = L.profile_read("blend_fwd")
    b_ms, b_n = L.profile_read("blend_bwd")
    chk = (float(out.double().sum()), float(uvg.grad.double().abs().sum()), float(feat.grad.double().abs().sum()))
    if ref is None:
        ref = chk
    print(f"[{var or 'default':40s}] fwd {f_ms / max(f_n, 1) * 1e3:8.1f} us   bwd {b_ms / max(b_n, 1) * 1e3:8.1f} us   "
          f"checksum out={chk[0]:.6e} |duv|={chk[1]:.6e} |df|={chk[2]:.6e}")
    for kv in filter(None, var.split(",")):
        os.environ.pop(kv.split("=")[0], None)
