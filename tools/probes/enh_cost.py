"""GPU box: forward compositing of the 23-channel render_iter row with and without the K = 20 id lists."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import splatter_a_video_amd._lib as L
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.synth import make_scene
N, W, H, F = 300000, 854, 480, 8
sc = make_scene(N, W, H, F=F, seed=1234)
t = lambda x: torch.tensor(x, device="cuda")
rng = np.random.default_rng(0)
off = t(np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32))
rgb, att = t(rng.uniform(size=(N, 3)).astype(np.float32)), t(rng.uniform(size=(N, 19)).astype(np.float32))
B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
sets = [dict(feature=rgb, taps=True), dict(feature="depth", bg=1.0), dict(feature=att, detach_opacity=True)]
for K in (0, 20, 0, 20):
    for rep in range(3):
        if rep == 1:
            torch.cuda.synchronize(); L.profile_reset(); L.profile_enable(True)
        with torch.no_grad():
            B.render_sets(t(sc.xyz), t(sc.scale), t(sc.rotate), t(sc.opacity), sets, off, t(sc.extr), K=K)
    torch.cuda.synchronize(); L.profile_enable(False)
    ms, n = L.profile_read("blend_fwd")
    print(f"K={K}: blend_fwd {ms / n / F * 1e3:.1f} us per frame")
