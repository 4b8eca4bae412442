"""per-frame rendering through FrameBatch(F = 1) (one C call per direction): frames/s of 25 frames forward + backward, 300k @ 480p"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dptr.gs as gs
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.synth import make_scene
from splatter_a_video_amd import densify as D
sc = make_scene(300000, 854, 480, F=50, C=0, seed=1234)
t = lambda a, g=False: torch.tensor(a, device="cuda", requires_grad=g)
W, H, N = sc.W, sc.H, sc.N
extr = t(sc.extr)
xyz0 = t(sc.xyz)
uv, *_ = gs.preprocess_ortho(xyz0, t(sc.scale), t(sc.rotate), extr, W, H, nearest=0.01)
perm = D.spatial_order(uv, W, H)
P = {k: t(np.ascontiguousarray(v[perm.cpu().numpy()]), True) for k, v in dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity).items()}
feat = torch.rand(N, 3, device="cuda", requires_grad=True)
offs = [t((sc.positions(f) - sc.xyz)[perm.cpu().numpy()][None]) for f in range(25)]
g = torch.randn(1, 3, H, W, device="cuda")
sink = {k: torch.zeros_like(v) for k, v in dict(xyz=P["xyz"], scales=P["scale"], uquats=P["rotate"], opacity=P["opacity"], feature=feat).items()}
B = FrameBatch(1, N, W, H, 3, "cuda")
def step():
    for off in offs:
        out = B.render(P["xyz"], P["scale"], P["rotate"], P["opacity"], feat, off, extr, bg=0.0, grad_sink=sink)
        out.backward(g)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in range(4): step()
host = (time.perf_counter() - t1) / 100 * 1e6
torch.cuda.synchronize()
print(f"FrameBatch(F=1): {200 / dt:.1f} frames/s, {dt / 200 * 1e6:.1f} us/frame wall, host enqueue {host:.1f} us/frame")
