"""can torch's HIP-graph capture take the library's launches?  One frame (forward + backward through frames.frame_rasterization) /
a 25-frame loop captured once and replayed; frames/s against the eager loop (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dptr.gs as gs
from splatter_a_video_amd.synth import make_scene
from splatter_a_video_amd import densify as D
sc = make_scene(300000, 854, 480, F=50, C=0, seed=1234)
t = lambda a, g=False: torch.tensor(a, device="cuda", requires_grad=g)
W, H, N = sc.W, sc.H, sc.N
extr = t(sc.extr)
uv, *_ = gs.preprocess_ortho(t(sc.xyz), t(sc.scale), t(sc.rotate), extr, W, H, nearest=0.01)
perm = D.spatial_order(uv, W, H).cpu().numpy()
P = {k: t(np.ascontiguousarray(v[perm]), True) for k, v in dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity).items()}
feat = torch.rand(N, 3, device="cuda", requires_grad=True)
offs = torch.stack([t((sc.positions(f) - sc.xyz)[perm]) for f in range(25)])
g = torch.randn(3, H, W, device="cuda")
sink = {k: torch.zeros_like(v) for k, v in dict(xyz=P["xyz"], scales=P["scale"], uquats=P["rotate"], opacity=P["opacity"], feature=feat).items()}
def step():
    for f in range(25):
        img = gs.rasterization_ortho(P["xyz"], P["scale"], P["rotate"], P["opacity"], feat, extr, W, H, 0.0, offset=offs[f], grad_sink=sink)
        img.backward(g)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); eager = 200 / (time.perf_counter() - t0)
ref = {k: v.clone() for k, v in sink.items()}
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
for v in sink.values(): v.zero_()
with torch.cuda.graph(graph):
    step()
torch.cuda.synchronize()
for v in sink.values(): v.zero_()
graph.replay(); torch.cuda.synchronize()
one = {k: v.clone() for k, v in sink.items()}
for v in sink.values(): v.zero_()
step(); torch.cuda.synchronize()
same = all(torch.equal(one[k], sink[k]) for k in sink)
t0 = time.perf_counter()
for _ in range(8): graph.replay()
torch.cuda.synchronize(); rep = 200 / (time.perf_counter() - t0)
print(f"eager {eager:.1f} frames/s | graph replay {rep:.1f} frames/s | gradients of a replay == eager step: {same}")
