"""scratch (GPU): time of knn_brute_batch on the bench scene with ascending / shuffled queries"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import dptr.gs as gs
from splatter_a_video_amd.synth import make_scene
from splatter_a_video_amd.densify import spatial_order
from splatter_a_video_amd.knn import knn_brute_batch
from splatter_a_video_amd import _lib as L
N, W, H, B, S = 300000, 854, 480, 25, 512
sc = make_scene(N, W, H, F=50, seed=1234)
dev = "cuda"
xyz = torch.tensor(sc.xyz, device=dev)
uv0, _ = gs.project_point_ortho(xyz, torch.tensor(sc.extr, device=dev), W, H, nearest=0.01)
order = spatial_order(uv0, W, H)
P = xyz[order].contiguous()
pts = P[None].repeat(B, 1, 1).contiguous()
rng = np.random.default_rng(0)
for name, prep in (("ascending", np.sort), ("shuffled", lambda a: a)):
    q = torch.from_numpy(np.stack([prep(rng.choice(N, S)) for _ in range(B)])).to(dev)
    for _ in range(2):
        d, i = knn_brute_batch(pts, q, 6)
    torch.cuda.synchronize()
    L.profile_reset(); L.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        d, i = knn_brute_batch(pts, q, 6)
    torch.cuda.synchronize()
    L.profile_enable(False)
    print(name, "wall ms/call", (time.perf_counter() - t0) / 5 * 1e3, {k: L.profile_read(k)[0] / 5 for k in ("knn_brute_box", "knn_brute_bound", "knn_brute_merge", "knn_brute")})
    print("  6th distance: median", float(d[..., 5].sqrt().median()), "max", float(d[..., 5].sqrt().max()))
# unordered point set for comparison
perm = torch.randperm(N, device=dev)
pts2 = P[perm][None].repeat(B, 1, 1).contiguous()
q = torch.from_numpy(np.stack([np.sort(rng.choice(N, S)) for _ in range(B)])).to(dev)
for _ in range(2):
    knn_brute_batch(pts2, q, 6)
torch.cuda.synchronize(); L.profile_reset(); L.profile_enable(True)
for _ in range(5):
    knn_brute_batch(pts2, q, 6)
torch.cuda.synchronize(); L.profile_enable(False)
print("random point order", {k: L.profile_read(k)[0] / 5 for k in ("knn_brute_box", "knn_brute_bound", "knn_brute_merge", "knn_brute")})
