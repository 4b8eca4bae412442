import sys, torch, numpy as np
sys.path.insert(0, ".")
import bench
from splatter_a_video_amd import gs
sc = bench.make_scene(300000, 854, 480, F=25, C=0, seed=1234)
dev = torch.device("cuda:0")
xyz = torch.tensor(sc.positions(0), device=dev); 
uv, depth, conic, radius, tiles = gs.preprocess_ortho(torch.tensor(sc.xyz, device=dev), torch.tensor(sc.scale, device=dev), torch.tensor(sc.rotate, device=dev), torch.tensor(sc.extr, device=dev), 854, 480, nearest=0.01)
t = tiles.cpu().numpy()
print("mean", t.mean(), "max", t.max())
h = np.bincount(t)
print({i: int(c) for i, c in enumerate(h) if c})
print("frac > 6:", (t > 6).mean(), "frac > 12:", (t > 12).mean())
