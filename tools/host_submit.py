"""How long does the host take to enqueue one step (25 frames) vs how long the GPU takes to run it?
   python tools/host_submit.py [--ops]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from splatter_a_video_amd.synth import make_scene

ops = "--ops" in sys.argv
dev = torch.device("cuda", 0)
sc = make_scene(300000, 854, 480, F=50, C=0, seed=1234)
R = bench.FrameRenderer(sc, dev, 0, fused=not ops, dynamic="--dynamic" in sys.argv)
offs = list(range(25)) if "--dynamic" in sys.argv else [R.offsets(i) for i in range(25)]
for _ in range(2):
    R.step(offs)
R.finish(); torch.cuda.synchronize()
sub, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    R.step(offs)
    t1 = time.perf_counter()
    R.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    sub.append((t1 - t0) / 25 * 1e3); tot.append((t2 - t0) / 25 * 1e3)
print("host enqueue ms/frame", [round(x, 3) for x in sub], "total ms/frame", [round(x, 3) for x in tot])
