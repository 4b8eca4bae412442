"""cProfile of the host side of one bench step (where do the Python microseconds of a frame go?)
   python tools/host_profile.py [--ops]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from splatter_a_video_amd.synth import make_scene

ops = "--ops" in sys.argv
dev = torch.device("cuda", 0)
sc = make_scene(300000, 854, 480, F=50, C=0, seed=1234)
R = bench.FrameRenderer(sc, dev, 0, fused=not ops)
offs = [R.offsets(i) for i in range(25)]
for _ in range(2):
    R.step(offs)
R.finish(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    R.step(offs)
pr.disable()
R.finish(); torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()))
