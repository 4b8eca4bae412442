"""Where do the microseconds of a PER-FRAME step go (bench.py --per-frame: the drop-in operators, frame by frame)?  Wall time per
frame against the sum of the kernels' HIP-event times, and a cProfile of the host side.
   python tools/host_profile.py [--ops]"""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import splatter_a_video_amd._lib as L
from splatter_a_video_amd.synth import make_scene

mode = "ops" if "--ops" in sys.argv else "frame"
dev = torch.device("cuda", 0)
sc = make_scene(300000, 854, 480, F=50, C=0, seed=1234)
frames = list(range(25))
R = bench.FrameRenderer(sc, dev, frames, mode=mode)
for _ in range(3):
    R.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    R.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / (8 * 25) * 1e6
t0 = time.perf_counter()
for _ in range(8):
    R.step()
host = (time.perf_counter() - t0) / (8 * 25) * 1e6      # enqueue only (no synchronise)
torch.cuda.synchronize()
L.profile_reset(); L.profile_enable(True)
R.step(); torch.cuda.synchronize()
L.profile_enable(False)
tot = 0.0
for n in ("sh_fwd", "preprocess_fwd", "bin_count", "bin_colscan", "bin_tilescan", "bin_scatter", "tile_sort", "blend_pack", "blend_fwd", "blend_bwd",
          "pair_reduce", "preprocess_bwd", "sh_bwd", "adam_step", "fill", "cull"):
    ms, cnt = L.profile_read(n)
    if cnt:
        print(f"{n:16s} {ms * 1e3 / 25:8.2f} us/frame  {cnt} launches")
        tot += ms * 1e3 / 25
L.profile_reset()
print(f"wall {wall:.1f} us/frame | host enqueue {host:.1f} us/frame | library kernels {tot:.1f} us/frame")
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    R.step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
print("\n".join(l[:160] for l in s.getvalue().splitlines()))
