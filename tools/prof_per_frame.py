"""GPU box: which torch ops (copies, fills, adds) the per-frame operator path launches per frame, by torch.profiler."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

mode = sys.argv[1] if len(sys.argv) > 1 else "frame"
dev = torch.device("cuda:0")
sc = bench.make_scene(300000, 854, 480, F=25, C=0, seed=1234)
frames = list(range(6))
R = bench.FrameRenderer(sc, dev, frames, 0, mode=mode)
for _ in range(2):
    R.step(collective=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    R.step(collective=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
# stacks of the memcpy launches
for ev in prof.events():
    n = ev.name
    if "emcpy" in n or n in ("aten::copy_", "aten::clone", "aten::fill_", "aten::add_", "aten::zero_", "aten::add", "aten::mul"):
        st = [s for s in (ev.stack or []) if "site-packages" not in s][:4]
        print(n, [tuple(i) for i in (ev.input_shapes or [])][:2] if hasattr(ev, "input_shapes") else "", "|", " <- ".join(st))
