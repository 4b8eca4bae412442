"""Kernel timing of the dynamic-evaluation row (a15) at the north-star size: 300k Gaussians, 250-frame clip."""
import json
import sys

import numpy as np
import torch

from splatter_a_video_amd import _lib as L
from splatter_a_video_amd.dynamics import FrameClock, SEGMENT_MAJOR, evaluate, to_segment_major

N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000, 250
torch.manual_seed(0)
clock = FrameClock(T)
I = clock.interval_num
dev = "cuda"
p = dict(position=torch.randn(N, 3, device=dev), pos_cubic_node=(0.1 * torch.randn(N, 4 * I * 3, device=dev)).requires_grad_(),
         rotation=torch.randn(N, 4, device=dev).requires_grad_(), rot_poly_feat=0.05 * torch.randn(N, 4, 4, device=dev),
         rot_fourier_feat=0.05 * torch.randn(N, 8, 4, device=dev), opacity=torch.randn(N, 1, device=dev).requires_grad_(),
         scaling=(torch.randn(N, 3, device=dev) - 4).requires_grad_())
sink = {k: torch.zeros_like(p[k]) for k in ("pos_cubic_node", "rotation", "opacity", "scaling")}
g = [torch.randn(N, w, device=dev) for w in (3, 4, 1, 3)]
res = {}
seg_p = dict(p); seg_p["pos_cubic_node"] = to_segment_major(p["pos_cubic_node"].detach(), I).requires_grad_()
seg_sink = dict(sink); seg_sink["pos_cubic_node"] = torch.zeros_like(seg_p["pos_cubic_node"])
for mode in ("dense", "sink", "sink_segmajor"):
    kw = dict(grad_sink=None) if mode == "dense" else dict(grad_sink=sink)
    pp = p
    if mode == "sink_segmajor":
        kw, pp = dict(grad_sink=seg_sink, cubic_layout=SEGMENT_MAJOR), seg_p
    for it in range(3):
        if it == 1:
            L.profile_enable(True); L.profile_reset()
        for t in range(100, 125):
            out = evaluate(clock, t, **kw, **pp)
            torch.autograd.backward(list(out), g)
        torch.cuda.synchronize()
    res[mode] = {k: L.profile_read(k) for k in ("dynamic_eval_fwd", "dynamic_eval_bwd")}
    L.profile_enable(False)
    # wall time incl. torch glue (zero fill of the dense spline gradient, AccumulateGrad)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(100, 125):
        out = evaluate(clock, t, **kw, **pp)
        torch.autograd.backward(list(out), g)
    e1.record(); torch.cuda.synchronize()
    res[mode]["wall_us_per_frame"] = e0.elapsed_time(e1) * 1000 / 25
alg_fwd = N * (12 + 48 + 16 + 64 + 128 + 4 + 12 + 12 + 16 + 4 + 12)
print(json.dumps({"N": N, "I": I, "alg_bytes_fwd": alg_fwd, "res": res}, default=str))
