"""Timing of the three blends of the reference's render_iter (dptr_ortho_enhanced.py:331-375) at BASELINE configs[1]:
rgb (3 ch, enhanced K=20, ndc + abs_ndc taps), depth (1 ch), attributes (19 ch, opacity detached)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dptr.gs as gs
from splatter_a_video_amd.synth import make_scene

N, W, H = 300000, 854, 480
dev = "cuda"
sc = make_scene(N, W, H, C=3, seed=1234)
t = lambda x: torch.tensor(x, device=dev)
uv, depth, conic, radius, tiles = gs.preprocess_ortho(t(sc.positions(0)), t(sc.scale), t(sc.rotate), t(sc.extr), W, H, nearest=0.01)
idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
rgb = t(sc.feature).requires_grad_(); op = t(sc.opacity).requires_grad_()
attrs = torch.rand(N, 19, device=dev).requires_grad_()
uvg = uv.detach().requires_grad_(); cg = conic.detach().requires_grad_(); dg = depth.detach().requires_grad_()
g3, g1, g19 = (torch.randn(c, H, W, device=dev) for c in (3, 1, 19))

def timeit(name, fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us")

def rgb_enh():
    ndc = torch.zeros_like(uv, requires_grad=True); andc = torch.zeros_like(uv, requires_grad=True)
    img, nc, gi = gs.alpha_blending_enhanced(uvg, cg, op, rgb, idx, tr, 0.0, W, H, ndc, andc, K=20)
    img.backward(g3)
def rgb_plain():
    ndc = torch.zeros_like(uv, requires_grad=True)
    gs.alpha_blending(uvg, cg, op, rgb, idx, tr, 0.0, W, H, ndc).backward(g3)
def dep():
    gs.alpha_blending(uvg, cg, op, dg, idx, tr, 1.0, W, H, None).backward(g1)
def att():
    gs.alpha_blending(uvg, cg, op.detach(), attrs, idx, tr, 0.0, W, H, None).backward(g19)
with torch.no_grad():
    timeit("rgb enhanced K=20 forward only", lambda: gs.alpha_blending_enhanced(uv, conic, op, rgb, idx, tr, 0.0, W, H, None, None, K=20))
    timeit("attrs 19ch forward only", lambda: gs.alpha_blending(uv, conic, op, attrs, idx, tr, 0.0, W, H, None))
    timeit("depth 1ch forward only", lambda: gs.alpha_blending(uv, conic, op, depth, idx, tr, 1.0, W, H, None))
timeit("rgb plain 3ch fwd+bwd (ndc)", rgb_plain)
timeit("rgb enhanced K=20 fwd+bwd (ndc+abs_ndc)", rgb_enh)
timeit("depth 1ch fwd+bwd", dep)
timeit("attrs 19ch fwd+bwd", att)


def flow_separate():
    ndc = torch.zeros_like(uv, requires_grad=True); andc = torch.zeros_like(uv, requires_grad=True)
    i1, nc, gi = gs.alpha_blending_enhanced(uvg, cg, op, rgb, idx, tr, 0.0, W, H, ndc, andc, K=20)
    i2 = gs.alpha_blending(uvg, cg, op, dg, idx, tr, 1.0, W, H, ndc.detach())
    i3 = gs.alpha_blending(uvg, cg, op.detach(), attrs, idx, tr, 0.0, W, H, ndc.detach())
    torch.autograd.backward([i1, i2, i3], [g3, g1, g19])
def flow_shared():
    ndc = torch.zeros_like(uv, requires_grad=True); andc = torch.zeros_like(uv, requires_grad=True)
    i1, i2, i3, nc, gi = gs.alpha_blending_shared(uvg, cg, op, [rgb, dg, attrs], idx, tr, [0.0, 1.0, 0.0], W, H, ndc, andc, K=20,
                                                  detach_opacity=[False, False, True], taps=[True, False, False])
    torch.autograd.backward([i1, i2, i3], [g3, g1, g19])
timeit("three blends of render_iter, separate calls", flow_separate)
timeit("three blends of render_iter, one shared forward", flow_shared)
