"""gs.rasterization_ortho / gs.rasterization through frames.frame_rasterization (a pooled one-frame batch: ONE call of the C ABI per
direction) against the operator chain they replace (reference: src/submodules/dptr/dptr/gs/__init__.py:28-100; the orthographic
chain of dptr_ortho_enhanced.py:282-349)."""
import numpy as np
import pytest
import torch

import dptr.gs as gs
from splatter_a_video_amd import frames as FR
from splatter_a_video_amd.gs import raster_ops as RO
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _params(sc, feat):
    return {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scale=sc.scale, rotate=sc.rotate, opacity=sc.opacity, feature=feat).items()}


def _close(a, b, what):
    tol = 2e-4 * b.abs() + 2e-6 * float(b.abs().max()) + 1e-12
    assert bool(((a - b).abs() <= tol).all()), (what, float((a - b).abs().max()))


@pytest.mark.parametrize("C", [3, 19])
def test_rasterization_ortho_equals_the_operator_chain(C):
    N, W, H = 8000, 200, 120
    sc = make_scene(N, W, H, seed=21 + C)
    rng = np.random.default_rng(C)
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(C, H, W)).astype(np.float32))
    off = _t((sc.positions(3) - sc.xyz).astype(np.float32))
    extr = _t(sc.extr)
    a = _params(sc, feat)
    img = gs.rasterization_ortho(a["xyz"], a["scale"], a["rotate"], a["opacity"], a["feature"], extr, W, H, 0.3, offset=off)
    img.backward(g)
    b = _params(sc, feat)
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(b["xyz"], b["scale"], b["rotate"], extr, W, H, nearest=0.01, offset=off)
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    ref = gs.alpha_blending(uv, conic, b["opacity"], b["feature"], idx, tr, 0.3, W, H)
    ref.backward(g)
    assert torch.equal(img, ref)                          # the same kernels: bit-identical images
    for k in a:
        _close(a[k].grad, b[k].grad, k)


def test_rasterization_takes_the_one_call_path_and_matches_the_operators():
    """the reference's gs.rasterization (pinhole camera): the pooled one-frame batch and the operator chain give the same image and
    gradients; two views before one backward get two batches of the pool"""
    N, W, H, C = 6000, 160, 112, 3
    sc = make_scene(N, W, H, seed=5, ortho=False)
    rng = np.random.default_rng(1)
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(C, H, W)).astype(np.float32))
    intr, extr = _t(sc.intr), _t(sc.extr)
    res = {}
    for one_call in (True, False):
        RO.OPTIONS["rasterization_one_call"] = one_call
        try:
            p = _params(sc, feat)
            img = gs.rasterization(p["xyz"], p["scale"], p["rotate"], p["opacity"], p["feature"], intr, extr, W, H, 0.1)
            img2 = gs.rasterization(p["xyz"] + 0.01, p["scale"], p["rotate"], p["opacity"], p["feature"], intr, extr, W, H, 0.1)
            (img * g).sum().backward(retain_graph=False) if False else torch.autograd.backward([img, img2], [g, 0.5 * g])
            res[one_call] = (img.detach(), img2.detach(), {k: v.grad for k, v in p.items()})
        finally:
            RO.OPTIONS["rasterization_one_call"] = True
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for k in res[True][2]:
        _close(res[True][2][k], res[False][2][k], k)
    key = next(k for k in FR._FRAME_POOL if k[1:] == (N, W, H, C))
    assert 2 <= len(FR._FRAME_POOL[key]) <= FR.POOL_MAX and not any(getattr(b, "_pending", False) for b in FR._FRAME_POOL[key])


def test_a_frame_loop_is_captured_in_a_hip_graph_and_replays_bit_equal():
    """the library launches on torch's current stream with caller-owned buffers and no host synchronisation after the first call, so
    torch.cuda.graph captures a loop of frames (forward + backward with gradient sinks) and a replay gives the eager loop's bits"""
    N, W, H, C, F = 5000, 160, 96, 3, 4
    sc = make_scene(N, W, H, seed=9)
    rng = np.random.default_rng(2)
    p = _params(sc, rng.uniform(size=(N, C)).astype(np.float32))
    offs = torch.stack([_t((sc.positions(f) - sc.xyz).astype(np.float32)) for f in range(F)])
    g = _t(rng.normal(size=(C, H, W)).astype(np.float32))
    extr = _t(sc.extr)
    sink = {k: torch.zeros_like(p[n]) for k, n in dict(xyz="xyz", scales="scale", uquats="rotate", opacity="opacity", feature="feature").items()}

    def loop():
        for f in range(F):
            img = gs.rasterization_ortho(p["xyz"], p["scale"], p["rotate"], p["opacity"], p["feature"], extr, W, H, 0.1, offset=offs[f],
                                         grad_sink=sink)
            img.backward(g)

    loop()                                            # (first call: the pooled batch sizes its pair buffers -- one host sync)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loop()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loop()
    for v in sink.values():
        v.zero_()
    graph.replay()
    torch.cuda.synchronize()
    got = {k: v.clone() for k, v in sink.items()}
    for v in sink.values():
        v.zero_()
    loop()
    torch.cuda.synchronize()
    for k in sink:
        assert float(got[k].abs().max()) > 0 and torch.equal(got[k], sink[k]), k
    FR.frame_rasterization.last.check()
