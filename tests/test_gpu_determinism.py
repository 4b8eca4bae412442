"""Deterministic mode and stream coverage ("launches on the current stream").

The reference's backward is all float atomics (src/alpha_blending.cu:229-246): two runs differ in the last bits.  Here the
frame-batch entry points and the per-frame operators on this package's own sort write one pair record per (tile, splat)
and sum a Gaussian's records in slot order -- no float atomics anywhere on that path -- so identical inputs must give
BIT-identical images, gradients and taps run to run.  ``splat_set_deterministic(1)`` closes the remaining doors (the
block-level kernel's carried survivors, the foreign-index atomic kernel)."""
import numpy as np
import pytest
import torch

import dptr.gs as gs
from splatter_a_video_amd import _lib as L
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.gs import raster_ops as RO
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _offsets(sc, F):
    return np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32)


def _run_render(B, sc, off, featv, g):
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=featv).items()}
    out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, _t(sc.extr), bg=0.1)
    out.backward(g)
    torch.cuda.synchronize()
    res = {k: v.grad.clone() for k, v in p.items()}
    res.update(out=out.detach().clone(), tap=B.tap.clone(), radii=B.radii_max.clone(), final_T=B.final_T.clone(),
               ncontrib=B.ncontrib.clone())
    if B.abs_tap is not None:
        res["abs_tap"] = B.abs_tap.clone()
    return res


@pytest.mark.parametrize("N,W,H,F,C,abs_tap", [(60000, 854, 480, 4, 3, False), (20000, 256, 192, 3, 3, True),
                                               (20000, 256, 192, 3, 32, False), (20000, 256, 192, 3, 19, False)])
def test_two_identical_batches_are_bit_identical(N, W, H, F, C, abs_tap):
    sc = make_scene(N, W, H, seed=3 + C)
    rng = np.random.default_rng(N)
    off = _t(_offsets(sc, F))
    featv = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    a = _run_render(FrameBatch(F, N, W, H, C, "cuda", want_abs=abs_tap), sc, off, featv, g)
    Bb = FrameBatch(F, N, W, H, C, "cuda", want_abs=abs_tap)          # other buffers, other addresses
    junk = torch.empty(12345677, device="cuda")                         # (shifts the allocator)
    b = _run_render(Bb, sc, off, featv, g)
    c = _run_render(Bb, sc, off, featv, g)                              # and the same object again
    del junk
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(b[k], c[k]), k
    assert float(a["xyz"].abs().max()) > 0


def test_three_set_training_frame_is_bit_identical_run_to_run():
    """render_sets (rgb enhanced K = 20 with taps + depth + 19 attribute channels): the one-pass three-set backward"""
    N, W, H, F, K = 30000, 320, 240, 3, 20
    sc = make_scene(N, W, H, seed=21)
    rng = np.random.default_rng(2)
    off = _t(_offsets(sc, F))
    base = dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, rgb=rng.uniform(size=(N, 3)).astype(np.float32),
                attrs=rng.uniform(-1, 1, size=(N, 19)).astype(np.float32))
    gr = [_t(rng.normal(size=(F, c, H, W)).astype(np.float32)) for c in (3, 1, 19)]

    def run():
        p = {k: _t(v, True) for k, v in base.items()}
        B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
        sets = [dict(feature=p["rgb"], bg=0.0, taps=True), dict(feature="depth", bg=1.0),
                dict(feature=p["attrs"], bg=0.0, detach_opacity=True)]
        o = B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, off, _t(sc.extr), K=K)
        torch.autograd.backward(list(o[:3]), gr)
        torch.cuda.synchronize()
        res = {k: v.grad.clone() for k, v in p.items()}
        res.update(rgb_img=o[0].detach().clone(), dep_img=o[1].detach().clone(), att_img=o[2].detach().clone(), ids=o[3].clone(),
                   tap=B.tap.clone(), abs_tap=B.abs_tap.clone())
        return res

    a, b = run(), run()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_per_frame_operators_are_bit_identical_and_the_flag_refuses_foreign_lists():
    N, W, H, C = 20000, 256, 192, 3
    sc = make_scene(N, W, H, seed=8)
    rng = np.random.default_rng(8)
    featv = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(C, H, W)).astype(np.float32))

    def run(foreign=False):
        p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=featv).items()}
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(p["xyz"], p["scales"], p["uquats"], _t(sc.extr), W, H, nearest=0.01)
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        if foreign:
            idx = idx.clone()          # a copy is a foreign index list: the wave-reduced atomic kernel
        ndc = torch.zeros_like(uv, requires_grad=True)
        img = gs.alpha_blending(uv, conic, p["opacity"], p["feature"], idx, tr, 0.2, W, H, ndc)
        img.backward(g)
        torch.cuda.synchronize()
        return dict({k: v.grad.clone() for k, v in p.items()}, img=img.detach().clone(), tap=ndc.grad.clone())

    a, b = run(), run()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert not L.deterministic()
    import warnings
    L.set_deterministic(True)
    try:
        assert L.deterministic()
        c = run()                      # the pair path is what it was
        for k in a:
            assert torch.equal(a[k], c[k]), k
        with warnings.catch_warnings(), pytest.raises(L.SplatError, match="deterministic"):
            warnings.simplefilter("ignore")
            run(foreign=True)
    finally:
        L.set_deterministic(False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")    # (the one-time "foreign index list" warning)
        run(foreign=True)              # flag off: the atomic kernel serves a foreign list again


def test_block_level_kernel_gives_way_in_deterministic_mode():
    """splat_alpha_blending_backward without the forward's cull words runs the block-level matrix-core kernel, whose carried
    survivors add with float atomics; under the flag the DPP pair kernel takes over: two runs are bit-identical and agree with
    the default kernel to summation order"""
    from splatter_a_video_amd.gs.raster_ops import _find_pairmap
    N, W, H, C = 12000, 192, 128, 3
    sc = make_scene(N, W, H, seed=31)
    rng = np.random.default_rng(31)
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(_t(sc.xyz), _t(sc.scale), _t(sc.rotate), _t(sc.extr), W, H, nearest=0.01)
    opac, feat = _t(sc.opacity), _t(rng.uniform(size=(N, C)).astype(np.float32))
    g = _t(rng.normal(size=(C, H, W)).astype(np.float32))
    idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    pm = _find_pairmap(idx, tr, N)
    M = idx.numel()
    lib = L.lib()

    def run():
        out = torch.empty(C, H, W, device="cuda"); fT = torch.empty(H, W, device="cuda")
        nc = torch.empty(H, W, dtype=torch.int32, device="cuda")
        pack = torch.empty(N * lib.splat_blend_pack_floats(C), device="cuda")
        L.check(lib.splat_alpha_blending_forward(
            L.ci(N), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opac), L.ptr(feat), L.ptr(None), L.ptr(idx), L.ptr(tr), L.cf(0.2),
            L.ptr(None), L.ci(W), L.ci(H), L.ci(0), L.ci(0), L.ptr(out), L.ptr(fT), L.ptr(nc), L.ptr(None), L.ptr(pack), L.stream()))
        duv = torch.empty(N, 2, device="cuda"); dcon = torch.empty(N, 3, device="cuda")
        dop = torch.empty(N, 1, device="cuda"); dfe = torch.empty(N, C, device="cuda")
        scratch = torch.empty(M * lib.splat_blend_pair_floats(C, 0), device="cuda")
        L.check(lib.splat_alpha_blending_backward(
            L.ci(N), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opac), L.ptr(feat), L.ptr(None), L.ptr(idx), L.ptr(tr), L.cf(0.2),
            L.ci(W), L.ci(H), L.ptr(fT), L.ptr(nc), L.ptr(g), L.ptr(duv), L.ptr(None), L.ptr(dcon), L.ptr(dop), L.ptr(dfe),
            L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(pm.goff), L.ptr(pm.slot_sorted), L.ptr(scratch), L.ptr(pack), L.ci(1),
            L.ptr(None), L.stream()))
        torch.cuda.synchronize()
        return dict(duv=duv, dcon=dcon, dop=dop, dfe=dfe)

    ref = run()
    L.set_deterministic(True)
    try:
        a, b = run(), run()
    finally:
        L.set_deterministic(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        d = (a[k] - ref[k]).abs()
        assert bool((d <= 2e-3 * ref[k].abs() + 2e-5 * float(ref[k].abs().max()) + 1e-12).all()), k


def test_operators_run_on_a_non_default_stream():
    """every launch goes to torch's CURRENT stream.  A frame batch rendered inside ``torch.cuda.stream(s)`` -- with
    the default stream kept busy by a long unrelated kernel queue -- equals the one rendered on the default stream bit for
    bit, and its work is ordered on ``s`` (an event recorded on ``s`` after the backward covers it)."""
    N, W, H, F, C = 20000, 256, 192, 3, 3
    sc = make_scene(N, W, H, seed=14)
    rng = np.random.default_rng(14)
    off = _t(_offsets(sc, F))
    featv = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    ref = _run_render(FrameBatch(F, N, W, H, C, "cuda"), sc, off, featv, g)

    s = torch.cuda.Stream()
    B = FrameBatch(F, N, W, H, C, "cuda")
    p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=featv).items()}
    extr = _t(sc.extr)
    torch.cuda.synchronize()
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(30):                 # ~ tens of ms of work queued on the DEFAULT stream
        busy = busy @ busy
        busy = busy / busy.abs().max()
    with torch.cuda.stream(s):
        assert L.stream().value == s.cuda_stream
        out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, extr, bg=0.1)
        out.backward(g)
        done = torch.cuda.Event()
        done.record(s)
    done.synchronize()                  # only stream s is waited for
    got = {k: v.grad for k, v in p.items()}
    got.update(out=out.detach(), tap=B.tap)
    for k in got:
        assert torch.equal(got[k], ref[k]), k
    torch.cuda.synchronize()
    assert torch.isfinite(busy).all()

    # the per-frame operators on the same side stream
    with torch.cuda.stream(s):
        q = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=featv).items()}
        uv, depth, conic, radius, tiles = gs.preprocess_ortho(q["xyz"], q["scales"], q["uquats"], extr, W, H, nearest=0.01,
                                                              offset=off[0])
        idx, tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
        img = gs.alpha_blending(uv, conic, q["opacity"], q["feature"], idx, tr, 0.1, W, H)
        s.synchronize()
        assert torch.equal(img, ref["out"][0])


def test_three_set_backward_from_the_forwards_records_is_bit_identical(monkeypatch):
    """the three-set tile kernel staging the records the FORWARD packed (default for the renderer's own plan: no packing launch
    in the backward) against the same kernel on its own packed records (SPLAT_SETS_FWDREC=0): the same staged values, the same
    arithmetic -- bit-identical gradients and taps"""
    N, W, H, F, K = 30000, 320, 240, 3, 20
    sc = make_scene(N, W, H, seed=23)
    rng = np.random.default_rng(6)
    off = _t(_offsets(sc, F))
    base = dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, rgb=rng.uniform(size=(N, 3)).astype(np.float32),
                attrs=rng.uniform(-1, 1, size=(N, 19)).astype(np.float32))
    gr = [_t(rng.normal(size=(F, c, H, W)).astype(np.float32)) for c in (3, 1, 19)]
    res = []
    for flag in ("1", "0"):
        monkeypatch.setitem(RO.OPTIONS, "sets_fwdrec", flag == "1")
        p = {k: _t(v, True) for k, v in base.items()}
        B = FrameBatch(F, N, W, H, 23, "cuda", want_abs=True)
        sets = [dict(feature=p["rgb"], bg=0.1, taps=True), dict(feature="depth", bg=1.0),
                dict(feature=p["attrs"], bg=0.0, detach_opacity=True)]
        o = B.render_sets(p["xyz"], p["scales"], p["uquats"], p["opacity"], sets, off, _t(sc.extr), K=K)
        torch.autograd.backward(list(o[:3]), gr)
        torch.cuda.synchronize()
        r = {k: v.grad.clone() for k, v in p.items()}
        r.update(tap=B.tap.clone(), abs_tap=B.abs_tap.clone())
        res.append(r)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    assert float(res[0]["attrs"].abs().max()) > 0
