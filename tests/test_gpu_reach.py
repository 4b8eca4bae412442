"""Reach masks (splat_bin_count_batch_reach / splat_bin_sort_batch_reach, ABI 21): the frame batch creates only the (Gaussian,
tile) pairs whose tile the splat can reach with alpha >= 1/255.  The reference creates a pair for every tile of the bounding
square (include/utils.h:17-37) and skips per pixel (src/alpha_blending.cu:78-95): a dropped pair must not change a bit of the
images or ids, and the gradients only by summation order."""
import ctypes

import numpy as np
import pytest
import torch

import splatter_a_video_amd._lib as L
from splatter_a_video_amd import frames as FR
from splatter_a_video_amd.frames import FrameBatch
from splatter_a_video_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), device="cuda", requires_grad=grad)


def _render(sc, F, C, feat, g, reach, big=False):
    old = FR.OPTIONS["reach"]
    FR.OPTIONS["reach"] = reach
    try:
        off = _t(np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32))
        p = {k: _t(v, True) for k, v in dict(xyz=sc.xyz, scales=sc.scale, uquats=sc.rotate, opacity=sc.opacity, feature=feat).items()}
        B = FrameBatch(F, sc.N, sc.W, sc.H, C, "cuda", want_abs=True)
        out = B.render(p["xyz"], p["scales"], p["uquats"], p["opacity"], p["feature"], off, _t(sc.extr), bg=0.2)
        out.backward(g)
        torch.cuda.synchronize()
        return out.detach(), B, p
    finally:
        FR.OPTIONS["reach"] = old


def _scene(N, W, H, kind):
    sc = make_scene(N, W, H, seed=N + 17)
    rng = np.random.default_rng(N)
    if kind == "big":          # splats whose rectangle holds 32 tiles and more (up to the whole image): the cell form of the reach word
        k = rng.choice(N, 60, replace=False)
        sc.scale[k] *= 12.0
        sc.scale[k[:20], 0] *= 4.0          # elongated: most of the bounding square is empty
        sc.scale[k[40:]] *= 4.0             # cells of 4 x 4 / 8 x 8 tiles
        sc.opacity[k[:10]] = 0.9
    if kind == "faint":        # below 1/255 a splat contributes nowhere: no pair at all
        sc.opacity[: N // 2] = 0.0035
    if kind == "elongated":
        sc.scale[:, 0] *= 4.0
        sc.scale[:, 1] *= 0.4
    return sc, rng


@pytest.mark.parametrize("kind,N,W,H,C", [("plain", 20000, 256, 192, 3), ("big", 4000, 320, 256, 3), ("faint", 6000, 128, 128, 3),
                                          ("elongated", 8000, 200, 120, 3), ("plain", 5000, 96, 64, 19), ("plain", 4000, 96, 64, 32)])
def test_reach_masks_change_no_bit_of_the_images(kind, N, W, H, C):
    F = 3
    sc, rng = _scene(N, W, H, kind)
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    g = _t(rng.normal(size=(F, C, H, W)).astype(np.float32))
    out_r, Br, pr = _render(sc, F, C, feat, g, True)
    out_f, Bf, pf = _render(sc, F, C, feat, g, False)
    assert torch.equal(out_r, out_f) and torch.equal(Br.final_T, Bf.final_T)
    assert torch.equal(Br.ncontrib > 0, Bf.ncontrib > 0) and bool((Br.ncontrib <= Bf.ncontrib).all())
    mr, mf = Br.check(), Bf.check()
    assert mr < mf
    if kind == "big":
        area = (Bf.goff[0].long() - torch.cat([Bf.goff[0, :1] * 0, Bf.goff[0, :-1]]).long())
        kept = (Br.goff[0].long() - torch.cat([Br.goff[0, :1] * 0, Br.goff[0, :-1]]).long())
        big = area >= 32
        assert int(area.max()) >= 32 and int((Br.reach[0] < 0).sum()) > 0     # rectangles of 32 tiles and more: the CELL form of the
        assert bool((kept <= area).all()) and int((kept[big] < area[big]).sum()) > 0   # reach word (bit 31) -- and it culls
    if kind == "faint":
        kept = Br.goff[0].long() - torch.cat([Br.goff[0, :1] * 0, Br.goff[0, :-1]]).long()
        assert int(kept[: N // 2].sum()) == 0
    for k in pr:
        a, b = pr[k].grad, pf[k].grad
        tol = 3e-4 * float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= tol, k
    assert float((Br.tap - Bf.tap).abs().max()) <= 3e-4 * float(Bf.tap.abs().max()) + 1e-12
    assert float((Br.abs_tap - Bf.abs_tap).abs().max()) <= 3e-4 * float(Bf.abs_tap.abs().max()) + 1e-12
    assert torch.equal(Br.radii_max, Bf.radii_max)


def test_reach_entry_points_raw():
    """the two entry points through the C ABI with F = 1: kept tiles per Gaussian, their prefix, M and the tile ranges agree; a
    kept pair list is a sub-list of the full one in the same order"""
    N, W, H = 5000, 160, 96
    sc = make_scene(N, W, H, seed=3)
    sc.scale[:60] *= 12.0            # rectangles of 32 tiles and more: the scatter step derives their kept count from the cell word
    sc.scale[:20, 1] *= 5.0
    lib = L.lib()
    import dptr.gs as gs
    uv, depth, conic, radius, tiles = gs.preprocess_ortho(_t(sc.xyz), _t(sc.scale), _t(sc.rotate), _t(sc.extr), W, H, nearest=0.01)
    op = _t(sc.opacity)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    i32 = dict(dtype=torch.int32, device="cuda")
    scratch = torch.empty(lib.splat_bin_scratch_bytes(N, W, H), dtype=torch.uint8, device="cuda")
    tr, m, gcount, reach = torch.empty(T, 2, **i32), torch.zeros(1, **i32), torch.empty(N, **i32), torch.empty(N, **i32)
    st = L.stream()
    L.check(lib.splat_bin_count_batch_reach(L.ci(1), L.ci(N), L.ptr(uv), L.ptr(radius), L.ptr(conic), L.ptr(op), ctypes.c_int64(0),
                                            L.ci(W), L.ci(H), L.ptr(scratch), L.ptr(tr), L.ptr(m), L.ptr(gcount), L.ptr(reach), st))
    M = int(m.item())
    assert M == int(gcount.sum()) and 0 < M < int(tiles.sum())
    assert int((tiles.view(-1) >= 32).sum()) > 10 and int((reach < 0).sum()) > 10
    assert bool((gcount <= tiles.view(-1)).all())
    keys, idx, owner, slot = torch.empty(M, dtype=torch.int64, device="cuda"), torch.empty(M, **i32), torch.empty(M, **i32), torch.empty(M, **i32)
    goff, ovf = torch.empty(N, **i32), torch.zeros(1, **i32)
    L.check(lib.splat_bin_sort_batch_reach(L.ci(1), L.ci(N), L.ptr(uv), L.ptr(depth), L.ptr(radius), L.ptr(reach), L.ci(W), L.ci(H),
                                           L.ptr(scratch), L.ptr(tr), ctypes.c_int64(M),
                                           L.ptr(keys), L.ptr(idx), L.ptr(ovf), L.ptr(goff), L.ptr(owner), L.ptr(slot), st))
    torch.cuda.synchronize()
    assert int(ovf.item()) == 0 and torch.equal(goff, torch.cumsum(gcount, 0).int())
    assert int(tr[:, 1].max()) == M and sorted(slot.tolist()) == list(range(M))
    full_idx, full_tr = gs.sort_gaussian(uv, depth, W, H, radius, tiles)
    full_idx, full_tr, idx_c, tr_c = full_idx.cpu().numpy(), full_tr.cpu().numpy(), idx.cpu().numpy(), tr.cpu().numpy()
    for t in range(T):
        a, b = idx_c[tr_c[t, 0]:tr_c[t, 1]], full_idx[full_tr[t, 0]:full_tr[t, 1]]
        it = iter(b.tolist())
        assert all(x in it for x in a.tolist()), t          # a sub-sequence of the full list, order kept
    # every pair slot belongs to the Gaussian the sorted entry names
    g_of_slot = torch.repeat_interleave(torch.arange(N, device="cuda"), gcount.long())
    assert torch.equal(g_of_slot[slot.long()].int(), idx)


@pytest.mark.parametrize("seed", range(12))
def test_reach_masks_on_random_scenes(seed):
    """random sizes, splat sizes from sub-pixel to a quarter of the image (cell form of the reach word), elongation, opacities down to
    below 1/255, centres off the image, clustered layouts: images, final_T and the contributing pixels of the masked lists equal the
    full lists' bit for bit (tools/reach_stress.py runs the same check over more cases)"""
    rng = np.random.default_rng(977 + seed)
    N = int(rng.integers(300, 12000)); W = int(rng.integers(40, 400)); H = int(rng.integers(40, 300)); F = int(rng.integers(1, 4))
    sig = float(np.exp(rng.uniform(np.log(0.4), np.log(0.25 * min(W, H)))))
    sc = make_scene(N, W, H, seed=int(rng.integers(1 << 30)), sigma_px=sig, clustered=float(rng.choice([0.0, 0.7])))
    kind = seed % 4
    if kind == 1:
        sc.scale[:, int(rng.integers(0, 2))] *= float(rng.uniform(2, 12))
    if kind == 2:
        sc.opacity[:] = (10.0 ** rng.uniform(-3.2, 0.0, size=sc.opacity.shape)).astype(np.float32)
    if kind == 3:
        sc.xyz[:, :2] *= 1.6
    C = int(rng.choice([3, 19]))
    feat = rng.uniform(size=(N, C)).astype(np.float32)
    off = _t(np.stack([sc.positions(f) - sc.xyz for f in range(F)]).astype(np.float32))
    res = []
    old = FR.OPTIONS["reach"]
    try:
        for reach in (True, False):
            FR.OPTIONS["reach"] = reach
            B = FrameBatch(F, N, W, H, C, "cuda")
            with torch.no_grad():
                out = B.render(_t(sc.xyz), _t(sc.scale), _t(sc.rotate), _t(sc.opacity), _t(feat), off, _t(sc.extr), bg=0.1)
            torch.cuda.synchronize()
            res.append((out.clone(), B.final_T.clone(), B.ncontrib.clone() > 0, B.check()))
    finally:
        FR.OPTIONS["reach"] = old
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert res[0][3] <= res[1][3]
