"""The block-level backward kernels of a frame batch (blend_bwd_mfma_kernel with the forward's cull flags,
blend_bwd_sets_kernel) are what the library option "bwd_quarters" = 0 selects; by default a batch runs the quarter-list kernels.
The option is set through the ABI (splat_set_option) for the duration of a test: the oracle tests of the batch path run again,
in this process, on the block-level kernels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_cases():
    import test_gpu_frames_oracle as T
    return [("render", T.test_frame_batch_render_against_oracle_and_reference_geometry, {"reach": True}),
            ("render_sets_std", T.test_render_sets_against_oracle, {"std": 1, "width": 19}),
            ("render_sets_generic", T.test_render_sets_against_oracle, {"std": 0, "width": 19}),
            ("render_sets_mask", T.test_render_sets_against_oracle, {"std": 1, "width": 1}),
            ("per_frame_cameras_render", T.test_per_frame_cameras_against_oracle, {"entry": "render"}),
            ("per_frame_cameras_render_sets", T.test_per_frame_cameras_against_oracle, {"entry": "render_sets"}),
            ("wide_row", T.test_wide_row_batch_against_oracle, {})]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["render", "render_sets_std", "render_sets_generic", "render_sets_mask", "per_frame_cameras_render",
                                  "per_frame_cameras_render_sets", "wide_row"])
def test_frame_batch_oracle_tests_with_block_level_backward(case, oracle_mod, lib_option):
    fn, kw = next((f, k) for n, f, k in _oracle_cases() if n == case)
    lib_option("bwd_quarters", 0)
    if "std" in kw:
        kw = dict(kw, lib_option=lib_option)
    fn(oracle_mod, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("C,with_abs", [(3, False), (3, True), (1, False), (19, False), (32, False)])
def test_single_frame_backward_with_and_without_the_forwards_cull_words(C, with_abs):
    """ABI 17: splat_alpha_blending_forward_flags / _backward_flags against the plain calls on the same inputs -- the backward that
    gets the forward's cull words (quarter-list kernels) and the one that culls again (block-level kernels) are two routes to
    the same gradients (other summation order: element-wise 2e-4 / 2e-6 of the maximum); images, final_T and ncontrib of the two
    forwards are identical."""
    import numpy as np
    import torch
    import dptr.gs as gs
    from splatter_a_video_amd import _lib as L
    from splatter_a_video_amd.gs.raster_ops import _find_pairmap
    from splatter_a_video_amd.synth import make_scene
    from test_gpu_parity import oracle_geometry
    import oracle as o

    N, W, H = 6000, 160, 112
    sc = make_scene(N, W, H, seed=5 + C)
    G = oracle_geometry(o, sc)
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), device="cuda", dtype=dt)
    uv, conic, opac = t(G["uv"]), t(G["conic"]), t(sc.opacity)
    rng = np.random.default_rng(C)
    feat = t(rng.uniform(size=(N, C)).astype(np.float32))
    g = t(rng.normal(size=(C, H, W)).astype(np.float32))
    idx, tr = gs.sort_gaussian(uv, t(G["depth"]), W, H, t(G["radius"]), t(G["tiles"]))
    pm = _find_pairmap(idx, tr, N)
    assert pm is not None
    M = idx.numel()
    lib = L.lib()
    res = []
    for use_flags in (False, True):
        out = torch.empty(C, H, W, device="cuda"); fT = torch.empty(H, W, device="cuda")
        nc = torch.empty(H, W, dtype=torch.int32, device="cuda")
        pack = torch.empty(N * lib.splat_blend_pack_floats(C), device="cuda")
        flags = torch.empty(M, dtype=torch.int32, device="cuda") if use_flags else None
        head = (L.ci(N), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opac), L.ptr(feat), L.ptr(None), L.ptr(idx), L.ptr(tr), L.cf(0.2),
                L.ptr(None), L.ci(W), L.ci(H), L.ci(0), L.ci(0), L.ptr(out), L.ptr(fT), L.ptr(nc), L.ptr(None), L.ptr(pack))
        if use_flags:
            L.check(lib.splat_alpha_blending_forward_flags(*head, L.ptr(flags), L.stream()))
        else:
            L.check(lib.splat_alpha_blending_forward(*head, L.stream()))
        duv = torch.empty(N, 2, device="cuda"); dabs = torch.empty(N, 2, device="cuda") if with_abs else None
        dcon = torch.empty(N, 3, device="cuda"); dop = torch.empty(N, 1, device="cuda"); dfe = torch.empty(N, C, device="cuda")
        scratch = torch.empty(M * lib.splat_blend_pair_floats(C, 0), device="cuda")
        args = (L.ci(N), L.ci(C), L.ptr(uv), L.ptr(conic), L.ptr(opac), L.ptr(feat), L.ptr(None), L.ptr(idx), L.ptr(tr), L.cf(0.2),
                L.ci(W), L.ci(H), L.ptr(fT), L.ptr(nc), L.ptr(g), L.ptr(duv), L.ptr(dabs), L.ptr(dcon), L.ptr(dop), L.ptr(dfe),
                L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(pm.goff), L.ptr(pm.slot_sorted), L.ptr(scratch), L.ptr(pack), L.ci(1),
                L.ptr(None))
        if use_flags:
            L.check(lib.splat_alpha_blending_backward_flags(*args, L.ptr(flags), L.stream()))
        else:
            L.check(lib.splat_alpha_blending_backward(*args, L.stream()))
        torch.cuda.synchronize()
        res.append(dict(out=out, fT=fT, nc=nc, duv=duv, dabs=dabs, dcon=dcon, dop=dop, dfe=dfe))
    a, b = res
    assert torch.equal(a["out"], b["out"]) and torch.equal(a["fT"], b["fT"]) and torch.equal(a["nc"], b["nc"])
    for k in ("duv", "dcon", "dop", "dfe") + (("dabs",) if with_abs else ()):
        x, y = b[k], a[k]
        d = (x - y).abs()
        tol = 2e-4 * y.abs() + 2e-6 * float(y.abs().max()) + 1e-12
        assert int((d > tol).sum()) <= max(2, x.numel() // 50000), (k, int((d > tol).sum()), float(d.max()))
        assert bool((d <= 10 * tol).all()), k


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sort_5000", "sort_crowded", "ties_and_empty", "render"])
def test_sort_and_pair_map_with_slot_keys(case, gpu, oracle_mod, lib_option):
    """The pair map's low key word is (Gaussian id, tile index inside the splat's rectangle) whenever the two fit 32 bits --
    every size under test.  The library option "bin_slot_keys" = 1 selects the other form (pair slot in the key, ids through the
    `owner` workspace: what 1M Gaussians on more than 4096 tiles get): the sort's bit-exactness tests and the batch's oracle test
    run again on that form."""
    import test_gpu_frames_oracle as T
    import test_gpu_parity as P
    lib_option("bin_slot_keys", 1)
    if case == "sort_5000":
        P.test_sort_gaussian_bit_exact(gpu, oracle_mod, 5000, 256, 256, 2.0)
    elif case == "sort_crowded":
        P.test_sort_gaussian_bit_exact(gpu, oracle_mod, *_crowded_sort_case())
    elif case == "ties_and_empty":
        P.test_sort_gaussian_ties_and_empty(gpu, oracle_mod)
    else:
        for reach in (True, False):      # slot keys count the kept tiles under reach masks as the packed keys do
            T.test_frame_batch_render_against_oracle_and_reference_geometry(oracle_mod, reach)


def _crowded_sort_case():
    """the largest parametrisation of test_sort_gaussian_bit_exact (a tile above 2048 keys)"""
    import test_gpu_parity as P
    marks = [m for m in P.test_sort_gaussian_bit_exact.pytestmark if m.name == "parametrize"]
    return max(marks[0].args[1], key=lambda c: c[0] * c[3])
