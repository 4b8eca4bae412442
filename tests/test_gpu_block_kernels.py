"""The block-level backward kernels of a frame batch (blend_bwd_mfma_kernel with the forward's cull flags,
blend_bwd_sets_kernel) are what SPLAT_BWD_QUARTERS=0 selects; by default a batch runs the quarter-list kernels.  The switch
is read once per process, so the oracle tests of the batch path run again in a child process with the block-level kernels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_frame_batch_oracle_tests_with_block_level_backward():
    env = dict(os.environ, SPLAT_BWD_QUARTERS="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_frames_oracle.py"), "-q", "-x", "-m", "gpu",
                        "-k", "render_against or render_sets_against or per_frame_cameras or wide", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


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
def test_sort_and_pair_map_with_slot_keys():
    """The pair map's low key word is (Gaussian id, tile index inside the splat's rectangle) whenever the two fit 32 bits --
    every size under test.  SPLAT_BIN_SLOT_KEYS=1 (read once per process) selects the other form (pair slot in the key, ids
    through the `owner` workspace: what 1M Gaussians on more than 4096 tiles get): the sort's bit-exactness tests, the pair-map
    test and the batch's oracle test run again in a child process on that form."""
    env = dict(os.environ, SPLAT_BIN_SLOT_KEYS="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_frames_oracle.py"), "-q", "-x", "-m", "gpu",
                        "-k", "sort or pair_map or render_against", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
