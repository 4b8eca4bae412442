"""The C-ABI library must build for gfx950 without a GPU, load, and export every symbol that
include/splat_hip.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "splatter_a_video_amd", "csrc"), "-j8"])
    import splatter_a_video_amd._lib as L
    return L


def test_header_symbols_exported(built_lib):
    header = open(os.path.join(ROOT, "include", "splat_hip.h")).read()
    declared = set(re.findall(r"\b(splat_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(built_lib.SYMBOLS), declared ^ set(built_lib.SYMBOLS)
    so = ctypes.CDLL(built_lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} not exported"


def test_abi_version_and_error_string(built_lib):
    lib = built_lib.lib()
    assert lib.splat_abi_version() == built_lib.ABI_VERSION
    assert isinstance(lib.splat_last_error(), bytes)


def test_argument_validation_without_gpu(built_lib):
    """Bad sizes are rejected before any HIP call, so this runs on a CPU-only box."""
    lib = built_lib.lib()
    rc = lib.splat_project_point_forward(ctypes.c_int(-1), None, None, None, 16, 16, ctypes.c_float(0.2),
                                         ctypes.c_float(1.3), 0, None, None, None)
    assert rc == -1
    assert b"bad sizes" in lib.splat_last_error()
    assert lib.splat_bin_scratch_bytes(1000, 64, 64) > 0
    assert lib.splat_bin_scratch_bytes(-1, 64, 64) == 0


def test_ops_refuse_cpu_tensors(built_lib):
    import torch
    import dptr.gs as gs
    with pytest.raises(ValueError):
        gs.compute_cov3d(torch.ones(4, 3), torch.ones(4, 4))
    with pytest.raises(ValueError):
        gs.project_point(torch.zeros(4, 3), torch.ones(4), torch.eye(4), 32, 32)


def test_operator_surface_matches_reference_names():
    import inspect
    import dptr.gs as gs
    expect = {
        "project_point": ["xyz", "intr", "extr", "W", "H", "nearest", "extent"],
        "compute_cov3d": ["scales", "uquats", "visible"],
        "ewa_project": ["xyz", "cov3d", "intr", "extr", "uv", "W", "H", "visible"],
        "sort_gaussian": ["uv", "depth", "W", "H", "radius", "tiles"],
        "compute_sh": ["shs", "degree", "view_dirs", "visible"],
        "compute_sh_free": ["shs", "degree", "view_dirs", "visible"],
        "alpha_blending": ["uv", "conic", "opacity", "feature", "idx_sorted", "title_bins", "bg", "W", "H", "ndc", "abs_ndc"],
        "alpha_blending_enhanced": ["uv", "conic", "opacity", "feature", "idx_sorted", "title_bins", "bg", "W", "H", "ndc",
                                    "abs_ndc", "K", "enable_truncation"],
        "alpha_blending_with_bias": ["uv", "conic", "opacity", "feature", "opacity_bias", "idx_sorted", "title_bins", "bg",
                                     "W", "H", "ndc", "abs_ndc"],
        "rasterization": ["xyz", "scale", "rotate", "opacity", "feature", "intr", "extr", "W", "H", "bg", "ndc"],
    }
    for name, params in expect.items():
        sig = inspect.signature(getattr(gs, name))
        assert list(sig.parameters) == params, name
    assert inspect.signature(gs.project_point).parameters["nearest"].default == 0.2
    assert inspect.signature(gs.project_point).parameters["extent"].default == 1.3
    assert inspect.signature(gs.alpha_blending_enhanced).parameters["K"].default == 10


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: without the HIP library the binding raises instead of computing something else."""
    import subprocess
    import sys
    code = ("import os; os.environ['SPLAT_LIB_PATH'] = r'%s/nope.so'\n"
            "import splatter_a_video_amd._lib as L\n"
            "try:\n    L.lib()\nexcept L.SplatError as e:\n    print('RAISED', 'no CPU fallback' in str(e).lower() or 'not found' in str(e))\n" % tmp_path)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_product_never_imports_oracle():
    """The shipped package must not reference the oracle (checker only)."""
    import pathlib
    for p in pathlib.Path(ROOT, "splatter_a_video_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p
    for p in pathlib.Path(ROOT, "dptr").rglob("*.py"):
        assert "oracle" not in p.read_text(), p


def test_extension_modules_refuse_cpu(built_lib):
    """the modules beyond the dptr.gs surface have no CPU fallback either"""
    import torch
    import dptr.gs as gs
    from splatter_a_video_amd.densify import DensifyState, compact
    from splatter_a_video_amd.dynamics import FrameClock, frame_preprocess
    from splatter_a_video_amd.knn import distCUDA2, knn_points
    with pytest.raises(ValueError):
        DensifyState(10, "cpu")
    with pytest.raises(ValueError):
        compact(torch.ones(4, dtype=torch.bool), {"x": torch.zeros(4, 3)})
    with pytest.raises(ValueError):
        knn_points(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3), None, None, K=3)
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(8, 3))
    with pytest.raises(ValueError):
        gs.preprocess_ortho(torch.zeros(4, 3), torch.ones(4, 3), torch.ones(4, 4), torch.eye(4), 32, 32)
    z = lambda *s: torch.zeros(*s)
    with pytest.raises(ValueError):
        frame_preprocess(FrameClock(10), 3, torch.eye(4), 32, 32, position=z(4, 3), pos_cubic_node=z(4, 24), rotation=z(4, 4),
                         rot_poly_feat=z(4, 4, 4), rot_fourier_feat=z(4, 8, 4), opacity=z(4, 1), scaling=z(4, 3))


def test_new_entry_points_validate_arguments(built_lib):
    lib = built_lib.lib()
    f = ctypes.c_float
    assert lib.splat_knn_search(ctypes.c_int(4), None, None, 4, None, None, None, ctypes.c_int(99), None, None, None) == -1
    assert b"K must be" in lib.splat_last_error()
    assert lib.splat_compact_rows(ctypes.c_int(4), None, None, ctypes.c_int(0), None, None, None) == -1
    assert lib.splat_dynamic_eval_forward(ctypes.c_int(4), ctypes.c_int(2), ctypes.c_int(5), f(0.0), None, None, None, 0, None,
                                          None, None, None, None, None, None, None, None, None) == -1
    assert b"segment index" in lib.splat_last_error()
    assert lib.splat_knn_grid_cells(ctypes.c_int(300000)) == 150000
    assert lib.splat_compact_scratch_bytes(ctypes.c_int(1000)) >= 16


def test_diff_gaussian_rasterization_shim_surface():
    """the names and fields the reference's alternate renderer uses (base_splatting.py:17,123-139) resolve, argument
    errors come before any device work, and CPU tensors are refused (no CPU fallback)"""
    import torch

    import diff_gaussian_rasterization as dgr
    from splatter_a_video_amd import diff_rasterizer

    assert dgr.GaussianRasterizer is diff_rasterizer.GaussianRasterizer
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    eye = torch.eye(4)
    s = dgr.GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                          scale_modifier=1.0, viewmatrix=eye, projmatrix=eye, sh_degree=0,
                                          campos=torch.zeros(3), prefiltered=False, debug=False)
    rast = dgr.GaussianRasterizer(raster_settings=s)
    xyz, op = torch.rand(8, 3), torch.rand(8, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(xyz, None, op, scales=torch.rand(8, 3), rotations=torch.rand(8, 4))
    with pytest.raises(Exception, match="3D covariance"):
        rast(xyz, None, op, colors_precomp=torch.rand(8, 3))
    with pytest.raises((ValueError, RuntimeError)):
        rast(xyz, None, op, colors_precomp=torch.rand(8, 3), scales=torch.rand(8, 3), rotations=torch.rand(8, 4))


def test_dptr_C_shim_exports_the_reference_pybind_names():
    """dptr.gs._C carries the 18 names of the reference's pybind module (src/submodules/dptr/dptr/gs/src/ext.cpp:14-33)"""
    import dptr.gs._C as _C
    names = ["project_point_forward", "project_point_backward", "compute_cov3d_forward", "compute_cov3d_backward",
             "ewa_project_forward", "ewa_project_backward", "compute_gaussian_key", "compute_tile_gaussian_range",
             "compute_sh_forward", "compute_sh_backward", "alpha_blending_forward", "alpha_blending_backward",
             "alpha_blending_forward_enhanced", "alpha_blending_backward_enhanced", "compute_sh_free_forward",
             "compute_sh_free_backward", "alpha_blending_forward_with_bias", "alpha_blending_backward_with_bias"]
    assert sorted(names) == sorted(_C.__all__)
    for n in names:
        assert callable(getattr(_C, n))


def test_frames_struct_mirror_matches_the_header(built_lib):
    """splat_frames_t (one-call frame batch): the ctypes mirror has the library's sizeof -- a struct of the right size
    gets past the ABI check and is then refused for its missing capacity; a wrong size is refused as such"""
    import ctypes
    from splatter_a_video_amd.frames import _SplatFrames
    lib = built_lib.lib()
    b = _SplatFrames()
    b.struct_bytes = ctypes.sizeof(_SplatFrames)
    assert lib.splat_frames_forward(ctypes.byref(b)) != 0
    assert b"capacity" in lib.splat_last_error()
    b.struct_bytes = ctypes.sizeof(_SplatFrames) - 8
    assert lib.splat_frames_forward(ctypes.byref(b)) != 0
    assert b"ABI version" in lib.splat_last_error()


def test_one_pass_plan_routes_feature_sets_to_the_three_groups():
    """frames._one_pass_plan: which set lists the three-set backward kernel can serve (one set per routing group -- taps /
    live opacity / detached opacity -- of at most 4 / 4 / 20 channels, C <= 28), and the (c0, cn, bg) tables it gets."""
    from splatter_a_video_amd.frames import _one_pass_plan, _set_groups
    real = (((3, 0.2, False, True), ("depth", 1.0, False, False), (19, 0.0, True, False)), [3, 1, 19])     # render_iter
    assert _set_groups(real[0]) == [0, 1, 2]
    c0, cn, bg, depth_ch, tap = _one_pass_plan(real[0], real[1], 23)
    assert (c0, cn, bg, depth_ch, tap) == ([0, 3, 4], [3, 1, 19], [0.2, 1.0, 0.0], 3, 0)
    # any order of the sets; a missing group has width 0
    c0, cn, bg, depth_ch, tap = _one_pass_plan(((8, 0.3, True, False), (3, 0.1, False, True)), [8, 3], 11)
    assert (c0, cn, depth_ch, tap) == ([8, 0, 0], [3, 0, 8], -1, 1)
    assert _one_pass_plan(((3, 0.0, False, True),), [3], 3)[1] == [3, 0, 0]
    # not servable: two sets of one group, a group too wide, taps from a detached set, a row wider than 28
    assert _one_pass_plan(((3, 0.0, False, False), (2, 0.0, False, False)), [3, 2], 5) is None
    assert _one_pass_plan(((5, 0.0, False, True),), [5], 5) is None
    assert _one_pass_plan(((21, 0.0, True, False),), [21], 21) is None
    assert _one_pass_plan(((3, 0.0, True, True),), [3], 3) is None
    assert _one_pass_plan(((4, 0.0, False, True), (4, 0.0, False, False), (20, 0.0, True, False)), [4, 4, 20], 28) is not None


def test_sets_entry_points_validate_arguments(built_lib):
    L = built_lib.lib()
    i3, f3 = ctypes.c_int32 * 3, ctypes.c_float * 3
    assert L.splat_blend_sets_pair_stride(23) == 36 and L.splat_blend_sets_pair_stride(28) == 40
    # sets that do not tile the row / too wide / null tables: refused before any launch
    one = ctypes.c_void_p(8)
    args = lambda c0, cn, C: (1, 10, C, i3(*c0), i3(*cn), f3(0, 0, 0), one, one, one, ctypes.c_int64(0), one, ctypes.c_int64(0), None,
                              None, one, one, ctypes.c_int64(100), 32, 32, one, one, one, None, 0, one, one, one, None, None, None)
    assert L.splat_alpha_blending_backward_batch_sets(*args([0, 3, 4], [3, 1, 18], 23)) == -1     # channel 22 uncovered
    assert b"cover" in L.splat_last_error()
    assert L.splat_alpha_blending_backward_batch_sets(*args([0, 3, 3], [3, 1, 19], 23)) == -1     # overlap
    assert L.splat_alpha_blending_backward_batch_sets(*args([0, 5, 6], [5, 1, 17], 23)) == -1     # tap set wider than 4
    assert L.splat_alpha_blending_backward_batch_sets(*args([0, 3, 4], [3, 1, 25], 29)) == -1     # C > 28
