"""ctypes binding of libsplat_hip.so (C ABI: include/splat_hip.h).

The product path has NO fallback: if the HIP library is missing or a tensor is not a contiguous
float32/int32 CUDA tensor the call raises.  (The CPU oracle lives in ``oracle/`` and is never
imported from here.)
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPLAT_LIB_PATH: alternative build of the same ABI (kernel tuning A/B runs); default = in-tree library
LIB_PATH = os.environ.get("SPLAT_LIB_PATH") or os.path.join(_HERE, "libsplat_hip.so")
_lib: Optional[ctypes.CDLL] = None

ABI_VERSION = 21

# every symbol include/splat_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "splat_last_error", "splat_abi_version", "splat_build_id", "splat_set_deterministic", "splat_get_deterministic",
    "splat_fill_f32", "splat_set_option", "splat_get_option",
    "splat_project_point_forward", "splat_project_point_backward",
    "splat_compute_cov3d_forward", "splat_compute_cov3d_backward",
    "splat_ewa_project_forward", "splat_ewa_project_backward",
    "splat_compute_sh_forward", "splat_compute_sh_backward",
    "splat_bin_scratch_bytes", "splat_bin_count", "splat_bin_sort",
    "splat_compute_gaussian_key", "splat_compute_tile_gaussian_range",
    "splat_alpha_blending_forward", "splat_alpha_blending_backward", "splat_alpha_blending_forward_flags",
    "splat_alpha_blending_backward_flags", "splat_blend_pair_floats", "splat_blend_pack_floats",
    "splat_dynamic_eval_forward", "splat_dynamic_eval_backward",
    "splat_position_poly_fourier_forward", "splat_position_poly_fourier_backward",
    "splat_preprocess_ortho_forward", "splat_preprocess_ortho_backward",
    "splat_frame_preprocess_forward", "splat_frame_preprocess_backward",
    "splat_densify_accumulate", "splat_densify_update", "splat_densify_masks",
    "splat_compact_scratch_bytes", "splat_compact_scan", "splat_compact_rows",
    "splat_gather_rows_repeat", "splat_densify_split_sample", "splat_morton_keys",
    "splat_knn_grid_cells", "splat_knn_plan_bytes", "splat_knn_build", "splat_knn_scatter", "splat_knn_search",
    "splat_adam_step", "splat_adam_step_pattern", "splat_arap_energy",
    "splat_preprocess_ortho_forward_batch", "splat_bin_count_batch", "splat_bin_sort_batch",
    "splat_alpha_blending_forward_batch", "splat_blend_pair_stride", "splat_alpha_blending_backward_batch",
    "splat_frame_preprocess_forward_batch", "splat_frames_gauss_backward_dynamic",
    "splat_frames_count", "splat_frames_forward", "splat_frames_backward",
    "splat_frames_gauss_backward_static", "splat_alpha_blending_backward_batch_set", "splat_frames_gauss_backward_static_set",
    "splat_blend_sets_pair_stride", "splat_blend_sets_pack_floats", "splat_alpha_blending_backward_batch_sets",
    "splat_alpha_blending_forward_batch_sets", "splat_pair_records_segment_sum", "splat_alpha_blending_backward_batch_sets_packed",
    "splat_frames_gauss_backward_static_sets", "splat_frames_gauss_backward_dynamic_sets",
    "splat_preprocess_forward_batch_cam", "splat_frames_gauss_backward_static_cam", "splat_frames_gauss_backward_static_sets_cam",
    "splat_preprocess_persp_forward", "splat_preprocess_persp_backward",
    "splat_alpha_blending_forward_batch_sources", "splat_frames_gauss_backward_dynamic_sources",
    "splat_frames_gauss_backward_static_sources_cam", "splat_blend_sets_uses_forward_pack",
    "splat_dynamic_positions_batch_forward", "splat_dynamic_positions_batch_backward",
    "splat_arap_energy_batch", "splat_knn_brute_scratch_bytes", "splat_knn_brute_batch", "splat_l1_loss_grad",
    "splat_alpha_blending_backward_batch_sets_l1", "splat_bin_count_batch_reach", "splat_bin_sort_batch_reach",
    "splat_profile_enable", "splat_profile_reset", "splat_profile_read",
]

# environment switches of earlier rounds, applied ONCE at load THROUGH the ABI (splat_set_option): the library itself reads no
# environment variable.  {variable: (option key, value that turns the non-default on, option value)}
_ENV_OPTIONS = {"SPLAT_BWD_QUARTERS": ("bwd_quarters", "0", 0), "SPLAT_BWD_KERNEL": ("bwd_kernel_dpp", "dpp", 1),
                "SPLAT_SETS_STD": ("sets_std", "0", 0), "SPLAT_BIN_SLOT_KEYS": ("bin_slot_keys", "1", 1)}


class SplatError(RuntimeError):
    pass


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SplatError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C splatter_a_video_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.splat_last_error.restype = ctypes.c_char_p
        L.splat_abi_version.restype = ctypes.c_int
        L.splat_build_id.restype = ctypes.c_char_p
        L.splat_set_deterministic.argtypes = [ctypes.c_int]
        L.splat_set_deterministic.restype = None
        L.splat_get_deterministic.restype = ctypes.c_int
        L.splat_fill_f32.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p]
        L.splat_bin_scratch_bytes.restype = ctypes.c_size_t
        L.splat_bin_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.splat_knn_plan_bytes.restype = ctypes.c_size_t
        L.splat_compact_scratch_bytes.restype = ctypes.c_size_t
        L.splat_compact_scratch_bytes.argtypes = [ctypes.c_int]
        L.splat_blend_pack_floats.restype = ctypes.c_size_t
        L.splat_blend_pack_floats.argtypes = [ctypes.c_int]
        L.splat_blend_pair_floats.restype = ctypes.c_size_t
        L.splat_blend_pair_floats.argtypes = [ctypes.c_int, ctypes.c_int]
        L.splat_blend_pair_stride.restype = ctypes.c_size_t
        L.splat_blend_pair_stride.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.splat_blend_sets_pair_stride.restype = ctypes.c_size_t
        L.splat_blend_sets_pair_stride.argtypes = [ctypes.c_int]
        L.splat_blend_sets_pack_floats.restype = ctypes.c_size_t
        L.splat_blend_sets_pack_floats.argtypes = []
        L.splat_profile_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
        L.splat_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.splat_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.splat_knn_brute_scratch_bytes.restype = ctypes.c_size_t
        L.splat_knn_brute_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        if L.splat_abi_version() != ABI_VERSION:
            raise SplatError("libsplat_hip.so ABI version mismatch; rebuild it")
        _lib = L
        for var, (key, on, value) in _ENV_OPTIONS.items():
            if os.environ.get(var) == on:
                check(L.splat_set_option(key.encode(), ctypes.c_int(value)))
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise SplatError(f"libsplat_hip error {rc}: {lib().splat_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream() -> ctypes.c_void_p:
    """the current HIP stream of the current device (raw handle).  torch.cuda.current_stream() builds a Stream object
    through several Python layers (~10 us per call, once per native launch); the raw getter is one C call."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def cf(x: float) -> ctypes.c_float:
    return ctypes.c_float(float(x))


def ci(x: int) -> ctypes.c_int:
    return ctypes.c_int(int(x))


def need(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    """Device / dtype gate + contiguity (the reference calls .contiguous() on every input too)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA (ROCm) tensor")
    if _cur_device is not None and t.device.index != _cur_device():
        # kernels are launched on the CURRENT device's stream (stream() below): a tensor of another device would be
        # dereferenced there.  One process drives one GPU (frame-sharded DP); otherwise wrap the call in torch.cuda.device(...)
        raise ValueError(f"{name} lives on cuda:{t.device.index} but the current device is cuda:{_cur_device()}")
    if t.dtype != dtype:
        if dtype == torch.uint8 and t.dtype == torch.bool:
            t = t.contiguous().view(torch.uint8)
        else:
            raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def build_id() -> str:
    """hash of the sources the loaded library was built from (csrc/Makefile); measurement records carry it"""
    return lib().splat_build_id().decode()


def set_option(key: str, value: int) -> None:
    """process-wide option of the library (include/splat_hip.h: splat_set_option): "bwd_quarters", "bwd_kernel_dpp", "sets_std",
    "bin_slot_keys", "deterministic"; read at launch time"""
    check(lib().splat_set_option(key.encode(), ctypes.c_int(int(value))))


def get_option(key: str) -> int:
    v = ctypes.c_int(0)
    check(lib().splat_get_option(key.encode(), ctypes.byref(v)))
    return v.value


class option:
    """``with L.option("bwd_quarters", 0): ...`` -- an option for the duration of a block (tests)"""

    def __init__(self, key: str, value: int):
        self.key, self.value = key, int(value)

    def __enter__(self):
        self.old = get_option(self.key)
        set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.old)
        return False


class FeatureSource(ctypes.Structure):
    """mirror of splat_feature_source_t (include/splat_hip.h)"""
    _fields_ = [("c0", ctypes.c_int32), ("cn", ctypes.c_int32), ("feature", ctypes.c_void_p), ("d_feature", ctypes.c_void_p),
                ("frame_stride", ctypes.c_int64)]


MAX_SOURCES = 8


def set_deterministic(on: bool) -> None:
    """process-wide deterministic mode (include/splat_hip.h: splat_set_deterministic): no backward launches a kernel with
    float atomics; a foreign idx_sorted (atomic backward) raises"""
    lib().splat_set_deterministic(1 if on else 0)


def deterministic() -> bool:
    return bool(lib().splat_get_deterministic())


def profile_enable(on: bool) -> None:
    lib().splat_profile_enable(ctypes.c_int(1 if on else 0))


def profile_reset() -> None:
    lib().splat_profile_reset()


def profile_read(prefix: str = ""):
    ms = ctypes.c_double(0.0)
    n = ctypes.c_int(0)
    lib().splat_profile_read(prefix.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value
