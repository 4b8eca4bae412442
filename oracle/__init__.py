"""CPU ORACLE -- test infrastructure, NOT product code.

numpy front-end of ``oracle/splat_oracle.c`` (a scalar float32 restatement of the reference's
``dptr.gs._C`` CUDA extension, see the header of that file for pinning status).  The function
names mirror the 18 entry points of the reference's pybind module
(reference: src/submodules/dptr/dptr/gs/src/ext.cpp:14-33) plus the two orthographic torch
twins of src/pointrix/renderer/dptr_ortho_enhanced.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``splatter_a_video_amd`` never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "splat_oracle.c")
    if force or (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_get_threads.restype = ctypes.c_int
        _lib.oracle_has_openmp.restype = ctypes.c_int
    return _lib


def set_threads(n: int) -> None:
    lib().oracle_set_threads(ctypes.c_int(int(n)))


def get_threads() -> int:
    return int(lib().oracle_get_threads())


def has_openmp() -> bool:
    return bool(lib().oracle_has_openmp())


def _f(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _b(a):
    return np.ascontiguousarray(np.asarray(a).astype(bool).reshape(-1), dtype=np.uint8)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _extr12(extr):
    e = _f(extr).reshape(-1)
    assert e.size >= 12, "extr must hold at least 3x4 floats"
    return np.ascontiguousarray(e[:12])


def _tiles(W, H):
    return ((W + 15) // 16) * ((H + 15) // 16)


# ------------------------------------------------------------------ project_point
def project_point_forward(xyz, intr, extr, W, H, nearest=0.2, extent=1.3):
    xyz = _f(xyz, (-1, 3)); intr = _f(intr); extr = _extr12(extr)
    P = xyz.shape[0]
    uv = np.zeros((P, 2), np.float32); depth = np.zeros((P, 1), np.float32)
    lib().oracle_project_point_forward(P, _p(xyz, _f32p), _p(intr, _f32p), _p(extr, _f32p), int(W), int(H),
                                       ctypes.c_float(nearest), ctypes.c_float(extent),
                                       _p(uv, _f32p), _p(depth, _f32p))
    return uv, depth


def project_point_backward(xyz, intr, extr, W, H, uv, depth, dL_duv, dL_ddepth,
                           need_intr=True, need_extr=True):
    xyz = _f(xyz, (-1, 3)); intr = _f(intr); extr = _extr12(extr)
    P = xyz.shape[0]
    uv = _f(uv); depth = _f(depth); dL_duv = _f(dL_duv); dL_ddepth = _f(dL_ddepth)
    dxyz = np.zeros((P, 3), np.float32)
    dintr = np.zeros(4, np.float32); dextr = np.zeros((3, 4), np.float32)
    lib().oracle_project_point_backward(P, _p(xyz, _f32p), _p(intr, _f32p), _p(extr, _f32p), int(W), int(H),
                                        _p(uv, _f32p), _p(depth, _f32p), _p(dL_duv, _f32p), _p(dL_ddepth, _f32p),
                                        _p(dxyz, _f32p), _p(dintr if need_intr else None, _f32p),
                                        _p(dextr if need_extr else None, _f32p))
    return dxyz, dintr, dextr


def project_point_ortho_forward(xyz, extr, W, H, nearest=0.2, extent=1.3):
    xyz = _f(xyz, (-1, 3)); extr = _extr12(extr)
    P = xyz.shape[0]
    uv = np.zeros((P, 2), np.float32); depth = np.zeros((P, 1), np.float32)
    lib().oracle_project_point_ortho_forward(P, _p(xyz, _f32p), _p(extr, _f32p), int(W), int(H),
                                             ctypes.c_float(nearest), ctypes.c_float(extent),
                                             _p(uv, _f32p), _p(depth, _f32p))
    return uv, depth


def project_point_ortho_backward(extr, W, H, depth, dL_duv, dL_ddepth):
    extr = _extr12(extr); depth = _f(depth); dL_duv = _f(dL_duv); dL_ddepth = _f(dL_ddepth)
    P = depth.size
    dxyz = np.zeros((P, 3), np.float32)
    lib().oracle_project_point_ortho_backward(P, _p(extr, _f32p), int(W), int(H), _p(depth, _f32p),
                                              _p(dL_duv, _f32p), _p(dL_ddepth, _f32p), _p(dxyz, _f32p))
    return dxyz


# ------------------------------------------------------------------ compute_cov3d
def compute_cov3d_forward(scales, uquats, visible=None):
    scales = _f(scales, (-1, 3)); uquats = _f(uquats, (-1, 4))
    P = scales.shape[0]
    vis = _b(np.ones(P, bool) if visible is None else visible)
    cov = np.zeros((P, 6), np.float32)
    lib().oracle_compute_cov3d_forward(P, _p(scales, _f32p), _p(uquats, _f32p), _p(vis, _u8p), _p(cov, _f32p))
    return cov


def compute_cov3d_backward(scales, uquats, visible, dL_dcov3d):
    scales = _f(scales, (-1, 3)); uquats = _f(uquats, (-1, 4)); g = _f(dL_dcov3d, (-1, 6))
    P = scales.shape[0]
    vis = _b(np.ones(P, bool) if visible is None else visible)
    ds = np.zeros((P, 3), np.float32); dq = np.zeros((P, 4), np.float32)
    lib().oracle_compute_cov3d_backward(P, _p(scales, _f32p), _p(uquats, _f32p), _p(vis, _u8p), _p(g, _f32p),
                                        _p(ds, _f32p), _p(dq, _f32p))
    return ds, dq


# ------------------------------------------------------------------ ewa_project
def ewa_project_forward(xyz, cov3d, intr, extr, uv, W, H, visible=None, ortho=False):
    xyz = _f(xyz, (-1, 3)); cov3d = _f(cov3d, (-1, 6)); extr = _extr12(extr); uv = _f(uv, (-1, 2))
    intr = _f(np.zeros(4) if intr is None else intr)
    P = xyz.shape[0]
    vis = _b(np.ones(P, bool) if visible is None else visible)
    conic = np.zeros((P, 3), np.float32); radius = np.zeros(P, np.int32); tiles = np.zeros(P, np.int32)
    lib().oracle_ewa_project_forward(int(bool(ortho)), P, _p(xyz, _f32p), _p(cov3d, _f32p), _p(intr, _f32p),
                                     _p(extr, _f32p), _p(uv, _f32p), int(W), int(H), _p(vis, _u8p),
                                     _p(conic, _f32p), _p(radius, _i32p), _p(tiles, _i32p))
    return conic, radius, tiles


def ewa_project_backward(xyz, cov3d, intr, extr, radius, dL_dconic, W=0, H=0, ortho=False,
                         need_intr=True, need_extr=True):
    xyz = _f(xyz, (-1, 3)); cov3d = _f(cov3d, (-1, 6)); extr = _extr12(extr)
    intr = _f(np.zeros(4) if intr is None else intr)
    radius = _i(radius); g = _f(dL_dconic, (-1, 3))
    P = xyz.shape[0]
    dxyz = np.zeros((P, 3), np.float32); dcov = np.zeros((P, 6), np.float32)
    dintr = np.zeros(4, np.float32); dextr = np.zeros((3, 4), np.float32)
    lib().oracle_ewa_project_backward(int(bool(ortho)), P, _p(xyz, _f32p), _p(cov3d, _f32p), _p(intr, _f32p),
                                      _p(extr, _f32p), int(W), int(H), _p(radius, _i32p), _p(g, _f32p),
                                      _p(dxyz, _f32p), _p(dcov, _f32p),
                                      _p(dintr if need_intr else None, _f32p),
                                      _p(dextr if need_extr else None, _f32p))
    return dxyz, dcov, dintr, dextr


# ------------------------------------------------------------------ compute_sh
def compute_sh_forward(shs, degree, view_dirs, visible=None, free=False):
    shs = _f(shs); dirs = _f(view_dirs, (-1, 3))
    P = shs.shape[0]
    assert shs.size >= P * (degree + 1) ** 2 * 3
    vis = _b(np.ones(P, bool) if visible is None else visible)
    colors = np.zeros((P, 3), np.float32)
    clamped = np.ones((P, 3), np.uint8)
    lib().oracle_compute_sh_forward(int(bool(free)), P, _p(shs, _f32p), int(degree), _p(dirs, _f32p), _p(vis, _u8p),
                                    _p(colors, _f32p), _p(clamped, _u8p))
    if free:
        return colors
    return colors, clamped.astype(bool)


def compute_sh_backward(shs, degree, view_dirs, visible, clamped, dL_dcolors, free=False):
    shs = _f(shs); dirs = _f(view_dirs, (-1, 3)); g = _f(dL_dcolors, (-1, 3))
    P = shs.shape[0]
    vis = _b(np.ones(P, bool) if visible is None else visible)
    cl = None if free else np.ascontiguousarray(np.asarray(clamped).astype(np.uint8))
    dshs = np.zeros_like(shs); ddirs = np.zeros((P, 3), np.float32)
    lib().oracle_compute_sh_backward(int(bool(free)), P, _p(shs, _f32p), int(degree), _p(dirs, _f32p), _p(vis, _u8p),
                                     _p(cl, _u8p), _p(g, _f32p), _p(dshs, _f32p), _p(ddirs, _f32p))
    return dshs, ddirs


# ------------------------------------------------------------------ sort_gaussian
def cumsum_i32(tiles):
    tiles = _i(tiles)
    out = np.zeros_like(tiles)
    lib().oracle_cumsum_i32(tiles.size, _p(tiles, _i32p), _p(out, _i32p))
    return out


def compute_gaussian_key(uv, depth, W, H, radius, tiles_cumsum):
    uv = _f(uv, (-1, 2)); depth = _f(depth).reshape(-1); radius = _i(radius); cs = _i(tiles_cumsum)
    P = uv.shape[0]
    M = int(cs[-1]) if P > 0 else 0
    key = np.zeros(M, np.int64); idx = np.zeros(M, np.int32)
    if P > 0:
        lib().oracle_compute_gaussian_key(P, _p(uv, _f32p), _p(depth, _f32p), int(W), int(H), _p(radius, _i32p),
                                          _p(cs, _i32p), _p(key, _i64p), _p(idx, _i32p))
    return key, idx


def compute_tile_gaussian_range(W, H, key_sorted):
    key_sorted = np.ascontiguousarray(key_sorted, dtype=np.int64)
    tr = np.zeros((_tiles(W, H), 2), np.int32)
    lib().oracle_compute_tile_range(key_sorted.size, _p(key_sorted, _i64p), _p(tr, _i32p))
    return tr


def sort_gaussian(uv, depth, W, H, radius, tiles) -> Tuple[np.ndarray, np.ndarray]:
    """Python-level glue of the reference op (dptr/gs/sort_gaussian.py:42-52), stable order."""
    cs = cumsum_i32(tiles)
    key, idx = compute_gaussian_key(uv, depth, W, H, radius, cs)
    lib().oracle_sort_pairs(key.size, _p(key, _i64p), _p(idx, _i32p))
    tr = compute_tile_gaussian_range(W, H, key)
    return idx, tr


# ------------------------------------------------------------------ alpha blending
def alpha_blending_forward(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H,
                           K: int = 0, enable_truncation: bool = False, opacity_bias=None):
    """Returns (out[C,H,W], final_T[H,W], ncontrib[H,W]) and gs_idx[H,W,K] when K>0 (enhanced)."""
    uv = _f(uv, (-1, 2)); conic = _f(conic, (-1, 3)); opacity = _f(opacity).reshape(-1)
    feature = _f(feature); feature = feature.reshape(uv.shape[0], -1)
    idx_sorted = _i(idx_sorted); tile_range = _i(tile_range)
    bias = None if opacity_bias is None else _f(opacity_bias).reshape(-1)
    P, C = feature.shape
    out = np.zeros((C, H, W), np.float32); fT = np.zeros((H, W), np.float32); nc = np.zeros((H, W), np.int32)
    gi = None
    if K > 0:
        gi = np.full((H, W, K), -1, np.int32)
    lib().oracle_alpha_blending_forward(P, C, _p(uv, _f32p), _p(conic, _f32p), _p(opacity, _f32p),
                                        _p(feature, _f32p), _p(bias, _f32p), _p(idx_sorted, _i32p),
                                        _p(tile_range, _i32p), ctypes.c_float(bg), int(W), int(H), int(K),
                                        int(bool(enable_truncation)), _p(out, _f32p), _p(fT, _f32p),
                                        _p(nc, _i32p), _p(gi, _i32p))
    if K > 0:
        return out, fT, nc, gi
    return out, fT, nc


def alpha_blending_backward(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H,
                            final_T, ncontrib, dL_dout, opacity_bias=None):
    """Returns (dL_duv, dL_dconic, dL_dopacity, dL_dfeature, dL_dabs_uv[, dL_dbias])."""
    uv = _f(uv, (-1, 2)); conic = _f(conic, (-1, 3)); opacity = _f(opacity).reshape(-1)
    feature = _f(feature); feature = feature.reshape(uv.shape[0], -1)
    idx_sorted = _i(idx_sorted); tile_range = _i(tile_range)
    bias = None if opacity_bias is None else _f(opacity_bias).reshape(-1)
    fT = _f(final_T); nc = _i(ncontrib); g = _f(dL_dout)
    P, C = feature.shape
    duv = np.zeros((P, 2), np.float32); dabs = np.zeros((P, 2), np.float32)
    dcon = np.zeros((P, 3), np.float32); dop = np.zeros((P, 1), np.float32)
    df = np.zeros((P, C), np.float32); db = np.zeros((P, 1), np.float32)
    lib().oracle_alpha_blending_backward(P, C, _p(uv, _f32p), _p(conic, _f32p), _p(opacity, _f32p),
                                         _p(feature, _f32p), _p(bias, _f32p), _p(idx_sorted, _i32p),
                                         _p(tile_range, _i32p), ctypes.c_float(bg), int(W), int(H),
                                         _p(fT, _f32p), _p(nc, _i32p), _p(g, _f32p),
                                         _p(duv, _f32p), _p(dabs, _f32p), _p(dcon, _f32p), _p(dop, _f32p),
                                         _p(df, _f32p), _p(db, _f32p))
    if bias is not None:
        return duv, dcon, dop, df, dabs, db
    return duv, dcon, dop, df, dabs


# ------------------------------------------------------------------ composed pipelines
def render_forward(xyz, scale, rotate, opacity, feature, intr, extr, W, H, bg,
                   ortho=False, nearest=None, shs=None, sh_degree=3, K=0):
    """5-op chain of dptr.gs.rasterization (dptr/gs/__init__.py:28-100); with ``shs`` the colours
    come from compute_sh with direction (0,0,1) as the video renderer does
    (dptr_ortho_enhanced.py:270-272) and are prepended to ``feature``."""
    xyz = _f(xyz, (-1, 3))
    P = xyz.shape[0]
    saved = {}
    if shs is not None:
        dirs = np.zeros((P, 3), np.float32); dirs[:, 2] = 1.0
        rgb, clamped = compute_sh_forward(shs, sh_degree, dirs)
        saved.update(dirs=dirs, clamped=clamped, rgb=rgb)
        feature = rgb if feature is None else np.concatenate([rgb, _f(feature).reshape(P, -1)], axis=1)
    if ortho:
        uv, depth = project_point_ortho_forward(xyz, extr, W, H, 0.01 if nearest is None else nearest)
    else:
        uv, depth = project_point_forward(xyz, intr, extr, W, H, 0.2 if nearest is None else nearest)
    visible = depth.reshape(-1) != 0
    cov3d = compute_cov3d_forward(scale, rotate, visible)
    conic, radius, tiles = ewa_project_forward(xyz, cov3d, intr, extr, uv, W, H, visible, ortho=ortho)
    idx_sorted, tile_range = sort_gaussian(uv, depth, W, H, radius, tiles)
    res = alpha_blending_forward(uv, conic, opacity, feature, idx_sorted, tile_range, bg, W, H, K=K)
    saved.update(uv=uv, depth=depth, visible=visible, cov3d=cov3d, conic=conic, radius=radius, tiles=tiles,
                 idx_sorted=idx_sorted, tile_range=tile_range, feature=feature, final_T=res[1], ncontrib=res[2])
    return res, saved


def render_backward(xyz, scale, rotate, opacity, intr, extr, W, H, bg, saved, dL_dout,
                    ortho=False, shs=None, sh_degree=3):
    """Backward of :func:`render_forward` -> dict of gradients (autograd order of SURVEY 3.3)."""
    s = saved
    duv, dcon, dop, df, dabs = alpha_blending_backward(s["uv"], s["conic"], opacity, s["feature"],
                                                       s["idx_sorted"], s["tile_range"], bg, W, H,
                                                       s["final_T"], s["ncontrib"], dL_dout)
    dxyz_e, dcov, _, _ = ewa_project_backward(xyz, s["cov3d"], intr, extr, s["radius"], dcon, W, H, ortho=ortho,
                                              need_intr=False, need_extr=False)
    P = s["uv"].shape[0]
    zero_d = np.zeros((P, 1), np.float32)
    if ortho:
        dxyz_p = project_point_ortho_backward(extr, W, H, s["depth"], duv, zero_d)
    else:
        dxyz_p, _, _ = project_point_backward(xyz, intr, extr, W, H, s["uv"], s["depth"], duv, zero_d,
                                              need_intr=False, need_extr=False)
    dscale, dquat = compute_cov3d_backward(scale, rotate, s["visible"], dcov)
    out = dict(xyz=dxyz_e + dxyz_p, scale=dscale, rotate=dquat, opacity=dop, uv=duv, abs_uv=dabs,
               conic=dcon, feature=df)
    if shs is not None:
        dshs, _ = compute_sh_backward(shs, sh_degree, s["dirs"], None, s["clamped"], df[:, :3])
        out["shs"] = dshs
        out["feature"] = df[:, 3:]
    return out


# ------------------------------------------------------------------ dynamic Gaussians (row a15)
def dynamic_time_scalars(time, num_frames, intervals, start_frame_id, time_len):
    """Host-side scalars of get_position / get_rotation (reference:
    src/dynamic_gaussian_with_base_point_cloud.py:184-198,236-250) in float32 arithmetic:
    segment index, d = t - knot, polynomial basis t^k (k<4), Fourier basis cos/sin(t k pi) (k=1..4)."""
    intervals = np.asarray(intervals, np.float32)
    nt = float(time) / float(num_frames - 1)
    seg = int(np.searchsorted(intervals, np.float32(nt - 1e-7), side="left")) - 1
    seg = max(seg, 0)
    d = np.float32(np.float32(nt) - intervals[seg])
    rt = np.float32((float(time) - float(start_frame_id)) / float(time_len))
    poly = np.power(rt, np.arange(4, dtype=np.float32)).astype(np.float32)
    k = (np.arange(4, dtype=np.float32) + np.float32(1.0))
    arg = (rt * k * np.float32(np.pi)).astype(np.float32)
    fourier = np.concatenate([np.cos(arg), np.sin(arg)]).astype(np.float32)
    return seg, float(d), poly, fourier


def dynamic_eval_forward(position, cubic, rotation, rot_poly, rot_fourier, opacity, scaling, seg, d, poly, fourier):
    position = _f(position, (-1, 3)); P = position.shape[0]
    cubic = _f(cubic).reshape(P, 4, -1, 3); I = cubic.shape[2]
    rotation = _f(rotation, (-1, 4)); rot_poly = _f(rot_poly).reshape(P, 4, 4); rot_fourier = _f(rot_fourier).reshape(P, 8, 4)
    opacity = _f(opacity).reshape(-1); scaling = _f(scaling, (-1, 3)); poly = _f(poly); fourier = _f(fourier)
    pos = np.zeros((P, 3), np.float32); rot = np.zeros((P, 4), np.float32)
    opa = np.zeros((P, 1), np.float32); scl = np.zeros((P, 3), np.float32)
    lib().oracle_dynamic_eval_forward(P, I, int(seg), ctypes.c_float(d), _p(position, _f32p), _p(cubic, _f32p),
                                      _p(rotation, _f32p), _p(rot_poly, _f32p), _p(rot_fourier, _f32p), _p(poly, _f32p),
                                      _p(fourier, _f32p), _p(opacity, _f32p), _p(scaling, _f32p), _p(pos, _f32p),
                                      _p(rot, _f32p), _p(opa, _f32p), _p(scl, _f32p))
    return pos, rot, opa, scl


def dynamic_eval_backward(cubic_shape, rotation, rot_poly, rot_fourier, opacity, scaling, seg, d, poly, fourier,
                          g_pos, g_rot, g_opa, g_scl):
    P, _, I, _ = cubic_shape
    rotation = _f(rotation, (-1, 4)); rot_poly = _f(rot_poly).reshape(P, 4, 4); rot_fourier = _f(rot_fourier).reshape(P, 8, 4)
    opacity = _f(opacity).reshape(-1); scaling = _f(scaling, (-1, 3)); poly = _f(poly); fourier = _f(fourier)
    g_pos = _f(g_pos, (-1, 3)); g_rot = _f(g_rot, (-1, 4)); g_opa = _f(g_opa).reshape(-1); g_scl = _f(g_scl, (-1, 3))
    dpos = np.zeros((P, 3), np.float32); dcub = np.zeros((P, 4, I, 3), np.float32); drot = np.zeros((P, 4), np.float32)
    dopa = np.zeros((P, 1), np.float32); dscl = np.zeros((P, 3), np.float32)
    lib().oracle_dynamic_eval_backward(P, I, int(seg), ctypes.c_float(d), _p(rotation, _f32p), _p(rot_poly, _f32p),
                                       _p(rot_fourier, _f32p), _p(poly, _f32p), _p(fourier, _f32p), _p(opacity, _f32p),
                                       _p(scaling, _f32p), _p(g_pos, _f32p), _p(g_rot, _f32p), _p(g_opa, _f32p),
                                       _p(g_scl, _f32p), _p(dpos, _f32p), _p(dcub, _f32p), _p(drot, _f32p),
                                       _p(dopa, _f32p), _p(dscl, _f32p))
    return dpos, dcub, drot, dopa, dscl


# ------------------------------------------------------------------ densification statistics
_u8p = ctypes.POINTER(ctypes.c_ubyte)
_i32p = ctypes.POINTER(ctypes.c_int)


def densify_accumulate(radius, tap, sx, sy, viewspace_grad, visible, radii):
    """in place on viewspace_grad [P,2] f32, visible [P] u8, radii [P] i32"""
    radius = np.ascontiguousarray(radius, np.int32); tap = _f(tap, (-1, 2))
    assert viewspace_grad.dtype == np.float32 and visible.dtype == np.uint8 and radii.dtype == np.int32
    lib().oracle_densify_accumulate(radius.size, _p(radius, _i32p), _p(tap, _f32p), ctypes.c_float(sx), ctypes.c_float(sy),
                                    _p(viewspace_grad, _f32p), _p(visible, _u8p), _p(radii, _i32p))


def densify_update(visible, viewspace_grad, radii, max_radii2D, pos_gradient_accum, denom):
    """in place on max_radii2D, pos_gradient_accum, denom (float32 [P])"""
    visible = np.ascontiguousarray(visible, np.uint8); viewspace_grad = _f(viewspace_grad, (-1, 2))
    radii = np.ascontiguousarray(radii, np.int32)
    for a in (max_radii2D, pos_gradient_accum, denom):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().oracle_densify_update(visible.size, _p(visible, _u8p), _p(viewspace_grad, _f32p), _p(radii, _i32p),
                                _p(max_radii2D, _f32p), _p(pos_gradient_accum, _f32p), _p(denom, _f32p))


def densify_masks(pos_gradient_accum, denom, max_radii2D, scaling_raw, opacity_raw, grad_threshold, percent_dense,
                  cameras_extent, min_opacity, size_threshold=20.0):
    acc = _f(pos_gradient_accum).reshape(-1); den = _f(denom).reshape(-1); mr = _f(max_radii2D).reshape(-1)
    sc = _f(scaling_raw, (-1, 3)); op = _f(opacity_raw).reshape(-1)
    P = acc.size
    clone = np.zeros(P, np.uint8); split = np.zeros(P, np.uint8); prune = np.zeros(P, np.uint8)
    lib().oracle_densify_masks(P, _p(acc, _f32p), _p(den, _f32p), _p(mr, _f32p), _p(sc, _f32p), _p(op, _f32p),
                               ctypes.c_float(grad_threshold), ctypes.c_float(percent_dense), ctypes.c_float(cameras_extent),
                               ctypes.c_float(min_opacity), ctypes.c_float(size_threshold), _p(clone, _u8p),
                               _p(split, _u8p), _p(prune, _u8p))
    return clone.astype(bool), split.astype(bool), prune.astype(bool)


def compact_rows(mask, src):
    """src[mask] for a 32-bit array of shape [P, ...]"""
    mask = np.ascontiguousarray(mask, np.uint8)
    src = np.ascontiguousarray(src)
    assert src.dtype.itemsize == 4 and src.shape[0] == mask.size
    rw = int(np.prod(src.shape[1:])) if src.ndim > 1 else 1
    dst = np.empty_like(src)
    f = lib().oracle_compact_rows
    f.restype = ctypes.c_int
    n = f(mask.size, _p(mask, _u8p), rw, ctypes.c_void_p(src.ctypes.data), ctypes.c_void_p(dst.ctypes.data))
    return dst[:n]


# ------------------------------------------------------------------ K nearest neighbours
def knn_points(query, points, K):
    """(dists [N,K] squared L2 ascending, idx [N,K] int32) by brute force"""
    q = _f(query, (-1, 3)); p = _f(points, (-1, 3))
    d = np.zeros((q.shape[0], K), np.float32); i = np.zeros((q.shape[0], K), np.int32)
    lib().oracle_knn_points(q.shape[0], _p(q, _f32p), p.shape[0], _p(p, _f32p), int(K), _p(d, _f32p), _p(i, _i32p))
    return d, i


# ------------------------------------------------------------------ structure surgery of densification (numpy restatement)
def _quat_R_np(q):
    q = q / np.sqrt((q * q).sum(1, keepdims=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3), np.float32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def structure_clone(params, moments, mask):
    """densify_clone (reference: src/pointrix/optimizer/atlas_gs_optimizer.py:289-304; extend_optimizer
    src/pointrix/point_cloud/points.py:332-375): selected rows appended, their Adam moments zero"""
    m = np.asarray(mask, bool)
    p = {k: np.concatenate([v, v[m]]) for k, v in params.items()}
    mo = {k: tuple(np.concatenate([x, np.zeros_like(x[m])]) for x in mv) for k, mv in moments.items()}
    return p, mo


def structure_split(params, moments, mask, split_num, unit_normals):
    """densify_split + new_pos_scale (atlas_gs_optimizer.py:255-287,306-349; prune_optimizer points.py:279-330) with the
    normal draws given: children appended (sampled position, log(scale / (0.8 split_num)), other attributes repeated, zero
    moments), parents removed"""
    m = np.asarray(mask, bool)
    n = int(m.sum())
    scl = np.exp(params["scaling"][m]).astype(np.float32)
    stds = np.tile(scl, (split_num, 1))
    samples = (np.asarray(unit_normals, np.float32) * stds).astype(np.float32)
    R = np.tile(_quat_R_np(params["rotation"][m].astype(np.float32)), (split_num, 1, 1))
    new_pos = np.einsum("nij,nj->ni", R, samples).astype(np.float32) + np.tile(params["position"][m], (split_num, 1))
    new_scl = np.log(stds / np.float32(0.8 * split_num)).astype(np.float32)
    valid = np.concatenate([~m, np.ones(split_num * n, bool)])
    p, mo = {}, {}
    for k, v in params.items():
        tail = new_pos if k == "position" else new_scl if k == "scaling" else np.tile(v[m], (split_num,) + (1,) * (v.ndim - 1))
        p[k] = np.concatenate([v, tail.astype(v.dtype)])[valid]
    for k, mv in moments.items():
        mo[k] = tuple(np.concatenate([x, np.zeros((split_num * n,) + x.shape[1:], x.dtype)])[valid] for x in mv)
    return p, mo, new_pos, new_scl


# ------------------------------------------------------------------ polynomial / Fourier position model (numpy restatement)
def time_basis(time, start_frame_id, time_len):
    """the 12 basis values of get_position / get_rotation (reference: src/dynamic_gaussian_points.py:138-150,169-180) in
    float32: t'^0..3, cos(t' l pi), sin(t' l pi), l = 1..4, t' = (time - start_frame_id) / time_len"""
    rt = np.float32((float(time) - float(start_frame_id)) / float(time_len))
    k = np.arange(4, dtype=np.float32)
    arg = (rt * (k + np.float32(1.0)) * np.float32(np.pi)).astype(np.float32)
    return np.concatenate([np.power(rt, k), np.cos(arg), np.sin(arg)]).astype(np.float32)


def position_poly_fourier_forward(position, pos_poly_feat, pos_fourier_feat, basis):
    """get_position (src/dynamic_gaussian_points.py:169-186)"""
    b = np.asarray(basis, np.float32)
    return (_f(position, (-1, 3)) + (np.asarray(pos_poly_feat, np.float32) * b[None, :4, None]).sum(1)
            + (np.asarray(pos_fourier_feat, np.float32) * b[None, 4:, None]).sum(1)).astype(np.float32)


def position_poly_fourier_backward(g_pos, basis):
    """-> (d_position, d_pos_poly_feat [N,4,3], d_pos_fourier_feat [N,8,3])"""
    g = _f(g_pos, (-1, 3)); b = np.asarray(basis, np.float32)
    return g.copy(), g[:, None, :] * b[None, :4, None], g[:, None, :] * b[None, 4:, None]


# ------------------------------------------------------------------ ARAP energy (numpy restatement)
def arap_energy(nodes, nbr, weight, sample_idx):
    """cal_arap_error + estimate_rotation (reference: src/geometry_utils.py:50-123) in numpy float32 / LAPACK SVD:
    nodes [Nt,Nv,3], nbr [Nv,K] (-1: none), weight [Nv,K] or None, sample_idx [S] -> (energy / Nt, d energy / d nodes with
    the rotations held constant, rotations [Nt-1,S,3,3])"""
    nodes = np.asarray(nodes, np.float32)
    Nt, Nv, _ = nodes.shape
    nbr = np.asarray(nbr)
    K = nbr.shape[1]
    has = nbr >= 0
    w_all = has.astype(np.float32) if weight is None else np.asarray(weight, np.float32)
    sidx = np.asarray(sample_idx, np.int64)

    def edges(v):
        E = np.zeros((Nv, K, 3), np.float32)
        ii, kk = np.nonzero(has)
        E[ii, kk] = v[ii] - v[nbr[ii, kk]]
        return E

    Es = edges(nodes[0])[sidx]
    w = w_all[sidx]
    total = 0.0
    grad = np.zeros(nodes.shape, np.float64)
    rots = np.zeros((Nt - 1, sidx.size, 3, 3), np.float32)
    for t in range(1, Nt):
        Et = edges(nodes[t])[sidx]
        S = np.einsum("ski,sk,skj->sij", Es, w, Et)
        unchanged = (Es == Et).all(axis=1).any(axis=1)      # (:66: reduced over the K edges, any axis -- as the reference)
        S[unchanged] = 0
        U, sig, Wt = np.linalg.svd(S.astype(np.float64))
        W = np.transpose(Wt, (0, 2, 1))
        R = W @ np.transpose(U, (0, 2, 1))
        flip = np.linalg.det(R) <= 0
        if flip.any():
            Um = U.copy()
            cols = np.argmin(sig[flip], axis=1)
            idxs = np.nonzero(flip)[0]
            Um[idxs, :, cols] *= -1
            R[flip] = W[flip] @ np.transpose(Um[flip], (0, 2, 1))
        rots[t - 1] = R.astype(np.float32)
        st = Et.astype(np.float64) - np.einsum("sij,skj->ski", R, Es.astype(np.float64))
        total += float((w * (st ** 2).sum(-1)).sum())
        gt = 2.0 * w[..., None] * st                                    # d / d e_tgt
        gs = -2.0 * w[..., None] * np.einsum("sji,skj->ski", R, st)      # d / d e_src = -2 w R^T st
        for s, i in enumerate(sidx):
            for k in range(K):
                j = nbr[i, k]
                if j < 0:
                    continue
                grad[t, i] += gt[s, k]; grad[t, j] -= gt[s, k]
                grad[0, i] += gs[s, k]; grad[0, j] -= gs[s, k]
    return np.float32(total / Nt), (grad / Nt).astype(np.float32), rots
