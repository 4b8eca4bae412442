// Per-frame evaluation of the dynamic Gaussians: canonical parameters + time -> the
// per-frame position / rotation / opacity / scaling the rasterizer chain consumes, and the matching backward.
// Replaces the eager torch of the reference's point-cloud class
// (src/dynamic_gaussian_with_base_point_cloud.py:171-198 get_opacity/get_scaling/get_rotation, :236-250 get_position).
//
// One quad (4 adjacent lanes) per Gaussian: lane j<3 owns position/scaling component j, lane 3 owns the opacity,
// lane a owns quaternion component a.  The [N,4,4] / [N,8,4] rotation tables are read as one float4 row per lane
// (a wave reads 1 KiB contiguous per instruction) and transposed-reduced inside the quad with DPP quad_perm;
// every output is written component-per-lane, i.e. contiguous across the wave.
#include "common.h"
#include "dynamics_dev.h"

namespace {

__global__ __launch_bounds__(DYN_BLOCK) void dynamic_eval_fwd_kernel(
    int P, CubicAddr ca, float d, DynBasis b, const float *__restrict__ position, const float *__restrict__ cubic,
    const float *__restrict__ rotation, const float4 *__restrict__ rot_poly, const float4 *__restrict__ rot_fourier,
    const float *__restrict__ opacity, const float *__restrict__ scaling, float *__restrict__ pos_t,
    float *__restrict__ rot_t, float *__restrict__ opa_t, float *__restrict__ scl_t) {
    const int t = blockIdx.x * DYN_BLOCK + threadIdx.x;
    const int n = t >> 2, j = t & 3;
    if (n >= P) return;  // whole quads leave together
    if (pos_t && j < 3) {
        // segment polynomial c3 + c2 d + c1 d^2 + c0 d^3 of the [N,4,I,3] table (:241-248)
        const float *c = cubic + ca.seg_off + (size_t)n * ca.stride_n + j;
        const size_t row = ca.stride_k;
        const float c0 = c[0], c1 = c[row], c2 = c[2 * row], c3 = c[3 * row];
        float p = c3 + c2 * d;
        p = p + c1 * (d * d);
        p = p + c0 * (d * d * d);
        pos_t[(size_t)n * 3 + j] = p + position[(size_t)n * 3 + j];
    }
    if (rot_t) {
        const float q = quat_component(n, j, b, rotation, rot_poly, rot_fourier);
        const float nrm = fmaxf(sqrtf(quad_sum(q * q)), 1e-12f);  // F.normalize: x / max(|x|, eps)
        rot_t[(size_t)n * 4 + j] = q / nrm;
    }
    if (scl_t && j < 3) scl_t[(size_t)n * 3 + j] = expf(scaling[(size_t)n * 3 + j]);
    if (opa_t && j == 3) opa_t[n] = 1.0f / (1.0f + expf(-opacity[n]));
}

template <bool ACC>
__device__ __forceinline__ void put(float *p, float v) {
    if (ACC)
        *p += v;
    else
        *p = v;
}

template <bool ACC>
__global__ __launch_bounds__(DYN_BLOCK) void dynamic_eval_bwd_kernel(
    int P, CubicAddr ca, float d, DynBasis b, const float *__restrict__ rotation,
    const float4 *__restrict__ rot_poly, const float4 *__restrict__ rot_fourier, const float *__restrict__ opacity,
    const float *__restrict__ scaling, const float *__restrict__ g_pos, const float *__restrict__ g_rot,
    const float *__restrict__ g_opa, const float *__restrict__ g_scl, float *__restrict__ d_position,
    float *__restrict__ d_cubic, float *__restrict__ d_rotation, float *__restrict__ d_opacity,
    float *__restrict__ d_scaling) {
    const int t = blockIdx.x * DYN_BLOCK + threadIdx.x;
    const int n = t >> 2, j = t & 3;
    if (n >= P) return;
    if (g_pos && j < 3) {
        const float g = g_pos[(size_t)n * 3 + j];
        if (d_position) put<ACC>(d_position + (size_t)n * 3 + j, g);
        if (d_cubic) {  // only the active segment of the spline table is touched
            float *c = d_cubic + ca.seg_off + (size_t)n * ca.stride_n + j;
            const size_t row = ca.stride_k;
            put<ACC>(c, g * (d * d * d));
            put<ACC>(c + row, g * (d * d));
            put<ACC>(c + 2 * row, g * d);
            put<ACC>(c + 3 * row, g);
        }
    }
    if (g_rot && d_rotation) {
        const float q = quat_component(n, j, b, rotation, rot_poly, rot_fourier);
        const float nrm = sqrtf(quad_sum(q * q));
        const float g = g_rot[(size_t)n * 4 + j];
        float dq;
        if (nrm < 1e-12f) {
            dq = g / 1e-12f;  // clamped norm: plain scaling
        } else {
            const float qh = q / nrm;
            const float dot = quad_sum(qh * g);
            dq = (g - qh * dot) / nrm;
        }
        put<ACC>(d_rotation + (size_t)n * 4 + j, dq);
    }
    if (g_scl && d_scaling && j < 3)
        put<ACC>(d_scaling + (size_t)n * 3 + j, g_scl[(size_t)n * 3 + j] * expf(scaling[(size_t)n * 3 + j]));
    if (g_opa && d_opacity && j == 3) {
        const float s = 1.0f / (1.0f + expf(-opacity[n]));
        put<ACC>(d_opacity + n, g_opa[n] * s * (1.0f - s));
    }
}

inline dim3 dyn_grid(int P) { return dim3((unsigned)(((size_t)P * 4 + DYN_BLOCK - 1) / DYN_BLOCK)); }

// ---- polynomial + Fourier position model of the reference's first dynamic point cloud
// (src/dynamic_gaussian_points.py:169-186 get_position): position(t) = position + sum_k pos_poly_feat[:, k, :] t'^k
//   + sum_m pos_fourier_feat[:, m, :] basis_m(t'),  t' = (time - start_frame_id) / time_len, basis = cos(t' l pi), sin(t' l pi)
// (l = 1..4).  Unlike the rotation tables these are NOT detached: both tables receive gradients.  One lane per
// (Gaussian, axis): the [N,4,3] / [N,8,3] tables are read with stride-3 accesses that together cover whole lines.
__global__ __launch_bounds__(DYN_BLOCK) void position_pf_fwd_kernel(int P, DynBasis b, const float *__restrict__ position,
                                                                    const float *__restrict__ pos_poly,
                                                                    const float *__restrict__ pos_fourier,
                                                                    float *__restrict__ pos_t) {
    const size_t t = (size_t)blockIdx.x * DYN_BLOCK + threadIdx.x;
    if (t >= (size_t)P * 3) return;
    const size_t n = t / 3, j = t - 3 * n;
    float acc = position[t];
    float ps = 0.f, fs = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) ps += pos_poly[n * 12 + k * 3 + j] * b.poly[k];
#pragma unroll
    for (int m = 0; m < 8; ++m) fs += pos_fourier[n * 24 + m * 3 + j] * b.fourier[m];
    pos_t[t] = (acc + ps) + fs;
}

template <bool ACC>
__global__ __launch_bounds__(DYN_BLOCK) void position_pf_bwd_kernel(int P, DynBasis b, const float *__restrict__ g_pos,
                                                                    float *__restrict__ d_position,
                                                                    float *__restrict__ d_pos_poly,
                                                                    float *__restrict__ d_pos_fourier) {
    const size_t t = (size_t)blockIdx.x * DYN_BLOCK + threadIdx.x;
    if (t >= (size_t)P * 3) return;
    const size_t n = t / 3, j = t - 3 * n;
    const float g = g_pos[t];
    if (d_position) put<ACC>(d_position + t, g);
    if (d_pos_poly) {
#pragma unroll
        for (int k = 0; k < 4; ++k) put<ACC>(d_pos_poly + n * 12 + k * 3 + j, g * b.poly[k]);
    }
    if (d_pos_fourier) {
#pragma unroll
        for (int m = 0; m < 8; ++m) put<ACC>(d_pos_fourier + n * 24 + m * 3 + j, g * b.fourier[m]);
    }
}


// ---- position(t) of every Gaussian at ALL frame times of a table (the pair frames of a training batch: track_gs =
// position(ids2), src/trainer_fragGS.py:486-507; the node sequence of the ARAP term, :671-675).  One thread per (Gaussian,
// axis) walks the frames: base position read once, the frame's 48-byte spline segment per frame, the arithmetic of
// dynamic_eval_fwd_kernel.  out[f * out_fs + 3 n + j]; grid.y slices the frames.
__global__ __launch_bounds__(DYN_BLOCK) void dynamic_positions_fwd_kernel(int F, int P, int I, int layout,
                                                                         const DynTab *__restrict__ tab,
                                                                         const float *__restrict__ position,
                                                                         const float *__restrict__ cubic, float *__restrict__ out,
                                                                         long long out_fs) {
    const size_t t = (size_t)blockIdx.x * DYN_BLOCK + threadIdx.x;
    if (t >= (size_t)P * 3) return;
    const size_t n = t / 3, j = t - 3 * n;
    const float base = position[t];
    const int per = (F + gridDim.y - 1) / gridDim.y;
    const int f0 = blockIdx.y * per, f1 = imin_(F, f0 + per);
    for (int f = f0; f < f1; ++f) {
        const CubicAddr ca = cubic_addr(layout, P, I, tab[f].seg);
        const float d = tab[f].d;
        const float *c = cubic + ca.seg_off + n * ca.stride_n + j;
        const size_t row = ca.stride_k;
        const float c0 = c[0], c1 = c[row], c2 = c[2 * row], c3 = c[3 * row];
        float p = c3 + c2 * d;
        p = p + c1 * (d * d);
        p = p + c0 * (d * d * d);
        out[(size_t)f * out_fs + t] = p + base;
    }
}

// Backward: g[f * g_fs + 3 n + j] = dL/dposition(t_f) -> d_position += sum_f g, the four coefficient rows of frame f's segment
// += g * (d^3, d^2, d, 1).  One thread per (Gaussian, axis) walks ALL frames (no atomics: it owns its rows); frames of one
// segment that follow each other accumulate in registers and are flushed when the walk leaves the segment.
__global__ __launch_bounds__(DYN_BLOCK) void dynamic_positions_bwd_kernel(int F, int P, int I, int layout,
                                                                         const DynTab *__restrict__ tab,
                                                                         const float *__restrict__ g, long long g_fs,
                                                                         float *__restrict__ d_position,
                                                                         float *__restrict__ d_cubic) {
    const size_t t = (size_t)blockIdx.x * DYN_BLOCK + threadIdx.x;
    if (t >= (size_t)P * 3) return;
    const size_t n = t / 3, j = t - 3 * n;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, dpos = 0.f;
    int cur = -1;
    auto flush = [&](int seg) {
        if (seg < 0 || !d_cubic) return;
        const CubicAddr ca = cubic_addr(layout, P, I, seg);
        float *c = d_cubic + ca.seg_off + n * ca.stride_n + j;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k * ca.stride_k] += acc[k];
    };
    for (int i = 0; i < F; ++i) {
        // walk order: entry i's pad[0] = 1 + the frame to visit i-th (frames of one segment next to each other: one flush per
        // touched segment instead of one per change of segment in table order -- the pair frames of a training batch arrive
        // interleaved (ids1, ids2, ids1, ..)); 0: table order.  Uniform: scalar loads.
        const int o = __float_as_int(tab[i].pad[0]);
        const int f = (o >= 1 && o <= F) ? o - 1 : i;
        const int seg = tab[f].seg;
        if (seg != cur) {
            flush(cur);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = 0.f;
            cur = seg;
        }
        const float d = tab[f].d, v = g[(size_t)f * g_fs + t];
        dpos += v;
        acc[0] += v * (d * d * d); acc[1] += v * (d * d); acc[2] += v * d; acc[3] += v;
    }
    flush(cur);
    if (d_position) d_position[t] += dpos;
}

}  // namespace

extern "C" int splat_position_poly_fourier_forward(int P, const float *basis_host, const float *position,
                                                   const float *pos_poly_feat, const float *pos_fourier_feat, float *pos_t,
                                                   void *stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad size");
    SPLAT_CHECK_ARG(basis_host != nullptr, "basis_host (12 host floats) is required");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(position && pos_poly_feat && pos_fourier_feat && pos_t, "null pointer");
    const unsigned blocks = (unsigned)(((size_t)P * 3 + DYN_BLOCK - 1) / DYN_BLOCK);
    SPLAT_LAUNCH("position_pf_fwd", position_pf_fwd_kernel, dim3(blocks), dim3(DYN_BLOCK), 0, (hipStream_t)stream, P,
                 load_basis(basis_host), position, pos_poly_feat, pos_fourier_feat, pos_t);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_position_poly_fourier_backward(int P, const float *basis_host, const float *g_pos, int accumulate,
                                                    float *d_position, float *d_pos_poly_feat, float *d_pos_fourier_feat,
                                                    void *stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad size");
    SPLAT_CHECK_ARG(basis_host != nullptr, "basis_host (12 host floats) is required");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(g_pos != nullptr, "null pointer");
    const unsigned blocks = (unsigned)(((size_t)P * 3 + DYN_BLOCK - 1) / DYN_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    if (accumulate)
        SPLAT_LAUNCH("position_pf_bwd", position_pf_bwd_kernel<true>, dim3(blocks), dim3(DYN_BLOCK), 0, s, P,
                     load_basis(basis_host), g_pos, d_position, d_pos_poly_feat, d_pos_fourier_feat);
    else
        SPLAT_LAUNCH("position_pf_bwd", position_pf_bwd_kernel<false>, dim3(blocks), dim3(DYN_BLOCK), 0, s, P,
                     load_basis(basis_host), g_pos, d_position, d_pos_poly_feat, d_pos_fourier_feat);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_dynamic_eval_forward(int P, int I, int seg, float d, const float *basis_host,
                                          const float *position, const float *cubic, int cubic_layout, const float *rotation,
                                          const float *rot_poly, const float *rot_fourier, const float *opacity,
                                          const float *scaling, float *pos_t, float *rot_t, float *opa_t, float *scl_t,
                                          void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && I >= 1, "P >= 0 and I >= 1 required");
    SPLAT_CHECK_ARG(seg >= 0 && seg < I, "segment index out of range");
    SPLAT_CHECK_ARG(basis_host != nullptr, "basis_host (12 host floats) is required");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR,
                    "unknown cubic_layout");
    SPLAT_CHECK_ARG(!pos_t || (position && cubic), "pos_t needs position and cubic");
    SPLAT_CHECK_ARG(!rot_t || (rotation && rot_poly && rot_fourier), "rot_t needs rotation, rot_poly, rot_fourier");
    SPLAT_CHECK_ARG(!opa_t || opacity, "opa_t needs opacity");
    SPLAT_CHECK_ARG(!scl_t || scaling, "scl_t needs scaling");
    if (P == 0) return SPLAT_OK;
    SPLAT_LAUNCH("dynamic_eval_fwd", dynamic_eval_fwd_kernel, dyn_grid(P), dim3(DYN_BLOCK), 0, (hipStream_t)stream, P,
                 cubic_addr(cubic_layout, P, I, seg), d, load_basis(basis_host), position, cubic, rotation, (const float4 *)rot_poly,
                 (const float4 *)rot_fourier, opacity, scaling, pos_t, rot_t, opa_t, scl_t);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_dynamic_eval_backward(int P, int I, int seg, float d, const float *basis_host,
                                           const float *rotation, const float *rot_poly, const float *rot_fourier,
                                           const float *opacity, const float *scaling, const float *g_pos,
                                           const float *g_rot, const float *g_opa, const float *g_scl, int accumulate,
                                           int cubic_layout, float *d_position, float *d_cubic, float *d_rotation, float *d_opacity,
                                           float *d_scaling, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && I >= 1, "P >= 0 and I >= 1 required");
    SPLAT_CHECK_ARG(seg >= 0 && seg < I, "segment index out of range");
    SPLAT_CHECK_ARG(basis_host != nullptr, "basis_host (12 host floats) is required");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR,
                    "unknown cubic_layout");
    SPLAT_CHECK_ARG(!(g_rot && d_rotation) || (rotation && rot_poly && rot_fourier),
                    "rotation gradient needs rotation, rot_poly, rot_fourier");
    SPLAT_CHECK_ARG(!(g_opa && d_opacity) || opacity, "opacity gradient needs opacity");
    SPLAT_CHECK_ARG(!(g_scl && d_scaling) || scaling, "scaling gradient needs scaling");
    if (P == 0) return SPLAT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (accumulate)
        SPLAT_LAUNCH("dynamic_eval_bwd", dynamic_eval_bwd_kernel<true>, dyn_grid(P), dim3(DYN_BLOCK), 0, s, P,
                     cubic_addr(cubic_layout, P, I, seg), d, load_basis(basis_host), rotation, (const float4 *)rot_poly, (const float4 *)rot_fourier, opacity,
                     scaling, g_pos, g_rot, g_opa, g_scl, d_position, d_cubic, d_rotation, d_opacity, d_scaling);
    else
        SPLAT_LAUNCH("dynamic_eval_bwd", dynamic_eval_bwd_kernel<false>, dyn_grid(P), dim3(DYN_BLOCK), 0, s, P,
                     cubic_addr(cubic_layout, P, I, seg), d, load_basis(basis_host), rotation, (const float4 *)rot_poly, (const float4 *)rot_fourier, opacity,
                     scaling, g_pos, g_rot, g_opa, g_scl, d_position, d_cubic, d_rotation, d_opacity, d_scaling);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// position(t_f) of every Gaussian for the F frame times of `tab` (device table of F 64-byte entries {int seg; float d; float
// basis[12]; pad}: the table of splat_frame_preprocess_forward_batch): out[f * out_frame_stride + 3 n + j].  The pair frames of
// a training batch (track_gs = position(ids2), src/trainer_fragGS.py:486-507) and the node sequence of its ARAP term.
extern "C" int splat_dynamic_positions_batch_forward(int F, int P, int I, const void *tab, const float *position,
                                                     const float *cubic, int cubic_layout, float *out,
                                                     int64_t out_frame_stride, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 0 && I >= 1, "F >= 1, P >= 0 and I >= 1 required");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR, "unknown cubic_layout");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(tab && position && cubic && out, "null pointer");
    SPLAT_CHECK_ARG(out_frame_stride >= (int64_t)P * 3, "out_frame_stride below P * 3");
    const unsigned blocks = (unsigned)(((size_t)P * 3 + DYN_BLOCK - 1) / DYN_BLOCK);
    unsigned slices = 1;   // enough workgroups to fill the chip: slice the frames when the Gaussians alone do not
    while (blocks * slices < 2048u && slices < (unsigned)F) slices *= 2;
    if (slices > (unsigned)F) slices = (unsigned)F;
    SPLAT_LAUNCH("dynamic_positions_fwd", dynamic_positions_fwd_kernel, dim3(blocks, slices), dim3(DYN_BLOCK), 0, (hipStream_t)stream,
                 F, P, I, cubic_layout, (const DynTab *)tab, position, cubic, out, (long long)out_frame_stride);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// Backward of the above: g[f * g_frame_stride + 3 n + j] is ADDED into d_position [P,3] (optional) and into the coefficient
// rows of every frame's segment of d_cubic (layout `cubic_layout`, optional).  Deterministic (no atomics).
extern "C" int splat_dynamic_positions_batch_backward(int F, int P, int I, const void *tab, const float *g,
                                                      int64_t g_frame_stride, int cubic_layout, float *d_position,
                                                      float *d_cubic, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 0 && I >= 1, "F >= 1, P >= 0 and I >= 1 required");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR, "unknown cubic_layout");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(tab && g, "null pointer");
    SPLAT_CHECK_ARG(g_frame_stride >= (int64_t)P * 3, "g_frame_stride below P * 3");
    if (!d_position && !d_cubic) return SPLAT_OK;
    const unsigned blocks = (unsigned)(((size_t)P * 3 + DYN_BLOCK - 1) / DYN_BLOCK);
    SPLAT_LAUNCH("dynamic_positions_bwd", dynamic_positions_bwd_kernel, dim3(blocks), dim3(DYN_BLOCK), 0, (hipStream_t)stream, F, P, I,
                 cubic_layout, (const DynTab *)tab, g, (long long)g_frame_stride, d_position, d_cubic);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
