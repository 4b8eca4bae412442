// Fused per-frame preprocess of the orthographic renderer: one pass over the Gaussians does what the reference's
// renderer spreads over ~80 eager kernels and three native operators per frame
// (src/pointrix/renderer/dptr_ortho_enhanced.py:282-310: project_point (ortho) -> compute_cov3d -> ewa_project (ortho)),
// forward and backward.  Intermediates (visible mask, cov3d, dL_dcov3d) never reach HBM; the backward can add its
// results straight into the caller's gradient buffers.  The arithmetic is the one-Gaussian code of pointwise_dev.h,
// i.e. the same functions the separate operators run.
#include "common.h"
#include "pointwise_dev.h"
#include "dynamics_dev.h"

namespace {

constexpr int PP_BLOCK = 256;
inline dim3 pp_grid(int P) { return dim3((unsigned)((P + PP_BLOCK - 1) / PP_BLOCK)); }

// ORTHO: the video renderer's camera (dptr_ortho_enhanced.py:177-202, 18-111); else the pinhole camera of gs.rasterization /
// DPTRRender (project_point.cu, ewa_project.cu: reference src/submodules/dptr/dptr/gs/__init__.py:28-100, dptr.py:107-161).
// The camera may differ from frame to frame (render_batch gives every batch element its own: dptr_ortho_enhanced.py:409-411):
// frame f reads extr + f * extr_fs (and intr + f * intr_fs).
template <bool ORTHO>
__global__ void __launch_bounds__(PP_BLOCK)
preprocess_fwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ offset,
                      const float *__restrict__ scales, const float4 *__restrict__ uquats,
                      const float *__restrict__ intr, const float *__restrict__ extr, long long intr_fs, long long extr_fs,
                      int W, int H, float nearest, float extent,
                      float2 *__restrict__ uv, float *__restrict__ depth, float *__restrict__ conic,
                      int *__restrict__ radius, int *__restrict__ tiles) {
    const int i = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (i >= P) return;
    {   // frame batch: blockIdx.y = frame; offsets [F,P,3] in, [F,P,..] out
        const size_t f = blockIdx.y;
        if (offset) offset += f * 3 * P;
        uv += f * P; depth += f * P; conic += f * 3 * P; radius += f * P;
        if (tiles) tiles += f * P;
        extr += f * extr_fs;
        if (!ORTHO) intr += f * intr_fs;
    }
    Cam c;
    load_cam(ORTHO ? nullptr : intr, extr, c);
    float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (offset) {
        p[0] += offset[3 * i]; p[1] += offset[3 * i + 1]; p[2] += offset[3 * i + 2];
    }
    float u, v, d;
    const bool cull = ORTHO ? project_ortho_pt(c, p[0], p[1], p[2], W, H, nearest, extent, u, v, d)
                            : project_persp_pt(c, p[0], p[1], p[2], W, H, nearest, extent, u, v, d);
    u = cull ? 0.f : u; v = cull ? 0.f : v; d = cull ? 0.f : d;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    int orad = 0, otiles = 0;
    if (d != 0.f) {  // the renderers' visibility mask (dptr_ortho_enhanced.py:288, gs/__init__.py:60)
        const float4 q4 = uquats[i];
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        float c3[6], a[3], b[3], t[3], Jm[4], cov[3];
        cov3d_pt(s, q, c3);
        ewa_T<ORTHO>(c, p, W, H, a, b, t, Jm);
        ewa_cov2d<ORTHO>(a, b, c3, cov);
        ewa_finish_pt<ORTHO>(cov, make_float2(u, v), W, H, o0, o1, o2, orad, otiles);
    }
    uv[i] = make_float2(u, v);
    depth[i] = d;
    conic[3 * i] = o0; conic[3 * i + 1] = o1; conic[3 * i + 2] = o2;
    radius[i] = orad;
    if (tiles) tiles[i] = otiles;
}

// Frame batch of STATIC Gaussians under ONE orthographic camera (the default workload of bench.py, FrameBatch.render / render_sets):
// the 3D covariance and -- the orthographic Jacobian being constant -- the 2D covariance do not depend on the frame, so a
// thread reads its Gaussian's parameters and runs cov3d + EWA ONCE and then walks its slice of the frames: per frame only the
// offset in (12 B) and uv / depth / conic / radius out (28 B), where the frame-as-grid.y kernel above re-read the 40 bytes of
// parameters and redid the covariance per frame (PMC round 4: 579 MB moved for 342 algorithmic per 25-frame launch).  Same
// The single-frame operator (splat_preprocess_ortho_forward) runs THIS kernel with F = 1, so that a frame of a batch and the
// per-frame operator give the same bits (the compiler contracts the covariance arithmetic differently in different kernels:
// the frame-as-grid.y kernel's conic differed in the last bit on 6 % of the Gaussians).
__global__ void __launch_bounds__(PP_BLOCK)
preprocess_fwd_frames_kernel(int F, int P, const float *__restrict__ xyz, const float *__restrict__ offset,
                             const float *__restrict__ scales, const float4 *__restrict__ uquats, const float *__restrict__ extr,
                             int W, int H, float nearest, float extent, float2 *__restrict__ uv, float *__restrict__ depth,
                             float *__restrict__ conic, int *__restrict__ radius, int *__restrict__ tiles) {
    const int i = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (i >= P) return;
    Cam c;
    load_cam(nullptr, extr, c);
    const float p0[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float4 q4 = uquats[i];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    float c3[6], a[3], b[3], t[3], Jm[4], cov[3];
    cov3d_pt(s, q, c3);
    ewa_T<true>(c, p0, W, H, a, b, t, Jm);     // (orthographic: a, b do not depend on the position)
    ewa_cov2d<true>(a, b, c3, cov);
    const int per = (F + gridDim.y - 1) / gridDim.y;
    const int f0 = blockIdx.y * per, f1 = imin_(F, f0 + per);
    for (int f = f0; f < f1; ++f) {
        const size_t fo = (size_t)f * P + i;
        float p[3] = {p0[0], p0[1], p0[2]};
        if (offset) {
            const float *o = offset + fo * 3;
            p[0] += o[0]; p[1] += o[1]; p[2] += o[2];
        }
        float u, v, d;
        const bool cull = project_ortho_pt(c, p[0], p[1], p[2], W, H, nearest, extent, u, v, d);
        u = cull ? 0.f : u; v = cull ? 0.f : v; d = cull ? 0.f : d;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        int orad = 0, otiles = 0;
        if (d != 0.f) ewa_finish_pt<true>(cov, make_float2(u, v), W, H, o0, o1, o2, orad, otiles);
        uv[fo] = make_float2(u, v);
        depth[fo] = d;
        conic[3 * fo] = o0; conic[3 * fo + 1] = o1; conic[3 * fo + 2] = o2;
        radius[fo] = orad;
        if (tiles) tiles[fo] = otiles;
    }
}

template <bool ACC>
__device__ __forceinline__ void put(float *p, float v) {
    if (ACC) *p += v;
    else *p = v;
}

template <bool ORTHO, bool ACC>
__global__ void __launch_bounds__(PP_BLOCK)
preprocess_bwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ offset,
                      const float *__restrict__ scales, const float4 *__restrict__ uquats,
                      const float *__restrict__ intr, const float *__restrict__ extr, int W, int H,
                      const float *__restrict__ depth,
                      const int *__restrict__ radius, const float *__restrict__ dL_duv,
                      const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dconic,
                      float *__restrict__ dL_dxyz, float *__restrict__ dL_dscales,
                      float *__restrict__ dL_duquats) {
    const int i = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (i >= P) return;
    float gp[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (depth[i] != 0.f) {
        Cam c;
        load_cam(ORTHO ? nullptr : intr, extr, c);
        float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        if (offset) {
            p[0] += offset[3 * i]; p[1] += offset[3 * i + 1]; p[2] += offset[3 * i + 2];
        }
        if (dL_dxyz) {
            const float gu = dL_duv ? dL_duv[2 * i] : 0.f, gv = dL_duv ? dL_duv[2 * i + 1] : 0.f, gd = dL_ddepth ? dL_ddepth[i] : 0.f;
            if (ORTHO) project_ortho_grad_pt(c, W, H, gu, gv, gd, gp);
            else project_persp_grad_pt(c, p[0], p[1], p[2], gu, gv, gd, gp);
        }
        if (radius[i] > 0 && dL_dconic && (dL_dscales || dL_duquats || (!ORTHO && dL_dxyz))) {
            const float4 q4 = uquats[i];
            const float q[4] = {q4.x, q4.y, q4.z, q4.w};
            const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
            float c3[6], a[3], b[3], t[3], Jm[4], cov[3];
            cov3d_pt(s, q, c3);
            ewa_T<ORTHO>(c, p, W, H, a, b, t, Jm);
            ewa_cov2d<ORTHO>(a, b, c3, cov);
            const float det = cov[0] * cov[2] - cov[1] * cov[1];
            if (det != 0.0f) {
                const float g3[3] = {dL_dconic[3 * i], dL_dconic[3 * i + 1], dL_dconic[3 * i + 2]};
                float dcx, dcy, dcz, g6[6];
                ewa_grad_cov_pt(a, b, cov, det, g3, dcx, dcy, dcz, g6);
                cov3d_grad_pt(s, q, g6, ds, dq);
                if (!ORTHO) {  // the perspective Jacobian depends on the point (ewa_project.cu:152-205)
                    float ge[3], da[3], db[3], dt[3];
                    ewa_grad_pos_persp_pt(c, t, a, b, c3, dcx, dcy, dcz, ge, da, db, dt);
                    gp[0] += ge[0]; gp[1] += ge[1]; gp[2] += ge[2];
                }
            }
        }
    }
    if (dL_dxyz) {
#pragma unroll
        for (int k = 0; k < 3; ++k) put<ACC>(dL_dxyz + 3 * i + k, gp[k]);
    }
    if (dL_dscales) {
#pragma unroll
        for (int k = 0; k < 3; ++k) put<ACC>(dL_dscales + 3 * i + k, ds[k]);
    }
    if (dL_duquats) {
#pragma unroll
        for (int k = 0; k < 4; ++k) put<ACC>(dL_duquats + 4 * i + k, dq[k]);
    }
}


// ------------------------------------------------------------------ dynamic Gaussians -> screen space in one pass
// the per-frame evaluation of the dynamic Gaussians (row a15) fused into the preprocess.  One
// quad of lanes per Gaussian as in dynamics.hip (coalesced float4 table rows, component-per-lane stores); after the
// quad has assembled position / unit quaternion / scale, all four lanes run the projection + cov3d + EWA of the
// Gaussian redundantly (a few hundred flops against ~330 bytes of HBM traffic) and each stores its own components.
struct DynFrame {
    float pos[3], q[4], scl[3];  // per-frame position, normalised rotation, activated scale
    float qraw[4], nrm;          // un-normalised rotation and its norm (backward)
};

__device__ __forceinline__ DynFrame dyn_frame(int n, int j, const CubicAddr &ca, float d, const DynBasis &b,
                                              const float *position, const float *cubic, const float *rotation,
                                              const float4 *rot_poly, const float4 *rot_fourier, const float *scaling) {
    float pj = 0.f, sj = 0.f;
    if (j < 3) {
        const float *c = cubic + ca.seg_off + (size_t)n * ca.stride_n + j;
        const size_t row = ca.stride_k;
        const float c0 = c[0], c1 = c[row], c2 = c[2 * row], c3 = c[3 * row];
        float p = c3 + c2 * d;
        p = p + c1 * (d * d);
        p = p + c0 * (d * d * d);
        pj = p + position[(size_t)n * 3 + j];
        sj = expf(scaling[(size_t)n * 3 + j]);
    }
    const float qj = quat_component(n, j, b, rotation, rot_poly, rot_fourier);
    const float nr = sqrtf(quad_sum(qj * qj));
    const float qn = qj / fmaxf(nr, 1e-12f);  // F.normalize: x / max(|x|, eps)
    DynFrame f;
    f.pos[0] = quad_bcast<0>(pj); f.pos[1] = quad_bcast<1>(pj); f.pos[2] = quad_bcast<2>(pj);
    f.scl[0] = quad_bcast<0>(sj); f.scl[1] = quad_bcast<1>(sj); f.scl[2] = quad_bcast<2>(sj);
    f.q[0] = quad_bcast<0>(qn); f.q[1] = quad_bcast<1>(qn); f.q[2] = quad_bcast<2>(qn); f.q[3] = quad_bcast<3>(qn);
    f.qraw[0] = quad_bcast<0>(qj); f.qraw[1] = quad_bcast<1>(qj); f.qraw[2] = quad_bcast<2>(qj); f.qraw[3] = quad_bcast<3>(qj);
    f.nrm = nr;
    return f;
}

// frame-loop form: position row, log-scale and the rotation table rows stay in registers across the frames
struct DynStatic {
    float pos_j, scl_j;  // position[n][j], exp(scaling[n][j])  (j < 3)
    QuatRows rows;
};
__device__ __forceinline__ DynStatic dyn_static(int n, int j, const float *position, const float *rotation,
                                                const float4 *rot_poly, const float4 *rot_fourier, const float *scaling) {
    DynStatic s;
    s.pos_j = j < 3 ? position[(size_t)n * 3 + j] : 0.f;
    s.scl_j = j < 3 ? expf(scaling[(size_t)n * 3 + j]) : 0.f;
    s.rows = load_quat_rows(n, j, rotation, rot_poly, rot_fourier);
    return s;
}
__device__ __forceinline__ DynFrame dyn_frame_rows(int n, int j, const CubicAddr &ca, float d, const float *basis12,
                                                   const DynStatic &st, const float *cubic) {
    float pj = 0.f;
    if (j < 3) {
        const float *c = cubic + ca.seg_off + (size_t)n * ca.stride_n + j;
        const size_t row = ca.stride_k;
        const float c0 = c[0], c1 = c[row], c2 = c[2 * row], c3 = c[3 * row];
        float p = c3 + c2 * d;
        p = p + c1 * (d * d);
        p = p + c0 * (d * d * d);
        pj = p + st.pos_j;
    }
    const float qj = quat_component_rows(st.rows, j, basis12[j], basis12[4 + j], basis12[8 + j]);
    const float nr = sqrtf(quad_sum(qj * qj));
    const float qn = qj / fmaxf(nr, 1e-12f);
    DynFrame f;
    f.pos[0] = quad_bcast<0>(pj); f.pos[1] = quad_bcast<1>(pj); f.pos[2] = quad_bcast<2>(pj);
    f.scl[0] = quad_bcast<0>(st.scl_j); f.scl[1] = quad_bcast<1>(st.scl_j); f.scl[2] = quad_bcast<2>(st.scl_j);
    f.q[0] = quad_bcast<0>(qn); f.q[1] = quad_bcast<1>(qn); f.q[2] = quad_bcast<2>(qn); f.q[3] = quad_bcast<3>(qn);
    f.qraw[0] = quad_bcast<0>(qj); f.qraw[1] = quad_bcast<1>(qj); f.qraw[2] = quad_bcast<2>(qj); f.qraw[3] = quad_bcast<3>(qj);
    f.nrm = nr;
    return f;
}

__global__ void __launch_bounds__(DYN_BLOCK)
frame_preprocess_fwd_kernel(int P, CubicAddr ca, float d, DynBasis b, const float *__restrict__ position,
                            const float *__restrict__ cubic, const float *__restrict__ rotation,
                            const float4 *__restrict__ rot_poly, const float4 *__restrict__ rot_fourier,
                            const float *__restrict__ opacity, const float *__restrict__ scaling,
                            const float *__restrict__ extr, int W, int H, float nearest, float extent,
                            float *__restrict__ uv, float *__restrict__ depth, float *__restrict__ conic,
                            int *__restrict__ radius, int *__restrict__ tiles, float *__restrict__ opa_t) {
    const int t = blockIdx.x * DYN_BLOCK + threadIdx.x;
    const int n = t >> 2, j = t & 3;
    if (n >= P) return;  // whole quads leave together
    const DynFrame f = dyn_frame(n, j, ca, d, b, position, cubic, rotation, rot_poly, rot_fourier, scaling);
    Cam c;
    load_cam(nullptr, extr, c);
    float u, v, dep;
    const bool cull = project_ortho_pt(c, f.pos[0], f.pos[1], f.pos[2], W, H, nearest, extent, u, v, dep);
    u = cull ? 0.f : u; v = cull ? 0.f : v; dep = cull ? 0.f : dep;
    float o[3] = {0.f, 0.f, 0.f};
    int orad = 0, otiles = 0;
    if (dep != 0.f) {
        float c3[6], a[3], bb[3], tt[3], Jm[4], cov[3];
        cov3d_pt(f.scl, f.q, c3);
        ewa_T<true>(c, f.pos, W, H, a, bb, tt, Jm);
        ewa_cov2d<true>(a, bb, c3, cov);
        ewa_finish_pt<true>(cov, make_float2(u, v), W, H, o[0], o[1], o[2], orad, otiles);
    }
    if (j < 2) uv[(size_t)n * 2 + j] = j == 0 ? u : v;
    if (j < 3) conic[(size_t)n * 3 + j] = j == 0 ? o[0] : j == 1 ? o[1] : o[2];
    if (j == 2) depth[n] = dep;
    if (j == 3) {
        radius[n] = orad;
        tiles[n] = otiles;
        opa_t[n] = 1.0f / (1.0f + expf(-opacity[n]));
    }
}

template <bool ACC>
__global__ void __launch_bounds__(DYN_BLOCK)
frame_preprocess_bwd_kernel(int P, CubicAddr ca, float d, DynBasis b, const float *__restrict__ position,
                            const float *__restrict__ cubic, const float *__restrict__ rotation,
                            const float4 *__restrict__ rot_poly, const float4 *__restrict__ rot_fourier,
                            const float *__restrict__ opacity, const float *__restrict__ scaling,
                            const float *__restrict__ extr, int W, int H, const float *__restrict__ depth,
                            const int *__restrict__ radius, const float *__restrict__ dL_duv,
                            const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dconic,
                            const float *__restrict__ dL_dopa, float *__restrict__ d_position,
                            float *__restrict__ d_cubic, float *__restrict__ d_rotation, float *__restrict__ d_opacity,
                            float *__restrict__ d_scaling) {
    const int t = blockIdx.x * DYN_BLOCK + threadIdx.x;
    const int n = t >> 2, j = t & 3;
    if (n >= P) return;
    const DynFrame f = dyn_frame(n, j, ca, d, b, position, cubic, rotation, rot_poly, rot_fourier, scaling);
    float gp[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (depth[n] != 0.f) {
        Cam c;
        load_cam(nullptr, extr, c);
        if (dL_duv) project_ortho_grad_pt(c, W, H, dL_duv[2 * n], dL_duv[2 * n + 1], dL_ddepth ? dL_ddepth[n] : 0.f, gp);
        if (radius[n] > 0 && dL_dconic) {
            float c3[6], a[3], bb[3], tt[3], Jm[4], cov[3];
            cov3d_pt(f.scl, f.q, c3);
            ewa_T<true>(c, f.pos, W, H, a, bb, tt, Jm);
            ewa_cov2d<true>(a, bb, c3, cov);
            const float det = cov[0] * cov[2] - cov[1] * cov[1];
            if (det != 0.0f) {
                const float g3[3] = {dL_dconic[3 * n], dL_dconic[3 * n + 1], dL_dconic[3 * n + 2]};
                float dcx, dcy, dcz, g6[6];
                ewa_grad_cov_pt(a, bb, cov, det, g3, dcx, dcy, dcz, g6);
                cov3d_grad_pt(f.scl, f.q, g6, ds, dq);
            }
        }
    }
    // chain through the activations: scale = exp(scaling), rotation = normalize(raw)  (poly / Fourier sums detached)
    if (j < 3) {
        const float g = j == 0 ? gp[0] : j == 1 ? gp[1] : gp[2];
        if (d_position) put<ACC>(d_position + (size_t)n * 3 + j, g);
        if (d_cubic) {  // only the active segment of the spline table is touched
            float *c = d_cubic + ca.seg_off + (size_t)n * ca.stride_n + j;
            const size_t row = ca.stride_k;
            put<ACC>(c, g * (d * d * d));
            put<ACC>(c + row, g * (d * d));
            put<ACC>(c + 2 * row, g * d);
            put<ACC>(c + 3 * row, g);
        }
        if (d_scaling) {
            const float dsj = j == 0 ? ds[0] : j == 1 ? ds[1] : ds[2];
            const float sj = j == 0 ? f.scl[0] : j == 1 ? f.scl[1] : f.scl[2];
            put<ACC>(d_scaling + (size_t)n * 3 + j, dsj * sj);
        }
    }
    if (d_rotation) {
        const float gq = j == 0 ? dq[0] : j == 1 ? dq[1] : j == 2 ? dq[2] : dq[3];
        float r;
        if (f.nrm < 1e-12f) {
            r = gq / 1e-12f;
        } else {
            const float dot = f.q[0] * dq[0] + f.q[1] * dq[1] + f.q[2] * dq[2] + f.q[3] * dq[3];
            const float qh = j == 0 ? f.q[0] : j == 1 ? f.q[1] : j == 2 ? f.q[2] : f.q[3];
            r = (gq - qh * dot) / f.nrm;
        }
        put<ACC>(d_rotation + (size_t)n * 4 + j, r);
    }
    if (d_opacity && j == 3) {
        const float s = 1.0f / (1.0f + expf(-opacity[n]));
        put<ACC>(d_opacity + n, (dL_dopa ? dL_dopa[n] : 0.f) * s * (1.0f - s));
    }
}

inline dim3 dyn_grid(int P) { return dim3((unsigned)(((size_t)P * 4 + DYN_BLOCK - 1) / DYN_BLOCK)); }

// ------------------------------------------------------------------ Gaussian-side backward of a frame batch
// After the tile kernels of a frame batch every (frame, tile, splat) pair owns one gradient record
// [ux uy ca cb cc o | ax ay | features] at frame * cap + slot, a Gaussian's records of one frame being contiguous
// (goff = inclusive prefix of tiles per Gaussian, per frame).  One quad of lanes per Gaussian streams them -- lane `sub`
// owns the 16-byte chunks sub, sub + 4, .. of every record, as pair_reduce does for one frame -- over ALL frames, then
// runs the preprocess backward and writes every parameter gradient once per batch.
//
// Static parameters + per-frame offsets under the orthographic camera: conic, radius and the projection Jacobian do not
// depend on the frame (the offsets only move the centre), and the whole backward chain is linear in (dL_duv,
// dL_dconic): the records of all frames are summed first and the chain runs ONCE per Gaussian.
// One-camera walk: a frame's first U records of a Gaussian are requested together (two register buffers: the next frame's while
// this frame's are summed), the records past them TAIL at a time -- every round of them is a dependent memory round trip, and 6 %
// of the bench scene's splats touch more than six tiles (most waves hold one).  Narrow (one-chunk-per-lane) records: U = 6
// (17.4 us per frame at 4, 16.5 at 6, 16.2 at 8 with the tail two at a time), TAIL = 6 (17.2 -> 14.4); wide records: U = 2 (at
// four the registers cost more waves than the loads in flight gain), TAIL = 4 (c5: 43.6 -> 40.9 us per frame).
#ifndef GAUSS_BWD_U
#define GAUSS_BWD_U 6
#endif
#ifndef GAUSS_BWD_TAIL
#define GAUSS_BWD_TAIL 6
#endif
#ifndef GAUSS_BWD_U_WIDE
#define GAUSS_BWD_U_WIDE 2
#endif
#ifndef GAUSS_BWD_TAIL_WIDE
#define GAUSS_BWD_TAIL_WIDE 4
#endif
struct GaussBwdArgs {
    int F, P, W, H;
    int C, cn;              // row stride of d_feature, channels carried by the records
    long long cap;          // records per frame
    const float *pair;      // [F, cap, NCP]
    const int *goff;        // [F, P]
    const int *radius;      // [F, P] (optional: radii_max / visibility output)
    const float *xyz, *scales;
    const float4 *uquats;
    const float *extr;
    // cameras that differ from frame to frame and / or the perspective camera: the projection chain runs per frame (CAM
    // template parameter); frame f's camera = extr + f * extr_fs (intr + f * intr_fs), its positions = xyz + offsets[f]
    const float *intr, *offsets;
    long long extr_fs, intr_fs;
    float *d_xyz, *d_scales, *d_uquats, *d_opacity, *d_feature;
    float *tap, *abs_tap;   // optional [P,2]: sum over the frames of dL_duv * (W/2, H/2) (and of its abs twin)
    int *radii_max;         // optional [P]: max over the frames of the screen radius (visibility = radii_max > 0)
    int accumulate;         // add to the parameter gradients instead of storing
    int skip_opacity;       // this feature set was blended with opacity.detach(): no opacity gradient
    int depth_channel;      // >= 0: that channel of the set is the per-frame depth feature -- its summed gradient is
                            // dL/ddepth of the projection (-> position), not a feature gradient
    // SETS records (blend_bwd_sets_kernel: [ux uy ca cb | cc o ax ay | tx ty 0 0 | row channels], common.h SETS_NG): per set the first row
    // channel, the width, the gradient rows (NULL: not wanted) and their stride
    // ... generalised to SOURCES (splat_feature_source_t): nsrc entries; sfs[g] != 0 marks a per-frame source (static kernels:
    // refused by the host side)
    int nsrc;
    int sc0[SPLAT_MAX_SOURCES], scn[SPLAT_MAX_SOURCES], sstride[SPLAT_MAX_SOURCES];
    float *sdf[SPLAT_MAX_SOURCES];
    long long sfs[SPLAT_MAX_SOURCES];
};

// component k of the quad's record sum (chunk k / 4 = lane (k / 4) & 3, register a[k / 16], element k & 3) on every lane
template <int NS>
__device__ __forceinline__ float record_component(const float4 (&a)[NS], int k, int sub) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const float e4[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k == 16 * c + 4 * sub + e) v = e4[e];
    }
    return quad_sum(v);  // exactly one lane holds it
}

// CAM: 0 = ONE orthographic camera for the batch: conic, radius and the projection Jacobian do not depend on the frame, the
// chain is linear in the summed (dL_duv, dL_dconic): it runs once on the sums of all frames.  1 = orthographic cameras that
// differ from frame to frame, 2 = perspective (its Jacobian depends on the point, hence on the frame's offset): the records
// of every frame go through that frame's projection / EWA backward; only dL/dcov3d (linear into scale / rotation) is summed
// over the frames first.
template <bool ABS, int NCP, bool SETS = false, int CAM = 0>
__global__ void __launch_bounds__(256)
frames_gauss_bwd_static_kernel(const GaussBwdArgs A) {
    constexpr int NG = SETS ? SETS_NG : GradLayout<ABS, false>::NG;
    constexpr int NQ = NCP / 4, NS = (NQ + 3) / 4;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 2, sub = t & 3;
    if (CAM != 0 && i >= A.P) return;  // whole quads leave together (CAM == 0: after the workgroup's barrier below)
    float4 a[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    int rmax = 0;
    bool any = false;
    int nbeg = 0, nend = 0;
    if (CAM != 0) { nbeg = i > 0 ? A.goff[i - 1] : 0; nend = A.goff[i]; }
    // per-frame chain (CAM > 0): position gradient and dL/dcov3d summed over the frames
    float gpf[3] = {0.f, 0.f, 0.f}, g6f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float c3s[6], ss[3], qs[4];
    if (CAM > 0) {
        const float4 q4 = A.uquats[i];
        qs[0] = q4.x; qs[1] = q4.y; qs[2] = q4.z; qs[3] = q4.w;
        ss[0] = A.scales[3 * i]; ss[1] = A.scales[3 * i + 1]; ss[2] = A.scales[3 * i + 2];
        cov3d_pt(ss, qs, c3s);
    }
    if (CAM == 0) {
        // One camera: the records of all frames are only summed.  The walk is bound by bytes in flight (two records per quad and
        // a slot range that depends on the previous frame's: 2.6-2.9 TB/s): the slot ranges of ALL frames of the workgroup's 64
        // Gaussians are read up front into LDS (coalesced; the record loads then depend on nothing in flight), a frame's first
        // U records are requested together, and the next frame's while this frame's are summed (two register buffers).
        constexpr int U = NS == 1 ? GAUSS_BWD_U : GAUSS_BWD_U_WIDE, FMAX = 32;
        constexpr int TU = NS == 1 ? GAUSS_BWD_TAIL : GAUSS_BWD_TAIL_WIDE;   // records a tail round requests together
        __shared__ int s_goff[FMAX][65];   // [frame][Gaussian of the workgroup + 1]: inclusive prefix, entry 0 = the Gaussian before
        const int i0 = (int)(blockIdx.x * 64);
        const int nf = imin_(A.F, FMAX);
        for (int c = threadIdx.x; c < nf * 65; c += 256) {
            const int f = c / 65, g = c - f * 65;
            const int gi = i0 + g - 1;
            s_goff[f][g] = (gi >= 0 && gi < A.P) ? A.goff[(size_t)f * A.P + gi] : 0;
        }
        __syncthreads();
        if (i >= A.P) return;
        const int li = i - i0;
        auto range = [&](int f, int &b, int &e) {
            if (f < FMAX) { b = s_goff[f][li]; e = s_goff[f][li + 1]; }
            else { const int *goff = A.goff + (size_t)f * A.P; b = i > 0 ? goff[i - 1] : 0; e = goff[i]; }
        };
        float4 bufA[U][NS], bufB[U][NS];
        int radA = 0, radB = 0;
        auto issue = [&](float4 (&v)[U][NS], int &rad, int f, int beg, int end) {
            const float *base = A.pair + (size_t)f * (size_t)A.cap * NCP + 4 * sub;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NS; ++c)
                    v[u][c] = (beg + u < end && 4 * c + sub < NQ)
                                  ? *reinterpret_cast<const float4 *>(base + (size_t)(beg + u) * NCP + 16 * c)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            rad = (A.radii_max && sub == 0) ? A.radius[(size_t)f * A.P + i] : 0;
        };
        auto consume = [&](const float4 (&v)[U][NS], int rad, int f, int beg, int end) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    a[c].x += v[u][c].x; a[c].y += v[u][c].y; a[c].z += v[u][c].z; a[c].w += v[u][c].w;
                }
            rmax = imax_(rmax, rad);
            any = any || end > beg;
            const float *base = A.pair + (size_t)f * (size_t)A.cap * NCP + 4 * sub;
            for (int j = beg + U; j < end; j += TU) {   // a Gaussian on more than U tiles: TU more records in flight
                float4 vt[TU][NS];
#pragma unroll
                for (int u = 0; u < TU; ++u)
#pragma unroll
                    for (int c = 0; c < NS; ++c)
                        vt[u][c] = (j + u < end && 4 * c + sub < NQ)
                                       ? *reinterpret_cast<const float4 *>(base + (size_t)(j + u) * NCP + 16 * c)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < TU; ++u)
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        a[c].x += vt[u][c].x; a[c].y += vt[u][c].y; a[c].z += vt[u][c].z; a[c].w += vt[u][c].w;
                    }
            }
        };
        int b0, e0, b1 = 0, e1 = 0;
        range(0, b0, e0);
        issue(bufA, radA, 0, b0, e0);
        for (int f = 0; f < A.F; f += 2) {
            if (f + 1 < A.F) { range(f + 1, b1, e1); issue(bufB, radB, f + 1, b1, e1); }
            consume(bufA, radA, f, b0, e0);
            if (f + 2 < A.F) { range(f + 2, b0, e0); issue(bufA, radA, f + 2, b0, e0); }
            if (f + 1 < A.F) consume(bufB, radB, f + 1, b1, e1);
        }
    }
    for (int f = 0; CAM > 0 && f < A.F; ++f) {
        float4 atot[CAM > 0 ? NS : 1];
        if (CAM > 0) {   // this frame's records are summed on their own (a = this frame, atot = the frames before)
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                atot[c] = a[c];
                a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const int beg = nbeg, end = nend;
        if (f + 1 < A.F) {  // the next frame's slot range is in flight while this frame's records are summed
            const int *goff = A.goff + (size_t)(f + 1) * A.P;
            nbeg = i > 0 ? goff[i - 1] : 0;
            nend = goff[i];
        }
        if (A.radii_max && sub == 0) rmax = imax_(rmax, A.radius[(size_t)f * A.P + i]);
        any = any || end > beg;
        const float *base = A.pair + (size_t)f * (size_t)A.cap * NCP + 4 * sub;
        int j = beg;
        for (; j + 1 < end; j += 2) {  // two records in flight
            float4 v0[NS], v1[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const bool mine = 4 * c + sub < NQ;
                v0[c] = mine ? *reinterpret_cast<const float4 *>(base + (size_t)j * NCP + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                v1[c] = mine ? *reinterpret_cast<const float4 *>(base + (size_t)(j + 1) * NCP + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                a[c].x += v0[c].x; a[c].y += v0[c].y; a[c].z += v0[c].z; a[c].w += v0[c].w;
                a[c].x += v1[c].x; a[c].y += v1[c].y; a[c].z += v1[c].z; a[c].w += v1[c].w;
            }
        }
        if (j < end) {
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                if (4 * c + sub < NQ) {
                    const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)j * NCP + 16 * c);
                    a[c].x += v.x; a[c].y += v.y; a[c].z += v.z; a[c].w += v.w;
                }
            }
        }
        if (CAM > 0) {
            if (end > beg) {  // a record exists: visible with radius > 0 in this frame (the chain's preconditions)
                float ux = quad_bcast<0>(a[0].x), uy = quad_bcast<0>(a[0].y);
                float g3[3] = {quad_bcast<0>(a[0].z), quad_bcast<0>(a[0].w), quad_bcast<1>(a[0].x)};
                const float gdep = A.depth_channel >= 0 ? record_component<NS>(a, NG + A.depth_channel, sub) : 0.f;
                Cam c;
                load_cam(CAM == 2 ? A.intr + (size_t)f * A.intr_fs : nullptr, A.extr + (size_t)f * A.extr_fs, c);
                float p[3] = {A.xyz[3 * i], A.xyz[3 * i + 1], A.xyz[3 * i + 2]};
                if (A.offsets) {
                    const float *o = A.offsets + ((size_t)f * A.P + i) * 3;
                    p[0] += o[0]; p[1] += o[1]; p[2] += o[2];
                }
                float g[3];
                if (CAM == 2) project_persp_grad_pt(c, p[0], p[1], p[2], ux, uy, gdep, g);
                else project_ortho_grad_pt(c, A.W, A.H, ux, uy, gdep, g);
                float ea[3], eb[3], et[3], Jm[4], cov[3];
                ewa_T<CAM != 2>(c, p, A.W, A.H, ea, eb, et, Jm);
                ewa_cov2d<CAM != 2>(ea, eb, c3s, cov);
                const float det = cov[0] * cov[2] - cov[1] * cov[1];
                if (det != 0.0f) {
                    float dcx, dcy, dcz, g6[6];
                    ewa_grad_cov_pt(ea, eb, cov, det, g3, dcx, dcy, dcz, g6);
#pragma unroll
                    for (int k = 0; k < 6; ++k) g6f[k] += g6[k];
                    if (CAM == 2) {
                        float ge[3], da[3], db[3], dt[3];
                        ewa_grad_pos_persp_pt(c, et, ea, eb, c3s, dcx, dcy, dcz, ge, da, db, dt);
                        g[0] += ge[0]; g[1] += ge[1]; g[2] += ge[2];
                    }
                }
                gpf[0] += g[0]; gpf[1] += g[1]; gpf[2] += g[2];
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                a[c].x += atot[c].x; a[c].y += atot[c].y; a[c].z += atot[c].z; a[c].w += atot[c].w;
            }
        }
    }
    // geometry sums: chunk 0 (lane 0) = ux uy ca cb, chunk 1 (lane 1) = cc o [ax ay]
    float ux = quad_bcast<0>(a[0].x), uy = quad_bcast<0>(a[0].y);
    float g3[3] = {quad_bcast<0>(a[0].z), quad_bcast<0>(a[0].w), quad_bcast<1>(a[0].x)};
    float dop = quad_bcast<1>(a[0].y);
    // dL/ddepth (a set whose channel `depth_channel` is the depth feature): component NG + channel
    const float gdep = (CAM == 0 && A.depth_channel >= 0) ? record_component<NS>(a, NG + A.depth_channel, sub) : 0.f;
    float gp[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (CAM > 0) {
        gp[0] = gpf[0]; gp[1] = gpf[1]; gp[2] = gpf[2];
        if (any) cov3d_grad_pt(ss, qs, g6f, ds, dq);
    } else if (any) {  // a record exists: the Gaussian was visible with radius > 0 in that frame (the chain's preconditions)
        Cam c;
        load_cam(nullptr, A.extr, c);
        project_ortho_grad_pt(c, A.W, A.H, ux, uy, gdep, gp);
        const float p[3] = {A.xyz[3 * i], A.xyz[3 * i + 1], A.xyz[3 * i + 2]};
        const float4 q4 = A.uquats[i];
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float s[3] = {A.scales[3 * i], A.scales[3 * i + 1], A.scales[3 * i + 2]};
        float c3[6], ea[3], eb[3], et[3], Jm[4], cov[3];
        cov3d_pt(s, q, c3);
        ewa_T<true>(c, p, A.W, A.H, ea, eb, et, Jm);
        ewa_cov2d<true>(ea, eb, c3, cov);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det != 0.0f) {
            float dcx, dcy, dcz, g6[6];
            ewa_grad_cov_pt(ea, eb, cov, det, g3, dcx, dcy, dcz, g6);
            cov3d_grad_pt(s, q, g6, ds, dq);
        }
    }
    const bool acc = A.accumulate != 0;
    auto put1 = [acc](float *p, float v) { *p = acc ? *p + v : v; };
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) put1(A.d_xyz + 3 * i + k, gp[k]);
        if (!A.skip_opacity) put1(A.d_opacity + i, dop);
        if (A.radii_max) A.radii_max[i] = rmax;
    } else if (sub == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) put1(A.d_scales + 3 * i + k, ds[k]);
    } else if (sub == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) put1(A.d_uquats + 4 * i + k, dq[k]);
    }
    {   // densification tap: d uv of the whole blend -- SETS records: of the tap set alone (components 8, 9 = chunk 2)
        const float tu = SETS ? quad_bcast<2>(a[0].x) : ux, tv = SETS ? quad_bcast<2>(a[0].y) : uy;
        if (sub == 3 && A.tap) {
            A.tap[2 * i] = tu * (0.5f * (float)A.W);
            A.tap[2 * i + 1] = tv * (0.5f * (float)A.H);
        }
    }
    if (ABS) {
        const float ax = quad_bcast<1>(a[0].z), ay = quad_bcast<1>(a[0].w);
        if (sub == 3 && A.abs_tap) {
            A.abs_tap[2 * i] = ax * (0.5f * (float)A.W);
            A.abs_tap[2 * i + 1] = ay * (0.5f * (float)A.H);
        }
    }
    // features: component k of the record lives in chunk k / 4 = lane (k / 4) & 3, register a[k / 16]
    if (SETS) {   // routed by source (a uniform loop over the table; the component tests are per lane)
        for (int g = 0; g < A.nsrc; ++g) {
            float *df = A.sdf[g];
            if (!df) continue;
            const int c0 = A.sc0[g], cn = A.scn[g], st = A.sstride[g];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                if (4 * c + sub < NQ) {
                    const float v[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ch = 16 * c + 4 * sub + e - NG;
                        if (ch >= c0 && ch < c0 + cn && ch != A.depth_channel) put1(df + (size_t)i * st + (ch - c0), v[e]);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        if (4 * c + sub < NQ) {
            const int k0 = 16 * c + 4 * sub;
            const float v[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = k0 + e - NG;
                if (ch >= 0 && ch < A.cn && ch != A.depth_channel && A.d_feature) {
                    put1(A.d_feature + (size_t)i * A.C + ch, v[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ frame batch of DYNAMIC Gaussians (rows a15 + f1)
// Forward: frame_preprocess_fwd_kernel with the frame as grid.y and the per-frame scalars read from a device table.
__global__ void __launch_bounds__(DYN_BLOCK)
frame_preprocess_fwd_batch_kernel(int F, int P, int I, int layout, const DynTab *__restrict__ tab, const float *__restrict__ position,
                                  const float *__restrict__ cubic, const float *__restrict__ rotation,
                                  const float4 *__restrict__ rot_poly, const float4 *__restrict__ rot_fourier,
                                  const float *__restrict__ opacity, const float *__restrict__ scaling,
                                  const float *__restrict__ extr, int W, int H, float nearest, float extent,
                                  float *__restrict__ uv, float *__restrict__ depth, float *__restrict__ conic,
                                  int *__restrict__ radius, float *__restrict__ opa_t) {
    // one quad per Gaussian loops over the frames of its slice (grid.y slices of the batch keep the chip full): the
    // frozen rotation tables (192 B), position and scale are read once, only the 48-byte spline segment per frame
    const int t = blockIdx.x * DYN_BLOCK + threadIdx.x;
    const int n = t >> 2, j = t & 3;
    if (n >= P) return;  // whole quads leave together
    const DynStatic stc = dyn_static(n, j, position, rotation, rot_poly, rot_fourier, scaling);
    Cam c;
    load_cam(nullptr, extr, c);
    const int per = (F + gridDim.y - 1) / gridDim.y;
    const int f0 = blockIdx.y * per, f1 = imin_(F, f0 + per);
    if (blockIdx.y == 0 && j == 3) opa_t[n] = 1.0f / (1.0f + expf(-opacity[n]));  // frame independent
    // four frames per trip: the quad evaluates frame f + k's position / rotation together (lane j = component j), lane k
    // keeps it and runs the projection / cov3d / EWA chain for that frame alone (the chain on all four lanes per frame was
    // 4x redundant), then writes its frame's seven outputs
    const float scl3[3] = {quad_bcast<0>(stc.scl_j), quad_bcast<1>(stc.scl_j), quad_bcast<2>(stc.scl_j)};
    for (int f = f0; f < f1; f += 4) {
        float m_pos[3] = {0.f, 0.f, 0.f}, m_q[4] = {1.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ff = imin_(f + k, f1 - 1);  // (a repeated last frame: its lane does not store)
            const CubicAddr ca = cubic_addr(layout, P, I, tab[ff].seg);
            const DynFrame fr = dyn_frame_rows(n, j, ca, tab[ff].d, tab[ff].basis, stc, cubic);
            if (j == k) {
                m_pos[0] = fr.pos[0]; m_pos[1] = fr.pos[1]; m_pos[2] = fr.pos[2];
                m_q[0] = fr.q[0]; m_q[1] = fr.q[1]; m_q[2] = fr.q[2]; m_q[3] = fr.q[3];
            }
        }
        if (f + j >= f1) continue;
        float u, v, dep;
        const bool cull = project_ortho_pt(c, m_pos[0], m_pos[1], m_pos[2], W, H, nearest, extent, u, v, dep);
        u = cull ? 0.f : u; v = cull ? 0.f : v; dep = cull ? 0.f : dep;
        float o[3] = {0.f, 0.f, 0.f};
        int orad = 0, otiles = 0;
        if (dep != 0.f) {
            float c3[6], a[3], bb[3], tt[3], Jm[4], cov[3];
            cov3d_pt(scl3, m_q, c3);
            ewa_T<true>(c, m_pos, W, H, a, bb, tt, Jm);
            ewa_cov2d<true>(a, bb, c3, cov);
            ewa_finish_pt<true>(cov, make_float2(u, v), W, H, o[0], o[1], o[2], orad, otiles);
        }
        const size_t fn = (size_t)(f + j) * P + n;
        *reinterpret_cast<float2 *>(uv + fn * 2) = make_float2(u, v);
        conic[fn * 3] = o[0]; conic[fn * 3 + 1] = o[1]; conic[fn * 3 + 2] = o[2];
        depth[fn] = dep;
        radius[fn] = orad;
    }
}

// Backward: one quad per Gaussian walks ALL frames: per frame it sums the Gaussian's pair records (lane `sub` owns the
// record chunks sub, sub + 4, ..), re-evaluates the frame's position / rotation (dyn_frame), runs projection + EWA +
// cov3d backward and chains through the activations -- scale = exp, rotation = normalize(raw + detached sums), opacity =
// sigmoid, position = base + cubic segment.  Everything accumulates in registers; the spline segment's four coefficient
// rows are flushed when the walk leaves the segment (frames of a batch are time-ordered: a handful of flushes).
#ifndef GAUSS_DYN_MINW
#define GAUSS_DYN_MINW 3   // 168 registers, no scratch (4: 128 registers and 156 bytes of scratch per lane -- 62 vs 54 us per frame for the
                           // three-set records at c2, round 4)
#endif
struct GaussDynArgs {
    int F, P, W, H, I, layout;
    int C, cn;
    long long cap;
    const float *pair;
    const int *goff;
    const int *radius;
    const DynTab *tab;
    const float *position, *cubic, *rotation;
    const float4 *rot_poly, *rot_fourier;
    const float *opacity, *scaling, *extr;
    float *d_position, *d_cubic, *d_rotation, *d_opacity, *d_scaling, *d_feature;  // d_cubic / d_* are ADDED to
    float *tap, *abs_tap;
    int *radii_max;
    // SETS records (blend_bwd_sets_kernel): see GaussBwdArgs
    // (sources: nsrc entries; sfs[g] != 0 = per-frame source whose gradient is ADDED per frame at sdf[g] + f * sfs[g]; npf = their count)
    int nsrc, npf;
    int spfq[SPLAT_MAX_SOURCES];   // per-frame source inside one 16-byte chunk of the record: that chunk's index, else -1
    int sc0[SPLAT_MAX_SOURCES], scn[SPLAT_MAX_SOURCES], sstride[SPLAT_MAX_SOURCES];
    float *sdf[SPLAT_MAX_SOURCES];
    long long sfs[SPLAT_MAX_SOURCES];
    int depth_channel;
};

// PF: the row has per-frame sources (A.npf > 0) -- its own instantiation: their block costs the walk ten registers
template <bool ABS, int NCP, bool SETS = false, bool PF = false>
__global__ void __launch_bounds__(256, GAUSS_DYN_MINW)
frames_gauss_bwd_dynamic_kernel(const GaussDynArgs A) {
    constexpr int NG = SETS ? SETS_NG : GradLayout<ABS, false>::NG;
    constexpr int NQ = NCP / 4, NS = (NQ + 3) / 4;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int n = t >> 2, j = t & 3;
    if (n >= A.P) return;
    float4 atot[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) atot[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    Cam cam;
    load_cam(nullptr, A.extr, cam);
    const DynStatic stc = dyn_static(n, j, A.position, A.rotation, A.rot_poly, A.rot_fourier, A.scaling);
    float acc_c[4] = {0.f, 0.f, 0.f, 0.f};  // gradient of the active segment's coefficient rows c0..c3, component j
    float d_pos = 0.f, d_rot = 0.f, d_scl = 0.f, tap_u = 0.f, tap_v = 0.f, atap_u = 0.f, atap_v = 0.f;
    int cur_seg = -1, rmax = 0;
    auto flush = [&](int seg) {
        if (seg < 0 || !A.d_cubic || j >= 3) return;
        const CubicAddr ca = cubic_addr(A.layout, A.P, A.I, seg);
        float *c = A.d_cubic + ca.seg_off + (size_t)n * ca.stride_n + j;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k * ca.stride_k] += acc_c[k];
    };
    // The frames are walked in groups of (up to) four of one spline segment: the quad sums frame f + k's records and
    // re-evaluates its position / rotation together (lane j = component j), lane k KEEPS frame f + k; then every lane runs
    // the projection / EWA / cov3d / normalisation chain ONCE, for its own frame (the chain on all four lanes for every
    // frame was 4x redundant), and quad sums hand component j of the four frames' results to lane j.
    float scl3[3] = {quad_bcast<0>(stc.scl_j), quad_bcast<1>(stc.scl_j), quad_bcast<2>(stc.scl_j)};
    int f = 0;
    while (f < A.F) {
        const int seg = A.tab[f].seg;
        if (seg != cur_seg) {
            flush(cur_seg);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc_c[k] = 0.f;
            cur_seg = seg;
        }
        const CubicAddr ca = cubic_addr(A.layout, A.P, A.I, seg);
        float m_ux = 0.f, m_uy = 0.f, m_g3[3] = {0.f, 0.f, 0.f}, m_pos[3] = {0.f, 0.f, 0.f}, m_q[4] = {1.f, 0.f, 0.f, 0.f};
        float m_nrm = 1.f, m_d = 0.f, m_gdep = 0.f;
        bool m_valid = false;
        int took = 0;
        // the slot ranges (and radii) of the group's four frames are requested together: one memory round trip in front of the
        // records' instead of one per frame
        int rb[4], re[4], rr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ff = imin_(f + k, A.F - 1);
            const int *goff = A.goff + (size_t)ff * A.P;
            rb[k] = n > 0 ? goff[n - 1] : 0;
            re[k] = goff[n];
            rr[k] = (A.radii_max && j == 0) ? A.radius[(size_t)ff * A.P + n] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ff = f + k;
            if (ff >= A.F || A.tab[ff].seg != seg) break;  // grid-uniform
            took = k + 1;
            const int beg = rb[k], end = re[k];
            rmax = imax_(rmax, rr[k]);
            if (end <= beg) continue;  // quad-uniform: no record of this Gaussian in frame ff
            float4 af[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) af[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *base = A.pair + (size_t)ff * (size_t)A.cap * NCP + 4 * j;
            if constexpr (NS == 1) {
                // narrow records: six requested together (one dependent round trip for 94 % of the splats; two at a time was 2 .. 3)
                constexpr int TD = 6;
                for (int r = beg; r < end; r += TD) {
                    float4 v[TD];
#pragma unroll
                    for (int u = 0; u < TD; ++u)
                        v[u] = (r + u < end && j < NQ) ? *reinterpret_cast<const float4 *>(base + (size_t)(r + u) * NCP) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < TD; ++u) { af[0].x += v[u].x; af[0].y += v[u].y; af[0].z += v[u].z; af[0].w += v[u].w; }
                }
            } else {   // wide records: two in flight (the registers of a third or fourth cost more than the round trips they save)
                int r = beg;
                for (; r + 1 < end; r += 2) {
                    float4 v0[NS], v1[NS];
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        const bool mine = 4 * c + j < NQ;
                        v0[c] = mine ? *reinterpret_cast<const float4 *>(base + (size_t)r * NCP + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                        v1[c] = mine ? *reinterpret_cast<const float4 *>(base + (size_t)(r + 1) * NCP + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        af[c].x += v0[c].x; af[c].y += v0[c].y; af[c].z += v0[c].z; af[c].w += v0[c].w;
                        af[c].x += v1[c].x; af[c].y += v1[c].y; af[c].z += v1[c].z; af[c].w += v1[c].w;
                    }
                }
                if (r < end) {
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        if (4 * c + j < NQ) {
                            const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)r * NCP + 16 * c);
                            af[c].x += v.x; af[c].y += v.y; af[c].z += v.z; af[c].w += v.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                atot[c].x += af[c].x; atot[c].y += af[c].y; atot[c].z += af[c].z; atot[c].w += af[c].w;
            }
            if (SETS && PF) {
                // per-frame sources (track_gs = position(ids2), src/trainer_fragGS.py:506-511): THIS frame's gradient of their
                // channels goes to the frame's own rows; the lane that holds a component adds it (a quad owns its Gaussian)
                for (int g = 0; g < A.nsrc; ++g) {
                    if (A.sfs[g] == 0 || !A.sdf[g]) continue;
                    const int c0 = A.sc0[g], cn = A.scn[g];
                    float *dst = A.sdf[g] + (size_t)ff * (size_t)A.sfs[g] + (size_t)n * A.sstride[g];
                    const int qc = A.spfq[g];
                    if (qc >= 0) {   // at most four channels inside ONE 16-byte chunk of the record (track_gs: chunk 4): one lane adds them
                        float4 v = af[0];
#pragma unroll
                        for (int c = 1; c < NS; ++c)
                            if ((qc >> 2) == c) v = af[c];
                        if (j == (qc & 3)) {
                            const float e4[4] = {v.x, v.y, v.z, v.w};
                            const int e0 = (NG + c0) & 3;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (e >= e0 && e - e0 < cn) dst[e - e0] += e4[e];
                        }
                        continue;
                    }
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        if (4 * c + j < NQ) {
                            const float v[4] = {af[c].x, af[c].y, af[c].z, af[c].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int ch = 16 * c + 4 * j + e - NG;
                                if (ch >= c0 && ch < c0 + cn) dst[ch - c0] += v[e];
                            }
                        }
                    }
                }
            }
            float ux = quad_bcast<0>(af[0].x), uy = quad_bcast<0>(af[0].y);
            float ga = quad_bcast<0>(af[0].z), gb = quad_bcast<0>(af[0].w), gc = quad_bcast<1>(af[0].x);
            float gdep = 0.f;  // SETS: the frame's gradient of the depth channel = dL/ddepth of the projection
            // densification taps: d uv of the whole blend -- SETS records: of the tap set alone (components 8, 9 = chunk 2)
            tap_u += SETS ? quad_bcast<2>(af[0].x) : ux;
            tap_v += SETS ? quad_bcast<2>(af[0].y) : uy;
            if (ABS) {
                atap_u += quad_bcast<1>(af[0].z); atap_v += quad_bcast<1>(af[0].w);
            }
            if (SETS && A.depth_channel >= 0) {
                const int kd = NG + A.depth_channel;  // chunk kd / 4 -> lane (kd / 4) & 3, register af[kd / 16], element kd & 3
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    const float e4[4] = {af[c].x, af[c].y, af[c].z, af[c].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kd == 16 * c + 4 * j + e) v = e4[e];
                }
                gdep = quad_sum(v);
            }
            const float dseg = A.tab[ff].d;
            const DynFrame fr = dyn_frame_rows(n, j, ca, dseg, A.tab[ff].basis, stc, A.cubic);
            if (j == k) {
                m_ux = ux; m_uy = uy; m_g3[0] = ga; m_g3[1] = gb; m_g3[2] = gc;
                m_pos[0] = fr.pos[0]; m_pos[1] = fr.pos[1]; m_pos[2] = fr.pos[2];
                m_q[0] = fr.q[0]; m_q[1] = fr.q[1]; m_q[2] = fr.q[2]; m_q[3] = fr.q[3];
                m_nrm = fr.nrm; m_d = dseg; m_gdep = gdep; m_valid = true;
            }
        }
        f += took;
        // ---- the chain, once per lane for the lane's own frame
        float gp[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f};
        if (m_valid) {
            float dq[4] = {0.f, 0.f, 0.f, 0.f};
            project_ortho_grad_pt(cam, A.W, A.H, m_ux, m_uy, m_gdep, gp);
            float c3[6], ea[3], eb[3], et[3], Jm[4], cov[3];
            cov3d_pt(scl3, m_q, c3);
            ewa_T<true>(cam, m_pos, A.W, A.H, ea, eb, et, Jm);
            ewa_cov2d<true>(ea, eb, c3, cov);
            const float det = cov[0] * cov[2] - cov[1] * cov[1];
            if (det != 0.0f) {
                float dcx, dcy, dcz, g6[6];
                ewa_grad_cov_pt(ea, eb, cov, det, m_g3, dcx, dcy, dcz, g6);
                cov3d_grad_pt(scl3, m_q, g6, ds, dq);
            }
            if (m_nrm < 1e-12f) {
#pragma unroll
                for (int c = 0; c < 4; ++c) rq[c] = dq[c] / 1e-12f;
            } else {
                const float dot = m_q[0] * dq[0] + m_q[1] * dq[1] + m_q[2] * dq[2] + m_q[3] * dq[3];
#pragma unroll
                for (int c = 0; c < 4; ++c) rq[c] = (dq[c] - m_q[c] * dot) / m_nrm;
            }
        }
        // ---- component jj of the four frames' results to lane jj
        const float d1 = m_d, d2 = m_d * m_d, d3 = m_d * m_d * m_d;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const float s0 = quad_sum(gp[jj]), s1 = quad_sum(gp[jj] * d1), s2 = quad_sum(gp[jj] * d2), s3 = quad_sum(gp[jj] * d3);
            const float ss = quad_sum(ds[jj] * scl3[jj]);
            if (j == jj) {
                d_pos += s0;
                acc_c[0] += s3; acc_c[1] += s2; acc_c[2] += s1; acc_c[3] += s0;
                d_scl += ss;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float sr = quad_sum(rq[jj]);
            if (j == jj) d_rot += sr;
        }
    }
    flush(cur_seg);
    if (j < 3) {
        if (A.d_position) A.d_position[(size_t)n * 3 + j] += d_pos;
        if (A.d_scaling) A.d_scaling[(size_t)n * 3 + j] += d_scl;
    }
    if (A.d_rotation) A.d_rotation[(size_t)n * 4 + j] += d_rot;
    const float dop = quad_bcast<1>(atot[0].y);
    if (j == 3) {
        if (A.d_opacity) {
            const float s = 1.0f / (1.0f + expf(-A.opacity[n]));
            A.d_opacity[n] += dop * s * (1.0f - s);
        }
        if (A.tap) {
            A.tap[2 * n] = tap_u * (0.5f * (float)A.W);
            A.tap[2 * n + 1] = tap_v * (0.5f * (float)A.H);
        }
        if (ABS && A.abs_tap) {
            A.abs_tap[2 * n] = atap_u * (0.5f * (float)A.W);
            A.abs_tap[2 * n + 1] = atap_v * (0.5f * (float)A.H);
        }
    }
    if (j == 0 && A.radii_max) A.radii_max[n] = rmax;
    if (SETS) {   // shared sources: the sum over the frames (per-frame sources were written frame by frame)
        for (int g = 0; g < A.nsrc; ++g) {
            float *df = A.sdf[g];
            if (!df || A.sfs[g] != 0) continue;
            const int c0 = A.sc0[g], cn = A.scn[g], st = A.sstride[g];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                if (4 * c + j < NQ) {
                    const float v[4] = {atot[c].x, atot[c].y, atot[c].z, atot[c].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ch = 16 * c + 4 * j + e - NG;
                        if (ch >= c0 && ch < c0 + cn && ch != A.depth_channel) df[(size_t)n * st + (ch - c0)] += v[e];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        if (4 * c + j < NQ) {
            const int k0 = 16 * c + 4 * j;
            const float v[4] = {atot[c].x, atot[c].y, atot[c].z, atot[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = k0 + e - NG;
                if (ch >= 0 && ch < A.cn && A.d_feature) {
                    A.d_feature[(size_t)n * A.C + ch] += v[e];
                }
            }
        }
    }
}

template <bool ABS>
int launch_gauss_bwd_dynamic(const GaussDynArgs &A, int ncp, hipStream_t s) {
    const dim3 grid((unsigned)(((size_t)A.P * 4 + 255) / 256)), block(256);
#define GD(N) case N: SPLAT_LAUNCH("gauss_bwd", (frames_gauss_bwd_dynamic_kernel<ABS, N>), grid, block, 0, s, A); break
    switch (ncp) {
        GD(8); GD(12); GD(16); GD(20); GD(24); GD(28); GD(32); GD(36); GD(40);
        default: splat_set_error("gauss_bwd: unsupported record stride %d", ncp); return SPLAT_E_ARG;
    }
#undef GD
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <bool ABS, int CAM = 0>
int launch_gauss_bwd_static(const GaussBwdArgs &A, int ncp, hipStream_t s) {
    const dim3 grid((unsigned)(((size_t)A.P * 4 + 255) / 256)), block(256);
#define GB(N) case N: SPLAT_LAUNCH("gauss_bwd", (frames_gauss_bwd_static_kernel<ABS, N, false, CAM>), grid, block, 0, s, A); break
    switch (ncp) {
        GB(8); GB(12); GB(16); GB(20); GB(24); GB(28); GB(32); GB(36); GB(40);
        default: splat_set_error("gauss_bwd: unsupported record stride %d", ncp); return SPLAT_E_ARG;
    }
#undef GB
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

int launch_gauss_bwd_dynamic_sets(const GaussDynArgs &A, int ncp, hipStream_t s) {
    const dim3 grid((unsigned)(((size_t)A.P * 4 + 255) / 256)), block(256);
#define GDS(N) case N: if (A.npf > 0) SPLAT_LAUNCH("gauss_bwd", (frames_gauss_bwd_dynamic_kernel<true, N, true, true>), grid, block, 0, s, A); \
                    else SPLAT_LAUNCH("gauss_bwd", (frames_gauss_bwd_dynamic_kernel<true, N, true>), grid, block, 0, s, A); break
    switch (ncp) {
        GDS(12); GDS(16); GDS(20); GDS(24); GDS(28); GDS(32); GDS(36); GDS(40);
        default: splat_set_error("gauss_bwd: unsupported record stride %d", ncp); return SPLAT_E_ARG;
    }
#undef GDS
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <int CAM = 0>
int launch_gauss_bwd_static_sets(const GaussBwdArgs &A, int ncp, hipStream_t s) {
    const dim3 grid((unsigned)(((size_t)A.P * 4 + 255) / 256)), block(256);
#define GS(N) case N: SPLAT_LAUNCH("gauss_bwd", (frames_gauss_bwd_static_kernel<true, N, true, CAM>), grid, block, 0, s, A); break
    switch (ncp) {
        GS(12); GS(16); GS(20); GS(24); GS(28); GS(32); GS(36); GS(40);
        default: splat_set_error("gauss_bwd: unsupported record stride %d", ncp); return SPLAT_E_ARG;
    }
#undef GS
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

}  // namespace

extern "C" size_t splat_blend_pair_stride(int C, int want_abs, int has_bias);
extern "C" size_t splat_blend_sets_pair_stride(int C);

// Gaussian side of splat_alpha_blending_backward_batch_sets: sums the SETS records of every Gaussian over the F frames and
// runs the projection chain once.  set_dfeature: HOST array of three device pointers (NULL entries: no gradient wanted for
// that set; the channel `depth_channel` of the row, if >= 0, is the per-frame depth and feeds d_xyz instead).
// camera of a batch -> kernel arguments; returns the chain mode (CAM of frames_gauss_bwd_static_kernel) or -1
static int set_camera(GaussBwdArgs &A, const splat_camera_t *cam, int F) {
    if (!cam || !cam->extr) return -1;
    A.extr = cam->extr; A.intr = cam->intr; A.offsets = cam->offsets;
    A.extr_fs = cam->extr_frame_stride; A.intr_fs = cam->intr_frame_stride;
    if (cam->perspective) return cam->intr ? 2 : -1;
    return (cam->extr_frame_stride != 0 && F > 1) ? 1 : 0;
}

extern "C" int splat_frames_gauss_backward_static_sets_cam(int F, int P, int C, int W, int H, int64_t capacity,
                                                           const float *pair_records, const int32_t *goff_incl,
                                                           const int32_t *radius, const float *xyz, const float *scales,
                                                           const float *uquats, const splat_camera_t *cam, int accumulate,
                                                           float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                           const int32_t *set_c0, const int32_t *set_cn,
                                                           float *const *set_dfeature, const int32_t *set_stride,
                                                           int depth_channel, float *tap, float *abs_tap,
                                                           int32_t *radii_max, void *stream);

extern "C" int splat_frames_gauss_backward_static_sets(int F, int P, int C, int W, int H, int64_t capacity,
                                                       const float *pair_records, const int32_t *goff_incl,
                                                       const int32_t *radius, const float *xyz, const float *scales,
                                                       const float *uquats, const float *extr, int accumulate,
                                                       float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                       const int32_t *set_c0, const int32_t *set_cn,
                                                       float *const *set_dfeature, const int32_t *set_stride,
                                                       int depth_channel, float *tap, float *abs_tap,
                                                       int32_t *radii_max, void *stream) {
    splat_camera_t cam;
    memset(&cam, 0, sizeof(cam));
    cam.extr = extr;
    return splat_frames_gauss_backward_static_sets_cam(F, P, C, W, H, capacity, pair_records, goff_incl, radius, xyz, scales, uquats,
                                                       &cam, accumulate, d_xyz, d_scales, d_uquats, d_opacity, set_c0, set_cn,
                                                       set_dfeature, set_stride, depth_channel, tap, abs_tap, radii_max, stream);
}

#define STAT_SETS_PARAMS int F, int P, int C, int W, int H, int64_t capacity, const float *pair_records, const int32_t *goff_incl, \
    const int32_t *radius, const float *xyz, const float *scales, const float *uquats, const splat_camera_t *cam, int accumulate, \
    float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity, const int32_t *set_c0, const int32_t *set_cn, \
    float *const *set_dfeature, const int32_t *set_stride, int depth_channel, float *tap, float *abs_tap, int32_t *radii_max, void *stream
#define STAT_SETS_ARGS F, P, C, W, H, capacity, pair_records, goff_incl, radius, xyz, scales, uquats, cam, accumulate, d_xyz, d_scales, \
    d_uquats, d_opacity, set_c0, set_cn, set_dfeature, set_stride, depth_channel, tap, abs_tap, radii_max, stream
extern "C" int splat_frames_gauss_backward_static_sets_cam(STAT_SETS_PARAMS) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= 28 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 28)");
    SPLAT_CHECK_ARG(pair_records && goff_incl && xyz && scales && uquats && cam, "null input pointer");
    SPLAT_CHECK_ARG(d_xyz && d_scales && d_uquats && d_opacity, "null gradient pointer");
    SPLAT_CHECK_ARG(set_c0 && set_cn && set_dfeature && set_stride, "null set table");
    SPLAT_CHECK_ARG(depth_channel < C, "depth_channel outside the row");
    SPLAT_CHECK_ARG(!radii_max || radius, "radii_max needs the per-frame radius");
    GaussBwdArgs A;
    memset(&A, 0, sizeof(A));
    A.F = F; A.P = P; A.W = W; A.H = H; A.C = C; A.cn = C; A.cap = capacity;
    A.skip_opacity = 0; A.depth_channel = depth_channel;
    A.pair = pair_records; A.goff = goff_incl; A.radius = radius;
    A.xyz = xyz; A.scales = scales; A.uquats = (const float4 *)uquats;
    const int mode = set_camera(A, cam, F);
    SPLAT_CHECK_ARG(mode >= 0, "camera: extr (and intr for the perspective camera) must be set");
    A.d_xyz = d_xyz; A.d_scales = d_scales; A.d_uquats = d_uquats; A.d_opacity = d_opacity;
    A.tap = tap; A.abs_tap = abs_tap; A.radii_max = radii_max; A.accumulate = accumulate;
    A.nsrc = 3;
    for (int g = 0; g < 3; ++g) {
        A.sc0[g] = set_c0[g]; A.scn[g] = set_cn[g]; A.sstride[g] = set_stride[g]; A.sdf[g] = set_dfeature[g];
        SPLAT_CHECK_ARG(set_cn[g] == 0 || !set_dfeature[g] || set_stride[g] >= set_cn[g], "feature stride below the set width");
    }
    const int ncp = (int)splat_blend_sets_pair_stride(C);
    return mode == 0 ? launch_gauss_bwd_static_sets<0>(A, ncp, (hipStream_t)stream)
         : mode == 1 ? launch_gauss_bwd_static_sets<1>(A, ncp, (hipStream_t)stream)
                     : launch_gauss_bwd_static_sets<2>(A, ncp, (hipStream_t)stream);
}

// the same with the row described by feature SOURCES (include/splat_hip.h: splat_feature_source_t); shared sources only
extern "C" int splat_frames_gauss_backward_static_sources_cam(int F, int P, int C, int W, int H, int64_t capacity,
                                                              const float *pair_records, const int32_t *goff_incl,
                                                              const int32_t *radius, const float *xyz, const float *scales,
                                                              const float *uquats, const splat_camera_t *cam, int accumulate,
                                                              float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                              int nsrc, const splat_feature_source_t *src, int depth_channel,
                                                              float *tap, float *abs_tap, int32_t *radii_max, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= 28 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 28)");
    SPLAT_CHECK_ARG(pair_records && goff_incl && xyz && scales && uquats && cam, "null input pointer");
    SPLAT_CHECK_ARG(d_xyz && d_scales && d_uquats && d_opacity, "null gradient pointer");
    SPLAT_CHECK_ARG(src && nsrc >= 0 && nsrc <= SPLAT_MAX_SOURCES, "null / oversized source table");
    SPLAT_CHECK_ARG(depth_channel < C, "depth_channel outside the row");
    SPLAT_CHECK_ARG(!radii_max || radius, "radii_max needs the per-frame radius");
    GaussBwdArgs A;
    memset(&A, 0, sizeof(A));
    A.F = F; A.P = P; A.W = W; A.H = H; A.C = C; A.cn = C; A.cap = capacity;
    A.skip_opacity = 0; A.depth_channel = depth_channel;
    A.pair = pair_records; A.goff = goff_incl; A.radius = radius;
    A.xyz = xyz; A.scales = scales; A.uquats = (const float4 *)uquats;
    const int mode = set_camera(A, cam, F);
    SPLAT_CHECK_ARG(mode >= 0, "camera: extr (and intr for the perspective camera) must be set");
    A.d_xyz = d_xyz; A.d_scales = d_scales; A.d_uquats = d_uquats; A.d_opacity = d_opacity;
    A.tap = tap; A.abs_tap = abs_tap; A.radii_max = radii_max; A.accumulate = accumulate;
    A.nsrc = nsrc;
    for (int g = 0; g < nsrc; ++g) {
        SPLAT_CHECK_ARG(src[g].cn >= 0 && src[g].c0 >= 0 && src[g].c0 + src[g].cn <= C, "source outside the row");
        SPLAT_CHECK_ARG(!(src[g].d_feature && src[g].frame_stride != 0 && F > 1),
                        "a per-frame source's gradient needs the dynamic entry point (static Gaussians: the records of all frames are summed)");
        A.sc0[g] = src[g].c0; A.scn[g] = src[g].cn; A.sstride[g] = src[g].cn; A.sdf[g] = src[g].d_feature;
    }
    const int ncp = (int)splat_blend_sets_pair_stride(C);
    return mode == 0 ? launch_gauss_bwd_static_sets<0>(A, ncp, (hipStream_t)stream)
         : mode == 1 ? launch_gauss_bwd_static_sets<1>(A, ncp, (hipStream_t)stream)
                     : launch_gauss_bwd_static_sets<2>(A, ncp, (hipStream_t)stream);
}

extern "C" int splat_preprocess_forward_batch_cam(int F, int P, const float *xyz, const float *offsets,
                                                  const float *scales, const float *uquats, const splat_camera_t *cam, int W,
                                                  int H, float nearest, float extent, float *uv, float *depth,
                                                  float *conic, int32_t *radius, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && F <= 65535 && P >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(xyz && scales && uquats && cam && cam->extr && uv && depth && conic && radius, "null pointer");
    SPLAT_CHECK_ARG(!cam->perspective || cam->intr, "the perspective camera needs intr");
    SPLAT_CHECK_ARG(F == 1 || offsets || cam->extr_frame_stride != 0,
                    "several frames of static Gaussians need per-frame offsets [F,P,3] or per-frame cameras");
    const dim3 grid = pp_grid(P);
    if (!cam->perspective && (cam->extr_frame_stride == 0 || F == 1)) {
        // one orthographic camera: covariance once per Gaussian, a thread walks its slice of the frames
        unsigned slices = 1;
        while (grid.x * slices < 4096u && slices * 2 <= (unsigned)F) slices *= 2;
        SPLAT_LAUNCH("preprocess_fwd", preprocess_fwd_frames_kernel, dim3(grid.x, slices), dim3(PP_BLOCK), 0, (hipStream_t)stream, F, P,
                     xyz, offsets, scales, (const float4 *)uquats, cam->extr, W, H, nearest, extent, (float2 *)uv, depth, conic, radius,
                     (int *)nullptr);
        SPLAT_POST_LAUNCH();
        return SPLAT_OK;
    }
    if (cam->perspective)
        SPLAT_LAUNCH("preprocess_fwd", preprocess_fwd_kernel<false>, dim3(grid.x, F), dim3(PP_BLOCK), 0, (hipStream_t)stream, P,
                     xyz, offsets, scales, (const float4 *)uquats, cam->intr, cam->extr, (long long)cam->intr_frame_stride,
                     (long long)cam->extr_frame_stride, W, H, nearest, extent, (float2 *)uv, depth, conic, radius, (int *)nullptr);
    else
        SPLAT_LAUNCH("preprocess_fwd", preprocess_fwd_kernel<true>, dim3(grid.x, F), dim3(PP_BLOCK), 0, (hipStream_t)stream, P,
                     xyz, offsets, scales, (const float4 *)uquats, (const float *)nullptr, cam->extr, 0ll,
                     (long long)cam->extr_frame_stride, W, H, nearest, extent, (float2 *)uv, depth, conic, radius, (int *)nullptr);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_preprocess_ortho_forward_batch(int F, int P, const float *xyz, const float *offsets,
                                                    const float *scales, const float *uquats, const float *extr, int W,
                                                    int H, float nearest, float extent, float *uv, float *depth,
                                                    float *conic, int32_t *radius, void *stream) {
    splat_camera_t cam;
    memset(&cam, 0, sizeof(cam));
    cam.extr = extr;
    SPLAT_CHECK_ARG(F == 1 || offsets, "several frames of static Gaussians need per-frame offsets [F,P,3]");
    return splat_preprocess_forward_batch_cam(F, P, xyz, offsets, scales, uquats, &cam, W, H, nearest, extent, uv, depth, conic,
                                              radius, stream);
}

static int gauss_backward_static(int F, int P, int C, int W, int H, int64_t capacity, int want_abs,
                                 const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                 const float *xyz, const float *scales, const float *uquats, const splat_camera_t *cam,
                                 int accumulate, float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                 float *d_feature, int feature_stride, int skip_opacity, int depth_channel, float *tap,
                                 float *abs_tap, int32_t *radii_max, void *stream);
static splat_camera_t ortho_camera(const float *extr) {
    splat_camera_t cam;
    memset(&cam, 0, sizeof(cam));
    cam.extr = extr;
    return cam;
}

// one feature set under any camera of splat_camera_t (per-frame cameras, perspective): the general form of the two below
extern "C" int splat_frames_gauss_backward_static_cam(int F, int P, int cn, int W, int H, int64_t capacity, int want_abs,
                                                      const float *pair_records, const int32_t *goff_incl,
                                                      const int32_t *radius, const float *xyz, const float *scales,
                                                      const float *uquats, const splat_camera_t *cam, int accumulate,
                                                      float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                      float *d_feature, int feature_stride, int skip_opacity,
                                                      int depth_channel, float *tap, float *abs_tap, int32_t *radii_max,
                                                      void *stream) {
    SPLAT_CHECK_ARG(skip_opacity || d_opacity, "null gradient pointer");
    SPLAT_CHECK_ARG(depth_channel < cn, "depth_channel outside the set");
    return gauss_backward_static(F, P, cn, W, H, capacity, want_abs, pair_records, goff_incl, radius, xyz, scales, uquats, cam,
                                 accumulate, d_xyz, d_scales, d_uquats, d_opacity, d_feature, feature_stride, skip_opacity,
                                 depth_channel, tap, abs_tap, radii_max, stream);
}

extern "C" int splat_frames_gauss_backward_static(int F, int P, int C, int W, int H, int64_t capacity, int want_abs,
                                                  const float *pair_records, const int32_t *goff_incl,
                                                  const int32_t *radius, const float *xyz, const float *scales,
                                                  const float *uquats, const float *extr, int accumulate,
                                                  float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                  float *d_feature, float *tap, float *abs_tap, int32_t *radii_max,
                                                  void *stream) {
    SPLAT_CHECK_ARG(d_opacity && d_feature, "null gradient pointer");
    const splat_camera_t cam = ortho_camera(extr);
    return gauss_backward_static(F, P, C, W, H, capacity, want_abs, pair_records, goff_incl, radius, xyz, scales, uquats, &cam,
                                 accumulate, d_xyz, d_scales, d_uquats, d_opacity, d_feature, C, 0, -1, tap, abs_tap, radii_max,
                                 stream);
}

// one feature SET of a multi-set batch (see splat_alpha_blending_backward_batch_set): cn channels per record,
// d_feature rows `feature_stride` floats apart (NULL: no feature gradient wanted), the set's routing flags
extern "C" int splat_frames_gauss_backward_static_set(int F, int P, int cn, int W, int H, int64_t capacity, int want_abs,
                                                      const float *pair_records, const int32_t *goff_incl,
                                                      const int32_t *radius, const float *xyz, const float *scales,
                                                      const float *uquats, const float *extr, int accumulate,
                                                      float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                      float *d_feature, int feature_stride, int skip_opacity,
                                                      int depth_channel, float *tap, float *abs_tap, int32_t *radii_max,
                                                      void *stream) {
    SPLAT_CHECK_ARG(skip_opacity || d_opacity, "null gradient pointer");
    SPLAT_CHECK_ARG(depth_channel < cn, "depth_channel outside the set");
    const splat_camera_t cam = ortho_camera(extr);
    return gauss_backward_static(F, P, cn, W, H, capacity, want_abs, pair_records, goff_incl, radius, xyz, scales, uquats, &cam,
                                 accumulate, d_xyz, d_scales, d_uquats, d_opacity, d_feature, feature_stride, skip_opacity,
                                 depth_channel, tap, abs_tap, radii_max, stream);
}

static int gauss_backward_static(int F, int P, int C, int W, int H, int64_t capacity, int want_abs,
                                 const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                 const float *xyz, const float *scales, const float *uquats, const splat_camera_t *cam,
                                 int accumulate, float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                 float *d_feature, int feature_stride, int skip_opacity, int depth_channel, float *tap,
                                 float *abs_tap, int32_t *radii_max, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= 32 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 32)");
    SPLAT_CHECK_ARG(pair_records && goff_incl && xyz && scales && uquats && cam, "null input pointer");
    SPLAT_CHECK_ARG(d_xyz && d_scales && d_uquats, "null gradient pointer");
    SPLAT_CHECK_ARG(!abs_tap || want_abs, "abs_tap needs records with the abs sums");
    SPLAT_CHECK_ARG(!radii_max || radius, "radii_max needs the per-frame radius");
    GaussBwdArgs A;
    memset(&A, 0, sizeof(A));
    A.F = F; A.P = P; A.W = W; A.H = H; A.C = feature_stride; A.cn = C; A.cap = capacity;
    A.skip_opacity = skip_opacity; A.depth_channel = depth_channel;
    A.pair = pair_records; A.goff = goff_incl; A.radius = radius;
    A.xyz = xyz; A.scales = scales; A.uquats = (const float4 *)uquats;
    const int mode = set_camera(A, cam, F);
    SPLAT_CHECK_ARG(mode >= 0, "camera: extr (and intr for the perspective camera) must be set");
    A.d_xyz = d_xyz; A.d_scales = d_scales; A.d_uquats = d_uquats; A.d_opacity = d_opacity; A.d_feature = d_feature;
    A.tap = tap; A.abs_tap = abs_tap; A.radii_max = radii_max; A.accumulate = accumulate;
    const int ncp = (int)splat_blend_pair_stride(C, want_abs, 0);
    hipStream_t hs = (hipStream_t)stream;
    if (mode == 0) return want_abs ? launch_gauss_bwd_static<true, 0>(A, ncp, hs) : launch_gauss_bwd_static<false, 0>(A, ncp, hs);
    if (mode == 1) return want_abs ? launch_gauss_bwd_static<true, 1>(A, ncp, hs) : launch_gauss_bwd_static<false, 1>(A, ncp, hs);
    return want_abs ? launch_gauss_bwd_static<true, 2>(A, ncp, hs) : launch_gauss_bwd_static<false, 2>(A, ncp, hs);
}

extern "C" int splat_preprocess_ortho_forward(int P, const float *xyz, const float *offset, const float *scales,
                                              const float *uquats, const float *extr, int W, int H, float nearest,
                                              float extent, float *uv, float *depth, float *conic, int32_t *radius,
                                              int32_t *tiles, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && scales && uquats && extr && uv && depth && conic && radius && tiles, "null pointer");
    // (the frame batch's kernel at F = 1: a batch's frame and this operator give the same bits)
    SPLAT_LAUNCH("preprocess_fwd", preprocess_fwd_frames_kernel, pp_grid(P), dim3(PP_BLOCK), 0, (hipStream_t)stream, 1, P, xyz, offset,
                 scales, (const float4 *)uquats, extr, W, H, nearest, extent, (float2 *)uv, depth, conic, radius, tiles);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// The pinhole camera's chain of gs.rasterization / DPTRRender (project_point -> compute_cov3d -> ewa_project; reference
// src/submodules/dptr/dptr/gs/__init__.py:55-77, dptr.py:107-147) in one pass, forward and backward (position gradient
// through the projection AND the EWA Jacobian; camera gradients: the separate operators).
extern "C" int splat_preprocess_persp_forward(int P, const float *xyz, const float *offset, const float *scales,
                                              const float *uquats, const float *intr, const float *extr, int W, int H,
                                              float nearest, float extent, float *uv, float *depth, float *conic,
                                              int32_t *radius, int32_t *tiles, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && scales && uquats && intr && extr && uv && depth && conic && radius && tiles, "null pointer");
    SPLAT_LAUNCH("preprocess_fwd", preprocess_fwd_kernel<false>, pp_grid(P), dim3(PP_BLOCK), 0, (hipStream_t)stream, P, xyz,
                 offset, scales, (const float4 *)uquats, intr, extr, 0ll, 0ll, W, H, nearest, extent, (float2 *)uv, depth, conic,
                 radius, tiles);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_preprocess_persp_backward(int P, const float *xyz, const float *offset, const float *scales,
                                               const float *uquats, const float *intr, const float *extr, int W, int H,
                                               const float *depth, const int32_t *radius, const float *dL_duv,
                                               const float *dL_ddepth, const float *dL_dconic, int accumulate,
                                               float *dL_dxyz, float *dL_dscales, float *dL_duquats, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && scales && uquats && intr && extr && depth && radius, "null pointer");
    SPLAT_CHECK_ARG(!(dL_dscales || dL_duquats) || dL_dconic, "scale / rotation gradients need dL_dconic");
    hipStream_t s = (hipStream_t)stream;
    if (accumulate)
        SPLAT_LAUNCH("preprocess_bwd", (preprocess_bwd_kernel<false, true>), pp_grid(P), dim3(PP_BLOCK), 0, s, P, xyz, offset,
                     scales, (const float4 *)uquats, intr, extr, W, H, depth, radius, dL_duv, dL_ddepth, dL_dconic, dL_dxyz,
                     dL_dscales, dL_duquats);
    else
        SPLAT_LAUNCH("preprocess_bwd", (preprocess_bwd_kernel<false, false>), pp_grid(P), dim3(PP_BLOCK), 0, s, P, xyz, offset,
                     scales, (const float4 *)uquats, intr, extr, W, H, depth, radius, dL_duv, dL_ddepth, dL_dconic, dL_dxyz,
                     dL_dscales, dL_duquats);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_preprocess_ortho_backward(int P, const float *xyz, const float *offset, const float *scales,
                                               const float *uquats, const float *extr, int W, int H,
                                               const float *depth, const int32_t *radius, const float *dL_duv,
                                               const float *dL_ddepth, const float *dL_dconic, int accumulate,
                                               float *dL_dxyz, float *dL_dscales, float *dL_duquats, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && scales && uquats && extr && depth && radius, "null pointer");
    SPLAT_CHECK_ARG(!dL_dxyz || dL_duv, "dL_dxyz needs dL_duv");
    SPLAT_CHECK_ARG(!(dL_dscales || dL_duquats) || dL_dconic, "scale / rotation gradients need dL_dconic");
    hipStream_t s = (hipStream_t)stream;
    if (accumulate)
        SPLAT_LAUNCH("preprocess_bwd", (preprocess_bwd_kernel<true, true>), pp_grid(P), dim3(PP_BLOCK), 0, s, P, xyz, offset,
                     scales, (const float4 *)uquats, (const float *)nullptr, extr, W, H, depth, radius, dL_duv, dL_ddepth,
                     dL_dconic, dL_dxyz, dL_dscales, dL_duquats);
    else
        SPLAT_LAUNCH("preprocess_bwd", (preprocess_bwd_kernel<true, false>), pp_grid(P), dim3(PP_BLOCK), 0, s, P, xyz, offset,
                     scales, (const float4 *)uquats, (const float *)nullptr, extr, W, H, depth, radius, dL_duv, dL_ddepth,
                     dL_dconic, dL_dxyz, dL_dscales, dL_duquats);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_frame_preprocess_forward(int P, int I, int seg, float d, const float *basis_host,
                                              const float *position, const float *cubic, int cubic_layout,
                                              const float *rotation, const float *rot_poly, const float *rot_fourier,
                                              const float *opacity, const float *scaling, const float *extr, int W,
                                              int H, float nearest, float extent, float *uv, float *depth,
                                              float *conic, int32_t *radius, int32_t *tiles, float *opa_t,
                                              void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && I >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(seg >= 0 && seg < I, "segment index out of range");
    SPLAT_CHECK_ARG(basis_host != nullptr, "basis_host (12 host floats) is required");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR,
                    "unknown cubic_layout");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(position && cubic && rotation && rot_poly && rot_fourier && opacity && scaling && extr,
                    "null input pointer");
    SPLAT_CHECK_ARG(uv && depth && conic && radius && tiles && opa_t, "null output pointer");
    SPLAT_LAUNCH("frame_preprocess_fwd", frame_preprocess_fwd_kernel, dyn_grid(P), dim3(DYN_BLOCK), 0, (hipStream_t)stream,
                 P, cubic_addr(cubic_layout, P, I, seg), d, load_basis(basis_host), position, cubic, rotation,
                 (const float4 *)rot_poly, (const float4 *)rot_fourier, opacity, scaling, extr, W, H, nearest, extent, uv,
                 depth, conic, radius, tiles, opa_t);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_frame_preprocess_backward(int P, int I, int seg, float d, const float *basis_host,
                                               const float *position, const float *cubic, int cubic_layout,
                                               const float *rotation, const float *rot_poly, const float *rot_fourier,
                                               const float *opacity, const float *scaling, const float *extr, int W,
                                               int H, const float *depth, const int32_t *radius, const float *dL_duv,
                                               const float *dL_ddepth, const float *dL_dconic, const float *dL_dopa,
                                               int accumulate, float *d_position, float *d_cubic, float *d_rotation,
                                               float *d_opacity, float *d_scaling, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && I >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(seg >= 0 && seg < I, "segment index out of range");
    SPLAT_CHECK_ARG(basis_host != nullptr, "basis_host (12 host floats) is required");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR,
                    "unknown cubic_layout");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(position && cubic && rotation && rot_poly && rot_fourier && opacity && scaling && extr && depth &&
                        radius,
                    "null input pointer");
    hipStream_t s = (hipStream_t)stream;
    const CubicAddr ca = cubic_addr(cubic_layout, P, I, seg);
    const DynBasis b = load_basis(basis_host);
    if (accumulate)
        SPLAT_LAUNCH("frame_preprocess_bwd", frame_preprocess_bwd_kernel<true>, dyn_grid(P), dim3(DYN_BLOCK), 0, s, P, ca, d,
                     b, position, cubic, rotation, (const float4 *)rot_poly, (const float4 *)rot_fourier, opacity, scaling,
                     extr, W, H, depth, radius, dL_duv, dL_ddepth, dL_dconic, dL_dopa, d_position, d_cubic, d_rotation,
                     d_opacity, d_scaling);
    else
        SPLAT_LAUNCH("frame_preprocess_bwd", frame_preprocess_bwd_kernel<false>, dyn_grid(P), dim3(DYN_BLOCK), 0, s, P, ca,
                     d, b, position, cubic, rotation, (const float4 *)rot_poly, (const float4 *)rot_fourier, opacity,
                     scaling, extr, W, H, depth, radius, dL_duv, dL_ddepth, dL_dconic, dL_dopa, d_position, d_cubic,
                     d_rotation, d_opacity, d_scaling);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}


// ---- frame batch of dynamic Gaussians (SURVEY 8 rows a15 + f1): `tab` = F entries {int seg; float d; float basis[12];
// float pad[2]} (64 bytes each) in DEVICE memory, one per frame; cubic in either layout; outputs [F,P,..], opa_t [P].
extern "C" int splat_frame_preprocess_forward_batch(int F, int P, int I, const void *tab, const float *position,
                                                    const float *cubic, int cubic_layout, const float *rotation,
                                                    const float *rot_poly, const float *rot_fourier, const float *opacity,
                                                    const float *scaling, const float *extr, int W, int H, float nearest,
                                                    float extent, float *uv, float *depth, float *conic, int32_t *radius,
                                                    float *opa_t, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && F <= 65535 && P >= 1 && I >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR, "unknown cubic_layout");
    SPLAT_CHECK_ARG(tab && position && cubic && rotation && rot_poly && rot_fourier && opacity && scaling && extr, "null input pointer");
    SPLAT_CHECK_ARG(uv && depth && conic && radius && opa_t, "null output pointer");
    const dim3 g = dyn_grid(P);
    // frames per thread: as many as keep >= ~4096 workgroups in flight
    int slices = (int)((4096 + g.x - 1) / g.x);
    if (slices < 1) slices = 1;
    if (slices > F) slices = F;
    SPLAT_LAUNCH("frame_preprocess_fwd", frame_preprocess_fwd_batch_kernel, dim3(g.x, slices), dim3(DYN_BLOCK), 0, (hipStream_t)stream,
                 F, P, I, cubic_layout, (const DynTab *)tab, position, cubic, rotation, (const float4 *)rot_poly,
                 (const float4 *)rot_fourier, opacity, scaling, extr, W, H, nearest, extent, uv, depth, conic, radius, opa_t);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_frames_gauss_backward_dynamic(int F, int P, int I, int C, int W, int H, int64_t capacity, int want_abs,
                                                   const float *pair_records, const int32_t *goff_incl,
                                                   const int32_t *radius, const void *tab, const float *position,
                                                   const float *cubic, int cubic_layout, const float *rotation,
                                                   const float *rot_poly, const float *rot_fourier, const float *opacity,
                                                   const float *scaling, const float *extr, float *d_position,
                                                   float *d_cubic, float *d_rotation, float *d_opacity, float *d_scaling,
                                                   float *d_feature, float *tap, float *abs_tap, int32_t *radii_max,
                                                   void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && I >= 1 && C >= 1 && C <= 32 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 32)");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR, "unknown cubic_layout");
    SPLAT_CHECK_ARG(pair_records && goff_incl && tab && position && cubic && rotation && rot_poly && rot_fourier && opacity &&
                        scaling && extr,
                    "null input pointer");
    SPLAT_CHECK_ARG(!abs_tap || want_abs, "abs_tap needs records with the abs sums");
    SPLAT_CHECK_ARG(!radii_max || radius, "radii_max needs the per-frame radius");
    GaussDynArgs A;
    memset(&A, 0, sizeof(A));
    A.F = F; A.P = P; A.W = W; A.H = H; A.I = I; A.layout = cubic_layout; A.C = C; A.cn = C; A.cap = capacity;
    A.pair = pair_records; A.goff = goff_incl; A.radius = radius; A.tab = (const DynTab *)tab;
    A.position = position; A.cubic = cubic; A.rotation = rotation; A.rot_poly = (const float4 *)rot_poly;
    A.rot_fourier = (const float4 *)rot_fourier; A.opacity = opacity; A.scaling = scaling; A.extr = extr;
    A.d_position = d_position; A.d_cubic = d_cubic; A.d_rotation = d_rotation; A.d_opacity = d_opacity; A.d_scaling = d_scaling;
    A.d_feature = d_feature; A.tap = tap; A.abs_tap = abs_tap; A.radii_max = radii_max;
    const int ncp = (int)splat_blend_pair_stride(C, want_abs, 0);
    return want_abs ? launch_gauss_bwd_dynamic<true>(A, ncp, (hipStream_t)stream)
                    : launch_gauss_bwd_dynamic<false>(A, ncp, (hipStream_t)stream);
}

// Gaussian side of splat_alpha_blending_backward_batch_sets for DYNAMIC Gaussians (the reference's real training frame:
// dynamic_gaussian_with_base_point_cloud.py getters + render_iter's three blends): splat_frames_gauss_backward_dynamic with
// the SETS records -- taps from the tap set, per-set feature gradients (ADDED to set_dfeature[g], HOST array of three device
// pointers), the row channel `depth_channel` (>= 0) is the per-frame depth and feeds the position through the projection.
static int gauss_backward_dynamic_sets_impl(int F, int P, int I, int C, int W, int H, int64_t capacity,
                                                        const float *pair_records, const int32_t *goff_incl,
                                                        const int32_t *radius, const void *tab, const float *position,
                                                        const float *cubic, int cubic_layout, const float *rotation,
                                                        const float *rot_poly, const float *rot_fourier, const float *opacity,
                                                        const float *scaling, const float *extr, float *d_position,
                                                        float *d_cubic, float *d_rotation, float *d_opacity, float *d_scaling,
                                                        const int32_t *set_c0, const int32_t *set_cn,
                                                        float *const *set_dfeature, const int32_t *set_stride,
                                                        int depth_channel, float *tap, float *abs_tap, int32_t *radii_max,
                                                        void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && I >= 1 && C >= 1 && C <= 28 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 28)");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR, "unknown cubic_layout");
    SPLAT_CHECK_ARG(pair_records && goff_incl && tab && position && cubic && rotation && rot_poly && rot_fourier && opacity &&
                        scaling && extr,
                    "null input pointer");
    SPLAT_CHECK_ARG(set_c0 && set_cn && set_dfeature && set_stride, "null set table");
    SPLAT_CHECK_ARG(depth_channel < C, "depth_channel outside the row");
    SPLAT_CHECK_ARG(!radii_max || radius, "radii_max needs the per-frame radius");
    GaussDynArgs A;
    memset(&A, 0, sizeof(A));
    A.F = F; A.P = P; A.W = W; A.H = H; A.I = I; A.layout = cubic_layout; A.C = C; A.cn = C; A.cap = capacity;
    A.pair = pair_records; A.goff = goff_incl; A.radius = radius; A.tab = (const DynTab *)tab;
    A.position = position; A.cubic = cubic; A.rotation = rotation; A.rot_poly = (const float4 *)rot_poly;
    A.rot_fourier = (const float4 *)rot_fourier; A.opacity = opacity; A.scaling = scaling; A.extr = extr;
    A.d_position = d_position; A.d_cubic = d_cubic; A.d_rotation = d_rotation; A.d_opacity = d_opacity; A.d_scaling = d_scaling;
    A.tap = tap; A.abs_tap = abs_tap; A.radii_max = radii_max; A.depth_channel = depth_channel;
    A.nsrc = 3;
    for (int g = 0; g < 3; ++g) {
        A.sc0[g] = set_c0[g]; A.scn[g] = set_cn[g]; A.sstride[g] = set_stride[g]; A.sdf[g] = set_dfeature[g];
        SPLAT_CHECK_ARG(set_cn[g] == 0 || !set_dfeature[g] || set_stride[g] >= set_cn[g], "feature stride below the set width");
    }
    return launch_gauss_bwd_dynamic_sets(A, (int)splat_blend_sets_pair_stride(C), (hipStream_t)stream);
}

// the same with the row described by feature SOURCES (include/splat_hip.h: splat_feature_source_t): shared sources receive the
// sum over the frames, per-frame sources every frame's own gradient
extern "C" int splat_frames_gauss_backward_dynamic_sources(int F, int P, int I, int C, int W, int H, int64_t capacity,
                                                           const float *pair_records, const int32_t *goff_incl,
                                                           const int32_t *radius, const void *tab, const float *position,
                                                           const float *cubic, int cubic_layout, const float *rotation,
                                                           const float *rot_poly, const float *rot_fourier, const float *opacity,
                                                           const float *scaling, const float *extr, float *d_position,
                                                           float *d_cubic, float *d_rotation, float *d_opacity, float *d_scaling,
                                                           int nsrc, const splat_feature_source_t *src, int depth_channel,
                                                           float *tap, float *abs_tap, int32_t *radii_max, void *stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && I >= 1 && C >= 1 && C <= 28 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 28)");
    SPLAT_CHECK_ARG(cubic_layout == SPLAT_CUBIC_GAUSSIAN_MAJOR || cubic_layout == SPLAT_CUBIC_SEGMENT_MAJOR, "unknown cubic_layout");
    SPLAT_CHECK_ARG(pair_records && goff_incl && tab && position && cubic && rotation && rot_poly && rot_fourier && opacity &&
                        scaling && extr,
                    "null input pointer");
    SPLAT_CHECK_ARG(src && nsrc >= 0 && nsrc <= SPLAT_MAX_SOURCES, "null / oversized source table");
    SPLAT_CHECK_ARG(depth_channel < C, "depth_channel outside the row");
    SPLAT_CHECK_ARG(!radii_max || radius, "radii_max needs the per-frame radius");
    GaussDynArgs A;
    memset(&A, 0, sizeof(A));
    A.F = F; A.P = P; A.W = W; A.H = H; A.I = I; A.layout = cubic_layout; A.C = C; A.cn = C; A.cap = capacity;
    A.pair = pair_records; A.goff = goff_incl; A.radius = radius; A.tab = (const DynTab *)tab;
    A.position = position; A.cubic = cubic; A.rotation = rotation; A.rot_poly = (const float4 *)rot_poly;
    A.rot_fourier = (const float4 *)rot_fourier; A.opacity = opacity; A.scaling = scaling; A.extr = extr;
    A.d_position = d_position; A.d_cubic = d_cubic; A.d_rotation = d_rotation; A.d_opacity = d_opacity; A.d_scaling = d_scaling;
    A.tap = tap; A.abs_tap = abs_tap; A.radii_max = radii_max; A.depth_channel = depth_channel;
    A.nsrc = nsrc;
    for (int g = 0; g < nsrc; ++g) {
        SPLAT_CHECK_ARG(src[g].cn >= 0 && src[g].c0 >= 0 && src[g].c0 + src[g].cn <= C, "source outside the row");
        SPLAT_CHECK_ARG(src[g].frame_stride == 0 || !src[g].d_feature || src[g].frame_stride >= (int64_t)P * src[g].cn,
                        "frame stride of a per-frame source below P * cn");
        A.sc0[g] = src[g].c0; A.scn[g] = src[g].cn; A.sstride[g] = src[g].cn; A.sdf[g] = src[g].d_feature;
        A.sfs[g] = src[g].d_feature ? src[g].frame_stride : 0;
        if (A.sfs[g] != 0) ++A.npf;
        const int k0 = SETS_NG + src[g].c0, k1 = k0 + src[g].cn - 1;
        A.spfq[g] = (src[g].cn >= 1 && k0 / 4 == k1 / 4) ? k0 / 4 : -1;
    }
    return launch_gauss_bwd_dynamic_sets(A, (int)splat_blend_sets_pair_stride(C), (hipStream_t)stream);
}

#define DYN_SETS_PARAMS int F, int P, int I, int C, int W, int H, int64_t capacity, const float *pair_records, const int32_t *goff_incl, \
    const int32_t *radius, const void *tab, const float *position, const float *cubic, int cubic_layout, const float *rotation, \
    const float *rot_poly, const float *rot_fourier, const float *opacity, const float *scaling, const float *extr, float *d_position, \
    float *d_cubic, float *d_rotation, float *d_opacity, float *d_scaling, const int32_t *set_c0, const int32_t *set_cn, \
    float *const *set_dfeature, const int32_t *set_stride, int depth_channel, float *tap, float *abs_tap, int32_t *radii_max, void *stream
#define DYN_SETS_ARGS F, P, I, C, W, H, capacity, pair_records, goff_incl, radius, tab, position, cubic, cubic_layout, rotation, rot_poly, \
    rot_fourier, opacity, scaling, extr, d_position, d_cubic, d_rotation, d_opacity, d_scaling, set_c0, set_cn, set_dfeature, set_stride, \
    depth_channel, tap, abs_tap, radii_max, stream
extern "C" int splat_frames_gauss_backward_dynamic_sets(DYN_SETS_PARAMS) { return gauss_backward_dynamic_sets_impl(DYN_SETS_ARGS); }
