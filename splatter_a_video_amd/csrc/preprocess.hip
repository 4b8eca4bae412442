// Fused per-frame preprocess of the orthographic renderer: one pass over the Gaussians does what the reference's
// renderer spreads over ~80 eager kernels and three native operators per frame
// (src/pointrix/renderer/dptr_ortho_enhanced.py:282-310: project_point (ortho) -> compute_cov3d -> ewa_project (ortho)),
// forward and backward.  Intermediates (visible mask, cov3d, dL_dcov3d) never reach HBM; the backward can add its
// results straight into the caller's gradient buffers.  The arithmetic is the one-Gaussian code of pointwise_dev.h,
// i.e. the same functions the separate operators run.
#include "common.h"
#include "pointwise_dev.h"

namespace {

constexpr int PP_BLOCK = 256;
inline dim3 pp_grid(int P) { return dim3((unsigned)((P + PP_BLOCK - 1) / PP_BLOCK)); }

__global__ void __launch_bounds__(PP_BLOCK)
preprocess_ortho_fwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ offset,
                            const float *__restrict__ scales, const float4 *__restrict__ uquats,
                            const float *__restrict__ extr, int W, int H, float nearest, float extent,
                            float2 *__restrict__ uv, float *__restrict__ depth, float *__restrict__ conic,
                            int *__restrict__ radius, int *__restrict__ tiles) {
    const int i = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (i >= P) return;
    Cam c;
    load_cam(nullptr, extr, c);
    float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (offset) {
        p[0] += offset[3 * i]; p[1] += offset[3 * i + 1]; p[2] += offset[3 * i + 2];
    }
    float u, v, d;
    const bool cull = project_ortho_pt(c, p[0], p[1], p[2], W, H, nearest, extent, u, v, d);
    u = cull ? 0.f : u; v = cull ? 0.f : v; d = cull ? 0.f : d;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    int orad = 0, otiles = 0;
    if (d != 0.f) {  // the renderer's visibility mask (dptr_ortho_enhanced.py:288)
        const float4 q4 = uquats[i];
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        float c3[6], a[3], b[3], t[3], Jm[4], cov[3];
        cov3d_pt(s, q, c3);
        ewa_T<true>(c, p, W, H, a, b, t, Jm);
        ewa_cov2d<true>(a, b, c3, cov);
        ewa_finish_pt<true>(cov, make_float2(u, v), W, H, o0, o1, o2, orad, otiles);
    }
    uv[i] = make_float2(u, v);
    depth[i] = d;
    conic[3 * i] = o0; conic[3 * i + 1] = o1; conic[3 * i + 2] = o2;
    radius[i] = orad;
    tiles[i] = otiles;
}

template <bool ACC>
__device__ __forceinline__ void put(float *p, float v) {
    if (ACC) *p += v;
    else *p = v;
}

template <bool ACC>
__global__ void __launch_bounds__(PP_BLOCK)
preprocess_ortho_bwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ offset,
                            const float *__restrict__ scales, const float4 *__restrict__ uquats,
                            const float *__restrict__ extr, int W, int H, const float *__restrict__ depth,
                            const int *__restrict__ radius, const float *__restrict__ dL_duv,
                            const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dconic,
                            float *__restrict__ dL_dxyz, float *__restrict__ dL_dscales,
                            float *__restrict__ dL_duquats) {
    const int i = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (i >= P) return;
    float gp[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (depth[i] != 0.f) {
        Cam c;
        load_cam(nullptr, extr, c);
        if (dL_dxyz) project_ortho_grad_pt(c, W, H, dL_duv[2 * i], dL_duv[2 * i + 1], dL_ddepth ? dL_ddepth[i] : 0.f, gp);
        if (radius[i] > 0 && (dL_dscales || dL_duquats)) {
            float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
            if (offset) {
                p[0] += offset[3 * i]; p[1] += offset[3 * i + 1]; p[2] += offset[3 * i + 2];
            }
            const float4 q4 = uquats[i];
            const float q[4] = {q4.x, q4.y, q4.z, q4.w};
            const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
            float c3[6], a[3], b[3], t[3], Jm[4], cov[3];
            cov3d_pt(s, q, c3);
            ewa_T<true>(c, p, W, H, a, b, t, Jm);
            ewa_cov2d<true>(a, b, c3, cov);
            const float det = cov[0] * cov[2] - cov[1] * cov[1];
            if (det != 0.0f) {
                const float g3[3] = {dL_dconic[3 * i], dL_dconic[3 * i + 1], dL_dconic[3 * i + 2]};
                float dcx, dcy, dcz, g6[6];
                ewa_grad_cov_pt(a, b, cov, det, g3, dcx, dcy, dcz, g6);
                cov3d_grad_pt(s, q, g6, ds, dq);
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

}  // namespace

extern "C" int splat_preprocess_ortho_forward(int P, const float *xyz, const float *offset, const float *scales,
                                              const float *uquats, const float *extr, int W, int H, float nearest,
                                              float extent, float *uv, float *depth, float *conic, int32_t *radius,
                                              int32_t *tiles, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && scales && uquats && extr && uv && depth && conic && radius && tiles, "null pointer");
    SPLAT_LAUNCH("preprocess_fwd", preprocess_ortho_fwd_kernel, pp_grid(P), dim3(PP_BLOCK), 0, (hipStream_t)stream, P, xyz,
                 offset, scales, (const float4 *)uquats, extr, W, H, nearest, extent, (float2 *)uv, depth, conic, radius,
                 tiles);
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
        SPLAT_LAUNCH("preprocess_bwd", preprocess_ortho_bwd_kernel<true>, pp_grid(P), dim3(PP_BLOCK), 0, s, P, xyz, offset,
                     scales, (const float4 *)uquats, extr, W, H, depth, radius, dL_duv, dL_ddepth, dL_dconic, dL_dxyz,
                     dL_dscales, dL_duquats);
    else
        SPLAT_LAUNCH("preprocess_bwd", preprocess_ortho_bwd_kernel<false>, pp_grid(P), dim3(PP_BLOCK), 0, s, P, xyz, offset,
                     scales, (const float4 *)uquats, extr, W, H, depth, radius, dL_duv, dL_ddepth, dL_dconic, dL_dxyz,
                     dL_dscales, dL_duquats);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
