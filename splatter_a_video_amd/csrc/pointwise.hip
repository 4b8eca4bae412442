// Per-Gaussian streaming kernels: project_point, compute_cov3d, ewa_project, compute_sh
// (forward + backward, perspective and orthographic).  One thread per Gaussian, 256-thread
// blocks; HBM-bound (24..250 B per point).  Semantics follow the reference kernels cited in
// include/splat_hip.h; the arithmetic mirrors oracle/splat_oracle.c.
#include "common.h"
#include "pointwise_dev.h"

#define PW_BLOCK 256
static inline dim3 pw_grid(int P) { return dim3((unsigned)((P + PW_BLOCK - 1) / PW_BLOCK)); }

// ------------------------------------------------------------------ project_point
// reference: src/project_point.cu:13-57 ; ortho: dptr_ortho_enhanced.py:177-202
template <bool ORTHO>
__global__ void __launch_bounds__(PW_BLOCK)
project_point_fwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ intr,
                         const float *__restrict__ extr, int W, int H, float nearest, float extent,
                         float *__restrict__ uv, float *__restrict__ depth) {
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= P) return;
    Cam c;
    load_cam(ORTHO ? nullptr : intr, extr, c);
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float u, v, d;
    const bool cull = ORTHO ? project_ortho_pt(c, x, y, z, W, H, nearest, extent, u, v, d)
                            : project_persp_pt(c, x, y, z, W, H, nearest, extent, u, v, d);
    uv[2 * i] = cull ? 0.f : u;
    uv[2 * i + 1] = cull ? 0.f : v;
    depth[i] = cull ? 0.f : d;
}

// reference: src/project_point.cu:59-145 ; ortho: autograd of the twin.
template <bool ORTHO, bool CAMGRAD>
__global__ void __launch_bounds__(PW_BLOCK)
project_point_bwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ intr,
                         const float *__restrict__ extr, int W, int H, const float *__restrict__ depth,
                         const float *__restrict__ dL_duv, const float *__restrict__ dL_ddepth,
                         float *__restrict__ dL_dxyz, float *__restrict__ dL_dintr, float *__restrict__ dL_dextr) {
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    const bool live = (i < P) && (depth[i] != 0);
    float cg[16];  // camera-gradient contributions: intr[0..3], extr[0..11]
#pragma unroll
    for (int k = 0; k < 16; ++k) cg[k] = 0.f;
    if (live) {
        Cam c;
        load_cam(ORTHO ? nullptr : intr, extr, c);
        const float gu = dL_duv[2 * i], gv = dL_duv[2 * i + 1], gd = dL_ddepth[i];
        if (ORTHO) {
            float g[3];
            project_ortho_grad_pt(c, W, H, gu, gv, gd, g);
#pragma unroll
            for (int j = 0; j < 3; ++j) dL_dxyz[3 * i + j] = g[j];
        } else {
            const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            float tx, ty, tz;
            cam_xform(c, x, y, z, tx, ty, tz);
            const float n1 = (float)(1.0 / (double)tz);
            const float n2 = (float)(1.0 / (double)(tz * tz));
            float g[3];
            project_persp_grad_pt(c, x, y, z, gu, gv, gd, g);
#pragma unroll
            for (int j = 0; j < 3; ++j) dL_dxyz[3 * i + j] = g[j];
            if (CAMGRAD) {
                cg[0] = tx * n1 * gu; cg[1] = ty * n1 * gv; cg[2] = gu; cg[3] = gv;
                const float p[4] = {x, y, z, 1.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cg[4 + j] = c.fx * p[j] * n1 * gu;
                    cg[8 + j] = c.fy * p[j] * n1 * gv;
                    cg[12 + j] = -c.fx * p[j] * tx * n2 * gu - c.fy * p[j] * ty * n2 * gv + p[j] * gd;
                }
            }
        }
    }
    if (!live && i < P) {
        dL_dxyz[3 * i] = 0.f; dL_dxyz[3 * i + 1] = 0.f; dL_dxyz[3 * i + 2] = 0.f;
    }
    if (CAMGRAD && !ORTHO) {  // whole wave participates: reduce, then one atomic per wave and component
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float s = wave_sum_to_lane63(cg[k]);
            if (lane == 63 && s != 0.f) {
                if (k < 4) { if (dL_dintr) atomic_add_f32(dL_dintr + k, s); }
                else if (dL_dextr) atomic_add_f32(dL_dextr + (k - 4), s);
            }
        }
    }
}

// ------------------------------------------------------------------ compute_cov3d
// reference: src/compute_cov3d.cu:14-58,119-129
__global__ void __launch_bounds__(PW_BLOCK)
cov3d_fwd_kernel(int P, const float *__restrict__ scales, const float4 *__restrict__ uquats,
                 const uint8_t *__restrict__ visible, float *__restrict__ cov3d) {
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= P) return;
    float *o = cov3d + 6 * (size_t)i;
    if (!visible[i]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = 0.f;
        return;
    }
    const float4 q4 = uquats[i];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    float c6[6];
    cov3d_pt(s, q, c6);
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = c6[k];
}

// reference: src/compute_cov3d.cu:60-117,131-147
__global__ void __launch_bounds__(PW_BLOCK)
cov3d_bwd_kernel(int P, const float *__restrict__ scales, const float4 *__restrict__ uquats,
                 const uint8_t *__restrict__ visible, const float *__restrict__ dL_dcov3d,
                 float *__restrict__ dL_dscales, float4 *__restrict__ dL_duquats) {
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= P) return;
    if (!visible[i]) {
        dL_dscales[3 * i] = 0.f; dL_dscales[3 * i + 1] = 0.f; dL_dscales[3 * i + 2] = 0.f;
        dL_duquats[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 q4 = uquats[i];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    float g[6], ds[3], dq[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = dL_dcov3d[6 * (size_t)i + k];
    cov3d_grad_pt(s, q, g, ds, dq);
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = ds[k];
    dL_duquats[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
}

// ------------------------------------------------------------------ ewa_project
// reference: src/ewa_project.cu:16-83 ; ortho: dptr_ortho_enhanced.py:18-111
// Every element of conic / radius / tiles is written (zeros for culled / degenerate splats).
template <bool ORTHO>
__global__ void __launch_bounds__(PW_BLOCK)
ewa_fwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ cov3d, const float *__restrict__ intr,
               const float *__restrict__ extr, const float2 *__restrict__ uv, int W, int H,
               const uint8_t *__restrict__ visible, float *__restrict__ conic, int *__restrict__ radius,
               int *__restrict__ tiles) {
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= P) return;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    int orad = 0, otiles = 0;
    if (visible[i]) {
        Cam c;
        load_cam(ORTHO ? nullptr : intr, extr, c);
        const float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        float c3[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c3[k] = cov3d[6 * (size_t)i + k];
        float a[3], b[3], t[3], Jm[4], cov[3];
        ewa_T<ORTHO>(c, p, W, H, a, b, t, Jm);
        ewa_cov2d<ORTHO>(a, b, c3, cov);
        ewa_finish_pt<ORTHO>(cov, uv[i], W, H, o0, o1, o2, orad, otiles);
    }
    conic[3 * i] = o0; conic[3 * i + 1] = o1; conic[3 * i + 2] = o2;
    radius[i] = orad;
    tiles[i] = otiles;
}

// reference: src/ewa_project.cu:85-252
template <bool ORTHO, bool CAMGRAD>
__global__ void __launch_bounds__(PW_BLOCK)
ewa_bwd_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ cov3d, const float *__restrict__ intr,
               const float *__restrict__ extr, int W, int H, const int *__restrict__ radius,
               const float *__restrict__ dL_dconic, float *__restrict__ dL_dxyz, float *__restrict__ dL_dcov3d,
               float *__restrict__ dL_dintr, float *__restrict__ dL_dextr) {
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    float cg[14];  // intr[0..1], extr[0..11]
#pragma unroll
    for (int k = 0; k < 14; ++k) cg[k] = 0.f;
    bool live = (i < P) && (radius[i] > 0);
    bool wrote = false;   // dL_dcov3d row written
    bool wrote_xyz = false;
    if (live) {
        Cam c;
        load_cam(ORTHO ? nullptr : intr, extr, c);
        const float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        float c3[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c3[k] = cov3d[6 * (size_t)i + k];
        float a[3], b[3], t[3], Jm[4], cov[3];
        ewa_T<ORTHO>(c, p, W, H, a, b, t, Jm);
        ewa_cov2d<ORTHO>(a, b, c3, cov);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det != 0.0f) {
            const float g3[3] = {dL_dconic[3 * i], dL_dconic[3 * i + 1], dL_dconic[3 * i + 2]};
            float dcx, dcy, dcz, o6[6];
            ewa_grad_cov_pt(a, b, cov, det, g3, dcx, dcy, dcz, o6);
            float *o = dL_dcov3d + 6 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = o6[k];
            wrote = true;
            if (!ORTHO) {
                float da[3], db[3], dt[3], gpos[3];
                ewa_grad_pos_persp_pt(c, t, a, b, c3, dcx, dcy, dcz, gpos, da, db, dt);
                const float dtx = dt[0], dty = dt[1], dtz = dt[2];
                const float dJ00 = c.e[0] * da[0] + c.e[1] * da[1] + c.e[2] * da[2];
                const float dJ02 = c.e[8] * da[0] + c.e[9] * da[1] + c.e[10] * da[2];
                const float dJ11 = c.e[4] * db[0] + c.e[5] * db[1] + c.e[6] * db[2];
                const float dJ12 = c.e[8] * db[0] + c.e[9] * db[1] + c.e[10] * db[2];
                const float tz = 1.f / t[2], tz2 = tz * tz;
                dL_dxyz[3 * i + 0] = gpos[0];
                dL_dxyz[3 * i + 1] = gpos[1];
                dL_dxyz[3 * i + 2] = gpos[2];
                wrote_xyz = true;
                if (CAMGRAD) {
                    cg[0] = tz * dJ00 - t[0] * tz2 * dJ02;
                    cg[1] = tz * dJ11 - t[1] * tz2 * dJ12;
                    const float pp[4] = {p[0], p[1], p[2], 1.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cg[2 + k] = pp[k] * dtx + (k < 3 ? Jm[0] * da[k] : 0.f);
                        cg[6 + k] = pp[k] * dty + (k < 3 ? Jm[1] * db[k] : 0.f);
                        cg[10 + k] = pp[k] * dtz + (k < 3 ? Jm[2] * da[k] + Jm[3] * db[k] : 0.f);
                    }
                }
            }
        }
    }
    if (i < P) {  // every element is written: zeros where nothing flows
        if (!wrote) {
            float *o = dL_dcov3d + 6 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = 0.f;
        }
        if (!wrote_xyz) {
            dL_dxyz[3 * i] = 0.f; dL_dxyz[3 * i + 1] = 0.f; dL_dxyz[3 * i + 2] = 0.f;
        }
    }
    if (CAMGRAD && !ORTHO) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            const float s = wave_sum_to_lane63(cg[k]);
            if (lane == 63 && s != 0.f) {
                if (k < 2) { if (dL_dintr) atomic_add_f32(dL_dintr + k, s); }
                else if (dL_dextr) atomic_add_f32(dL_dextr + (k - 2), s);
            }
        }
    }
}

// ------------------------------------------------------------------ compute_sh
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float B[16]) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                         0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    B[0] = C0;
    if (deg < 1) return;
    B[1] = -C1 * y; B[2] = C1 * z; B[3] = -C1 * x;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = C2[0] * xy; B[5] = C2[1] * yz; B[6] = C2[2] * (2.0f * zz - xx - yy); B[7] = C2[3] * xz; B[8] = C2[4] * (xx - yy);
    if (deg < 3) return;
    B[9] = C3[0] * y * (3.0f * xx - yy);
    B[10] = C3[1] * xy * z;
    B[11] = C3[2] * y * (4.0f * zz - xx - yy);
    B[12] = C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    B[13] = C3[4] * x * (4.0f * zz - xx - yy);
    B[14] = C3[5] * z * (xx - yy);
    B[15] = C3[6] * x * (xx - 3.0f * yy);
}

// reference: src/compute_sh.cu:32-80 (free variant: compute_sh_free.cu)
template <int DEG, bool FREE>
__global__ void __launch_bounds__(PW_BLOCK)
sh_fwd_kernel(int P, const float *__restrict__ shs, const float *__restrict__ dirs,
              const uint8_t *__restrict__ visible, float *__restrict__ colors, uint8_t *__restrict__ clamped) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= P) return;
    if (!visible[i]) {  // reference: colors zero-init, clamped ones-init
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            colors[3 * i + c] = 0.f;
            if (!FREE) clamped[3 * i + c] = 1;
        }
        return;
    }
    float B[16];
    sh_basis(DEG, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], B);
    const float *sh = shs + (size_t)i * NB * 3;
    float r[3] = {B[0] * sh[0], B[0] * sh[1], B[0] * sh[2]};
#pragma unroll
    for (int k = 1; k < NB; ++k) {
        r[0] = r[0] + B[k] * sh[3 * k + 0];
        r[1] = r[1] + B[k] * sh[3 * k + 1];
        r[2] = r[2] + B[k] * sh[3 * k + 2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (FREE) {
            colors[3 * i + c] = r[c];
        } else {
            const float v = r[c] + 0.5f;
            clamped[3 * i + c] = (v < 0);
            colors[3 * i + c] = v > 0.f ? v : 0.f;
        }
    }
}

// reference: src/compute_sh.cu:82-195
// dL/dshs[i][k][c] = B_k(dir_i) * g_i[c] is a rank-1 block of (DEG+1)^2 x 3 floats per Gaussian.  Phase 1 (thread per
// Gaussian) puts the basis values and the masked colour gradient into LDS, phase 2 streams the workgroup's
// contiguous output range with consecutive lanes on consecutive addresses (float4 when the row length allows),
// storing or -- gradient-bucket use -- adding.
template <int DEG, bool FREE, bool ACC>
__global__ void __launch_bounds__(PW_BLOCK)
sh_bwd_kernel(int P, const float *__restrict__ shs, const float *__restrict__ dirs,
              const uint8_t *__restrict__ visible, const uint8_t *__restrict__ clamped,
              const float *__restrict__ dL_dcolors, float *__restrict__ dL_dshs, float *__restrict__ dL_ddirs) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NB3 = NB * 3;
    __shared__ float sB[PW_BLOCK][NB + 1];
    __shared__ float sg[PW_BLOCK][3];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * PW_BLOCK;
    const int i = i0 + tid;
    const int nG = imin_(PW_BLOCK, P - i0);
    if (i < P) {
        float B[16], g[3] = {0.f, 0.f, 0.f};
        const bool vis = visible[i] != 0;
        const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        sh_basis(DEG, x, y, z, B);
        if (vis) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g[c] = dL_dcolors[3 * i + c];
                if (!FREE && clamped[3 * i + c]) g[c] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) sB[tid][k] = B[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) sg[tid][c] = g[c];   // zero for invisible points: their rows become zero / unchanged
        if (dL_ddirs) {  // direction gradient (reads the coefficients; skipped when not requested)
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
            if (vis && DEG > 0) {
                const float C1 = 0.4886025119029199f;
                const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                     0.5462742152960396f};
                const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                     -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
                const float *sh = shs + (size_t)i * NB * 3;
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#define SHC(k) sh[3 * (k) + c]
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float dx = -C1 * SHC(3), dy = -C1 * SHC(1), dz = C1 * SHC(2);
                    if (DEG > 1) {
                        dx += C2[0] * y * SHC(4) + C2[2] * 2.f * -x * SHC(6) + C2[3] * z * SHC(7) + C2[4] * 2.f * x * SHC(8);
                        dy += C2[0] * x * SHC(4) + C2[1] * z * SHC(5) + C2[2] * 2.f * -y * SHC(6) + C2[4] * 2.f * -y * SHC(8);
                        dz += C2[1] * y * SHC(5) + C2[2] * 2.f * 2.f * z * SHC(6) + C2[3] * x * SHC(7);
                    }
                    if (DEG > 2) {
                        dx += (C3[0] * SHC(9) * 3.f * 2.f * xy + C3[1] * SHC(10) * yz + C3[2] * SHC(11) * -2.f * xy +
                               C3[3] * SHC(12) * -3.f * 2.f * xz + C3[4] * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                               C3[5] * SHC(14) * 2.f * xz + C3[6] * SHC(15) * 3.f * (xx - yy));
                        dy += (C3[0] * SHC(9) * 3.f * (xx - yy) + C3[1] * SHC(10) * xz + C3[2] * SHC(11) * (-3.f * yy + 4.f * zz - xx) +
                               C3[3] * SHC(12) * -3.f * 2.f * yz + C3[4] * SHC(13) * -2.f * xy + C3[5] * SHC(14) * -2.f * yz +
                               C3[6] * SHC(15) * -3.f * 2.f * xy);
                        dz += (C3[1] * SHC(10) * xy + C3[2] * SHC(11) * 4.f * 2.f * yz + C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) +
                               C3[4] * SHC(13) * 4.f * 2.f * xz + C3[5] * SHC(14) * (xx - yy));
                    }
                    ddx += dx * g[c]; ddy += dy * g[c]; ddz += dz * g[c];
                }
#undef SHC
            }
            dL_ddirs[3 * i] = ddx; dL_ddirs[3 * i + 1] = ddy; dL_ddirs[3 * i + 2] = ddz;
        }
    }
    __syncthreads();
    float *out = dL_dshs + (size_t)i0 * NB3;
    if (NB3 % 4 == 0) {
        constexpr int Q = NB3 / 4;  // float4 chunks per Gaussian
        float4 *out4 = reinterpret_cast<float4 *>(out);
        for (int e = tid; e < nG * Q; e += PW_BLOCK) {
            const int gi = e / Q, j = e - gi * Q;
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int idx = 4 * j + t, k = idx / 3, c = idx - 3 * k;
                v[t] = sB[gi][k] * sg[gi][c];
            }
            float4 o = ACC ? out4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
            out4[e] = o;
        }
    } else {
        for (int e = tid; e < nG * NB3; e += PW_BLOCK) {
            const int gi = e / NB3, r = e - gi * NB3, k = r / 3, c = r - 3 * k;
            const float v = sB[gi][k] * sg[gi][c];
            out[e] = ACC ? out[e] + v : v;
        }
    }
}

// ================================================================== C ABI
extern "C" int splat_project_point_forward(int P, const float *xyz, const float *intr, const float *extr, int W, int H,
                                           float nearest, float extent, int ortho, float *uv, float *depth,
                                           splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && extr && uv && depth && (ortho || intr), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (ortho)
        SPLAT_LAUNCH("project_point_fwd", project_point_fwd_kernel<true>, pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, intr, extr, W, H, nearest, extent, uv, depth);
    else
        SPLAT_LAUNCH("project_point_fwd", project_point_fwd_kernel<false>, pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, intr, extr, W, H, nearest, extent, uv, depth);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_project_point_backward(int P, const float *xyz, const float *intr, const float *extr, int W, int H,
                                            int ortho, const float *depth, const float *dL_duv, const float *dL_ddepth,
                                            float *dL_dxyz, float *dL_dintr, float *dL_dextr, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && extr && depth && dL_duv && dL_ddepth && dL_dxyz && (ortho || intr), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool cam = (dL_dintr != nullptr) || (dL_dextr != nullptr);
    if (ortho)
        SPLAT_LAUNCH("project_point_bwd", (project_point_bwd_kernel<true, false>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, intr, extr, W, H, depth, dL_duv, dL_ddepth, dL_dxyz, dL_dintr, dL_dextr);
    else if (cam)
        SPLAT_LAUNCH("project_point_bwd", (project_point_bwd_kernel<false, true>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, intr, extr, W, H, depth, dL_duv, dL_ddepth, dL_dxyz, dL_dintr, dL_dextr);
    else
        SPLAT_LAUNCH("project_point_bwd", (project_point_bwd_kernel<false, false>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, intr, extr, W, H, depth, dL_duv, dL_ddepth, dL_dxyz, dL_dintr, dL_dextr);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_compute_cov3d_forward(int P, const float *scales, const float *uquats, const uint8_t *visible,
                                           float *cov3d, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(scales && uquats && visible && cov3d, "null pointer");
    SPLAT_LAUNCH("cov3d_fwd", cov3d_fwd_kernel, pw_grid(P), dim3(PW_BLOCK), 0, (hipStream_t)stream, P, scales, (const float4 *)uquats, visible, cov3d);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_compute_cov3d_backward(int P, const float *scales, const float *uquats, const uint8_t *visible,
                                            const float *dL_dcov3d, float *dL_dscales, float *dL_duquats,
                                            splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(scales && uquats && visible && dL_dcov3d && dL_dscales && dL_duquats, "null pointer");
    SPLAT_LAUNCH("cov3d_bwd", cov3d_bwd_kernel, pw_grid(P), dim3(PW_BLOCK), 0, (hipStream_t)stream, P, scales, (const float4 *)uquats, visible, dL_dcov3d, dL_dscales, (float4 *)dL_duquats);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_ewa_project_forward(int P, const float *xyz, const float *cov3d, const float *intr,
                                         const float *extr, const float *uv, int W, int H, const uint8_t *visible,
                                         int ortho, float *conic, int32_t *radius, int32_t *tiles,
                                         splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && cov3d && extr && uv && visible && conic && radius && tiles && (ortho || intr), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (ortho)
        SPLAT_LAUNCH("ewa_fwd", ewa_fwd_kernel<true>, pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, cov3d, intr, extr, (const float2 *)uv, W, H, visible, conic, radius, tiles);
    else
        SPLAT_LAUNCH("ewa_fwd", ewa_fwd_kernel<false>, pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, cov3d, intr, extr, (const float2 *)uv, W, H, visible, conic, radius, tiles);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_ewa_project_backward(int P, const float *xyz, const float *cov3d, const float *intr,
                                          const float *extr, int W, int H, int ortho, const int32_t *radius,
                                          const float *dL_dconic, float *dL_dxyz, float *dL_dcov3d, float *dL_dintr,
                                          float *dL_dextr, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(xyz && cov3d && extr && radius && dL_dconic && dL_dxyz && dL_dcov3d && (ortho || intr), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool cam = (dL_dintr != nullptr) || (dL_dextr != nullptr);
    if (ortho)
        SPLAT_LAUNCH("ewa_bwd", (ewa_bwd_kernel<true, false>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, cov3d, intr, extr, W, H, radius, dL_dconic, dL_dxyz, dL_dcov3d, dL_dintr, dL_dextr);
    else if (cam)
        SPLAT_LAUNCH("ewa_bwd", (ewa_bwd_kernel<false, true>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, cov3d, intr, extr, W, H, radius, dL_dconic, dL_dxyz, dL_dcov3d, dL_dintr, dL_dextr);
    else
        SPLAT_LAUNCH("ewa_bwd", (ewa_bwd_kernel<false, false>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, xyz, cov3d, intr, extr, W, H, radius, dL_dconic, dL_dxyz, dL_dcov3d, dL_dintr, dL_dextr);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <bool FREE>
static int sh_fwd_dispatch(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                           float *colors, uint8_t *clamped, hipStream_t s) {
    switch (degree) {
        case 0: SPLAT_LAUNCH("sh_fwd", (sh_fwd_kernel<0, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, colors, clamped); break;
        case 1: SPLAT_LAUNCH("sh_fwd", (sh_fwd_kernel<1, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, colors, clamped); break;
        case 2: SPLAT_LAUNCH("sh_fwd", (sh_fwd_kernel<2, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, colors, clamped); break;
        default: SPLAT_LAUNCH("sh_fwd", (sh_fwd_kernel<3, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, colors, clamped); break;
    }
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <bool FREE, bool ACC>
static int sh_bwd_dispatch(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                           const uint8_t *clamped, const float *g, float *dshs, float *ddirs, hipStream_t s) {
    switch (degree) {
        case 0: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<0, FREE, ACC>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
        case 1: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<1, FREE, ACC>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
        case 2: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<2, FREE, ACC>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
        default: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<3, FREE, ACC>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
    }
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_compute_sh_forward(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                                        int free_variant, float *colors, uint8_t *clamped, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && degree >= 0 && degree <= 3, "degree must be 0..3");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(shs && dirs && visible && colors && (free_variant || clamped), "null pointer");
    return free_variant ? sh_fwd_dispatch<true>(P, shs, degree, dirs, visible, colors, clamped, (hipStream_t)stream)
                        : sh_fwd_dispatch<false>(P, shs, degree, dirs, visible, colors, clamped, (hipStream_t)stream);
}

extern "C" int splat_compute_sh_backward(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                                         const uint8_t *clamped, int free_variant, const float *dL_dcolors,
                                         int accumulate, float *dL_dshs, float *dL_ddirs, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && degree >= 0 && degree <= 3, "degree must be 0..3");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(shs && dirs && visible && dL_dcolors && dL_dshs && (free_variant || clamped), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (accumulate)
        return free_variant ? sh_bwd_dispatch<true, true>(P, shs, degree, dirs, visible, clamped, dL_dcolors, dL_dshs, dL_ddirs, s)
                            : sh_bwd_dispatch<false, true>(P, shs, degree, dirs, visible, clamped, dL_dcolors, dL_dshs, dL_ddirs, s);
    return free_variant ? sh_bwd_dispatch<true, false>(P, shs, degree, dirs, visible, clamped, dL_dcolors, dL_dshs, dL_ddirs, s)
                        : sh_bwd_dispatch<false, false>(P, shs, degree, dirs, visible, clamped, dL_dcolors, dL_dshs, dL_ddirs, s);
}
