// Per-Gaussian streaming kernels: project_point, compute_cov3d, ewa_project, compute_sh
// (forward + backward, perspective and orthographic).  One thread per Gaussian, 256-thread
// blocks; HBM-bound (24..250 B per point).  Semantics follow the reference kernels cited in
// include/splat_hip.h; the arithmetic mirrors oracle/splat_oracle.c.
#include "common.h"

#define PW_BLOCK 256
static inline dim3 pw_grid(int P) { return dim3((unsigned)((P + PW_BLOCK - 1) / PW_BLOCK)); }

struct Cam {
    float e[12];  // extr rows
    float fx, fy, cx, cy;
};


// camera constants are read once per thread from global (L2/K$ resident, 16 floats)
__device__ __forceinline__ void load_cam(const float *intr, const float *extr, Cam &c) {
#pragma unroll
    for (int k = 0; k < 12; ++k) c.e[k] = extr[k];
    if (intr) {
        c.fx = intr[0]; c.fy = intr[1]; c.cx = intr[2]; c.cy = intr[3];
    } else {
        c.fx = c.fy = c.cx = c.cy = 0.f;
    }
}

__device__ __forceinline__ void cam_xform(const Cam &c, float x, float y, float z, float &tx, float &ty, float &tz) {
    tx = c.e[0] * x + c.e[1] * y + c.e[2] * z + c.e[3];
    ty = c.e[4] * x + c.e[5] * y + c.e[6] * z + c.e[7];
    tz = c.e[8] * x + c.e[9] * y + c.e[10] * z + c.e[11];
}

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
    float tx, ty, tz;
    cam_xform(c, x, y, z, tx, ty, tz);
    float u, v, d;
    bool cull = false;
    if (ORTHO) {
        u = ((tx + 1.f) * (float)W) / 2.f - 0.5f;
        v = ((ty + 1.f) * (float)H) / 2.f - 0.5f;
        d = tz;
        if (isnan(d)) d = 0.f;
        else if (isinf(d)) d = d > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
        const float xlo = (float)((1.0 - (double)extent) * W * 0.5), xhi = (float)((1.0 + (double)extent) * W * 0.5);
        const float ylo = (float)((1.0 - (double)extent) * H * 0.5), yhi = (float)((1.0 + (double)extent) * H * 0.5);
        cull = (d <= nearest) || (u < xlo) || (u > xhi) || (v < ylo) || (v > yhi);
    } else {
        const float inv = (float)(1.0 / ((double)tz + 1e-7));
        u = (float)((double)(c.fx * tx * inv + c.cx) - 0.5);
        v = (float)((double)(c.fy * ty * inv + c.cy) - 0.5);
        d = tz;
        if (nearest > 0) cull = cull || (tz <= nearest);
        if (extent > 0) {
            const float xlo = (float)((double)((1 - extent) * W) * 0.5), xhi = (float)((double)((1 + extent) * W) * 0.5);
            const float ylo = (float)((double)((1 - extent) * H) * 0.5), yhi = (float)((double)((1 + extent) * H) * 0.5);
            cull = cull || (u < xlo) || (u > xhi) || (v < ylo) || (v > yhi);
        }
    }
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
            const float gx = gu * ((float)W / 2.f), gy = gv * ((float)H / 2.f);
#pragma unroll
            for (int j = 0; j < 3; ++j) dL_dxyz[3 * i + j] = c.e[j] * gx + c.e[4 + j] * gy + c.e[8 + j] * gd;
        } else {
            const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            float tx, ty, tz;
            cam_xform(c, x, y, z, tx, ty, tz);
            const float n1 = (float)(1.0 / (double)tz);
            const float n2 = (float)(1.0 / (double)(tz * tz));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float g = 0.f;
                g += (c.fx * (c.e[j] * tz - tx * c.e[8 + j]) * n2) * gu;
                g += (c.fy * (c.e[4 + j] * tz - ty * c.e[8 + j]) * n2) * gv;
                g += c.e[8 + j] * gd;
                dL_dxyz[3 * i + j] = g;
            }
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
__device__ __forceinline__ void quat_R(const float *q, float R[3][3]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

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
    float R[3][3], M[3][3];
    quat_R(q, R);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[k][j] = s[k] * R[j][k];
    int n = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b) o[n++] = M[0][a] * M[0][b] + M[1][a] * M[1][b] + M[2][a] * M[2][b];
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
    const float *g = dL_dcov3d + 6 * (size_t)i;
    float R[3][3], M[3][3];
    quat_R(q, R);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[k][j] = s[k] * R[j][k];
    const float G[3][3] = {{g[0], 0.5f * g[1], 0.5f * g[2]}, {0.5f * g[1], g[3], 0.5f * g[4]}, {0.5f * g[2], 0.5f * g[4], g[5]}};
    float dM[3][3], D[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) dM[a][b] = 2.0f * (M[a][0] * G[0][b] + M[a][1] * G[1][b] + M[a][2] * G[2][b]);
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) D[a][b] = s[a] * dM[a][b];
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float4 o;
    o.x = 2 * z * (D[0][1] - D[1][0]) + 2 * y * (D[2][0] - D[0][2]) + 2 * x * (D[1][2] - D[2][1]);
    o.y = 2 * y * (D[1][0] + D[0][1]) + 2 * z * (D[2][0] + D[0][2]) + 2 * r * (D[1][2] - D[2][1]) - 4 * x * (D[2][2] + D[1][1]);
    o.z = 2 * x * (D[1][0] + D[0][1]) + 2 * r * (D[2][0] - D[0][2]) + 2 * z * (D[1][2] + D[2][1]) - 4 * y * (D[2][2] + D[0][0]);
    o.w = 2 * r * (D[0][1] - D[1][0]) + 2 * x * (D[2][0] + D[0][2]) + 2 * y * (D[1][2] + D[2][1]) - 4 * z * (D[1][1] + D[0][0]);
    dL_duquats[i] = o;
}

// ------------------------------------------------------------------ ewa_project
template <bool ORTHO>
__device__ __forceinline__ void ewa_T(const Cam &c, const float p[3], int W, int H, float a[3], float b[3],
                                      float t[3], float Jm[4]) {
    cam_xform(c, p[0], p[1], p[2], t[0], t[1], t[2]);
    float J00, J11, J02, J12;
    if (ORTHO) {
        J00 = (float)W / 2.f; J11 = (float)H / 2.f; J02 = 0.f; J12 = 0.f;
    } else {
        J00 = c.fx / t[2]; J11 = c.fy / t[2];
        J02 = -(c.fx * t[0]) / (t[2] * t[2]);
        J12 = -(c.fy * t[1]) / (t[2] * t[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = J00 * c.e[k] + 0.0f * c.e[4 + k] + J02 * c.e[8 + k];
        b[k] = 0.0f * c.e[k] + J11 * c.e[4 + k] + J12 * c.e[8 + k];
    }
    Jm[0] = J00; Jm[1] = J11; Jm[2] = J02; Jm[3] = J12;
}

template <bool ORTHO>
__device__ __forceinline__ void ewa_cov2d(const float a[3], const float b[3], const float c3[6], float cov[3]) {
    const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float Xa[3], Xb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Xa[c] = a[0] * S[0][c] + a[1] * S[1][c] + a[2] * S[2][c];
        Xb[c] = b[0] * S[0][c] + b[1] * S[1][c] + b[2] * S[2][c];
    }
    cov[0] = (Xa[0] * a[0] + Xa[1] * a[1] + Xa[2] * a[2]) + 0.3f;
    cov[1] = ORTHO ? (Xa[0] * b[0] + Xa[1] * b[1] + Xa[2] * b[2]) : (Xb[0] * a[0] + Xb[1] * a[1] + Xb[2] * a[2]);
    cov[2] = (Xb[0] * b[0] + Xb[1] * b[1] + Xb[2] * b[2]) + 0.3f;
}

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
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        const bool bad = (det == 0.0f) || (ORTHO && isnan(det));
        if (!bad) {
            const float mid = 0.5f * (cov[0] + cov[2]);
            const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float l1 = mid + sq, l2 = mid - sq;
            const int r = (int)ceilf(3.f * sqrtf(fmaxf(l1, l2)));
            const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
            const float2 q = uv[i];
            int x0, y0, x1, y1;
            tile_rect(q.x, q.y, r, gx, gy, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) != 0) {
                if (ORTHO) {
                    o0 = cov[2] / det; o1 = -cov[1] / det; o2 = cov[0] / det;
                } else {
                    const float di = 1.f / det;
                    o0 = cov[2] * di; o1 = -cov[1] * di; o2 = cov[0] * di;
                }
                orad = r;
                otiles = (y1 - y0) * (x1 - x0);
            }
        }
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
            const float nom = 1.0f / (det * det);
            const float gx_ = dL_dconic[3 * i], gy_ = dL_dconic[3 * i + 1], gz_ = dL_dconic[3 * i + 2];
            const float dcx = nom * (-cov[2] * cov[2] * gx_ + cov[1] * cov[2] * gy_ + (det - cov[0] * cov[2]) * gz_);
            const float dcy = nom * (2 * cov[1] * cov[2] * gx_ - (det + 2 * cov[1] * cov[1]) * gy_ + 2 * cov[0] * cov[1] * gz_);
            const float dcz = nom * ((det - cov[0] * cov[2]) * gx_ + cov[0] * cov[1] * gy_ - cov[0] * cov[0] * gz_);
            float *o = dL_dcov3d + 6 * (size_t)i;
            o[0] = a[0] * a[0] * dcx + a[0] * b[0] * dcy + b[0] * b[0] * dcz;
            o[1] = 2 * a[0] * a[1] * dcx + (a[0] * b[1] + b[0] * a[1]) * dcy + 2 * b[0] * b[1] * dcz;
            o[2] = 2 * a[0] * a[2] * dcx + (a[0] * b[2] + b[0] * a[2]) * dcy + 2 * b[0] * b[2] * dcz;
            o[3] = a[1] * a[1] * dcx + a[1] * b[1] * dcy + b[1] * b[1] * dcz;
            o[4] = 2 * a[1] * a[2] * dcx + (a[1] * b[2] + b[1] * a[2]) * dcy + 2 * b[1] * b[2] * dcz;
            o[5] = a[2] * a[2] * dcx + a[2] * b[2] * dcy + b[2] * b[2] * dcz;
            wrote = true;
            if (!ORTHO) {
                const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
                float da[3], db[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float Sa = a[0] * S[0][k] + a[1] * S[1][k] + a[2] * S[2][k];
                    const float Sb = b[0] * S[0][k] + b[1] * S[1][k] + b[2] * S[2][k];
                    da[k] = 2 * Sa * dcx + Sb * dcy;
                    db[k] = Sa * dcy + 2 * Sb * dcz;
                }
                const float dJ00 = c.e[0] * da[0] + c.e[1] * da[1] + c.e[2] * da[2];
                const float dJ02 = c.e[8] * da[0] + c.e[9] * da[1] + c.e[10] * da[2];
                const float dJ11 = c.e[4] * db[0] + c.e[5] * db[1] + c.e[6] * db[2];
                const float dJ12 = c.e[8] * db[0] + c.e[9] * db[1] + c.e[10] * db[2];
                const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
                const float dtx = -c.fx * tz2 * dJ02;
                const float dty = -c.fy * tz2 * dJ12;
                const float dtz = -c.fx * tz2 * dJ00 - c.fy * tz2 * dJ11 + (2 * c.fx * t[0]) * tz3 * dJ02 +
                                  (2 * c.fy * t[1]) * tz3 * dJ12;
                dL_dxyz[3 * i + 0] = c.e[0] * dtx + c.e[4] * dty + c.e[8] * dtz;
                dL_dxyz[3 * i + 1] = c.e[1] * dtx + c.e[5] * dty + c.e[9] * dtz;
                dL_dxyz[3 * i + 2] = c.e[2] * dtx + c.e[6] * dty + c.e[10] * dtz;
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
template <int DEG, bool FREE>
__global__ void __launch_bounds__(PW_BLOCK)
sh_bwd_kernel(int P, const float *__restrict__ shs, const float *__restrict__ dirs,
              const uint8_t *__restrict__ visible, const uint8_t *__restrict__ clamped,
              const float *__restrict__ dL_dcolors, float *__restrict__ dL_dshs, float *__restrict__ dL_ddirs) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= P) return;
    if (!visible[i]) {
        float *oz = dL_dshs + (size_t)i * NB * 3;
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) oz[k] = 0.f;
        if (dL_ddirs) {
            dL_ddirs[3 * i] = 0.f; dL_ddirs[3 * i + 1] = 0.f; dL_ddirs[3 * i + 2] = 0.f;
        }
        return;
    }
    const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
    float B[16];
    sh_basis(DEG, x, y, z, B);
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g[c] = dL_dcolors[3 * i + c];
        if (!FREE && clamped[3 * i + c]) g[c] = 0.f;
    }
    float *o = dL_dshs + (size_t)i * NB * 3;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        o[3 * k + 0] = B[k] * g[0];
        o[3 * k + 1] = B[k] * g[1];
        o[3 * k + 2] = B[k] * g[2];
    }
    if (!dL_ddirs) return;  // direction gradient not requested: the coefficients are never read
    if (DEG == 0) {
        dL_ddirs[3 * i] = 0.f; dL_ddirs[3 * i + 1] = 0.f; dL_ddirs[3 * i + 2] = 0.f;
        return;
    }
    const float C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                         0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const float *sh = shs + (size_t)i * NB * 3;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
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
    dL_ddirs[3 * i] = ddx; dL_ddirs[3 * i + 1] = ddy; dL_ddirs[3 * i + 2] = ddz;
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

template <bool FREE>
static int sh_bwd_dispatch(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                           const uint8_t *clamped, const float *g, float *dshs, float *ddirs, hipStream_t s) {
    switch (degree) {
        case 0: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<0, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
        case 1: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<1, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
        case 2: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<2, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
        default: SPLAT_LAUNCH("sh_bwd", (sh_bwd_kernel<3, FREE>), pw_grid(P), dim3(PW_BLOCK), 0, s, P, shs, dirs, visible, clamped, g, dshs, ddirs); break;
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
                                         float *dL_dshs, float *dL_ddirs, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && degree >= 0 && degree <= 3, "degree must be 0..3");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(shs && dirs && visible && dL_dcolors && dL_dshs && (free_variant || clamped), "null pointer");
    return free_variant ? sh_bwd_dispatch<true>(P, shs, degree, dirs, visible, clamped, dL_dcolors, dL_dshs, dL_ddirs, (hipStream_t)stream)
                        : sh_bwd_dispatch<false>(P, shs, degree, dirs, visible, clamped, dL_dcolors, dL_dshs, dL_ddirs, (hipStream_t)stream);
}
