// One-Gaussian device functions of the streaming operators (project_point, compute_cov3d, ewa_project).
#pragma once
#include "common.h"

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

__device__ __forceinline__ void quat_R(const float *q, float R[3][3]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

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


// ---- one-Gaussian forms of the streaming operators (shared by the per-operator kernels of pointwise.hip and the
//      fused per-frame kernels of preprocess.hip: one source of truth for the arithmetic)

// orthographic projection + culling (reference twin: dptr_ortho_enhanced.py:177-202); returns the cull flag
__device__ __forceinline__ bool project_ortho_pt(const Cam &c, float x, float y, float z, int W, int H, float nearest,
                                                 float extent, float &u, float &v, float &d) {
    float tx, ty, tz;
    cam_xform(c, x, y, z, tx, ty, tz);
    u = ((tx + 1.f) * (float)W) / 2.f - 0.5f;
    v = ((ty + 1.f) * (float)H) / 2.f - 0.5f;
    d = tz;
    if (isnan(d)) d = 0.f;
    else if (isinf(d)) d = d > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    const float xlo = (float)((1.0 - (double)extent) * W * 0.5), xhi = (float)((1.0 + (double)extent) * W * 0.5);
    const float ylo = (float)((1.0 - (double)extent) * H * 0.5), yhi = (float)((1.0 + (double)extent) * H * 0.5);
    return (d <= nearest) || (u < xlo) || (u > xhi) || (v < ylo) || (v > yhi);
}

// perspective projection + culling (reference: src/project_point.cu:23-56); returns the cull flag
__device__ __forceinline__ bool project_persp_pt(const Cam &c, float x, float y, float z, int W, int H, float nearest,
                                                 float extent, float &u, float &v, float &d) {
    float tx, ty, tz;
    cam_xform(c, x, y, z, tx, ty, tz);
    const float inv = (float)(1.0 / ((double)tz + 1e-7));
    u = (float)((double)(c.fx * tx * inv + c.cx) - 0.5);
    v = (float)((double)(c.fy * ty * inv + c.cy) - 0.5);
    d = tz;
    bool cull = false;
    if (nearest > 0) cull = cull || (tz <= nearest);
    if (extent > 0) {
        const float xlo = (float)((double)((1 - extent) * W) * 0.5), xhi = (float)((double)((1 + extent) * W) * 0.5);
        const float ylo = (float)((double)((1 - extent) * H) * 0.5), yhi = (float)((double)((1 + extent) * H) * 0.5);
        cull = cull || (u < xlo) || (u > xhi) || (v < ylo) || (v > yhi);
    }
    return cull;
}

// gradient of the perspective projection w.r.t. the point (reference: src/project_point.cu:72-105; no epsilon in 1 / tz)
__device__ __forceinline__ void project_persp_grad_pt(const Cam &c, float x, float y, float z, float gu, float gv, float gd,
                                                      float g[3]) {
    float tx, ty, tz;
    cam_xform(c, x, y, z, tx, ty, tz);
    const float n2 = (float)(1.0 / (double)(tz * tz));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float a = 0.f;
        a += (c.fx * (c.e[j] * tz - tx * c.e[8 + j]) * n2) * gu;
        a += (c.fy * (c.e[4 + j] * tz - ty * c.e[8 + j]) * n2) * gv;
        a += c.e[8 + j] * gd;
        g[j] = a;
    }
}

// position gradient of the perspective EWA projection (the Jacobian depends on the camera-space point): from dL/dcov2d
// (dcx, dcy, dcz), the rows a, b of T = J R, the camera-space point t and the 3D covariance (reference: src/ewa_project.cu:152-205)
__device__ __forceinline__ void ewa_grad_pos_persp_pt(const Cam &c, const float t[3], const float a[3], const float b[3],
                                                      const float c3[6], float dcx, float dcy, float dcz, float g[3],
                                                      float da[3], float db[3], float dt[3]) {
    const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
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
    dt[0] = -c.fx * tz2 * dJ02;
    dt[1] = -c.fy * tz2 * dJ12;
    dt[2] = -c.fx * tz2 * dJ00 - c.fy * tz2 * dJ11 + (2 * c.fx * t[0]) * tz3 * dJ02 + (2 * c.fy * t[1]) * tz3 * dJ12;
    g[0] = c.e[0] * dt[0] + c.e[4] * dt[1] + c.e[8] * dt[2];
    g[1] = c.e[1] * dt[0] + c.e[5] * dt[1] + c.e[9] * dt[2];
    g[2] = c.e[2] * dt[0] + c.e[6] * dt[1] + c.e[10] * dt[2];
}

// gradient of the orthographic projection w.r.t. the point
__device__ __forceinline__ void project_ortho_grad_pt(const Cam &c, int W, int H, float gu, float gv, float gd,
                                                      float g[3]) {
    const float gx = gu * ((float)W / 2.f), gy = gv * ((float)H / 2.f);
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] = c.e[j] * gx + c.e[4 + j] * gy + c.e[8 + j] * gd;
}

// Sigma = (S R^T)^T (S R^T), upper triangle (reference: src/compute_cov3d.cu:14-58)
__device__ __forceinline__ void cov3d_pt(const float s[3], const float q[4], float o[6]) {
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

// reference: src/compute_cov3d.cu:60-117
__device__ __forceinline__ void cov3d_grad_pt(const float s[3], const float q[4], const float g[6], float ds[3],
                                              float dq[4]) {
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
    for (int k = 0; k < 3; ++k) ds[k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) D[a][b] = s[a] * dM[a][b];
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2 * z * (D[0][1] - D[1][0]) + 2 * y * (D[2][0] - D[0][2]) + 2 * x * (D[1][2] - D[2][1]);
    dq[1] = 2 * y * (D[1][0] + D[0][1]) + 2 * z * (D[2][0] + D[0][2]) + 2 * r * (D[1][2] - D[2][1]) - 4 * x * (D[2][2] + D[1][1]);
    dq[2] = 2 * x * (D[1][0] + D[0][1]) + 2 * r * (D[2][0] - D[0][2]) + 2 * z * (D[1][2] + D[2][1]) - 4 * y * (D[2][2] + D[0][0]);
    dq[3] = 2 * r * (D[0][1] - D[1][0]) + 2 * x * (D[2][0] + D[0][2]) + 2 * y * (D[1][2] + D[2][1]) - 4 * z * (D[1][1] + D[0][0]);
}

// conic / radius / tile count of one splat from its 2D covariance (reference: src/ewa_project.cu:43-83;
// ortho: dptr_ortho_enhanced.py:60-111); everything stays 0 for degenerate or off-screen splats
template <bool ORTHO>
__device__ __forceinline__ void ewa_finish_pt(const float cov[3], float2 q, int W, int H, float &o0, float &o1, float &o2,
                                              int &orad, int &otiles) {
    o0 = o1 = o2 = 0.f;
    orad = otiles = 0;
    const float det = cov[0] * cov[2] - cov[1] * cov[1];
    const bool bad = (det == 0.0f) || (ORTHO && isnan(det));
    if (bad) return;
    const float mid = 0.5f * (cov[0] + cov[2]);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float l1 = mid + sq, l2 = mid - sq;
    const int r = (int)ceilf(3.f * sqrtf(fmaxf(l1, l2)));
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int x0, y0, x1, y1;
    tile_rect(q.x, q.y, r, gx, gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) return;
    if (ORTHO) {
        o0 = cov[2] / det; o1 = -cov[1] / det; o2 = cov[0] / det;
    } else {
        const float di = 1.f / det;
        o0 = cov[2] * di; o1 = -cov[1] * di; o2 = cov[0] * di;
    }
    orad = r;
    otiles = (y1 - y0) * (x1 - x0);
}

// dL/dcov2d (dcx, dcy, dcz) and dL/dcov3d (o[6]) from dL/dconic (reference: src/ewa_project.cu:117-150)
__device__ __forceinline__ void ewa_grad_cov_pt(const float a[3], const float b[3], const float cov[3], float det,
                                                const float g[3], float &dcx, float &dcy, float &dcz, float o[6]) {
    const float nom = 1.0f / (det * det);
    dcx = nom * (-cov[2] * cov[2] * g[0] + cov[1] * cov[2] * g[1] + (det - cov[0] * cov[2]) * g[2]);
    dcy = nom * (2 * cov[1] * cov[2] * g[0] - (det + 2 * cov[1] * cov[1]) * g[1] + 2 * cov[0] * cov[1] * g[2]);
    dcz = nom * ((det - cov[0] * cov[2]) * g[0] + cov[0] * cov[1] * g[1] - cov[0] * cov[0] * g[2]);
    o[0] = a[0] * a[0] * dcx + a[0] * b[0] * dcy + b[0] * b[0] * dcz;
    o[1] = 2 * a[0] * a[1] * dcx + (a[0] * b[1] + b[0] * a[1]) * dcy + 2 * b[0] * b[1] * dcz;
    o[2] = 2 * a[0] * a[2] * dcx + (a[0] * b[2] + b[0] * a[2]) * dcy + 2 * b[0] * b[2] * dcz;
    o[3] = a[1] * a[1] * dcx + a[1] * b[1] * dcy + b[1] * b[1] * dcz;
    o[4] = 2 * a[1] * a[2] * dcx + (a[1] * b[2] + b[1] * a[2]) * dcy + 2 * b[1] * b[2] * dcz;
    o[5] = a[2] * a[2] * dcx + a[2] * b[2] * dcy + b[2] * b[2] * dcz;
}
