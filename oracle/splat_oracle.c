/*
 * splat_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar float32 restatement, by code reading, of the algorithm implemented by the
 * reference's CUDA extension `dptr.gs._C` (reference: src/submodules/dptr/dptr/gs/src/ *.cu files)
 * plus the two pure-torch orthographic twins in
 * src/pointrix/renderer/dptr_ortho_enhanced.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped path
 * (splatter_a_video_amd/) never does and fails loudly without its HIP library.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - ortho project_point, ortho EWA, SH basis  : pinned against golden vectors generated in the
 *     build container from the importable reference torch functions (tests/golden/).
 *   - perspective project/EWA, cov3d, sort, alpha blending (all variants): the reference holds
 *     no tests / golden vectors and its CUDA cannot be built here (no nvcc, GLM un-vendored)
 *     => "parity unpinned" for those: restated by code reading, cross-checked by an
 *     independent float64 torch twin + autograd and by mathematical identities.
 *
 * Conventions: all tensors row-major float32 / int32 / uint8(bool); "extr" = first 12 floats
 * of a row-major [R|T] (3x4 or 4x4); "intr" = [fx, fy, cx, cy]; tile = 16x16 pixels.
 * Gradient accumulators that the reference fills with float atomicAdd (order-dependent) are
 * accumulated here in double and rounded once, i.e. the oracle returns the order-independent
 * centre value of the reference's possible results.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; optional -fopenmp for the
 * cpu_baseline timing leg; tests run it single-threaded).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define TILE_PIX 256

static int g_threads = 1;

void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int oracle_get_threads(void) { return g_threads; }
int oracle_has_openmp(void) {
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Tile rectangle of a splat. Reference: include/utils.h:17-37 (get_rect).
 * float arithmetic, C truncation toward zero, then clamp to [0, grid]. */
static inline void tile_rect(float px, float py, int r, int gx, int gy,
                             int *x0, int *y0, int *x1, int *y1) {
    const float fr = (float)r;
    *x0 = imin(gx, imax(0, (int)((px - fr) / (float)TILE)));
    *y0 = imin(gy, imax(0, (int)((py - fr) / (float)TILE)));
    *x1 = imin(gx, imax(0, (int)((((px + fr) + (float)TILE) - 1.0f) / (float)TILE)));
    *y1 = imin(gy, imax(0, (int)((((py + fr) + (float)TILE) - 1.0f) / (float)TILE)));
}

/* ------------------------------------------------------------------------------------------
 * project_point (perspective). Reference: src/project_point.cu:13-57 (fwd), :59-145 (bwd).
 * Outputs must be zero-initialised by the caller (culled points keep 0).
 * ---------------------------------------------------------------------------------------- */
void oracle_project_point_forward(int P, const float *xyz, const float *intr, const float *extr,
                                  int W, int H, float nearest, float extent,
                                  float *uv, float *depth) {
    for (int i = 0; i < P; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float tx = extr[0] * x + extr[1] * y + extr[2] * z + extr[3];
        const float ty = extr[4] * x + extr[5] * y + extr[6] * z + extr[7];
        const float tz = extr[8] * x + extr[9] * y + extr[10] * z + extr[11];
        /* the reference's literals 1.0, 1e-7, 0.5 are doubles */
        const float inv = (float)(1.0 / ((double)tz + 1e-7));
        const float u = (float)((double)(intr[0] * tx * inv + intr[2]) - 0.5);
        const float v = (float)((double)(intr[1] * ty * inv + intr[3]) - 0.5);
        int cull = 0;
        if (nearest > 0) cull |= (tz <= nearest);
        if (extent > 0) {
            const float xlo = (float)((double)((1 - extent) * W) * 0.5);
            const float xhi = (float)((double)((1 + extent) * W) * 0.5);
            const float ylo = (float)((double)((1 - extent) * H) * 0.5);
            const float yhi = (float)((double)((1 + extent) * H) * 0.5);
            cull |= (u < xlo) || (u > xhi) || (v < ylo) || (v > yhi);
        }
        if (cull) continue;
        uv[2 * i] = u;
        uv[2 * i + 1] = v;
        depth[i] = tz;
    }
}

/* dL_dintr[4] / dL_dextr[12] may be NULL (the reference only fills them when requires_grad). */
void oracle_project_point_backward(int P, const float *xyz, const float *intr, const float *extr,
                                   int W, int H, const float *uv, const float *depth,
                                   const float *dL_duv, const float *dL_ddepth,
                                   float *dL_dxyz, float *dL_dintr, float *dL_dextr) {
    (void)W; (void)H; (void)uv;
    double ai[4] = {0, 0, 0, 0};
    double ae[12] = {0};
    for (int i = 0; i < P; ++i) {
        if (depth[i] == 0) continue;
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float tx = extr[0] * x + extr[1] * y + extr[2] * z + extr[3];
        const float ty = extr[4] * x + extr[5] * y + extr[6] * z + extr[7];
        const float tz = extr[8] * x + extr[9] * y + extr[10] * z + extr[11];
        const float n1 = (float)(1.0 / (double)tz);
        const float n2 = (float)(1.0 / (double)(tz * tz));
        const float gu = dL_duv[2 * i], gv = dL_duv[2 * i + 1], gd = dL_ddepth[i];
        for (int j = 0; j < 3; ++j) {
            float g = 0.f;
            g += (intr[0] * (extr[j] * tz - tx * extr[8 + j]) * n2) * gu;
            g += (intr[1] * (extr[4 + j] * tz - ty * extr[8 + j]) * n2) * gv;
            g += extr[8 + j] * gd;
            dL_dxyz[3 * i + j] += g;
        }
        if (dL_dintr) {
            ai[0] += tx * n1 * gu;
            ai[1] += ty * n1 * gv;
            ai[2] += gu;
            ai[3] += gv;
        }
        if (dL_dextr) {
            const float p[4] = {x, y, z, 1.f};
            for (int j = 0; j < 4; ++j) {
                ae[j] += intr[0] * p[j] * n1 * gu;
                ae[4 + j] += intr[1] * p[j] * n1 * gv;
                ae[8 + j] += -intr[0] * p[j] * tx * n2 * gu;
                ae[8 + j] += -intr[1] * p[j] * ty * n2 * gv;
                ae[8 + j] += p[j] * gd;
            }
        }
    }
    if (dL_dintr) for (int k = 0; k < 4; ++k) dL_dintr[k] += (float)ai[k];
    if (dL_dextr) for (int k = 0; k < 12; ++k) dL_dextr[k] += (float)ae[k];
}

/* ------------------------------------------------------------------------------------------
 * project_point (orthographic twin). Reference: src/pointrix/renderer/dptr_ortho_enhanced.py
 * :145-202. uv = (p_cam.xy + 1) * [W,H] / 2 - 0.5, depth = nan_to_num(p_cam.z), culled -> 0.
 * Backward = autograd of that torch code: gradient flows through clone/index-assign, i.e.
 * zero for culled points; nan_to_num passes gradient where finite.
 * ---------------------------------------------------------------------------------------- */
static inline float nan_to_num_f(float v) {
    if (isnan(v)) return 0.f;
    if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return v;
}

void oracle_project_point_ortho_forward(int P, const float *xyz, const float *extr, int W, int H,
                                        float nearest, float extent, float *uv, float *depth) {
    for (int i = 0; i < P; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float tx = extr[0] * x + extr[1] * y + extr[2] * z + extr[3];
        const float ty = extr[4] * x + extr[5] * y + extr[6] * z + extr[7];
        const float tz = extr[8] * x + extr[9] * y + extr[10] * z + extr[11];
        const float u = ((tx + 1.f) * (float)W) / 2.f - 0.5f;
        const float v = ((ty + 1.f) * (float)H) / 2.f - 0.5f;
        const float d = nan_to_num_f(tz);
        /* python-float thresholds are doubles multiplied in double, compared after the tensor's
         * float32 scalar-promotion: torch compares float32 tensor with a python scalar in
         * float32 (the scalar is cast to the tensor dtype). */
        const float xlo = (float)((1.0 - (double)extent) * W * 0.5);
        const float xhi = (float)((1.0 + (double)extent) * W * 0.5);
        const float ylo = (float)((1.0 - (double)extent) * H * 0.5);
        const float yhi = (float)((1.0 + (double)extent) * H * 0.5);
        const int cull = (d <= nearest) || (u < xlo) || (u > xhi) || (v < ylo) || (v > yhi);
        if (cull) continue;
        uv[2 * i] = u;
        uv[2 * i + 1] = v;
        depth[i] = d;
    }
}

void oracle_project_point_ortho_backward(int P, const float *extr, int W, int H,
                                         const float *depth, const float *dL_duv,
                                         const float *dL_ddepth, float *dL_dxyz) {
    for (int i = 0; i < P; ++i) {
        /* culled <=> depth == 0 (a surviving point has depth > nearest >= 0 ... the reference
         * renderer always passes nearest=0.01; with nearest<=0 a surviving exact-0 depth has
         * measure zero). */
        if (depth[i] == 0) continue;
        const float gx = dL_duv[2 * i] * ((float)W / 2.f);
        const float gy = dL_duv[2 * i + 1] * ((float)H / 2.f);
        const float gz = dL_ddepth[i];
        for (int j = 0; j < 3; ++j)
            dL_dxyz[3 * i + j] += extr[j] * gx + extr[4 + j] * gy + extr[8 + j] * gz;
    }
}

/* ------------------------------------------------------------------------------------------
 * compute_cov3d. Reference: src/compute_cov3d.cu:14-58 (fwd), :60-117 (bwd).
 * Sigma = R diag(s^2) R^T with R the standard rotation matrix of unit quaternion (r,x,y,z).
 * ---------------------------------------------------------------------------------------- */
static inline void quat_to_R(const float *q, float R[3][3]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

void oracle_compute_cov3d_forward(int P, const float *scales, const float *uquats,
                                  const uint8_t *visible, float *cov3d) {
    for (int i = 0; i < P; ++i) {
        if (!visible[i]) continue;
        float R[3][3];
        quat_to_R(uquats + 4 * i, R);
        const float *s = scales + 3 * i;
        /* M = diag(s) R^T  (M[k][j] = s_k R[j][k]);  Sigma = M^T M */
        float M[3][3];
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) M[k][j] = s[k] * R[j][k];
        float S[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                S[a][b] = M[0][a] * M[0][b] + M[1][a] * M[1][b] + M[2][a] * M[2][b];
        float *o = cov3d + 6 * i;
        o[0] = S[0][0]; o[1] = S[0][1]; o[2] = S[0][2];
        o[3] = S[1][1]; o[4] = S[1][2]; o[5] = S[2][2];
    }
}

void oracle_compute_cov3d_backward(int P, const float *scales, const float *uquats,
                                   const uint8_t *visible, const float *dL_dcov3d,
                                   float *dL_dscales, float *dL_duquats) {
    for (int i = 0; i < P; ++i) {
        if (!visible[i]) continue;
        float R[3][3];
        const float *q = uquats + 4 * i;
        quat_to_R(q, R);
        const float *s = scales + 3 * i;
        const float *g = dL_dcov3d + 6 * i;
        float M[3][3];
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) M[k][j] = s[k] * R[j][k];
        /* symmetric gradient with halved off-diagonals (:70-78) */
        const float G[3][3] = {{g[0], 0.5f * g[1], 0.5f * g[2]},
                               {0.5f * g[1], g[3], 0.5f * g[4]},
                               {0.5f * g[2], 0.5f * g[4], g[5]}};
        /* dL_dM = 2 M G */
        float dM[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                dM[a][b] = 2.0f * (M[a][0] * G[0][b] + M[a][1] * G[1][b] + M[a][2] * G[2][b]);
        /* dL_ds_k = sum_j R[j][k] dM[k][j] */
        for (int k = 0; k < 3; ++k)
            dL_dscales[3 * i + k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
        /* D[a][b] = s_a dM[a][b] = dL/dR[b][a] */
        float D[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) D[a][b] = s[a] * dM[a][b];
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        float *o = dL_duquats + 4 * i;
        o[0] = 2 * z * (D[0][1] - D[1][0]) + 2 * y * (D[2][0] - D[0][2]) + 2 * x * (D[1][2] - D[2][1]);
        o[1] = 2 * y * (D[1][0] + D[0][1]) + 2 * z * (D[2][0] + D[0][2]) + 2 * r * (D[1][2] - D[2][1]) -
               4 * x * (D[2][2] + D[1][1]);
        o[2] = 2 * x * (D[1][0] + D[0][1]) + 2 * r * (D[2][0] - D[0][2]) + 2 * z * (D[1][2] + D[2][1]) -
               4 * y * (D[2][2] + D[0][0]);
        o[3] = 2 * r * (D[0][1] - D[1][0]) + 2 * x * (D[2][0] + D[0][2]) + 2 * y * (D[1][2] + D[2][1]) -
               4 * z * (D[1][1] + D[0][0]);
    }
}

/* ------------------------------------------------------------------------------------------
 * ewa_project. Perspective: src/ewa_project.cu:16-83 (fwd), :85-252 (bwd).
 * Ortho twin: dptr_ortho_enhanced.py:18-111 (J = diag(W/2, H/2), backward = autograd: only
 * conic -> cov3d, no position gradient).
 * a[k] = T(0,k), b[k] = T(1,k) with T = J R_w2c (2x3).
 * ---------------------------------------------------------------------------------------- */
static inline void ewa_T(int ortho, const float *p, const float *intr, const float *extr, int W, int H,
                         float a[3], float b[3], float t[3], float Jm[4]) {
    t[0] = extr[0] * p[0] + extr[1] * p[1] + extr[2] * p[2] + extr[3];
    t[1] = extr[4] * p[0] + extr[5] * p[1] + extr[6] * p[2] + extr[7];
    t[2] = extr[8] * p[0] + extr[9] * p[1] + extr[10] * p[2] + extr[11];
    float J00, J11, J02, J12;
    if (ortho) {
        J00 = (float)W / 2.f; J11 = (float)H / 2.f; J02 = 0.f; J12 = 0.f;
    } else {
        const float fx = intr[0], fy = intr[1];
        J00 = fx / t[2]; J11 = fy / t[2];
        J02 = -(fx * t[0]) / (t[2] * t[2]);
        J12 = -(fy * t[1]) / (t[2] * t[2]);
    }
    for (int k = 0; k < 3; ++k) {
        a[k] = J00 * extr[k] + 0.0f * extr[4 + k] + J02 * extr[8 + k];
        b[k] = 0.0f * extr[k] + J11 * extr[4 + k] + J12 * extr[8 + k];
    }
    Jm[0] = J00; Jm[1] = J11; Jm[2] = J02; Jm[3] = J12;
}

static inline void ewa_cov2d(int ortho, const float a[3], const float b[3], const float *c3, float cov[3]) {
    const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float Xa[3], Xb[3];
    for (int c = 0; c < 3; ++c) {
        Xa[c] = a[0] * S[0][c] + a[1] * S[1][c] + a[2] * S[2][c];
        Xb[c] = b[0] * S[0][c] + b[1] * S[1][c] + b[2] * S[2][c];
    }
    cov[0] = (Xa[0] * a[0] + Xa[1] * a[1] + Xa[2] * a[2]) + 0.3f;
    /* CUDA reads cov2D[0][1] (GLM column 0, row 1) = row(b).Sigma.a ; the torch twin reads
     * cov2d[...,0,1] = row(a).Sigma.b -- equal mathematically, kept apart for rounding. */
    cov[1] = ortho ? (Xa[0] * b[0] + Xa[1] * b[1] + Xa[2] * b[2])
                   : (Xb[0] * a[0] + Xb[1] * a[1] + Xb[2] * a[2]);
    cov[2] = (Xb[0] * b[0] + Xb[1] * b[1] + Xb[2] * b[2]) + 0.3f;
}

void oracle_ewa_project_forward(int ortho, int P, const float *xyz, const float *cov3d,
                                const float *intr, const float *extr, const float *uv,
                                int W, int H, const uint8_t *visible,
                                float *conic, int32_t *radius, int32_t *tiles) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int i = 0; i < P; ++i) {
        if (!visible[i]) continue;
        float a[3], b[3], t[3], Jm[4], cov[3];
        ewa_T(ortho, xyz + 3 * i, intr, extr, W, H, a, b, t, Jm);
        ewa_cov2d(ortho, a, b, cov3d + 6 * i, cov);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        if (ortho && isnan(det)) continue; /* twin: nan_to_num + mask => zeros */
        const float mid = 0.5f * (cov[0] + cov[2]);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + sq, l2 = mid - sq;
        const float fr = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        const int r = (int)fr;
        int x0, y0, x1, y1;
        tile_rect(uv[2 * i], uv[2 * i + 1], r, gx, gy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        const float di = 1.f / det;
        if (ortho) { /* twin divides, CUDA multiplies by reciprocal */
            conic[3 * i] = cov[2] / det; conic[3 * i + 1] = -cov[1] / det; conic[3 * i + 2] = cov[0] / det;
        } else {
            conic[3 * i] = cov[2] * di; conic[3 * i + 1] = -cov[1] * di; conic[3 * i + 2] = cov[0] * di;
        }
        radius[i] = r;
        tiles[i] = (y1 - y0) * (x1 - x0);
    }
}

/* dL_dxyz / dL_dcov3d zero-initialised by caller; dL_dintr / dL_dextr may be NULL.
 * Gate: radius > 0 (reference saves radius, :97). */
void oracle_ewa_project_backward(int ortho, int P, const float *xyz, const float *cov3d,
                                 const float *intr, const float *extr, int W, int H,
                                 const int32_t *radius, const float *dL_dconic,
                                 float *dL_dxyz, float *dL_dcov3d, float *dL_dintr, float *dL_dextr) {
    double ai[2] = {0, 0};
    double ae[12] = {0};
    for (int i = 0; i < P; ++i) {
        if (!(radius[i] > 0)) continue;
        float a[3], b[3], t[3], Jm[4], cov[3];
        const float *p = xyz + 3 * i;
        const float *c3 = cov3d + 6 * i;
        ewa_T(ortho, p, intr, extr, W, H, a, b, t, Jm);
        ewa_cov2d(ortho, a, b, c3, cov);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        const float nom = 1.0f / (det * det);
        const float gx_ = dL_dconic[3 * i], gy_ = dL_dconic[3 * i + 1], gz_ = dL_dconic[3 * i + 2];
        const float dcx = nom * (-cov[2] * cov[2] * gx_ + cov[1] * cov[2] * gy_ + (det - cov[0] * cov[2]) * gz_);
        const float dcy = nom * (2 * cov[1] * cov[2] * gx_ - (det + 2 * cov[1] * cov[1]) * gy_ + 2 * cov[0] * cov[1] * gz_);
        const float dcz = nom * ((det - cov[0] * cov[2]) * gx_ + cov[0] * cov[1] * gy_ - cov[0] * cov[0] * gz_);
        float *o = dL_dcov3d + 6 * i;
        o[0] += a[0] * a[0] * dcx + a[0] * b[0] * dcy + b[0] * b[0] * dcz;
        o[1] += 2 * a[0] * a[1] * dcx + (a[0] * b[1] + b[0] * a[1]) * dcy + 2 * b[0] * b[1] * dcz;
        o[2] += 2 * a[0] * a[2] * dcx + (a[0] * b[2] + b[0] * a[2]) * dcy + 2 * b[0] * b[2] * dcz;
        o[3] += a[1] * a[1] * dcx + a[1] * b[1] * dcy + b[1] * b[1] * dcz;
        o[4] += 2 * a[1] * a[2] * dcx + (a[1] * b[2] + b[1] * a[2]) * dcy + 2 * b[1] * b[2] * dcz;
        o[5] += a[2] * a[2] * dcx + a[2] * b[2] * dcy + b[2] * b[2] * dcz;
        if (ortho) continue; /* T is constant in the twin: nothing flows to xyz / camera */

        const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
        float da[3], db[3]; /* dL/da_k, dL/db_k */
        for (int k = 0; k < 3; ++k) {
            const float Sa = a[0] * S[0][k] + a[1] * S[1][k] + a[2] * S[2][k];
            const float Sb = b[0] * S[0][k] + b[1] * S[1][k] + b[2] * S[2][k];
            da[k] = 2 * Sa * dcx + Sb * dcy;
            db[k] = Sa * dcy + 2 * Sb * dcz;
        }
        const float dJ00 = extr[0] * da[0] + extr[1] * da[1] + extr[2] * da[2];
        const float dJ02 = extr[8] * da[0] + extr[9] * da[1] + extr[10] * da[2];
        const float dJ11 = extr[4] * db[0] + extr[5] * db[1] + extr[6] * db[2];
        const float dJ12 = extr[8] * db[0] + extr[9] * db[1] + extr[10] * db[2];
        const float fx = intr[0], fy = intr[1];
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = -fx * tz2 * dJ02;
        const float dty = -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
        if (dL_dintr) {
            ai[0] += tz * dJ00; ai[0] += -t[0] * tz2 * dJ02;
            ai[1] += tz * dJ11; ai[1] += -t[1] * tz2 * dJ12;
        }
        if (dL_dextr) {
            for (int k = 0; k < 3; ++k) {
                ae[k] += Jm[0] * da[k];
                ae[4 + k] += Jm[1] * db[k];
                ae[8 + k] += Jm[2] * da[k] + Jm[3] * db[k];
            }
            const float pp[4] = {p[0], p[1], p[2], 1.f};
            for (int k = 0; k < 4; ++k) {
                ae[k] += pp[k] * dtx;
                ae[4 + k] += pp[k] * dty;
                ae[8 + k] += pp[k] * dtz;
            }
        }
        dL_dxyz[3 * i + 0] = extr[0] * dtx + extr[4] * dty + extr[8] * dtz;
        dL_dxyz[3 * i + 1] = extr[1] * dtx + extr[5] * dty + extr[9] * dtz;
        dL_dxyz[3 * i + 2] = extr[2] * dtx + extr[6] * dty + extr[10] * dtz;
    }
    if (dL_dintr) { dL_dintr[0] += (float)ai[0]; dL_dintr[1] += (float)ai[1]; }
    if (dL_dextr) for (int k = 0; k < 12; ++k) dL_dextr[k] += (float)ae[k];
}

/* ------------------------------------------------------------------------------------------
 * compute_sh / compute_sh_free. Reference: src/compute_sh.cu:15-195, src/compute_sh_free.cu.
 * shs is addressed FLAT with stride nb=(deg+1)^2 coefficient triplets per point
 * (compute_sh.cu:45) -- kept as is.  free != 0: no +0.5, no clamp, no clamped mask.
 * ---------------------------------------------------------------------------------------- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* basis[k] = d rgb / d sh_k (same scalar for the three channels) */
static inline int sh_basis(int deg, float x, float y, float z, float B[16]) {
    B[0] = SH_C0;
    if (deg < 1) return 1;
    B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
    if (deg < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.0f * zz - xx - yy);
    B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return 9;
    B[9] = SH_C3[0] * y * (3.0f * xx - yy);
    B[10] = SH_C3[1] * xy * z;
    B[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
    B[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    B[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
    B[14] = SH_C3[5] * z * (xx - yy);
    B[15] = SH_C3[6] * x * (xx - 3.0f * yy);
    return 16;
}

void oracle_compute_sh_forward(int free_variant, int P, const float *shs, int deg, const float *dirs,
                               const uint8_t *visible, float *colors, uint8_t *clamped) {
    const int nb = (deg + 1) * (deg + 1);
    for (int i = 0; i < P; ++i) {
        if (!visible[i]) continue;
        float B[16];
        sh_basis(deg, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], B);
        const float *sh = shs + (size_t)i * nb * 3;
        for (int c = 0; c < 3; ++c) {
            float r = B[0] * sh[c];
            for (int k = 1; k < nb; ++k) r = r + B[k] * sh[3 * k + c];
            if (free_variant) {
                colors[3 * i + c] = r;
            } else {
                r += 0.5f;
                clamped[3 * i + c] = (r < 0);
                colors[3 * i + c] = r > 0.f ? r : 0.f;
            }
        }
    }
}

void oracle_compute_sh_backward(int free_variant, int P, const float *shs, int deg, const float *dirs,
                                const uint8_t *visible, const uint8_t *clamped, const float *dL_dcolors,
                                float *dL_dshs, float *dL_ddirs) {
    const int nb = (deg + 1) * (deg + 1);
    for (int i = 0; i < P; ++i) {
        if (!visible[i]) continue;
        const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        float B[16];
        sh_basis(deg, x, y, z, B);
        const float *sh = shs + (size_t)i * nb * 3;
        float g[3];
        for (int c = 0; c < 3; ++c) {
            g[c] = dL_dcolors[3 * i + c];
            if (!free_variant && clamped[3 * i + c]) g[c] = 0.f;
        }
        float *o = dL_dshs + (size_t)i * nb * 3;
        for (int k = 0; k < nb; ++k)
            for (int c = 0; c < 3; ++c) o[3 * k + c] = B[k] * g[c];
        /* d rgb_c / d(x,y,z) (compute_sh.cu:121-188) */
        float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
#define SHC(k, c) sh[3 * (k) + (c)]
        if (deg > 0) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            for (int c = 0; c < 3; ++c) {
                dx[c] = -SH_C1 * SHC(3, c);
                dy[c] = -SH_C1 * SHC(1, c);
                dz[c] = SH_C1 * SHC(2, c);
                if (deg > 1) {
                    dx[c] += SH_C2[0] * y * SHC(4, c) + SH_C2[2] * 2.f * -x * SHC(6, c) +
                             SH_C2[3] * z * SHC(7, c) + SH_C2[4] * 2.f * x * SHC(8, c);
                    dy[c] += SH_C2[0] * x * SHC(4, c) + SH_C2[1] * z * SHC(5, c) +
                             SH_C2[2] * 2.f * -y * SHC(6, c) + SH_C2[4] * 2.f * -y * SHC(8, c);
                    dz[c] += SH_C2[1] * y * SHC(5, c) + SH_C2[2] * 2.f * 2.f * z * SHC(6, c) +
                             SH_C2[3] * x * SHC(7, c);
                    if (deg > 2) {
                        dx[c] += (SH_C3[0] * SHC(9, c) * 3.f * 2.f * xy + SH_C3[1] * SHC(10, c) * yz +
                                  SH_C3[2] * SHC(11, c) * -2.f * xy + SH_C3[3] * SHC(12, c) * -3.f * 2.f * xz +
                                  SH_C3[4] * SHC(13, c) * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * SHC(14, c) * 2.f * xz + SH_C3[6] * SHC(15, c) * 3.f * (xx - yy));
                        dy[c] += (SH_C3[0] * SHC(9, c) * 3.f * (xx - yy) + SH_C3[1] * SHC(10, c) * xz +
                                  SH_C3[2] * SHC(11, c) * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * SHC(12, c) * -3.f * 2.f * yz + SH_C3[4] * SHC(13, c) * -2.f * xy +
                                  SH_C3[5] * SHC(14, c) * -2.f * yz + SH_C3[6] * SHC(15, c) * -3.f * 2.f * xy);
                        dz[c] += (SH_C3[1] * SHC(10, c) * xy + SH_C3[2] * SHC(11, c) * 4.f * 2.f * yz +
                                  SH_C3[3] * SHC(12, c) * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * SHC(13, c) * 4.f * 2.f * xz + SH_C3[5] * SHC(14, c) * (xx - yy));
                    }
                }
            }
        }
#undef SHC
        dL_ddirs[3 * i + 0] = dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2];
        dL_ddirs[3 * i + 1] = dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2];
        dL_ddirs[3 * i + 2] = dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2];
    }
}

/* ------------------------------------------------------------------------------------------
 * sort_gaussian. Reference: dptr/gs/sort_gaussian.py:42-52, src/sort_gaussian.cu:16-70.
 *  key = (tile_id << 32) | bits(depth); value = gaussian index; pairs written at the
 *  Gaussian's slot range in the inclusive cumsum of `tiles`.
 *  Ordering rule fixed by this build (reference uses an unstable torch.sort): stable,
 *  i.e. ties (same tile, bit-equal depth) by ascending Gaussian index.
 * ---------------------------------------------------------------------------------------- */
void oracle_cumsum_i32(int P, const int32_t *in, int32_t *out) {
    int32_t acc = 0;
    for (int i = 0; i < P; ++i) { acc += in[i]; out[i] = acc; }
}

void oracle_compute_gaussian_key(int P, const float *uv, const float *depth, int W, int H,
                                 const int32_t *radius, const int32_t *tiles_cumsum,
                                 int64_t *key, int32_t *idx) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int i = 0; i < P; ++i) {
        if (radius[i] <= 0) continue;
        int x0, y0, x1, y1;
        tile_rect(uv[2 * i], uv[2 * i + 1], radius[i], gx, gy, &x0, &y0, &x1, &y1);
        int cur = (i == 0) ? 0 : tiles_cumsum[i - 1];
        int32_t bits;
        memcpy(&bits, depth + i, 4);
        const int64_t did = (int64_t)bits; /* sign-extending, as the reference's cast does */
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                const int64_t tid = (int64_t)ty * gx + tx;
                key[cur] = (tid << 32) | did;
                idx[cur] = i;
                ++cur;
            }
    }
}

typedef struct { int64_t k; int32_t v; } kv_t;
static int kv_cmp(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return x->v < y->v ? -1 : (x->v > y->v ? 1 : 0); /* values ascend within a Gaussian-order emit => stable */
}

/* sorts (key, idx) ascending by key, ties by idx (== stable w.r.t. emit order) */
void oracle_sort_pairs(int M, int64_t *key, int32_t *idx) {
    if (M <= 0) return;
    kv_t *t = (kv_t *)malloc(sizeof(kv_t) * (size_t)M);
    for (int m = 0; m < M; ++m) { t[m].k = key[m]; t[m].v = idx[m]; }
    qsort(t, (size_t)M, sizeof(kv_t), kv_cmp);
    for (int m = 0; m < M; ++m) { key[m] = t[m].k; idx[m] = t[m].v; }
    free(t);
}

/* tile_range zero-initialised by caller ([T,2]); untouched tiles stay (0,0). */
void oracle_compute_tile_range(int M, const int64_t *key_sorted, int32_t *tile_range) {
    for (int m = 0; m < M; ++m) {
        const int cur = (int)(key_sorted[m] >> 32);
        if (m == 0) tile_range[2 * cur] = 0;
        if (m == M - 1) tile_range[2 * cur + 1] = M;
        if (m == 0) continue;
        const int prev = (int)(key_sorted[m - 1] >> 32);
        if (prev != cur) {
            tile_range[2 * prev + 1] = m;
            tile_range[2 * cur] = m;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * alpha blending. Reference: src/alpha_blending.cu:16-249 (+ launch/chunk loop :287-394,
 * :440-580), src/alpha_blending_enhanced.cu:57-133, src/alpha_blending_with_bias.cu:88-89,
 * :211-214,:259-261.
 *  feature is [P,C] row-major here (the reference transposes on the host; same numbers).
 *  out [C,H,W], final_T [H,W], ncontrib [H,W], gs_idx [H,W,K] (caller pre-fills -1).
 *  bias == NULL -> plain; K <= 0 / gs_idx == NULL -> not enhanced.
 * ---------------------------------------------------------------------------------------- */
static inline int chunk_size(int rem) { return rem > 32 ? 32 : rem; }

void oracle_alpha_blending_forward(int P, int C, const float *uv, const float *conic,
                                   const float *opacity, const float *feature, const float *bias,
                                   const int32_t *idx_sorted, const int32_t *tile_range,
                                   float bg, int W, int H, int K, int enable_truncation,
                                   float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx) {
    (void)P;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int enhanced = (gs_idx != NULL && K > 0);
    const size_t HW = (size_t)H * W;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
#endif
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const int r0 = tile_range[2 * tile], r1 = tile_range[2 * tile + 1];
        float *F = (float *)malloc(sizeof(float) * (size_t)(C > 0 ? C : 1));
        for (int py = ty * TILE; py < imin(H, ty * TILE + TILE); ++py)
            for (int px = tx * TILE; px < imin(W, tx * TILE + TILE); ++px) {
                const size_t pix = (size_t)W * py + px;
                float T = 1.0f;
                int contributor = 0, last = 0, layer = 0, done = 0;
                for (int c = 0; c < C; ++c) F[c] = 0.f;
                for (int s = r0; s < r1 && !done; ++s) {
                    ++contributor;
                    const int id = idx_sorted[s];
                    const float dx = uv[2 * id] - (float)px;
                    const float dy = uv[2 * id + 1] - (float)py;
                    const float *cn = conic + 3 * id;
                    const float power = -0.5f * (cn[0] * dx * dx + cn[2] * dy * dy) - cn[1] * dx * dy;
                    if (power > 0) continue;
                    float araw = opacity[id] * expf(power);
                    if (bias) araw = araw + bias[id];
                    const float alpha = fminf(0.99f, araw);
                    if ((double)alpha < 1.0 / (double)255.0f) continue;
                    const float nT = T * (1 - alpha);
                    if (nT < 0.0001f) { done = 1; continue; }
                    const float *f = feature + (size_t)id * C;
                    for (int c = 0; c < C; ++c) F[c] += f[c] * alpha * T;
                    T = nT;
                    last = contributor;
                    if (enhanced) {
                        if (enable_truncation) {
                            gs_idx[pix * K + layer] = id;
                            ++layer;
                            if (layer >= K) { done = 1; continue; }
                        } else if (layer < K) {
                            gs_idx[pix * K + layer] = id;
                            ++layer;
                        }
                    }
                }
                final_T[pix] = T;
                ncontrib[pix] = last;
                for (int c = 0; c < C; ++c) out[(size_t)c * HW + pix] = F[c] + T * bg;
            }
        free(F);
    }
}

/* All gradient outputs zero-initialised by the caller. dL_dfeature is [P,C].
 * Channel chunks (<=32 channels each, alpha_blending.cu:440-577) only matter for dL_dabs_uv,
 * which sums |.| per chunk. with_bias stops a pixel once its reconstructed T < 1e-4. */
void oracle_alpha_blending_backward(int P, int C, const float *uv, const float *conic,
                                    const float *opacity, const float *feature, const float *bias,
                                    const int32_t *idx_sorted, const int32_t *tile_range,
                                    float bg, int W, int H, const float *final_T,
                                    const int32_t *ncontrib, const float *dL_dout,
                                    float *dL_duv, float *dL_dabs_uv, float *dL_dconic,
                                    float *dL_dopacity, float *dL_dfeature, float *dL_dbias) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    const size_t NP = (size_t)P;
    double *a_uv = (double *)calloc(NP * 2, sizeof(double));
    double *a_abs = (double *)calloc(NP * 2, sizeof(double));
    double *a_con = (double *)calloc(NP * 3, sizeof(double));
    double *a_op = (double *)calloc(NP, sizeof(double));
    double *a_bias = (double *)calloc(NP, sizeof(double));
    double *a_f = (double *)calloc(NP * (size_t)(C > 0 ? C : 1), sizeof(double));

    for (int C0 = 0; C0 < C; C0 += 32) {
        const int cn_ = chunk_size(C - C0);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
#endif
        for (int tile = 0; tile < gx * gy; ++tile) {
            const int tx = tile % gx, ty = tile / gx;
            const int r0 = tile_range[2 * tile], r1 = tile_range[2 * tile + 1];
            float acc[32], gp[32], lastf[32];
            for (int py = ty * TILE; py < imin(H, ty * TILE + TILE); ++py)
                for (int px = tx * TILE; px < imin(W, tx * TILE + TILE); ++px) {
                    const size_t pix = (size_t)W * py + px;
                    const float Tf = final_T[pix];
                    float T = Tf;
                    int contributor = r1 - r0;
                    const int last = ncontrib[pix];
                    float last_alpha = 0.f;
                    int done = 0;
                    for (int c = 0; c < cn_; ++c) {
                        acc[c] = 0.f; lastf[c] = 0.f;
                        gp[c] = dL_dout[(size_t)(C0 + c) * HW + pix];
                    }
                    for (int s = r1 - 1; s >= r0 && !done; --s) {
                        --contributor;
                        if (contributor >= last) continue;
                        const int id = idx_sorted[s];
                        const float dx = uv[2 * id] - (float)px;
                        const float dy = uv[2 * id + 1] - (float)py;
                        const float *cn = conic + 3 * id;
                        const float power = -0.5f * (cn[0] * dx * dx + cn[2] * dy * dy) - cn[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float opac = opacity[id];
                        float araw = opac * G;
                        if (bias) araw = araw + bias[id];
                        const float alpha = fminf(0.99f, araw);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / (1.f - alpha);
                        const float w = alpha * T;
                        float dLa = 0.f;
                        const float *f = feature + (size_t)id * C + C0;
                        for (int c = 0; c < cn_; ++c) {
                            acc[c] = last_alpha * lastf[c] + (1.f - last_alpha) * acc[c];
                            lastf[c] = f[c];
                            dLa += (f[c] - acc[c]) * gp[c];
                            const double v = (double)(w * gp[c]);
#ifdef _OPENMP
#pragma omp atomic
#endif
                            a_f[(size_t)id * C + C0 + c] += v;
                        }
                        dLa *= T;
                        last_alpha = alpha;
                        float bgdot = 0.f;
                        for (int c = 0; c < cn_; ++c) bgdot += bg * gp[c];
                        dLa += (-Tf / (1.f - alpha)) * bgdot;
                        const float dLG = opac * dLa;
                        const float dGx = -G * dx * cn[0] - G * dy * cn[1];
                        const float dGy = -G * dy * cn[2] - G * dx * cn[1];
                        const double v0 = (double)(dLG * dGx), v1 = (double)(dLG * dGy);
                        const double v2 = (double)fabsf(dLG * dGx), v3 = (double)fabsf(dLG * dGy);
                        const double v4 = (double)(-0.5f * G * dx * dx * dLG);
                        const double v5 = (double)(-G * dx * dy * dLG);
                        const double v6 = (double)(-0.5f * G * dy * dy * dLG);
                        const double v7 = (double)(G * dLa);
                        const double v8 = (double)dLa;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_uv[2 * (size_t)id] += v0;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_uv[2 * (size_t)id + 1] += v1;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_abs[2 * (size_t)id] += v2;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_abs[2 * (size_t)id + 1] += v3;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_con[3 * (size_t)id] += v4;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_con[3 * (size_t)id + 1] += v5;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_con[3 * (size_t)id + 2] += v6;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a_op[id] += v7;
                        if (bias) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                            a_bias[id] += v8;
                            done = T < 0.0001f;
                        }
                    }
                }
        }
    }
    for (size_t i = 0; i < NP; ++i) {
        dL_duv[2 * i] += (float)a_uv[2 * i]; dL_duv[2 * i + 1] += (float)a_uv[2 * i + 1];
        dL_dabs_uv[2 * i] += (float)a_abs[2 * i]; dL_dabs_uv[2 * i + 1] += (float)a_abs[2 * i + 1];
        for (int k = 0; k < 3; ++k) dL_dconic[3 * i + k] += (float)a_con[3 * i + k];
        dL_dopacity[i] += (float)a_op[i];
        if (bias && dL_dbias) dL_dbias[i] += (float)a_bias[i];
        for (int c = 0; c < C; ++c) dL_dfeature[i * C + c] += (float)a_f[i * C + c];
    }
    free(a_uv); free(a_abs); free(a_con); free(a_op); free(a_bias); free(a_f);
}

/* ------------------------------------------------------------------------------------------
 * Per-frame evaluation of the dynamic Gaussians ("next" row 8f-1).
 * Reference: src/dynamic_gaussian_with_base_point_cloud.py
 *   get_position :236-250  position + cubic-spline segment  c3 + c2 d + c1 d^2 + c0 d^3
 *                          (coeff [N,4,I,3]; segment index and d = t - knot are host scalars)
 *   get_rotation :184-198  normalize(rotation + sum_k poly_k * t^k + sum_m fourier_m * basis_m),
 *                          the two time-dependent sums are .detach()'ed (no gradient to them)
 *   get_opacity  :171-172  sigmoid ; get_scaling :175-177 exp
 * Pinned by tests/golden/dynamic_*.npz (generated from the reference methods).
 * ---------------------------------------------------------------------------------------- */
void oracle_dynamic_eval_forward(int P, int I, int seg, float d, const float *position, const float *cubic,
                                 const float *rotation, const float *rot_poly, const float *rot_fourier,
                                 const float *poly_basis /*[4]*/, const float *fourier_basis /*[8]*/,
                                 const float *opacity, const float *scaling,
                                 float *pos_t, float *rot_t, float *opa_t, float *scl_t) {
    const float d2 = d * d, d3 = d * d * d;
    for (int i = 0; i < P; ++i) {
        const float *c = cubic + (size_t)i * 4 * I * 3;
        for (int a = 0; a < 3; ++a) {
            const float c0 = c[(0 * I + seg) * 3 + a], c1 = c[(1 * I + seg) * 3 + a];
            const float c2 = c[(2 * I + seg) * 3 + a], c3 = c[(3 * I + seg) * 3 + a];
            float p = c3 + c2 * d;
            p = p + c1 * d2;
            p = p + c0 * d3;
            pos_t[3 * i + a] = p + position[3 * i + a];
        }
        float q[4];
        for (int a = 0; a < 4; ++a) {
            float sp = 0.f, sf = 0.f;
            for (int k = 0; k < 4; ++k) sp += rot_poly[(i * 4 + k) * 4 + a] * poly_basis[k];
            for (int m = 0; m < 8; ++m) sf += rot_fourier[(i * 8 + m) * 4 + a] * fourier_basis[m];
            q[a] = rotation[4 * i + a] + sp + sf;
        }
        float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (n < 1e-12f) n = 1e-12f;
        for (int a = 0; a < 4; ++a) rot_t[4 * i + a] = q[a] / n;
        opa_t[i] = 1.0f / (1.0f + expf(-opacity[i]));
        for (int a = 0; a < 3; ++a) scl_t[3 * i + a] = expf(scaling[3 * i + a]);
    }
}

/* gradients w.r.t. position, cubic (dense, zero outside the active segment), rotation, opacity, scaling */
void oracle_dynamic_eval_backward(int P, int I, int seg, float d, const float *rotation, const float *rot_poly,
                                  const float *rot_fourier, const float *poly_basis, const float *fourier_basis,
                                  const float *opacity, const float *scaling,
                                  const float *g_pos, const float *g_rot, const float *g_opa, const float *g_scl,
                                  float *d_position, float *d_cubic, float *d_rotation, float *d_opacity,
                                  float *d_scaling) {
    const float pw[4] = {d * d * d, d * d, d, 1.0f}; /* derivative of the segment value w.r.t. c0..c3 */
    memset(d_cubic, 0, sizeof(float) * (size_t)P * 4 * I * 3);
    for (int i = 0; i < P; ++i) {
        for (int a = 0; a < 3; ++a) {
            d_position[3 * i + a] = g_pos[3 * i + a];
            for (int k = 0; k < 4; ++k) d_cubic[((size_t)i * 4 + k) * I * 3 + seg * 3 + a] = g_pos[3 * i + a] * pw[k];
        }
        float q[4];
        for (int a = 0; a < 4; ++a) {
            float sp = 0.f, sf = 0.f;
            for (int k = 0; k < 4; ++k) sp += rot_poly[(i * 4 + k) * 4 + a] * poly_basis[k];
            for (int m = 0; m < 8; ++m) sf += rot_fourier[(i * 8 + m) * 4 + a] * fourier_basis[m];
            q[a] = rotation[4 * i + a] + sp + sf;
        }
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (n < 1e-12f) { /* F.normalize clamps the norm: plain scaling */
            for (int a = 0; a < 4; ++a) d_rotation[4 * i + a] = g_rot[4 * i + a] / 1e-12f;
        } else {
            float dot = 0.f;
            for (int a = 0; a < 4; ++a) dot += (q[a] / n) * g_rot[4 * i + a];
            for (int a = 0; a < 4; ++a) d_rotation[4 * i + a] = (g_rot[4 * i + a] - (q[a] / n) * dot) / n;
        }
        const float s = 1.0f / (1.0f + expf(-opacity[i]));
        d_opacity[i] = g_opa[i] * s * (1.0f - s);
        for (int a = 0; a < 3; ++a) d_scaling[3 * i + a] = g_scl[3 * i + a] * expf(scaling[3 * i + a]);
    }
}

/* ----------------------------------------------------------------------------------------
 * Densification statistics (test infrastructure like the rest of this file).
 *   batch reduction of the renderer      src/pointrix/renderer/dptr_ortho_enhanced.py:425-431
 *   accumulate_viewspace_grad            src/pointrix/optimizer/atlas_gs_optimizer.py:414-433
 *   update_structure (statistics part)   atlas_gs_optimizer.py:110-121
 *   generate_clone_mask / split_mask     atlas_gs_optimizer.py:199-251
 *   prune filter                         atlas_gs_optimizer.py:363-375
 * Pinned by tests/golden/densify_5000.npz (generated from those methods).
 * ---------------------------------------------------------------------------------------- */
/* one frame of the batch: viewspace_grad += tap * (sx, sy); visible |= radius > 0; radii = max(radii, radius) */
void oracle_densify_accumulate(int P, const int *radius, const float *tap, float sx, float sy, float *viewspace_grad,
                               unsigned char *visible, int *radii) {
    for (int i = 0; i < P; ++i) {
        viewspace_grad[2 * i] += tap[2 * i] * sx;
        viewspace_grad[2 * i + 1] += tap[2 * i + 1] * sy;
        if (radius[i] > 0) visible[i] = 1;
        if (radius[i] > radii[i]) radii[i] = radius[i];
    }
}

/* once per step, for the Gaussians visible in the batch */
void oracle_densify_update(int P, const unsigned char *visible, const float *viewspace_grad, const int *radii,
                           float *max_radii2D, float *pos_gradient_accum, float *denom) {
    for (int i = 0; i < P; ++i) {
        if (!visible[i]) continue;
        const float r = (float)radii[i];
        if (r > max_radii2D[i]) max_radii2D[i] = r;
        const float gx = viewspace_grad[2 * i], gy = viewspace_grad[2 * i + 1];
        pos_gradient_accum[i] += sqrtf(gx * gx + gy * gy);
        denom[i] += 1.0f;
    }
}

/* clone / split / prune decisions from the accumulated statistics; scaling and opacity are the RAW parameters
 * (exp / sigmoid applied here, as get_scaling / get_opacity do) */
void oracle_densify_masks(int P, const float *pos_gradient_accum, const float *denom, const float *max_radii2D,
                          const float *scaling_raw, const float *opacity_raw, float grad_threshold, float percent_dense,
                          float cameras_extent, float min_opacity, float size_threshold, unsigned char *clone,
                          unsigned char *split, unsigned char *prune) {
    for (int i = 0; i < P; ++i) {
        float g = pos_gradient_accum[i] / denom[i];
        if (isnan(g)) g = 0.0f;
        float smax = expf(scaling_raw[3 * i]);
        const float s1 = expf(scaling_raw[3 * i + 1]), s2 = expf(scaling_raw[3 * i + 2]);
        if (s1 > smax) smax = s1;
        if (s2 > smax) smax = s2;
        const float dense = percent_dense * cameras_extent;
        clone[i] = (fabsf(g) >= grad_threshold) && (smax <= dense);
        split[i] = (g >= grad_threshold) && (smax > dense);
        const float op = 1.0f / (1.0f + expf(-opacity_raw[i]));
        prune[i] = (op < min_opacity) || (size_threshold > 0.0f && (max_radii2D[i] > size_threshold || smax > 0.1f * cameras_extent));
    }
}

/* stream compaction of rows of `row_words` 32-bit words: dst = src[mask]; returns the number of rows kept */
int oracle_compact_rows(int P, const unsigned char *mask, int row_words, const unsigned int *src, unsigned int *dst) {
    int n = 0;
    for (int i = 0; i < P; ++i) {
        if (!mask[i]) continue;
        memcpy(dst + (size_t)n * row_words, src + (size_t)i * row_words, sizeof(unsigned int) * (size_t)row_words);
        ++n;
    }
    return n;
}

/* ----------------------------------------------------------------------------------------
 * K nearest neighbours.  The reference calls pytorch3d.ops.knn_points(points[None],
 * points[None], K=K+1) (src/geometry_utils.py:17-19; pytorch3d is an un-vendored dependency, absent from the
 * reference tree: parity unpinned, anchored on the call site and on knn_points' published contract: squared
 * Euclidean distances, the K smallest per query in ascending order, with their indices).  Brute force, exact;
 * ties between equal distances are broken by the smaller index.
 * ---------------------------------------------------------------------------------------- */
void oracle_knn_points(int N, const float *query, int M, const float *points, int K, float *dists, int *idx) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float *bd = dists + (size_t)i * K;
        int *bi = idx + (size_t)i * K;
        int n = 0;
        const float qx = query[3 * i], qy = query[3 * i + 1], qz = query[3 * i + 2];
        for (int j = 0; j < M; ++j) {
            const float dx = qx - points[3 * j], dy = qy - points[3 * j + 1], dz = qz - points[3 * j + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (n == K && !(d < bd[K - 1])) continue;
            int p = n < K ? n : K - 1;
            while (p > 0 && d < bd[p - 1]) {
                bd[p] = bd[p - 1];
                bi[p] = bi[p - 1];
                --p;
            }
            bd[p] = d;
            bi[p] = j;
            if (n < K) ++n;
        }
        for (int p = n; p < K; ++p) {
            bd[p] = 0.0f;  /* fewer than K points: knn_points pads with 0 / -1 */
            bi[p] = -1;
        }
    }
}
