// Internal helpers shared by the HIP translation units of libsplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/splat_hip.h"

#define TILE 16
#define TILE_PIX 256
#define WAVE 64

// ---------------------------------------------------------------- errors
void splat_set_error(const char *fmt, ...);

#define SPLAT_CHECK_ARG(cond, msg)                     \
    do {                                               \
        if (!(cond)) {                                 \
            splat_set_error("%s: %s", __func__, msg);  \
            return SPLAT_E_ARG;                        \
        }                                              \
    } while (0)

#define SPLAT_CHECK_HIP(expr)                                                          \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            splat_set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(e_));   \
            return SPLAT_E_LAUNCH;                                                     \
        }                                                                              \
    } while (0)

bool splat_deterministic();   // splat_set_deterministic (runtime.hip)
// splat_set_option keys (runtime.hip): kernel selection through the ABI, read at launch time
enum { SPLAT_OPT_BWD_QUARTERS = 0, SPLAT_OPT_BWD_KERNEL_DPP, SPLAT_OPT_SETS_STD, SPLAT_OPT_BIN_SLOT_KEYS, SPLAT_OPT_COUNT };
int splat_option(int id);

// ---------------------------------------------------------------- profiled launches
// When splat_profile_enable(1) was called, every kernel launch is bracketed by two hipEvents
// recorded on the launch stream; splat_profile_read() sums them per kernel-name prefix.
bool splat_profile_on();
void splat_profile_begin(const char *name, hipStream_t s);
void splat_profile_end(hipStream_t s);

struct ProfiledLaunch {
    hipStream_t s;
    bool on;
    ProfiledLaunch(const char *name, hipStream_t st) : s(st), on(splat_profile_on()) {
        if (on) splat_profile_begin(name, s);
    }
    ~ProfiledLaunch() {
        if (on) splat_profile_end(s);
    }
};

#define SPLAT_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                        \
    do {                                                                                   \
        ProfiledLaunch pl_(name, stream);                                                  \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);               \
    } while (0)

#define SPLAT_POST_LAUNCH() SPLAT_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }

// Tile rectangle of a splat: float arithmetic, truncation toward zero, clamp to [0, grid]
// (reference: include/utils.h:17-37).
__device__ __forceinline__ void tile_rect(float px, float py, int r, int gx, int gy, int &x0, int &y0, int &x1,
                                          int &y1) {
    const float fr = (float)r;
    x0 = imin_(gx, imax_(0, (int)((px - fr) / (float)TILE)));
    y0 = imin_(gy, imax_(0, (int)((py - fr) / (float)TILE)));
    x1 = imin_(gx, imax_(0, (int)((((px + fr) + (float)TILE) - 1.0f) / (float)TILE)));
    y1 = imin_(gy, imax_(0, (int)((((py + fr) + (float)TILE) - 1.0f) / (float)TILE)));
}

// ---- wave64 reductions with DPP (gfx9 row_shr / row_bcast); result valid in lane 63 ----
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

// Sum over the 64 lanes; the total lands in lane 63 (other lanes hold partials).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_f<0x111, 0xf, 0xf, true>(v);  // row_shr:1
    v += dpp_f<0x112, 0xf, 0xf, true>(v);  // row_shr:2
    v += dpp_f<0x114, 0xf, 0xf, true>(v);  // row_shr:4
    v += dpp_f<0x118, 0xf, 0xf, true>(v);  // row_shr:8  -> lane 15 of every row = row total
    v += dpp_f<0x142, 0xa, 0xf, true>(v);  // row_bcast:15 into rows 1,3
    v += dpp_f<0x143, 0xc, 0xf, true>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
    return v;
}

__device__ __forceinline__ float wave_sum_bcast(float v) {
    v = wave_sum_to_lane63(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = imax_(v, __shfl_xor(v, o));
    return v;
}

// hardware float atomic add, agent scope, no return (global_atomic_add_f32)
__device__ __forceinline__ void atomic_add_f32(float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- conservative reach test (tile_cull of blend.hip; reach masks of binning.hip)
// Can the splat reach alpha >= 1/255 at any pixel centre of the block [bx0,bx1]x[by0,by1]?
// alpha >= 1/255 needs q(d) = d^T Q d <= tau = 2 ln(255 o).  Two conservative stages (never false for a
// splat that contributes; every bound is inflated by the rounding error it can carry):
//   1. axis-aligned box of the ellipse {q <= tau} against the block;
//   2. exact: the minimum of the convex q over the block rectangle (centre inside, else on an edge).
// The block-independent part (cull_params: log, two sqrt, three rcp) is evaluated once per Gaussian and frame.
struct CullP {
    float hx, hy;   // half extents of the ellipse's bounding box (hx < 0: the splat never contributes)
    float tauq;     // inflated tau for the exact test (+inf: keep whenever the box test passes)
    float ia, ic;   // 1/A, 1/C
};

__device__ __forceinline__ CullP cull_params(float a, float b, float c, float o) {
    CullP p;
    const float INF = __builtin_inff();
    p.hx = INF; p.hy = INF; p.tauq = INF; p.ia = 0.f; p.ic = 0.f;  // degenerate conic: always keep
    const float t = 255.f * o;
    if (t < 0.999f) {  // alpha <= o < 1/255 everywhere
        p.hx = -1.f;
        return p;
    }
    const float det = a * c - b * b;
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return p;
    const float relerr = 4e-7f * (a * c + b * b) / det;  // rounding bound of det (cancellation)
    if (!(relerr < 0.25f)) return p;
    const float tau0 = fmaxf(2.f * __logf(t), 0.f);
    const float tau = tau0 * (1.f + 2.f * relerr) * 1.002f + 2e-3f;
    const float inv = 1.f / det;
    p.hx = sqrtf(tau * c * inv) * 1.001f + 0.01f;
    p.hy = sqrtf(tau * a * inv) * 1.001f + 0.01f;
    p.tauq = tau * 1.002f + 2e-3f;
    p.ia = 1.f / a;
    p.ic = 1.f / c;
    return p;
}

__device__ __forceinline__ bool cull_test(float u, float v, float a, float b, float c, const CullP &p, float bx0,
                                          float bx1, float by0, float by1) {
    const float dx0 = bx0 - u, dx1 = bx1 - u, dy0 = by0 - v, dy1 = by1 - v;  // block relative to the centre
    const float ddx = fmaxf(fmaxf(dx0, -dx1), 0.f);
    const float ddy = fmaxf(fmaxf(dy0, -dy1), 0.f);
    if (!((ddx <= p.hx) && (ddy <= p.hy))) return false;
    if (ddx == 0.f && ddy == 0.f) return true;  // centre inside the block
    // q is convex with its minimum at the centre, and the centre lies outside the block: the block minimum sits on an
    // edge that FACES the centre (the segment from the centre to any point of the block enters it through such an
    // edge, at a point where q is smaller) -- one vertical and/or one horizontal edge; on each, clamp the
    // unconstrained minimiser of the edge line
    const float INF = __builtin_inff();
    const float xe = dx0 > 0.f ? dx0 : dx1, ye = dy0 > 0.f ? dy0 : dy1;  // the facing edges (meaningful where dd* > 0)
    const float ys = fminf(fmaxf(-b * xe * p.ic, dy0), dy1), xs = fminf(fmaxf(-b * ye * p.ia, dx0), dx1);
    const float qx = a * xe * xe + 2.f * b * xe * ys + c * ys * ys;
    const float qy = a * xs * xs + 2.f * b * xs * ye + c * ye * ye;
    const float qmin = fminf(ddx > 0.f ? qx : INF, ddy > 0.f ? qy : INF);
    // q is evaluated with ~1e-6 relative error of its largest term; the terms are bounded by (a+c+2|b|) * r^2
    const float r2 = fmaxf(dx0 * dx0, dx1 * dx1) + fmaxf(dy0 * dy0, dy1 * dy1);
    const float qerr = 4e-6f * (a + c + 2.f * fabsf(b)) * r2;
    return qmin <= p.tauq + qerr;
}

// ---------------------------------------------------------------- pair records of the atomic-free blend backward
// One record per (tile, splat) pair at its Gaussian-major slot: r[] = [ux uy ca cb cc o | ax ay (ABS) | bias (BIAS) |
// CH feature terms] -- written by the tile kernels of blend.hip, summed per Gaussian by pair_reduce (blend.hip) or by
// the frame-batch Gaussian-side backward (preprocess.hip).
template <bool ABS, bool BIAS>
struct GradLayout {
    static constexpr int NG = 6 + (ABS ? 2 : 0) + (BIAS ? 1 : 0);
    static constexpr int I_ABS = 6;
    static constexpr int I_BIAS = 6 + (ABS ? 2 : 0);
};
// stride of a pair record in floats: the used floats rounded up to whole 16-byte chunks (a Gaussian's records are
// contiguous and streamed with float4 loads; padding every record to a 64-byte sector cost 30 % more traffic)
#define PAIR_STRIDE(nc) (((nc) + 3) & ~3)
// SETS records (the renderer's three feature sets in one backward pass, blend.hip): [ux uy ca cb | cc o ax ay | tx ty 0 0 |
// dL_dfeature of the row's channels 0 .. C-1] -- the geometry part fills three whole 16-byte chunks, so the channel gradients
// start on a chunk boundary (the tile kernel stores them as float4, the Gaussian-side walk reads whole chunks anyway; for the
// renderer's 23 channels the stride stays PAIR_STRIDE(12 + 23) = 36 floats)
constexpr int SETS_NG = 12;
// Workgroups are handed to the 8 XCDs round-robin by linear block id, and every XCD has its own L2.  Neighbouring
// tiles gather largely the same packed records (a splat touches ~4 tiles), so runs of BLEND_XCD_RUN consecutive tiles
// of the row-major order go to the same XCD (its consecutive blocks, i.e. roughly concurrently resident), and the runs
// are dealt round-robin so that every XCD sees the whole image (a contiguous band per XCD halves the HBM reads as
// well but leaves the XCDs that own the image borders idle early).  Block b = XCD b & 7, its (b >> 3)-th block.
#ifndef BLEND_XCD_RUN
#define BLEND_XCD_RUN 64   // a run is about one tile row of a 480p frame plus the start of the next (54 tiles per row)
#endif
__device__ __forceinline__ int xcd_tile(int b, int T) {
#if BLEND_XCD_RUN > 1
    constexpr int S = BLEND_XCD_RUN;
    const int full = (T / (8 * S)) * (8 * S);  // tiles covered by complete rounds of 8 runs; the tail maps linearly
    if (b >= full) return b;
    const int x = b & 7, q = b >> 3;
    return ((q / S) * 8 + x) * S + (q % S);
#else
    return b;
#endif
}


