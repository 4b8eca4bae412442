// Tile-based front-to-back alpha compositing, forward and backward (+ "enhanced" first-K index
// capture and the opacity-bias variant).  Reference semantics: src/alpha_blending.cu:16-249,
// src/alpha_blending_enhanced.cu:57-133, src/alpha_blending_with_bias.cu:88-89,211-214,259-261.
//
// MI355X design (differs from the reference's 256-thread tile block + per-pixel global atomics):
//   * one WAVE (64 lanes) per workgroup; a 16x16 tile is covered by 4/PPL independent waves, each
//     lane owning PPL pixels (rows y0+4p).  No cross-wave barrier, each wave stops as soon as its
//     own pixels are saturated (the reference waits for the whole tile).
//   * splats are fetched 64 at a time: lane j gathers splat j of the batch (id, uv, conic,
//     opacity, C features; the next batch is prefetched into registers while the current one is
//     composited) and parks it in LDS; the pixel loop reads them back as broadcast ds_read_b128.
//     Features come from the [P,C] row-major tensor directly (no host transpose/copy).
//   * backward: per-lane partial gradients of the PPL pixels are summed across the wave with
//     DPP row_shr/row_bcast adds and lane 63 issues ONE hardware float atomic per
//     (wave, splat, component) instead of one per (pixel, splat, component); splats that no pixel
//     of the wave touches are skipped wave-uniformly; the walk starts at the wave's largest
//     ncontrib instead of the end of the tile list.
#include "common.h"

struct BlendArgs {
    int P, C;          // C = row stride of feature / dL_dfeature
    int c0, cn;        // channel chunk [c0, c0+cn)
    const float2 *uv;
    const float *conic;
    const float *opacity;
    const float *feature;
    const float *bias;
    const int *idx_sorted;
    const int2 *tile_range;
    float bg;
    int W, H, gx;
    int K, trunc;
    // forward outputs
    float *out;        // [C,H,W]
    float *final_T;
    int *ncontrib;
    int *gs_idx;
    // backward inputs / outputs
    const float *dL_dout;
    float *dL_duv, *dL_dabs_uv, *dL_dconic, *dL_dopacity, *dL_dfeature, *dL_dbias;
};

template <int CH>
struct Splat {  // one gathered splat, held by one lane
    float u, v, a, b, c, o, bias;
    int id;
    float f[CH];
};

template <int CH, bool BIAS>
__device__ __forceinline__ void gather_splat(const BlendArgs &A, int pos, bool valid, Splat<CH> &s) {
    s.id = 0; s.u = s.v = s.a = s.b = s.c = s.o = s.bias = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) s.f[k] = 0.f;
    if (valid) {
        const int id = A.idx_sorted[pos];
        s.id = id;
        const float2 q = A.uv[id];
        s.u = q.x; s.v = q.y;
        s.a = A.conic[3 * id]; s.b = A.conic[3 * id + 1]; s.c = A.conic[3 * id + 2];
        s.o = A.opacity[id];
        if (BIAS) s.bias = A.bias[id];
        const float *f = A.feature + (size_t)id * A.C + A.c0;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < A.cn) s.f[k] = f[k];
    }
}

template <int CH>
struct SplatLDS {
    static constexpr int CHP = (CH + 3) & ~3;
    float4 g0[WAVE];  // u v a b
    float4 g1[WAVE];  // c o bias id(bits)
    float f[WAVE * CHP];
};

template <int CH>
__device__ __forceinline__ void park_splat(SplatLDS<CH> &L, int lane, const Splat<CH> &s) {
    L.g0[lane] = make_float4(s.u, s.v, s.a, s.b);
    L.g1[lane] = make_float4(s.c, s.o, s.bias, __int_as_float(s.id));
    constexpr int CHP = SplatLDS<CH>::CHP;
#pragma unroll
    for (int k = 0; k < CHP; k += 4) {
        float4 v;
        v.x = k + 0 < CH ? s.f[k + 0] : 0.f;
        v.y = k + 1 < CH ? s.f[k + 1] : 0.f;
        v.z = k + 2 < CH ? s.f[k + 2] : 0.f;
        v.w = k + 3 < CH ? s.f[k + 3] : 0.f;
        *reinterpret_cast<float4 *>(&L.f[lane * CHP + k]) = v;
    }
}

template <int CH>
__device__ __forceinline__ void read_feat(const SplatLDS<CH> &L, int j, float f[CH]) {
    constexpr int CHP = SplatLDS<CH>::CHP;
#pragma unroll
    for (int k = 0; k < CHP; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(&L.f[j * CHP + k]);
        if (k + 0 < CH) f[k + 0] = v.x;
        if (k + 1 < CH) f[k + 1] = v.y;
        if (k + 2 < CH) f[k + 2] = v.z;
        if (k + 3 < CH) f[k + 3] = v.w;
    }
}

// ------------------------------------------------------------------ forward
template <int CH, int PPL, bool ENH, bool BIAS>
__global__ void __launch_bounds__(WAVE)
blend_fwd_kernel(const BlendArgs A) {
    constexpr int WPT = 4 / PPL;
    __shared__ SplatLDS<CH> L;
    const int lane = threadIdx.x;
    const int tile = blockIdx.x / WPT, sub = blockIdx.x - tile * WPT;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int px = tx * TILE + (lane & 15);
    const int pyb = ty * TILE + sub * 4 * PPL + (lane >> 4);
    const float pxf = (float)px;

    float T[PPL], F[PPL][CH], pyf[PPL];
    int last[PPL], layer[PPL];
    bool done[PPL], inside[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int py = pyb + 4 * p;
        pyf[p] = (float)py;
        inside[p] = (px < A.W) && (py < A.H);
        done[p] = !inside[p];
        T[p] = 1.0f; last[p] = 0; layer[p] = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) F[p][k] = 0.f;
    }
    const int2 range = A.tile_range[tile];
    const int n = range.y - range.x;

    Splat<CH> nxt;
    gather_splat<CH, BIAS>(A, range.x + lane, lane < n, nxt);
    for (int base = 0; base < n; base += WAVE) {
        bool alld = true;
#pragma unroll
        for (int p = 0; p < PPL; ++p) alld = alld && done[p];
        if (__all(alld)) break;
        __syncthreads();  // previous batch fully consumed (single wave: s_barrier is ~free)
        park_splat<CH>(L, lane, nxt);
        const int nb = imin_(WAVE, n - base);
        const int nbase = base + WAVE;
        gather_splat<CH, BIAS>(A, range.x + nbase + lane, nbase + lane < n, nxt);  // prefetch
        __syncthreads();
        for (int j = 0; j < nb; ++j) {
            const float4 g0 = L.g0[j], g1 = L.g1[j];
            float alpha[PPL];
            bool ok[PPL];
            bool any_ok = false;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const float dx = g0.x - pxf, dy = g0.y - pyf[p];
                const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
                float araw = g1.y * __expf(power);
                if (BIAS) araw = araw + g1.z;
                alpha[p] = fminf(0.99f, araw);
                ok[p] = !done[p] && !(power > 0.f) && !(alpha[p] < (1.0f / 255.0f));
                any_ok = any_ok || ok[p];
            }
            if (!__any(any_ok)) continue;
            float f[CH];
            read_feat<CH>(L, j, f);
            const int id = __float_as_int(g1.w);
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const float nT = T[p] * (1.f - alpha[p]);
                const bool sat = ok[p] && (nT < 0.0001f);
                const bool app = ok[p] && !sat;
                done[p] = done[p] || sat;
                const float w = app ? alpha[p] * T[p] : 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) F[p][k] += f[k] * w;
                T[p] = app ? nT : T[p];
                last[p] = app ? base + j + 1 : last[p];
                if (ENH) {
                    if (app && (A.trunc || layer[p] < A.K)) {
                        const size_t pix = (size_t)A.W * (size_t)(pyb + 4 * p) + px;
                        A.gs_idx[pix * A.K + layer[p]] = id;
                        layer[p]++;
                        if (A.trunc && layer[p] >= A.K) done[p] = true;
                    }
                }
            }
        }
    }
    const size_t HW = (size_t)A.H * A.W;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        if (!inside[p]) continue;
        const size_t pix = (size_t)A.W * (size_t)(pyb + 4 * p) + px;
        A.final_T[pix] = T[p];
        A.ncontrib[pix] = last[p];
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < A.cn) A.out[(size_t)(A.c0 + k) * HW + pix] = F[p][k] + T[p] * A.bg;
    }
}

// ------------------------------------------------------------------ backward
template <int CH, int PPL, bool BIAS>
__global__ void __launch_bounds__(WAVE)
blend_bwd_kernel(const BlendArgs A) {
    constexpr int WPT = 4 / PPL;
    __shared__ SplatLDS<CH> L;
    const int lane = threadIdx.x;
    const int tile = blockIdx.x / WPT, sub = blockIdx.x - tile * WPT;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int px = tx * TILE + (lane & 15);
    const int pyb = ty * TILE + sub * 4 * PPL + (lane >> 4);
    const float pxf = (float)px;
    const size_t HW = (size_t)A.H * A.W;

    float Tf[PPL], T[PPL], pyf[PPL], bgdot[PPL], acc[PPL][CH], gp[PPL][CH];
    int last[PPL];
    bool done[PPL];
    int maxlast = 0;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int py = pyb + 4 * p;
        pyf[p] = (float)py;
        const bool inside = (px < A.W) && (py < A.H);
        const size_t pix = (size_t)A.W * (size_t)py + px;
        Tf[p] = inside ? A.final_T[pix] : 0.f;
        T[p] = Tf[p];
        last[p] = inside ? A.ncontrib[pix] : 0;
        done[p] = !inside;
        maxlast = imax_(maxlast, last[p]);
        float bd = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            acc[p][k] = 0.f;
            gp[p][k] = (inside && k < A.cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
            if (k < A.cn) bd += A.bg * gp[p][k];
        }
        bgdot[p] = bd;
    }
    const int2 range = A.tile_range[tile];
    const int n = imin_(range.y - range.x, wave_max_i(maxlast));  // entries >= max ncontrib are never used
    if (n <= 0) return;

    // reverse walk: batch b covers list positions q = top-64b-lane  (top = n-1)
    Splat<CH> nxt;
    {
        const int q = n - 1 - lane;
        gather_splat<CH, BIAS>(A, range.x + q, q >= 0, nxt);
    }
    for (int top = n - 1; top >= 0; top -= WAVE) {
        __syncthreads();
        park_splat<CH>(L, lane, nxt);
        const int nb = imin_(WAVE, top + 1);
        {
            const int q = top - WAVE - lane;
            gather_splat<CH, BIAS>(A, range.x + q, q >= 0, nxt);  // prefetch
        }
        __syncthreads();
        for (int j = 0; j < nb; ++j) {
            const int q = top - j;  // 0-based list position == reference's `contributor` after decrement
            const float4 g0 = L.g0[j], g1 = L.g1[j];
            float alpha[PPL], G[PPL], dx[PPL], dy[PPL];
            bool ok[PPL];
            bool any_ok = false;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                dx[p] = g0.x - pxf; dy[p] = g0.y - pyf[p];
                const float power = -0.5f * (g0.z * dx[p] * dx[p] + g1.x * dy[p] * dy[p]) - g0.w * dx[p] * dy[p];
                G[p] = __expf(power);
                float araw = g1.y * G[p];
                if (BIAS) araw = araw + g1.z;
                alpha[p] = fminf(0.99f, araw);
                ok[p] = !done[p] && (q < last[p]) && !(power > 0.f) && !(alpha[p] < (1.0f / 255.0f));
                any_ok = any_ok || ok[p];
            }
            if (!__any(any_ok)) continue;
            float f[CH];
            read_feat<CH>(L, j, f);
            float s_ux = 0.f, s_uy = 0.f, s_ax = 0.f, s_ay = 0.f, s_ca = 0.f, s_cb = 0.f, s_cc = 0.f, s_o = 0.f, s_b = 0.f;
            float s_f[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) s_f[k] = 0.f;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                if (ok[p]) {
                    const float a = alpha[p];
                    T[p] = T[p] / (1.f - a);
                    const float w = a * T[p];
                    float dLa = 0.f;
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        dLa += (f[k] - acc[p][k]) * gp[p][k];
                        s_f[k] += w * gp[p][k];
                        acc[p][k] = a * f[k] + (1.f - a) * acc[p][k];  // == reference's deferred update
                    }
                    dLa *= T[p];
                    dLa += (-Tf[p] / (1.f - a)) * bgdot[p];
                    const float dLG = g1.y * dLa;
                    const float gx_ = -G[p] * dx[p] * g0.z - G[p] * dy[p] * g0.w;
                    const float gy_ = -G[p] * dy[p] * g1.x - G[p] * dx[p] * g0.w;
                    s_ux += dLG * gx_; s_uy += dLG * gy_;
                    s_ax += fabsf(dLG * gx_); s_ay += fabsf(dLG * gy_);
                    s_ca += -0.5f * G[p] * dx[p] * dx[p] * dLG;
                    s_cb += -G[p] * dx[p] * dy[p] * dLG;
                    s_cc += -0.5f * G[p] * dy[p] * dy[p] * dLG;
                    s_o += G[p] * dLa;
                    if (BIAS) {
                        s_b += dLa;
                        done[p] = T[p] < 0.0001f;
                    }
                }
            }
            // wave reduction -> lane 63, one atomic per component
            s_ux = wave_sum_to_lane63(s_ux); s_uy = wave_sum_to_lane63(s_uy);
            s_ax = wave_sum_to_lane63(s_ax); s_ay = wave_sum_to_lane63(s_ay);
            s_ca = wave_sum_to_lane63(s_ca); s_cb = wave_sum_to_lane63(s_cb); s_cc = wave_sum_to_lane63(s_cc);
            s_o = wave_sum_to_lane63(s_o);
            if (BIAS) s_b = wave_sum_to_lane63(s_b);
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < A.cn) s_f[k] = wave_sum_to_lane63(s_f[k]);
            if (lane == 63) {
                const int id = __float_as_int(g1.w);
                atomic_add_f32(A.dL_duv + 2 * id, s_ux);
                atomic_add_f32(A.dL_duv + 2 * id + 1, s_uy);
                atomic_add_f32(A.dL_dabs_uv + 2 * id, s_ax);
                atomic_add_f32(A.dL_dabs_uv + 2 * id + 1, s_ay);
                atomic_add_f32(A.dL_dconic + 3 * id, s_ca);
                atomic_add_f32(A.dL_dconic + 3 * id + 1, s_cb);
                atomic_add_f32(A.dL_dconic + 3 * id + 2, s_cc);
                atomic_add_f32(A.dL_dopacity + id, s_o);
                if (BIAS) atomic_add_f32(A.dL_dbias + id, s_b);
                float *df = A.dL_dfeature + (size_t)id * A.C + A.c0;
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k < A.cn) atomic_add_f32(df + k, s_f[k]);
            }
        }
    }
}

// ================================================================== launch tables
template <int CH, int PPL>
static int launch_fwd(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    const dim3 grid((unsigned)(T * (4 / PPL))), block(WAVE);
    if (enh) {
        if (bias) SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, PPL, true, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, PPL, true, false>), grid, block, 0, s, A);
    } else {
        if (bias) SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, PPL, false, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, PPL, false, false>), grid, block, 0, s, A);
    }
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <int CH, int PPL>
static int launch_bwd(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    const dim3 grid((unsigned)(T * (4 / PPL))), block(WAVE);
    if (bias) SPLAT_LAUNCH("blend_bwd", (blend_bwd_kernel<CH, PPL, true>), grid, block, 0, s, A);
    else SPLAT_LAUNCH("blend_bwd", (blend_bwd_kernel<CH, PPL, false>), grid, block, 0, s, A);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

static int fwd_chunk(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    const int cn = A.cn;
    if (cn <= 1) return launch_fwd<1, 1>(A, T, enh, bias, s);
    if (cn <= 3) return launch_fwd<3, 1>(A, T, enh, bias, s);
    if (cn <= 8) return launch_fwd<8, 1>(A, T, enh, bias, s);
    if (cn <= 16) return launch_fwd<16, 1>(A, T, enh, bias, s);
    if (cn <= 24) return launch_fwd<24, 1>(A, T, enh, bias, s);
    return launch_fwd<32, 1>(A, T, enh, bias, s);
}

static int bwd_chunk(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    const int cn = A.cn;
    if (cn <= 1) return launch_bwd<1, 4>(A, T, bias, s);
    if (cn <= 3) return launch_bwd<3, 4>(A, T, bias, s);
    if (cn <= 8) return launch_bwd<8, 2>(A, T, bias, s);
    if (cn <= 16) return launch_bwd<16, 2>(A, T, bias, s);
    if (cn <= 24) return launch_bwd<24, 1>(A, T, bias, s);
    return launch_bwd<32, 1>(A, T, bias, s);
}

// ================================================================== C ABI
extern "C" int splat_alpha_blending_forward(int P, int C, const float *uv, const float *conic, const float *opacity,
                                            const float *feature, const float *opacity_bias,
                                            const int32_t *idx_sorted, const int32_t *tile_range, float bg, int W,
                                            int H, int K, int enable_truncation, float *out, float *final_T,
                                            int32_t *ncontrib, int32_t *gs_idx, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && C >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(tile_range && out && final_T && ncontrib, "null pointer");
    SPLAT_CHECK_ARG(P == 0 || (uv && conic && opacity && feature), "null pointer");
    const bool enh = (gs_idx != nullptr) && K > 0;
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity; A.feature = feature; A.bias = opacity_bias;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bg = bg; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.K = enh ? K : 0; A.trunc = enable_truncation ? 1 : 0;
    A.out = out; A.final_T = final_T; A.ncontrib = ncontrib; A.gs_idx = gs_idx;
    const int T = A.gx * ((H + TILE - 1) / TILE);
    for (int c0 = 0; c0 < C; c0 += 32) {  // reference chunking: src/alpha_blending.cu:287-394
        A.c0 = c0;
        A.cn = C - c0 > 32 ? 32 : C - c0;
        const int rc = fwd_chunk(A, T, enh, opacity_bias != nullptr, (hipStream_t)stream);
        if (rc != SPLAT_OK) return rc;
    }
    return SPLAT_OK;
}

extern "C" int splat_alpha_blending_backward(int P, int C, const float *uv, const float *conic, const float *opacity,
                                             const float *feature, const float *opacity_bias,
                                             const int32_t *idx_sorted, const int32_t *tile_range, float bg, int W,
                                             int H, const float *final_T, const int32_t *ncontrib,
                                             const float *dL_dout, float *dL_duv, float *dL_dabs_uv, float *dL_dconic,
                                             float *dL_dopacity, float *dL_dfeature, float *dL_dopacity_bias,
                                             splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && C >= 1 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(uv && conic && opacity && feature && idx_sorted && tile_range && final_T && ncontrib && dL_dout,
                    "null pointer");
    SPLAT_CHECK_ARG(dL_duv && dL_dabs_uv && dL_dconic && dL_dopacity && dL_dfeature, "null gradient pointer");
    SPLAT_CHECK_ARG(!opacity_bias || dL_dopacity_bias, "bias given without dL_dopacity_bias");
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity; A.feature = feature; A.bias = opacity_bias;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bg = bg; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.final_T = const_cast<float *>(final_T); A.ncontrib = const_cast<int *>(ncontrib);
    A.dL_dout = dL_dout;
    A.dL_duv = dL_duv; A.dL_dabs_uv = dL_dabs_uv; A.dL_dconic = dL_dconic; A.dL_dopacity = dL_dopacity;
    A.dL_dfeature = dL_dfeature; A.dL_dbias = dL_dopacity_bias;
    const int T = A.gx * ((H + TILE - 1) / TILE);
    for (int c0 = 0; c0 < C; c0 += 32) {  // reference chunking: src/alpha_blending.cu:440-577
        A.c0 = c0;
        A.cn = C - c0 > 32 ? 32 : C - c0;
        const int rc = bwd_chunk(A, T, opacity_bias != nullptr, (hipStream_t)stream);
        if (rc != SPLAT_OK) return rc;
    }
    return SPLAT_OK;
}
