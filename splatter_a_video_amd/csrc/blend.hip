// Tile-based front-to-back alpha compositing, forward and backward (+ "enhanced" first-K index
// capture and the opacity-bias variant).  Reference semantics: src/alpha_blending.cu:16-249,
// src/alpha_blending_enhanced.cu:57-133, src/alpha_blending_with_bias.cu:88-89,211-214,259-261.
//
// MI355X design (differs from the reference's 256-thread tile block + per-pixel global atomics):
//   * one WAVE (64 lanes) per workgroup.  A 16x16 tile is split into BWxBH pixel blocks, one
//     wave each, every lane owning PPL = BW*BH/64 pixels.  No cross-wave barrier; a wave stops
//     as soon as its own pixels are saturated.
//   * the tile's depth-sorted splat list is consumed 64 entries at a time: lane j gathers entry
//     j (id, uv, conic, opacity, C features from the [P,C] row-major tensor -- no host
//     transpose), two batches ahead for the id and one ahead for the payload, so no load sits
//     on the critical path of the pixel loop.
//   * CULL + COMPACT at staging: a splat can only reach alpha >= 1/255 inside the ellipse
//     d^T Q d <= 2 ln(255 o); its axis-aligned box (inflated by a rounding bound) is tested against
//     the wave's pixel block, survivors are compacted into LDS with ballot/popcount, order
//     preserved.  The pixel loop therefore only visits splats that can touch the block (the
//     reference visits every splat of the 3-sigma tile list for all 256 pixels).  The test is
//     conservative, so results are unchanged.
//   * pixel loop: splat records are read back as broadcast ds_read_b128.
//   * backward: per-lane partial gradients are summed across the wave with DPP row_shr /
//     row_bcast adds and lane 63 issues ONE hardware float atomic per (wave, splat, component)
//     instead of one per (pixel, splat, component); the walk starts at the wave's largest
//     ncontrib instead of the end of the tile list; 1/(1-alpha) is one v_rcp_f32.
#include <stdlib.h>

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
    int no_atomics;    // experiment switch (timing only)
    // forward outputs
    float *out;        // [C,H,W]
    float *final_T;
    int *ncontrib;
    int *gs_idx;
    // backward inputs / outputs
    const float *dL_dout;
    float *dL_duv, *dL_dabs_uv, *dL_dconic, *dL_dopacity, *dL_dfeature, *dL_dbias;
    // atomic-free backward: per-(tile,splat) partial sums + inverse pair map
    float *pair_buf;        // [M, NC] partial gradients in sorted-pair order
    const int *goff_incl;   // [P] inclusive prefix of tiles per Gaussian
    const int *inv_pos;     // [M] pair slot -> sorted position
    int accumulate;         // reduce: add to the geometry gradients (channel chunks > 0)
};

template <int CH>
struct Splat {  // one gathered list entry, held by one lane
    float u, v, a, b, c, o, bias;
    int id;
    float f[CH];
};

template <int CH, bool BIAS>
__device__ __forceinline__ void gather_splat(const BlendArgs &A, int id, bool valid, Splat<CH> &s) {
    s.id = id; s.u = s.v = s.a = s.b = s.c = s.o = s.bias = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) s.f[k] = 0.f;
    if (valid) {
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

// Can the splat reach alpha >= 1/255 anywhere in the pixel block [bx0,bx1]x[by0,by1]?
// Conservative: never false for a splat that contributes.
__device__ __forceinline__ bool splat_touches(float u, float v, float a, float b, float c, float o, float bx0,
                                              float bx1, float by0, float by1) {
    const float t = 255.f * o;
    if (t < 0.999f) return false;  // alpha <= o < 1/255 everywhere
    const float det = a * c - b * b;
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return true;
    const float relerr = 4e-7f * (a * c + b * b) / det;  // rounding bound of det (cancellation)
    if (!(relerr < 0.25f)) return true;
    const float tau = fmaxf(2.f * __logf(t), 0.f) * (1.f + 2.f * relerr) * 1.002f + 2e-3f;
    const float inv = 1.f / det;
    const float hx = sqrtf(tau * c * inv) * 1.001f + 0.01f;
    const float hy = sqrtf(tau * a * inv) * 1.001f + 0.01f;
    const float ddx = fmaxf(fmaxf(bx0 - u, u - bx1), 0.f);
    const float ddy = fmaxf(fmaxf(by0 - v, v - by1), 0.f);
    return (ddx <= hx) && (ddy <= hy);
}

template <int CH>
struct SplatLDS {
    static constexpr int CHP = (CH + 3) & ~3;
    float4 g0[WAVE];  // u v a b
    float4 g1[WAVE];  // c o bias id(bits)
    int q[WAVE];      // list position of the entry
    float f[WAVE * CHP];
};

template <int CH>
__device__ __forceinline__ void park_splat(SplatLDS<CH> &L, int slot, int q, const Splat<CH> &s) {
    L.g0[slot] = make_float4(s.u, s.v, s.a, s.b);
    L.g1[slot] = make_float4(s.c, s.o, s.bias, __int_as_float(s.id));
    L.q[slot] = q;
    constexpr int CHP = SplatLDS<CH>::CHP;
#pragma unroll
    for (int k = 0; k < CHP; k += 4) {
        float4 v;
        v.x = k + 0 < CH ? s.f[k + 0] : 0.f;
        v.y = k + 1 < CH ? s.f[k + 1] : 0.f;
        v.z = k + 2 < CH ? s.f[k + 2] : 0.f;
        v.w = k + 3 < CH ? s.f[k + 3] : 0.f;
        *reinterpret_cast<float4 *>(&L.f[slot * CHP + k]) = v;
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

// cull + order-preserving compaction of one gathered batch into LDS; returns the survivor count
template <int CH, bool BIAS>
__device__ __forceinline__ int stage_batch(SplatLDS<CH> &L, int lane, const Splat<CH> &s, bool valid, int q,
                                           float bx0, float bx1, float by0, float by1) {
    bool keep = valid;
    if (!BIAS) keep = keep && splat_touches(s.u, s.v, s.a, s.b, s.c, s.o, bx0, bx1, by0, by1);
    const unsigned long long m = __ballot(keep);
    const int slot = __popcll(m & ((1ull << lane) - 1ull));
    if (keep) park_splat<CH>(L, slot, q, s);
    return __popcll(m);
}

// pixel-block geometry of a wave
template <int BW, int BH>
struct Block {
    static constexpr int PPL = BW * BH / WAVE;
    static constexpr int ROWS = WAVE / BW;        // rows covered by one pass of the 64 lanes
    static constexpr int NBX = TILE / BW, NBY = TILE / BH;
    static constexpr int WPT = NBX * NBY;         // waves per tile
};

// ------------------------------------------------------------------ forward
template <int CH, int BW, int BH, bool ENH, bool BIAS>
__global__ void __launch_bounds__(WAVE)
blend_fwd_kernel(const BlendArgs A) {
    using B = Block<BW, BH>;
    constexpr int PPL = B::PPL;
    __shared__ SplatLDS<CH> L;
    const int lane = threadIdx.x;
    const int tile = blockIdx.x / B::WPT, sub = blockIdx.x - tile * B::WPT;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (sub % B::NBX) * BW, by = ty * TILE + (sub / B::NBX) * BH;
    const int px = bx + (lane % BW);
    const int pyb = by + (lane / BW);
    const float pxf = (float)px;
    const float bx0 = (float)bx, bx1 = (float)(bx + BW - 1), by0 = (float)by, by1 = (float)(by + BH - 1);

    float T[PPL], F[PPL][CH], pyf[PPL];
    int last[PPL], layer[PPL];
    bool done[PPL], inside[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int py = pyb + B::ROWS * p;
        pyf[p] = (float)py;
        inside[p] = (px < A.W) && (py < A.H);
        done[p] = !inside[p];
        T[p] = 1.0f; last[p] = 0; layer[p] = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) F[p][k] = 0.f;
    }
    const int2 range = A.tile_range[tile];
    const int n = range.y - range.x;

    // software pipeline: ids two batches ahead, payload one batch ahead
    int id1 = (lane < n) ? A.idx_sorted[range.x + lane] : 0;
    int id2 = (WAVE + lane < n) ? A.idx_sorted[range.x + WAVE + lane] : 0;
    Splat<CH> cur;
    gather_splat<CH, BIAS>(A, id1, lane < n, cur);
    for (int base = 0; base < n; base += WAVE) {
        bool alld = true;
#pragma unroll
        for (int p = 0; p < PPL; ++p) alld = alld && done[p];
        if (__all(alld)) break;
        __syncthreads();  // previous batch consumed (single wave: s_barrier is ~free)
        const int nb = stage_batch<CH, BIAS>(L, lane, cur, base + lane < n, base + lane, bx0, bx1, by0, by1);
        gather_splat<CH, BIAS>(A, id2, base + WAVE + lane < n, cur);
        id2 = (base + 2 * WAVE + lane < n) ? A.idx_sorted[range.x + base + 2 * WAVE + lane] : 0;
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
            const int q = L.q[j];
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
                last[p] = app ? q + 1 : last[p];
                if (ENH) {
                    if (app && (A.trunc || layer[p] < A.K)) {
                        const size_t pix = (size_t)A.W * (size_t)(pyb + B::ROWS * p) + px;
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
        const size_t pix = (size_t)A.W * (size_t)(pyb + B::ROWS * p) + px;
        A.final_T[pix] = T[p];
        A.ncontrib[pix] = last[p];
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < A.cn) A.out[(size_t)(A.c0 + k) * HW + pix] = F[p][k] + T[p] * A.bg;
    }
}

// ------------------------------------------------------------------ backward
template <int CH, int BW, int BH, bool BIAS>
__global__ void __launch_bounds__(WAVE)
blend_bwd_kernel(const BlendArgs A) {
    using B = Block<BW, BH>;
    constexpr int PPL = B::PPL;
    __shared__ SplatLDS<CH> L;
    const int lane = threadIdx.x;
    const int tile = blockIdx.x / B::WPT, sub = blockIdx.x - tile * B::WPT;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (sub % B::NBX) * BW, by = ty * TILE + (sub / B::NBX) * BH;
    const int px = bx + (lane % BW);
    const int pyb = by + (lane / BW);
    const float pxf = (float)px;
    const float bx0 = (float)bx, bx1 = (float)(bx + BW - 1), by0 = (float)by, by1 = (float)(by + BH - 1);
    const size_t HW = (size_t)A.H * A.W;

    float Tf[PPL], T[PPL], pyf[PPL], bgdot[PPL], acc[PPL][CH], gp[PPL][CH];
    int last[PPL];
    bool done[PPL];
    int maxlast = 0;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int py = pyb + B::ROWS * p;
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

    // reverse walk: the batch starting at `top` covers list positions q = top - lane
    int id1 = (n - 1 - lane >= 0) ? A.idx_sorted[range.x + n - 1 - lane] : 0;
    int id2 = (n - 1 - WAVE - lane >= 0) ? A.idx_sorted[range.x + n - 1 - WAVE - lane] : 0;
    Splat<CH> cur;
    gather_splat<CH, BIAS>(A, id1, n - 1 - lane >= 0, cur);
    for (int top = n - 1; top >= 0; top -= WAVE) {
        __syncthreads();
        const int nb = stage_batch<CH, BIAS>(L, lane, cur, top - lane >= 0, top - lane, bx0, bx1, by0, by1);
        gather_splat<CH, BIAS>(A, id2, top - WAVE - lane >= 0, cur);
        {
            const int q2 = top - 2 * WAVE - lane;
            id2 = (q2 >= 0) ? A.idx_sorted[range.x + q2] : 0;
        }
        __syncthreads();
        for (int j = 0; j < nb; ++j) {
            const float4 g0 = L.g0[j], g1 = L.g1[j];
            const int q = L.q[j];  // 0-based list position == reference's `contributor` after decrement
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
                    const float r1a = __builtin_amdgcn_rcpf(1.f - a);
                    T[p] = T[p] * r1a;
                    const float w = a * T[p];
                    float dLa = 0.f;
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        dLa += (f[k] - acc[p][k]) * gp[p][k];
                        s_f[k] += w * gp[p][k];
                        acc[p][k] = a * f[k] + (1.f - a) * acc[p][k];  // == reference's deferred update
                    }
                    dLa *= T[p];
                    dLa += (-Tf[p] * r1a) * bgdot[p];
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
            if (lane == 63 && !A.no_atomics) {
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


// ------------------------------------------------------------------ backward, atomic-free ("pair" mode)
// One 256-thread workgroup per tile = four waves, each owning an 8x8 pixel block.  The tile list is
// walked back-to-front in super-batches of SB entries staged ONCE for the whole tile in LDS; every
// wave culls the super-batch against its own block (ballot/popcount -> private index list), runs
// the per-pixel replay only over its survivors, DPP-reduces the per-lane partials and adds them
// into the super-batch's LDS accumulator (ds_add_f32 by one lane).  After a barrier the
// accumulator rows go out as ONE coalesced store per super-batch into pair_buf[sorted position];
// pair_reduce_kernel then sums each Gaussian's rows through the inverse pair map.  No global
// atomics, deterministic up to the order of the four LDS adds.
template <int CH, bool BIAS>
struct PairCfg {
    static constexpr int NG = BIAS ? 9 : 8;                 // ux uy ax ay ca cb cc o [bias]
    static constexpr int NC = NG + CH;                      // floats per pair record
    static constexpr int SB = CH <= 8 ? 256 : (CH <= 16 ? 128 : 64);
    static constexpr int CHP = (CH + 3) & ~3;
};

template <int CH, bool BIAS>
struct PairLDS {
    using Cfg = PairCfg<CH, BIAS>;
    float4 g0[Cfg::SB];             // u v a b
    float4 g1[Cfg::SB];             // c o bias id
    float f[Cfg::SB * Cfg::CHP];
    float acc[Cfg::SB * Cfg::NC];
    unsigned short list[4][Cfg::SB];
};

template <int CH, bool BIAS>
__global__ void __launch_bounds__(256)
blend_bwd_pair_kernel(const BlendArgs A) {
    using Cfg = PairCfg<CH, BIAS>;
    constexpr int SB = Cfg::SB, NC = Cfg::NC, NG = Cfg::NG, CHP = Cfg::CHP;
    __shared__ PairLDS<CH, BIAS> L;
    __shared__ int s_wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)bx, bx1 = (float)(bx + 7), by0 = (float)by, by1 = (float)(by + 7);
    const size_t HW = (size_t)A.H * A.W;

    const bool inside = (px < A.W) && (py < A.H);
    const size_t pix = (size_t)A.W * (size_t)py + px;
    const float Tf = inside ? A.final_T[pix] : 0.f;
    float T = Tf;
    const int last = inside ? A.ncontrib[pix] : 0;
    bool done = !inside;
    float acc[CH], gp[CH];
    float bgdot = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        acc[k] = 0.f;
        gp[k] = (inside && k < A.cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
        if (k < A.cn) bgdot += A.bg * gp[k];
    }
    const int wmax = wave_max_i(last);  // this wave never needs entries q >= wmax
    if (lane == 0) s_wmax[w] = wmax;
    __syncthreads();
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    float *pb = A.pair_buf + (size_t)range.x * NC;
    // entries nobody replays still get a (zero) record
    for (int i = n * NC + tid; i < len * NC; i += 256) pb[i] = 0.f;
    if (n <= 0) return;

    // per-thread software pipeline over the entries this thread stages (thread e < SB <-> entry top - e)
    const bool stager = tid < SB;
    int id1 = (stager && n - 1 - tid >= 0) ? A.idx_sorted[range.x + n - 1 - tid] : 0;
    int id2 = (stager && n - 1 - SB - tid >= 0) ? A.idx_sorted[range.x + n - 1 - SB - tid] : 0;
    Splat<CH> cur;
    gather_splat<CH, BIAS>(A, id1, stager && n - 1 - tid >= 0, cur);

    for (int top = n - 1; top >= 0; top -= SB) {
        const int nb = imin_(SB, top + 1);
        // ---- stage the super-batch once for the tile, clear the accumulator
        if (stager) {
            L.g0[tid] = make_float4(cur.u, cur.v, cur.a, cur.b);
            L.g1[tid] = make_float4(cur.c, cur.o, cur.bias, __int_as_float(cur.id));
#pragma unroll
            for (int k = 0; k < CHP; k += 4) {
                float4 v;
                v.x = k + 0 < CH ? cur.f[k + 0] : 0.f;
                v.y = k + 1 < CH ? cur.f[k + 1] : 0.f;
                v.z = k + 2 < CH ? cur.f[k + 2] : 0.f;
                v.w = k + 3 < CH ? cur.f[k + 3] : 0.f;
                *reinterpret_cast<float4 *>(&L.f[tid * CHP + k]) = v;
            }
            gather_splat<CH, BIAS>(A, id2, top - SB - tid >= 0, cur);  // prefetch next super-batch
            const int q2 = top - 2 * SB - tid;
            id2 = (q2 >= 0) ? A.idx_sorted[range.x + q2] : 0;
        }
        for (int i = tid; i < nb * NC; i += 256) L.acc[i] = 0.f;
        __syncthreads();

        // ---- per-wave cull -> private, order-preserving index list
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < SB / WAVE; ++r) {
            const int e = r * WAVE + lane;
            bool keep = (e < nb) && (top - e < wmax);
            if (keep && !BIAS) {
                const float4 a0 = L.g0[e], a1 = L.g1[e];
                keep = splat_touches(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, bx0, bx1, by0, by1);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) L.list[w][cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)e;
            cnt += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- replay over the survivors
        for (int j = 0; j < cnt; ++j) {
            const int e = L.list[w][j];
            const float4 g0 = L.g0[e], g1 = L.g1[e];
            const int q = top - e;
            const float dx = g0.x - pxf, dy = g0.y - pyf;
            const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
            const float G = __expf(power);
            float araw = g1.y * G;
            if (BIAS) araw = araw + g1.z;
            const float alpha = fminf(0.99f, araw);
            const bool ok = !done && (q < last) && !(power > 0.f) && !(alpha < (1.0f / 255.0f));
            if (!__any(ok)) continue;
            float f[CH];
#pragma unroll
            for (int k = 0; k < CHP; k += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(&L.f[e * CHP + k]);
                if (k + 0 < CH) f[k + 0] = v.x;
                if (k + 1 < CH) f[k + 1] = v.y;
                if (k + 2 < CH) f[k + 2] = v.z;
                if (k + 3 < CH) f[k + 3] = v.w;
            }
            float s[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) s[k] = 0.f;
            float s_f[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) s_f[k] = 0.f;
            if (ok) {
                const float r1a = __builtin_amdgcn_rcpf(1.f - alpha);
                T = T * r1a;
                const float wgt = alpha * T;
                float dLa = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    dLa += (f[k] - acc[k]) * gp[k];
                    s_f[k] = wgt * gp[k];
                    acc[k] = alpha * f[k] + (1.f - alpha) * acc[k];
                }
                dLa *= T;
                dLa += (-Tf * r1a) * bgdot;
                const float dLG = g1.y * dLa;
                const float gx_ = -G * dx * g0.z - G * dy * g0.w;
                const float gy_ = -G * dy * g1.x - G * dx * g0.w;
                s[0] = dLG * gx_; s[1] = dLG * gy_;
                s[2] = fabsf(s[0]); s[3] = fabsf(s[1]);
                s[4] = -0.5f * G * dx * dx * dLG;
                s[5] = -G * dx * dy * dLG;
                s[6] = -0.5f * G * dy * dy * dLG;
                s[7] = G * dLa;
                if (BIAS) {
                    s[8] = dLa;
                    done = T < 0.0001f;
                }
            }
#pragma unroll
            for (int k = 0; k < NG; ++k) s[k] = wave_sum_to_lane63(s[k]);
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < A.cn) s_f[k] = wave_sum_to_lane63(s_f[k]);
            if (lane == 63) {
                float *a = &L.acc[e * NC];
#pragma unroll
                for (int k = 0; k < NG; ++k) atomicAdd(a + k, s[k]);
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k < A.cn) atomicAdd(a + NG + k, s_f[k]);
            }
        }
        __syncthreads();
        // ---- one coalesced store of the super-batch's records: entry e <-> sorted position top - e
        {
            const int lo = top - nb + 1;  // lowest list position of this super-batch
            float *dst = pb + (size_t)lo * NC;
            for (int i = tid; i < nb * NC; i += 256) {
                const int ql = i / NC, c = i - ql * NC;
                dst[i] = L.acc[(nb - 1 - ql) * NC + c];
            }
        }
        __syncthreads();
    }
}

// sums each Gaussian's pair records (inverse pair map) into the final gradients -- plain stores.
template <bool BIAS>
__global__ void __launch_bounds__(256)
pair_reduce_kernel(const BlendArgs A, int NC, int CHk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.P) return;
    constexpr int NG = BIAS ? 9 : 8;
    const int beg = i > 0 ? A.goff_incl[i - 1] : 0, end = A.goff_incl[i];
    float g[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) g[k] = 0.f;
    float *df = A.dL_dfeature + (size_t)i * A.C + A.c0;
    // features: processed in register groups of 8 to bound register use for wide chunks
    for (int k0 = 0; k0 < A.cn; k0 += 8) {
        float fs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) fs[k] = 0.f;
        for (int j = beg; j < end; ++j) {
            const float *rec = A.pair_buf + (size_t)A.inv_pos[j] * NC;
            if (k0 == 0) {
#pragma unroll
                for (int k = 0; k < NG; ++k) g[k] += rec[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < A.cn) fs[k] += rec[NG + k0 + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k0 + k < A.cn) df[k0 + k] = fs[k];
    }
    (void)CHk;
    if (A.accumulate) {
        A.dL_duv[2 * i] += g[0]; A.dL_duv[2 * i + 1] += g[1];
        A.dL_dabs_uv[2 * i] += g[2]; A.dL_dabs_uv[2 * i + 1] += g[3];
        A.dL_dconic[3 * i] += g[4]; A.dL_dconic[3 * i + 1] += g[5]; A.dL_dconic[3 * i + 2] += g[6];
        A.dL_dopacity[i] += g[7];
        if (BIAS) A.dL_dbias[i] += g[NG - 1];
    } else {
        A.dL_duv[2 * i] = g[0]; A.dL_duv[2 * i + 1] = g[1];
        A.dL_dabs_uv[2 * i] = g[2]; A.dL_dabs_uv[2 * i + 1] = g[3];
        A.dL_dconic[3 * i] = g[4]; A.dL_dconic[3 * i + 1] = g[5]; A.dL_dconic[3 * i + 2] = g[6];
        A.dL_dopacity[i] = g[7];
        if (BIAS) A.dL_dbias[i] = g[NG - 1];
    }
}

template <int CH>
static int launch_bwd_pair(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    if (bias) SPLAT_LAUNCH("blend_bwd", (blend_bwd_pair_kernel<CH, true>), dim3((unsigned)T), dim3(256), 0, s, A);
    else SPLAT_LAUNCH("blend_bwd", (blend_bwd_pair_kernel<CH, false>), dim3((unsigned)T), dim3(256), 0, s, A);
    SPLAT_POST_LAUNCH();
    const int NC = (bias ? 9 : 8) + CH;
    const dim3 grid((unsigned)((A.P + 255) / 256));
    if (bias) SPLAT_LAUNCH("pair_reduce", pair_reduce_kernel<true>, grid, dim3(256), 0, s, A, NC, CH);
    else SPLAT_LAUNCH("pair_reduce", pair_reduce_kernel<false>, grid, dim3(256), 0, s, A, NC, CH);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

static int bwd_pair_chunk(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    const int cn = A.cn;
    if (cn <= 1) return launch_bwd_pair<1>(A, T, bias, s);
    if (cn <= 3) return launch_bwd_pair<3>(A, T, bias, s);
    if (cn <= 8) return launch_bwd_pair<8>(A, T, bias, s);
    if (cn <= 16) return launch_bwd_pair<16>(A, T, bias, s);
    if (cn <= 24) return launch_bwd_pair<24>(A, T, bias, s);
    return launch_bwd_pair<32>(A, T, bias, s);
}

extern "C" size_t splat_blend_pair_floats(int C, int has_bias) {
    // floats per pair record for the widest channel chunk of a C-channel backward
    const int cn = C > 32 ? 32 : C;
    const int ch = cn <= 1 ? 1 : cn <= 3 ? 3 : cn <= 8 ? 8 : cn <= 16 ? 16 : cn <= 24 ? 24 : 32;
    return (size_t)((has_bias ? 9 : 8) + ch);
}

// ================================================================== launch tables
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int CH, int BW, int BH>
static int launch_fwd(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    const dim3 grid((unsigned)(T * Block<BW, BH>::WPT)), block(WAVE);
    if (enh) {
        if (bias) SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, BW, BH, true, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, BW, BH, true, false>), grid, block, 0, s, A);
    } else {
        if (bias) SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, BW, BH, false, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, BW, BH, false, false>), grid, block, 0, s, A);
    }
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <int CH, int BW, int BH>
static int launch_bwd(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    const dim3 grid((unsigned)(T * Block<BW, BH>::WPT)), block(WAVE);
    if (bias) SPLAT_LAUNCH("blend_bwd", (blend_bwd_kernel<CH, BW, BH, true>), grid, block, 0, s, A);
    else SPLAT_LAUNCH("blend_bwd", (blend_bwd_kernel<CH, BW, BH, false>), grid, block, 0, s, A);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// block shape per channel width: 8x8 (1 px/lane) by default; SPLAT_FWD_SHAPE / SPLAT_BWD_SHAPE
// (0: 8x8, 1: 16x8, 2: 16x16) are tuning switches for experiments on narrow channel counts.
template <int CH>
static int fwd_shape(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    if constexpr (CH <= 8) {
        const int shape = env_int("SPLAT_FWD_SHAPE", 0);
        if (shape == 1) return launch_fwd<CH, 16, 8>(A, T, enh, bias, s);
        if (shape == 2) return launch_fwd<CH, 16, 16>(A, T, enh, bias, s);
    }
    return launch_fwd<CH, 8, 8>(A, T, enh, bias, s);
}

template <int CH>
static int bwd_shape(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    if constexpr (CH <= 8) {
        const int shape = env_int("SPLAT_BWD_SHAPE", 0);
        if (shape == 1) return launch_bwd<CH, 16, 8>(A, T, bias, s);
        if (shape == 2) return launch_bwd<CH, 16, 16>(A, T, bias, s);
    }
    return launch_bwd<CH, 8, 8>(A, T, bias, s);
}

static int fwd_chunk(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    const int cn = A.cn;
    if (cn <= 1) return fwd_shape<1>(A, T, enh, bias, s);
    if (cn <= 3) return fwd_shape<3>(A, T, enh, bias, s);
    if (cn <= 8) return fwd_shape<8>(A, T, enh, bias, s);
    if (cn <= 16) return fwd_shape<16>(A, T, enh, bias, s);
    if (cn <= 24) return fwd_shape<24>(A, T, enh, bias, s);
    return fwd_shape<32>(A, T, enh, bias, s);
}

static int bwd_chunk(const BlendArgs &A, int T, bool bias, hipStream_t s) {
    const int cn = A.cn;
    if (cn <= 1) return bwd_shape<1>(A, T, bias, s);
    if (cn <= 3) return bwd_shape<3>(A, T, bias, s);
    if (cn <= 8) return bwd_shape<8>(A, T, bias, s);
    if (cn <= 16) return bwd_shape<16>(A, T, bias, s);
    if (cn <= 24) return bwd_shape<24>(A, T, bias, s);
    return bwd_shape<32>(A, T, bias, s);
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
                                             const int32_t *goff_incl, const int32_t *inv_pos, float *pair_scratch,
                                             splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && C >= 1 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    const bool pair_mode = goff_incl && inv_pos && pair_scratch && !env_int("SPLAT_BWD_ATOMIC", 0);
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
    A.no_atomics = env_int("SPLAT_EXP_NO_ATOMICS", 0);
    A.goff_incl = goff_incl; A.inv_pos = inv_pos; A.pair_buf = pair_scratch;
    const int T = A.gx * ((H + TILE - 1) / TILE);
    for (int c0 = 0; c0 < C; c0 += 32) {  // reference chunking: src/alpha_blending.cu:440-577
        A.c0 = c0;
        A.cn = C - c0 > 32 ? 32 : C - c0;
        A.accumulate = c0 > 0;
        const int rc = pair_mode ? bwd_pair_chunk(A, T, opacity_bias != nullptr, (hipStream_t)stream)
                                 : bwd_chunk(A, T, opacity_bias != nullptr, (hipStream_t)stream);
        if (rc != SPLAT_OK) return rc;
    }
    return SPLAT_OK;
}
