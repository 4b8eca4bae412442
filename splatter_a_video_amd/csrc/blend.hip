// Tile-based front-to-back alpha compositing, forward and backward (+ "enhanced" first-K index
// capture and the opacity-bias variant).  Reference semantics: src/alpha_blending.cu:16-249,
// src/alpha_blending_enhanced.cu:57-133, src/alpha_blending_with_bias.cu:88-89,211-214,259-261.
//
// MI355X design (the reference: 256-thread tile block, every pixel walks the whole 3-sigma tile
// list, one global float atomic per (pixel, splat, component) in the backward):
//
//   * one 256-thread workgroup per 16x16 tile = four waves, each owning an 8x8 pixel block.
//   * the tile's depth-sorted list is consumed in super-batches of SB entries that are gathered
//     ONCE per tile (thread e gathers entry e: id, uv, conic, opacity, C features straight from
//     the [P,C] row-major tensor -- no host transpose) with the ids two super-batches and the
//     payload one super-batch ahead in registers, then parked in LDS.
//   * CULL: a splat can only reach alpha >= 1/255 inside the ellipse d^T Q d <= 2 ln(255 o); every
//     wave tests the axis-aligned box of that ellipse (inflated by a rounding bound) against its own
//     8x8 block and builds a private order-preserving survivor list with ballot/popcount.  The
//     pixel loop only visits survivors (about 40 % of the tile list for 2-pixel-sigma splats).
//     The test is conservative, so results are unchanged.
//   * pixel loop: several survivors per trip (independent chains for ILP), records read back as
//     broadcast ds_read_b128.
//   * backward ("pair" mode, used when idx_sorted comes from our sort_gaussian): per-lane partial
//     gradients are wave-reduced with in-place DPP adds (row_shr x4, row_bcast x2), lane 63
//     stores the (8+C)-float record into the wave's private LDS slab; after the super-batch the
//     four slabs are summed and every record (padded to whole 16-byte chunks) is stored at its
//     Gaussian-major pair slot (slot_sorted[] comes out of the tile sort); pair_reduce_kernel then
//     streams each Gaussian's contiguous records.  No global atomics at all.
//   * backward ("atomic" mode, foreign idx_sorted): same replay per wave, one hardware float
//     atomic per (wave, splat, component).
#include <stdlib.h>

#include <utility>

#include "common.h"

// tuning knobs (compile time; tools/build_variants.sh builds alternatives for A/B runs)
#ifndef BLEND_FWD8_MINW
#define BLEND_FWD8_MINW 4  // 5 .. 8 channels (the trainer's small plans: rgb + depth + 1 .. 4 attributes, with the K = 20 id lists): at the
                           // narrow rows' 80 registers the enhanced instantiation spills 172 bytes per lane (128 us per frame at c2)
#endif
#ifndef BLEND_FWD_U
#define BLEND_FWD_U 2      // survivors evaluated per trip in the forward (narrow channel counts).  Round 6: 4 -> 2 (the quarter lists of a
                           // wave are padded to a multiple of U: with the denser lists of the reach masks 46.4 -> 44.9 us per frame)
#endif
#ifndef BLEND_FWD_QDONE
#define BLEND_FWD_QDONE 1      // forward: a 4x4 quarter whose pixels are all saturated keeps no further entries (its block's may go on)
#endif
#ifndef BLEND_FWD_TRIPTEST
#define BLEND_FWD_TRIPTEST 0   // forward: skip a trip's compositing when no lane of the wave has a splat to apply (see the kernel)
#endif
#ifndef BLEND_FWD_SB
#define BLEND_FWD_SB 128   // forward super-batch (narrow channel counts): 14 KB of LDS per workgroup
#endif
#ifndef BLEND_FWD_MINW
#define BLEND_FWD_MINW 6   // __launch_bounds__ min waves per SIMD for the forward: 80 VGPRs, 6 workgroups per CU (with the
                           // frame batch's 40 500 tiles per launch; 7 -- all tiles of ONE 480p frame resident -- costs 18
                           // spilled registers once the cull flags are written: 84 vs 77 us per frame; 5: 80 us, 4: 88 us)
#endif
#ifndef BLEND_BWD_U
#define BLEND_BWD_U 2
#endif
#ifndef BLEND_BWD_MINW
#define BLEND_BWD_MINW 1
#endif
#ifndef BLEND_MFMA_SB
#define BLEND_MFMA_SB 128  // super-batch of the MFMA backward (narrow channel counts)
#endif
#ifndef BLEND_MFMA_MINW
#define BLEND_MFMA_MINW 4  // 128 VGPRs: 4 workgroups per CU (what 40 KB of LDS allow as well)
#endif
#ifndef BLEND_WIDE_SB
#define BLEND_WIDE_SB 64
#endif
#ifndef BLEND_WIDE_MINW
#define BLEND_WIDE_MINW 1
#endif
#ifndef BLEND_CARRY
#define BLEND_CARRY 1
#endif
#ifndef BLEND_WIDE_HOIST
#define BLEND_WIDE_HOIST 1
#endif
#ifndef BLEND_ABL
#define BLEND_ABL 0        // ablation of the matrix-core backward for timing experiments (1: no combine, 2: no chunks); results invalid
#endif
#ifndef BLEND_SETS_MINW
#define BLEND_SETS_MINW 2  // waves per SIMD the three-set backward is compiled for (3: 168 registers = 39 scratch accesses
                           // inside the chunk loop, 1061 instead of 615 us per frame; its LDS would allow 3)
#endif
#ifndef BLEND_SETS_CAP
#define BLEND_SETS_CAP 32  // slab rows per wave of the three-set backward (list positions per round)
#endif
#ifndef BLEND_SETS_LE_AHEAD
#define BLEND_SETS_LE_AHEAD 0  // three-set quarter kernel: read a step's list entry one step ahead (1) or at the top of the step (0).
                               // Measured (round 4, c2 training frame): 438 vs 434 us per frame -- the round trip it saves is hidden
                               // by the second wave of the SIMD, the address arithmetic it adds is not
#endif
#ifndef BLEND_SETS_TWO_BARRIERS
#define BLEND_SETS_TWO_BARRIERS 1  // three-set quarter kernel: no barrier behind the combine (it reads the staged geometry in front of its barrier)
#endif
#ifndef BLEND_SETS_EARLY_ROWS
#define BLEND_SETS_EARLY_ROWS 1  // three-set quarter kernel: request the survivor's slab row at the top of the step (13 registers
                                 // across the step) instead of in front of the epilogue's adds
#endif
#ifndef BLEND_LATE_STAGE
#define BLEND_LATE_STAGE 1 // matrix-core backward: gather the next super-batch's records behind the chunk loop (1) or in front of it (0)
#endif
#ifndef BLEND_SLOT_EARLY
#define BLEND_SLOT_EARLY 0 // matrix-core backward: load the combine's pair slots at the top of the super-batch (1) or behind the chunks (0)
#endif
#ifndef BLEND_REC_SWZ
#define BLEND_REC_SWZ 1    // strip-walk backward kernels: XOR swizzle of the staged records' 16-byte parts (bank conflicts)
#endif
#ifndef BLEND_STATE_SKEW
#define BLEND_STATE_SKEW 1 // strip-walk backward kernels: per-pixel replay state rows skewed by lane group (bank conflicts)
#endif
// The reference skips a splat when power > 0 (src/alpha_blending.cu:93).  power is a negative-semidefinite form -- EWA only
// emits positive-definite conics (cov2d + 0.3 I) -- so it exceeds 0 by rounding only (the expanded polynomial carries ~1e-5 of
// absolute noise in log2 units; at a splat's centre the reference evaluates exp(0) = 1).  No compare per (pixel, splat) is spent
// on it: the raw alpha is exp2 of the polynomial (opacity included, see power_coeffs) with the clamp bit of v_exp_f32 set --
// alpha_raw = min(o exp(power), 1), and a NaN (o < 0, garbage conic) becomes 0 under the DX10 clamp, i.e. "skipped".  Forward
// and every backward kernel share exp2_guard(), so decisions stay reproducible.
// raw alpha = exp2(pw) = o * exp(power); `ok` is always true (kept for the callers' predicate chains)
__device__ __forceinline__ float exp2_guard(float pw, bool &ok) {
    ok = true;
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(pw), 0.f, 1.f);   // folds into v_exp_f32 ... clamp
}

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
    const float *bgc;  // optional per-channel background [C] (forward); NULL: bg for every channel
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
    float *dL_dndc, *dL_dabs_ndc;  // optional densification taps (pair mode): dL_duv / dL_dabs_uv scaled by (W/2, H/2)
    // atomic-free backward: per-(tile,splat) partial sums + inverse pair map
    float *pair_buf;        // [M, NCP] partial gradients, one 16-B-aligned record per pair SLOT (Gaussian-major)
    const int *goff_incl;   // [P] inclusive prefix of tiles per Gaussian (slot ranges)
    const int *slot_sorted; // [M] sorted position -> pair slot
    int accumulate;         // reduce: add to the geometry gradients (channel chunks > 0)
    int pack_valid;         // backward: pack already holds this chunk's records (kept from the forward)
    float *pack;            // [P, Rec<CH>::RS] packed records of the current channel chunk (scratch)
    float *dbg_T_front;     // optional [H,W] (backward): the transmittance the replay arrives at in front of the first splat
                            // (1 up to rounding iff every inclusion decision of the forward was reproduced)
    // frame batch: F frames of one Gaussian set in a single launch (workgroup -> (frame, tile)); every per-frame array
    // is the frame-0 pointer plus frame * stride.  F = 1: the single-frame operators.
    int F, T;               // frames, tiles per frame
    int tile_only;          // backward: stop after the pair records (the caller reduces them over the frames)
    long long cap;          // pair capacity per frame: stride of idx_sorted / slot_sorted / pair_buf records
    long long pack_fs;      // floats between two frames' packed records
    long long opacity_fs, feature_fs, bias_fs;  // element strides of the per-Gaussian inputs (0: shared by all frames)
    int feature_row;        // pack_sets_kernel: floats between two Gaussians' rows of `feature` (0: C -- a dense [P, C] row; the forward's
                            // packed records serve as the row with their record stride)
    // several feature sets in one backward pass (blend_bwd_sets_kernel): first row channel, width and background of the
    // tap set (0), the second set (1) and the opacity-detached set (2); width 0 = no such set
    int s0c0, s0cn, s1c0, s1cn, s2c0, s2cn;
    float s0bg, s1bg, s2bg;
    // ... each set's features [P, cn] (frame stride sfs: 0 = shared by the frames) and image gradient [F, cn, H, W] in its own
    // tensor (the caller does not concatenate a row / a gradient image); NULL: `feature` / `dL_dout` hold the whole row
    const float *sf0, *sf1, *sf2, *sdl0, *sdl1, *sdl2;
    long long sfs0, sfs1, sfs2;
    // forward packing of a row whose channels come from several SOURCES (splat_alpha_blending_forward_batch_sources): source s =
    // row channels [src_c0[s], + src_cn[s]) from dense rows src_f[s][P, cn] (+ frame * src_fs[s]); nsrc = 0: sf0 .. sf2 above
    int nsrc;
    int src_c0[SPLAT_MAX_SOURCES], src_cn[SPLAT_MAX_SOURCES];
    const float *src_f[SPLAT_MAX_SOURCES];
    long long src_fs[SPLAT_MAX_SOURCES];
    // optional [F, cap] words, one per sorted tile entry: byte w != 0 = the forward's cull kept the entry for the tile's
    // block w (the keep word of tile_cull).  The forward writes them, the strip-walk backward kernels read them instead
    // of repeating the cull (13 % of their VALU instructions); NULL: every kernel culls for itself.
    unsigned int *cull_flags;
    // pair records inside a wider record: floats [rec_off, rec_off + ..) of records rec_stride floats apart (0: the kernel's
    // own stride).  The renderer's row in two passes -- the tap set through the narrow matrix-core kernel, depth + attributes
    // through blend_bwd_attr_kernel -- shares one record [A: ux uy ca cb cc o ax ay f0 f1 f2 . | B: ux uy ca cb cc o dz a0 ..]
    int rec_stride, rec_off;
    // loss-fused three-set backward (splat_alpha_blending_backward_batch_sets_l1): the image gradient is that of the L1 loss,
    // computed where the kernel hoists dL_dout -- l1_pred = the forward's output row [F, C, H, W], sdl0 .. sdl2 hold the sets'
    // TARGET images, the gradient of a channel is l1_s<g> * sign(pred - target), and sum |pred - target| of every (frame, tile,
    // set) is WRITTEN to l1_sum[(frame * T + tile) * 3 + g] (the caller adds them up).  NULL: sdl* / dL_dout are gradients.
    const float *l1_pred;
    float l1_s0, l1_s1, l1_s2;
    float *l1_sum;
};

// per-frame view of the argument block (all uniform: scalar address arithmetic)
__device__ __forceinline__ BlendArgs frame_args(const BlendArgs &B, int f) {
    BlendArgs A = B;
    if (B.F > 1) {
        const size_t HW = (size_t)B.H * B.W, fz = (size_t)f;
        if (B.uv) {  // inputs of the pack kernel (absent when the backward reuses the forward's records)
            A.uv = B.uv + fz * B.P;
            A.conic = B.conic + fz * 3 * B.P;
            A.opacity = B.opacity + fz * B.opacity_fs;
            A.feature = B.feature ? B.feature + fz * B.feature_fs : nullptr;
        }
        if (B.bias) A.bias = B.bias + fz * B.bias_fs;
        A.idx_sorted = B.idx_sorted + fz * B.cap;
        A.tile_range = B.tile_range + fz * B.T;
        A.pack = B.pack + fz * B.pack_fs;
        if (B.out) A.out = B.out + fz * B.C * HW;
        if (B.final_T) A.final_T = B.final_T + fz * HW;
        if (B.ncontrib) A.ncontrib = B.ncontrib + fz * HW;
        if (B.gs_idx) A.gs_idx = B.gs_idx + fz * HW * B.K;
        if (B.dL_dout) A.dL_dout = B.dL_dout + fz * B.C * HW;
        if (B.slot_sorted) A.slot_sorted = B.slot_sorted + fz * B.cap;
        if (B.goff_incl) A.goff_incl = B.goff_incl + fz * B.P;
        if (B.dbg_T_front) A.dbg_T_front = B.dbg_T_front + fz * HW;
        if (B.cull_flags) A.cull_flags = B.cull_flags + fz * B.cap;
        if (B.sf0) A.sf0 = B.sf0 + fz * B.sfs0;
        if (B.sf1) A.sf1 = B.sf1 + fz * B.sfs1;
        if (B.sf2) A.sf2 = B.sf2 + fz * B.sfs2;
        if (B.sdl0) A.sdl0 = B.sdl0 + fz * B.s0cn * HW;
        if (B.sdl1) A.sdl1 = B.sdl1 + fz * B.s1cn * HW;
        if (B.sdl2) A.sdl2 = B.sdl2 + fz * B.s2cn * HW;
        if (B.l1_pred) A.l1_pred = B.l1_pred + fz * B.C * HW;
    }
    if (B.l1_sum) A.l1_sum = B.l1_sum + 3 * (size_t)f * (size_t)B.T;
    return A;
}

// ---- the splat's exponent: ONE arithmetic for the forward and every backward kernel, so that a pixel's backward
// reproduces its forward's alpha bit for bit (the reference's two kernels share their expression as well,
// src/alpha_blending.cu:78-87 vs :196-203; a decision alpha >= 1/255 that flips between the two passes would corrupt
// the T /= (1 - alpha) replay of that pixel).
//   log2(o) + power(x, y) * log2(e) = q0 + qx x + qy y + qxx x^2 + qxy x y + qyy y^2      x, y: pixel relative to the tile centre
// evaluated as the fused-multiply-add chain q0 -> +x qx -> +y qy -> +xx qxx -> +xy qxy -> +yy qyy.  The matrix-core
// backward gets exactly this chain from two v_mfma_f32_16x16x4_f32 (an f32 MFMA is the ascending fma chain over k
// starting from C: profiles/r02_mfma_fma_chain_probe.json, 2^20 of 2^20 random products bit-equal); the lane = pixel
// kernels run it on the VALU.  The coefficients come from power_coeffs() everywhere (explicit fma, no contraction).
// exp2 of it is the raw alpha o * exp(power) itself.  The reference's "power > 0" guard: see exp2_guard().
#define BLEND_L2E 1.4426950408889634f
struct PowerCoef {
    float q0, qx, qy, qxx, qxy, qyy;
};
// The opacity rides in the constant term: q0 = log2(o) - 0.5 log2(e) c^T Q c, so that exp2 of the polynomial IS o * exp(power)
// (the raw alpha) -- one multiply less per (pixel, splat) evaluation in every kernel.  o <= 0 gives -inf / NaN -> alpha 0
// (the reference: o * G < 1/255 -> skipped).
__device__ __forceinline__ PowerCoef power_coeffs(float u, float v, float cA, float cB, float cC, float o, float cx, float cy) {
#pragma clang fp contract(off)
    const float uc = u - cx, vc = v - cy;  // splat centre relative to the tile centre
    const float tx = __builtin_fmaf(cA, uc, cB * vc);
    const float ty = __builtin_fmaf(cB, uc, cC * vc);
    PowerCoef k;
    k.q0 = __builtin_fmaf(-0.5f * BLEND_L2E, __builtin_fmaf(uc, tx, vc * ty), __builtin_amdgcn_logf(o));
    k.qx = BLEND_L2E * tx;
    k.qy = BLEND_L2E * ty;
    k.qxx = (-0.5f * BLEND_L2E) * cA;
    k.qxy = (-BLEND_L2E) * cB;
    k.qyy = (-0.5f * BLEND_L2E) * cC;
    return k;
}
// lane = pixel evaluation; c0 = [q0 qx qy qxx], c1 = [qxy qyy . .]; x, y, xx = x^2, xy, yy = y^2 exact in f32
__device__ __forceinline__ float power_poly(const float4 &c0, const float4 &c1, float x, float y, float xx, float xy,
                                            float yy) {
    float pw = __builtin_fmaf(x, c0.y, c0.x);
    pw = __builtin_fmaf(y, c0.z, pw);
    pw = __builtin_fmaf(xx, c0.w, pw);
    pw = __builtin_fmaf(xy, c1.x, pw);
    return __builtin_fmaf(yy, c1.y, pw);
}
// opacity-bias variant (no cull, no matrix-core kernel): the reference's factored expression, shared by its forward
// and backward kernels; returns -power
__device__ __forceinline__ float neg_power_factored(float dx, float dy, float cA, float cB, float cC) {
#pragma clang fp contract(off)
    const float h = __builtin_fmaf(cA, 0.5f * dx, cB * dy);
    return __builtin_fmaf(dx, h, (cC * (0.5f * dy)) * dy);
}
__device__ __forceinline__ float exp_neg(float q) { return __builtin_amdgcn_exp2f(-BLEND_L2E * q); }

struct __attribute__((packed, aligned(4))) F3 {
    float x, y, z;
};

// ---- packed per-Gaussian record (built once per blend call by pack_kernel):
//   [u v A B | C o bias id | f0 .. f(CH-1) | pad]  RS floats, RS a multiple of 16 (whole 64-B sectors).
// The tile kernels gather ONE contiguous record per list entry (4 lanes x 16 B per sector) instead
// of touching four separate arrays (uv, conic, opacity, feature) -- a 128-B line fetched per 4..12-B
// access was 4-7x the algorithmic bytes (profiles/: TCC_EA0_RDREQ_128B).
template <int CH>
struct Rec {
    static constexpr int RS = (8 + CH + 15) & ~15;  // floats per record
    static constexpr int RQ = RS / 4;               // float4 chunks per record
    // culling parameters [hx hy tauq 1/A 1/C] of the splat (cull_params) live in the last 5 pad floats when the
    // record has that much padding; otherwise the tile kernels derive them from the conic
    static constexpr int CULL = (RS - 8 - CH >= 5) ? RS - 5 : -1;
};

// (CullP, cull_params, cull_test: common.h -- the binning kernels run the same test on whole tiles when they create the pairs)

// rows [i0, i0 + nrec) of a set's [P, cn] feature tensor into the staged records (LDS rows of LS floats, float offset `at`):
// the 256 threads read the nrec * cn consecutive floats coalesced (a thread reading its own 76-byte row of the attribute set
// touched 64 lines per load instruction: 2.4 TB/s for the whole packing kernel)
// SETS: `at` is the set's first SLOT and channel ch goes to float sets_fpos(at + ch) of the row (transposed slots, see there)
template <bool SETS = false>
__device__ __forceinline__ void stage_feature_rows(float *s_rec, int LS, int at, const float *src, int cn, int i0, int nrec) {
    if (!src || cn <= 0) return;
    const float *p = src + (size_t)i0 * cn;
    for (int e = threadIdx.x; e < nrec * cn; e += 256) {
        const int row = e / cn, ch = e - row * cn;
        s_rec[row * LS + (SETS ? 8 + 8 * ((at + ch) & 3) + ((at + ch) >> 2) : at + ch)] = p[e];
    }
}

template <int CH, bool BIAS, bool EXACT>
__global__ void __launch_bounds__(256)
pack_kernel(const BlendArgs B) {
    const BlendArgs A = frame_args(B, blockIdx.y);
    constexpr int RS = Rec<CH>::RS, RQ = Rec<CH>::RQ;
    constexpr int LS = RS + 4;  // LDS row stride in floats: 16-B aligned, conflict-free for the float4 row writes
    __shared__ __attribute__((aligned(16))) float s_rec[256 * LS];
    const int i0 = blockIdx.x * 256;
    const int i = i0 + threadIdx.x;
    float r[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) r[k] = 0.f;
    if (i < A.P) {
        const float2 q = A.uv[i];
        r[0] = q.x; r[1] = q.y;
        r[2] = A.conic[3 * i]; r[3] = A.conic[3 * i + 1]; r[4] = A.conic[3 * i + 2];
        r[5] = A.opacity[i];
        if (BIAS) r[6] = A.bias[i];
        r[7] = __int_as_float(i);
        if (A.feature) {
            const float *f = A.feature + (size_t)i * A.C + A.c0;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (EXACT || k < A.cn) r[8 + k] = f[k];
        }   // (else: the row's sets live in their own tensors -- staged cooperatively below)
        if (Rec<CH>::CULL >= 0) {
            constexpr int CO = Rec<CH>::CULL >= 0 ? Rec<CH>::CULL : 0;
            const CullP cp = cull_params(r[2], r[3], r[4], r[5]);
            r[CO] = cp.hx; r[CO + 1] = cp.hy; r[CO + 2] = cp.tauq; r[CO + 3] = cp.ia; r[CO + 4] = cp.ic;
        }
    }
    {
        // rows through LDS so that consecutive lanes store consecutive 16-byte chunks of the record array
        float4 *row = reinterpret_cast<float4 *>(s_rec + threadIdx.x * LS);
#pragma unroll
        for (int k = 0; k < RS; k += 4) row[k / 4] = make_float4(r[k], r[k + 1], r[k + 2], r[k + 3]);
        __syncthreads();
        const int nrec = imin_(256, A.P - i0);
        if (!A.feature) {  // row channel c of source s sits at float 8 + c - c0 of the record (chunk [c0, c0 + cn) = the whole row)
            // (the table is read from the kernel argument block B itself: indexing a modified copy would put it in scratch)
            for (int sidx = 0; sidx < B.nsrc; ++sidx)
                stage_feature_rows(s_rec, LS, 8 + B.src_c0[sidx] - B.c0, B.src_f[sidx] + (size_t)blockIdx.y * B.src_fs[sidx],
                                   B.src_cn[sidx], i0, nrec);
            __syncthreads();
        }
        float4 *dst = reinterpret_cast<float4 *>(A.pack + (size_t)i0 * RS);
        for (int c = threadIdx.x; c < nrec * RQ; c += 256) {
            const int g = c / RQ, part = c - g * RQ;
            dst[c] = *reinterpret_cast<const float4 *>(s_rec + g * LS + 4 * part);
        }
    }
}

// ---- staging area of one super-batch (shared by the four waves of a tile): SB packed records,
// slot SB = inert (all-zero) record for the padded tail of the survivor lists
// COEF: the staging threads also leave the exponent's polynomial [q0 qx qy qxx | qxy qyy o id] of every entry (tile-centred,
// power_coeffs) for the lane = pixel kernels; slot SB is the inert entry (opacity 0).
// XR: extra record rows behind the inert one (rows SB + 1 + 15 w + j: wave w's carried survivors, see CarryLDS)
// SWZ: part p (16 bytes) of entry e's record sits at part (p & ~3) | ((p & 3) ^ ((e >> 2) & 3)).  A record is 16 or 48
// dwords, so part p of every entry starts on one of FOUR bank quads (e mod 4): the matrix-core kernels' lanes read part 0 / 1
// of 16 different survivors at once (ds_read_b128, 64 banks: 4-way conflict) and a K-slice of the features (ds_read_b32,
// 32 banks: 8-way).  With the swizzle 16 consecutive entries hit 16 different quads.
// RQL: the leading 16-byte parts of a record that are staged (default: all) -- rows of RQL parts
// CS: float4 per coefficient block -- 2, or 3 when the block also carries the entry's (at most four) channels behind the
// coefficients (the narrow forward: one base address per evaluated survivor instead of two)
template <int CH, int SB, bool COEF = false, int XR = 0, bool SWZ = false, int RQL = Rec<CH>::RQ, int CS = 2>
struct TileLDS {
    static constexpr int RQ = RQL;
    static constexpr bool SWIZZLED = SWZ;
    static_assert(!SWZ || RQL % 4 == 0, "the swizzle permutes groups of four parts");
    float4 rec[(SB + 1 + XR) * RQ];
    static constexpr int CSTRIDE = CS;
    float4 coef[COEF ? CS * (SB + 1) : 1];
    unsigned int keep[SB];  // byte w of entry e: wave w's 8x8 block can be reached by the splat (and passes its predicate)
    unsigned short list[4][SB + 16 + XR / 4];
    static __device__ __forceinline__ int part(int e, int p) {  // float4 index of part p of entry e
        return SWZ ? e * RQ + ((p & ~3) | ((p & 3) ^ ((e >> 2) & 3))) : e * RQ + p;
    }
    __device__ __forceinline__ const float4 &g0(int e) const { return rec[part(e, 0)]; }  // u v A B
    __device__ __forceinline__ const float4 &g1(int e) const { return rec[part(e, 1)]; }  // C o bias id
    // float k of entry e's record (k need not be a constant)
    __device__ __forceinline__ float f(int e, int k) const { return reinterpret_cast<const float *>(&rec[part(e, k >> 2)])[k & 3]; }
};

template <int CH, int SB, bool COEF, int XR>
__device__ __forceinline__ void read_feat(const TileLDS<CH, SB, COEF, XR> &L, int e, float f[CH]) {
    constexpr int RQ = Rec<CH>::RQ;
#pragma unroll
    for (int k = 0; k < CH; k += 4) {
        const float4 v = L.rec[e * RQ + 2 + k / 4];
        if (k + 0 < CH) f[k + 0] = v.x;
        if (k + 1 < CH) f[k + 1] = v.y;
        if (k + 2 < CH) f[k + 2] = v.z;
        if (k + 3 < CH) f[k + 3] = v.w;
    }
}

// Software-pipelined gather of packed records: a super-batch is SB*RQ float4 chunks, chunk c belongs
// to entry c / RQ; thread t moves chunks t, t+256, ...  Ids run two super-batches ahead, payload one.
// `pos(e, b)` maps (entry, batch) to the list position or -1.
template <int CH, int SB>
struct Stager {
    static constexpr int RQ = Rec<CH>::RQ;
    static constexpr int NCHUNK = SB * RQ;
    static constexpr int K = (NCHUNK + 255) / 256;
    int id_next[K];
    float4 v[K];

    template <typename Pos>
    __device__ __forceinline__ void load_ids(const BlendArgs &A, int tid, int base, Pos pos, int batch) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = tid + 256 * k;
            const int q = (c < NCHUNK) ? pos(c / RQ, batch) : -1;
            id_next[k] = q >= 0 ? A.idx_sorted[base + q] : -1;
        }
    }
    __device__ __forceinline__ void load_payload(const BlendArgs &A, int tid) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = tid + 256 * k;
            const int part = c % RQ;
            v[k] = id_next[k] >= 0
                       ? *reinterpret_cast<const float4 *>(A.pack + (size_t)id_next[k] * Rec<CH>::RS + 4 * part)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    template <bool COEF, int XR, bool SWZ, int RQL, int CS>
    __device__ __forceinline__ void park(TileLDS<CH, SB, COEF, XR, SWZ, RQL, CS> &L, int tid) const {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = tid + 256 * k;
            if (RQL == RQ) {
                if (c < NCHUNK) L.rec[SWZ ? L.part(c / RQ, c % RQ) : c] = v[k];
            } else if (c < NCHUNK && c % RQ < RQL) {   // only the leading parts are staged
                L.rec[L.part(c / RQ, c % RQ)] = v[k];
            }
        }
    }
};

// Cooperative cull of the staged super-batch: thread e tests entry e once against the four 8x8 blocks of the tile
// (wave w owns block (w & 1, w >> 1)) and records one flag byte per wave.  pred(e, w) drops entries a wave does not
// need before the geometric test.  Callers put a __syncthreads() between tile_cull and build_list.
// SUB: the flag byte of a kept block carries one bit per 4x4 quarter (bit sx + 2 sy) from a bounding-box test of the
// quarter's pixel centres, for kernels that keep a survivor list per quarter; otherwise the byte is 0 / 1.
// gflags: optional global copy of the staged entries' keep words (BlendArgs::cull_flags + the super-batch's first position).
// Quarter bits of a kept block: bounding-box test of the quarter's pixel centres, and -- rows of 16 channels and more
// (BLEND_EXACTQ 1) -- the exact ellipse test of cull_test on the quarter's rectangle for the quarters the box lets through:
// 12 % fewer list entries and 7 % fewer steps in the three-set backward (391 -> 364 us per frame at c2) for + 10 us in the
// 24-channel forward, whose cull threads own two blocks each.  Narrow rows (BLEND_EXACTQ 2) lose: the forward's cull is a
// quarter of the little it does per super-batch (+ 17 us against - 6.5 us in the backward).
#ifndef BLEND_EXACTQ
#define BLEND_EXACTQ 1
#endif
#ifndef BLEND_TANQ
#define BLEND_TANQ 1     // rows below 16 channels: tangent-plane test of the quarters the bounding box lets through (see tile_cull)
#endif
#ifndef BLEND_TANQ_NOBOX
#define BLEND_TANQ_NOBOX 0
#endif
#ifndef BLEND_TANQ_GATE
#define BLEND_TANQ_GATE 0   // ... behind the exact test of the 8x8 block (1) or on their own (0: narrow forward 60.2 -> 54.9 us per frame,
                            // backward 130.7 -> 132.8 at c2: the block test cost the cull threads more than the 1.7 % of entries it removed)
#endif
__device__ __forceinline__ unsigned live_quarters(bool b) { return b ? 15u : 0u; }
__device__ __forceinline__ unsigned live_quarters(unsigned m) { return m; }

template <int CH, int SB, bool BIAS, bool SUB, bool COEF, int XR, bool SWZ, int RQL, int CS, typename Pred>
__device__ __forceinline__ void tile_cull(TileLDS<CH, SB, COEF, XR, SWZ, RQL, CS> &L, int tid, int nb, float tx0, float ty0, Pred pred,
                                          unsigned int *gflags = nullptr) {
    constexpr int TPE = 256 / SB;  // threads per entry (1, 2 or 4): each tests 4 / TPE of the blocks
    constexpr int BPT = 4 / TPE;
    static_assert(SB == 64 || SB == 128 || SB == 256, "super-batch sizes the 256-thread cull supports");
    const int e = tid & (SB - 1), part = tid / SB;
    unsigned char *flags = reinterpret_cast<unsigned char *>(L.keep) + 4 * e + BPT * part;
    unsigned k[BPT];
#pragma unroll
    for (int j = 0; j < BPT; ++j) k[j] = 0u;
    if (e < nb) {
        if (BIAS) {  // the opacity bias lifts alpha everywhere: no geometric cull
#pragma unroll
            for (int j = 0; j < BPT; ++j) k[j] = pred(e, BPT * part + j) ? (SUB ? 15u : 1u) : 0u;
        } else {
            const float4 a0 = L.g0(e), a1 = L.g1(e);
            if (COEF && part == 0) {
                const PowerCoef k = power_coeffs(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, tx0 + 7.5f, ty0 + 7.5f);
                L.coef[CS * e] = make_float4(k.q0, k.qx, k.qy, k.qxx);
                L.coef[CS * e + 1] = make_float4(k.qxy, k.qyy, a1.y, a1.w);
                if (CS == 3) L.coef[CS * e + 2] = L.rec[L.part(e, 2)];   // channels 0 .. 3 of the record
            }
            CullP cp;
            if (Rec<CH>::CULL >= 0 && RQL == Rec<CH>::RQ) {
                constexpr int CO = Rec<CH>::CULL >= 0 ? Rec<CH>::CULL : 0;
                cp.hx = L.f(e, CO); cp.hy = L.f(e, CO + 1); cp.tauq = L.f(e, CO + 2); cp.ia = L.f(e, CO + 3); cp.ic = L.f(e, CO + 4);
            } else {
                cp = cull_params(a0.z, a0.w, a1.x, a1.y);
            }
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const int ww = BPT * part + j;
                const float x0 = tx0 + (float)(8 * (ww & 1)), y0 = ty0 + (float)(8 * (ww >> 1));
                // quarter bits from per-quarter tests that are exact on their own (rows of 16 channels and more) subsume the block's
                // exact test -- the quarters' rectangles of pixel centres tile the block's -- so the block is not tested there
                // (70 VALU per thread and super-batch less in the wide forward's cull); BLEND_TANQ_GATE 0: the same for the
                // tangent-plane quarters of the narrow rows (3.00 instead of 2.95 quarters per pair, tools/cull_model.py)
                constexpr bool QEX = BLEND_EXACTQ == 2 || (BLEND_EXACTQ == 1 && CH >= 16);
                constexpr bool GATE = !(SUB && (QEX || (BLEND_TANQ && CH < 16 && !BLEND_TANQ_GATE)));
                // pred: does block ww still need entry e?  bool, or (forward) the mask of its 4x4 quarters that do -- a quarter whose 16
                // pixels are all saturated keeps nothing while the block's other quarters go on
                const unsigned live = live_quarters(pred(e, ww));
                const bool kb = live != 0u && (!GATE || cull_test(a0.x, a0.y, a0.z, a0.w, a1.x, cp, x0, x0 + 7.f, y0, y0 + 7.f));
                if (SUB) {
                    unsigned m = 0u;
                    if (kb) {
                        // distance from the centre to the pixel-centre range of each half along x and y (0 inside)
                        const float ax0 = fmaxf(fmaxf(x0 - a0.x, a0.x - (x0 + 3.f)), 0.f), ax1 = fmaxf(fmaxf(x0 + 4.f - a0.x, a0.x - (x0 + 7.f)), 0.f);
                        const float ay0 = fmaxf(fmaxf(y0 - a0.y, a0.y - (y0 + 3.f)), 0.f), ay1 = fmaxf(fmaxf(y0 + 4.f - a0.y, a0.y - (y0 + 7.f)), 0.f);
                        const bool bx0_ = ax0 <= cp.hx, bx1_ = ax1 <= cp.hx, by0_ = ay0 <= cp.hy, by1_ = ay1 <= cp.hy;
                        m = (bx0_ && by0_ ? 1u : 0u) | (bx1_ && by0_ ? 2u : 0u) | (bx0_ && by1_ ? 4u : 0u) | (bx1_ && by1_ ? 8u : 0u);
#if BLEND_TANQ_NOBOX
                        if (BLEND_TANQ && CH < 16) m = cp.hx < 0.f ? 0u : 15u;   // experiment: the tangent test alone
#endif
                        if (BLEND_TANQ && CH < 16) {
                            // TANGENT-PLANE test of the four quarters, branch-free: q(d) = d^T Q d is convex, so over the quarter's
                            // rectangle of pixel centres (half extents 1.5 around its centre c) q >= q(c) - 3 (|Qc|_x + |Qc|_y); a
                            // quarter whose bound exceeds tau holds no pixel with alpha >= 1/255.  Keeps 2.95 quarters per (tile,
                            // splat) pair where the bounding box keeps 3.20 and the exact test 2.81 (tools/cull_model.py) for
                            // ~35 VALU per block, the four quarters sharing Q c by increments.
                            const float dxc = x0 + 1.5f - a0.x, dyc = y0 + 1.5f - a0.y;
                            const float cA = a0.z, cB = a0.w, cC = a1.x;
                            const float tx0 = cA * dxc + cB * dyc, ty0 = cB * dxc + cC * dyc;
                            const float ex = fmaxf(fabsf(dxc - 1.5f), fabsf(dxc + 5.5f)), ey = fmaxf(fabsf(dyc - 1.5f), fabsf(dyc + 5.5f));
                            const float thr = cp.tauq + 4e-6f * (cA + cC + 2.f * fabsf(cB)) * (ex * ex + ey * ey);   // cull_test's rounding bound
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float fa = (float)(4 * (q & 1)), fb = (float)(4 * (q >> 1));
                                const float tx = tx0 + (fa * cA + fb * cB), ty = ty0 + (fa * cB + fb * cC);
                                const float qc = (dxc + fa) * tx + (dyc + fb) * ty;
                                if (qc - 3.f * (fabsf(tx) + fabsf(ty)) > thr) m &= ~(1u << q);   // (NaN / inf threshold: kept)
                            }
                        }
                        if (BLEND_EXACTQ == 2 || (BLEND_EXACTQ == 1 && CH >= 16)) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float qx0 = x0 + (float)(4 * (q & 1)), qy0 = y0 + (float)(4 * (q >> 1));
                                if (((m >> q) & 1u) && !cull_test(a0.x, a0.y, a0.z, a0.w, a1.x, cp, qx0, qx0 + 3.f, qy0, qy0 + 3.f)) m &= ~(1u << q);
                            }
                        }
                    }
                    k[j] = m & live;
                } else {
                    k[j] = kb ? 1u : 0u;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j) flags[j] = (unsigned char)k[j];
    if (gflags && e < nb) {  // the same bytes, for the backward (thread (e, part) owns bytes BPT part .. of word e)
        unsigned char *g = reinterpret_cast<unsigned char *>(gflags) + 4 * e + BPT * part;
        if (BPT == 4) *reinterpret_cast<unsigned int *>(g) = k[0] | (k[1 % BPT] << 8) | (k[2 % BPT] << 16) | (k[3 % BPT] << 24);
        else if (BPT == 2) *reinterpret_cast<unsigned short *>(g) = (unsigned short)(k[0] | (k[1 % BPT] << 8));
        else *g = (unsigned char)k[0];
    }
}

// wave w's order-preserving survivor list from the flag bytes; returns the count.
template <int CH, int SB, bool COEF, int XR, bool SWZ, int RQL>
__device__ __forceinline__ int build_list(TileLDS<CH, SB, COEF, XR, SWZ, RQL> &L, int w, int lane) {
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < SB / WAVE; ++r) {
        const int e = r * WAVE + lane;
        const bool keep = ((L.keep[e] >> (8 * w)) & 0xffu) != 0u;
        const unsigned long long m = __ballot(keep);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep) L.list[w][cnt + before] = (unsigned short)e;
        cnt += __popcll(m);
    }
    if (lane < 16) L.list[w][cnt + lane] = (unsigned short)SB;  // pad: the unrolled loops read slot SB (inert record)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

// The forward's cull decisions instead of a second cull (BlendArgs::cull_flags): thread e turns the entry's flag byte into
// the keep word tile_cull would have produced (byte w != 0: wave w replays the entry), with the same predicate.
template <int CH, int SB, bool COEF, int XR, bool SWZ, int RQL, typename Pred>
__device__ __forceinline__ void keep_from_flags(TileLDS<CH, SB, COEF, XR, SWZ, RQL> &L, int tid, int nb, unsigned flags, Pred pred) {
    if (tid < SB) {
        unsigned kw = 0u;
        if (tid < nb) {
#pragma unroll
            for (int ww = 0; ww < 4; ++ww)
                if (((flags >> (8 * ww)) & 0xffu) && pred(tid, ww)) kw |= 1u << (8 * ww);
        }
        L.keep[tid] = kw;
    }
}

// ---- survivors carried over a super-batch boundary (strip-walk backward kernels).  A chunk of the matrix-core kernels
// replays 16 survivors of a wave's list; a list of 38 costs three chunks, the third 62 % empty.  Instead, the last
// (list length mod 16) survivors of a super-batch wait in the wave's carry rows of the staging area (payload copied, list
// position and pair slot kept) and head the wave's list of the next super-batch -- they are the deepest splats of it, the
// replay order is unchanged; only the tile's last super-batch pads its final chunk.  The slabs are indexed by list POSITION
// (CAP rows per wave; a longer list takes another round of chunks + combine); an entry's record is stored by the combine
// of its own super-batch without the waves that carried it, and those add their part one super-batch later with float
// atomics (<= 15 records per wave and super-batch).
template <int SB, bool POS = true>
struct CarryLDS {
    static constexpr int CQ = 15;
    // POS (slab rows by list position): byte w = list position of entry e in wave w's list; 255 = not replayed by that wave now
    unsigned int pos4[POS ? SB : 1];
    int cslot[4][16];          // pair slot of the carried survivors
    int more[4];               // wave w has chunks left for another round
};

// build_list behind `base` carried survivors: positions base .. base + cnt - 1, and the entries' positions
template <int CH, int SB, bool COEF, int XR, bool SWZ, int RQL, bool POS>
__device__ __forceinline__ int build_list_at(TileLDS<CH, SB, COEF, XR, SWZ, RQL> &L, CarryLDS<SB, POS> &C, int w, int lane, int base) {
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < SB / WAVE; ++r) {
        const int e = r * WAVE + lane;
        const bool keep = ((L.keep[e] >> (8 * w)) & 0xffu) != 0u;
        const unsigned long long m = __ballot(keep);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep) L.list[w][base + cnt + before] = (unsigned short)e;
        if (POS) reinterpret_cast<unsigned char *>(C.pos4)[4 * e + w] = keep ? (unsigned char)(base + cnt + before) : (unsigned char)255;
        cnt += __popcll(m);
    }
    if (lane < 16) L.list[w][base + cnt + lane] = (unsigned short)SB;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

// The last `left` survivors of wave w's list (positions first .. first + left - 1, all entries of the staged super-batch)
// become its carry rows j0 .. j0 + left - 1: payload (the RQC leading chunks: geometry + features; chunk 1's w = the list
// position in the tile, which the replay's `q < ncontrib` test needs), pair slot, list head; their position bytes become
// 255 so that the combine leaves this wave out.
template <int CH, int SB, bool COEF, int XR, bool SWZ, bool POS>
__device__ __forceinline__ void carry_out(TileLDS<CH, SB, COEF, XR, SWZ> &L, CarryLDS<SB, POS> &C, int w, int lane, int first, int left,
                                          int j0, int top, const int *slots) {
    constexpr int RQC = 2 + (CH + 3) / 4, CQ = CarryLDS<SB>::CQ;
    for (int idx = lane; idx < left * RQC; idx += WAVE) {
        const int k = idx / RQC, part = idx - k * RQC;
        const int e = L.list[w][first + k];
        float4 v = L.rec[L.part(e, part)];
        if (part == 1) v.w = __int_as_float(top - e);
        L.rec[L.part(SB + 1 + CQ * w + j0 + k, part)] = v;
    }
    int e = 0;
    if (lane < left) {
        e = L.list[w][first + lane];
        C.cslot[w][j0 + lane] = slots[top - e];
        if (POS) reinterpret_cast<unsigned char *>(C.pos4)[4 * e + w] = (unsigned char)255;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // (the list reads above are done before the head of the list is rewritten)
    if (lane < left) L.list[w][j0 + lane] = (unsigned short)(SB + 1 + CQ * w + j0 + lane);
}

// ------------------------------------------------------------------ forward
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
struct FwdCfg {
    static constexpr int SB = CH <= 8 ? BLEND_FWD_SB : 128;
};

// Wide rows (16 .. 32 channels), BLEND_FWD_MFMA: the channel sums F[p, c] += w[p, s] f[s, c] are a product over (pixels x
// splats x channels) and run on the matrix pipe.  Lane = pixel as before for the alpha / transmittance chain, but the 16 lanes of
// a DPP row ARE one 4x4 quarter (lane -> pixel map below), a trip evaluates four survivors of the quarter's list, and a 4x4
// transpose of the four weight registers across the rows (two v_permlane32_swap + two v_permlane16_swap) yields the A operand
// of every quarter at once: A_g[m = pixel i of quarter g][k = trip] in lane 16 k + i.  B_g[k][n] = channel n of quarter g's
// k-th survivor is one ds_read_b32 per lane from the staged record; D_g (16 pixels x 16 channels) accumulates over the trips.
// An f32 MFMA is the ascending fma chain over k starting from C (profiles/r02_mfma_fma_chain_probe.json), i.e. bit for bit the
// F += f * w chain of the lane = pixel loop in the same splat order.  Per four survivors: 8 MFMAs (256 FP32-pipe cycles; 4 with
// 16 channels) + 4 swaps instead of 4 x CH FMAs (384 cycles at 24 channels, 512 at 32) and CH / 4 broadcast ds_read_b128 each.
#ifndef BLEND_FWD_MFMA
#define BLEND_FWD_MFMA 1
#endif

#ifndef BLEND_FWD_MF_MINW
#define BLEND_FWD_MF_MINW 3
#endif
template <int CH, bool ENH, bool BIAS, bool EXACT>
__global__ void __launch_bounds__(256, (CH <= 4 ? BLEND_FWD_MINW : CH <= 8 ? BLEND_FWD8_MINW : CH <= 16 ? 4 : (BLEND_FWD_MFMA && !BIAS) ? BLEND_FWD_MF_MINW : 3))
blend_fwd_kernel(const BlendArgs B) {
    constexpr int SB = FwdCfg<CH>::SB;
    constexpr bool MF = BLEND_FWD_MFMA && CH >= 16 && !BIAS;
    constexpr int NCB = (CH + 15) / 16;           // MF: 16-channel blocks of the product
    constexpr int U = MF ? 4 : CH <= 8 ? BLEND_FWD_U : 2;  // survivors evaluated per trip
    constexpr int RB = Rec<CH>::RS * 4;           // bytes per record
    constexpr int RM = RB / 32;                   // record offset = RM * coefficient-block offset
    // rows of at most four channels carry them in the coefficient block (48 bytes): the survivor's block offset is the only
    // address of an evaluation (no record offset to derive from it: one VALU instruction per evaluated (pixel, splat) less)
    constexpr int CS = (!BIAS && CH <= 4) ? 3 : 2;
    constexpr int CSB = 16 * CS;                  // bytes per coefficient block = unit of the list entries
    static_assert((SB + 1) * CSB <= 65536 && SB % U == 0 && RB % 32 == 0, "offsets must fit the 16-bit list entries");
    static_assert(BIAS ? CSB == 32 : true, "with a bias the list entries address the records (32-byte unit)");
    __shared__ TileLDS<CH, SB, !BIAS, 0, false, Rec<CH>::RQ, CS> L;
    __shared__ __attribute__((aligned(16))) unsigned int s_qlist[4][4][SB];  // [wave][quarter] survivor lists (32 e: coefficient-block byte offsets)
    __shared__ int s_done[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    // pixel of the lane inside the wave's 8x8 block: row-major, or (MF) quarter-major -- DPP row q = lanes 16 q .. 16 q + 15 = the
    // 4x4 quarter q (x half = q & 1, y half = q >> 1), row-major inside it
    const int lx = MF ? ((lane >> 4) & 1) * 4 + (lane & 3) : (lane & 7);
    const int ly = MF ? (lane >> 5) * 4 + ((lane >> 2) & 3) : (lane >> 3);
    const int px = bx + lx, py = by + ly;
    const float pxf = (float)px, pyf = (float)py;
    // pixel relative to the tile centre and its monomials (exact in f32)
    const float x = (float)((w & 1) * 8 + lx) - 7.5f, y = (float)((w >> 1) * 8 + ly) - 7.5f;
    const float xx = x * x, xy = x * y, yy = y * y;
    const int cn = EXACT ? CH : A.cn;
    f32x4 D[MF ? 4 : 1][MF ? NCB : 1];   // MF: D[g][c][j] = sum of pixel 4 (lane >> 4) + j of quarter g, channel 16 c + (lane & 15)
    if (MF) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < NCB; ++c) D[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const bool inside = (px < A.W) && (py < A.H);
    float T = inside ? 1.0f : -1.0f, F[CH];   // T < 0: the pixel is finished, |T| its final transmittance
    int last = 0, layer = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) F[k] = 0.f;
    const int2 range = A.tile_range[tile];
    const int n = range.y - range.x;

    if (tid < Rec<CH>::RQ) L.rec[SB * Rec<CH>::RQ + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // inert slot SB
    if (!BIAS && tid < CS) L.coef[CS * SB + tid] = make_float4(tid == 0 ? -__builtin_inff() : 0.f, 0.f, 0.f, 0.f);  // inert entry: q0 = log2(0)
    // list position of (entry e, super-batch b): forward walk
    auto pos = [n](int e, int b) { const int q = b * SB + e; return q < n ? q : -1; };
    Stager<CH, SB> st;
    st.load_ids(A, tid, range.x, pos, 0);
    st.load_payload(A, tid);
    st.load_ids(A, tid, range.x, pos, 1);

    for (int base = 0, batch = 0; base < n; base += SB, ++batch) {
        // quarters of the wave's block whose 16 pixels are all finished (bit q; 15 = the whole block): their lists end here -- a
        // quarter saturates before its block does, and the backward gets the shorter lists through the cull words
        unsigned qd;
        {
            const unsigned long long dm = __ballot(T < 0.f);
            if (MF) {   // DPP row q = quarter q
                qd = ((dm & 0xffffull) == 0xffffull ? 1u : 0u) | (((dm >> 16) & 0xffffull) == 0xffffull ? 2u : 0u) |
                     (((dm >> 32) & 0xffffull) == 0xffffull ? 4u : 0u) | ((dm >> 48) == 0xffffull ? 8u : 0u);
            } else {    // lane = 8 y + x: quarter (x >> 2) + 2 (y >> 2)
                constexpr unsigned long long Q0 = 0x000000000F0F0F0Full, Q1 = 0x00000000F0F0F0F0ull;
                qd = ((dm & Q0) == Q0 ? 1u : 0u) | ((dm & Q1) == Q1 ? 2u : 0u) | ((dm & (Q0 << 32)) == (Q0 << 32) ? 4u : 0u) |
                     ((dm & (Q1 << 32)) == (Q1 << 32) ? 8u : 0u);
            }
        }
        const bool alld = qd == 15u;
        if (lane == 0) s_done[w] = (int)qd;
        const int nb = imin_(SB, n - base);
        st.park(L, tid);
        st.load_payload(A, tid);                       // payload of the next super-batch
        st.load_ids(A, tid, range.x, pos, batch + 2);  // ids two ahead
        __syncthreads();
        if ((s_done[0] & s_done[1] & s_done[2] & s_done[3]) == 15) break;  // every pixel of the tile is saturated
        // (the keep words also go to A.cull_flags: what the backward needs of this cull -- a saturated block keeps nothing)
        tile_cull<CH, SB, BIAS, true, !BIAS>(L, tid, nb, (float)(tx * TILE), (float)(ty * TILE), [&](int, int ww) -> unsigned { return BLEND_FWD_QDONE ? 15u & ~(unsigned)s_done[ww] : (s_done[ww] != 15 ? 15u : 0u); },
                                             A.cull_flags ? A.cull_flags + range.x + base : nullptr);
        __syncthreads();
        if (!alld) {
            // one order-preserving survivor list per 4x4 quarter of the wave's block: the 16 lanes of a quarter walk
            // their own list (a splat is evaluated only on the quarters its bounding box reaches), the wave loops to
            // the longest of the four.  List entries are byte offsets of the entries' coefficient blocks (32 e); every
            // list is padded with the inert entry up to the common trip count, so a trip is one 8-byte list read + U
            // block reads, with no bounds test.
            int cq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cq[q] = 0;
#pragma unroll
            for (int r = 0; r < SB / WAVE; ++r) {
                const int e = r * WAVE + lane;
                const unsigned bits = (L.keep[e] >> (8 * w)) & 0xffu;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool keep = (bits >> q) & 1u;
                    const unsigned long long m = __ballot(keep);
                    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (keep) s_qlist[w][q][cq[q] + before] = (unsigned)(e * CSB);
                    cq[q] += __popcll(m);
                }
            }
            const int cnt = imax_(imax_(cq[0], cq[1]), imax_(cq[2], cq[3]));
            const int cntU = (cnt + U - 1) / U * U;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll 1
                for (int i = cq[q] + lane; i < cntU; i += WAVE) s_qlist[w][q][i] = (unsigned)(SB * CSB);  // log2(o) = -inf -> alpha 0
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int myq = MF ? (lane >> 4) : ((lane >> 2) & 1) + 2 * ((lane >> 5) & 1);  // quarter of this lane's pixel
            const unsigned int *mylist = s_qlist[w][myq];
            const char *recb = reinterpret_cast<const char *>(L.rec);
            const char *cfb = reinterpret_cast<const char *>(L.coef);
            int lastoff = -1;  // block offset of the last splat applied in this super-batch
            // MF: byte offsets of this lane's column (channel) in the B operands of block 0 and, relative to it, of block 1
            const int chan_off = 4 * (lane & 15);
            const int chan_off1 = (16 + (lane & 15) < CH) ? 64 : -chan_off;
            for (int j0 = 0; j0 < cntU; j0 += U) {
                unsigned off[U];
                static_assert(U == 2 || U == 4, "a trip reads its list entries as one 8- or 16-byte word");
#pragma unroll
                for (int u = 0; u < U; ++u) off[u] = mylist[j0 + u];  // adjacent 32-bit entries: one ds_read_b64 / b128 per trip, no unpacking
                float bq[MF ? 4 : 1][MF ? NCB : 1];   // MF: B operands, requested before the trip's arithmetic (two LDS round trips)
                if constexpr (MF) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned og = s_qlist[w][g][j0 + (lane >> 4)];   // quarter g's survivor of trip lane >> 4
                        const char *fb = recb + og * RM + 32 + chan_off;
                        // (block 1 of a 20 / 24-channel row: columns past the row read channel 0 again -- finite, never stored)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) bq[g][c] = *reinterpret_cast<const float *>(fb + (c == 0 ? 0 : chan_off1));
                    }
                }
                float4 g0[U], g1[U];
                float alpha[U];  // 0 where the splat does not touch the pixel
                bool aok[U];     // alpha[u] != 0 (a wave mask in scalar registers: the compare that zeroed alpha[u])
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const char *src = BIAS ? recb + off[u] * RM : cfb + off[u];
                    g0[u] = *reinterpret_cast<const float4 *>(src);
                    g1[u] = *reinterpret_cast<const float4 *>(src + 16);
                }
                bool any = false;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (BIAS) {
                        const float q = neg_power_factored(g0[u].x - pxf, g0[u].y - pyf, g0[u].z, g0[u].w, g1[u].x);
                        const float a = fminf(0.99f, __builtin_fmaf(g1[u].y, exp_neg(q), g1[u].z));
                        aok[u] = !(q < 0.f) && !(a < (1.0f / 255.0f));
                        alpha[u] = aok[u] ? a : 0.f;
                    } else {
                        const float pw = power_poly(g0[u], g1[u], x, y, xx, xy, yy);
                        bool pw_ok;
                        const float a = fminf(0.99f, exp2_guard(pw, pw_ok));
                        aok[u] = pw_ok && !(a < (1.0f / 255.0f));
                        alpha[u] = aok[u] ? a : 0.f;
                    }
                    if (BLEND_FWD_TRIPTEST) any = any || (alpha[u] > 0.f);
                }
                if (BLEND_FWD_TRIPTEST && __builtin_amdgcn_ballot_w64(any && T > 0.f) == 0ull) continue;
                // Branch-free compositing.  A finished pixel carries its final T NEGATED: T (1 - alpha) < 0.0001 holds for it
                // again ("saturated"), so nothing applies and no `done` predicate is kept; alpha = 0 (splat skipped on this
                // pixel) multiplies T by 1 and adds f * 0: the pixel's values do not change by a bit.
                float wq[MF ? U : 1];   // MF: the trip's weights, then (transposed) the quarters' A operands
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float f[MF ? 1 : CH];
                    if (!MF) {
                        // (CS == 3: the address is ready with the list entry -- without this tie to alpha[u] the scheduler requests
                        // the channels of all U survivors up front and the twelve registers they hold spill elsewhere)
                        if (CS == 3) asm volatile("" : "+v"(off[u]) : "v"(alpha[u]));
                        const float4 *fq = CS == 3 ? reinterpret_cast<const float4 *>(cfb + off[u] + 32)             // behind the coefficients
                                                   : reinterpret_cast<const float4 *>(recb + off[u] * RM + 32);  // 16-byte chunks of the record
#pragma unroll
                        for (int k = 0; k < CH; k += 4) {
                            const float4 v = fq[k / 4];
                            if (k + 0 < CH) f[k + 0] = v.x;
                            if (k + 1 < CH) f[k + 1] = v.y;
                            if (k + 2 < CH) f[k + 2] = v.z;
                            if (k + 3 < CH) f[k + 3] = v.w;
                        }
                    }
                    const float nT = T * (1.f - alpha[u]);
                    const bool sat = nT < 0.0001f;   // reference: the splat that would take T below 1e-4 ends the pixel, unapplied
                    const float wgt = sat ? 0.f : alpha[u] * T;
                    // applied: alpha >= 1/255 and not saturating -- then T (1 - alpha) >= 1e-4, so T > 0 and alpha T > 0: the
                    // same predicate as wgt > 0, from the two masks already in scalar registers (no third compare)
                    // (narrow rows; the wide kernels keep the compare: their scalar unit is the busier one)
                    const bool app = CH <= 8 ? (aok[u] && !sat) : wgt > 0.f;
                    if (MF) {
                        wq[u] = wgt;
                    } else {
#pragma unroll
                        for (int k = 0; k < CH; ++k) F[k] += f[k] * wgt;
                    }
                    T = sat ? -fabsf(T) : nT;
                    lastoff = app ? (int)off[u] : lastoff;
                    if (ENH) {
                        if (app && (A.trunc || layer < A.K)) {
                            // (staging the ids in LDS and writing the tile's rows coalesced at the end was slower:
                            // 23-channel row, K = 20: 179 vs 161 us per frame)
                            const size_t pix = (size_t)A.W * (size_t)py + px;
                            A.gs_idx[pix * A.K + layer] = __float_as_int(g1[u].w);
                            layer++;
                            if (A.trunc && layer >= A.K) T = -T;
                        }
                    }
                }
                if constexpr (MF) {
                    // rows <-> registers: wq[g] becomes (w_0 | w_1 | w_2 | w_3)[row g] = quarter g's A operand
                    typedef unsigned u32x2_f __attribute__((ext_vector_type(2)));
                    const u32x2_f s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wq[0]), __float_as_uint(wq[2]), false, false);
                    const u32x2_f s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wq[1]), __float_as_uint(wq[3]), false, false);
                    const u32x2_f a01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                    const u32x2_f a23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                    const float Aq[4] = {__uint_as_float(a01[0]), __uint_as_float(a01[1]), __uint_as_float(a23[0]), __uint_as_float(a23[1])};
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) D[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aq[g], bq[g][c], D[g][c], 0, 0, 0);
                }
            }
            last = lastoff >= 0 ? base + lastoff / CSB + 1 : last;
        }
        __syncthreads();
    }
    if constexpr (MF) {
        // D -> the lane's own F[]: a quarter at a time through the wave's 16 x 36 floats of the (dead) staging area.  (Every
        // thread left the loop behind a barrier: nobody reads the records any more.)
        constexpr int ST = 36;
        static_assert(4 * 16 * ST * 4 <= (int)sizeof(L.rec), "four waves' quarter buffers fit the staged records");
        float *so = reinterpret_cast<float *>(L.rec) + w * (16 * ST);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) so[(4 * (lane >> 4) + j) * ST + 16 * c + (lane & 15)] = D[g][c][j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float4 *src = reinterpret_cast<const float4 *>(so + (lane & 15) * ST);
            const bool mine = (lane >> 4) == g;
#pragma unroll
            for (int k = 0; k < CH; k += 4) {
                const float4 v = src[k / 4];
                if (k + 0 < CH) F[k + 0] = mine ? v.x : F[k + 0];
                if (k + 1 < CH) F[k + 1] = mine ? v.y : F[k + 1];
                if (k + 2 < CH) F[k + 2] = mine ? v.z : F[k + 2];
                if (k + 3 < CH) F[k + 3] = mine ? v.w : F[k + 3];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (inside) {
        const size_t HW = (size_t)A.H * A.W;
        const size_t pix = (size_t)A.W * (size_t)py + px;
        T = fabsf(T);
        A.final_T[pix] = T;
        A.ncontrib[pix] = last;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < cn) A.out[(size_t)(A.c0 + k) * HW + pix] = F[k] + T * (A.bgc ? A.bgc[A.c0 + k] : A.bg);
        if (ENH)
            for (int l = layer; l < A.K; ++l) A.gs_idx[pix * A.K + l] = -1;  // unused slots (reference: -1 init)
    }
}

// ------------------------------------------------------------------ wave reductions for the backward
// In-place DPP adds (the compiler only fuses the row_shr steps; row_bcast needs the "keep masked
// rows" form).  Four values per block so that dependent DPP instructions are >= 2 issue slots apart
// (VALU-write -> DPP-read hazard); s_nop covers the producers in front of the block.  Sum lands in lane 63.
#define DPP4(op)                                   \
    "v_add_f32_dpp %0, %0, %0 " op "\n\t"          \
    "v_add_f32_dpp %1, %1, %1 " op "\n\t"          \
    "v_add_f32_dpp %2, %2, %2 " op "\n\t"          \
    "v_add_f32_dpp %3, %3, %3 " op "\n\t"

__device__ __forceinline__ void wave_sum4_to_lane63(float &a, float &b, float &c, float &d) {
    asm volatile("s_nop 1\n\t" DPP4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                     DPP4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                         DPP4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                             DPP4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                                 DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                                     DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

#define DPP3(op)                                   \
    "v_add_f32_dpp %0, %0, %0 " op "\n\t"          \
    "v_add_f32_dpp %1, %1, %1 " op "\n\t"          \
    "v_add_f32_dpp %2, %2, %2 " op "\n\t"

// three interleaved chains: dependent DPP adds are still 2 issue slots apart
__device__ __forceinline__ void wave_sum3_to_lane63(float &a, float &b, float &c) {
    asm volatile("s_nop 1\n\t" DPP3("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                     DPP3("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                         DPP3("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                             DPP3("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                                 DPP3("row_bcast:15 row_mask:0xa bank_mask:0xf")
                                     DPP3("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c));
}

// N values -> lane 63, in blocks of 4 and 3 (N = 4a + 3b whenever N >= 6; small N fall back to the builtin chain)
template <int N>
__device__ __forceinline__ void wave_sum_n_to_lane63(float (&v)[N]) {
    constexpr int B3 = (N >= 6 || N == 3) ? ((4 - (N & 3)) & 3) : 0;  // number of 3-blocks: N - 3*B3 divisible by 4
    constexpr int N3 = 3 * B3;
    constexpr int N4 = (N - N3) & ~3;
#pragma unroll
    for (int k = 0; k < N4; k += 4) wave_sum4_to_lane63(v[k], v[k + 1], v[k + 2], v[k + 3]);
#pragma unroll
    for (int k = N4; k < N4 + N3; k += 3) wave_sum3_to_lane63(v[k], v[k + 1], v[k + 2]);
#pragma unroll
    for (int k = N4 + N3; k < N; ++k) v[k] = wave_sum_to_lane63(v[k]);
}

// per-pixel replay of one survivor in the backward (shared by pair and atomic kernels); GradLayout: common.h
template <int CH, bool ABS, bool BIAS>
__device__ __forceinline__ void replay_one(const float4 &g0, const float4 &g1, const float (&f)[CH], float dx, float dy,
                                           float G, float a, float Tf, float bgdot, const float (&gp)[CH], float &T,
                                           float (&acc)[CH], bool &done, float (&r)[GradLayout<ABS, BIAS>::NG + CH]) {
    using GL = GradLayout<ABS, BIAS>;
    const float r1a = __builtin_amdgcn_rcpf(1.f - a);
    T = T * r1a;
    const float wgt = a * T;
    float dLa = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        dLa += (f[k] - acc[k]) * gp[k];
        r[GL::NG + k] = wgt * gp[k];
        acc[k] = a * f[k] + (1.f - a) * acc[k];  // == the reference's deferred accum_rec update
    }
    dLa *= T;
    dLa += (-Tf * r1a) * bgdot;
    const float dLG = g1.y * dLa;
    const float gx_ = -G * dx * g0.z - G * dy * g0.w;
    const float gy_ = -G * dy * g1.x - G * dx * g0.w;
    r[0] = dLG * gx_; r[1] = dLG * gy_;
    r[2] = -0.5f * G * dx * dx * dLG;
    r[3] = -G * dx * dy * dLG;
    r[4] = -0.5f * G * dy * dy * dLG;
    r[5] = G * dLa;
    if constexpr (ABS) {
        r[GL::I_ABS] = fabsf(r[0]); r[GL::I_ABS + 1] = fabsf(r[1]);
    }
    if constexpr (BIAS) {
        r[GL::I_BIAS] = dLa;
        done = T < 0.0001f;
    }
}

// ------------------------------------------------------------------ backward, atomic-free ("pair" mode)
template <int CH, bool ABS, bool BIAS>
struct PairCfg {
    static constexpr int NG = GradLayout<ABS, BIAS>::NG;  // ux uy ca cb cc o [ax ay] [bias]
    static constexpr int NC = NG + CH;       // used floats per pair record
    static constexpr int NCP = PAIR_STRIDE(NC);  // record stride in pair_buf
    static constexpr int SB = 64;
};

template <int CH, bool ABS, bool BIAS, bool EXACT>
__global__ void __launch_bounds__(256, (CH <= 8 ? BLEND_BWD_MINW : 1))
blend_bwd_pair_kernel(const BlendArgs B) {
    using Cfg = PairCfg<CH, ABS, BIAS>;
    constexpr int SB = Cfg::SB, NC = Cfg::NC, NCP = Cfg::NCP;
    constexpr int U = CH <= 8 ? BLEND_BWD_U : 1;
    __shared__ TileLDS<CH, SB, !BIAS> L;
    __shared__ float s_acc[4][SB * NC];          // private slab per wave: plain stores, no atomics
    __shared__ unsigned long long s_mask[4];     // which entries of the super-batch the wave wrote
    __shared__ int s_wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    float *const pair_buf = B.pair_buf + (size_t)frame * (size_t)B.cap * NCP;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float x = (float)((w & 1) * 8 + (lane & 7)) - 7.5f, y = (float)((w >> 1) * 8 + (lane >> 3)) - 7.5f;
    const float xx = x * x, xy = x * y, yy = y * y;
    const size_t HW = (size_t)A.H * A.W;
    const int cn = EXACT ? CH : A.cn;

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
        gp[k] = (inside && k < cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
        if (k < cn) bgdot += A.bg * gp[k];
    }
    const int wmax = wave_max_i(last);  // this wave never needs entries q >= wmax
    if (lane == 0) s_wmax[w] = wmax;
    if (tid < Rec<CH>::RQ) L.rec[SB * Rec<CH>::RQ + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // inert slot SB
    if (!BIAS && tid < 2) L.coef[2 * SB + tid] = make_float4(tid == 0 ? -__builtin_inff() : 0.f, 0.f, 0.f, 0.f);  // inert entry: q0 = log2(0)
    __syncthreads();
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    const int *slots = A.slot_sorted + range.x;
    for (int i = n * NCP + tid; i < len * NCP; i += 256) {  // entries nobody replays: zero record
        const int ql = i / NCP;
        pair_buf[(size_t)slots[ql] * NCP + (i - ql * NCP)] = 0.f;
    }
    if (n <= 0) {
        if (A.dbg_T_front && inside) A.dbg_T_front[pix] = T;
        return;
    }

    // reverse walk: entry e of super-batch b sits at list position n-1 - b*SB - e
    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };  // negative = past the front
    Stager<CH, SB> st;
    st.load_ids(A, tid, range.x, pos, 0);
    st.load_payload(A, tid);
    st.load_ids(A, tid, range.x, pos, 1);

    int batch = 0;
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        st.park(L, tid);
        st.load_payload(A, tid);                       // payload of the next super-batch
        st.load_ids(A, tid, range.x, pos, batch + 2);  // ids two ahead
        __syncthreads();

        tile_cull<CH, SB, BIAS, false, !BIAS>(L, tid, nb, (float)(tx * TILE), (float)(ty * TILE),
                                [&](int e, int ww) { return top - e < s_wmax[ww]; });
        __syncthreads();
        const int cnt = build_list(L, w, lane);
        unsigned long long wrote = 0ull;
        float *slab = s_acc[w];
        for (int j0 = 0; j0 < cnt; j0 += U) {
            int e[U];
            float4 g0[U], g1[U];
            float dx[U], dy[U], G[U], alpha[U];
            bool ok[U];
            bool any_ok = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                e[u] = L.list[w][j0 + u];
                g0[u] = L.g0(e[u]);
                g1[u] = L.g1(e[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                dx[u] = g0[u].x - pxf; dy[u] = g0[u].y - pyf;
                bool pw_ok;
                float araw;
                if (BIAS) {  // the forward's expressions (blend_fwd_kernel), bit for bit
                    const float q = neg_power_factored(dx[u], dy[u], g0[u].z, g0[u].w, g1[u].x);
                    G[u] = exp_neg(q);
                    araw = __builtin_fmaf(g1[u].y, G[u], g1[u].z);
                    pw_ok = !(q < 0.f);
                } else {
                    const float pw = power_poly(L.coef[2 * e[u]], L.coef[2 * e[u] + 1], x, y, xx, xy, yy);
                    araw = exp2_guard(pw, pw_ok);                      // = o * G (the forward's value, bit for bit)
                    G[u] = araw * __builtin_amdgcn_rcpf(g1[u].y);      // (araw > 0 only where o > 0)
                }
                alpha[u] = fminf(0.99f, araw);
                ok[u] = (j0 + u < cnt) && !done && (top - e[u] < last) && pw_ok && !(alpha[u] < (1.0f / 255.0f));
                any_ok = any_ok || ok[u];
            }
            if (__builtin_amdgcn_ballot_w64(any_ok) == 0ull) continue;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (BIAS) ok[u] = ok[u] && !done;
                if (__builtin_amdgcn_ballot_w64(ok[u]) == 0ull) continue;
                float f[CH];
                read_feat(L, e[u], f);
                float r[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) r[k] = 0.f;
                if (ok[u]) replay_one<CH, ABS, BIAS>(g0[u], g1[u], f, dx[u], dy[u], G[u], alpha[u], Tf, bgdot, gp, T, acc, done, r);
                wave_sum_n_to_lane63<NC>(r);
                if (lane == 63) {
                    float *a = slab + e[u] * NC;
#pragma unroll
                    for (int k = 0; k < NC; ++k) a[k] = r[k];
                }
                wrote |= 1ull << e[u];
            }
        }
        if (lane == 0) s_mask[w] = wrote;
        __syncthreads();
        // ---- combine the four slabs; each record (NCP floats) goes to its pair slot, consecutive lanes write
        //      consecutive floats of it: entry e <-> sorted position top - e
        {
            const int lo = top - nb + 1;
            const unsigned long long m0 = s_mask[0], m1 = s_mask[1], m2 = s_mask[2], m3 = s_mask[3];
            for (int i = tid; i < nb * NCP; i += 256) {
                const int ql = i / NCP, c = i - ql * NCP;
                const int e = nb - 1 - ql;
                float v = 0.f;
                if (c < NC) {
                    if ((m0 >> e) & 1ull) v += s_acc[0][e * NC + c];
                    if ((m1 >> e) & 1ull) v += s_acc[1][e * NC + c];
                    if ((m2 >> e) & 1ull) v += s_acc[2][e * NC + c];
                    if ((m3 >> e) & 1ull) v += s_acc[3][e * NC + c];
                }
                pair_buf[(size_t)slots[lo + ql] * NCP + c] = v;
            }
        }
        __syncthreads();
    }
    if (A.dbg_T_front && inside) A.dbg_T_front[pix] = T;
}

// sums each Gaussian's pair records (contiguous slots [goff[i-1], goff[i])) into the final gradients.
// Four lanes per Gaussian: lane `sub` owns the 16-byte chunks sub, sub + 4, ... of every record, so a quad reads up to
// 64 contiguous bytes per step (coalesced), accumulates in registers with no cross-lane traffic, and writes its own
// components with plain stores.
template <bool ABS, bool BIAS>
__device__ __forceinline__ void store_component(const BlendArgs &A, int i, int k, float v) {
    using GL = GradLayout<ABS, BIAS>;
    constexpr int NG = GL::NG;
    float *dst;
    if (k < 2) {
        dst = A.dL_duv + 2 * i + k;
        if (A.dL_dndc) {  // the tap the reference fills with dL_duv * [W/2, H/2] (alpha_blending.py:140-147)
            float *t = A.dL_dndc + 2 * i + k;
            const float tv = v * (k == 0 ? 0.5f * (float)A.W : 0.5f * (float)A.H);
            *t = A.accumulate ? *t + tv : tv;
        }
    } else if (k < 5) dst = A.dL_dconic + 3 * i + (k - 2);
    else if (k == 5) dst = A.dL_dopacity + i;
    else if (ABS && k < GL::I_ABS + 2) {
        dst = A.dL_dabs_uv + 2 * i + (k - GL::I_ABS);
        if (A.dL_dabs_ndc) {
            float *t = A.dL_dabs_ndc + 2 * i + (k - GL::I_ABS);
            const float tv = v * (k == GL::I_ABS ? 0.5f * (float)A.W : 0.5f * (float)A.H);
            *t = A.accumulate ? *t + tv : tv;
        }
    }
    else if (BIAS && k == GL::I_BIAS) dst = A.dL_dbias + i;
    else {
        if (k - NG >= A.cn) return;  // padding
        A.dL_dfeature[(size_t)i * A.C + A.c0 + (k - NG)] = v;  // features of this chunk: always plain store
        return;
    }
    *dst = A.accumulate ? *dst + v : v;
}

template <bool ABS, bool BIAS, int NCP>
__global__ void __launch_bounds__(256)
pair_reduce_kernel(const BlendArgs A) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 2, sub = t & 3;
    if (i >= A.P) return;
    const int beg = i > 0 ? A.goff_incl[i - 1] : 0, end = A.goff_incl[i];
    constexpr int NQ = NCP / 4;         // 16-byte chunks per record; lane `sub` owns chunks sub, sub + 4, ...
    constexpr int NS = (NQ + 3) / 4;
    // records requested together: a splat's records are one dependent round trip per TU of them (94 % of the splats of the
    // bench scene touch at most six tiles; two at a time was 2 .. 3 round trips for the average splat)
    constexpr int TU = NS == 1 ? 6 : NS == 2 ? 3 : 2;
    float4 a[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *base = A.pair_buf + 4 * sub;
    for (int j = beg; j < end; j += TU) {
        float4 v[TU][NS];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int c = 0; c < NS; ++c)
                v[u][c] = (j + u < end && 4 * c + sub < NQ) ? *reinterpret_cast<const float4 *>(base + (size_t)(j + u) * NCP + 16 * c)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                a[c].x += v[u][c].x; a[c].y += v[u][c].y; a[c].z += v[u][c].z; a[c].w += v[u][c].w;
            }
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        if (4 * c + sub < NQ) {
            const int k = 16 * c + 4 * sub;
            store_component<ABS, BIAS>(A, i, k + 0, a[c].x);
            store_component<ABS, BIAS>(A, i, k + 1, a[c].y);
            store_component<ABS, BIAS>(A, i, k + 2, a[c].z);
            store_component<ABS, BIAS>(A, i, k + 3, a[c].w);
        }
    }
}

// ------------------------------------------------------------------ backward, pair mode, matrix-core version
// Everything that is a product over (pixels x splats) runs on the matrix cores (v_mfma_f32_16x16x4_f32: exact
// f32 FMA chains, 16x16 outputs, K = 4 per instruction):
//   power[p,n]  = phi(p) . q(n)          K = 6 monomials (1 x y xx xy yy) of the block-centred pixel coordinates
//   cg[p,n]     = dL_dout[p,:] . f[n,:]  K = channels
//   moments[n]  = sum_p phi(p) dL/dpower[p,n]      K = pixels  -> d uv, d conic, d opacity (power is quadratic in
//   dfeat[n,c]  = sum_p dL_dout[p,c] w[p,n]        K = pixels     (u - x, v - y), so its gradients are linear in
//                                                                 the six moments)
// instead of 9..40 DPP wave reductions per (wave, splat).  Lane roles inside a wave: n = lane & 15 is one of 16
// survivors (a "chunk", deepest first), kk = lane >> 4.  The wave's 8x8 pixel block is walked as four 2x8
// strips G; in strip G lane (n, kk) owns pixels m = 4 kk + i (i = 0..3, row-major in the strip), which is exactly
// the C/D layout of the 16x16 MFMA (row 4 kk + i, column n) -- the power and cg products land in the lane that
// consumes them.  With lanes = splats the per-pixel recurrences of the reference's backward loop
// (src/alpha_blending.cu:152-249) become prefix scans over the 16-lane DPP row:
//     T_n   = T_state * prod_{q<=n} 1/(1-a_q)                (transmittance in front of splat n)
//     R_n   = R_state + sum_{q<n} a_q T_q cg_q                (colour behind splat n, already dotted with dL_dout)
//     dL/da = T_n cg_n - (R_n + T_final bg.g) / (1-a_n)
// The records written to the slabs / pair_buf are the same as blend_bwd_pair_kernel's (pair_reduce is shared).

template <int CH, bool ABS>
struct MfmaCfg {
    static constexpr int NG = GradLayout<ABS, false>::NG;
    static constexpr int NC = NG + CH;
    static constexpr int NCP = PAIR_STRIDE(NC);
    static constexpr int SB = CH <= 8 ? BLEND_MFMA_SB : (CH <= 20 ? BLEND_WIDE_SB : 64);
    static constexpr int PZ = CH;                 // a zero slot (lanes without a channel feed it to the MFMAs)
    static constexpr int PS = (CH + 1 + 3) & ~3;  // state block [T_final*bg.g, ncontrib, T_state, R_state] (16-B aligned)
    static constexpr int PW = PS + 4;             // floats per pixel record: g[CH], 0.., state block
    static constexpr int NA = (CH + 15) / 16;     // feature-gradient accumulators (16 channels each)
    static constexpr int NK = (CH + 3) / 4;       // K-slabs of the cg product (4 channels each)
    // widest records: the four waves add into ONE slab with LDS float atomics (ds_add_f32) instead of owning a private
    // slab each -- at 32 channels that is 10 KB instead of 39 KB of LDS, two workgroups per CU instead of one (828 ->
    // 748 us at BASELINE configs[4]); at 16 / 20 channels the atomics cost more than the third workgroup gains
    // (19-channel blend 750 -> 805 us), so those keep the private slabs
    // (narrow records as well -- 5 KB of LDS less, one row per entry in the combine -- was tried in round 3: ds_add_f32 costs
    // about 16 cycles per instruction of 16 lanes, 159 -> 184 us per frame at BASELINE configs[1])
    static constexpr bool SHARED = CH > 20;
    static constexpr int NSLAB = SHARED ? 1 : 4;
};

// four interleaved inclusive row scans (lane 0 first): dependent DPP instructions are 4 issue slots apart; lanes
// without a source (bound_ctrl off) keep their own value, which is what an inclusive scan needs
#define SCAN4(op, sh)                                                     \
    op " %0, %0, %0 row_shr:" sh " row_mask:0xf bank_mask:0xf\n\t"        \
    op " %1, %1, %1 row_shr:" sh " row_mask:0xf bank_mask:0xf\n\t"        \
    op " %2, %2, %2 row_shr:" sh " row_mask:0xf bank_mask:0xf\n\t"        \
    op " %3, %3, %3 row_shr:" sh " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row_scan_mul4(float &a, float &b, float &c, float &d) {
    asm volatile("s_nop 1\n\t" SCAN4("v_mul_f32_dpp", "1") SCAN4("v_mul_f32_dpp", "2") SCAN4("v_mul_f32_dpp", "4")
                     SCAN4("v_mul_f32_dpp", "8")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void row_scan_add4(float &a, float &b, float &c, float &d) {
    asm volatile("s_nop 1\n\t" SCAN4("v_add_f32_dpp", "1") SCAN4("v_add_f32_dpp", "2") SCAN4("v_add_f32_dpp", "4")
                     SCAN4("v_add_f32_dpp", "8")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// two floats to LDS from the last lane of every 16-lane row (EXEC is all ones in the callers: full waves, uniform flow)
__device__ __forceinline__ void lds_store2_lane15(float *p, float a, float b) {
    const unsigned addr = (unsigned)(size_t)p;  // LDS byte address = low 32 bits of the generic shared pointer
    asm volatile(
        "s_mov_b64 exec, %3\n\t"
        "ds_write2_b32 %0, %1, %2 offset1:1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(addr), "v"(a), "v"(b), "s"(0x8000800080008000ull)
        : "memory");
}

// four such pairs behind ONE exec switch, the pairs' places given as constant dword offsets from `base` (the instruction's own
// offset fields: no address arithmetic per pair).  Per step of the strip-walk kernels: 4 VALU and 6 SALU instructions less than
// four lds_store2_lane15 calls.
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void lds_store2x4_lane15(float *base, float a0, float b0, float a1, float b1, float a2, float b2, float a3,
                                                    float b3) {
    static_assert(O0 >= 0 && O1 >= 0 && O2 >= 0 && O3 >= 0 && O0 < 255 && O1 < 255 && O2 < 255 && O3 < 255, "ds_write2_b32 offsets are 8-bit dword counts");
    const unsigned addr = (unsigned)(size_t)base;
    asm volatile(
        "s_mov_b64 exec, %9\n\t"
        "ds_write2_b32 %0, %1, %2 offset0:%c10 offset1:%c11\n\t"
        "ds_write2_b32 %0, %3, %4 offset0:%c12 offset1:%c13\n\t"
        "ds_write2_b32 %0, %5, %6 offset0:%c14 offset1:%c15\n\t"
        "ds_write2_b32 %0, %7, %8 offset0:%c16 offset1:%c17\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(addr), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "s"(0x8000800080008000ull), "n"(O0),
          "n"(O0 + 1), "n"(O1), "n"(O1 + 1), "n"(O2), "n"(O2 + 1), "n"(O3), "n"(O3 + 1)
        : "memory");
}

// sum over the four 16-lane rows of a wave (lanes n, n + 16, n + 32, n + 48): two VALU-only swaps (gfx950 v_permlane16_swap /
// v_permlane32_swap), every lane ends up with the total
typedef unsigned u32x2_b __attribute__((ext_vector_type(2)));
// (a swap of (v, v) leaves the even rows' / lower half's values in every lane of the first result and the odd rows' / upper
// half's in the second: their sum is v + the partner's v in EVERY lane -- no select, and the same two operands as before)
__device__ __forceinline__ float rows_sum(float v, int lane) {
    (void)lane;
    {
        const u32x2_b r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // + value of lane ^ 16
    }
    {
        const u32x2_b r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // + value of lane ^ 32
    }
    return v;
}

// R[i] = base[i] + (rs[i] of the previous lane of the row; 0 for lane 0): exclusive scan from the inclusive one
__device__ __forceinline__ void row_shr1_add4(float (&R)[4], const float (&rs)[4], const float (&base)[4]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %5, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %6, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %7, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3])
        : "v"(rs[0]), "v"(rs[1]), "v"(rs[2]), "v"(rs[3]), "v"(base[0]), "v"(base[1]), "v"(base[2]), "v"(base[3]));
}

template <int CH, bool ABS, bool EXACT>
__global__ void __launch_bounds__(256, (CH <= 3 ? BLEND_MFMA_MINW : CH <= 8 ? 2 : (CH <= 20 ? BLEND_WIDE_MINW : 1)))   // (8 channels: the registers of two waves per SIMD)
blend_bwd_mfma_kernel(const BlendArgs B) {
    using Cfg = MfmaCfg<CH, ABS>;
    constexpr int SB = Cfg::SB, NG = Cfg::NG, NC = Cfg::NC, NCP = Cfg::NCP, PW = Cfg::PW, NA = Cfg::NA, NK = Cfg::NK;
    constexpr int PZ = Cfg::PZ, PS = Cfg::PS;
    constexpr int I_ABS = GradLayout<ABS, false>::I_ABS;
    constexpr bool SHARED = Cfg::SHARED;
    // survivors that do not fill a chunk are carried into the next super-batch (CarryLDS)
    constexpr bool CARRY = BLEND_CARRY && CH <= 20;
    constexpr int CQ = CarryLDS<SB>::CQ, XR = CARRY ? 4 * CQ : 0;
    // private slabs (16 / 20 channels): rows by list position, CAP per wave (whole chunks; a longer list takes another round);
    // shared slab: one row per staged entry + one per carry row, no limit
    constexpr int CAP = SHARED ? (1 << 20) : (!CARRY ? SB : ((SB == 128 && NC <= 9) ? 80 : 64));
    // the combine gives TPR threads to a record, three floats each (one 12-byte store per thread); shared slab: rows of NCS floats
    constexpr int TPR = (NC + 2) / 3, NCS = SHARED ? 3 * TPR : NC;
    constexpr int SROWS = SHARED ? SB + XR : CAP + (CARRY ? 1 : 0);
    __shared__ TileLDS<CH, SB, false, XR, BLEND_REC_SWZ != 0> L;
    __shared__ CarryLDS<SB, !SHARED> CL;
    // narrow feature rows: dL_dfeature = sum_p g[p,c] w[p,n] as per-lane FMAs over the lane's own pixels + one cross-row sum
    // per chunk, instead of four MFMAs per strip whose A operand would use 3 of its 16 rows (the matrix pipe's time is on
    // this kernel's critical path: dropping those products saved 13 % of it, the VALU form gives back a third)
    constexpr bool FEAT_VALU = CH <= 4;
    // one shared slab (LDS float atomics; rows are cleared by whoever reads them out), or a private slab per wave whose
    // row CAP stays zero (what the combine reads for a wave that has nothing for an entry)
    __shared__ float s_acc[Cfg::NSLAB][SROWS * NCS];
    // per-pixel rows [g[CH] 0.. | T_final*bg.g, ncontrib, T_state, R_state]: pixel q = 16 G + 4 kk + i of the wave's block at
    // float G * GS + kk * KS + i * PW.  The four lane groups kk of a wave read / write the rows of their own pixels in one
    // instruction; with dense rows (KS = 4 PW, a multiple of 32 dwords) all four hit the same banks (ds_write2_b32 from
    // lanes 15 / 31 / 47 / 63: 4-way) -- 8 floats of skew per group spread them
    constexpr int KS = 4 * PW + (BLEND_STATE_SKEW ? 8 : 0), GS = 4 * KS;
    __shared__ __attribute__((aligned(16))) float s_pix[4][4 * GS];
    auto pixoff = [](int q) { return (q >> 4) * GS + ((q >> 2) & 3) * KS + (q & 3) * PW; };
    __shared__ float s_mom[16 * 64];         // A operand of the moment product: [step 4 G + i][lane]
    __shared__ int s_wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int RST = B.rec_stride ? B.rec_stride : NCP;   // floats between two pair records
    float *const pair_buf = B.pair_buf + (size_t)frame * (size_t)B.cap * RST + B.rec_off;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    const float bx0 = (float)bx, by0 = (float)by;
    const float tcx = (float)(tx * TILE) + 7.5f, tcy = (float)(ty * TILE) + 7.5f;  // tile centre: origin of the power polynomial
    const float ox = (float)((w & 1) * 8) - 7.5f, oy = (float)((w >> 1) * 8) - 7.5f;  // block origin relative to it
    const int cn = EXACT ? CH : A.cn;
    const int nl = lane & 15, kk = lane >> 4;
    int wmax;
    // ---- operand tables (block-centred pixel coordinates x, y in [-3.5, 3.5]; pixel q = 8 row + column)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int st = 4 * w + r, Gs = st >> 2, is = st & 3;  // step = (strip, sub-step)
        const int q = 16 * Gs + 4 * kk + is;                 // the pixel lane-group kk handles in this step
        const float x = (float)(q & 7) - 3.5f, y = (float)(q >> 3) - 3.5f;
        // rows 0-3: 1 x y xx | rows 4-7: 1 x y xy | rows 8-11: 1 y yy 0 | rows 12-15: 0
        const int grp = nl >> 2, i = nl & 3;
        float v = 0.f;
        if (grp == 0) v = i == 0 ? 1.f : i == 1 ? x : i == 2 ? y : x * x;
        if (grp == 1) v = i == 0 ? 1.f : i == 1 ? x : i == 2 ? y : x * y;
        if (grp == 2) v = i == 0 ? 1.f : i == 1 ? y : i == 2 ? y * y : 0.f;
        s_mom[64 * st + lane] = v;
    }
    // A operand of the power product, A[m = pixel nl of strip G][k = kk]: monomials 1 x y xx | xy yy 0 0 of the pixel's
    // position relative to the TILE centre -- with the coefficients of power_coeffs() the two MFMAs below are the fma
    // chain power_poly() runs in the forward kernel, bit for bit
    float phi1[4], phi2[4];
#pragma unroll
    for (int Gs = 0; Gs < 4; ++Gs) {
        const int q = 16 * Gs + nl;
        const float x = (float)(q & 7) + ox, y = (float)(q >> 3) + oy;
        phi1[Gs] = kk == 0 ? 1.f : kk == 1 ? x : kk == 2 ? y : x * x;
        phi2[Gs] = kk == 0 ? x * y : kk == 1 ? y * y : 0.f;
    }
    {   // per-pixel constants: lane q <-> pixel (q & 7, q >> 3) of the wave's block
        const int px = bx + (lane & 7), py = by + (lane >> 3);
        const size_t HW = (size_t)A.H * A.W;
        const bool inside = (px < A.W) && (py < A.H);
        const size_t pix = (size_t)A.W * (size_t)py + px;
        const float Tf = inside ? A.final_T[pix] : 0.f;
        const int last = inside ? A.ncontrib[pix] : 0;
        float *r = s_pix[w] + pixoff(lane);
        float bgdot = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float g = (inside && k < cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
            r[k] = g;
            bgdot += A.bg * g;
        }
#pragma unroll
        for (int k = CH; k < PS; ++k) r[k] = 0.f;  // slot PZ: what lanes without a channel feed to the MFMAs
        r[PS] = 0.f;
        r[PS + 1] = __int_as_float(last);
        r[PS + 2] = Tf;   // T_state: transmittance behind the splats replayed so far
        r[PS + 3] = Tf * bgdot;  // R_state: colour behind them (dotted with dL_dout) -- the background is the deepest layer
        wmax = wave_max_i(last);  // this wave never needs entries q >= wmax
        if (lane == 0) s_wmax[w] = wmax;
    }
    if (tid < Rec<CH>::RQ) L.rec[SB * Rec<CH>::RQ + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // inert slot SB
    if (SHARED) {
        for (int i = tid; i < SROWS * NCS; i += 256) s_acc[0][i] = 0.f;
    } else if (CARRY && lane < NC) {
        s_acc[w][CAP * NC + lane] = 0.f;                                                     // the slab's zero row
    }
    __syncthreads();
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    const int *slots = A.slot_sorted + range.x;
    // combine: thread (ce, cc) of a pass writes floats [CW cc, CW cc + CW) of entry ce's record
    constexpr int CW = 3, TPE = TPR;
    constexpr int EPI = 256 / TPE;                  // pair records the 256 threads write per pass
    constexpr int NPASS = (SB + EPI - 1) / EPI;
    const int ce = tid / TPE, cc = tid - ce * TPE;  // this thread's (entry within the pass, part of the record)
    auto store_part = [&](int slot, float v0, float v1, float v2) {
        float *dst = pair_buf + (size_t)slot * RST + CW * cc;
        if (CW * cc + 2 < NCP) {
            F3 t; t.x = v0; t.y = v1; t.z = v2;
            *reinterpret_cast<F3 *>(dst) = t;       // (floats NC .. NCP - 1 of a record are padding: whatever lands there is ignored)
        } else {
            if (CW * cc < NCP) dst[0] = v0;
            if (CW * cc + 1 < NCP) dst[1] = v1;
        }
    };
    if (ce < EPI)
        for (int ql = n + ce; ql < len; ql += EPI)  // entries nobody replays: zero record
            store_part(slots[ql], 0.f, 0.f, 0.f);
    if (n <= 0) {
        if (A.dbg_T_front) {
            const int px = bx + (lane & 7), py = by + (lane >> 3);
            if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_pix[w][pixoff(lane) + PS + 2];
        }
        return;
    }

    // ---- per-lane addressing: pixel (G, kk, i) is q = 16 G + 4 kk + i
    float *pixrow = s_pix[w] + kk * KS;            // own pixel of step (G, i): pixrow + G * GS + i * PW
    const float *pixcol = s_pix[w] + pixoff(nl);   // pixel nl of strip G (cg product, A operand): pixcol + G * GS
    const float *momrow = s_mom + lane;            // + 64 * (4 G + i)
    int gch[NA], kch[NK];
#pragma unroll
    for (int q = 0; q < NA; ++q) gch[q] = 16 * q + nl < CH ? 16 * q + nl : PZ;  // channel row of the feature product
#pragma unroll
    for (int j = 0; j < NK; ++j) kch[j] = 4 * j + kk < CH ? 4 * j + kk : PZ;    // K index of the cg product

    // wide rows: the A operands of the colour and feature-gradient products are the same in every chunk (dL_dout of the
    // wave's pixels) -- held in registers instead of re-read from LDS per chunk
    constexpr bool HOIST = BLEND_WIDE_HOIST && CH > 8;
    float hcg[HOIST ? 4 : 1][HOIST ? NK : 1], hft[HOIST ? 16 : 1][HOIST ? NA : 1], hmom[HOIST ? 16 : 1];
    if (HOIST) {
#pragma unroll
        for (int G = 0; G < 4; ++G)
#pragma unroll
            for (int j = 0; j < NK; ++j) hcg[G][j] = pixcol[G * GS + kch[j]];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int q = 0; q < NA; ++q) hft[s][q] = pixrow[(s >> 2) * GS + (s & 3) * PW + gch[q]];
            hmom[s] = momrow[64 * s];
        }
    }

    // reverse walk: entry e of super-batch b sits at list position n-1 - b*SB - e
    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };  // negative = past the front
    Stager<CH, SB> st;
    st.load_ids(A, tid, range.x, pos, 0);
    st.load_payload(A, tid);
    st.load_ids(A, tid, range.x, pos, 1);
    // the forward's cull flags of the super-batch (thread e: entry e), one super-batch ahead
    auto load_flags = [&](int topb) -> unsigned {
        const int q = topb - tid;
        return (A.cull_flags && tid < SB && q >= 0) ? (unsigned)A.cull_flags[range.x + q] : 0u;
    };
    unsigned fl_next = load_flags(n - 1);

    int batch = 0;
    int ncarry = 0;  // survivors waiting in this wave's carry rows
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        st.park(L, tid);
        if (!BLEND_LATE_STAGE) {
            st.load_payload(A, tid);                       // payload of the next super-batch
            st.load_ids(A, tid, range.x, pos, batch + 2);  // ids two ahead
        }
        const unsigned fl = fl_next;
        fl_next = load_flags(top - SB);
        // pair slots of the entries this thread writes in the combine: loaded ahead of the barrier in front of it (the
        // combine used to wait for them iteration by iteration: 4.6 serial L2 round trips per super-batch)
        int sl[NPASS <= 3 ? NPASS : 1];
        auto load_slots = [&]() {
            if (NPASS <= 3) {
#pragma unroll
                for (int r = 0; r < (NPASS <= 3 ? NPASS : 1); ++r) {
                    const int ql = ce + r * EPI;
                    sl[r] = (ce < EPI && ql < nb) ? slots[top - nb + 1 + ql] : 0;
                }
            }
        };
        if (BLEND_SLOT_EARLY) load_slots();
        if (A.cull_flags) {
            // (the forward's keep words need nothing of the staged records: one barrier serves the park and the keep words)
            keep_from_flags(L, tid, nb, fl, [&](int e, int ww) { return top - e < s_wmax[ww]; });
        } else {
            __syncthreads();
            tile_cull<CH, SB, false, false, false>(L, tid, nb, (float)(tx * TILE), (float)(ty * TILE),
                                 [&](int e, int ww) { return top - e < s_wmax[ww]; });
        }
        __syncthreads();
        // every survivor gets a slab record: keep flags = written flags
        const int ncin = ncarry;
        const int total = ncin + (CARRY ? build_list_at(L, CL, w, lane, ncin) : build_list(L, w, lane));
        const int nproc = (!CARRY || top - SB < 0) ? total : (total & ~15);  // whole chunks; the tile's last batch pads
        float *slab = s_acc[SHARED ? 0 : w];
        for (int p0 = 0;; p0 += CAP) {  // rounds of at most CAP list positions (one, unless more than CAP survive)
        const int p1 = (BLEND_ABL & 2) ? p0 : CARRY ? imin_(nproc, p0 + CAP) : nproc;
        for (int j0 = p0; j0 < p1; j0 += 16) {
            const int e = L.list[w][j0 + nl];  // ascending e = back to front; slot SB (inert) past the end
            const float4 g0 = L.g0(e), g1 = L.g1(e);
            const float cA = g0.z, cB = g0.w, cC = g1.x, o = g1.y;
            const float uc = g0.x - bx0 - 3.5f, vc = g0.y - by0 - 3.5f;  // centre in block-centred pixel coordinates
            // list position of this survivor in the tile (negative for the inert slot: harmless, alpha = 0)
            const int qn = (CARRY && e > SB) ? __float_as_int(g1.w) : top - e;
            // B operands: this lane's K-slice of the splat's power coefficients (x log2 e) and features
            float bq1, bq2, blx = 0.f, bly = 0.f, bf[NK];
            {
                const PowerCoef pc = power_coeffs(g0.x, g0.y, cA, cB, cC, o, tcx, tcy);
                bq1 = kk == 0 ? pc.q0 : kk == 1 ? pc.qx : kk == 2 ? pc.qy : pc.qxx;
                bq2 = kk == 0 ? pc.qxy : kk == 1 ? pc.qyy : 0.f;
                if (ABS) {
                    // conic (centre - pixel), the factor of |d uv|, is affine in the pixel: one product with the monomials
                    // (1, x, y, .) of phi1 per axis instead of six VALU operations per (pixel, splat)
                    const float ut = g0.x - tcx, vt = g0.y - tcy;   // centre relative to the tile centre (phi1's origin)
                    blx = kk == 0 ? cA * ut + cB * vt : kk == 1 ? -cA : kk == 2 ? -cB : 0.f;
                    bly = kk == 0 ? cB * ut + cC * vt : kk == 1 ? -cB : kk == 2 ? -cC : 0.f;
                }
#pragma unroll
                for (int j = 0; j < NK; ++j)  // this lane's K index of slab j: float 8 + 4 j + kk (zeros past CH: pack_kernel)
                    bf[j] = reinterpret_cast<const float *>(&L.rec[L.part(e, 2 + j)])[kk];
            }
            f32x4 d_mom = {0.f, 0.f, 0.f, 0.f};
            float s_ax = 0.f, s_ay = 0.f;  // |d uv| sums over the lane's own pixels (summed over the rows at the chunk's end)
            f32x4 d_f[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) d_f[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            float dfv[FEAT_VALU ? CH : 1];
#pragma unroll
            for (int c = 0; c < (FEAT_VALU ? CH : 1); ++c) dfv[c] = 0.f;
            asm volatile("" ::: "memory");  // keep the per-pixel LDS reads inside the chunk (registers, not hoisted copies)
#pragma unroll
            for (int G = 0; G < 4; ++G) {  // strip G: four independent sub-steps, their DPP scans interleave
                f32x4 pw = {0.f, 0.f, 0.f, 0.f}, cgv = {0.f, 0.f, 0.f, 0.f};
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bq1, pw, 0, 0, 0);
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi2[G], bq2, pw, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NK; ++j)
                    cgv = __builtin_amdgcn_mfma_f32_16x16x4f32(HOIST ? hcg[HOIST ? G : 0][HOIST ? j : 0] : pixcol[G * GS + kch[j]],
                                                               bf[j], cgv, 0, 0, 0);
                float cg[4], araw[4], a[4], r1a[4], rp[4], Ts4[4], Rs4[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 stv = *reinterpret_cast<const float4 *>(pixrow + G * GS + i * PW + PS);
                    cg[i] = cgv[i];  // (the colour dot product stays on the matrix cores: as per-lane FMAs it was 6 % slower)
                    const int last = __float_as_int(stv.y);
                    Ts4[i] = stv.z;
                    Rs4[i] = stv.w;
                    bool pw_ok;
                    araw[i] = exp2_guard(pw[i], pw_ok);   // o * G: the opacity is part of the polynomial's constant term
                    // (min(0.99, .) is monotone and 0.99 > 1/255: alpha < 1/255 <=> araw < 1/255 -- the forward's decision)
                    ok[i] = (qn < last) && pw_ok && !(araw[i] < (1.0f / 255.0f));
                    araw[i] = ok[i] ? araw[i] : 0.f;   // 0 for a splat this pixel does not replay: alpha = 0, dL/dpower = 0
                    a[i] = fminf(0.99f, araw[i]);
                    r1a[i] = __builtin_amdgcn_rcpf(1.f - a[i]);
                    rp[i] = r1a[i];
                }
                row_scan_mul4(rp[0], rp[1], rp[2], rp[3]);
                float T[4], wgt[4], rs[4], R[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T[i] = Ts4[i] * rp[i];  // transmittance in front of this splat
                    wgt[i] = a[i] * T[i];
                    rs[i] = cg[i] * wgt[i];
                }
                row_scan_add4(rs[0], rs[1], rs[2], rs[3]);
                row_shr1_add4(R, rs, Rs4);  // R = R_state + colour of the deeper splats of this chunk
                // the row's last lane holds the new state of its pixel: store from lanes 15/31/47/63 only, without a
                // branch (the chunk stays one basic block, so the four strips' instruction streams interleave)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    lds_store2_lane15(pixrow + G * GS + i * PW + PS + 2, T[i], Rs4[i] + rs[i]);
                f32x4 lx = {0.f, 0.f, 0.f, 0.f}, ly = {0.f, 0.f, 0.f, 0.f};
                if (ABS) {
                    lx = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], blx, lx, 0, 0, 0);
                    ly = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bly, ly, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = 4 * G + i;
                    const float dLa = T[i] * cg[i] - R[i] * r1a[i];   // (R includes the background term: the state starts from it)
                    const float dLp = araw[i] * dLa;  // dL/dpower (araw = 0 where the splat is not replayed)
                    d_mom = __builtin_amdgcn_mfma_f32_16x16x4f32(HOIST ? hmom[HOIST ? s : 0] : momrow[64 * s], dLp, d_mom, 0, 0, 0);
                    if (FEAT_VALU) {
                        const float4 gq = *reinterpret_cast<const float4 *>(pixrow + G * GS + i * PW);  // g[own pixel][0..3]
                        dfv[0] = __builtin_fmaf(gq.x, wgt[i], dfv[0]);
                        if (CH > 1) dfv[1 % (FEAT_VALU ? CH : 1)] = __builtin_fmaf(gq.y, wgt[i], dfv[1 % (FEAT_VALU ? CH : 1)]);
                        if (CH > 2) dfv[2 % (FEAT_VALU ? CH : 1)] = __builtin_fmaf(gq.z, wgt[i], dfv[2 % (FEAT_VALU ? CH : 1)]);
                        if (CH > 3) dfv[3 % (FEAT_VALU ? CH : 1)] = __builtin_fmaf(gq.w, wgt[i], dfv[3 % (FEAT_VALU ? CH : 1)]);
                    } else {
#pragma unroll
                        for (int q = 0; q < NA; ++q)
                            d_f[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                HOIST ? hft[HOIST ? s : 0][HOIST ? q : 0] : pixrow[G * GS + i * PW + gch[q]], wgt[i], d_f[q], 0, 0, 0);
                    }
                    if (ABS) {
                        s_ax += fabsf(dLp * lx[i]);
                        s_ay += fabsf(dLp * ly[i]);
                    }
                }
            }
            // ---- chunk epilogue: lane (n, kk) holds rows 4kk..4kk+3 of every accumulator for survivor n
            if (FEAT_VALU) {
#pragma unroll
                for (int c = 0; c < (FEAT_VALU ? CH : 1); ++c) dfv[c] = rows_sum(dfv[c], lane);  // over the four pixel groups
            }
            if (ABS) {
                s_ax = rows_sum(s_ax, lane);
                s_ay = rows_sum(s_ay, lane);
            }
            if (j0 + nl < nproc) {
                // shared slab: the entry's row (carried survivors: entries SB + 1 + .. -> rows SB + ..); private: list position
                float *rec = slab + (SHARED ? (e > SB ? e - 1 : e) : (CARRY ? j0 + nl - p0 : e)) * NCS;
                auto put = [](float *p, float v) {
                    if (SHARED) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_f32
                    else *p = v;
                };
                const float D0 = d_mom[0];
                if (kk == 0) {
                    const float Dx = d_mom[1], Dy = d_mom[2], Dxx = d_mom[3];
                    put(rec + 0, cA * Dx + cB * Dy - (cA * uc + cB * vc) * D0);
                    put(rec + 1, cB * Dx + cC * Dy - (cB * uc + cC * vc) * D0);
                    put(rec + 2, -0.5f * (uc * uc * D0 - 2.f * uc * Dx + Dxx));
                    put(rec + 5, o > 0.f ? D0 / o : 0.f);
                    if (ABS) {
                        put(rec + I_ABS, s_ax);
                        put(rec + I_ABS + 1, s_ay);
                    }
                } else if (kk == 1) {
                    const float Dx = d_mom[1], Dy = d_mom[2], Dxy = d_mom[3];
                    put(rec + 3, -(uc * vc * D0 - uc * Dy - vc * Dx + Dxy));
                } else if (kk == 2) {
                    const float Dy = d_mom[1], Dyy = d_mom[2];
                    put(rec + 4, -0.5f * (vc * vc * D0 - 2.f * vc * Dy + Dyy));
                }
                if (FEAT_VALU) {
                    if (kk == 3) {   // (kk 0..2 write the geometry terms above)
#pragma unroll
                        for (int c = 0; c < (FEAT_VALU ? CH : 1); ++c) put(rec + NG + c, dfv[c]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NA; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = 16 * q + 4 * kk + i;
                            if (c < CH) put(rec + NG + c, d_f[q][i]);
                        }
                }
            }
        }
        if (CARRY && p0 == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (nproc > 0)  // the carried survivors were replayed (their rows: the wave's first ncin): their part of last batch's records
                for (int idx = lane; idx < ncin * NC; idx += 64) {
                    const int k = idx / NC, c = idx - k * NC;
                    float *src = SHARED ? slab + (SB + CQ * w + k) * NCS + c : slab + idx;
                    __hip_atomic_fetch_add(pair_buf + (size_t)CL.cslot[w][k] * RST + c, *src, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    if (SHARED) *src = 0.f;   // (only this wave adds to its carry rows)
                }
            const int left = total - nproc;  // 0 .. 15 survivors wait for the next super-batch
            if (left > 0) carry_out<CH, SB>(L, CL, w, lane, nproc > 0 ? nproc : ncin, nproc > 0 ? left : left - ncin,
                                            nproc > 0 ? 0 : ncin, top, slots);
            ncarry = left;
        }
        if (CARRY && !SHARED && lane == 0) CL.more[w] = nproc > p0 + CAP;
        if (!BLEND_SLOT_EARLY && p0 == 0) load_slots();
        if (BLEND_LATE_STAGE && p0 == 0) {
            // the next super-batch's payload (8 registers) is requested behind the chunks, not across them: its latency
            // hides under the barrier, the combine and the other workgroups of the CU
            st.load_payload(A, tid);
            st.load_ids(A, tid, range.x, pos, batch + 2);
        }
        __syncthreads();
        // ---- combine the four slabs: thread (ce, cc) sums component cc of every EPI-th entry and stores it at the
        //      entry's pair slot (the NCP - NC pad floats of a record are never written; pair_reduce ignores them)
        if (ce < EPI && !(BLEND_ABL & 1)) {
            const int lo = top - nb + 1;
            if (SHARED) {
                // one row per entry: read it out, clear it for the next super-batch, store it at the entry's pair slot
#pragma unroll
                for (int r = 0; r < NPASS; ++r) {
                    const int ql = ce + r * EPI;
                    if (ql < nb) {
                        float *row = s_acc[0] + (nb - 1 - ql) * NCS + CW * cc;
                        const float v0 = row[0], v1 = row[CW > 1 ? 1 : 0], v2 = row[CW > 1 ? 2 : 0];
                        row[0] = 0.f;
                        if (CW > 1) { row[1] = 0.f; row[2] = 0.f; }
                        store_part(NPASS <= 3 ? sl[NPASS <= 3 ? r : 0] : slots[lo + ql], v0, v1, v2);
                    }
                }
            } else {
                // private slabs, rows by list position: the entry's row in every wave's slab (the zero row CAP where the wave
                // did not replay it in this round), three floats per thread
#pragma unroll
                for (int r = 0; r < NPASS; ++r) {
                    const int ql = ce + r * EPI;
                    if (ql < nb) {
                        const unsigned int p4 = CL.pos4[nb - 1 - ql];
                        float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            const unsigned int pp = umin_(((p4 >> (8 * ww)) & 0xffu) - (unsigned)p0, (unsigned)CAP);
                            const float *row = s_acc[ww] + pp * NC + CW * cc;
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                if (CW * (TPR - 1) + j < NC || CW * cc + j < NC) v[j] += row[j];   // (floats >= NC: padding)
                        }
                        const int slot = NPASS <= 3 ? sl[NPASS <= 3 ? r : 0] : slots[lo + ql];
                        if (p0 > 0) {  // a further round of the same super-batch: the same thread stored the record before
                            const float *old = pair_buf + (size_t)slot * RST + CW * cc;
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                if (CW * cc + j < NCP) v[j] += old[j];
                        }
                        store_part(slot, v[0], v[1], v[2]);
                    }
                }
            }
        }
        if (SHARED) break;   // (no rounds; the next super-batch's barrier orders its slab adds behind this combine's clears)
        const bool more = CARRY && (CL.more[0] | CL.more[1] | CL.more[2] | CL.more[3]);
        __syncthreads();
        if (!more) break;
        }
    }
    if (A.dbg_T_front) {  // per-pixel transmittance after the last (front-most) replayed splat: lane q <-> pixel q of the block
        const int px = bx + (lane & 7), py = by + (lane >> 3);
        if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_pix[w][pixoff(lane) + PS + 2];
    }
}

// ------------------------------------------------------------------ narrow matrix-core backward on QUARTER lists
// blend_bwd_mfma_kernel replays a block-level survivor over the 64 pixels of the wave's 8x8 block; on BASELINE configs[1] a
// (tile, splat) pair keeps 1.23 of its 4 blocks but only 3.2 of its 16 quarters (4x4 pixels), and 32 % of the evaluated
// (pixel, splat) pairs are active against the forward's 49 %.  Here every 4x4 quarter of the wave's block walks
// its OWN order-preserving list (the quarter bits of the forward's cull flags; pixels of different quarters are independent,
// only the order inside a quarter matters), a step = 16 survivors of one quarter's list x its 16 pixels = the strip step of
// blend_bwd_mfma_kernel with the strip being the quarter.  What makes a step per quarter pay (round 2's experiment with the
// full chunk prologue / epilogue per step did not):
//   * the power polynomial's coefficients are computed ONCE per staged entry, by the lanes that park its record (quad exchange
//     of the two geometry parts), into the record's unused floats: part 3 = q0 qx qy qxx, part 1 .zw = qxy qyy;
//   * a step adds RAW moments of dL/dpower about the wave's block centre (moved to the tile centre in the combine) and the feature sums
//     into the survivor's slab row (rows by position in the wave's block-level list, float4 read-add-write by three lane
//     groups); the map to d uv / d conic / d opacity runs once per entry in the combine, which has the entry's geometry;
//   * no carried survivors: lists of a quarter are short, the super-batch is what fills the steps.
// (Wave w walking quarter w of each of the tile's four blocks instead -- a more even share of the tile: barrier-coupled steps
// 1.087 -> 1.047 of the mean in the CPU model, and what blend_bwd_sets_quarter_kernel does -- doubles the rows a wave needs
// (a splat's quarters inside one block share a row, its quarters in different blocks meet in different waves): second rounds
// everywhere at 64 rows, 138 -> 166 us per frame; 128 rows do not fit four workgroups per CU.)
// Same arithmetic as blend_bwd_mfma_kernel per (pixel, splat): exponent chain, guards, scans.  Narrow rows; ABS: with the |d uv|
// sums of the abs taps (two more products per step for conic (centre - pixel), as in blend_bwd_sets_quarter_kernel).
#ifndef BLEND_Q_SB
#define BLEND_Q_SB 128
#endif
#ifndef BLEND_Q_CAP
#define BLEND_Q_CAP 80   // slab rows per wave and round (a second round costs a list rebuild, two barriers and a combine).  Round 6: with
                         // reach masks a super-batch of 128 entries keeps 59 rows per wave on average (39 before): 64 rows -> 80 (137 -> 132 us per
                         // frame; 45 KB of LDS = three workgroups per CU, which the compiler answers with 155 registers; 112 rows = two: 167 us)
#endif
#ifndef BLEND_Q_SWZ
#define BLEND_Q_SWZ 1   // part p of entry e at 4 e + (p ^ ((e >> 2) & 3)): 16 survivors' reads of one part spread over the banks
#endif
#ifndef BLEND_Q_MINW
#define BLEND_Q_MINW 4
#endif
#ifndef BLEND_QABL
#define BLEND_QABL 0    // timing ablation of the quarter-list backward (1: no steps, 2: no combine); results invalid
#endif
template <int CH, bool ABS = false>
struct QuarterCfg {
    static_assert(CH <= 3 && Rec<CH>::RQ == 4 && Rec<CH>::CULL >= 11, "floats 11-15 of the record (cull parameters) are free for the coefficients");
    static constexpr int SB = BLEND_Q_SB, CAP = BLEND_Q_CAP;
    static constexpr int NG = GradLayout<ABS, false>::NG, NC = NG + CH, NCP = PAIR_STRIDE(NC);
    // slab row.  ABS: [M0 Mx My Mxx | Mxy Myy ax ay | f0 f1 f2 . of lane groups 0 + 2 | of lane groups 1 + 3], two float4 per lane
    // group 0 / 1.  Without the abs taps (round 6): 12 floats, THREE per lane group -- [M0 Mx My | Mxx Mxy Myy | f of groups 0 + 2 | f
    // of groups 1 + 3] (the moment operand's rows are ordered 1 x y . xx xy yy . for it) -- one 12-byte read-add-write per lane and
    // step, and 80 rows per wave fit the 40 KB of FOUR workgroups per CU (16-float rows: 45 KB = three; 130 -> 112 us per frame)
    static constexpr int RW = ABS ? 16 : 12;
    static constexpr int PW = 8;    // pixel row: [g0 g1 g2 . | . ncontrib T_state R_state]
};

template <int CH, bool ABS, bool EXACT>
__global__ void __launch_bounds__(256, BLEND_Q_MINW)
blend_bwd_quarter_kernel(const BlendArgs B) {
    using Cfg = QuarterCfg<CH, ABS>;
    constexpr int SB = Cfg::SB, CAP = Cfg::CAP, NCP = Cfg::NCP, RW = Cfg::RW, PW = Cfg::PW, RQ = 4;
    static_assert(SB <= 128 && CAP < 255, "list entries are (entry | position << 8) in 16 bits");
    __shared__ float4 s_rec[(SB + 1) * RQ];              // staged records, slot SB = inert; part p of entry e at qpart(e, p)
    auto qpart = [](int e, int p) { return BLEND_Q_SWZ ? 4 * e + (p ^ ((e >> 2) & 3)) : 4 * e + p; };
    __shared__ unsigned int s_keep[SB];
    __shared__ unsigned short s_qlist[4][4][CAP + 16];   // [wave][quarter], entries of the current round: entry | (slab row) << 8
    __shared__ unsigned int s_pos4[SB];                  // byte w: position of entry e in wave w's block-level list (255: none)
    __shared__ __attribute__((aligned(16))) float s_acc[4][(CAP + 1) * RW];
    constexpr int KS = 4 * PW + 8, GS = 4 * KS;          // pixel rows skewed per lane group (see blend_bwd_mfma_kernel)
    __shared__ __attribute__((aligned(16))) float s_pix[4][4 * GS];
    auto pixoff = [](int q) { return (q >> 4) * GS + ((q >> 2) & 3) * KS + (q & 3) * PW; };
    __shared__ float s_mom[16 * 32];   // [step][lane group][row & 7]: rows 0-5 of the moment operand (rows 8-15 alias them: their products are not used)
    __shared__ int s_wmax[4];
    __shared__ int s_more[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int RST = B.rec_stride ? B.rec_stride : NCP;
    float *const pair_buf = B.pair_buf + (size_t)frame * (size_t)B.cap * RST + B.rec_off;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    const float tcx = (float)(tx * TILE) + 7.5f, tcy = (float)(ty * TILE) + 7.5f;   // tile centre: origin of the polynomial AND of the moments
    const float ox = (float)((w & 1) * 8) - 7.5f, oy = (float)((w >> 1) * 8) - 7.5f;
    const int cn = EXACT ? CH : A.cn;
    const int nl = lane & 15, kk = lane >> 4;
    // pixel index of the strip walk: q = 16 G + 4 kk + i  <->  quarter G = (sx, sy), pixel (x, y) = (4 sx + i, 4 sy + kk)
    auto qx = [](int q) { return 4 * ((q >> 4) & 1) + (q & 3); };
    auto qy = [](int q) { return 4 * (q >> 5) + ((q >> 2) & 3); };
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // A operand of the moment product (one table for the four waves): rows 0-3 = 1 x y xx, rows
                                    // 4-5 = xy yy of the pixel in BLOCK-centred coordinates; the combine moves a wave's sums to the tile centre
        const int st = 4 * w + r, Gs = st >> 2, is = st & 3;
        const int q = 16 * Gs + 4 * kk + is;
        const float x = (float)qx(q) - 3.5f, y = (float)qy(q) - 3.5f;
        float v = 0.f;
        if (ABS) {
            if (nl < 4) v = nl == 0 ? 1.f : nl == 1 ? x : nl == 2 ? y : x * x;
            else if (nl < 6) v = nl == 4 ? x * y : y * y;
        } else {   // rows 0-2 = 1 x y, rows 4-6 = xx xy yy: three sums per lane group (see QuarterCfg::RW)
            if (nl < 3) v = nl == 0 ? 1.f : nl == 1 ? x : y;
            else if (nl >= 4 && nl < 7) v = nl == 4 ? x * x : nl == 5 ? x * y : y * y;
        }
        if (nl < 8) s_mom[32 * st + 8 * kk + nl] = v;
    }
    float phi1[4], phi2[4];
#pragma unroll
    for (int Gs = 0; Gs < 4; ++Gs) {
        const int q = 16 * Gs + nl;
        const float x = (float)qx(q) + ox, y = (float)qy(q) + oy;
        phi1[Gs] = kk == 0 ? 1.f : kk == 1 ? x : kk == 2 ? y : x * x;
        phi2[Gs] = kk == 0 ? x * y : kk == 1 ? y * y : 0.f;
    }
    const int lx = lane & 7, ly = lane >> 3;                                       // lane <-> pixel (lx, ly) of the block
    const int myq = 16 * ((lx >> 2) + 2 * (ly >> 2)) + 4 * (ly & 3) + (lx & 3);    // its index in the strip walk
    {
        const int px = bx + lx, py = by + ly;
        const size_t HW = (size_t)A.H * A.W;
        const bool inside = (px < A.W) && (py < A.H);
        const size_t pix = (size_t)A.W * (size_t)py + px;
        const float Tf = inside ? A.final_T[pix] : 0.f;
        const int last = inside ? A.ncontrib[pix] : 0;
        float *r = s_pix[w] + pixoff(myq);
        float bgdot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = (k < CH && inside && k < cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
            r[k] = g;
            bgdot += A.bg * g;
        }
        r[4] = 0.f;
        r[5] = __int_as_float(last);
        r[6] = Tf;             // T_state
        r[7] = Tf * bgdot;     // R_state: starts from the background
        const int wmax = wave_max_i(last);
        if (lane == 0) s_wmax[w] = wmax;
    }
    if (tid < RQ) s_rec[qpart(SB, tid)] = make_float4(tid == 3 ? -__builtin_inff() : 0.f, 0.f, 0.f, 0.f);   // inert slot SB: q0 = log2(0)
    if (lane < RW) s_acc[w][CAP * RW + lane] = 0.f;   // the slab's zero row
    __syncthreads();
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    const int *slots = A.slot_sorted + range.x;
    auto store_rec = [&](int slot, const float (&v)[12]) {
        float4 *dst = reinterpret_cast<float4 *>(pair_buf + (size_t)slot * RST);
#pragma unroll
        for (int c = 0; c < NCP / 4; ++c) dst[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    };
    {
        const float z[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ql = n + tid; ql < len; ql += 256) store_rec(slots[ql], z);   // entries nobody replays: zero record
    }
    if (n <= 0) {
        if (A.dbg_T_front) {
            const int px = bx + lx, py = by + ly;
            if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_pix[w][pixoff(myq) + 6];
        }
        return;
    }
    float *pixrow = s_pix[w] + kk * KS;            // own pixel of step (G, i): pixrow + G * GS + i * PW
    const float *pixcol = s_pix[w] + pixoff(nl);   // pixel nl of quarter G: pixcol + G * GS
    const float *momrow = s_mom + 8 * kk + (nl & 7);
    const int kch = kk < CH ? kk : 4;              // K index of the cg product (slot 4 of a pixel row is zero)

    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };
    Stager<CH, SB> st;
    static_assert(Stager<CH, SB>::NCHUNK % 256 == 0, "every thread parks K chunks");
    st.load_ids(A, tid, range.x, pos, 0);
    st.load_payload(A, tid);
    st.load_ids(A, tid, range.x, pos, 1);
    auto load_flags = [&](int topb) -> unsigned {
        const int q = topb - tid;
        return (tid < SB && q >= 0) ? (unsigned)A.cull_flags[range.x + q] : 0u;
    };
    unsigned fl_next = load_flags(n - 1);

    int batch = 0;
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        {   // park the records; the lanes holding part 0 / part 1 of an entry exchange them and leave the polynomial's
            // coefficients in the record: part 3 = q0 qx qy qxx, part 1 = C o qxy qyy
            const int p = tid & 3;
#pragma unroll
            for (int k = 0; k < Stager<CH, SB>::K; ++k) {
                const int e = (tid >> 2) + 64 * k;
                const float4 mine = st.v[k];
                float4 other;
                other.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.x), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                other.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.y), 0xB1, 0xf, 0xf, true));
                other.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.z), 0xB1, 0xf, 0xf, true));
                other.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.w), 0xB1, 0xf, 0xf, true));
                const float4 g0 = p == 0 ? mine : other, g1 = p == 0 ? other : mine;
                const PowerCoef pc = power_coeffs(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, tcx, tcy);
                if (p == 0) {
                    s_rec[qpart(e, 0)] = mine;
                    s_rec[qpart(e, 3)] = make_float4(pc.q0, pc.qx, pc.qy, pc.qxx);
                } else if (p == 1) {
                    s_rec[qpart(e, 1)] = make_float4(mine.x, mine.y, pc.qxy, pc.qyy);
                } else if (p == 2) {
                    s_rec[qpart(e, 2)] = make_float4(mine.x, mine.y, mine.z, 0.f);   // (.w: a cull parameter, possibly inf, would meet a zero of the cg product)
                }
            }
        }
        const unsigned fl = fl_next;
        fl_next = load_flags(top - SB);
        if (tid < SB) {   // keep word of entry tid: byte w = the quarter bits of the forward's cull, if wave w still needs the entry
            unsigned kw = 0u;
            if (tid < nb) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww)
                    if (top - tid < s_wmax[ww]) kw |= fl & (0xfu << (8 * ww));
            }
            s_keep[tid] = kw;
        }
        __syncthreads();
        float *slab = s_acc[w];
        int slot_mine = 0;
        for (int p0 = 0;; p0 += CAP) {   // rounds of CAP positions of the wave's block-level list (one, unless more survive)
            // ---- this wave's lists: block-level positions (slab rows, the combine's lookup) and one list per quarter
            // (the four quarter counts ride in the bytes of one word: one wave-wide prefix sum serves the four lists)
            unsigned cqw = 0u;   // packed list lengths, byte q = quarter q (at most CAP entries of a round in a list)
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < SB / WAVE; ++r) {
                const int e = r * WAVE + lane;
                unsigned bits = (s_keep[e] >> (8 * w)) & 0xfu;
                const bool kb = bits != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(kb);
                const int ps = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (p0 == 0) reinterpret_cast<unsigned char *>(s_pos4)[4 * e + w] = kb ? (unsigned char)imin_(ps, 254) : (unsigned char)255;
                bits = ((unsigned)(ps - p0) < (unsigned)CAP) ? bits : 0u;       // entries of this round
                const unsigned word = (bits * 0x00204081u) & 0x01010101u;      // bit q -> byte q
                unsigned incl = word;
                asm volatile("s_nop 1\n\t"
                             "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\t"
                             "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\t"
                             "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\t"
                             "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\t"
                             "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                             "s_nop 1\n\t"
                             "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                             : "+v"(incl));
                const unsigned posw = cqw + incl - word;                       // packed list positions of this lane's entry
                const unsigned short ent = (unsigned short)(e | ((ps - p0) << 8));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((bits >> q) & 1u) s_qlist[w][q][(posw >> (8 * q)) & 0xffu] = ent;
                cqw += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                cnt += __popcll(m);
            }
            int cq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cq[q] = (int)((cqw >> (8 * q)) & 0xffu);
            if (lane < 16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s_qlist[w][q][cq[q] + lane] = (unsigned short)(SB | (CAP << 8));   // pad: inert entry, zero row
            }
            {   // rows of this round start from zero
                const int rows = imin_(imax_(cnt - p0, 0), CAP);
                float4 *z = reinterpret_cast<float4 *>(slab);
                for (int c = lane; c < rows * (RW / 4); c += WAVE) z[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (lane == 0) s_more[w] = cnt > p0 + CAP;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                for (int j0 = 0; j0 < ((BLEND_QABL & 1) ? 0 : cq[G]); j0 += 16) {
                    const unsigned le = s_qlist[w][G][j0 + nl];
                    const int e = le & 0xffu, row = le >> 8;
                    const int qn = top - e;
                    const float *er = reinterpret_cast<const float *>(s_rec) + e * (4 * RQ);
                    const int sw = BLEND_Q_SWZ ? 4 * ((e >> 2) & 3) : 0;   // float offset of part p: 4 p ^ sw
                    const float bq1 = er[(12 ^ sw) + kk];           // q0 qx qy qxx
                    const float bq2 = er[(4 ^ sw) + 2 + (kk & 1)];  // qxy qyy (lane groups 2, 3: their monomial operand is zero)
                    const float bf = er[(8 ^ sw) + kk];             // feature kk (zero past CH)
                    float blx = 0.f, bly = 0.f;
                    if (ABS) {   // -conic (centre - pixel) as a product with the monomials (1, x, y, .): operands -c0x cA cB 0 / -c0y cB cC 0
                        const float4 ge = *reinterpret_cast<const float4 *>(er + (0 ^ sw));   // u v cA cB
                        const float cCe = er[(4 ^ sw)];
                        const float ut = ge.x - tcx, vt = ge.y - tcy;
                        blx = kk == 0 ? -(ge.z * ut + ge.w * vt) : kk == 1 ? ge.z : kk == 2 ? ge.w : 0.f;
                        bly = kk == 0 ? -(ge.w * ut + cCe * vt) : kk == 1 ? ge.w : kk == 2 ? cCe : 0.f;
                    }
                    float s_ax = 0.f, s_ay = 0.f;
                    f32x4 d_mom = {0.f, 0.f, 0.f, 0.f};
                    float dfv[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) dfv[c] = 0.f;
                    asm volatile("" ::: "memory");
                    f32x4 pw = {0.f, 0.f, 0.f, 0.f}, cgv = {0.f, 0.f, 0.f, 0.f};
                    pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bq1, pw, 0, 0, 0);
                    pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi2[G], bq2, pw, 0, 0, 0);
                    cgv = __builtin_amdgcn_mfma_f32_16x16x4f32(pixcol[G * GS + kch], bf, cgv, 0, 0, 0);
                    float cg[4], araw[4], a[4], r1a[4], rp[4], Ts4[4], Rs4[4];
                    float4 gq[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        gq[i] = *reinterpret_cast<const float4 *>(pixrow + G * GS + i * PW);
                        const float4 stv = *reinterpret_cast<const float4 *>(pixrow + G * GS + i * PW + 4);
                        cg[i] = cgv[i];
                        const int last = __float_as_int(stv.y);
                        Ts4[i] = stv.z;
                        Rs4[i] = stv.w;
                        bool pw_ok;
                        araw[i] = exp2_guard(pw[i], pw_ok);
                        const bool ok = (qn < last) && pw_ok && !(araw[i] < (1.0f / 255.0f));
                        araw[i] = ok ? araw[i] : 0.f;
                        a[i] = fminf(0.99f, araw[i]);
                        r1a[i] = __builtin_amdgcn_rcpf(1.f - a[i]);
                        rp[i] = r1a[i];
                    }
                    row_scan_mul4(rp[0], rp[1], rp[2], rp[3]);
                    float T[4], wgt[4], rs[4], R[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        T[i] = Ts4[i] * rp[i];
                        wgt[i] = a[i] * T[i];
                        rs[i] = cg[i] * wgt[i];
                    }
                    row_scan_add4(rs[0], rs[1], rs[2], rs[3]);
                    row_shr1_add4(R, rs, Rs4);
                    lds_store2x4_lane15<6, PW + 6, 2 * PW + 6, 3 * PW + 6>(pixrow + G * GS, T[0], Rs4[0] + rs[0], T[1], Rs4[1] + rs[1], T[2],
                                                                           Rs4[2] + rs[2], T[3], Rs4[3] + rs[3]);
                    f32x4 lx4 = {0.f, 0.f, 0.f, 0.f}, ly4 = {0.f, 0.f, 0.f, 0.f};
                    if (ABS) {
                        lx4 = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], blx, lx4, 0, 0, 0);
                        ly4 = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bly, ly4, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float dLa = T[i] * cg[i] - R[i] * r1a[i];
                        const float dLp = araw[i] * dLa;
                        d_mom = __builtin_amdgcn_mfma_f32_16x16x4f32(momrow[32 * (4 * G + i)], dLp, d_mom, 0, 0, 0);
                        if (ABS) {
                            s_ax += fabsf(dLp * lx4[i]);
                            s_ay += fabsf(dLp * ly4[i]);
                        }
                        dfv[0] = __builtin_fmaf(gq[i].x, wgt[i], dfv[0]);
                        if (CH > 1) dfv[1 % CH] = __builtin_fmaf(gq[i].y, wgt[i], dfv[1 % CH]);
                        if (CH > 2) dfv[2 % CH] = __builtin_fmaf(gq[i].z, wgt[i], dfv[2 % CH]);
                    }
                    // ---- step epilogue: raw sums into the survivor's row (this wave's quarters run one after the other: plain
                    //      read-add-write; lane group kk owns floats 4 kk .. 4 kk + 3 of the row)
                    if (ABS) {
                        s_ax = rows_sum(s_ax, lane);
                        s_ay = rows_sum(s_ay, lane);
                    }
                    float fs[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < CH; ++c) {   // lane groups kk and kk ^ 2 together (one swap); groups 0 and 1 keep a half each
                        const u32x2_b r = __builtin_amdgcn_permlane32_swap(__float_as_uint(dfv[c]), __float_as_uint(dfv[c]), false, false);
                        fs[c] = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // own + the value of lane ^ 32, in every lane
                    }
                    if (!ABS) {
                        if (j0 + nl < cq[G]) {   // lane groups 0 / 1: their three moments; 2 / 3: the pooled feature sums (own + lane ^ 32)
                            F3 *r3 = reinterpret_cast<F3 *>(slab + row * RW + 3 * kk);
                            F3 v3 = *r3;
                            v3.x += kk < 2 ? d_mom[0] : fs[0];
                            v3.y += kk < 2 ? d_mom[1] : fs[1];
                            v3.z += kk < 2 ? d_mom[2] : fs[2];
                            *r3 = v3;
                        }
                    } else if (kk < 2 && j0 + nl < cq[G]) {   // lane group kk: floats 4 kk .. of the moments, 8 + 4 kk .. of the features
                        float4 *rm = reinterpret_cast<float4 *>(slab + row * RW + 4 * kk), *rf = rm + 2;
                        float4 m4 = *rm, f4 = *rf;
                        m4.x += d_mom[0]; m4.y += d_mom[1];
                        m4.z += ABS ? d_mom[2] + (kk == 1 ? s_ax : 0.f) : d_mom[2];   // (lane group 1: rows 6, 7 of the moment product are zero)
                        m4.w += ABS ? d_mom[3] + (kk == 1 ? s_ay : 0.f) : d_mom[3];
                        f4.x += fs[0]; f4.y += fs[1]; f4.z += fs[2];
                        *rm = m4;
                        *rf = f4;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next step may add to the same rows
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (p0 == 0) {
                slot_mine = tid < nb ? slots[top - tid] : 0;   // this thread's entry in the combine: entry tid
                st.load_payload(A, tid);
                st.load_ids(A, tid, range.x, pos, batch + 2);
            }
            __syncthreads();
            // ---- combine: thread e sums entry e's rows of the four slabs and maps the raw moments to the record
            if (tid < nb && !(BLEND_QABL & 2)) {
                const int e = tid;
                const unsigned int p4 = s_pos4[e];
                float s[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned int pp = umin_(((p4 >> (8 * ww)) & 0xffu) - (unsigned)p0, (unsigned)CAP);
                    const float4 *rw = reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW);
                    float4 m, m2, fa, fb;
                    if (ABS) {
                        m = rw[0]; m2 = rw[1]; fa = rw[2]; fb = rw[3];
                    } else {   // [M0 Mx My Mxx | Mxy Myy fA0 fA1 | fA2 fB0 fB1 fB2]
                        const float4 r1 = rw[1], r2 = rw[2];
                        m = rw[0];
                        m2 = make_float4(r1.x, r1.y, 0.f, 0.f);
                        fa = make_float4(r1.z, r1.w, r2.x, 0.f);
                        fb = make_float4(r2.y, r2.z, r2.w, 0.f);
                    }
                    // wave ww's moments are about its block centre (bxw, byw) = (-4 | 4, -4 | 4) from the tile centre:
                    // X = x + bxw, Y = y + byw
                    const float bxw = (ww & 1) ? 4.f : -4.f, byw = (ww >> 1) ? 4.f : -4.f;
                    s[0] += m.x;
                    s[1] += m.y + bxw * m.x;
                    s[2] += m.z + byw * m.x;
                    s[3] += m.w + 2.f * bxw * m.y + (bxw * bxw) * m.x;
                    s[4] += m2.x + bxw * m.z + byw * m.y + (bxw * byw) * m.x;
                    s[5] += m2.y + 2.f * byw * m.z + (byw * byw) * m.x;
                    s[6] += fa.x + fb.x; s[7] += fa.y + fb.y; s[8] += fa.z + fb.z;
                    s[9] += m2.z; s[10] += m2.w;
                }
                const float4 g0 = s_rec[qpart(e, 0)], g1 = s_rec[qpart(e, 1)];
                const float cA = g0.z, cB = g0.w, cC = g1.x, o = g1.y;
                const float uc = g0.x - tcx, vc = g0.y - tcy;
                const float M0 = s[0], Mx = s[1], My = s[2], Mxx = s[3], Mxy = s[4], Myy = s[5];
                float rec[12];
                rec[0] = cA * Mx + cB * My - (cA * uc + cB * vc) * M0;
                rec[1] = cB * Mx + cC * My - (cB * uc + cC * vc) * M0;
                rec[2] = -0.5f * (uc * uc * M0 - 2.f * uc * Mx + Mxx);
                rec[3] = -(uc * vc * M0 - uc * My - vc * Mx + Mxy);
                rec[4] = -0.5f * (vc * vc * M0 - 2.f * vc * My + Myy);
                rec[5] = o > 0.f ? M0 / o : 0.f;
                rec[6] = 0.f; rec[7] = 0.f; rec[8] = 0.f; rec[9] = 0.f; rec[10] = 0.f; rec[11] = 0.f;
                if (ABS) { rec[6] = s[9]; rec[7] = s[10]; }
                rec[Cfg::NG] = s[6];
                if (CH > 1) rec[Cfg::NG + 1 % CH] = s[7];
                if (CH > 2) rec[Cfg::NG + 2 % CH] = s[8];
                if (p0 > 0) {   // a further round of the same super-batch: this thread stored the record before
                    const float *old = pair_buf + (size_t)slot_mine * RST;
#pragma unroll
                    for (int c = 0; c < Cfg::NC; ++c) rec[c] += old[c];
                }
                store_rec(slot_mine, rec);
            }
            const bool more = s_more[0] | s_more[1] | s_more[2] | s_more[3];
            __syncthreads();
            if (!more) break;
        }
    }
    if (A.dbg_T_front) {
        const int px = bx + lx, py = by + ly;
        if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_pix[w][pixoff(myq) + 6];
    }
}

// ------------------------------------------------------------------ backward of several feature SETS in ONE pass
// The reference renderer composites up to three feature sets over one geometry (reference:
// src/pointrix/renderer/dptr_ortho_enhanced.py:331-375): the TAP set (alpha_blending_enhanced: every gradient + the ndc /
// abs_ndc taps), a SECOND set blended with the live opacity (the depth) and a DETACHED set (alpha_blending with
// opacity.detach(): the extra attributes).  The three share the alpha / transmittance chain and differ only in where their
// dL/dalpha goes:
//       d uv, d conic <- every set        d opacity <- tap + second set        taps, |taps| <- tap set
// One replay of the chain serves all of them: a colour dot product, a behind-colour scan and a dL/dalpha per set, the moment
// product on their sum, the opacity and tap sums as per-lane sums over the lane's own pixels (+ one cross-row sum per chunk).
// Channel slots: [0,4) tap set | [4,8) second set | [8,28) detached set, zero padded (whole K-slabs of the cg product).
// dL_dout of the wave's pixels lives in REGISTERS in both MFMA operand layouts (staged once through the slab memory); LDS
// keeps the 8-float replay state of a pixel.  Pair record:
//       [ux uy ca cb | cc o ax ay | tx ty 0 0 | dL_dfeature of the row's channels 0 .. C-1]      (NG = SETS_NG = 12, NC = 12 + C)
struct SetsCfg {
    static constexpr int CH = 28, NK = 7, NA = 2, SB = 64;
    static constexpr int NG = SETS_NG, NCMAX = NG + CH;
    static constexpr int PS = 8;  // state floats per pixel: [. . . ncontrib | T, R of set 0 1 2 (starting from T_final bg.g)]
};

// Float of feature slot s inside the packed SETS record (and its staged copy): the 28 slots are stored TRANSPOSED -- slot
// 4 j + kk at 8 + 8 kk + j -- so that the B operand rows a lane group kk feeds to the colour products (slots kk, 4 + kk, ..,
// 24 + kk) are 7 consecutive floats: two ds_read_b128 per step instead of seven ds_read_b32 whose 16 lanes hit 8 bank pairs
// (record stride 48 floats = 16 mod 32).  Floats 8 kk + 15 are padding (zero).
__device__ __forceinline__ constexpr int sets_fpos(int slot) { return 8 + 8 * (slot & 3) + (slot >> 2); }

// slot -> row channel (or -1): uniform selects on the kernel arguments
__device__ __forceinline__ int sets_slot_channel(const BlendArgs &A, int slot) {
    const int g = slot < 4 ? 0 : slot < 8 ? 1 : 2;
    const int o = slot - (g == 0 ? 0 : g == 1 ? 4 : 8);
    const int c0 = g == 0 ? A.s0c0 : g == 1 ? A.s1c0 : A.s2c0, cn = g == 0 ? A.s0cn : g == 1 ? A.s1cn : A.s2cn;
    return (slot < SetsCfg::CH && o < cn) ? c0 + o : -1;
}

__global__ void __launch_bounds__(256)
pack_sets_kernel(const BlendArgs B) {
    const BlendArgs A = frame_args(B, blockIdx.y);
    constexpr int CH = SetsCfg::CH, RS = Rec<CH>::RS, CO = Rec<CH>::CULL;
    static_assert(CO >= 0, "the cull parameters ride in the record's padding");
    constexpr int RQ = Rec<CH>::RQ, LS = RS + 4;  // LDS row stride: 16-B aligned, conflict-free for the float4 row writes
    __shared__ __attribute__((aligned(16))) float s_rec[256 * LS];
    const int i0 = blockIdx.x * 256;
    const int i = i0 + threadIdx.x;
    float r[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) r[k] = 0.f;
    if (i < A.P) {
        const float2 q = A.uv[i];
        r[0] = q.x; r[1] = q.y;
        r[2] = A.conic[3 * i]; r[3] = A.conic[3 * i + 1]; r[4] = A.conic[3 * i + 2];
        r[5] = A.opacity[i];
        r[7] = __int_as_float(i);
        if (A.feature) {
            const float *f = A.feature + (size_t)i * (A.feature_row ? A.feature_row : A.C);
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int c = sets_slot_channel(A, k);
                if (c >= 0) r[sets_fpos(k)] = f[c];
            }
        }   // (else: the sets' own tensors, staged cooperatively below)
        const CullP cp = cull_params(r[2], r[3], r[4], r[5]);
        r[CO] = cp.hx; r[CO + 1] = cp.hy; r[CO + 2] = cp.tauq; r[CO + 3] = cp.ia; r[CO + 4] = cp.ic;
    }
    // rows through LDS: consecutive lanes store consecutive 16-byte chunks of the record array (a thread writing its own
    // 192-byte record was 32 us per frame at 300k Gaussians, 1.8 TB/s)
    float4 *row = reinterpret_cast<float4 *>(s_rec + threadIdx.x * LS);
#pragma unroll
    for (int k = 0; k < RS; k += 4) row[k / 4] = make_float4(r[k], r[k + 1], r[k + 2], r[k + 3]);
    __syncthreads();
    const int nrec = imin_(256, A.P - i0);
    if (!A.feature) {  // slots [0,4) | [4,8) | [8,28) of the three sets
        stage_feature_rows<true>(s_rec, LS, 0, A.sf0, A.s0cn, i0, nrec);
        stage_feature_rows<true>(s_rec, LS, 4, A.sf1, A.s1cn, i0, nrec);
        stage_feature_rows<true>(s_rec, LS, 8, A.sf2, A.s2cn, i0, nrec);
        __syncthreads();
    }
    float4 *dst = reinterpret_cast<float4 *>(A.pack + (size_t)i0 * RS);
    for (int c = threadIdx.x; c < nrec * RQ; c += 256) {
        const int g = c / RQ, part = c - g * RQ;
        dst[c] = *reinterpret_cast<const float4 *>(s_rec + g * LS + 4 * part);
    }
}

template <bool ABS>
__global__ void __launch_bounds__(256, BLEND_SETS_MINW)
blend_bwd_sets_kernel(const BlendArgs B) {
    using Cfg = SetsCfg;
    constexpr int CH = Cfg::CH, SB = Cfg::SB, NG = Cfg::NG, NK = Cfg::NK, NA = Cfg::NA, PS = Cfg::PS;
    // private slab per wave, rows by the wave's list POSITION (CAP rows + a zero row; a longer survivor list takes another round
    // of chunks + combine): 25 KB instead of the 39 KB of one row per staged entry -- three workgroups per CU.  (First: staging
    // of the wave's dL_dout, half a block at a time.)
    constexpr int CAP = BLEND_SETS_CAP;
    static_assert(CAP % 16 == 0 && (CAP + 1) * Cfg::NCMAX >= 32 * CH, "whole chunks; the staging of 32 pixels fits a slab");
    __shared__ TileLDS<CH, SB, false, 0, BLEND_REC_SWZ != 0> L;
    __shared__ CarryLDS<SB, true> CL;              // (position bytes of the entries; nothing is carried here)
    __shared__ float s_acc[4][(CAP + 1) * Cfg::NCMAX];
    // replay state rows (8 floats per pixel), pixel q = 16 G + 4 kk + i at float G * GS + kk * KS + i * PS: 8 floats of skew
    // per lane group, or the four groups' rows of one step share their banks (see blend_bwd_mfma_kernel)
    constexpr int KS = 4 * PS + (BLEND_STATE_SKEW ? 8 : 0), GS = 4 * KS;
    __shared__ __attribute__((aligned(16))) float s_state[4][4 * GS];
    auto pixoff = [](int q) { return (q >> 4) * GS + ((q >> 2) & 3) * KS + (q & 3) * PS; };
    __shared__ float s_mom[16 * 64];             // A operand of the moment product: [step 4 G + i][lane]
    __shared__ int s_wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int NC = NG + A.C, NCP = PAIR_STRIDE(NC);
    float *const pair_buf = B.pair_buf + (size_t)frame * (size_t)B.cap * NCP;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    const float bx0 = (float)bx, by0 = (float)by;
    const float tcx = (float)(tx * TILE) + 7.5f, tcy = (float)(ty * TILE) + 7.5f;
    const float ox = (float)((w & 1) * 8) - 7.5f, oy = (float)((w >> 1) * 8) - 7.5f;
    const int nl = lane & 15, kk = lane >> 4;
    int wmax;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // moment operand table (see blend_bwd_mfma_kernel)
        const int st = 4 * w + r, Gs = st >> 2, is = st & 3;
        const int q = 16 * Gs + 4 * kk + is;
        const float x = (float)(q & 7) - 3.5f, y = (float)(q >> 3) - 3.5f;
        const int grp = nl >> 2, i = nl & 3;
        float v = 0.f;
        if (grp == 0) v = i == 0 ? 1.f : i == 1 ? x : i == 2 ? y : x * x;
        if (grp == 1) v = i == 0 ? 1.f : i == 1 ? x : i == 2 ? y : x * y;
        if (grp == 2) v = i == 0 ? 1.f : i == 1 ? y : i == 2 ? y * y : 0.f;
        s_mom[64 * st + lane] = v;
    }
    float phi1[4], phi2[4];
#pragma unroll
    for (int Gs = 0; Gs < 4; ++Gs) {
        const int q = 16 * Gs + nl;
        const float x = (float)(q & 7) + ox, y = (float)(q >> 3) + oy;
        phi1[Gs] = kk == 0 ? 1.f : kk == 1 ? x : kk == 2 ? y : x * x;
        phi2[Gs] = kk == 0 ? x * y : kk == 1 ? y * y : 0.f;
    }
    float *stage = s_acc[w];  // [pixel of the half block][slot]
    float gpix[CH];           // dL_dout of this lane's pixel, slot by slot
    {   // per-pixel constants: lane q <-> pixel (q & 7, q >> 3) of the wave's block
        const int px = bx + (lane & 7), py = by + (lane >> 3);
        const size_t HW = (size_t)A.H * A.W;
        const bool inside = (px < A.W) && (py < A.H);
        const size_t pix = (size_t)A.W * (size_t)py + px;
        const float Tf = inside ? A.final_T[pix] : 0.f;
        const int last = inside ? A.ncontrib[pix] : 0;
        float bgd[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = sets_slot_channel(A, k);
            const int gi = k < 4 ? 0 : k < 8 ? 1 : 2;
            float g = 0.f;
            if (inside && c >= 0) {
                if (A.dL_dout) {
                    g = A.dL_dout[(size_t)c * HW + pix];
                } else {  // the set's own gradient image [cn, H, W]
                    const float *d = gi == 0 ? A.sdl0 : gi == 1 ? A.sdl1 : A.sdl2;
                    const int o = k - (gi == 0 ? 0 : gi == 1 ? 4 : 8);
                    g = d[(size_t)o * HW + pix];
                }
            }
            gpix[k] = g;
            bgd[gi] += (gi == 0 ? A.s0bg : gi == 1 ? A.s1bg : A.s2bg) * g;
        }
        float *r = s_state[w] + pixoff(lane);
        r[0] = 0.f; r[1] = 0.f; r[2] = 0.f;
        r[3] = __int_as_float(last);
        r[4] = Tf;    // T_state: transmittance behind the splats replayed so far
        r[5] = Tf * bgd[0]; r[6] = Tf * bgd[1]; r[7] = Tf * bgd[2];  // R_state of the three sets: starts from the background (the deepest layer)
        wmax = wave_max_i(last);
        if (lane == 0) s_wmax[w] = wmax;
    }
    if (tid < Rec<CH>::RQ) L.rec[SB * Rec<CH>::RQ + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // inert slot SB
    // ---- the two MFMA operand layouts of dL_dout into registers: the wave's own slab memory stages 32 pixels at a time
    //      (strips 0-1, then 2-3); wave-private, so wave barriers order it
    float hcg[4][NK], hft[16][NA];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int k = 0; k < CH; ++k) stage[(lane & 31) * CH + k] = gpix[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int G = 2 * h; G < 2 * h + 2; ++G)
#pragma unroll
            for (int j = 0; j < NK; ++j) hcg[G][j] = stage[(16 * (G & 1) + nl) * CH + 4 * j + kk];  // A[m = pixel nl of strip G][k = slot]
#pragma unroll
        for (int st = 8 * h; st < 8 * h + 8; ++st) {
#pragma unroll
            for (int q = 0; q < NA; ++q)  // A[m = slot 16 q + nl][k = own pixel of step st]
                hft[st][q] = 16 * q + nl < CH ? stage[(16 * ((st >> 2) & 1) + 4 * kk + (st & 3)) * CH + 16 * q + nl] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < NC) s_acc[w][CAP * NC + lane] = 0.f;   // the slab's zero row (what the combine reads where the wave has nothing)
    __syncthreads();
    const float *momrow = s_mom + lane;  // + 64 * (4 G + i)
    // the feature accumulators of this lane hold slots 16 q + 4 kk + i (rows of the product): one group per (q, kk) --
    // record component of the first of them and how many are real channels
    int fbase[NA], fcnt[NA];
    {
        const int c0[3] = {__builtin_amdgcn_readfirstlane(A.s0c0), __builtin_amdgcn_readfirstlane(A.s1c0),
                           __builtin_amdgcn_readfirstlane(A.s2c0)};
        const int cn[3] = {__builtin_amdgcn_readfirstlane(A.s0cn), __builtin_amdgcn_readfirstlane(A.s1cn),
                           __builtin_amdgcn_readfirstlane(A.s2cn)};
        const int o2 = 4 * (kk - 2);  // q = 0, kk >= 2: slots 8 + o2 ..
        fbase[0] = NG + (kk == 0 ? c0[0] : kk == 1 ? c0[1] : c0[2] + o2);
        fcnt[0] = kk == 0 ? cn[0] : kk == 1 ? cn[1] : cn[2] - o2;
        fbase[1] = NG + c0[2] + 8 + 4 * kk;  // q = 1: slots 16 + 4 kk .. = offset 8 + 4 kk of the detached set
        fcnt[1] = cn[2] - (8 + 4 * kk);
    }
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    const int *slots = A.slot_sorted + range.x;
    // combine: TPR threads per record, three floats each (one 12-byte store)
    const int TPR = (NC + 2) / 3, EPI = 256 / TPR;   // pair records the 256 threads write per pass
    const int ce = tid / TPR, cc = tid - ce * TPR;   // this thread's (entry within the pass, part of the record)
    auto store_part = [&](int slot, float v0, float v1, float v2) {
        float *dst = pair_buf + (size_t)slot * NCP + 3 * cc;
        if (3 * cc + 2 < NCP) {
            F3 t; t.x = v0; t.y = v1; t.z = v2;
            *reinterpret_cast<F3 *>(dst) = t;        // (floats NC .. NCP - 1 of a record are padding)
        } else {
            if (3 * cc < NCP) dst[0] = v0;
            if (3 * cc + 1 < NCP) dst[1] = v1;
        }
    };
    if (ce < EPI)
        for (int ql = n + ce; ql < len; ql += EPI)   // entries nobody replays: zero record
            store_part(slots[ql], 0.f, 0.f, 0.f);
    float *state = s_state[w] + kk * KS;             // own pixel of step (G, i): state + G * GS + i * PS
    if (n <= 0) {
        if (A.dbg_T_front) {
            const int px = bx + (lane & 7), py = by + (lane >> 3);
            if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_state[w][pixoff(lane) + 4];
        }
        return;
    }

    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };
    Stager<CH, SB> st;
    st.load_ids(A, tid, range.x, pos, 0);
    st.load_payload(A, tid);
    st.load_ids(A, tid, range.x, pos, 1);
    // the forward's cull flags of the super-batch (thread e: entry e), one super-batch ahead
    auto load_flags = [&](int topb) -> unsigned {
        const int q = topb - tid;
        return (A.cull_flags && tid < SB && q >= 0) ? (unsigned)A.cull_flags[range.x + q] : 0u;
    };
    unsigned fl_next = load_flags(n - 1);

    int batch = 0;
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        st.park(L, tid);
        const unsigned fl = fl_next;
        fl_next = load_flags(top - SB);
        if (A.cull_flags) {   // (the keep words need nothing of the staged records: one barrier serves both)
            keep_from_flags(L, tid, nb, fl, [&](int e, int ww) { return top - e < s_wmax[ww]; });
        } else {
            __syncthreads();
            tile_cull<CH, SB, false, false, false>(L, tid, nb, (float)(tx * TILE), (float)(ty * TILE),
                                               [&](int e, int ww) { return top - e < s_wmax[ww]; });
        }
        __syncthreads();
        const int cnt = build_list_at(L, CL, w, lane, 0);   // + the entries' list positions (CL.pos4)
        float *slab = s_acc[w];
        int sl[3];   // pair slots of the entries this thread writes in the combine (up to three passes)
        for (int p0 = 0;; p0 += CAP) {   // rounds of at most CAP list positions (one, unless more than CAP survive)
        const int p1 = imin_(cnt, p0 + CAP);
        for (int j0 = p0; j0 < p1; j0 += 16) {
            const int e = L.list[w][j0 + nl];
            const float4 g0 = L.g0(e), g1 = L.g1(e);
            const float cA = g0.z, cB = g0.w, cC = g1.x, o = g1.y;
            const float uc = g0.x - bx0 - 3.5f, vc = g0.y - by0 - 3.5f;
            const int qn = top - e;
            float bq1, bq2, blx, bly, bf[NK];
            {
                const PowerCoef pc = power_coeffs(g0.x, g0.y, cA, cB, cC, o, tcx, tcy);
                bq1 = kk == 0 ? pc.q0 : kk == 1 ? pc.qx : kk == 2 ? pc.qy : pc.qxx;
                bq2 = kk == 0 ? pc.qxy : kk == 1 ? pc.qyy : 0.f;
                // conic (centre - pixel), the tap gradient's factor, is affine in the pixel: one product with the monomials
                // (1, x, y, .) of phi1 per axis instead of six VALU operations per (pixel, splat)
                const float ut = g0.x - tcx, vt = g0.y - tcy;   // centre relative to the tile centre (phi1's origin)
                blx = kk == 0 ? cA * ut + cB * vt : kk == 1 ? -cA : kk == 2 ? -cB : 0.f;
                bly = kk == 0 ? cB * ut + cC * vt : kk == 1 ? -cB : kk == 2 ? -cC : 0.f;
#pragma unroll
                for (int j = 0; j < NK; ++j) bf[j] = L.f(e, 8 + 8 * kk + j);   // slot 4 j + kk (transposed slots: sets_fpos)
            }
            f32x4 d_mom = {0.f, 0.f, 0.f, 0.f};
            f32x4 d_f[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) d_f[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            float s_op = 0.f, s_tx = 0.f, s_ty = 0.f, s_ax = 0.f, s_ay = 0.f;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                f32x4 pw = {0.f, 0.f, 0.f, 0.f}, cv0 = {0.f, 0.f, 0.f, 0.f}, cv1 = {0.f, 0.f, 0.f, 0.f}, cv2 = {0.f, 0.f, 0.f, 0.f};
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bq1, pw, 0, 0, 0);
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi2[G], bq2, pw, 0, 0, 0);
                cv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][0], bf[0], cv0, 0, 0, 0);
                cv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][1], bf[1], cv1, 0, 0, 0);
#pragma unroll
                for (int j = 2; j < NK; ++j) cv2 = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][j], bf[j], cv2, 0, 0, 0);
                float araw[4], a[4], r1a[4], rp[4], Ts4[4], Rs[3][4], cg[3][4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int last = __float_as_int(state[G * GS + i * PS + 3]);
                    const float4 sb = *reinterpret_cast<const float4 *>(state + G * GS + i * PS + 4);
                    Ts4[i] = sb.x;
                    Rs[0][i] = sb.y; Rs[1][i] = sb.z; Rs[2][i] = sb.w;
                    cg[0][i] = cv0[i]; cg[1][i] = cv1[i]; cg[2][i] = cv2[i];
                    bool pw_ok;
                    araw[i] = exp2_guard(pw[i], pw_ok);   // o * G: the opacity is part of the polynomial's constant term
                    const float alpha = fminf(0.99f, araw[i]);
                    ok[i] = (qn < last) && pw_ok && !(alpha < (1.0f / 255.0f));
                    a[i] = ok[i] ? alpha : 0.f;
                    r1a[i] = __builtin_amdgcn_rcpf(1.f - a[i]);
                    rp[i] = r1a[i];
                }
                row_scan_mul4(rp[0], rp[1], rp[2], rp[3]);
                float T[4], wgt[4], rs[3][4], R[3][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T[i] = Ts4[i] * rp[i];
                    wgt[i] = a[i] * T[i];
#pragma unroll
                    for (int g = 0; g < 3; ++g) rs[g][i] = cg[g][i] * wgt[i];
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    row_scan_add4(rs[g][0], rs[g][1], rs[g][2], rs[g][3]);
                    row_shr1_add4(R[g], rs[g], Rs[g]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lds_store2_lane15(state + G * GS + i * PS + 4, T[i], Rs[0][i] + rs[0][i]);
                    lds_store2_lane15(state + G * GS + i * PS + 6, Rs[1][i] + rs[1][i], Rs[2][i] + rs[2][i]);
                }
                f32x4 lx = {0.f, 0.f, 0.f, 0.f}, ly = {0.f, 0.f, 0.f, 0.f};
                lx = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], blx, lx, 0, 0, 0);
                ly = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bly, ly, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = 4 * G + i;
                    const float dLa0 = T[i] * cg[0][i] - R[0][i] * r1a[i];   // (R includes the set's background term)
                    const float dLa1 = T[i] * cg[1][i] - R[1][i] * r1a[i];
                    const float dLa2 = T[i] * cg[2][i] - R[2][i] * r1a[i];
                    const float am = ok[i] ? araw[i] : 0.f;
                    const float dLp_tap = am * dLa0;            // dL/dpower of the tap set
                    const float dLp_op = am * (dLa0 + dLa1);     // ... of the sets blended with the live opacity
                    const float dLp = am * (dLa0 + dLa1 + dLa2);  // ... of all sets
                    d_mom = __builtin_amdgcn_mfma_f32_16x16x4f32(momrow[64 * s], dLp, d_mom, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NA; ++q) d_f[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(hft[s][q], wgt[i], d_f[q], 0, 0, 0);
                    s_op += dLp_op;
                    const float gx = dLp_tap * lx[i], gy = dLp_tap * ly[i];
                    s_tx += gx;
                    s_ty += gy;
                    if (ABS) {
                        s_ax += fabsf(gx);
                        s_ay += fabsf(gy);
                    }
                }
            }
            // ---- chunk epilogue
            s_op = rows_sum(s_op, lane);
            s_tx = rows_sum(s_tx, lane);
            s_ty = rows_sum(s_ty, lane);
            if (ABS) {
                s_ax = rows_sum(s_ax, lane);
                s_ay = rows_sum(s_ay, lane);
            }
            if (j0 + nl < cnt) {
                float *rec = slab + (j0 + nl - p0) * NC;   // row = position in this round
                const float D0 = d_mom[0];
                if (kk == 0) {
                    const float Dx = d_mom[1], Dy = d_mom[2], Dxx = d_mom[3];
                    rec[0] = cA * Dx + cB * Dy - (cA * uc + cB * vc) * D0;
                    rec[1] = cB * Dx + cC * Dy - (cB * uc + cC * vc) * D0;
                    rec[2] = -0.5f * (uc * uc * D0 - 2.f * uc * Dx + Dxx);
                    rec[5] = o > 0.f ? s_op / o : 0.f;
                } else if (kk == 1) {
                    const float Dx = d_mom[1], Dy = d_mom[2], Dxy = d_mom[3];
                    rec[3] = -(uc * vc * D0 - uc * Dy - vc * Dx + Dxy);
                    rec[6] = ABS ? s_ax : 0.f;
                    rec[7] = ABS ? s_ay : 0.f;
                } else if (kk == 2) {
                    const float Dy = d_mom[1], Dyy = d_mom[2];
                    rec[4] = -0.5f * (vc * vc * D0 - 2.f * vc * Dy + Dyy);
                } else {
                    rec[8] = -s_tx;   // d uv of the tap set alone: sum dLp (conic (pixel - centre))
                    rec[9] = -s_ty;
                    rec[10] = 0.f;    // (padding of the geometry part: whole 16-byte chunks)
                    rec[11] = 0.f;
                }
#pragma unroll
                for (int q = 0; q < NA; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < fcnt[q]) rec[fbase[q] + i] = d_f[q][i];
            }
        }
        if (lane == 0) CL.more[w] = cnt > p0 + CAP;
        if (p0 == 0) {
            // behind the chunks, ahead of the barrier: the pair slots of this thread's combine passes and the next
            // super-batch's records and ids (requested here their registers are not live across the chunk loop)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ql = ce + r * EPI;
                sl[r] = (ce < EPI && ql < nb) ? slots[top - nb + 1 + ql] : 0;
            }
            st.load_payload(A, tid);
            st.load_ids(A, tid, range.x, pos, batch + 2);
        }
        __syncthreads();
        if (ce < EPI) {  // combine the four slabs into the entries' pair slots: the entry's row in every wave's slab (the zero
                         // row CAP where the wave did not replay it in this round), three floats per thread
            const int lo = top - nb + 1;
            int r = 0;
            for (int ql = ce; ql < nb; ql += EPI, ++r) {
                const unsigned int p4 = CL.pos4[nb - 1 - ql];
                float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned int pp = umin_(((p4 >> (8 * ww)) & 0xffu) - (unsigned)p0, (unsigned)CAP);
                    const float *row = s_acc[ww] + pp * NC + 3 * cc;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (3 * cc + j < NC) v[j] += row[j];
                }
                const int slot = r == 0 ? sl[0] : r == 1 ? sl[1] : r == 2 ? sl[2] : slots[lo + ql];
                if (p0 > 0) {   // a further round of the same super-batch: the same thread stored the record before
                    const float *old = pair_buf + (size_t)slot * NCP + 3 * cc;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (3 * cc + j < NCP) v[j] += old[j];
                }
                store_part(slot, v[0], v[1], v[2]);
            }
        }
        const bool more = CL.more[0] | CL.more[1] | CL.more[2] | CL.more[3];
        __syncthreads();
        if (!more) break;
        }
    }
    if (A.dbg_T_front) {
        const int px = bx + (lane & 7), py = by + (lane >> 3);
        if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_state[w][pixoff(lane) + 4];
    }
}

// ------------------------------------------------------------------ three-set backward on QUARTER lists
// blend_bwd_sets_kernel with the strip walk of blend_bwd_quarter_kernel: every 4x4 quarter of the wave's block walks its own
// list (the forward's quarter bits), a step = 16 survivors of one quarter's list x its 16 pixels.  Per staged entry, once (the
// lanes that park its two geometry parts): the polynomial's coefficients and the constant terms of the tap factor into the
// record's free floats (part 9 = q0 qx qy qxx, part 10 = qxy qyy -c0x -c0y, part 11 = 0).  A step adds raw sums into the
// survivor's slab row (rows by position in the wave's block-level list; float4 read-add-write, a lane group per float4):
//   [M0 Mx My Mxx | Mxy Myy ay(B) . | op tx ty ax (A) | op tx ty ax (B) | ay(A) . . . | dL_dfeature of the 28 slots]
// (block-centred moments of dL/dpower; A / B: the per-lane sums of lane groups 0 + 2 / 1 + 3); the combine moves the moments to
// the tile centre, maps them to d uv / d conic and routes the slots to the row's channels -- once per entry.  Wave w walks
// quarter w of each of the tile's four blocks (see blend_bwd_quarter_kernel).
struct SetsQCfg {
    static constexpr int CH = SetsCfg::CH, NK = SetsCfg::NK, NA = SetsCfg::NA, SB = SetsCfg::SB, NG = SetsCfg::NG, PS = SetsCfg::PS;
    static constexpr int CAP = SB;          // a row per staged entry at most: no rounds
    static constexpr int RW = 20 + CH + 4;  // 48 floats + 4 of padding: the float4 k of 16 consecutive rows sit on 16 different bank quads
    static constexpr int RQ = Rec<CH>::RQ;  // 12 parts of the packed record
    // staged record (LDS): parts 0-1 geometry | parts 2-9 the transposed feature slots (sets_fpos) | parts 10-13 the step's
    // other B operands by lane group, Q[kk] = (q1 q2 lx ly)[kk] with q1 = q0 qx qy qxx, q2 = qxy qyy 0 0 (the polynomial),
    // lx = -c0x cA cB 0, ly = -c0y cB cC 0 (the tap factor) | part 14 padding: 15 parts, an ODD number of 16-byte slots, so
    // consecutive entries start on different slots of the 256-byte bank row and no swizzle is needed
    static constexpr int RQL = 15;
};

// STD: the renderer's own plan -- rgb (3 channels, taps) at row channels 0-2, the depth at channel 3, 19 detached attributes
// at channels 4-22 -- whose channel gradients the combine stores as whole float4 (record chunk 3 = slots 0 1 2 4, chunks 4 .. 8 =
// slots 8 .. 27); other plans route every slot to its channel with scalar stores.
// FWDREC (with STD): the records the FORWARD packed for this row (Rec<24>: [u v A B | C o . id | channels 0 .. 22 .], 32 floats)
// are staged directly -- the park scatters a record's channels to their transposed slot positions -- so the backward needs no
// packing launch of its own (17 us per frame at c2) and gathers 128 instead of 192 bytes per entry.
// SMALL = 3 / 4: a plan whose detached set holds at most four / eight channels (the trainer's ['mask_attribute'] /
// ['dino_attribute'] rows, alone or behind track_gs: src/trainer_fragGS.py:657,1214) -- slots 12 .. / 16 .. 27 are empty, so the
// K-slabs j >= SMALL of the colour dot product, the second 16-slot block of the feature-gradient product, their hoisted operands
// (28 / 32 registers instead of 88) and their slab traffic are skipped; records and staging keep the 28-slot layout, the slab rows
// hold 16 slots (36 floats).
#ifndef BLEND_SETS_SMALL_SB
#define BLEND_SETS_SMALL_SB 48   // SMALL: super-batch (= slab rows per wave) with which 36-float rows leave 50.9 KB of LDS = three workgroups per
                                 // CU (52 entries are 54.3 KB: three on paper, two on the chip -- 313 us per frame against 256 at 48; 40: 270)
#endif
#ifndef BLEND_SETS_SMALL_MINW
#define BLEND_SETS_SMALL_MINW 3
#endif
template <bool ABS, bool STD, bool FWDREC = false, int SMALL = 0>   // SMALL: K-slabs of the colour product in use (3 / 4; 0: all seven)
__global__ void __launch_bounds__(256, (SMALL ? BLEND_SETS_SMALL_MINW : BLEND_SETS_MINW))
blend_bwd_sets_quarter_kernel(const BlendArgs B) {
    static_assert(!FWDREC || STD, "the forward's records are only understood for the renderer's own plan");
    static_assert(!SMALL || !STD, "the renderer's own plan fills the 28 slots");
    static_assert(SMALL == 0 || SMALL == 3 || SMALL == 4, "slots 0 .. 11 or 0 .. 15 (one 16-slot block of the feature-gradient product)");
    constexpr int NKU = SMALL ? SMALL : SetsQCfg::NK, NAU = SMALL ? 1 : SetsQCfg::NA, CHU = SMALL ? 4 * SMALL : SetsQCfg::CH;   // slabs / blocks / slots in use
    using Cfg = SetsQCfg;
    constexpr int CH = Cfg::CH, NG = Cfg::NG, NK = Cfg::NK, RQ = Cfg::RQ;
    // SMALL (round 6): slots 12 .. 27 are empty, so a slab row is 20 + 12 (+ 4: an odd number of float4) floats instead of 52, and
    // with a 48-entry super-batch the workgroup holds 50.9 instead of 81 KB of LDS -- THREE workgroups per CU at 168 registers
    // (the full plans' hoisted operands keep them at 220 registers and two): 3|1|4 plan backward 310 -> 256 us per frame
    constexpr int SB = (SMALL && BLEND_SETS_SMALL_MINW >= 3) ? BLEND_SETS_SMALL_SB : Cfg::SB, CAP = SB;
    constexpr int RW = (SMALL && BLEND_SETS_SMALL_MINW >= 3) ? 36 : Cfg::RW;
    constexpr int RQL = Cfg::RQL;
    static_assert(RQ == 12 && Rec<CH>::CULL == 43 && SB <= 64 && CAP * RW >= 32 * CH, "record floats 40-42 are free; the staging of 32 pixels fits a slab");
    __shared__ float4 s_rec[(SB + 1) * RQL];            // staged records (SetsQCfg::RQL parts each), slot SB = inert
    auto qpart = [](int e, int p) { return e * RQL + p; };
    __shared__ unsigned int s_keep[64];                 // (one word per lane of the list build: entries SB .. 63 stay zero)
    __shared__ unsigned int s_pos4[64];
    constexpr int QL = SB + 16;                         // a list and its padding (the inert entry, 16 times)
    __shared__ unsigned short s_qlist[(4 * 4 + 1) * QL];   // [wave][quarter][QL]: entry | (slab row) << 8 (+ one list of slack: the
                                                           // step loop reads the NEXT step's entry one step ahead, also past the last list)
    __shared__ __attribute__((aligned(16))) float s_acc[4][(CAP + 1) * RW];
    // replay state of a pixel: [T, R of set 0 1 2] (float4 rows, the four lane groups' rows of a step skewed by 4 floats) and
    // its ncontrib (own array)
    constexpr int KS = 4 * 4 + 4, GS = 4 * KS;
    __shared__ __attribute__((aligned(16))) float s_state[4][4 * GS];
    __shared__ int s_last[4][64];
    auto pixoff = [](int q) { return (q >> 4) * GS + ((q >> 2) & 3) * KS + (q & 3) * 4; };
    __shared__ float s_mom[16 * 32];                    // [step][lane group][row & 7], see blend_bwd_quarter_kernel
    __shared__ int s_wmax[4];
    __shared__ float s_l1[4][3];                        // loss-fused: the waves' sums of |pred - target| per set
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int NC = NG + A.C, NCP = PAIR_STRIDE(NC);
    float *const pair_buf = B.pair_buf + (size_t)frame * (size_t)B.cap * NCP;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int wx = 4 * (w & 1), wy = 4 * (w >> 1);   // this wave's quarter inside every block
    const float tcx = (float)(tx * TILE) + 7.5f, tcy = (float)(ty * TILE) + 7.5f;
    const float ox = (float)wx - 7.5f, oy = (float)wy - 7.5f;
    const int nl = lane & 15, kk = lane >> 4;
    // strip walk: q = 16 G + 4 kk + i  <->  block G = (gx, gy), pixel (8 gx + i, 8 gy + kk) + (wx, wy) of the tile
    auto qx = [](int q) { return 8 * ((q >> 4) & 1) + (q & 3); };
    auto qy = [](int q) { return 8 * (q >> 5) + ((q >> 2) & 3); };
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // moment operand about the centre of the wave's pixel set: rows 0-3 = 1 x y xx, rows 4-5 = xy yy
        const int st = 4 * w + r, Gs = st >> 2, is = st & 3;
        const int q = 16 * Gs + 4 * kk + is;
        const float x = (float)qx(q) - 5.5f, y = (float)qy(q) - 5.5f;
        float v = 0.f;
        if (nl < 4) v = nl == 0 ? 1.f : nl == 1 ? x : nl == 2 ? y : x * x;
        else if (nl < 6) v = nl == 4 ? x * y : y * y;
        if (nl < 8) s_mom[32 * st + 8 * kk + nl] = v;
    }
    float phi1[4], phi2[4];
#pragma unroll
    for (int Gs = 0; Gs < 4; ++Gs) {
        const int q = 16 * Gs + nl;
        const float x = (float)qx(q) + ox, y = (float)qy(q) + oy;
        phi1[Gs] = kk == 0 ? 1.f : kk == 1 ? x : kk == 2 ? y : x * x;
        phi2[Gs] = kk == 0 ? x * y : kk == 1 ? y * y : 0.f;
    }
    const int myq = lane;                              // lane <-> pixel `lane` of the strip walk
    const int lx = wx + qx(myq), ly = wy + qy(myq);    // its position in the tile
    float *stage = s_acc[w];  // [pixel of the half block (raster)][slot]
    float gpix[CH];
    {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        const size_t HW = (size_t)A.H * A.W;
        const bool inside = (px < A.W) && (py < A.H);
        const size_t pix = inside ? (size_t)A.W * (size_t)py + px : 0;   // (a lane outside the image loads pixel 0 and drops it)
        const float Tf = inside ? A.final_T[pix] : 0.f;
        const int last = inside ? A.ncontrib[pix] : 0;
        // every load of the hoist is issued UNCONDITIONALLY at a clamped address and masked afterwards: under `if (inside && c >= 0)`
        // each load sat in its own exec-masked block with its wait behind it -- 23 (loss-fused: 46) dependent round trips per wave,
        // ~9 us per workgroup at c2
        float tgt[CH];   // loss-fused: the targets (else unused)
#pragma unroll
        for (int k = CHU; k < CH; ++k) gpix[k] = 0.f;
#pragma unroll
        for (int k = 0; k < CHU; ++k) {
            const int c = sets_slot_channel(A, k);
            const int gi = k < 4 ? 0 : k < 8 ? 1 : 2;
            const int cc = c >= 0 ? c : 0;
#if BLEND_ABL == 3
            gpix[k] = 0.001f * (float)(k + lane); tgt[k] = 0.f; continue;   // timing ablation: no image loads (results invalid)
#endif
            if (A.dL_dout) {
                gpix[k] = A.dL_dout[(size_t)cc * HW + pix];
            } else {
                const float *d = gi == 0 ? A.sdl0 : gi == 1 ? A.sdl1 : A.sdl2;
                const int o = c >= 0 ? k - (gi == 0 ? 0 : gi == 1 ? 4 : 8) : 0;
                const float *dd = d ? d : A.final_T;      // (a set without channels has no image: any valid address, dropped below)
                if (A.l1_pred) {   // (uniform) d holds the TARGET, the gradient follows from the forward's output row
                    tgt[k] = dd[(size_t)o * HW + pix];
                    gpix[k] = A.l1_pred[(size_t)cc * HW + pix];
                } else {
                    gpix[k] = dd[(size_t)o * HW + pix];
                }
            }
        }
        float bgd[3] = {0.f, 0.f, 0.f};
        float l1a[3] = {0.f, 0.f, 0.f};   // loss-fused: sum |pred - target| of this pixel, per set
#pragma unroll
        for (int k = 0; k < CHU; ++k) {
            const int c = sets_slot_channel(A, k);
            const int gi = k < 4 ? 0 : k < 8 ? 1 : 2;
            float g = gpix[k];
            if (A.l1_pred) {   // the gradient of scale * |pred - target| (splat_l1_loss_grad's rule)
                const float diff = g - tgt[k];
                const float sc = gi == 0 ? A.l1_s0 : gi == 1 ? A.l1_s1 : A.l1_s2;
                g = diff > 0.f ? sc : (diff < 0.f ? -sc : 0.f);
                l1a[gi] += (inside && c >= 0) ? fabsf(diff) : 0.f;
            }
            g = (inside && c >= 0) ? g : 0.f;
            gpix[k] = g;
            bgd[gi] += (gi == 0 ? A.s0bg : gi == 1 ? A.s1bg : A.s2bg) * g;
        }
        if (A.l1_pred) {   // the wave's three sums; thread g < 3 writes the workgroup's behind the barrier below
#pragma unroll
            for (int g_ = 0; g_ < 3; ++g_) {
                const float t_ = wave_sum_to_lane63(l1a[g_]);
                if (lane == 63) s_l1[w][g_] = t_;
            }
        }
        float *r = s_state[w] + pixoff(myq);
        r[0] = Tf;
        r[1] = Tf * bgd[0]; r[2] = Tf * bgd[1]; r[3] = Tf * bgd[2];
        s_last[w][myq] = last;
        const int wmax = wave_max_i(last);
        if (lane == 0) s_wmax[w] = wmax;
    }
    if (FWDREC) {   // the park writes a record's channels float by float: the slots no channel maps to and the padding stay zero
        for (int c = tid; c < SB * RQL; c += 256) s_rec[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < RQL) s_rec[qpart(SB, tid)] = make_float4(tid == 10 ? -__builtin_inff() : 0.f, 0.f, 0.f, 0.f);   // inert slot SB: q0 = log2(0)
    if (SB < 64 && tid < 64) s_keep[tid] = 0u;          // (entries SB .. 63 of the list build: never kept)
    // ---- dL_dout of the wave's pixels into registers in both MFMA operand layouts, 32 pixels (two quarters) at a time
    float hcg[4][NKU], hft[16][NAU];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int k = 0; k < CH; ++k) stage[(lane & 31) * CH + k] = gpix[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int G = 2 * h; G < 2 * h + 2; ++G)
#pragma unroll
            for (int j = 0; j < NKU; ++j)   // A[m = pixel nl of quarter G][k = slot]: pixel 16 (G & 1) + nl of the half (lanes in walk order)
                hcg[G][j] = stage[(16 * (G & 1) + nl) * CH + 4 * j + kk];
#pragma unroll
        for (int st = 8 * h; st < 8 * h + 8; ++st) {
#pragma unroll
            for (int q = 0; q < NAU; ++q)  // A[m = slot 16 q + nl][k = own pixel of step st = (G, i)]: pixel 16 (G & 1) + 4 kk + i of the half
                hft[st][q] = 16 * q + nl < CH ? stage[(16 * ((st >> 2) & 1) + 4 * kk + (st & 3)) * CH + 16 * q + nl] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < RW) s_acc[w][CAP * RW + lane] = 0.f;   // the slab's zero row
    __syncthreads();
    // (one plain store per workgroup and set; the caller adds them up: reproducible, no atomics)
    if (A.l1_pred && A.l1_sum && tid < 3) A.l1_sum[3 * (size_t)tile + tid] = (s_l1[0][tid] + s_l1[1][tid]) + (s_l1[2][tid] + s_l1[3][tid]);
    const float *momrow = s_mom + 8 * kk + (nl & 7);
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    const int *slots = A.slot_sorted + range.x;
    // combine: entry = lane, role = wave -- wave 0: geometry (floats 0 .. 11 of the record), waves 1 .. 3: the slots' channel gradients
    const int ce = lane, cp = w;
    auto zero_rec = [&](int slot) {
        float *dst = pair_buf + (size_t)slot * NCP;
        for (int c = cp; c < NCP / 4; c += 4) reinterpret_cast<float4 *>(dst)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    for (int ql = n + ce; ql < len; ql += 64) zero_rec(slots[ql]);   // entries nobody replays: zero record
    float *state = s_state[w] + kk * KS;
    const int *lastp = s_last[w] + 4 * kk;   // ncontrib of the own pixel of step (G, i): lastp[16 G + i]
    if (n <= 0) {
        if (A.dbg_T_front) {
            const int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_state[w][pixoff(myq)];
        }
        return;
    }

    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };
    // staging: a QUAD per entry -- thread t <-> entry se = t >> 2, 16-byte part sp = t & 3 of each of the record's three 64-byte
    // sectors (four lanes per sector: coalesced).  One index load and three payload loads per thread and super-batch (of the
    // third sector only parts 8 and 9, the last feature slots, are used; the cull parameters behind them are not), no
    // division, and the coefficient block runs ONCE per wave (the generic Stager -- chunk c = t + 256 k, entry c / 12 -- made every
    // wave run it for each of its three chunks: 230 of the 830 VALU instructions a wave spent per super-batch outside the steps).
    static_assert(SB <= 64 && RQ == 12, "256 threads = 64 entries x 4 parts (entries SB .. 63 idle); three sectors per record");
    const int se = tid >> 2, sp = tid & 3;
    int sid_next;              // Gaussian of entry se, two super-batches ahead (-1: past the list)
    float4 sv0, sv1, sv2 = make_float4(0.f, 0.f, 0.f, 0.f);   // parts sp, 4 + sp, 8 + sp (sp < 2) of entry se, one super-batch ahead
#define SETSQ_STAGE_IDS(b)                                                        \
    do {                                                                          \
        const int q_ = pos(se, (b));                                              \
        sid_next = q_ >= 0 ? A.idx_sorted[range.x + q_] : -1;                     \
    } while (0)
    // (past the list the quad loads Gaussian 0's record: staged entries >= nb are in no list and the combine skips them)
#define SETSQ_STAGE_PAYLOAD()                                                                                                   \
    do {                                                                                                                        \
        const float4 *src_ = reinterpret_cast<const float4 *>(A.pack + (size_t)imax_(sid_next, 0) * (FWDREC ? 32 : Rec<CH>::RS)) + sp; \
        sv0 = src_[0];                                                                                                          \
        sv1 = src_[4];                                                                                                          \
        if (!FWDREC && sp < 2) sv2 = src_[8];                                                                                   \
    } while (0)
    SETSQ_STAGE_IDS(0);
    SETSQ_STAGE_PAYLOAD();
    SETSQ_STAGE_IDS(1);
    auto load_flags = [&](int topb) -> unsigned {
        const int q = topb - tid;
        return (tid < SB && q >= 0) ? (unsigned)A.cull_flags[range.x + q] : 0u;
    };
    unsigned fl_next = load_flags(n - 1);

    int batch = 0;
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        {   // park: the quad's lanes copy their parts and each leaves ONE of the four operand rows Q[sp] (the two geometry parts
            // sit in lanes 0 / 1 of the quad: broadcast)
            const float4 mine = sv0;
            float4 g0, g1;
#define QB_(v, c) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), c, 0xf, 0xf, true))
            g0.x = QB_(mine.x, 0x00); g0.y = QB_(mine.y, 0x00); g0.z = QB_(mine.z, 0x00); g0.w = QB_(mine.w, 0x00);   // quad_perm [0,0,0,0]
            g1.x = QB_(mine.x, 0x55); g1.y = QB_(mine.y, 0x55);                                                       // quad_perm [1,1,1,1]
#undef QB_
            const PowerCoef pc = power_coeffs(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, tcx, tcy);
            const float ut = g0.x - tcx, vt = g0.y - tcy;
            float4 *rb = s_rec + imin_(se, SB) * RQL;
            if (SB < 64 && se >= SB) {
                // (a super-batch below 64 entries: the quads past it park nothing)
            } else if (FWDREC) {
                // forward record: parts 0 1 = geometry, part 2 = r g b depth, part 3 = attributes 0-3, parts 4 + sp = attributes
                // 4 + 4 sp ..; channel -> slot (rgb 0-2 | depth 4 | attribute a 8 + a) -> float sets_fpos(slot)
                float *rf = reinterpret_cast<float *>(rb);
                if (sp < 2) rb[sp] = mine;
                else if (sp == 2) { rf[8] = mine.x; rf[16] = mine.y; rf[24] = mine.z; rf[9] = mine.w; }
                else { rf[10] = mine.x; rf[18] = mine.y; rf[26] = mine.z; rf[34] = mine.w; }
                rf[11 + sp] = sv1.x; rf[19 + sp] = sv1.y; rf[27 + sp] = sv1.z;   // slots 12 + 4 sp + i at 11 + sp + 8 i
                if (sp < 3) rf[35 + sp] = sv1.w;                                   // (sp 3: channel 23 is the record's padding -> slot 27 stays zero)
            } else {
                rb[sp] = mine;                                // parts 0 .. 3
                rb[4 + sp] = sv1;                             // parts 4 .. 7
                if (sp < 2) rb[8 + sp] = sv2;                 // parts 8, 9
            }
            float4 qv;                                    // Q[sp] = (q1 q2 lx ly)[sp]
            qv.x = sp == 0 ? pc.q0 : sp == 1 ? pc.qx : sp == 2 ? pc.qy : pc.qxx;
            qv.y = sp == 0 ? pc.qxy : sp == 1 ? pc.qyy : 0.f;
            qv.z = sp == 0 ? -(g0.z * ut + g0.w * vt) : sp == 1 ? g0.z : sp == 2 ? g0.w : 0.f;
            qv.w = sp == 0 ? -(g0.w * ut + g1.x * vt) : sp == 1 ? g0.w : sp == 2 ? g1.x : 0.f;
            if (SB == 64 || se < SB) rb[10 + sp] = qv;
        }
        const unsigned fl = fl_next;
        fl_next = load_flags(top - SB);
        if (tid < SB) {   // byte w' bit G = quarter w' of block G in the forward's cull flags (byte G bit w')
            unsigned kw = 0u;
            if (tid < nb) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned t4 = (fl >> ww) & 0x01010101u;
                    const unsigned nib = ((t4 * 0x01020408u) >> 24) & 0xfu;
                    if (top - tid < s_wmax[ww]) kw |= nib << (8 * ww);
                }
            }
            s_keep[tid] = kw;
        }
        __syncthreads();
        float *slab = s_acc[w];
        // ---- this wave's lists (SB = 64: one entry per lane)
        unsigned cqw = 0u;
        int cnt;
        {
            const int e = lane;
            const unsigned bits = (s_keep[e] >> (8 * w)) & 0xfu;
            const bool kb = bits != 0u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(kb);
            const int ps = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            reinterpret_cast<unsigned char *>(s_pos4)[4 * e + w] = kb ? (unsigned char)ps : (unsigned char)255;
            const unsigned word = (bits * 0x00204081u) & 0x01010101u;
            unsigned incl = word;
            asm volatile("s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                         : "+v"(incl));
            const unsigned posw = incl - word;
            const unsigned short ent = (unsigned short)(e | (ps << 8));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((bits >> q) & 1u) s_qlist[(4 * w + q) * QL + ((posw >> (8 * q)) & 0xffu)] = ent;
            cqw = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            cnt = __popcll(m);
        }
        int cq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cq[q] = (int)((cqw >> (8 * q)) & 0xffu);
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_qlist[(4 * w + q) * QL + cq[q] + lane] = (unsigned short)(SB | (CAP << 8));   // pad: inert entry, zero row
        }
        {   // rows start from zero
            float4 *z = reinterpret_cast<float4 *>(slab);
            for (int c = lane; c < cnt * (RW / 4); c += WAVE) z[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the list entry of a step is read ONE STEP AHEAD (the step's operand reads hang on it: one LDS round trip less on the
        // step's critical path).  Steps in walk order: quarter 0's, then quarter 1's ...; the entry behind the last step of a
        // quarter is the first of the next quarter that has any (read past the last list: slack, unused).
#if BLEND_SETS_LE_AHEAD
        int gfirst = 0;
#pragma unroll
        for (int q = 3; q >= 0; --q) gfirst = cq[q] > 0 ? q : gfirst;
        unsigned le_next = s_qlist[(4 * w + gfirst) * QL + nl];
#endif
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            for (int j0 = 0; j0 < ((BLEND_QABL & 4) ? 0 : cq[G]); j0 += 16) {   // (BLEND_QABL 4: timing ablation, no steps)
#if BLEND_SETS_LE_AHEAD
                const unsigned le = le_next;
                {
                    int gn = 4;       // the next quarter with a list (4: none -- the slack list, or the next wave's first)
#pragma unroll
                    for (int q = 3; q > G; --q) gn = cq[q] > 0 ? q : gn;
                    const bool lastq = j0 + 16 >= cq[G];
                    le_next = s_qlist[(lastq ? (4 * w + gn) * QL : (4 * w + G) * QL + j0 + 16) + nl];
                }
#else
                const unsigned le = s_qlist[(4 * w + G) * QL + j0 + nl];
#endif
                const int e = le & 0xffu, row = le >> 8;
                const int qn = top - e;
                // the step's B operands of this lane group: three 16-byte reads (Q[kk], slots kk 4+kk 8+kk 12+kk, slots 16+kk 20+kk 24+kk)
                const float4 *er = s_rec + e * RQL;
                const float4 qv = er[10 + kk], t0 = er[2 + 2 * kk];
                const float4 t1 = SMALL ? make_float4(0.f, 0.f, 0.f, 0.f) : er[3 + 2 * kk];
                const float bq1 = qv.x, bq2 = qv.y, blx = qv.z, bly = qv.w;
                float bf[NK];
                bf[0] = t0.x; bf[1] = t0.y; bf[2] = t0.z; bf[3] = t0.w; bf[4] = t1.x; bf[5] = t1.y; bf[6] = t1.z;
                static_assert(NK == 7, "slots 4 j + kk, j = 0 .. 6");
#if BLEND_SETS_EARLY_ROWS
                // the survivor's slab row -- what the step's epilogue adds into -- is requested NOW (a padded list slot reads the
                // zero row), so that its LDS round trip runs under the step instead of in front of the epilogue's adds
                float *const rr = slab + row * RW;
                float4 *const p1 = reinterpret_cast<float4 *>(rr + 4 * kk);
                float4 *const pf = reinterpret_cast<float4 *>(rr + 20 + 4 * kk);   // slots 4 kk .. (q = 0) and 16 + 4 kk .. (q = 1)
                const float4 o1_in = *p1, f0_in = pf[0];
                const float4 f1_in = SMALL ? make_float4(0.f, 0.f, 0.f, 0.f) : pf[4];   // (lane group 3: pf[4] is the row's padding)
                const float ay_in = rr[16];
#endif
                f32x4 d_mom = {0.f, 0.f, 0.f, 0.f};
                f32x4 d_f[NAU];
#pragma unroll
                for (int a = 0; a < NAU; ++a) d_f[a] = f32x4{0.f, 0.f, 0.f, 0.f};
                float s_op = 0.f, s_tx = 0.f, s_ty = 0.f, s_ax = 0.f, s_ay = 0.f;
                asm volatile("" ::: "memory");
                f32x4 pw = {0.f, 0.f, 0.f, 0.f}, cv0 = {0.f, 0.f, 0.f, 0.f}, cv1 = {0.f, 0.f, 0.f, 0.f}, cv2 = {0.f, 0.f, 0.f, 0.f};
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bq1, pw, 0, 0, 0);
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi2[G], bq2, pw, 0, 0, 0);
                cv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][0], bf[0], cv0, 0, 0, 0);
                cv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][1], bf[1], cv1, 0, 0, 0);
#pragma unroll
                for (int j = 2; j < NKU; ++j) cv2 = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][j], bf[j], cv2, 0, 0, 0);
                float araw[4], a[4], r1a[4], rp[4], Ts4[4], Rs[3][4], cg[3][4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int last = lastp[16 * G + i];
                    const float4 sb = *reinterpret_cast<const float4 *>(state + G * GS + i * 4);
                    Ts4[i] = sb.x;
                    Rs[0][i] = sb.y; Rs[1][i] = sb.z; Rs[2][i] = sb.w;
                    cg[0][i] = cv0[i]; cg[1][i] = cv1[i]; cg[2][i] = cv2[i];
                    bool pw_ok;
                    araw[i] = exp2_guard(pw[i], pw_ok);
                    const float alpha = fminf(0.99f, araw[i]);
                    ok[i] = (qn < last) && pw_ok && !(alpha < (1.0f / 255.0f));
                    a[i] = ok[i] ? alpha : 0.f;
                    r1a[i] = __builtin_amdgcn_rcpf(1.f - a[i]);
                    rp[i] = r1a[i];
                }
                row_scan_mul4(rp[0], rp[1], rp[2], rp[3]);
                float T[4], wgt[4], rs[3][4], R[3][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T[i] = Ts4[i] * rp[i];
                    wgt[i] = a[i] * T[i];
#pragma unroll
                    for (int g = 0; g < 3; ++g) rs[g][i] = cg[g][i] * wgt[i];
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    row_scan_add4(rs[g][0], rs[g][1], rs[g][2], rs[g][3]);
                    row_shr1_add4(R[g], rs[g], Rs[g]);
                }
#pragma unroll
                for (int i = 0; i < 4; i += 2)   // two pixels (four pairs) per exec switch
                    lds_store2x4_lane15<0, 2, 4, 6>(state + G * GS + i * 4, T[i], Rs[0][i] + rs[0][i], Rs[1][i] + rs[1][i],
                                                    Rs[2][i] + rs[2][i], T[i + 1], Rs[0][i + 1] + rs[0][i + 1], Rs[1][i + 1] + rs[1][i + 1],
                                                    Rs[2][i + 1] + rs[2][i + 1]);
                f32x4 lx4 = {0.f, 0.f, 0.f, 0.f}, ly4 = {0.f, 0.f, 0.f, 0.f};   // -conic (centre - pixel): the sign returns in the combine
                lx4 = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], blx, lx4, 0, 0, 0);
                ly4 = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bly, ly4, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = 4 * G + i;
                    const float dLa0 = T[i] * cg[0][i] - R[0][i] * r1a[i];
                    const float dLa1 = T[i] * cg[1][i] - R[1][i] * r1a[i];
                    const float dLa2 = T[i] * cg[2][i] - R[2][i] * r1a[i];
                    const float am = ok[i] ? araw[i] : 0.f;
                    const float dLp_tap = am * dLa0;
                    const float dLp_op = am * (dLa0 + dLa1);
                    const float dLp = am * (dLa0 + dLa1 + dLa2);
                    d_mom = __builtin_amdgcn_mfma_f32_16x16x4f32(momrow[32 * s], dLp, d_mom, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NAU; ++q) d_f[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(hft[s][q], wgt[i], d_f[q], 0, 0, 0);
                    s_op += dLp_op;
                    const float gx = dLp_tap * lx4[i], gy = dLp_tap * ly4[i];
                    s_tx += gx;
                    s_ty += gy;
                    if (ABS) {
                        s_ax += fabsf(gx);
                        s_ay += fabsf(gy);
                    }
                }
                // ---- step epilogue: lane groups kk and kk ^ 2 pool their sums (one swap each); then a float4 read-add-write per
                //      lane group: kk 0: moments 0-3 | kk 1: Mxy Myy ay(B) | kk 2: sums A (+ ay(A)) | kk 3: sums B; and the feature quads
                auto half = [&](float v) {
                    const u32x2_b r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // own + the value of lane ^ 32, in every lane
                };
                s_op = half(s_op); s_tx = half(s_tx); s_ty = half(s_ty);
                if (ABS) { s_ax = half(s_ax); s_ay = half(s_ay); }
#if BLEND_SETS_EARLY_ROWS
                if (j0 + nl < cq[G]) {
                    float4 o1 = o1_in;
                    o1.x += kk < 2 ? d_mom[0] : s_op;
                    o1.y += kk < 2 ? d_mom[1] : s_tx;
                    o1.z += kk == 0 ? d_mom[2] : kk == 1 ? s_ay : s_ty;
                    o1.w += kk == 0 ? d_mom[3] : kk == 1 ? 0.f : s_ax;
                    *p1 = o1;
                    if (ABS && kk == 2) rr[16] = ay_in + s_ay;
                    float4 f0 = f0_in;
                    f0.x += d_f[0][0]; f0.y += d_f[0][1]; f0.z += d_f[0][2]; f0.w += d_f[0][3];
                    pf[0] = f0;
                    if (!SMALL && kk < 3) {
                        float4 f1 = f1_in;
                        f1.x += d_f[NAU - 1][0]; f1.y += d_f[NAU - 1][1]; f1.z += d_f[NAU - 1][2]; f1.w += d_f[NAU - 1][3];
                        pf[4] = f1;
                    }
                }
#else
                if (j0 + nl < cq[G]) {
                    float *rr = slab + row * RW;
                    float4 v1;
                    v1.x = kk < 2 ? d_mom[0] : s_op;
                    v1.y = kk < 2 ? d_mom[1] : s_tx;
                    v1.z = kk == 0 ? d_mom[2] : kk == 1 ? s_ay : s_ty;
                    v1.w = kk == 0 ? d_mom[3] : kk == 1 ? 0.f : s_ax;
                    float4 *p1 = reinterpret_cast<float4 *>(rr + 4 * kk);
                    float4 o1 = *p1;
                    o1.x += v1.x; o1.y += v1.y; o1.z += v1.z; o1.w += v1.w;
                    *p1 = o1;
                    if (ABS && kk == 2) rr[16] += s_ay;
                    float4 *pf = reinterpret_cast<float4 *>(rr + 20 + 4 * kk);   // slots 4 kk .. (q = 0) and 16 + 4 kk .. (q = 1)
                    float4 f0 = pf[0];
                    f0.x += d_f[0][0]; f0.y += d_f[0][1]; f0.z += d_f[0][2]; f0.w += d_f[0][3];
                    pf[0] = f0;
                    if (!SMALL && kk < 3) {
                        float4 f1 = pf[4];
                        f1.x += d_f[NAU - 1][0]; f1.y += d_f[NAU - 1][1]; f1.z += d_f[NAU - 1][2]; f1.w += d_f[NAU - 1][3];
                        pf[4] = f1;
                    }
                }
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        const int slot_mine = ce < nb ? slots[top - ce] : 0;   // entry ce of the combine
        SETSQ_STAGE_PAYLOAD();
        SETSQ_STAGE_IDS(batch + 2);
#if BLEND_SETS_TWO_BARRIERS
        // the combine's only reads of the staged records, taken BEFORE the barrier: behind it a fast wave may already park the next
        // super-batch over them while a slow one still combines (the combine otherwise reads the slabs and position bytes, which
        // nobody writes before the next super-batch's barrier) -- two barriers per super-batch instead of three
        float4 cg0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float cgC = 0.f, cgo = 0.f;
        if (cp <= 1) {
            cg0 = s_rec[qpart(ce, 0)];
            const float4 t_ = s_rec[qpart(ce, 1)];
            cgC = t_.x; cgo = t_.y;
        }
#endif
        __syncthreads();
#if !BLEND_SETS_TWO_BARRIERS
        float4 cg0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float cgC = 0.f, cgo = 0.f;
        if (cp <= 1) {
            cg0 = s_rec[qpart(ce, 0)];
            const float4 t_ = s_rec[qpart(ce, 1)];
            cgC = t_.x; cgo = t_.y;
        }
#endif
        // ---- combine: entry ce = lane, role cp = wave -- wave 0: the geometry part of the record, waves 1 .. 3: the channel
        //      gradients.  Wave-uniform roles: a wave runs ONE of the two paths (four parts per entry in neighbouring lanes made
        //      every wave run both).
        if (ce < nb) {
            const int e = ce;
            const unsigned int p4 = s_pos4[e];
            float *dst = pair_buf + (size_t)slot_mine * NCP;
            if (STD && cp == 0) {
                // the moments (about the centre of each wave's pixel set) to the tile centre, then to d uv / d conic: floats 0 .. 4
                float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // M0 Mx My Mxx Mxy Myy
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                    const float4 *rw = reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW);
                    const float4 m = rw[0];
                    const float2 m2 = *reinterpret_cast<const float2 *>(rw + 1);
                    const float bxw = (ww & 1) ? 2.f : -2.f, byw = (ww >> 1) ? 2.f : -2.f;   // centre of the wave's pixel set from the tile centre
                    s[0] += m.x;
                    s[1] += m.y + bxw * m.x;
                    s[2] += m.z + byw * m.x;
                    s[3] += m.w + 2.f * bxw * m.y + (bxw * bxw) * m.x;
                    s[4] += m2.x + bxw * m.z + byw * m.y + (bxw * byw) * m.x;
                    s[5] += m2.y + 2.f * byw * m.z + (byw * byw) * m.x;
                }
                const float4 g0 = cg0;
                const float cC = cgC;
                const float cA = g0.z, cB = g0.w;
                const float uc = g0.x - tcx, vc = g0.y - tcy;
                const float M0 = s[0], Mx = s[1], My = s[2], Mxx = s[3], Mxy = s[4], Myy = s[5];
                float4 r0;
                r0.x = cA * Mx + cB * My - (cA * uc + cB * vc) * M0;
                r0.y = cB * Mx + cC * My - (cB * uc + cC * vc) * M0;
                r0.z = -0.5f * (uc * uc * M0 - 2.f * uc * Mx + Mxx);
                r0.w = -(uc * vc * M0 - uc * My - vc * Mx + Mxy);
                reinterpret_cast<float4 *>(dst)[0] = r0;
                dst[4] = -0.5f * (vc * vc * M0 - 2.f * vc * My + Myy);
            } else if (cp == 0) {
                float s[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // M0 Mx My Mxx Mxy Myy | op tx ty ax ay
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                    const float4 *rw = reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW);
                    const float4 m = rw[0], m2 = rw[1], sa = rw[2], sb2 = rw[3];
                    const float ayA = s_acc[ww][pp * RW + 16];
                    const float bxw = (ww & 1) ? 2.f : -2.f, byw = (ww >> 1) ? 2.f : -2.f;   // centre of the wave's pixel set from the tile centre
                    s[0] += m.x;
                    s[1] += m.y + bxw * m.x;
                    s[2] += m.z + byw * m.x;
                    s[3] += m.w + 2.f * bxw * m.y + (bxw * bxw) * m.x;
                    s[4] += m2.x + bxw * m.z + byw * m.y + (bxw * byw) * m.x;
                    s[5] += m2.y + 2.f * byw * m.z + (byw * byw) * m.x;
                    s[6] += sa.x + sb2.x; s[7] += sa.y + sb2.y; s[8] += sa.z + sb2.z; s[9] += sa.w + sb2.w;
                    s[10] += ayA + m2.z;
                }
                const float4 g0 = cg0;
                const float cA = g0.z, cB = g0.w, cC = cgC, o = cgo;
                const float uc = g0.x - tcx, vc = g0.y - tcy;
                const float M0 = s[0], Mx = s[1], My = s[2], Mxx = s[3], Mxy = s[4], Myy = s[5];
                float4 r0, r1;
                r0.x = cA * Mx + cB * My - (cA * uc + cB * vc) * M0;
                r0.y = cB * Mx + cC * My - (cB * uc + cC * vc) * M0;
                r0.z = -0.5f * (uc * uc * M0 - 2.f * uc * Mx + Mxx);
                r0.w = -(uc * vc * M0 - uc * My - vc * Mx + Mxy);
                r1.x = -0.5f * (vc * vc * M0 - 2.f * vc * My + Myy);
                r1.y = o > 0.f ? s[6] / o : 0.f;
                r1.z = ABS ? s[9] : 0.f;
                r1.w = ABS ? s[10] : 0.f;
                reinterpret_cast<float4 *>(dst)[0] = r0;
                reinterpret_cast<float4 *>(dst)[1] = r1;
                // d uv of the tap set alone (the step's factor carries the minus sign), and the chunk's padding
                reinterpret_cast<float4 *>(dst)[2] = make_float4(s[7], s[8], 0.f, 0.f);
            } else if (STD) {
                // record chunk 3 = slots 0 1 2 4 (r g b | depth), chunk j >= 4 = slots 4 j - 8 .. 4 j - 5, i.e. float4 j + 3 of a
                // slab row (slot s at row float 20 + s)
                // Roles (the moments' move to the tile centre makes wave 0's part the longest; the waves meet at the barrier
                // behind the combine): wave 1 = the per-lane sums (d opacity, taps, |taps|: floats 5 .. 9) + chunk 3, wave 2 =
                // chunks 4 5 6, wave 3 = chunks 7 8.
                if (cp == 1) {
                    float op = 0.f, tx_ = 0.f, ty_ = 0.f, ax = 0.f, ay = 0.f;
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                        const float *rf = s_acc[ww] + pp * RW;
                        const float4 *rw = reinterpret_cast<const float4 *>(rf);
                        const float4 sa = rw[2], sb2 = rw[3], v5 = rw[5];
                        op += sa.x + sb2.x; tx_ += sa.y + sb2.y; ty_ += sa.z + sb2.z; ax += sa.w + sb2.w;
                        ay += rf[16] + rf[6];
                        a.x += v5.x; a.y += v5.y; a.z += v5.z; a.w += rf[24];
                    }
                    const float o = cgo;
                    dst[5] = o > 0.f ? op / o : 0.f;
                    dst[6] = ABS ? ax : 0.f;
                    dst[7] = ABS ? ay : 0.f;
                    reinterpret_cast<float4 *>(dst)[2] = make_float4(tx_, ty_, 0.f, 0.f);   // d uv of the tap set alone | padding
                    reinterpret_cast<float4 *>(dst)[3] = a;                                 // r g b | depth
                } else {
                    // record chunks 4 5 6 (wave 2) / 7 8 (wave 3) = float4 7 8 9 / 10 11 of a slab row
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
                    const int r0_ = cp == 2 ? 7 : 10;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                        const float4 *rw = reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW) + r0_;
                        const float4 va = rw[0], vb = rw[1];
                        a.x += va.x; a.y += va.y; a.z += va.z; a.w += va.w;
                        b.x += vb.x; b.y += vb.y; b.z += vb.z; b.w += vb.w;
                        if (cp == 2) {
                            const float4 vc_ = rw[2];
                            c.x += vc_.x; c.y += vc_.y; c.z += vc_.z; c.w += vc_.w;
                        }
                    }
                    float4 *d4 = reinterpret_cast<float4 *>(dst) + (cp == 2 ? 4 : 7);
                    d4[0] = a;
                    d4[1] = b;   // (wave 3: channels 20 21 22 and the record's last padding float)
                    if (cp == 2) d4[2] = c;
                }
            } else {
                // slots [8 (cp - 1) .. ) of the 28 (cp 1: 0-7, cp 2: 8-19, cp 3: 20-27), summed over the waves, to their row channels
                const int s0 = cp == 1 ? 0 : cp == 2 ? 8 : 20, ns = SMALL ? (cp == 1 ? 8 : cp == 2 ? 4 * SMALL - 8 : 0) : (cp == 2 ? 12 : 8);
                float f[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) f[k] = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                    const float4 *rw = reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW + 20 + s0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (4 * c < ns) {
                            const float4 v = rw[c];
                            f[4 * c] += v.x; f[4 * c + 1] += v.y; f[4 * c + 2] += v.z; f[4 * c + 3] += v.w;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    if (k < ns) {
                        const int c = sets_slot_channel(A, s0 + k);
                        if (c >= 0) dst[NG + c] = f[k];
                    }
                }
                // (floats NC .. NCP - 1 of a record are padding)
            }
        }
#if !BLEND_SETS_TWO_BARRIERS
        __syncthreads();
#endif
    }
    if (A.dbg_T_front) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_state[w][pixoff(myq)];
    }
}

#undef SETSQ_STAGE_IDS
#undef SETSQ_STAGE_PAYLOAD

// ------------------------------------------------------------------ wide single-set backward on QUARTER lists
// The strip walk of blend_bwd_sets_quarter_kernel for ONE feature set of 16 .. 32 channels (the wide instantiations of
// blend_bwd_mfma_kernel run 2 workgroups per CU on block lists, their four waves adding into one slab with LDS float atomics
// at 32 channels): quarter lists from the forward's quarter bits, wave w on quarter w of each of the tile's four blocks,
// dL_dout of the wave's pixels in registers in both MFMA operand layouts, the polynomial's coefficients once per staged entry,
// raw block-set-centred moments + the feature gradients added into slab rows by position (float4 read-add-write), the map to
// d uv / d conic / d opacity once per entry in the combine.  Record: GradLayout<false, false> = [ux uy ca cb cc o | features].
template <int CH>
struct WideQCfg {
    static_assert(CH % 4 == 0 && CH >= 16 && CH <= 32, "whole K-slabs");
    static constexpr int SB = 64, CAP = SB;
    static constexpr int NK = CH / 4, NA = (CH + 15) / 16;
    static constexpr int NG = GradLayout<false, false>::NG, NC = NG + CH, NCP = PAIR_STRIDE(NC);
    static constexpr int RW = (((8 + CH) / 4) | 1) * 4;   // slab row [M0 Mx My Mxx | Mxy Myy . . | CH feature gradients], an odd number of float4
    static constexpr int RQ = Rec<CH>::RQ;
};

template <int CH, bool EXACT>
__global__ void __launch_bounds__(256, (CH <= 20 ? 3 : 2))   // 16 / 20 channels: 168 registers and 48 KB of LDS -- three workgroups per CU
blend_bwd_wide_quarter_kernel(const BlendArgs B) {
    using Cfg = WideQCfg<CH>;
    constexpr int SB = Cfg::SB, CAP = Cfg::CAP, NK = Cfg::NK, NA = Cfg::NA, NG = Cfg::NG, NCP = Cfg::NCP, RW = Cfg::RW, RQ = Cfg::RQ;
    static_assert(RQ % 4 == 0 && CAP * RW >= 32 * CH && SB == 64, "swizzle groups; staging of 32 pixels fits a slab; a quad per staged entry");
    __shared__ float4 s_rec[(SB + 1) * RQ];
    auto qpart = [](int e, int p) { return e * RQ + ((p & ~3) | ((p & 3) ^ ((e >> 2) & 3))); };
    __shared__ float4 s_coef[(SB + 1) * 2];             // [q0 qx qy qxx | qxy qyy 0 0] of the staged entries, slot SB = inert
    __shared__ unsigned int s_keep[SB];
    __shared__ unsigned int s_pos4[SB];
    __shared__ unsigned short s_qlist[4][4][SB + 16];
    __shared__ __attribute__((aligned(16))) float s_acc[4][(CAP + 1) * RW];
    constexpr int KS = 4 * 2 + 2, GS = 4 * KS;          // replay state [T, R] per pixel, lane groups' rows skewed
    __shared__ __attribute__((aligned(8))) float s_state[4][4 * GS];
    __shared__ int s_last[4][64];
    auto pixoff = [](int q) { return (q >> 4) * GS + ((q >> 2) & 3) * KS + (q & 3) * 2; };
    __shared__ float s_mom[16 * 32];
    __shared__ int s_wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int RST = B.rec_stride ? B.rec_stride : NCP;
    float *const pair_buf = B.pair_buf + (size_t)frame * (size_t)B.cap * RST + B.rec_off;
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int wx = 4 * (w & 1), wy = 4 * (w >> 1);
    const float tcx = (float)(tx * TILE) + 7.5f, tcy = (float)(ty * TILE) + 7.5f;
    const float ox = (float)wx - 7.5f, oy = (float)wy - 7.5f;
    const int cn = EXACT ? CH : A.cn;
    const int nl = lane & 15, kk = lane >> 4;
    auto qx = [](int q) { return 8 * ((q >> 4) & 1) + (q & 3); };
    auto qy = [](int q) { return 8 * (q >> 5) + ((q >> 2) & 3); };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int st = 4 * w + r, Gs = st >> 2, is = st & 3;
        const int q = 16 * Gs + 4 * kk + is;
        const float x = (float)qx(q) - 5.5f, y = (float)qy(q) - 5.5f;
        float v = 0.f;
        if (nl < 4) v = nl == 0 ? 1.f : nl == 1 ? x : nl == 2 ? y : x * x;
        else if (nl < 6) v = nl == 4 ? x * y : y * y;
        if (nl < 8) s_mom[32 * st + 8 * kk + nl] = v;
    }
    float phi1[4], phi2[4];
#pragma unroll
    for (int Gs = 0; Gs < 4; ++Gs) {
        const int q = 16 * Gs + nl;
        const float x = (float)qx(q) + ox, y = (float)qy(q) + oy;
        phi1[Gs] = kk == 0 ? 1.f : kk == 1 ? x : kk == 2 ? y : x * x;
        phi2[Gs] = kk == 0 ? x * y : kk == 1 ? y * y : 0.f;
    }
    const int myq = lane;
    const int lx = wx + qx(myq), ly = wy + qy(myq);
    float *stage = s_acc[w];
    float gpix[CH];
    {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        const size_t HW = (size_t)A.H * A.W;
        const bool inside = (px < A.W) && (py < A.H);
        const size_t pix = (size_t)A.W * (size_t)py + px;
        const float Tf = inside ? A.final_T[pix] : 0.f;
        const int last = inside ? A.ncontrib[pix] : 0;
        float bgdot = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float g = (inside && k < cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
            gpix[k] = g;
            bgdot += A.bg * g;
        }
        float *r = s_state[w] + pixoff(myq);
        r[0] = Tf;
        r[1] = Tf * bgdot;
        s_last[w][myq] = last;
        const int wmax = wave_max_i(last);
        if (lane == 0) s_wmax[w] = wmax;
    }
    if (tid < RQ) s_rec[qpart(SB, tid)] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 2) s_coef[2 * SB + tid] = make_float4(tid == 0 ? -__builtin_inff() : 0.f, 0.f, 0.f, 0.f);   // inert slot: q0 = log2(0)
    float hcg[4][NK], hft[16][NA];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int k = 0; k < CH; ++k) stage[(lane & 31) * CH + k] = gpix[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int G = 2 * h; G < 2 * h + 2; ++G)
#pragma unroll
            for (int j = 0; j < NK; ++j) hcg[G][j] = stage[(16 * (G & 1) + nl) * CH + 4 * j + kk];
#pragma unroll
        for (int st = 8 * h; st < 8 * h + 8; ++st) {
#pragma unroll
            for (int q = 0; q < NA; ++q)
                hft[st][q] = 16 * q + nl < CH ? stage[(16 * ((st >> 2) & 1) + 4 * kk + (st & 3)) * CH + 16 * q + nl] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < RW) s_acc[w][CAP * RW + lane] = 0.f;   // the slab's zero row
    __syncthreads();
    const float *momrow = s_mom + 8 * kk + (nl & 7);
    const int2 range = A.tile_range[tile];
    const int len = range.y - range.x;
    const int n = imin_(len, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    const int *slots = A.slot_sorted + range.x;
    const int ce = lane, cp = w;   // combine: entry = lane, role = wave (geometry | three shares of the feature chunks): wave-uniform paths
    auto zero_rec = [&](int slot) {
        float *dst = pair_buf + (size_t)slot * RST;
        for (int c = cp; c < NCP / 4; c += 4) reinterpret_cast<float4 *>(dst)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    for (int ql = n + ce; ql < len; ql += 64) zero_rec(slots[ql]);
    float *state = s_state[w] + kk * KS;
    const int *lastp = s_last[w] + 4 * kk;
    if (n <= 0) {
        if (A.dbg_T_front) {
            const int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_state[w][pixoff(myq)];
        }
        return;
    }
    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };
    // staging: a quad per entry (see blend_bwd_sets_quarter_kernel) -- thread t <-> entry se = t >> 2, part sp = t & 3 of each of the
    // record's NSEC 64-byte sectors: one index load per thread and super-batch, no division, the coefficient block once per wave
    constexpr int NSEC = RQ / 4;
    const int se = tid >> 2, sp = tid & 3;
    int sid_next;
    float4 sv[NSEC];
#define WIDEQ_STAGE_IDS(b)                                                        \
    do {                                                                          \
        const int q_ = pos(se, (b));                                              \
        sid_next = q_ >= 0 ? A.idx_sorted[range.x + q_] : -1;                     \
    } while (0)
    // (past the list the quad loads Gaussian 0's record: staged entries >= nb are in no list and the combine skips them)
#define WIDEQ_STAGE_PAYLOAD()                                                                                                   \
    do {                                                                                                                        \
        const float4 *src_ = reinterpret_cast<const float4 *>(A.pack + (size_t)imax_(sid_next, 0) * Rec<CH>::RS) + sp;          \
        _Pragma("unroll") for (int k_ = 0; k_ < NSEC; ++k_) sv[k_] = src_[4 * k_];                                              \
    } while (0)
    WIDEQ_STAGE_IDS(0);
    WIDEQ_STAGE_PAYLOAD();
    WIDEQ_STAGE_IDS(1);
    auto load_flags = [&](int topb) -> unsigned {
        const int q = topb - tid;
        return (tid < SB && q >= 0) ? (unsigned)A.cull_flags[range.x + q] : 0u;
    };
    unsigned fl_next = load_flags(n - 1);

    int batch = 0;
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        {   // park; the lanes holding parts 0 / 1 of an entry exchange them and leave the polynomial's coefficients
            const float4 mine = sv[0];
            float4 other;
            other.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.x), 0xB1, 0xf, 0xf, true));
            other.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.y), 0xB1, 0xf, 0xf, true));
            other.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.z), 0xB1, 0xf, 0xf, true));
            other.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.w), 0xB1, 0xf, 0xf, true));
            const float4 g0 = (sp & 1) == 0 ? mine : other, g1 = (sp & 1) == 0 ? other : mine;   // (lanes 2, 3 of a quad: unused)
            const PowerCoef pc = power_coeffs(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, tcx, tcy);
            float4 *rb = s_rec + se * RQ + (sp ^ ((se >> 2) & 3));   // part 4 k + sp at 4 k + (sp ^ swizzle)
#pragma unroll
            for (int k = 0; k < NSEC; ++k) rb[4 * k] = sv[k];
            if (sp == 0) s_coef[2 * se] = make_float4(pc.q0, pc.qx, pc.qy, pc.qxx);
            else if (sp == 1) s_coef[2 * se + 1] = make_float4(pc.qxy, pc.qyy, 0.f, 0.f);
        }
        const unsigned fl = fl_next;
        fl_next = load_flags(top - SB);
        if (tid < SB) {   // byte w' bit G = quarter w' of block G in the forward's cull flags (byte G bit w')
            unsigned kw = 0u;
            if (tid < nb) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned t4 = (fl >> ww) & 0x01010101u;
                    const unsigned nib = ((t4 * 0x01020408u) >> 24) & 0xfu;
                    if (top - tid < s_wmax[ww]) kw |= nib << (8 * ww);
                }
            }
            s_keep[tid] = kw;
        }
        __syncthreads();
        float *slab = s_acc[w];
        unsigned cqw = 0u;
        int cnt;
        {
            const int e = lane;
            const unsigned bits = (s_keep[e] >> (8 * w)) & 0xfu;
            const bool kb = bits != 0u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(kb);
            const int ps = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            reinterpret_cast<unsigned char *>(s_pos4)[4 * e + w] = kb ? (unsigned char)ps : (unsigned char)255;
            const unsigned word = (bits * 0x00204081u) & 0x01010101u;
            unsigned incl = word;
            asm volatile("s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "s_nop 1\n\t"
                         "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                         : "+v"(incl));
            const unsigned posw = incl - word;
            const unsigned short ent = (unsigned short)(e | (ps << 8));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((bits >> q) & 1u) s_qlist[w][q][(posw >> (8 * q)) & 0xffu] = ent;
            cqw = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            cnt = __popcll(m);
        }
        int cq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cq[q] = (int)((cqw >> (8 * q)) & 0xffu);
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_qlist[w][q][cq[q] + lane] = (unsigned short)(SB | (CAP << 8));
        }
        {
            float4 *z = reinterpret_cast<float4 *>(slab);
            for (int c = lane; c < cnt * (RW / 4); c += WAVE) z[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            for (int j0 = 0; j0 < cq[G]; j0 += 16) {
                const unsigned le = s_qlist[w][G][j0 + nl];
                const int e = le & 0xffu, row = le >> 8;
                const int qn = top - e;
                const float *er = reinterpret_cast<const float *>(s_rec) + e * (4 * RQ);
                const float *ec = reinterpret_cast<const float *>(s_coef) + e * 8;
                const int sw = 4 * ((e >> 2) & 3);
                const float bq1 = ec[kk];
                const float bq2 = ec[4 + (kk & 1)];
                float bf[NK];
#pragma unroll
                for (int j = 0; j < NK; ++j) bf[j] = er[(8 + 4 * j + kk) ^ sw];
                // the survivor's slab row is requested now (a padded list slot reads the zero row): its round trip runs under the step
                // (not at 20 channels: the 168-register budget of three waves per SIMD has no room for the 12 registers)
                constexpr bool EARLY = CH != 20;
                float *const rr = slab + row * RW;
                float4 *const p1 = reinterpret_cast<float4 *>(rr + 4 * (kk & 1));
                float4 o1_in, fq_in[NA];
                if (EARLY) {
                    o1_in = *p1;
#pragma unroll
                    for (int q = 0; q < NA; ++q)
                        fq_in[q] = *reinterpret_cast<const float4 *>(rr + 8 + ((16 * q + 4 * kk < CH) ? 16 * q + 4 * kk : 0));
                }
                f32x4 d_mom = {0.f, 0.f, 0.f, 0.f};
                f32x4 d_f[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) d_f[a] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" ::: "memory");
                f32x4 pw = {0.f, 0.f, 0.f, 0.f}, cva = {0.f, 0.f, 0.f, 0.f}, cvb = {0.f, 0.f, 0.f, 0.f};
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi1[G], bq1, pw, 0, 0, 0);
                pw = __builtin_amdgcn_mfma_f32_16x16x4f32(phi2[G], bq2, pw, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NK; j += 2) {   // two accumulation chains
                    cva = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][j], bf[j], cva, 0, 0, 0);
                    if (j + 1 < NK) cvb = __builtin_amdgcn_mfma_f32_16x16x4f32(hcg[G][j + 1], bf[j + 1], cvb, 0, 0, 0);
                }
                float cg[4], araw[4], a[4], r1a[4], rp[4], Ts4[4], Rs4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int last = lastp[16 * G + i];
                    const float2 sv = *reinterpret_cast<const float2 *>(state + G * GS + i * 2);
                    Ts4[i] = sv.x;
                    Rs4[i] = sv.y;
                    cg[i] = cva[i] + cvb[i];
                    bool pw_ok;
                    araw[i] = exp2_guard(pw[i], pw_ok);
                    const bool ok = (qn < last) && pw_ok && !(araw[i] < (1.0f / 255.0f));
                    araw[i] = ok ? araw[i] : 0.f;
                    a[i] = fminf(0.99f, araw[i]);
                    r1a[i] = __builtin_amdgcn_rcpf(1.f - a[i]);
                    rp[i] = r1a[i];
                }
                row_scan_mul4(rp[0], rp[1], rp[2], rp[3]);
                float T[4], wgt[4], rs[4], R[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T[i] = Ts4[i] * rp[i];
                    wgt[i] = a[i] * T[i];
                    rs[i] = cg[i] * wgt[i];
                }
                row_scan_add4(rs[0], rs[1], rs[2], rs[3]);
                row_shr1_add4(R, rs, Rs4);
                lds_store2x4_lane15<0, 2, 4, 6>(state + G * GS, T[0], Rs4[0] + rs[0], T[1], Rs4[1] + rs[1], T[2], Rs4[2] + rs[2], T[3],
                                                Rs4[3] + rs[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = 4 * G + i;
                    const float dLa = T[i] * cg[i] - R[i] * r1a[i];
                    const float dLp = araw[i] * dLa;
                    d_mom = __builtin_amdgcn_mfma_f32_16x16x4f32(momrow[32 * s], dLp, d_mom, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NA; ++q) d_f[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(hft[s][q], wgt[i], d_f[q], 0, 0, 0);
                }
                if (j0 + nl < cq[G]) {
                    if (kk < 2) {
                        float4 o1 = EARLY ? o1_in : *p1;
                        o1.x += d_mom[0]; o1.y += d_mom[1]; o1.z += d_mom[2]; o1.w += d_mom[3];
                        *p1 = o1;
                    }
#pragma unroll
                    for (int q = 0; q < NA; ++q) {
                        if (16 * q + 4 * kk < CH) {
                            float4 f0 = EARLY ? fq_in[q] : *reinterpret_cast<const float4 *>(rr + 8 + 16 * q + 4 * kk);
                            f0.x += d_f[q][0]; f0.y += d_f[q][1]; f0.z += d_f[q][2]; f0.w += d_f[q][3];
                            *reinterpret_cast<float4 *>(rr + 8 + 16 * q + 4 * kk) = f0;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        const int slot_mine = ce < nb ? slots[top - ce] : 0;
        WIDEQ_STAGE_PAYLOAD();
        WIDEQ_STAGE_IDS(batch + 2);
        // the combine's reads of the staged records, in front of its barrier: behind it a fast wave may already park the next
        // super-batch (two barriers per super-batch, see blend_bwd_sets_quarter_kernel)
        float4 cg0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float cgC = 0.f, cgo = 0.f;
        if (cp == 0) {
            cg0 = s_rec[qpart(ce, 0)];
            const float4 t_ = s_rec[qpart(ce, 1)];
            cgC = t_.x; cgo = t_.y;
        }
        __syncthreads();
        if (ce < nb) {
            const int e = ce;
            const unsigned int p4 = s_pos4[e];
            float *dst = pair_buf + (size_t)slot_mine * RST;
            if (cp == 0) {
                float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                    const float4 *rw = reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW);
                    const float4 m = rw[0], m2 = rw[1];
                    const float bxw = (ww & 1) ? 2.f : -2.f, byw = (ww >> 1) ? 2.f : -2.f;
                    s[0] += m.x;
                    s[1] += m.y + bxw * m.x;
                    s[2] += m.z + byw * m.x;
                    s[3] += m.w + 2.f * bxw * m.y + (bxw * bxw) * m.x;
                    s[4] += m2.x + bxw * m.z + byw * m.y + (bxw * byw) * m.x;
                    s[5] += m2.y + 2.f * byw * m.z + (byw * byw) * m.x;
                }
                const float4 g0 = cg0;
                const float cA = g0.z, cB = g0.w, cC = cgC, o = cgo;
                const float uc = g0.x - tcx, vc = g0.y - tcy;
                const float M0 = s[0], Mx = s[1], My = s[2], Mxx = s[3], Mxy = s[4], Myy = s[5];
                float4 r0;
                r0.x = cA * Mx + cB * My - (cA * uc + cB * vc) * M0;
                r0.y = cB * Mx + cC * My - (cB * uc + cC * vc) * M0;
                r0.z = -0.5f * (uc * uc * M0 - 2.f * uc * Mx + Mxx);
                r0.w = -(uc * vc * M0 - uc * My - vc * Mx + Mxy);
                reinterpret_cast<float4 *>(dst)[0] = r0;
                float2 r1;
                r1.x = -0.5f * (vc * vc * M0 - 2.f * vc * My + Myy);
                r1.y = o > 0.f ? M0 / o : 0.f;
                reinterpret_cast<float2 *>(dst)[2] = r1;
            } else {
                constexpr int NQF = CH / 4, PER = (NQF + 2) / 3;   // feature float4 chunks, shared by three threads
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int c = (cp - 1) * PER + k;
                    if (c < NQF) {
                        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            const unsigned int pp = umin_((p4 >> (8 * ww)) & 0xffu, (unsigned)CAP);
                            const float4 v = *reinterpret_cast<const float4 *>(s_acc[ww] + pp * RW + 8 + 4 * c);
                            f.x += v.x; f.y += v.y; f.z += v.z; f.w += v.w;
                        }
                        float2 *d2 = reinterpret_cast<float2 *>(dst + NG + 4 * c);   // (NG = 6: 8-byte aligned)
                        d2[0] = make_float2(f.x, f.y);
                        d2[1] = make_float2(f.z, f.w);
                    }
                }
            }
        }
    }
    if (A.dbg_T_front) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px < A.W && py < A.H) A.dbg_T_front[(size_t)A.W * py + px] = s_state[w][pixoff(myq)];
    }
}
#undef WIDEQ_STAGE_IDS
#undef WIDEQ_STAGE_PAYLOAD

// ------------------------------------------------------------------ backward, atomic mode (foreign idx_sorted)
// Same tile structure; every wave reduces its partials and lane 63 issues one hardware float atomic
// per (wave, splat, component).  Gradient outputs must be zero-initialised.
template <int CH, bool ABS, bool BIAS, bool EXACT>
__global__ void __launch_bounds__(256)
blend_bwd_atomic_kernel(const BlendArgs B) {
    using GL = GradLayout<ABS, BIAS>;
    constexpr int SB = 64, NG = GL::NG, NC = NG + CH;
    __shared__ TileLDS<CH, SB, !BIAS> L;
    __shared__ int s_wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int frame = gtile / B.T, tile = gtile - frame * B.T;
    const BlendArgs A = frame_args(B, frame);
    const int tx = tile % A.gx, ty = tile / A.gx;
    const int bx = tx * TILE + (w & 1) * 8, by = ty * TILE + (w >> 1) * 8;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float x = (float)((w & 1) * 8 + (lane & 7)) - 7.5f, y = (float)((w >> 1) * 8 + (lane >> 3)) - 7.5f;
    const float xx = x * x, xy = x * y, yy = y * y;
    const size_t HW = (size_t)A.H * A.W;
    const int cn = EXACT ? CH : A.cn;

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
        gp[k] = (inside && k < cn) ? A.dL_dout[(size_t)(A.c0 + k) * HW + pix] : 0.f;
        if (k < cn) bgdot += A.bg * gp[k];
    }
    const int wmax = wave_max_i(last);
    if (lane == 0) s_wmax[w] = wmax;
    if (tid < Rec<CH>::RQ) L.rec[SB * Rec<CH>::RQ + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // inert slot SB
    if (!BIAS && tid < 2) L.coef[2 * SB + tid] = make_float4(tid == 0 ? -__builtin_inff() : 0.f, 0.f, 0.f, 0.f);  // inert entry: q0 = log2(0)
    __syncthreads();
    const int2 range = A.tile_range[tile];
    const int n = imin_(range.y - range.x, imax_(imax_(s_wmax[0], s_wmax[1]), imax_(s_wmax[2], s_wmax[3])));
    if (n <= 0) {
        if (A.dbg_T_front && inside) A.dbg_T_front[pix] = T;
        return;
    }
    auto pos = [n](int e, int b) { return n - 1 - b * SB - e; };
    Stager<CH, SB> st;
    st.load_ids(A, tid, range.x, pos, 0);
    st.load_payload(A, tid);
    st.load_ids(A, tid, range.x, pos, 1);
    int batch = 0;
    for (int top = n - 1; top >= 0; top -= SB, ++batch) {
        const int nb = imin_(SB, top + 1);
        st.park(L, tid);
        st.load_payload(A, tid);
        st.load_ids(A, tid, range.x, pos, batch + 2);
        __syncthreads();
        tile_cull<CH, SB, BIAS, false, !BIAS>(L, tid, nb, (float)(tx * TILE), (float)(ty * TILE),
                                [&](int e, int ww) { return top - e < s_wmax[ww]; });
        __syncthreads();
        const int cnt = build_list(L, w, lane);
        for (int j = 0; j < cnt; ++j) {
            const int e = L.list[w][j];
            const float4 g0 = L.g0(e), g1 = L.g1(e);
            const float dx = g0.x - pxf, dy = g0.y - pyf;
            bool pw_ok;
            float G, araw;
            if (BIAS) {  // the forward's expressions (blend_fwd_kernel), bit for bit
                const float q = neg_power_factored(dx, dy, g0.z, g0.w, g1.x);
                G = exp_neg(q);
                araw = __builtin_fmaf(g1.y, G, g1.z);
                pw_ok = !(q < 0.f);
            } else {
                const float pw = power_poly(L.coef[2 * e], L.coef[2 * e + 1], x, y, xx, xy, yy);
                araw = exp2_guard(pw, pw_ok);                          // = o * G (the forward's value, bit for bit)
                G = araw * __builtin_amdgcn_rcpf(g1.y);
            }
            const float alpha = fminf(0.99f, araw);
            const bool ok = !done && (top - e < last) && pw_ok && !(alpha < (1.0f / 255.0f));
            if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
            float f[CH];
            read_feat(L, e, f);
            float r[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) r[k] = 0.f;
            if (ok) replay_one<CH, ABS, BIAS>(g0, g1, f, dx, dy, G, alpha, Tf, bgdot, gp, T, acc, done, r);
            wave_sum_n_to_lane63<NC>(r);
            if (lane == 63) {
                const int id = __float_as_int(g1.w);
                atomic_add_f32(A.dL_duv + 2 * id, r[0]);
                atomic_add_f32(A.dL_duv + 2 * id + 1, r[1]);
                atomic_add_f32(A.dL_dconic + 3 * id, r[2]);
                atomic_add_f32(A.dL_dconic + 3 * id + 1, r[3]);
                atomic_add_f32(A.dL_dconic + 3 * id + 2, r[4]);
                atomic_add_f32(A.dL_dopacity + id, r[5]);
                if (ABS) {
                    atomic_add_f32(A.dL_dabs_uv + 2 * id, r[GL::I_ABS]);
                    atomic_add_f32(A.dL_dabs_uv + 2 * id + 1, r[GL::I_ABS + 1]);
                }
                if (BIAS) atomic_add_f32(A.dL_dbias + id, r[GL::I_BIAS]);
                float *df = A.dL_dfeature + (size_t)id * A.C + A.c0;
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k < cn) atomic_add_f32(df + k, r[NG + k]);
            }
        }
        __syncthreads();
    }
    if (A.dbg_T_front && inside) A.dbg_T_front[pix] = T;
}

// ================================================================== launch tables
template <int CH>
static int launch_pack(const BlendArgs &A, bool bias, hipStream_t s) {
    if (A.P == 0) return SPLAT_OK;
    const dim3 grid((unsigned)((A.P + 255) / 256), (unsigned)A.F), block(256);
    const bool exact = A.cn == CH;
    if (bias) { if (exact) SPLAT_LAUNCH("blend_pack", (pack_kernel<CH, true, true>), grid, block, 0, s, A); else SPLAT_LAUNCH("blend_pack", (pack_kernel<CH, true, false>), grid, block, 0, s, A); }
    else { if (exact) SPLAT_LAUNCH("blend_pack", (pack_kernel<CH, false, true>), grid, block, 0, s, A); else SPLAT_LAUNCH("blend_pack", (pack_kernel<CH, false, false>), grid, block, 0, s, A); }
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

template <int CH>
static int launch_fwd(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    const dim3 grid((unsigned)(T * A.F)), block(256);
    const bool exact = A.cn == CH;
    {
        const int rc = launch_pack<CH>(A, bias, s);
        if (rc != SPLAT_OK) return rc;
    }
#define FWD(E, B, X) SPLAT_LAUNCH("blend_fwd", (blend_fwd_kernel<CH, E, B, X>), grid, block, 0, s, A)
    if (enh) {
        if (bias) { if (exact) FWD(true, true, true); else FWD(true, true, false); }
        else { if (exact) FWD(true, false, true); else FWD(true, false, false); }
    } else {
        if (bias) { if (exact) FWD(false, true, true); else FWD(false, true, false); }
        else { if (exact) FWD(false, false, true); else FWD(false, false, false); }
    }
#undef FWD
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// kernel selection: options of the ABI (splat_set_option), read at launch time
static bool bwd_use_mfma() { return splat_option(SPLAT_OPT_BWD_KERNEL_DPP) == 0; }        // 1: the DPP-reduction pair kernel (A/B)
static bool bwd_use_quarters() { return splat_option(SPLAT_OPT_BWD_QUARTERS) != 0; }      // 0: block-level kernels everywhere
static bool sets_std_plan_enabled() { return splat_option(SPLAT_OPT_SETS_STD) != 0; }     // 0: generic slot -> channel routing

template <int CH, bool ABS, bool BIAS>
static int launch_bwd_ab(const BlendArgs &A, int T, bool pair, hipStream_t s) {
    const dim3 grid((unsigned)(T * A.F)), block(256);
    const bool exact = A.cn == CH;
    if (pair && !BIAS && CH <= 3 && A.cull_flags && bwd_use_mfma() && bwd_use_quarters()) {
        // narrow row with the forward's cull words: one survivor list per 4x4 quarter
        constexpr int QC = CH <= 3 ? CH : 3;
        if (exact) SPLAT_LAUNCH("blend_bwd", (blend_bwd_quarter_kernel<QC, ABS, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_bwd", (blend_bwd_quarter_kernel<QC, ABS, false>), grid, block, 0, s, A);
    } else if (pair && !BIAS && !ABS && CH >= 16 && A.cull_flags && !A.rec_stride && bwd_use_mfma() && bwd_use_quarters()) {
        // frame batch, wide row without |taps|: quarter lists
        constexpr int WC = CH >= 16 ? CH : 16;
        if (exact) SPLAT_LAUNCH("blend_bwd", (blend_bwd_wide_quarter_kernel<WC, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_bwd", (blend_bwd_wide_quarter_kernel<WC, false>), grid, block, 0, s, A);
    } else if (pair && !BIAS && bwd_use_mfma() && !(BLEND_CARRY && splat_deterministic())) {
        if (exact) SPLAT_LAUNCH("blend_bwd", (blend_bwd_mfma_kernel<CH, ABS, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_bwd", (blend_bwd_mfma_kernel<CH, ABS, false>), grid, block, 0, s, A);
    } else if (pair) {
        if (exact) SPLAT_LAUNCH("blend_bwd", (blend_bwd_pair_kernel<CH, ABS, BIAS, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_bwd", (blend_bwd_pair_kernel<CH, ABS, BIAS, false>), grid, block, 0, s, A);
    } else {
        if (splat_deterministic()) {
            splat_set_error("deterministic mode: the backward needs this library's pair map (goff_incl, slot_sorted, pair_scratch); "
                            "a foreign idx_sorted would take the float-atomic kernel");
            return SPLAT_E_ARG;
        }
        if (exact) SPLAT_LAUNCH("blend_bwd", (blend_bwd_atomic_kernel<CH, ABS, BIAS, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_bwd", (blend_bwd_atomic_kernel<CH, ABS, BIAS, false>), grid, block, 0, s, A);
    }
    SPLAT_POST_LAUNCH();
    if (pair && !A.tile_only) {  // frame batches reduce their records in the Gaussian-side backward (preprocess.hip)
        const dim3 rgrid((unsigned)(((size_t)A.P * 4 + 255) / 256));
        SPLAT_LAUNCH("pair_reduce", (pair_reduce_kernel<ABS, BIAS, PairCfg<CH, ABS, BIAS>::NCP>), rgrid, dim3(256), 0, s, A);
        SPLAT_POST_LAUNCH();
    }
    return SPLAT_OK;
}

template <int CH>
static int launch_bwd(const BlendArgs &A, int T, bool bias, bool pair, hipStream_t s) {
    if (!A.pack_valid) {
        const int rc = launch_pack<CH>(A, bias, s);
        if (rc != SPLAT_OK) return rc;
    }
    const bool abs_ = A.dL_dabs_uv != nullptr;  // |d uv| sums are only produced when the caller asks for them
    if (abs_) return bias ? launch_bwd_ab<CH, true, true>(A, T, pair, s) : launch_bwd_ab<CH, true, false>(A, T, pair, s);
    return bias ? launch_bwd_ab<CH, false, true>(A, T, pair, s) : launch_bwd_ab<CH, false, false>(A, T, pair, s);
}

// channel-chunk width -> kernel instantiation
static inline int chunk_ch(int cn) { return cn <= 1 ? 1 : cn <= 3 ? 3 : cn <= 8 ? 8 : cn <= 16 ? 16 : cn <= 20 ? 20 : cn <= 24 ? 24 : 32; }

static int fwd_chunk(const BlendArgs &A, int T, bool enh, bool bias, hipStream_t s) {
    switch (chunk_ch(A.cn)) {
        case 1: return launch_fwd<1>(A, T, enh, bias, s);
        case 3: return launch_fwd<3>(A, T, enh, bias, s);
        case 8: return launch_fwd<8>(A, T, enh, bias, s);
        case 16: return launch_fwd<16>(A, T, enh, bias, s);
        case 20: return launch_fwd<20>(A, T, enh, bias, s);
        case 24: return launch_fwd<24>(A, T, enh, bias, s);
        default: return launch_fwd<32>(A, T, enh, bias, s);
    }
}

static int bwd_chunk(const BlendArgs &A, int T, bool bias, bool pair, hipStream_t s) {
    switch (chunk_ch(A.cn)) {
        case 1: return launch_bwd<1>(A, T, bias, pair, s);
        case 3: return launch_bwd<3>(A, T, bias, pair, s);
        case 8: return launch_bwd<8>(A, T, bias, pair, s);
        case 16: return launch_bwd<16>(A, T, bias, pair, s);
        case 20: return launch_bwd<20>(A, T, bias, pair, s);
        case 24: return launch_bwd<24>(A, T, bias, pair, s);
        default: return launch_bwd<32>(A, T, bias, pair, s);
    }
}

extern "C" size_t splat_blend_pack_floats(int C) {
    // floats per packed Gaussian record for the widest channel chunk: [u v A B | C o bias id | features | pad]
    return (size_t)((8 + chunk_ch(C > 32 ? 32 : C) + 15) & ~15);
}

extern "C" size_t splat_blend_pair_floats(int C, int has_bias) {
    // floats per pair record (upper bound over the abs / no-abs layouts) for the widest channel chunk
    return (size_t)PAIR_STRIDE((has_bias ? 9 : 8) + chunk_ch(C > 32 ? 32 : C));
}

// ================================================================== C ABI
static int blend_forward_impl(int P, int C, const float *uv, const float *conic, const float *opacity,
                              const float *feature, const float *opacity_bias,
                              const int32_t *idx_sorted, const int32_t *tile_range, float bg,
                              const float *bg_channels, int W, int H, int K, int enable_truncation,
                              float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx,
                              float *pack_scratch, uint32_t *cull_flags, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && C >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(tile_range && out && final_T && ncontrib, "null pointer");
    SPLAT_CHECK_ARG(P == 0 || (uv && conic && opacity && feature && pack_scratch), "null pointer");
    const bool enh = (gs_idx != nullptr) && K > 0;
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity; A.feature = feature; A.bias = opacity_bias;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bg = bg; A.bgc = bg_channels; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.K = enh ? K : 0; A.trunc = enable_truncation ? 1 : 0;
    A.out = out; A.final_T = final_T; A.ncontrib = ncontrib; A.gs_idx = gs_idx;
    A.pack = pack_scratch;
    A.cull_flags = cull_flags;   // optional: the cull's keep words of every sorted entry, for the backward
    const int T = A.gx * ((H + TILE - 1) / TILE);
    A.F = 1; A.T = T;
    for (int c0 = 0; c0 < C; c0 += 32) {  // reference chunking: src/alpha_blending.cu:287-394
        A.c0 = c0;
        A.cn = C - c0 > 32 ? 32 : C - c0;
        const int rc = fwd_chunk(A, T, enh, opacity_bias != nullptr, (hipStream_t)stream);
        if (rc != SPLAT_OK) return rc;
    }
    return SPLAT_OK;
}

extern "C" int splat_alpha_blending_forward(int P, int C, const float *uv, const float *conic, const float *opacity,
                                            const float *feature, const float *opacity_bias,
                                            const int32_t *idx_sorted, const int32_t *tile_range, float bg,
                                            const float *bg_channels, int W, int H, int K, int enable_truncation,
                                            float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx,
                                            float *pack_scratch, splat_stream_t stream) {
    return blend_forward_impl(P, C, uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, bg_channels, W, H, K,
                              enable_truncation, out, final_T, ncontrib, gs_idx, pack_scratch, nullptr, stream);
}

extern "C" int splat_alpha_blending_forward_flags(int P, int C, const float *uv, const float *conic, const float *opacity,
                                                  const float *feature, const float *opacity_bias,
                                                  const int32_t *idx_sorted, const int32_t *tile_range, float bg,
                                                  const float *bg_channels, int W, int H, int K, int enable_truncation,
                                                  float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx,
                                                  float *pack_scratch, uint32_t *cull_flags, splat_stream_t stream) {
    return blend_forward_impl(P, C, uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, bg_channels, W, H, K,
                              enable_truncation, out, final_T, ncontrib, gs_idx, pack_scratch, cull_flags, stream);
}

static int blend_backward_impl(int P, int C, const float *uv, const float *conic, const float *opacity,
                               const float *feature, const float *opacity_bias,
                               const int32_t *idx_sorted, const int32_t *tile_range, float bg, int W,
                               int H, const float *final_T, const int32_t *ncontrib,
                               const float *dL_dout, float *dL_duv, float *dL_dabs_uv, float *dL_dconic,
                               float *dL_dopacity, float *dL_dfeature, float *dL_dopacity_bias,
                               float *dL_dndc, float *dL_dabs_ndc, const int32_t *goff_incl,
                               const int32_t *slot_sorted, float *pair_scratch, float *pack_scratch,
                               int pack_is_valid, float *dbg_T_front, const uint32_t *cull_flags, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && C >= 1 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    // idx_sorted may be NULL when no Gaussian touches any tile (M = 0: every tile range is empty, nothing dereferences it)
    SPLAT_CHECK_ARG(uv && conic && opacity && feature && tile_range && final_T && ncontrib && dL_dout && pack_scratch,
                    "null pointer");
    SPLAT_CHECK_ARG(dL_duv && dL_dconic && dL_dopacity && dL_dfeature, "null gradient pointer");
    SPLAT_CHECK_ARG(!opacity_bias || dL_dopacity_bias, "bias given without dL_dopacity_bias");
    const int npm = (goff_incl != nullptr) + (slot_sorted != nullptr) + (pair_scratch != nullptr);
    SPLAT_CHECK_ARG(npm == 0 || npm == 3, "goff_incl, slot_sorted and pair_scratch go together");
    const bool pair_mode = npm == 3;
    if (!pair_mode && splat_deterministic()) {   // refused BEFORE any launch (the packing kernel used to run first)
        splat_set_error("deterministic mode: the backward needs this library's pair map (goff_incl, slot_sorted, pair_scratch); "
                        "a foreign idx_sorted would take the float-atomic kernel");
        return SPLAT_E_ARG;
    }
    SPLAT_CHECK_ARG(pair_mode || (!dL_dndc && !dL_dabs_ndc), "the tap outputs need the pair-mode backward");
    SPLAT_CHECK_ARG(!dL_dabs_ndc || dL_dabs_uv, "dL_dabs_ndc needs dL_dabs_uv");
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
    A.dL_dndc = dL_dndc; A.dL_dabs_ndc = dL_dabs_ndc;
    A.goff_incl = goff_incl; A.slot_sorted = slot_sorted; A.pair_buf = pair_scratch;
    A.pack = pack_scratch;
    A.dbg_T_front = dbg_T_front;
    A.cull_flags = pair_mode ? const_cast<uint32_t *>(cull_flags) : nullptr;   // (the forward's keep words: quarter-list kernels)
    A.pack_valid = (pack_is_valid && C <= 32) ? 1 : 0;  // one chunk only: later chunks overwrite the scratch
    const int T = A.gx * ((H + TILE - 1) / TILE);
    A.F = 1; A.T = T;
    for (int c0 = 0; c0 < C; c0 += 32) {  // reference chunking: src/alpha_blending.cu:440-577
        A.c0 = c0;
        A.cn = C - c0 > 32 ? 32 : C - c0;
        A.accumulate = c0 > 0;
        const int rc = bwd_chunk(A, T, opacity_bias != nullptr, pair_mode, (hipStream_t)stream);
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
                                             float *dL_dndc, float *dL_dabs_ndc, const int32_t *goff_incl,
                                             const int32_t *slot_sorted, float *pair_scratch, float *pack_scratch,
                                             int pack_is_valid, float *dbg_T_front, splat_stream_t stream) {
    return blend_backward_impl(P, C, uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, W, H, final_T, ncontrib,
                               dL_dout, dL_duv, dL_dabs_uv, dL_dconic, dL_dopacity, dL_dfeature, dL_dopacity_bias, dL_dndc,
                               dL_dabs_ndc, goff_incl, slot_sorted, pair_scratch, pack_scratch, pack_is_valid, dbg_T_front, nullptr,
                               stream);
}

extern "C" int splat_alpha_blending_backward_flags(int P, int C, const float *uv, const float *conic, const float *opacity,
                                                   const float *feature, const float *opacity_bias,
                                                   const int32_t *idx_sorted, const int32_t *tile_range, float bg, int W,
                                                   int H, const float *final_T, const int32_t *ncontrib,
                                                   const float *dL_dout, float *dL_duv, float *dL_dabs_uv, float *dL_dconic,
                                                   float *dL_dopacity, float *dL_dfeature, float *dL_dopacity_bias,
                                                   float *dL_dndc, float *dL_dabs_ndc, const int32_t *goff_incl,
                                                   const int32_t *slot_sorted, float *pair_scratch, float *pack_scratch,
                                                   int pack_is_valid, float *dbg_T_front, const uint32_t *cull_flags,
                                                   splat_stream_t stream) {
    return blend_backward_impl(P, C, uv, conic, opacity, feature, opacity_bias, idx_sorted, tile_range, bg, W, H, final_T, ncontrib,
                               dL_dout, dL_duv, dL_dabs_uv, dL_dconic, dL_dopacity, dL_dfeature, dL_dopacity_bias, dL_dndc,
                               dL_dabs_ndc, goff_incl, slot_sorted, pair_scratch, pack_scratch, pack_is_valid, dbg_T_front,
                               cull_flags, stream);
}


// ================================================================== frame batch (F frames of one Gaussian set per launch)
// The tile kernels take workgroup b -> (frame, tile) through the same XCD-aware mapping over the F * T tiles; per-frame
// arrays are stacked [F, ...], the pair arrays (idx_sorted, slot_sorted, pair records) hold `capacity` entries per frame.
// C <= 32 (one channel chunk).  opacity / feature may be shared by all frames (stride 0) or per frame.
extern "C" size_t splat_blend_pair_stride(int C, int want_abs, int has_bias) {
    // exact record stride (floats) the backward kernels use for this configuration
    const int ng = 6 + (want_abs ? 2 : 0) + (has_bias ? 1 : 0);
    return (size_t)PAIR_STRIDE(ng + chunk_ch(C > 32 ? 32 : C));
}

extern "C" int splat_alpha_blending_forward_batch(int F, int P, int C, const float *uv, const float *conic,
                                                  const float *opacity, int64_t opacity_frame_stride,
                                                  const float *feature, int64_t feature_frame_stride,
                                                  const int32_t *idx_sorted, const int32_t *tile_range, int64_t capacity,
                                                  float bg, const float *bg_channels, int W, int H, int K,
                                                  int enable_truncation, float *out, float *final_T, int32_t *ncontrib,
                                                  int32_t *gs_idx, float *pack_scratch, uint32_t *cull_flags,
                                                  splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= 32 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 32)");
    SPLAT_CHECK_ARG(uv && conic && opacity && feature && idx_sorted && tile_range && out && final_T && ncontrib &&
                        pack_scratch,
                    "null pointer");
    const bool enh = (gs_idx != nullptr) && K > 0;
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity; A.feature = feature;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bg = bg; A.bgc = bg_channels; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.K = enh ? K : 0; A.trunc = enable_truncation ? 1 : 0;
    A.out = out; A.final_T = final_T; A.ncontrib = ncontrib; A.gs_idx = gs_idx;
    A.pack = pack_scratch;
    A.cull_flags = cull_flags;
    const int T = A.gx * ((H + TILE - 1) / TILE);
    SPLAT_CHECK_ARG((long long)F * T < (1ll << 31), "too many tiles");
    A.F = F; A.T = T; A.cap = capacity;
    A.pack_fs = (long long)P * (long long)splat_blend_pack_floats(C);
    A.opacity_fs = opacity_frame_stride; A.feature_fs = feature_frame_stride;
    A.c0 = 0; A.cn = C;
    return fwd_chunk(A, T, enh, false, (hipStream_t)stream);
}

// splat_alpha_blending_forward_batch over a row whose channels come from several SOURCES (no concatenated [F,P,C] row):
// source s = row channels [c0, c0 + cn) from dense rows feature[P, cn] (frame f at feature + f * frame_stride floats; 0 = shared
// by the frames -- a per-frame source is e.g. the depth [F,P,1] or track_gs = position(ids2) [F,P,3],
// src/trainer_fragGS.py:506-511).  The sources must tile the row; d_feature is not used by the forward.
extern "C" int splat_alpha_blending_forward_batch_sources(int F, int P, int C, int nsrc, const splat_feature_source_t *src,
                                                          const float *uv, const float *conic, const float *opacity,
                                                          int64_t opacity_frame_stride, const int32_t *idx_sorted,
                                                          const int32_t *tile_range, int64_t capacity,
                                                          const float *bg_channels, int W, int H, int K, int enable_truncation,
                                                          float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx,
                                                          float *pack_scratch, uint32_t *cull_flags, splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= 32 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 32)");
    SPLAT_CHECK_ARG(src && nsrc >= 1 && nsrc <= SPLAT_MAX_SOURCES && bg_channels, "null / oversized source table");
    SPLAT_CHECK_ARG(uv && conic && opacity && idx_sorted && tile_range && out && final_T && ncontrib && pack_scratch, "null pointer");
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    {
        unsigned long long covered = 0ull;
        for (int g = 0; g < nsrc; ++g) {
            SPLAT_CHECK_ARG(src[g].cn >= 0 && (src[g].cn == 0 || (src[g].c0 >= 0 && src[g].c0 + src[g].cn <= C && src[g].feature)),
                            "source outside the row / null source pointer");
            for (int k = 0; k < src[g].cn; ++k) {
                SPLAT_CHECK_ARG(!(covered & (1ull << (src[g].c0 + k))), "the sources overlap");
                covered |= 1ull << (src[g].c0 + k);
            }
            A.src_c0[g] = src[g].c0; A.src_cn[g] = src[g].cn; A.src_f[g] = src[g].feature; A.src_fs[g] = src[g].frame_stride;
        }
        SPLAT_CHECK_ARG(covered == (1ull << C) - 1ull, "the sources do not cover the row's channels");
    }
    A.nsrc = nsrc;
    const bool enh = (gs_idx != nullptr) && K > 0;
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bgc = bg_channels; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.K = enh ? K : 0; A.trunc = enable_truncation ? 1 : 0;
    A.out = out; A.final_T = final_T; A.ncontrib = ncontrib; A.gs_idx = gs_idx;
    A.pack = pack_scratch;
    A.cull_flags = cull_flags;
    const int T = A.gx * ((H + TILE - 1) / TILE);
    SPLAT_CHECK_ARG((long long)F * T < (1ll << 31), "too many tiles");
    A.F = F; A.T = T; A.cap = capacity;
    A.pack_fs = (long long)P * (long long)splat_blend_pack_floats(C);
    A.opacity_fs = opacity_frame_stride;
    A.c0 = 0; A.cn = C;
    return fwd_chunk(A, T, enh, false, (hipStream_t)stream);
}

// The same with (at most) three sets given as parallel HOST arrays of three entries (set_cn[g] = 0: no such set).
extern "C" int splat_alpha_blending_forward_batch_sets(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                                       const float *const *set_feature, const int64_t *set_feature_fs,
                                                       const float *uv, const float *conic, const float *opacity,
                                                       int64_t opacity_frame_stride, const int32_t *idx_sorted,
                                                       const int32_t *tile_range, int64_t capacity,
                                                       const float *bg_channels, int W, int H, int K, int enable_truncation,
                                                       float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx,
                                                       float *pack_scratch, uint32_t *cull_flags, splat_stream_t stream) {
    SPLAT_CHECK_ARG(set_c0 && set_cn && set_feature && set_feature_fs, "null set table");
    splat_feature_source_t src[3];
    memset(src, 0, sizeof(src));
    for (int g = 0; g < 3; ++g) {
        src[g].c0 = set_c0[g]; src[g].cn = set_cn[g]; src[g].feature = set_feature[g]; src[g].frame_stride = set_feature_fs[g];
    }
    return splat_alpha_blending_forward_batch_sources(F, P, C, 3, src, uv, conic, opacity, opacity_frame_stride, idx_sorted,
                                                      tile_range, capacity, bg_channels, W, H, K, enable_truncation, out, final_T,
                                                      ncontrib, gs_idx, pack_scratch, cull_flags, stream);
}

extern "C" int splat_alpha_blending_backward_batch(int F, int P, int C, const int32_t *idx_sorted,
                                                   const int32_t *tile_range, int64_t capacity, float bg, int W, int H,
                                                   const float *final_T, const int32_t *ncontrib, const float *dL_dout,
                                                   int want_abs, const int32_t *slot_sorted, float *pair_records,
                                                   const float *pack, const uint32_t *cull_flags, float *dbg_T_front,
                                                   splat_stream_t stream) {
    // writes one gradient record per (frame, tile, splat) pair at frame * capacity + slot; the records are summed per
    // Gaussian by the Gaussian-side backward (splat_frames_gauss_backward_*).  `pack` = the forward's packed records.
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= 32 && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 32)");
    SPLAT_CHECK_ARG(idx_sorted && tile_range && final_T && ncontrib && dL_dout && slot_sorted && pair_records && pack,
                    "null pointer");
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bg = bg; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.final_T = const_cast<float *>(final_T); A.ncontrib = const_cast<int *>(ncontrib);
    A.dL_dout = dL_dout;
    A.slot_sorted = slot_sorted; A.pair_buf = pair_records;
    A.pack = const_cast<float *>(pack); A.pack_valid = 1;
    A.dbg_T_front = dbg_T_front;
    A.cull_flags = const_cast<uint32_t *>(cull_flags);
    const int T = A.gx * ((H + TILE - 1) / TILE);
    A.F = F; A.T = T; A.cap = capacity; A.tile_only = 1;
    A.pack_fs = (long long)P * (long long)splat_blend_pack_floats(C);
    A.c0 = 0; A.cn = C;
    // tile kernels only: the abs sums are requested through a non-null marker (nothing is written through it here)
    A.dL_dabs_uv = want_abs ? pair_records : nullptr;
    return bwd_chunk(A, T, false, true, (hipStream_t)stream);
}

// One feature SET [c0, c0 + cn) of a wider composited row (the reference's renderer blends rgb / depth / attributes of one
// geometry, dptr_ortho_enhanced.py:331-375): packs its own records from (uv, conic, opacity, feature[.., c0:c0+cn]) and
// writes the set's pair records (stride splat_blend_pair_stride(cn, want_abs, 0)).  dL_dout is the [F,C,H,W] gradient of
// the whole row.
extern "C" int splat_alpha_blending_backward_batch_set(int F, int P, int C, int c0, int cn, const float *uv,
                                                       const float *conic, const float *opacity,
                                                       int64_t opacity_frame_stride, const float *feature,
                                                       int64_t feature_frame_stride, const int32_t *idx_sorted,
                                                       const int32_t *tile_range, int64_t capacity, float bg, int W, int H,
                                                       const float *final_T, const int32_t *ncontrib,
                                                       const float *dL_dout, int want_abs, const int32_t *slot_sorted,
                                                       float *pair_records, float *pack_scratch, float *dbg_T_front,
                                                       splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && W > 0 && H > 0 && capacity >= 1, "bad sizes");
    SPLAT_CHECK_ARG(c0 >= 0 && cn >= 1 && cn <= 32 && c0 + cn <= C, "the set must be 1..32 channels inside the row");
    SPLAT_CHECK_ARG(uv && conic && opacity && feature && idx_sorted && tile_range && final_T && ncontrib && dL_dout &&
                        slot_sorted && pair_records && pack_scratch,
                    "null pointer");
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity; A.feature = feature;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.bg = bg; A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.final_T = const_cast<float *>(final_T); A.ncontrib = const_cast<int *>(ncontrib);
    A.dL_dout = dL_dout;
    A.slot_sorted = slot_sorted; A.pair_buf = pair_records;
    A.pack = pack_scratch; A.pack_valid = 0;
    A.dbg_T_front = dbg_T_front;
    const int T = A.gx * ((H + TILE - 1) / TILE);
    A.F = F; A.T = T; A.cap = capacity; A.tile_only = 1;
    A.pack_fs = (long long)P * (long long)splat_blend_pack_floats(cn);
    A.opacity_fs = opacity_frame_stride; A.feature_fs = feature_frame_stride;
    A.c0 = c0; A.cn = cn;
    A.dL_dabs_uv = want_abs ? pair_records : nullptr;
    return bwd_chunk(A, T, false, true, (hipStream_t)stream);
}

// Backward tile pass of up to three feature sets of one geometry in ONE launch (blend_bwd_sets_kernel): set 0 = the tap set
// (<= 4 channels: full gradients + the densification taps), set 1 = a second set blended with the live opacity (<= 4
// channels), set 2 = the set blended with opacity.detach() (<= 20 channels); set_cn[g] = 0: no such set.  The sets must tile
// the row's C channels exactly.  set_c0 / set_cn / set_bg are HOST arrays of three entries.  Records: stride
// splat_blend_sets_pair_stride(C), layout [ux uy ca cb | cc o ax ay | tx ty 0 0 | dL_dfeature[0..C-1]] (reduce them with
// splat_frames_gauss_backward_static_sets); pack_scratch: F * P * splat_blend_sets_pack_floats() floats.
extern "C" size_t splat_blend_sets_pair_stride(int C) { return (size_t)PAIR_STRIDE(SetsCfg::NG + C); }
extern "C" size_t splat_blend_sets_pack_floats(void) { return (size_t)Rec<SetsCfg::CH>::RS; }

// Does splat_alpha_blending_backward_batch_sets_packed stage the FORWARD's packed records for this plan (no packing launch, no
// pack_scratch needed)?  The one place the condition lives: callers ask instead of re-deriving it.
extern "C" int splat_blend_sets_uses_forward_pack(int C, const int32_t *set_c0, const int32_t *set_cn, int has_cull_flags) {
    if (!set_c0 || !set_cn) return 0;
    const bool std_plan = C == 23 && set_c0[0] == 0 && set_cn[0] == 3 && set_c0[1] == 3 && set_cn[1] == 1 && set_c0[2] == 4 &&
                          set_cn[2] == 19 && sets_std_plan_enabled();
    return (std_plan && has_cull_flags && bwd_use_quarters()) ? 1 : 0;
}

static int backward_batch_sets_impl(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                    const float *set_bg, const float *uv, const float *conic,
                                    const float *opacity, int64_t opacity_frame_stride,
                                    const float *feature, int64_t feature_frame_stride,
                                    const float *const *set_feature, const int64_t *set_feature_fs,
                                    const int32_t *idx_sorted, const int32_t *tile_range,
                                    int64_t capacity, int W, int H, const float *final_T,
                                    const int32_t *ncontrib, const float *dL_dout,
                                    const float *const *set_dL, int want_abs,
                                    const int32_t *slot_sorted, float *pair_records,
                                    float *pack_scratch, const uint32_t *cull_flags,
                                    float *dbg_T_front, const float *forward_pack, splat_stream_t stream,
                                    const float *l1_pred = nullptr, const float *l1_scale = nullptr, float *l1_sum = nullptr) {
    SPLAT_CHECK_ARG(F >= 1 && P >= 1 && C >= 1 && C <= SetsCfg::CH && W > 0 && H > 0 && capacity >= 1, "bad sizes (C <= 28)");
    SPLAT_CHECK_ARG(!l1_pred || (l1_scale && set_dL && !dL_dout && cull_flags && bwd_use_quarters()),
                    "loss-fused backward: needs the sets' target images, the scales, the forward's cull words and the quarter-list kernels");
    SPLAT_CHECK_ARG(set_c0 && set_cn && set_bg, "null set table");
    SPLAT_CHECK_ARG(uv && conic && opacity && idx_sorted && tile_range && final_T && ncontrib && slot_sorted && pair_records &&
                        (pack_scratch || forward_pack),
                    "null pointer");
    SPLAT_CHECK_ARG(feature || (set_feature && set_feature_fs), "features: the row [F,P,C] or the sets' own tensors");
    SPLAT_CHECK_ARG(dL_dout || set_dL, "image gradient: the row [F,C,H,W] or the sets' own tensors");
    SPLAT_CHECK_ARG(set_cn[0] >= 0 && set_cn[0] <= 4 && set_cn[1] >= 0 && set_cn[1] <= 4 && set_cn[2] >= 0 && set_cn[2] <= 20,
                    "set widths: tap set <= 4, second set <= 4, detached set <= 20 channels");
    {   // the sets tile [0, C): every row channel has exactly one slot (every record component is written)
        unsigned covered = 0u;
        for (int g = 0; g < 3; ++g) {
            SPLAT_CHECK_ARG(set_cn[g] == 0 || (set_c0[g] >= 0 && set_c0[g] + set_cn[g] <= C), "set outside the row");
            for (int k = 0; k < set_cn[g]; ++k) {
                SPLAT_CHECK_ARG(!(covered & (1u << (set_c0[g] + k))), "the sets overlap");
                covered |= 1u << (set_c0[g] + k);
            }
        }
        SPLAT_CHECK_ARG(covered == (C == 32 ? 0xffffffffu : (1u << C) - 1u), "the sets do not cover the row's channels");
    }
    BlendArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.C = C;
    A.uv = (const float2 *)uv; A.conic = conic; A.opacity = opacity; A.feature = feature;
    A.idx_sorted = idx_sorted; A.tile_range = (const int2 *)tile_range;
    A.W = W; A.H = H; A.gx = (W + TILE - 1) / TILE;
    A.final_T = const_cast<float *>(final_T); A.ncontrib = const_cast<int *>(ncontrib);
    A.dL_dout = dL_dout;
    A.slot_sorted = slot_sorted; A.pair_buf = pair_records;
    A.pack = pack_scratch;
    A.dbg_T_front = dbg_T_front;
    A.cull_flags = const_cast<uint32_t *>(cull_flags);
    const int T = A.gx * ((H + TILE - 1) / TILE);
    A.F = F; A.T = T; A.cap = capacity; A.tile_only = 1;
    A.pack_fs = (long long)P * (long long)Rec<SetsCfg::CH>::RS;
    A.opacity_fs = opacity_frame_stride; A.feature_fs = feature_frame_stride;
    A.c0 = 0; A.cn = C;
    A.s0c0 = set_c0[0]; A.s0cn = set_cn[0]; A.s0bg = set_bg[0];
    A.s1c0 = set_c0[1]; A.s1cn = set_cn[1]; A.s1bg = set_bg[1];
    A.s2c0 = set_c0[2]; A.s2cn = set_cn[2]; A.s2bg = set_bg[2];
    const bool from_forward = forward_pack && splat_blend_sets_uses_forward_pack(C, set_c0, set_cn, cull_flags != nullptr);
    if (!feature) {
        A.sf0 = set_feature[0]; A.sf1 = set_feature[1]; A.sf2 = set_feature[2];
        A.sfs0 = set_feature_fs[0]; A.sfs1 = set_feature_fs[1]; A.sfs2 = set_feature_fs[2];
        const bool have_sets = (!set_cn[0] || A.sf0) && (!set_cn[1] || A.sf1) && (!set_cn[2] || A.sf2);
        // (the forward's packed records carry the row: the sets' own tensors are not read, e.g. a row described by sources)
        SPLAT_CHECK_ARG(from_forward || have_sets || forward_pack, "null set feature pointer");
        if (!from_forward && !have_sets) {
            // a row described by SOURCES under a plan the tile kernel does not stage from the forward's records: the packing
            // launch reads the row's channels out of those records ([u v A B | C o . id | channels 0 .. C-1 | pad], one per
            // Gaussian and frame) -- any one-pass plan serves feature lists and per-frame tensors (round 6)
            const int rsf = (int)splat_blend_pack_floats(C);
            A.feature = forward_pack + 8;
            A.feature_fs = (long long)P * rsf;
            A.feature_row = rsf;
            A.sf0 = A.sf1 = A.sf2 = nullptr;
        }
    }
    if (!dL_dout) {
        A.sdl0 = set_dL[0]; A.sdl1 = set_dL[1]; A.sdl2 = set_dL[2];
        SPLAT_CHECK_ARG((!set_cn[0] || A.sdl0) && (!set_cn[1] || A.sdl1) && (!set_cn[2] || A.sdl2), "null set gradient pointer");
    }
    if (l1_pred) {
        A.l1_pred = l1_pred; A.l1_sum = l1_sum;
        A.l1_s0 = l1_scale[0]; A.l1_s1 = l1_scale[1]; A.l1_s2 = l1_scale[2];
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(T * F)), block(256);
    // the renderer's own plan (rgb 0-2 | depth 3 | 19 attributes 4-22): channel gradients stored as whole float4
    const bool std_plan = C == 23 && set_c0[0] == 0 && set_cn[0] == 3 && set_c0[1] == 3 && set_cn[1] == 1 && set_c0[2] == 4 &&
                          set_cn[2] == 19 && sets_std_plan_enabled();
    if (from_forward) {
        // the forward's packed records of this row (splat_alpha_blending_forward_batch_sets / _forward with C = 23: 32 floats per
        // Gaussian and frame) are staged directly: no packing launch
        A.pack = const_cast<float *>(forward_pack);
        A.pack_fs = (long long)P * 32;
        if (want_abs) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<true, true, true>), grid, block, 0, s, A);
        else SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<false, true, true>), grid, block, 0, s, A);
        SPLAT_POST_LAUNCH();
        return SPLAT_OK;
    }
    SPLAT_CHECK_ARG(pack_scratch, "pack_scratch is needed: the forward's records do not serve this plan / these kernels");
    SPLAT_LAUNCH("blend_pack", pack_sets_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)F), dim3(256), 0, s, A);
    SPLAT_POST_LAUNCH();
    if (A.cull_flags && bwd_use_quarters()) {   // quarter lists (the forward's quarter bits)
        // a detached set of at most four / eight channels leaves slots 12 .. / 16 .. 27 empty: the SMALL instantiations
        const int small = std_plan ? 0 : set_cn[2] <= 4 ? 3 : set_cn[2] <= 8 ? 4 : 0;
        if (want_abs) {
            if (std_plan) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<true, true>), grid, block, 0, s, A);
            else if (small == 3) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<true, false, false, 3>), grid, block, 0, s, A);
            else if (small == 4) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<true, false, false, 4>), grid, block, 0, s, A);
            else SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<true, false>), grid, block, 0, s, A);
        } else {
            if (std_plan) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<false, true>), grid, block, 0, s, A);
            else if (small == 3) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<false, false, false, 3>), grid, block, 0, s, A);
            else if (small == 4) SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<false, false, false, 4>), grid, block, 0, s, A);
            else SPLAT_LAUNCH("blend_bwd", (blend_bwd_sets_quarter_kernel<false, false>), grid, block, 0, s, A);
        }
    } else if (want_abs) SPLAT_LAUNCH("blend_bwd", blend_bwd_sets_kernel<true>, grid, block, 0, s, A);
    else SPLAT_LAUNCH("blend_bwd", blend_bwd_sets_kernel<false>, grid, block, 0, s, A);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_alpha_blending_backward_batch_sets(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                                        const float *set_bg, const float *uv, const float *conic,
                                                        const float *opacity, int64_t opacity_frame_stride,
                                                        const float *feature, int64_t feature_frame_stride,
                                                        const float *const *set_feature, const int64_t *set_feature_fs,
                                                        const int32_t *idx_sorted, const int32_t *tile_range,
                                                        int64_t capacity, int W, int H, const float *final_T,
                                                        const int32_t *ncontrib, const float *dL_dout,
                                                        const float *const *set_dL, int want_abs,
                                                        const int32_t *slot_sorted, float *pair_records,
                                                        float *pack_scratch, const uint32_t *cull_flags,
                                                        float *dbg_T_front, splat_stream_t stream) {
    return backward_batch_sets_impl(F, P, C, set_c0, set_cn, set_bg, uv, conic, opacity, opacity_frame_stride, feature,
                                    feature_frame_stride, set_feature, set_feature_fs, idx_sorted, tile_range, capacity, W, H,
                                    final_T, ncontrib, dL_dout, set_dL, want_abs, slot_sorted, pair_records, pack_scratch,
                                    cull_flags, dbg_T_front, nullptr, stream);
}

// The same with the packed records the FORWARD left for this row (`forward_pack`: the pack_scratch of
// splat_alpha_blending_forward_batch_sets / splat_alpha_blending_forward[_flags] called with the same C = 23 channels, F * P * 32
// floats, untouched since): for the renderer's own plan (rgb at channels 0-2 with the taps, the depth at channel 3, 19 detached
// attributes at channels 4-22) with the forward's cull words the tile kernel stages them directly and no packing launch runs;
// any other plan (or forward_pack = NULL) behaves like splat_alpha_blending_backward_batch_sets.
extern "C" int splat_alpha_blending_backward_batch_sets_packed(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                                               const float *set_bg, const float *uv, const float *conic,
                                                               const float *opacity, int64_t opacity_frame_stride,
                                                               const float *feature, int64_t feature_frame_stride,
                                                               const float *const *set_feature, const int64_t *set_feature_fs,
                                                               const int32_t *idx_sorted, const int32_t *tile_range,
                                                               int64_t capacity, int W, int H, const float *final_T,
                                                               const int32_t *ncontrib, const float *dL_dout,
                                                               const float *const *set_dL, int want_abs,
                                                               const int32_t *slot_sorted, float *pair_records,
                                                               float *pack_scratch, const uint32_t *cull_flags,
                                                               float *dbg_T_front, const float *forward_pack,
                                                               splat_stream_t stream) {
    return backward_batch_sets_impl(F, P, C, set_c0, set_cn, set_bg, uv, conic, opacity, opacity_frame_stride, feature,
                                    feature_frame_stride, set_feature, set_feature_fs, idx_sorted, tile_range, capacity, W, H,
                                    final_T, ncontrib, dL_dout, set_dL, want_abs, slot_sorted, pair_records, pack_scratch,
                                    cull_flags, dbg_T_front, forward_pack, stream);
}

// Per-Gaussian sums of pair records of ANY layout: out[i, :] = sum of the records in Gaussian i's slots [goff_incl[i-1],
// goff_incl[i]) (stride `ncp` floats, a multiple of 4).  The SINGLE-FRAME use of splat_alpha_blending_backward_batch_sets
// (F = 1: the per-frame operator gs.alpha_blending_shared, where the three blends are separate from the projection and the
// gradients go back to autograd as d uv / d conic / d opacity / d feature): the caller slices the summed SETS record
// [ux uy ca cb | cc o ax ay | tx ty 0 0 | row channels].  A thread per (Gaussian, 16-byte chunk): the threads of a record read
// it contiguously.  out is fully written ([P, ncp]); deterministic (slot order).
// splat_alpha_blending_backward_batch_sets_packed with the L1 image loss FUSED into the tile kernel's hoist of the image gradient:
// set_target[g] = the set's target image [F, cn, H, W] (instead of a gradient image), pred_row = the forward's output row
// [F, C, H, W], the gradient of a channel of set g is l1_scale_host[g] * sign(pred - target) and l1_sum[(f * tiles + t) * 3 + g]
// (optional, [F, tiles, 3]: every entry is WRITTEN) = sum |pred - target| over the pixels of tile t of frame f, set g -- the caller
// adds them up (bit-reproducible; no atomics).  What splat_l1_loss_grad + the backward did in four launches and 1.9 GB of gradient
// images written and read back (src/trainer_fragGS.py:573-600: l1_loss on the three rendered images).  Quarter-list kernels only.
extern "C" int splat_alpha_blending_backward_batch_sets_l1(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                                           const float *set_bg, const float *uv, const float *conic,
                                                           const float *opacity, int64_t opacity_frame_stride,
                                                           const float *const *set_feature, const int64_t *set_feature_fs,
                                                           const int32_t *idx_sorted, const int32_t *tile_range,
                                                           int64_t capacity, int W, int H, const float *final_T,
                                                           const int32_t *ncontrib, const float *pred_row,
                                                           const float *const *set_target, const float *l1_scale_host,
                                                           float *l1_sum, int want_abs, const int32_t *slot_sorted,
                                                           float *pair_records, float *pack_scratch, const uint32_t *cull_flags,
                                                           float *dbg_T_front, const float *forward_pack, splat_stream_t stream) {
    SPLAT_CHECK_ARG(pred_row && set_target && l1_scale_host, "null pointer");
    return backward_batch_sets_impl(F, P, C, set_c0, set_cn, set_bg, uv, conic, opacity, opacity_frame_stride, nullptr, 0,
                                    set_feature, set_feature_fs, idx_sorted, tile_range, capacity, W, H, final_T, ncontrib,
                                    nullptr, set_target, want_abs, slot_sorted, pair_records, pack_scratch, cull_flags, dbg_T_front,
                                    forward_pack, stream, pred_row, l1_scale_host, l1_sum);
}

namespace {
__global__ void __launch_bounds__(256)
records_segment_sum_kernel(int P, int nq, const float4 *__restrict__ rec, const int *__restrict__ goff, float4 *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(t / nq), c = (int)(t - (long long)i * nq);
    if (i >= P) return;
    const int beg = i > 0 ? goff[i - 1] : 0, end = goff[i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    int j = beg;
    for (; j + 1 < end; j += 2) {   // two records in flight
        const float4 v0 = rec[(size_t)j * nq + c], v1 = rec[(size_t)(j + 1) * nq + c];
        a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
        b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
    }
    if (j < end) {
        const float4 v0 = rec[(size_t)j * nq + c];
        a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
    }
    out[(size_t)i * nq + c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
}  // namespace

extern "C" int splat_pair_records_segment_sum(int P, int ncp, const float *pair_records, const int32_t *goff_incl, float *out,
                                              splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && ncp >= 4 && ncp % 4 == 0, "bad sizes (the record stride is a multiple of 4 floats)");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(pair_records && goff_incl && out, "null pointer");
    SPLAT_CHECK_ARG(((uintptr_t)pair_records | (uintptr_t)out) % 16 == 0, "records and out must be 16-byte aligned");
    const int nq = ncp / 4;
    const long long threads = (long long)P * nq;
    SPLAT_LAUNCH("pair_reduce", records_segment_sum_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                 P, nq, (const float4 *)pair_records, goff_incl, (float4 *)out);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
