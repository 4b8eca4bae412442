// Densification statistics and structure updates on the device.
//   densify_accumulate  one frame of a batch: the gradient tap (ndc / abs_ndc .grad, reference
//                       src/pointrix/renderer/dptr_ortho_enhanced.py:331-349,379) summed into viewspace_grad, visibility
//                       OR-ed, radii max-ed (render_batch, :425-431; accumulate_viewspace_grad,
//                       src/pointrix/optimizer/atlas_gs_optimizer.py:414-433) -- one pass instead of 3 F eager kernels
//   densify_update      once per step (update_structure, atlas_gs_optimizer.py:110-121)
//   densify_masks       clone / split / prune decisions (:199-251, :363-375)
//   compact_scan/rows   stream compaction of every per-Gaussian tensor (parameters, Adam moments, statistics) with ONE
//                       prefix sum of the keep mask (prune_optimizer, src/pointrix/point_cloud/points.py:282-312 does a
//                       boolean-index pass with its own nonzero() per tensor)
// All HBM streaming, 5..30 bytes per Gaussian.
#include "common.h"

namespace {

constexpr int DB = 256;
inline dim3 dgrid(size_t n) { return dim3((unsigned)((n + DB - 1) / DB)); }

__global__ void __launch_bounds__(DB)
densify_accumulate_kernel(int P, const int *__restrict__ radius, const float2 *__restrict__ tap, float sx, float sy,
                          float2 *__restrict__ viewspace_grad, unsigned char *__restrict__ visible,
                          int *__restrict__ radii) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= P) return;
    if (tap && viewspace_grad) {
        const float2 t = tap[i];
        float2 g = viewspace_grad[i];
        // rounded product first (the tap itself is a rounded float32 in the reference): no FMA contraction here
        float px = t.x * sx, py = t.y * sy;
        asm volatile("" : "+v"(px), "+v"(py));
        g.x += px;
        g.y += py;
        viewspace_grad[i] = g;
    }
    const int r = radius[i];
    if (r > 0) visible[i] = 1;
    if (r > radii[i]) radii[i] = r;
}

__global__ void __launch_bounds__(DB)
densify_update_kernel(int P, const unsigned char *__restrict__ visible, const float2 *__restrict__ viewspace_grad,
                      const int *__restrict__ radii, float *__restrict__ max_radii2D,
                      float *__restrict__ pos_gradient_accum, float *__restrict__ denom) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= P || !visible[i]) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)radii[i]);
    const float2 g = viewspace_grad[i];
    pos_gradient_accum[i] += sqrtf(g.x * g.x + g.y * g.y);
    denom[i] += 1.0f;
}

__global__ void __launch_bounds__(DB)
densify_masks_kernel(int P, const float *__restrict__ accum, const float *__restrict__ denom,
                     const float *__restrict__ max_radii2D, const float *__restrict__ scaling_raw,
                     const float *__restrict__ opacity_raw, float grad_threshold, float dense, float big_ws,
                     float min_opacity, float size_threshold, unsigned char *__restrict__ clone,
                     unsigned char *__restrict__ split, unsigned char *__restrict__ prune) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= P) return;
    float g = accum[i] / denom[i];
    if (isnan(g)) g = 0.f;  // never visible: 0 / 0
    const float smax = fmaxf(fmaxf(expf(scaling_raw[3 * i]), expf(scaling_raw[3 * i + 1])), expf(scaling_raw[3 * i + 2]));
    if (clone) clone[i] = (fabsf(g) >= grad_threshold) && (smax <= dense);
    if (split) split[i] = (g >= grad_threshold) && (smax > dense);
    if (prune) {
        const float op = 1.0f / (1.0f + expf(-opacity_raw[i]));
        prune[i] = (op < min_opacity) || (size_threshold > 0.f && (max_radii2D[i] > size_threshold || smax > big_ws));
    }
}

// ---- prefix sum of a byte mask: per-block counts, single-workgroup scan of the counts, block-local positions
__global__ void __launch_bounds__(DB)
compact_count_kernel(int P, const unsigned char *__restrict__ mask, int *__restrict__ block_sum) {
    __shared__ int ws[DB / WAVE];
    const int i = blockIdx.x * DB + threadIdx.x;
    const bool k = i < P && mask[i] != 0;
    const unsigned long long m = __ballot(k);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(1024)
compact_blockscan_kernel(int nb, int *__restrict__ block_sum, int *__restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int t = base + threadIdx.x;
        const int v = t < nb ? block_sum[t] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o);
            if (lane >= o) inc += n;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (t < nb) block_sum[t] = carry + woff + inc - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = carry_s;
}

__global__ void __launch_bounds__(DB)
compact_index_kernel(int P, const unsigned char *__restrict__ mask, const int *__restrict__ block_off,
                     int *__restrict__ index) {
    __shared__ int ws[DB / WAVE];
    const int i = blockIdx.x * DB + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool k = i < P && mask[i] != 0;
    const unsigned long long m = __ballot(k);
    if (lane == 0) ws[w] = __popcll(m);
    __syncthreads();
    int off = block_off[blockIdx.x];
    for (int q = 0; q < w; ++q) off += ws[q];
    if (i < P) index[i] = off + __popcll(m & ((1ull << lane) - 1ull));  // destination row if kept
}

// row i (row_words 32-bit words) -> row index[i] of dst when mask[i]; consecutive lanes move consecutive words
__global__ void __launch_bounds__(DB)
compact_rows_kernel(size_t total_words, int row_words, const unsigned char *__restrict__ mask,
                    const int *__restrict__ index, const unsigned int *__restrict__ src, unsigned int *__restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * DB + threadIdx.x;
    if (e >= total_words) return;
    const size_t i = e / (size_t)row_words;
    if (!mask[i]) return;
    const size_t w = e - i * (size_t)row_words;
    dst[(size_t)index[i] * row_words + w] = src[e];
}

// ---- structure surgery: selected rows appended `repeat` times (clone: 1, split: split_num) behind the existing rows
// dst row (r * n_sel + index[i]) = src row i for every selected i (the order of torch's x[mask].repeat(r, 1, ...))
__global__ void __launch_bounds__(DB)
gather_rows_repeat_kernel(size_t total_words, int row_words, const unsigned char *__restrict__ mask,
                          const int *__restrict__ index, int n_sel, int repeat, const unsigned int *__restrict__ src,
                          unsigned int *__restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * DB + threadIdx.x;
    if (e >= total_words) return;
    const size_t i = e / (size_t)row_words;
    if (!mask[i]) return;
    const size_t w = e - i * (size_t)row_words;
    const unsigned int v = src[e];
    for (int r = 0; r < repeat; ++r) dst[((size_t)r * n_sel + index[i]) * row_words + w] = v;
}

// Philox-4x32-10 (Salmon et al., SC'11): counter-based, so every rank draws the SAME normals for Gaussian i, replica r
// from (seed, i, r) without any shared generator state -- data-parallel replicas split identically.
__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                           unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0, 1)

// new_pos_scale of the reference (atlas_gs_optimizer.py:255-287): samples ~ N(0, diag(scaling^2)) rotated by the
// Gaussian's (normalised) rotation and added to its position; new scaling = log(scaling / (0.8 split_num)).
__global__ void __launch_bounds__(DB)
split_sample_kernel(int P, const unsigned char *__restrict__ mask, const int *__restrict__ index, int n_sel, int split_num,
                    unsigned seed_lo, unsigned seed_hi, const float *__restrict__ position,
                    const float *__restrict__ scaling_raw, const float4 *__restrict__ rotation_raw,
                    const float *__restrict__ unit_normals, float *__restrict__ new_pos, float *__restrict__ new_scaling) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= P || !mask[i]) return;
    const float s[3] = {expf(scaling_raw[3 * i]), expf(scaling_raw[3 * i + 1]), expf(scaling_raw[3 * i + 2])};
    float4 q = rotation_raw[i];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= nrm; q.y /= nrm; q.z /= nrm; q.w /= nrm;
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    for (int rep = 0; rep < split_num; ++rep) {
        const size_t row = (size_t)rep * n_sel + index[i];
        float zn[3];
        if (unit_normals) {
            zn[0] = unit_normals[3 * row]; zn[1] = unit_normals[3 * row + 1]; zn[2] = unit_normals[3 * row + 2];
        } else {
            unsigned u[4];
            philox4x32((unsigned)i, (unsigned)rep, 0u, 0u, seed_lo, seed_hi, u);
            const float r0 = sqrtf(-2.f * logf(u01(u[0]))), a0 = 6.283185307179586f * u01(u[1]);
            const float r1 = sqrtf(-2.f * logf(u01(u[2]))), a1 = 6.283185307179586f * u01(u[3]);
            zn[0] = r0 * cosf(a0); zn[1] = r0 * sinf(a0); zn[2] = r1 * cosf(a1);
        }
        const float sm[3] = {zn[0] * s[0], zn[1] * s[1], zn[2] * s[2]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            new_pos[3 * row + k] = R[k][0] * sm[0] + R[k][1] * sm[1] + R[k][2] * sm[2] + position[3 * i + k];
            new_scaling[3 * row + k] = logf(s[k] / (0.8f * (float)split_num));
        }
    }
}

// ---- spatial order of the Gaussians (Morton / Z-curve of the screen position).  The binning kernels walk the Gaussians in
// index order and write every (Gaussian, tile) pair into its tile's key segment; with neighbours in the image next to each
// other in memory those writes, the sort's owner gathers and the compositing kernels' record gathers touch a few tiles'
// worth of lines per workgroup instead of the whole image's (c2: bin_scatter 17.8 -> 9.4 us, tile_sort 17.1 -> 12.3 us per
// frame).  Densification rebuilds every per-Gaussian array anyway -- that is when the order is (re)established; between two
// densifications the Gaussians move by a fraction of a tile.  Results do not depend on the order (ties of exactly equal
// depths aside, which sort by Gaussian index as in the reference).
__device__ __forceinline__ unsigned int spread15(unsigned int v) {  // bits of a 15-bit number to the even positions
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    return (v | (v << 1)) & 0x55555555u;
}

__global__ void __launch_bounds__(DB)
morton_keys_kernel(int P, const float2 *__restrict__ uv, float sx, float sy, int *__restrict__ keys) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= P) return;
    const float2 q = uv[i];
    const float fx = fminf(fmaxf(q.x * sx, 0.f), 32767.f), fy = fminf(fmaxf(q.y * sy, 0.f), 32767.f);  // NaN -> 0
    keys[i] = (int)(spread15((unsigned int)fx) | (spread15((unsigned int)fy) << 1));
}

}  // namespace

// keys [P] int32 (>= 0): Morton code of (u / W, v / H) quantised to 15 bits each (positions outside the image clamp to its
// border).  argsort(keys, stable) is the spatial order; see densify.py::spatial_order.
extern "C" int splat_morton_keys(int P, const float *uv, int W, int H, int32_t *keys, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(uv && keys, "null pointer");
    SPLAT_LAUNCH("morton_keys", morton_keys_kernel, dgrid(P), dim3(DB), 0, (hipStream_t)stream, P, (const float2 *)uv,
                 32768.f / (float)W, 32768.f / (float)H, keys);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_gather_rows_repeat(int P, const uint8_t *mask, const int32_t *index, int n_sel, int repeat,
                                        int row_words, const void *src, void *dst, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && row_words >= 1 && repeat >= 1 && n_sel >= 0, "bad sizes");
    if (P == 0 || n_sel == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(mask && index && src && dst, "null pointer");
    const size_t total = (size_t)P * (size_t)row_words;
    SPLAT_LAUNCH("gather_rows", gather_rows_repeat_kernel, dgrid(total), dim3(DB), 0, (hipStream_t)stream, total, row_words,
                 mask, index, n_sel, repeat, (const unsigned int *)src, (unsigned int *)dst);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_densify_split_sample(int P, const uint8_t *mask, const int32_t *index, int n_sel, int split_num,
                                          uint64_t seed, const float *position, const float *scaling_raw,
                                          const float *rotation_raw, const float *unit_normals, float *new_pos,
                                          float *new_scaling, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && n_sel >= 0 && split_num >= 1, "bad sizes");
    if (P == 0 || n_sel == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(mask && index && position && scaling_raw && rotation_raw && new_pos && new_scaling, "null pointer");
    SPLAT_LAUNCH("split_sample", split_sample_kernel, dgrid(P), dim3(DB), 0, (hipStream_t)stream, P, mask, index, n_sel,
                 split_num, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), position, scaling_raw,
                 (const float4 *)rotation_raw, unit_normals, new_pos, new_scaling);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_densify_accumulate(int P, const int32_t *radius, const float *tap, float sx, float sy,
                                        float *viewspace_grad, uint8_t *visible, int32_t *radii, void *stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad size");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(radius && visible && radii, "null pointer");
    SPLAT_CHECK_ARG((tap == nullptr) == (viewspace_grad == nullptr), "tap and viewspace_grad go together");
    SPLAT_LAUNCH("densify_accumulate", densify_accumulate_kernel, dgrid(P), dim3(DB), 0, (hipStream_t)stream, P, radius,
                 (const float2 *)tap, sx, sy, (float2 *)viewspace_grad, visible, radii);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_densify_update(int P, const uint8_t *visible, const float *viewspace_grad, const int32_t *radii,
                                    float *max_radii2D, float *pos_gradient_accum, float *denom, void *stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad size");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(visible && viewspace_grad && radii && max_radii2D && pos_gradient_accum && denom, "null pointer");
    SPLAT_LAUNCH("densify_update", densify_update_kernel, dgrid(P), dim3(DB), 0, (hipStream_t)stream, P, visible,
                 (const float2 *)viewspace_grad, radii, max_radii2D, pos_gradient_accum, denom);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_densify_masks(int P, const float *pos_gradient_accum, const float *denom,
                                   const float *max_radii2D, const float *scaling_raw, const float *opacity_raw,
                                   float grad_threshold, float percent_dense, float cameras_extent, float min_opacity,
                                   float size_threshold, uint8_t *clone, uint8_t *split, uint8_t *prune, void *stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad size");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(pos_gradient_accum && denom && scaling_raw, "null pointer");
    SPLAT_CHECK_ARG(!prune || (opacity_raw && max_radii2D), "prune needs opacity_raw and max_radii2D");
    SPLAT_LAUNCH("densify_masks", densify_masks_kernel, dgrid(P), dim3(DB), 0, (hipStream_t)stream, P, pos_gradient_accum,
                 denom, max_radii2D, scaling_raw, opacity_raw, grad_threshold, percent_dense * cameras_extent,
                 0.1f * cameras_extent, min_opacity, size_threshold, clone, split, prune);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" size_t splat_compact_scratch_bytes(int P) {
    if (P < 0) return 0;
    return ((size_t)((P + DB - 1) / DB) + 64) * sizeof(int);
}

extern "C" int splat_compact_scan(int P, const uint8_t *mask, int32_t *index, int32_t *count, void *scratch,
                                  void *stream) {
    SPLAT_CHECK_ARG(P >= 0, "bad size");
    SPLAT_CHECK_ARG(count != nullptr, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {
        SPLAT_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int), s));
        return SPLAT_OK;
    }
    SPLAT_CHECK_ARG(mask && index && scratch, "null pointer");
    const int nb = (P + DB - 1) / DB;
    int *block_sum = (int *)scratch;
    SPLAT_LAUNCH("compact_count", compact_count_kernel, dim3(nb), dim3(DB), 0, s, P, mask, block_sum);
    SPLAT_POST_LAUNCH();
    SPLAT_LAUNCH("compact_blockscan", compact_blockscan_kernel, dim3(1), dim3(1024), 0, s, nb, block_sum, count);
    SPLAT_POST_LAUNCH();
    SPLAT_LAUNCH("compact_index", compact_index_kernel, dim3(nb), dim3(DB), 0, s, P, mask, block_sum, index);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_compact_rows(int P, const uint8_t *mask, const int32_t *index, int row_words, const void *src,
                                  void *dst, void *stream) {
    SPLAT_CHECK_ARG(P >= 0 && row_words >= 1, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(mask && index && src && dst, "null pointer");
    const size_t total = (size_t)P * (size_t)row_words;
    SPLAT_LAUNCH("compact_rows", compact_rows_kernel, dgrid(total), dim3(DB), 0, (hipStream_t)stream, total, row_words, mask,
                 index, (const unsigned int *)src, (unsigned int *)dst);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
