// Adam step over the flat parameter buffer of the frame-sharded data-parallel renderer ("one all-reduce of
// the flat bucket per optimiser step, then identical Adam on every rank").  The reference steps torch.optim.Adam
// parameter group by parameter group (src/pointrix/optimizer/optimizer.py, atlas_gs_optimizer.py: Adam with eps = 1e-15,
// per-group learning rates); this is the same update rule (no weight decay, no amsgrad) as ONE streaming launch over the
// contiguous buffer the collective just reduced, with the per-group learning rates looked up by segment:
//     m <- b1 m + (1 - b1) g        v <- b2 v + (1 - b2) g^2
//     p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 4 reads + 3 writes of 4 bytes per element, float4 lanes.
#include <math.h>

#include "common.h"

#ifndef ADAM_NONTEMPORAL
#define ADAM_NONTEMPORAL 1   // streaming loads / stores: 12.1 -> 10.7 us per frame on the 1.48 GB dynamic-model buffer, neutral on the static one
#endif
namespace {
constexpr int ADAM_MAX_SEG = SPLAT_ADAM_MAX_SEGMENTS;
struct AdamSegs {
    int n;
    long long end[ADAM_MAX_SEG];  // exclusive end of segment k (elements); segment k uses step size ss[k]
    float ss[ADAM_MAX_SEG];       // lr_k / (1 - b1^t)
    // optional pattern INSIDE a segment: of every `period[k]` consecutive elements (from the segment's start) the first `head[k]`
    // use ss_head[k] instead -- the SH block [N, 16, 3] of the reference holds two parameter groups interleaved per Gaussian
    // (features: the 3 DC floats, lr 0.0025; features_rest: the 45 others, lr 0.000125; src/configs/frag_gs_v10.yaml:44-47)
    int period[ADAM_MAX_SEG], head[ADAM_MAX_SEG];
    float ss_head[ADAM_MAX_SEG];
};

__device__ __forceinline__ float seg_step(const AdamSegs &S, long long i) {
    float s = S.ss[0], sh = S.ss_head[0];
    long long start = 0;
    int per = S.period[0], hd = S.head[0];
#pragma unroll
    for (int k = 1; k < ADAM_MAX_SEG; ++k)
        if (k < S.n && i >= S.end[k - 1]) { s = S.ss[k]; sh = S.ss_head[k]; start = S.end[k - 1]; per = S.period[k]; hd = S.head[k]; }
    if (per > 0 && (int)((unsigned long long)(i - start) % (unsigned)per) < hd) s = sh;
    return s;
}

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float ss, float b1, float b2, float rbc2,
                                      float eps) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= ss * m / (sqrtf(v) * rbc2 + eps);
}

__global__ void __launch_bounds__(256)
adam_kernel(long long n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
            AdamSegs S, float b1, float b2, float rbc2, float eps, float gscale) {
    const long long n4 = n >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
#if ADAM_NONTEMPORAL
        // a pure stream: every byte is touched once per step -- keep it out of the caches the next forward wants
        typedef float nt4 __attribute__((ext_vector_type(4)));
        const nt4 p_ = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(p) + q), m_ = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(m) + q);
        const nt4 v_ = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(v) + q), g_ = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(g) + q);
        float4 P = make_float4(p_.x, p_.y, p_.z, p_.w), M = make_float4(m_.x, m_.y, m_.z, m_.w), V = make_float4(v_.x, v_.y, v_.z, v_.w);
        const float4 G = make_float4(g_.x, g_.y, g_.z, g_.w);
#else
        float4 P = reinterpret_cast<float4 *>(p)[q], M = reinterpret_cast<float4 *>(m)[q], V = reinterpret_cast<float4 *>(v)[q];
        const float4 G = reinterpret_cast<const float4 *>(g)[q];
#endif
        const long long i = q << 2;
        adam1(P.x, G.x * gscale, M.x, V.x, seg_step(S, i), b1, b2, rbc2, eps);
        adam1(P.y, G.y * gscale, M.y, V.y, seg_step(S, i + 1), b1, b2, rbc2, eps);
        adam1(P.z, G.z * gscale, M.z, V.z, seg_step(S, i + 2), b1, b2, rbc2, eps);
        adam1(P.w, G.w * gscale, M.w, V.w, seg_step(S, i + 3), b1, b2, rbc2, eps);
#if ADAM_NONTEMPORAL
        __builtin_nontemporal_store((nt4){P.x, P.y, P.z, P.w}, reinterpret_cast<nt4 *>(p) + q);
        __builtin_nontemporal_store((nt4){M.x, M.y, M.z, M.w}, reinterpret_cast<nt4 *>(m) + q);
        __builtin_nontemporal_store((nt4){V.x, V.y, V.z, V.w}, reinterpret_cast<nt4 *>(v) + q);
#else
        reinterpret_cast<float4 *>(p)[q] = P;
        reinterpret_cast<float4 *>(m)[q] = M;
        reinterpret_cast<float4 *>(v)[q] = V;
#endif
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {  // tail
        const long long i = (n4 << 2) + threadIdx.x;
        adam1(p[i], g[i] * gscale, m[i], v[i], seg_step(S, i), b1, b2, rbc2, eps);
    }
}
// dst[0..n) = value: float4 stores, grid-stride (the gradient bucket is zeroed once per step: 70.8 MB at c2)
__global__ void __launch_bounds__(256)
fill_kernel(float *__restrict__ dst, long long n, float value) {
    const long long head = imin_((int)((16 - ((uintptr_t)dst & 15)) & 15) >> 2, (int)(n < 4 ? n : 4));   // floats up to 16-byte alignment
    float4 *d4 = reinterpret_cast<float4 *>(dst + head);
    const long long n4 = (n - head) >> 2;
    const float4 v4 = make_float4(value, value, value, value);
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) d4[q] = v4;
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) dst[threadIdx.x] = value;
        const long long tail0 = head + (n4 << 2);
        if (tail0 + threadIdx.x < n && threadIdx.x < 4) dst[tail0 + threadIdx.x] = value;
    }
}

// L1 image loss and its gradient in one pass (the photometric / depth / attribute terms of a training step,
// src/trainer_fragGS.py:573-600 -- l1_loss on the rendered frames): grad = scale * sign(pred - target), loss_sum += sum |pred -
// target|.  pred may be a channel slice of a wider image row: F blocks of `inner` contiguous floats, `pred_fs` floats apart;
// target and grad are dense [F, inner].  grid.y = frame.
__global__ void __launch_bounds__(256)
l1_loss_grad_kernel(long long inner, const float *__restrict__ pred, long long pred_fs, const float *__restrict__ target,
                    float scale, float *__restrict__ grad, float *__restrict__ loss_sum) {
    const long long f = blockIdx.y;
    const float *p = pred + f * pred_fs, *t = target + f * inner;
    float *g = grad + f * inner;
    float acc = 0.f;
    const bool vec = (((uintptr_t)p | (uintptr_t)t | (uintptr_t)g) & 15) == 0;
    const long long n4 = vec ? inner >> 2 : 0;
    typedef float nt4 __attribute__((ext_vector_type(4)));
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        // three streams touched once each (a step's images: ~1 GB each at 25 frames of 480p x 23 channels): streaming accesses
        const nt4 a = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(p) + q), b = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(t) + q);
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        auto sg = [scale](float d) { return d > 0.f ? scale : (d < 0.f ? -scale : 0.f); };
        __builtin_nontemporal_store((nt4){sg(d0), sg(d1), sg(d2), sg(d3)}, reinterpret_cast<nt4 *>(g) + q);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < inner; i += (long long)gridDim.x * 256) {
        const float d = p[i] - t[i];
        acc += fabsf(d);
        g[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    }
    acc = wave_sum_to_lane63(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum) atomic_add_f32(loss_sum, (part[0] + part[1]) + (part[2] + part[3]));
}
}  // namespace

// grad[f, i] = scale * sign(pred[f * pred_frame_stride + i] - target[f, i]), *loss_sum += sum |pred - target| (zero-init; float
// atomics: the reported sum is not bit-reproducible, the gradient is).  The mean-reduced L1 loss of the reference
// (l1_loss, src/trainer_fragGS.py:573) and its backward: scale = weight / (F * inner).
extern "C" int splat_l1_loss_grad(int F, int64_t inner, const float *pred, int64_t pred_frame_stride, const float *target,
                                  float scale, float *grad, float *loss_sum, splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 0 && F <= 65535 && inner >= 0, "bad sizes");
    if (F == 0 || inner == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(pred && target && grad, "null pointer");
    SPLAT_CHECK_ARG(pred_frame_stride >= inner, "pred_frame_stride below the block size");
    // few, fat workgroups: every workgroup ends with ONE atomic on the same address (25 600 of them serialised were half of the
    // kernel's time at 25 frames x 1024 workgroups)
    long long blocks = ((inner >> 2) + 255) / 256;
    const long long cap = F >= 16 ? 96 : (F >= 4 ? 384 : 1536);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    SPLAT_LAUNCH("l1_loss_grad", l1_loss_grad_kernel, dim3((unsigned)blocks, (unsigned)F), dim3(256), 0, (hipStream_t)stream,
                 (long long)inner, pred, (long long)pred_frame_stride, target, scale, grad, loss_sum);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_fill_f32(float *dst, size_t n, float value, splat_stream_t stream) {
    if (n == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(dst && ((uintptr_t)dst & 3) == 0, "dst must be a 4-byte aligned device pointer");
    long long blocks = ((long long)(n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    SPLAT_LAUNCH("fill", fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, (long long)n, value);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

static int adam_step_impl(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int nseg,
                          const int64_t *seg_end_host, const float *seg_lr_host, const int32_t *seg_period_host,
                          const int32_t *seg_head_host, const float *seg_lr_head_host, float beta1, float beta2, float eps, int step,
                          float grad_scale, splat_stream_t stream) {
    SPLAT_CHECK_ARG(n >= 0 && step >= 1, "bad sizes");
    SPLAT_CHECK_ARG(nseg >= 1 && nseg <= ADAM_MAX_SEG && seg_end_host && seg_lr_host, "1..SPLAT_ADAM_MAX_SEGMENTS segments");
    SPLAT_CHECK_ARG(seg_end_host[nseg - 1] == n, "the last segment must end at n");
    if (n == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "null pointer");
    SPLAT_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                    "buffers must be 16-byte aligned");
    AdamSegs S;
    S.n = nseg;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int k = 0; k < ADAM_MAX_SEG; ++k) {
        S.end[k] = k < nseg ? seg_end_host[k] : n;
        S.ss[k] = k < nseg ? (float)((double)seg_lr_host[k] / bc1) : 0.f;
        S.period[k] = (k < nseg && seg_period_host) ? seg_period_host[k] : 0;
        S.head[k] = (k < nseg && seg_head_host) ? seg_head_host[k] : 0;
        S.ss_head[k] = (k < nseg && seg_lr_head_host) ? (float)((double)seg_lr_head_host[k] / bc1) : 0.f;
        SPLAT_CHECK_ARG(S.period[k] >= 0 && S.head[k] >= 0 && S.head[k] <= S.period[k], "pattern: 0 <= head <= period");
        SPLAT_CHECK_ARG(k == 0 || k >= nseg || seg_end_host[k] >= seg_end_host[k - 1], "segments must be ascending");
    }
    const long long n4 = n >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    SPLAT_LAUNCH("adam_step", adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, param,
                 grad, exp_avg, exp_avg_sq, S, beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, grad_scale);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int nseg,
                               const int64_t *seg_end_host, const float *seg_lr_host, float beta1, float beta2, float eps,
                               int step, float grad_scale, splat_stream_t stream) {
    return adam_step_impl(n, param, grad, exp_avg, exp_avg_sq, nseg, seg_end_host, seg_lr_host, nullptr, nullptr, nullptr, beta1,
                          beta2, eps, step, grad_scale, stream);
}

// The same with a learning-rate PATTERN inside segments: of every seg_period[k] consecutive elements of segment k the first
// seg_head[k] take seg_lr_head[k] (period 0: none) -- two parameter groups interleaved in one tensor, e.g. the reference's
// features / features_rest inside the [N, 16, 3] SH block (period 48, head 3).
extern "C" int splat_adam_step_pattern(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int nseg,
                                       const int64_t *seg_end_host, const float *seg_lr_host, const int32_t *seg_period_host,
                                       const int32_t *seg_head_host, const float *seg_lr_head_host, float beta1, float beta2,
                                       float eps, int step, float grad_scale, splat_stream_t stream) {
    SPLAT_CHECK_ARG(seg_period_host && seg_head_host && seg_lr_head_host, "null pattern table");
    return adam_step_impl(n, param, grad, exp_avg, exp_avg_sq, nseg, seg_end_host, seg_lr_host, seg_period_host, seg_head_host,
                          seg_lr_head_host, beta1, beta2, eps, step, grad_scale, stream);
}
