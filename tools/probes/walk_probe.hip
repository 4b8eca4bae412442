// Standalone probe (GPU box: hipcc --offload-arch=gfx950 -O3 walk_probe.hip -o walk_probe && ./walk_probe): how fast can the
// Gaussian-side walk over the pair records (F frames x Gaussian-major slots, ~3.84 records of NCP floats per Gaussian and frame)
// read them?  Compares the product kernel's access pattern with a pure stream and a few re-orderings.  Prints GB/s of record bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NCP = 12;

__global__ void k_stream(const float4 *p, size_t n4, float *out) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

// quad per Gaussian, lane `sub` reads chunk sub of every record; frames in the order order(f, blockIdx)
template <int U, int STAGGER>
__global__ void __launch_bounds__(256) k_quad(int F, int P, long long cap, const float *pair, const int *goff, float *out) {
    __shared__ int s_goff[32][65];
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, sub = t & 3;
    const int i0 = blockIdx.x * 64;
    for (int c = threadIdx.x; c < F * 65; c += 256) {
        const int f = c / 65, g = c - f * 65, gi = i0 + g - 1;
        s_goff[f][g] = (gi >= 0 && gi < P) ? goff[(size_t)f * P + gi] : 0;
    }
    __syncthreads();
    if (i >= P) return;
    const int li = i - i0;
    float4 a = make_float4(0, 0, 0, 0);
    float4 bufA[U], bufB[U];
    auto issue = [&](float4 (&v)[U], int f) {
        const int beg = s_goff[f][li], end = s_goff[f][li + 1];
        const float *base = pair + (size_t)f * cap * NCP + 4 * sub;
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = (beg + u < end && sub < 3) ? *reinterpret_cast<const float4 *>(base + (size_t)(beg + u) * NCP) : make_float4(0, 0, 0, 0);
    };
    auto consume = [&](const float4 (&v)[U], int f) {
#pragma unroll
        for (int u = 0; u < U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        const int beg = s_goff[f][li], end = s_goff[f][li + 1];
        const float *base = pair + (size_t)f * cap * NCP + 4 * sub;
        for (int j = beg + U; j < end; ++j)
            if (sub < 3) { const float4 v1 = *reinterpret_cast<const float4 *>(base + (size_t)j * NCP); a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w; }
    };
    auto fr = [&](int k) { return STAGGER ? (k + (int)blockIdx.x * STAGGER) % F : k; };
    issue(bufA, fr(0));
    for (int f = 0; f < F; f += 2) {
        if (f + 1 < F) issue(bufB, fr(f + 1));
        consume(bufA, fr(f));
        if (f + 2 < F) issue(bufA, fr(f + 2));
        if (f + 1 < F) consume(bufB, fr(f + 1));
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

// one thread per Gaussian: its records of a frame as one contiguous run of float4
__global__ void __launch_bounds__(256) k_thread(int F, int P, long long cap, const float *pair, const int *goff, float *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float4 a = make_float4(0, 0, 0, 0);
    for (int f = 0; f < F; ++f) {
        const int *g = goff + (size_t)f * P;
        const int beg = i > 0 ? g[i - 1] : 0, end = g[i];
        const float4 *base = reinterpret_cast<const float4 *>(pair + (size_t)f * cap * NCP);
        for (int c = 3 * beg; c < 3 * end; ++c) { const float4 v = base[c]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

// frame-major: blockIdx.y = frame; every workgroup sums its 64 Gaussians' records of ONE frame and writes the sums [F, P, NCP]
__global__ void __launch_bounds__(256) k_frame_major(int F, int P, long long cap, const float *pair, const int *goff, float *sums) {
    const int f = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, sub = t & 3;
    if (i >= P) return;
    const int *g = goff + (size_t)f * P;
    const int beg = i > 0 ? g[i - 1] : 0, end = g[i];
    const float *base = pair + (size_t)f * cap * NCP + 4 * sub;
    float4 a = make_float4(0, 0, 0, 0);
    if (sub < 3) {
        for (int j = beg; j < end; ++j) { const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)j * NCP); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        *reinterpret_cast<float4 *>(sums + ((size_t)f * P + i) * NCP + 4 * sub) = a;
    }
}

// wave per run of Gaussians: the wave streams the contiguous records of its 64 Gaussians of a frame as float4, lane-contiguous
// (no per-Gaussian ownership: measures what the access ORDER alone can give)
__global__ void __launch_bounds__(256) k_wave_stream(int F, int P, long long cap, const float *pair, const int *goff, float *out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int i0 = wave * 64;
    if (i0 >= P) return;
    const int i1 = min(P, i0 + 64);
    float4 a = make_float4(0, 0, 0, 0);
    for (int f = 0; f < F; ++f) {
        const int *g = goff + (size_t)f * P;
        const int beg = i0 > 0 ? g[i0 - 1] : 0, end = g[i1 - 1];
        const float4 *base = reinterpret_cast<const float4 *>(pair + (size_t)f * cap * NCP);
        for (int c = 3 * beg + lane; c < 3 * end; c += 64) { const float4 v = base[c]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__device__ __forceinline__ void st16(float4 *p, const float4 &v) {
    const v4f t = {v.x, v.y, v.z, v.w};
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
    else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(t) : "memory");
    else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(t) : "memory");
}
// the tile kernel's side: every record written once as three 16-byte stores by one thread, slots in a scrambled order
template <int MODE>
__global__ void __launch_bounds__(256) k_write(long long cap, int F, float *pair) {
    const long long n = cap * F;
    for (long long r = blockIdx.x * 256ll + threadIdx.x; r < n; r += (long long)gridDim.x * 256) {
        const long long f = r / cap, j = r - f * cap;
        const long long slot = (j * 7919) % cap;   // scattered inside the frame
        float4 *dst = reinterpret_cast<float4 *>(pair + (f * cap + slot) * NCP);
        st16<MODE>(dst, make_float4(1.f, 2.f, 3.f, 4.f)); st16<MODE>(dst + 1, make_float4(1.f, 2.f, 3.f, 4.f)); st16<MODE>(dst + 2, make_float4(1.f, 2.f, 3.f, 4.f));
    }
}

// time of `fn` alone, each repetition right behind `pre` (not timed)
template <typename Pre, typename Fn>
static float time_after_ms(Pre pre, Fn fn, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0.f;
    for (int r = 0; r < reps + 1; ++r) {
        pre();
        CK(hipEventRecord(e0));
        fn();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) tot += ms;
    }
    return tot / reps;
}

template <typename Fn>
static float time_ms(Fn fn, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int F = 25, P = 300000;
    std::vector<int> goff((size_t)F * P);
    unsigned s = 12345u;
    long long M = 0;
    for (int f = 0; f < F; ++f) {
        int acc = 0;
        for (int i = 0; i < P; ++i) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = (s >> 8) % 100;
            const int tiles = r < 10 ? 1 : r < 30 ? 2 : r < 75 ? 4 : r < 92 ? 6 : 9;   // mean 3.9
            acc += tiles;
            goff[(size_t)f * P + i] = acc;
        }
        if (acc > M) M = acc;
    }
    const long long cap = M + 1024;
    const size_t nfl = (size_t)F * cap * NCP;
    printf("F %d P %d pairs/frame <= %lld cap %lld records %.1f MB/frame total %.2f GB\n", F, P, M, cap, M * NCP * 4 / 1e6, nfl * 4 / 1e9);
    float *pair, *out, *sums;
    int *dgoff;
    CK(hipMalloc(&pair, nfl * 4)); CK(hipMemset(pair, 0, nfl * 4));
    CK(hipMalloc(&out, 1024)); CK(hipMalloc(&sums, (size_t)F * P * NCP * 4));
    CK(hipMalloc(&dgoff, goff.size() * 4)); CK(hipMemcpy(dgoff, goff.data(), goff.size() * 4, hipMemcpyHostToDevice));
    double used = 0;
    for (int f = 0; f < F; ++f) used += (double)goff[(size_t)f * P + P - 1] * NCP * 4;
    auto rep = [&](const char *name, float ms, double bytes) { printf("%-44s %8.3f ms  %7.1f GB/s (%.1f us per frame)\n", name, ms, bytes / ms / 1e6, ms * 1e3 / F); };
    rep("stream (whole buffer, float4 grid-stride)", time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)pair, nfl / 4, out); }), nfl * 4.0);
    const int gq = (P * 4 + 255) / 256;
    rep("quad walk U=6 (product pattern)", time_ms([&] { hipLaunchKernelGGL((k_quad<6, 0>), dim3(gq), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    rep("quad walk U=4", time_ms([&] { hipLaunchKernelGGL((k_quad<4, 0>), dim3(gq), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    rep("quad walk U=10", time_ms([&] { hipLaunchKernelGGL((k_quad<10, 0>), dim3(gq), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    rep("quad walk U=6, frames staggered by 1 per wg", time_ms([&] { hipLaunchKernelGGL((k_quad<6, 1>), dim3(gq), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    rep("quad walk U=6, frames staggered by 7 per wg", time_ms([&] { hipLaunchKernelGGL((k_quad<6, 7>), dim3(gq), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    rep("thread per Gaussian", time_ms([&] { hipLaunchKernelGGL(k_thread, dim3((P + 255) / 256), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    rep("frame-major quads (+ sums written)", time_ms([&] { hipLaunchKernelGGL(k_frame_major, dim3(gq, F), dim3(256), 0, 0, F, P, cap, pair, dgoff, sums); }), used);
    rep("wave streams its 64 Gaussians' run", time_ms([&] { hipLaunchKernelGGL(k_wave_stream, dim3((P + 255) / 256), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); }), used);
    auto walk = [&] { hipLaunchKernelGGL((k_quad<6, 0>), dim3(gq), dim3(256), 0, 0, F, P, cap, pair, dgoff, out); };
#define MODE_ROWS(MODE, label)                                                                                              \
    {                                                                                                                        \
        auto wr = [&] { hipLaunchKernelGGL(k_write<MODE>, dim3(256 * 16), dim3(256), 0, 0, cap, F, pair); };                  \
        rep("writer alone: " label, time_ms(wr), nfl * 4.0);                                                                 \
        rep("  quad walk U=6 right behind it", time_after_ms(wr, walk), used);                                               \
    }
    MODE_ROWS(0, "plain stores");
    MODE_ROWS(1, "__builtin_nontemporal_store");
    MODE_ROWS(2, "sc0 sc1");
    MODE_ROWS(3, "sc0 sc1 nt");
    MODE_ROWS(4, "sc1");
    MODE_ROWS(5, "nt");
    return 0;
}
