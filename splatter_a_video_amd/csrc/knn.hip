// Exact K nearest neighbours on a uniform grid.
// The reference calls pytorch3d.ops.knn_points(points[None], points[None], K=6) on all Gaussians every training step
// (src/geometry_utils.py:17-19, call site src/trainer_fragGS.py:671-675); pytorch3d is an un-vendored CUDA dependency
// with no ROCm build.  Contract reproduced here: squared Euclidean distances of the K nearest points per query in
// ascending order, with their indices (ties: smaller index first).
//
//   K1 knn_bbox     bounding box of the point set (block reduce + order-preserving integer atomics)
//   K2 knn_count    cell of every point (grid of G^3 cells over the box), histogram with global atomics
//   (exclusive scan of the histogram: caller, any device scan)
//   K3 knn_scatter  counting-sort scatter: points (xyz + original id) stored cell by cell
//   K4 knn_search   one thread per query: shells of cells of growing Chebyshev radius r around the query's cell until
//                   K candidates are held and the worst is closer than the nearest possible unvisited point
//                   (r cell widths minus a rounding slack); sorted insertion into K registers
// HBM-bound gather work; queries are expected in (roughly) spatial order for cache locality -- the wrapper passes them
// cell-sorted when query and point set coincide.
#include "common.h"

namespace {

constexpr int KB = 256;
inline dim3 kgrid(size_t n) { return dim3((unsigned)((n + KB - 1) / KB)); }

struct KnnGrid {  // lives in device memory: written by K1, read by the others
    float lo[3], hi[3];
};

__device__ __forceinline__ unsigned f2ord(float f) {  // order-preserving map float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void __launch_bounds__(KB) knn_bbox_kernel(int M, const float *__restrict__ pts, unsigned *__restrict__ box) {
    // box[0..2] = ordered min, box[3..5] = ordered max (initialised by the host to 0xffffffff / 0)
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * KB + threadIdx.x; i < M; i += gridDim.x * KB) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned o = f2ord(pts[3 * i + a]);
            mn[a] = o < mn[a] ? o : mn[a];
            mx[a] = o > mx[a] ? o : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned on = __shfl_xor(mn[a], s), ox = __shfl_xor(mx[a], s);
            mn[a] = on < mn[a] ? on : mn[a];
            mx[a] = ox > mx[a] ? ox : mx[a];
        }
    }
    __shared__ unsigned smn[KB / WAVE][3], smx[KB / WAVE][3];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            smn[w][a] = mn[a];
            smx[w][a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        unsigned lo = smn[0][a], hi = smx[0][a];
        for (int q = 1; q < KB / WAVE; ++q) {
            lo = smn[q][a] < lo ? smn[q][a] : lo;
            hi = smx[q][a] > hi ? smx[q][a] : hi;
        }
        atomicMin(box + a, lo);
        atomicMax(box + 3 + a, hi);
    }
}

// Grid over the bounding box: isotropic cell size h (the stopping rule of the search needs one width), per-axis
// cell counts; chosen on the device by knn_plan_kernel so that no host round trip is needed.
struct KnnPlan {
    float lo[3];
    float h, inv_h;
    int G[3];
};

__global__ void knn_plan_kernel(const unsigned *__restrict__ box, int budget, KnnPlan *__restrict__ plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float e[3], emax = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        plan->lo[a] = ord2f(box[a]);
        e[a] = fmaxf(ord2f(box[3 + a]) - plan->lo[a], 0.f);
        if (!(e[a] < 3.0e38f)) e[a] = 0.f;  // empty set / non-finite input
        emax = fmaxf(emax, e[a]);
    }
    float h = 1.f;
    int G[3] = {1, 1, 1};
    if (emax > 0.f) {
        // smallest cell width whose grid fits the budget (at most 1024 cells per axis): bisection on h
        float lo_h = emax / 1024.f, hi_h = emax;
        for (int it = 0; it < 48; ++it) {
            const float mid = 0.5f * (lo_h + hi_h);
            double cells = 1.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) cells *= (double)fminf(fmaxf(ceilf(e[a] / mid), 1.f), 1024.f);
            if (cells <= (double)budget) hi_h = mid;
            else lo_h = mid;
        }
        h = hi_h;
#pragma unroll
        for (int a = 0; a < 3; ++a) G[a] = (int)fminf(fmaxf(ceilf(e[a] / h), 1.f), 1024.f);
        while ((long long)G[0] * G[1] * G[2] > (long long)budget) {  // rounding at the boundary: coarsen once more
            h *= 1.01f;
#pragma unroll
            for (int a = 0; a < 3; ++a) G[a] = (int)fminf(fmaxf(ceilf(e[a] / h), 1.f), 1024.f);
        }
    }
    plan->h = h;
    plan->inv_h = 1.f / h;
#pragma unroll
    for (int a = 0; a < 3; ++a) plan->G[a] = G[a];
}

struct CellMap {
    float lo[3], h, inv;
    int G[3];
    __device__ __forceinline__ void load(const KnnPlan *p) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = p->lo[a];
            G[a] = p->G[a];
        }
        h = p->h;
        inv = p->inv_h;
    }
    __device__ __forceinline__ int coord(float x, int a) const {
        const int c = (int)floorf((x - lo[a]) * inv);
        return c < 0 ? 0 : (c >= G[a] ? G[a] - 1 : c);
    }
};

__global__ void __launch_bounds__(KB) knn_count_kernel(int M, const float *__restrict__ pts, const KnnPlan *__restrict__ plan,
                                                       int *__restrict__ cell_of, int *__restrict__ count) {
    const int i = blockIdx.x * KB + threadIdx.x;
    if (i >= M) return;
    CellMap cm;
    cm.load(plan);
    const int c = (cm.coord(pts[3 * i + 2], 2) * cm.G[1] + cm.coord(pts[3 * i + 1], 1)) * cm.G[0] + cm.coord(pts[3 * i], 0);
    cell_of[i] = c;
    atomicAdd(count + c, 1);
}

__global__ void __launch_bounds__(KB) knn_scatter_kernel(int M, const float *__restrict__ pts, const int *__restrict__ cell_of,
                                                         const int *__restrict__ cell_start, int *__restrict__ fill,
                                                         float4 *__restrict__ sorted) {
    const int i = blockIdx.x * KB + threadIdx.x;
    if (i >= M) return;
    const int c = cell_of[i];
    const int slot = cell_start[c] + atomicAdd(fill + c, 1);
    sorted[slot] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float(i));
}

// sorted insertion of (d, id) into KT registers; order: smaller d first, equal d -> smaller id first
template <int KT>
__device__ __forceinline__ void knn_insert(float (&bd)[KT], int (&bi)[KT], float d, int id) {
    if (!((d < bd[KT - 1]) || (d == bd[KT - 1] && id < bi[KT - 1]))) return;
    bd[KT - 1] = d;
    bi[KT - 1] = id;
#pragma unroll
    for (int p = KT - 1; p > 0; --p) {
        const bool sw = (bd[p] < bd[p - 1]) || (bd[p] == bd[p - 1] && bi[p] < bi[p - 1]);
        const float td = bd[p];
        const int ti = bi[p];
        bd[p] = sw ? bd[p - 1] : td;
        bi[p] = sw ? bi[p - 1] : ti;
        bd[p - 1] = sw ? td : bd[p - 1];
        bi[p - 1] = sw ? ti : bi[p - 1];
    }
}

template <int KT>
__global__ void __launch_bounds__(KB)
knn_search_kernel(int N, const float *__restrict__ query, const int *__restrict__ qorder, int M,
                  const float4 *__restrict__ sorted, const int *__restrict__ cell_start, const KnnPlan *__restrict__ plan,
                  int K, int rcap, float *__restrict__ dists, int *__restrict__ idx) {
    const int t = blockIdx.x * KB + threadIdx.x;
    if (t >= N) return;
    const int qi = qorder ? qorder[t] : t;  // queries are walked in spatial order, results land at the original index
    CellMap cm;
    cm.load(plan);
    const int Gx = cm.G[0], Gy = cm.G[1], Gz = cm.G[2];
    const float qx = query[3 * qi], qy = query[3 * qi + 1], qz = query[3 * qi + 2];
    const int cx = cm.coord(qx, 0), cy = cm.coord(qy, 1), cz = cm.coord(qz, 2);
    const float hmin = cm.h;
    const float ext = cm.h * (float)imax_(Gx, imax_(Gy, Gz));
    // cell boundaries are monotone thresholds of the coordinate but sit a few ulps off lo + c h; a query outside the
    // box is clamped into a border cell: its true distance to the r-th shell is larger, never smaller
    const float slack = 8e-6f * (ext + fabsf(cm.lo[0]) + fabsf(cm.lo[1]) + fabsf(cm.lo[2]));
    float bd[KT];
    int bi[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        bd[p] = __builtin_inff();
        bi[p] = 0x7fffffff;
    }
    const int want = K < M ? K : M;
    bool done = false;
    for (int r = 0; r <= rcap && !done; ++r) {
        const int z0 = imax_(cz - r, 0), z1 = imin_(cz + r, Gz - 1);
        const int y0 = imax_(cy - r, 0), y1 = imin_(cy + r, Gy - 1);
        const int x0 = imax_(cx - r, 0), x1 = imin_(cx + r, Gx - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool face = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                // rows on a face of the shell are scanned whole (contiguous cells along x); interior rows only at their ends
                for (int side = 0; side < (face ? 1 : 2); ++side) {
                    int xa, xb;
                    if (face) { xa = x0; xb = x1; }
                    else {
                        const int x = side == 0 ? cx - r : cx + r;
                        if (x < 0 || x >= Gx || (side == 1 && r == 0)) continue;
                        xa = xb = x;
                    }
                    const int row = (z * Gy + y) * Gx;
                    const int beg = cell_start[row + xa], end = cell_start[row + xb + 1];
                    for (int j = beg; j < end; ++j) {
                        const float4 p = sorted[j];
                        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
                        knn_insert<KT>(bd, bi, dx * dx + dy * dy + dz * dz, __float_as_int(p.w));
                    }
                }
            }
        const float reach = fmaxf((float)r * hmin - slack, 0.f);
        float kth = 0.f;  // distance of the want-th candidate (+inf while fewer are held); static register indices only
#pragma unroll
        for (int p = 0; p < KT; ++p) kth = (p == want - 1) ? bd[p] : kth;
        done = kth <= reach * reach;
        if ((x0 == 0 && x1 == Gx - 1) && (y0 == 0 && y1 == Gy - 1) && (z0 == 0 && z1 == Gz - 1)) done = true;  // whole grid seen
    }
    if (!done) {  // sparse neighbourhood: exhaustive scan (exact, rare)
#pragma unroll
        for (int p = 0; p < KT; ++p) {
            bd[p] = __builtin_inff();
            bi[p] = 0x7fffffff;
        }
        for (int j = 0; j < M; ++j) {
            const float4 p = sorted[j];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            knn_insert<KT>(bd, bi, dx * dx + dy * dy + dz * dz, __float_as_int(p.w));
        }
    }
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        if (p < K) {
            const bool have = bi[p] != 0x7fffffff;
            dists[(size_t)qi * K + p] = have ? bd[p] : 0.f;  // fewer than K points: 0 / -1 padding as knn_points
            idx[(size_t)qi * K + p] = have ? bi[p] : -1;
        }
    }
}


// ---- K nearest neighbours of a FEW query vertices in each of B point sets, brute force (the ARAP term of a training batch
// needs the neighbours of its 512 sampled vertices only, src/geometry_utils.py:90-123 with :17-19; a grid build per point set
// costs more than scanning the set for 512 queries).  Block = 256 queries x one chunk of the points, staged through LDS in
// tiles of KNB_TILE (every lane reads the same point: a broadcast); partial top-KT lists per chunk, merged by knn_brute_merge.
// Exact; ties -> smaller index (knn_insert), the arithmetic of knn_search_kernel.
constexpr int KNB_TILE = 1024, KNB_TILES = 4, KNB_CHUNK = KNB_TILE * KNB_TILES;

// Bound pass: an upper bound of the query's K-th neighbour distance from the KNB_WIN points around its own index (the Gaussians
// are kept in Morton order of their screen positions, densify.spatial_order: index neighbours are mostly space neighbours; any K
// points give a valid bound).  The scan then inserts only candidates within the bound -- a fresh list per (query, chunk) would
// take ~8 ln(chunk / 8) insertions per thread, and a wave runs the insertion whenever ANY of its 64 queries inserts: three
// times the work of the distance loop itself.
constexpr int KNB_WIN = 128;
// A WAVE per query (round 6; a thread per query walked its 128 window points one insertion after the other: 100 us of latency at
// 12 800 threads): lane l holds the distances of window points l and 64 + l, the K-th smallest of the 128 is extracted by K
// rounds of wave minimum + removal of ONE instance of it.
__device__ __forceinline__ float knn_wave_min(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    return m;
}
template <int KT>
__global__ void __launch_bounds__(KB)
knn_brute_bound_kernel(int N, int S, int K, const float *__restrict__ pts, long long pts_bs, const long long *__restrict__ qidx,
                       float *__restrict__ tau) {
    static_assert(KNB_WIN == 2 * WAVE, "two window points per lane");
    const int lane = threadIdx.x & 63, q = blockIdx.x * (KB / WAVE) + (threadIdx.x >> 6), b = blockIdx.y;
    if (q >= S) return;   // (wave-uniform)
    const float *P = pts + (size_t)b * pts_bs;
    const int v = imin_(imax_((int)qidx[(size_t)b * S + q], 0), N - 1);   // memory safety: ids are clamped into the set
    const float qx = P[3 * (size_t)v], qy = P[3 * (size_t)v + 1], qz = P[3 * (size_t)v + 2];
    const int lo = imax_(0, imin_(v - KNB_WIN / 2, N - KNB_WIN)), hi = imin_(N, lo + KNB_WIN);
    float d[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = lo + WAVE * j + lane;
        d[j] = __builtin_inff();
        if (id < hi) {
            const float dx = qx - P[3 * (size_t)id], dy = qy - P[3 * (size_t)id + 1], dz = qz - P[3 * (size_t)id + 2];
            d[j] = dx * dx + dy * dy + dz * dz;
        }
    }
    float kth = __builtin_inff();   // fewer than K points in the window (tiny sets): no bound
    for (int r = 0; r < K; ++r) {
        kth = knn_wave_min(fminf(d[0], d[1]));
        const unsigned long long has = __builtin_amdgcn_ballot_w64(d[0] == kth || d[1] == kth);
        if (has == 0ull) break;   // (NaN input: no bound either way)
        if (lane == __builtin_ctzll(has)) {   // one instance leaves (equal distances count separately)
            if (d[0] == kth) d[0] = __builtin_inff();
            else d[1] = __builtin_inff();
        }
    }
    if (lane == 0) tau[(size_t)b * S + q] = kth * 1.000002f;   // (the scan may round the same distance an ulp differently: keep the bound's own points)
}

// bounding box of every tile of KNB_TILE consecutive points (Morton order: compact boxes): box[(b * NT + t) * 6 + {lo xyz, hi xyz}]
__global__ void __launch_bounds__(KB)
knn_tile_box_kernel(int N, int NT, const float *__restrict__ pts, long long pts_bs, float *__restrict__ box) {
    const int t = blockIdx.x, b = blockIdx.y;
    const float *P = pts + (size_t)b * pts_bs;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int e = threadIdx.x; e < KNB_TILE; e += KB) {
        const int id = t * KNB_TILE + e;
        if (id < N) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = P[3 * (size_t)id + a];
                lo[a] = fminf(lo[a], v);
                hi[a] = fmaxf(hi[a], v);
            }
        }
    }
    __shared__ float s_lo[KB / WAVE][3], s_hi[KB / WAVE][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float l = s_lo[0][a], h = s_hi[0][a];
        for (int w = 1; w < KB / WAVE; ++w) { l = fminf(l, s_lo[w][a]); h = fmaxf(h, s_hi[w][a]); }
        box[((size_t)b * NT + t) * 6 + a] = l;
        box[((size_t)b * NT + t) * 6 + 3 + a] = h;
    }
}

// Partial lists: a WAVE = 64 queries x one chunk of KNB_TILES tiles; a wave scans a tile only if one of its queries can have a
// candidate in the tile's box -- the callers hand the queries in ascending index = Morton order, so a wave's queries are
// neighbours and four tiles of five are far from all of them.  The waves of a workgroup share nothing: the tile's points are
// read through the SCALAR cache (their address is wave-uniform), eight points per trip -- no LDS staging, no barrier (with a
// tile staged in LDS by the workgroup a tile was scanned whenever ANY of its 256 queries needed it, and the skipping waves
// waited at the barriers; a wave staging for itself ran at a third of the occupancy).  Only non-empty lists are written:
// cnt[(b * G + c) * S + q] = entries of the list at pd / pi[.. * KT].
template <int KT>
__global__ void __launch_bounds__(KB)
knn_brute_partial_kernel(int N, int S, int G, int NT, const float *__restrict__ pts, long long pts_bs, const long long *__restrict__ qidx,
                         const float *__restrict__ tau, const float *__restrict__ box, float *__restrict__ pd, int *__restrict__ pi,
                         unsigned char *__restrict__ cnt) {
    const int c = blockIdx.x, b = blockIdx.z;
    const int q = blockIdx.y * KB + threadIdx.x;
    const float *__restrict__ P = pts + (size_t)b * pts_bs;
    float qx = 0.f, qy = 0.f, qz = 0.f, bound = -1.f;   // (a lane past the queries takes no candidate)
    if (q < S) {
        const long long v = imin_(imax_((int)qidx[(size_t)b * S + q], 0), N - 1);
        qx = P[3 * v]; qy = P[3 * v + 1]; qz = P[3 * v + 2];
        bound = tau[(size_t)b * S + q];
    }
    float bd[KT];
    int bi[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) { bd[p] = __builtin_inff(); bi[p] = 0x7fffffff; }
    for (int tl = 0; tl < KNB_TILES; ++tl) {
        const int tile = c * KNB_TILES + tl;
        const int base = tile * KNB_TILE;
        if (base >= N) break;   // uniform
        const float *bx = box + ((size_t)b * NT + tile) * 6;
        const float ex = fmaxf(fmaxf(bx[0] - qx, qx - bx[3]), 0.f), ey = fmaxf(fmaxf(bx[1] - qy, qy - bx[4]), 0.f),
                    ez = fmaxf(fmaxf(bx[2] - qz, qz - bx[5]), 0.f);
        // (the box distance is a lower bound of every point's distance up to rounding: compare against an inflated bound)
        if (__builtin_amdgcn_ballot_w64((ex * ex + ey * ey + ez * ez) * 0.999998f <= bound) == 0ull) continue;   // wave-uniform
        const int n = imin_(KNB_TILE, N - base);
        const float *__restrict__ T = P + 3 * (size_t)base;
        int e = 0;
        for (; e + 8 <= n; e += 8) {   // 24 consecutive floats per trip: scalar loads
            float v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = T[3 * e + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dx = qx - v[3 * k], dy = qy - v[3 * k + 1], dz = qz - v[3 * k + 2];
                const float d = dx * dx + dy * dy + dz * dz;
                if (d <= bound) knn_insert<KT>(bd, bi, d, base + e + k);   // the K nearest all lie within the bound
            }
        }
        for (; e < n; ++e) {
            const float dx = qx - T[3 * e], dy = qy - T[3 * e + 1], dz = qz - T[3 * e + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d <= bound) knn_insert<KT>(bd, bi, d, base + e);
        }
    }
    if (q < S) {
        const size_t o = ((size_t)b * G + c) * S + q;
        int nv = 0;
#pragma unroll
        for (int p = 0; p < KT; ++p) nv += bi[p] != 0x7fffffff ? 1 : 0;
        cnt[o] = (unsigned char)nv;
#pragma unroll
        for (int p = 0; p < KT; ++p)
            if (p < nv) { pd[o * KT + p] = bd[p]; pi[o * KT + p] = bi[p]; }
    }
}

// Merge of the sparse partial lists: a WAVE per query (round 6; a thread per query walked its G chunks' counters one dependent
// load after the other).  Lane l gathers the lists of chunks l, l + 64, .. into a private sorted list (almost always empty), then K
// rounds of a lexicographic wave minimum over the lanes' heads pop the result in order (ties -> smaller index, as knn_insert).
template <int KT>
__global__ void __launch_bounds__(KB)
knn_brute_merge_kernel(int S, int G, int K, const float *__restrict__ pd, const int *__restrict__ pi,
                       const unsigned char *__restrict__ cnt, float *__restrict__ dists, int *__restrict__ idx) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * (KB / WAVE) + (threadIdx.x >> 6), b = blockIdx.y;
    if (q >= S) return;   // (wave-uniform)
    float bd[KT];
    int bi[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) { bd[p] = __builtin_inff(); bi[p] = 0x7fffffff; }
    for (int c = lane; c < G; c += WAVE) {
        const size_t o = ((size_t)b * G + c) * S + q;
        const int nv = cnt[o];
        for (int p = 0; p < nv; ++p) knn_insert<KT>(bd, bi, pd[o * KT + p], pi[o * KT + p]);
    }
    float *od = dists + ((size_t)b * S + q) * K;
    int *oi = idx + ((size_t)b * S + q) * K;
    for (int r = 0; r < K; ++r) {
        float hd = bd[0];
        int hi = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float xd = __shfl_xor(hd, o);
            const int xi = __shfl_xor(hi, o);
            const bool take = (xd < hd) || (xd == hd && xi < hi);
            hd = take ? xd : hd;
            hi = take ? xi : hi;
        }
        const bool have = hi != 0x7fffffff;
        if (have && bi[0] == hi) {   // the owner pops its head (point ids are unique across the lanes' lists)
#pragma unroll
            for (int p = 0; p + 1 < KT; ++p) { bd[p] = bd[p + 1]; bi[p] = bi[p + 1]; }
            bd[KT - 1] = __builtin_inff();
            bi[KT - 1] = 0x7fffffff;
        }
        if (lane == 0) {
            od[r] = have ? hd : 0.f;   // fewer than K points: 0 / -1 padding as knn_points
            oi[r] = have ? hi : -1;
        }
    }
}
}  // namespace

extern "C" int splat_knn_grid_cells(int M) {
    // cell budget of the grid: about one cell per two points (a couple of points per occupied cell), 64 .. 2^22
    long long c = (long long)M / 2;
    if (c < 64) c = 64;
    if (c > (1ll << 22)) c = 1ll << 22;
    return (int)c;
}

extern "C" size_t splat_knn_plan_bytes(void) { return 256; }

extern "C" int splat_knn_build(int M, const float *points, int budget, void *plan /*splat_knn_plan_bytes()*/,
                               int32_t *cell_of /*[M]*/, int32_t *cell_count /*[budget + 1], zero-filled*/, void *stream) {
    SPLAT_CHECK_ARG(M >= 0 && budget >= 1, "bad sizes");
    SPLAT_CHECK_ARG(plan && cell_count, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned *box = (unsigned *)plan + 32;  // second half of the plan block: ordered-int bounds
    const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    SPLAT_CHECK_HIP(hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, s));
    if (M > 0) {
        SPLAT_CHECK_ARG(points && cell_of, "null pointer");
        const int nb = (M + KB - 1) / KB;
        SPLAT_LAUNCH("knn_bbox", knn_bbox_kernel, dim3(nb < 256 ? nb : 256), dim3(KB), 0, s, M, points, box);
        SPLAT_POST_LAUNCH();
    }
    SPLAT_LAUNCH("knn_plan", knn_plan_kernel, dim3(1), dim3(64), 0, s, box, budget, (KnnPlan *)plan);
    SPLAT_POST_LAUNCH();
    if (M > 0) {
        SPLAT_LAUNCH("knn_count", knn_count_kernel, kgrid(M), dim3(KB), 0, s, M, points, (const KnnPlan *)plan, cell_of,
                     cell_count);
        SPLAT_POST_LAUNCH();
    }
    return SPLAT_OK;
}

extern "C" int splat_knn_scatter(int M, const float *points, const int32_t *cell_of, const int32_t *cell_start,
                                 int32_t *fill /*[budget], zero-filled*/, float *sorted /*[M,4]*/, void *stream) {
    SPLAT_CHECK_ARG(M >= 0, "bad size");
    if (M == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(points && cell_of && cell_start && fill && sorted, "null pointer");
    SPLAT_LAUNCH("knn_scatter", knn_scatter_kernel, kgrid(M), dim3(KB), 0, (hipStream_t)stream, M, points, cell_of, cell_start,
                 fill, (float4 *)sorted);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_knn_search(int N, const float *query, const int32_t *query_order, int M, const float *sorted,
                                const int32_t *cell_start, const void *plan, int K, float *dists, int32_t *idx,
                                void *stream) {
    SPLAT_CHECK_ARG(N >= 0 && M >= 0, "bad sizes");
    SPLAT_CHECK_ARG(K >= 1 && K <= 16, "K must be 1..16");
    if (N == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(query && dists && idx && plan && cell_start && (M == 0 || sorted), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rcap = 8;
    if (K <= 8)
        SPLAT_LAUNCH("knn_search", knn_search_kernel<8>, kgrid(N), dim3(KB), 0, s, N, query, query_order, M,
                     (const float4 *)sorted, cell_start, (const KnnPlan *)plan, K, rcap, dists, idx);
    else
        SPLAT_LAUNCH("knn_search", knn_search_kernel<16>, kgrid(N), dim3(KB), 0, s, N, query, query_order, M,
                     (const float4 *)sorted, cell_start, (const KnnPlan *)plan, K, rcap, dists, idx);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// K nearest points of S query VERTICES (query_idx [B, S] int64: indices into the set itself) among the N points of each of B
// point sets (set b at points + b * points_batch_stride floats), brute force: dists / idx [B, S, K] ascending, ties -> smaller
// index, the query itself included (distance 0) like knn_points(points, points).  scratch: splat_knn_brute_scratch_bytes().
extern "C" size_t splat_knn_brute_scratch_bytes(int B, int N, int S) {
    const size_t n = (size_t)(N > 0 ? N : 1);
    const size_t G = (n + KNB_CHUNK - 1) / KNB_CHUNK, NT = (n + KNB_TILE - 1) / KNB_TILE;
    const size_t bs = (size_t)(B > 0 ? B : 1) * (size_t)(S > 0 ? S : 1);
    // partial lists + the bound of every query + the tiles' boxes + the lists' lengths
    return bs * G * 8 * (sizeof(float) + sizeof(int)) + bs * sizeof(float) + (size_t)(B > 0 ? B : 1) * NT * 6 * sizeof(float) + bs * G + 16;
}

extern "C" int splat_knn_brute_batch(int B, int N, int S, int K, const float *points, int64_t points_batch_stride,
                                     const int64_t *query_idx, float *dists, int32_t *idx, void *scratch, void *stream) {
    SPLAT_CHECK_ARG(B >= 1 && B <= 65535 && N >= 1 && S >= 0, "bad sizes");
    SPLAT_CHECK_ARG(K >= 1 && K <= 8, "K must be 1..8");
    if (S == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(points && query_idx && dists && idx && scratch, "null pointer");
    SPLAT_CHECK_ARG(points_batch_stride >= (int64_t)N * 3 || B == 1, "points_batch_stride below N * 3");
    hipStream_t s = (hipStream_t)stream;
    const int G = (N + KNB_CHUNK - 1) / KNB_CHUNK;
    float *pd = (float *)scratch;
    int *pi = (int *)(pd + (size_t)B * G * S * 8);
    float *tau = (float *)(pi + (size_t)B * G * S * 8);
    const int NT = (N + KNB_TILE - 1) / KNB_TILE;
    float *box = tau + (size_t)B * S;
    unsigned char *cnt = (unsigned char *)(box + (size_t)B * NT * 6);
    SPLAT_LAUNCH("knn_brute_box", knn_tile_box_kernel, dim3((unsigned)NT, (unsigned)B), dim3(KB), 0, s, N, NT, points,
                 (long long)points_batch_stride, box);
    SPLAT_POST_LAUNCH();
    const unsigned wq = (unsigned)((S + KB / WAVE - 1) / (KB / WAVE));   // a wave per query
    SPLAT_LAUNCH("knn_brute_bound", knn_brute_bound_kernel<8>, dim3(wq, (unsigned)B), dim3(KB), 0, s, N, S, K,
                 points, (long long)points_batch_stride, (const long long *)query_idx, tau);
    SPLAT_POST_LAUNCH();
    SPLAT_LAUNCH("knn_brute", knn_brute_partial_kernel<8>, dim3((unsigned)G, (unsigned)((S + KB - 1) / KB), (unsigned)B), dim3(KB), 0, s,
                 N, S, G, NT, points, (long long)points_batch_stride, (const long long *)query_idx, tau, box, pd, pi, cnt);
    SPLAT_POST_LAUNCH();
    SPLAT_LAUNCH("knn_brute_merge", knn_brute_merge_kernel<8>, dim3(wq, (unsigned)B), dim3(KB), 0, s, S, G, K,
                 pd, pi, cnt, dists, idx);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
