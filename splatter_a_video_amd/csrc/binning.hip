// Tile binning + per-tile depth sort (the reference's sort_gaussian op).
//
// The reference emits one 64-bit key (tile << 32 | depth bits) per (Gaussian, tile) pair and runs
// a global 64-bit radix sort (torch.sort) over all M pairs (dptr/gs/sort_gaussian.py:42-52,
// src/sort_gaussian.cu:16-70).  Here the tile id never enters a sort key:
//
//   K1 bin_count    each workgroup owns a contiguous chunk of Gaussians and histograms the tiles
//                   they touch in LDS (ds_add, no global atomics), then stores its row of the
//                   [NB, T] count matrix;
//   K2a bin_colscan exclusive scan of every tile's column over the NB chunks (32 columns x 32 row groups per
//                   workgroup, rows held in registers), column total -> tile_count[T];
//   K2b bin_tilescan single workgroup: exclusive scan of tile_count -> tile_range[T,2], M;
//   K3 bin_scatter  same chunking as K1; LDS counters start at the chunk's base, ds_add_rtn
//                   hands out the slot; writes key = (depth bits << 32 | gaussian id) -- pair-map mode: the low word
//                   is (id << kbits | index of the tile in the splat's rectangle), or the pair slot (pair_key_kbits);
//   K4 tile_sort    one workgroup per tile: all-ascending bitonic network over the tile's keys in registers (<= 2048
//                   keys; waves whose elements all lie past the list idle; crowded tiles: 2048-key blocks in registers +
//                   the strides of whole blocks through the tile's global segment), writes ids (and, in pair-map mode,
//                   the Gaussian-major pair slot of every sorted entry).
//
// Since keys inside a tile are unique (id in the low word) the result is deterministic and equals
// a STABLE sort of the reference keys (ties: ascending Gaussian id).  No host synchronisation.
#include "common.h"
#include <cstdlib>

#define BIN_BLOCK 256
#define BIN_MAX_NB 512          // rows of the count matrix (= workgroups of K1/K3)
#ifndef BIN_CHUNK
#define BIN_CHUNK 512           // Gaussians per row (chunk) before BIN_MAX_NB caps the row count
#endif
#ifndef BIN_CHUNK_BATCH
#define BIN_CHUNK_BATCH 2048    // ... in a frame batch of at least BIN_BATCH_FRAMES frames (see make_plan)
#endif
#define BIN_BATCH_FRAMES 4
#define BIN_LDS_TILES 12288     // <= 48 KB of LDS counters; larger tile grids use global atomics
#define BIN_GLOBAL_BLOCKS 2048   // grid of K1/K3 on the global-atomic path
#define SORT_BLOCK 256
#ifndef SORT_SWIZZLE_ALL
#define SORT_SWIZZLE_ALL 0
#endif
#ifndef SORT_SWIZZLE4
#define SORT_SWIZZLE4 1         // thread distance 4: ds_swizzle (one LDS-crossbar instruction per word) instead of two DPP moves + a select
#endif
#define SORT_LDS_KEYS 2048      // 16 KB of LDS per sort workgroup: the exchange buffer of the two widest strides of a 2048-key block

struct BinPlan {
    int T, gx, gy;
    int NB;        // number of Gaussian chunks (= workgroups of K1/K3)
    int chunk;     // Gaussians per chunk
    bool lds;      // LDS histogram path
    int nchunk;    // workgroups of K1/K3 on the path in use (NB, or the global-atomic path's own grid)
    size_t off_matrix, off_tilecount, off_total, off_chunksum;  // byte offsets inside scratch
    size_t bytes;
};

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

// F: frames of the launch.  The scratch LAYOUT (offsets, bytes per frame: what splat_bin_scratch_bytes reports) is that of the
// single-frame plan whatever F is; a frame batch (F >= BIN_BATCH_FRAMES) only uses FEWER, larger chunks of it: with grid.y = F
// there are workgroups enough, and every row less is 2 T counters less for bin_count to write, bin_colscan to scan and
// bin_scatter's workgroups to load (c2, 25 frames: count + scans + scatter 14.0 -> 11.8 us per frame; a single frame needs the
// small chunks: 512 workgroups are all it has).
static int bits_for(long long n) {  // bits that hold 0 .. n - 1
    int b = 1;
    while ((1ll << b) < n) ++b;
    return b;
}

// Pair-map keys.  The low key word has to order a tile's equal depths by ascending Gaussian id and to lead the sort to both
// the Gaussian and its pair slot.  BIN_PACKED_KEYS: the word is (id << kbits) | k, k = index of the tile inside the splat's
// rectangle (k < T): the sort reads the slot as goff_excl[id] + k from the per-Gaussian prefix the scatter writes anyway
// (4 P bytes per frame, L2-resident) -- no `owner` array written per pair by the scatter and gathered per pair, at sector
// granularity, by the sort.  When id and k do not fit 32 bits (1M Gaussians on more than 4096 tiles) the word is the pair
// slot and `owner` resolves the id as before.  Both forms sort identically (slots grow with (id, k)).
#ifndef BIN_PACKED_KEYS
#define BIN_PACKED_KEYS 1
#endif
static int pair_key_kbits(int P, int T) {
    if (!BIN_PACKED_KEYS) return 0;
    const bool slot_keys = splat_option(SPLAT_OPT_BIN_SLOT_KEYS) != 0;   // the slot form everywhere -- how the tests reach it at small sizes
    if (slot_keys) return 0;
    const int kb = bits_for(T), ib = bits_for(P);
    return kb + ib <= 32 ? kb : 0;
}

static BinPlan make_plan(int P, int W, int H, int F = 1) {
    BinPlan p;
    p.gx = (W + TILE - 1) / TILE;
    p.gy = (H + TILE - 1) / TILE;
    p.T = p.gx * p.gy;
    p.lds = p.T <= BIN_LDS_TILES;
    int nb = (P + BIN_CHUNK - 1) / BIN_CHUNK;
    if (nb < 1) nb = 1;
    if (nb > BIN_MAX_NB) nb = BIN_MAX_NB;
    if (!p.lds) nb = 1;  // global-atomic path keeps a single row
    const int nb_layout = nb;
    if (p.lds && F >= BIN_BATCH_FRAMES) {
        int nbb = (P + BIN_CHUNK_BATCH - 1) / BIN_CHUNK_BATCH;
        if (nbb < 1) nbb = 1;
        if (nbb < nb) nb = nbb;
    }
    p.NB = nb;
    p.chunk = (P + nb - 1) / nb;
    if (p.chunk < 1) p.chunk = 1;
    size_t o = 0;
    p.off_matrix = o; o += (size_t)(nb_layout < 2 ? 2 : nb_layout) * p.T * sizeof(int);  // global path: [counts, fill]
    o = (o + 255) & ~(size_t)255;
    p.off_tilecount = o; o += (size_t)p.T * sizeof(int);
    o = (o + 255) & ~(size_t)255;
    p.off_total = o; o += 256;
    p.nchunk = p.lds ? p.NB : imax(1, imin((P + BIN_BLOCK - 1) / BIN_BLOCK, BIN_GLOBAL_BLOCKS));
    const int nchunk_layout = p.lds ? nb_layout : p.nchunk;
    p.off_chunksum = o; o += (size_t)(nchunk_layout + 1) * sizeof(int);  // pairs per chunk -> exclusive chunk offsets
    o = (o + 255) & ~(size_t)255;
    p.bytes = o;
    return p;
}

// ------------------------------------------------------------------ reach masks (pairs that cannot contribute are never created)
// The reference creates a (Gaussian, tile) pair for every tile of the splat's bounding SQUARE (radius = 3 sigma of the major
// axis; include/utils.h:17-37) and lets alpha_blending skip, per pixel, whatever stays below alpha = 1/255
// (src/alpha_blending.cu:78-95).  On the bench scene a third of those pairs (33.7 %, tools/cull_model.py: "zero") reaches no
// pixel centre of their tile at all: they are sorted, staged and culled by both compositing kernels, get a zero pair record
// written and read again -- for nothing.  Callers that own the whole chain (frame batches, the fused per-frame operators; NOT
// the reference's sort_gaussian operator, whose idx_sorted / tile_range are its results) hand conic and opacity to the count
// kernel, which runs the compositing kernels' own conservative test (cull_test, common.h) on the tile's rectangle of pixel
// centres and leaves one word per Gaussian for the scatter kernel:
//     bit 31 clear: bits 0 .. 30 = keep flags of the rectangle's tiles in row-major order (rectangles of at most 31 tiles);
//     bit 31 set:   a larger rectangle (radius above ~35 px) is cut into CELLS of c x c tiles, c the smallest power of two that
//                   leaves at most 31 cells (an integer function of the rectangle: reach_cell_shift); bits 0 .. 30 = keep flags of
//                   the cells in row-major order -- a cell is tested as ONE rectangle of pixel centres and keeps or drops all its
//                   tiles (a splat of any size is culled; the corner cells of a large elongated splat's square are most of it).
// The scatter kernel only reads the word (and counts the kept tiles of a large rectangle from the word and the rectangle, in
// integers): the two kernels cannot disagree about a pair.  (A test repeated in the scatter kernel through a shared non-inlined
// function was the first form for large rectangles, but a device function call gives both kernels a stack: single-frame launches
// 9.5 -> 40 us and 22 -> 45 us.)
// A dropped pair holds no pixel with alpha >= 1/255, so images, ids and gradients are those of the full list bit for bit; only
// list positions (ncontrib) and M change.
#define REACH_BIG 0x80000000u
// log2 of the cell edge (in tiles) of a large rectangle of w x h tiles: the smallest cell that leaves at most 31 cells
__device__ __forceinline__ int reach_cell_shift(int w, int h) {
    int sh = 1;
    while ((((w - 1) >> sh) + 1) * (((h - 1) >> sh) + 1) > 31) ++sh;
    return sh;
}
// kept tiles of a large rectangle: the clipped areas of the cells whose bit is set
__device__ __forceinline__ int reach_big_count(unsigned word, int w, int h) {
    const int sh = reach_cell_shift(w, h), c = 1 << sh, ncx = ((w - 1) >> sh) + 1, ncy = ((h - 1) >> sh) + 1;
    int n = 0;
    unsigned bit = 1u;
    for (int cy = 0; cy < ncy; ++cy)
        for (int cx = 0; cx < ncx; ++cx, bit <<= 1)
            if (word & bit) n += imin_(c, w - cx * c) * imin_(c, h - cy * c);
    return n;
}

// ------------------------------------------------------------------ K1
template <bool LDS>
__global__ void __launch_bounds__(BIN_BLOCK)
bin_count_kernel(int P, const float2 *__restrict__ uv, const int *__restrict__ radius, int gx, int gy, int T,
                 int chunk, int *__restrict__ matrix, int *__restrict__ gcount, int *__restrict__ chunk_sum,
                 long long S, const float *__restrict__ conic, const float *__restrict__ opacity, long long opacity_fs,
                 unsigned *__restrict__ reach) {
    extern __shared__ __attribute__((aligned(16))) int cnt[];
    __shared__ int wave_pairs[BIN_BLOCK / 64];
    const int wg = blockIdx.x;
    {   // frame batch: blockIdx.y = frame; per-Gaussian arrays are [F,P,..], scratch blocks S ints apart
        const size_t f = blockIdx.y;
        uv += f * P; radius += f * P; matrix += f * S; chunk_sum += f * S;
        if (gcount) gcount += f * P;
        if (reach) { conic += f * 3 * (size_t)P; opacity += f * (size_t)opacity_fs; reach += f * P; }
    }
    int pairs = 0;  // this thread's share of the chunk's pair count
    if (LDS) {
        for (int t = threadIdx.x; t < T; t += BIN_BLOCK) cnt[t] = 0;
        __syncthreads();
    }
    const int beg = wg * chunk, end = imin_(P, beg + chunk);
    for (int i = beg + threadIdx.x; i < end; i += BIN_BLOCK) {
        const int r = radius[i];
        if (r <= 0) {
            if (gcount) gcount[i] = 0;
            if (reach) reach[i] = 0u;
            continue;
        }
        const float2 q = uv[i];
        int x0, y0, x1, y1;
        tile_rect(q.x, q.y, r, gx, gy, x0, y0, x1, y1);
        const int area = (x1 - x0) * (y1 - y0);
        if (reach) {   // only the tiles the splat can reach with alpha >= 1/255 (see "reach masks")
            const float cA = conic[3 * (size_t)i], cB = conic[3 * (size_t)i + 1], cC = conic[3 * (size_t)i + 2], o = opacity[i];
            unsigned word;
            int kept = 0;
            if (area < 32) {
                const CullP cp = cull_params(cA, cB, cC, o);
                unsigned m = 0u, bit = 1u;
                for (int ty = y0; ty < y1; ++ty)
                    for (int tx = x0; tx < x1; ++tx, bit <<= 1) {
                        const float fx = (float)(tx * TILE), fy = (float)(ty * TILE);
                        if (cull_test(q.x, q.y, cA, cB, cC, cp, fx, fx + (float)(TILE - 1), fy, fy + (float)(TILE - 1))) {
                            m |= bit;
                            if (LDS) atomicAdd(&cnt[ty * gx + tx], 1);
                            else atomicAdd(&matrix[ty * gx + tx], 1);
                        }
                    }
                kept = __popc(m);
                word = m;
            } else {   // a large rectangle: cells of c x c tiles, each tested as one rectangle of pixel centres
                const CullP cp = cull_params(cA, cB, cC, o);
                const int w = x1 - x0, h = y1 - y0, sh = reach_cell_shift(w, h), c = 1 << sh;
                const int ncx = ((w - 1) >> sh) + 1, ncy = ((h - 1) >> sh) + 1;
                unsigned m = 0u, bit = 1u;
                for (int cy = 0; cy < ncy; ++cy)
                    for (int cx = 0; cx < ncx; ++cx, bit <<= 1) {
                        const int tx0 = x0 + cx * c, ty0 = y0 + cy * c, tx1 = imin_(x1, tx0 + c), ty1 = imin_(y1, ty0 + c);
                        if (cull_test(q.x, q.y, cA, cB, cC, cp, (float)(tx0 * TILE), (float)(tx1 * TILE - 1), (float)(ty0 * TILE),
                                      (float)(ty1 * TILE - 1))) {
                            m |= bit;
                            kept += (tx1 - tx0) * (ty1 - ty0);
                            for (int ty = ty0; ty < ty1; ++ty)
                                for (int tx = tx0; tx < tx1; ++tx) {
                                    if (LDS) atomicAdd(&cnt[ty * gx + tx], 1);
                                    else atomicAdd(&matrix[ty * gx + tx], 1);
                                }
                        }
                    }
                word = REACH_BIG | m;
            }
            reach[i] = word;
            if (gcount) gcount[i] = kept;
            pairs += kept;
            continue;
        }
        if (gcount) gcount[i] = area;
        pairs += area;
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                if (LDS) atomicAdd(&cnt[ty * gx + tx], 1);
                else atomicAdd(&matrix[ty * gx + tx], 1);
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pairs += __shfl_xor(pairs, o);
    if ((threadIdx.x & 63) == 0) wave_pairs[threadIdx.x >> 6] = pairs;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < BIN_BLOCK / 64; ++k) s += wave_pairs[k];
        chunk_sum[wg] = s;
    }
    if (LDS) {
        int *row = matrix + (size_t)wg * T;
        for (int t = threadIdx.x; t < T; t += BIN_BLOCK) row[t] = cnt[t];
    }
}

// ------------------------------------------------------------------ K2a: one wave per tile column
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}

// 32 tile columns x 32 row groups per workgroup: every thread loads its (at most 16) rows of one column in one
// batch -- a wave reads two 128-byte row segments per instruction --, scans them in registers, and the group totals
// are exchanged through LDS.  The matrix is read once and written once.
#ifndef COLSCAN_COLS
#define COLSCAN_COLS 32
#endif
#define COLSCAN_GROUPS (1024 / COLSCAN_COLS)
#define COLSCAN_ROWS (BIN_MAX_NB / COLSCAN_GROUPS)
__global__ void __launch_bounds__(COLSCAN_COLS * COLSCAN_GROUPS)
bin_colscan_kernel(int T, int NB, int *__restrict__ matrix, int *__restrict__ tile_count, long long S) {
    __shared__ int gsum[COLSCAN_GROUPS][COLSCAN_COLS + 1];
    matrix += (size_t)blockIdx.y * S; tile_count += (size_t)blockIdx.y * S;
    const int c = threadIdx.x & (COLSCAN_COLS - 1), g = threadIdx.x / COLSCAN_COLS;
    const int t = blockIdx.x * COLSCAN_COLS + c;
    const int rpg = (NB + COLSCAN_GROUPS - 1) / COLSCAN_GROUPS;  // rows per group, <= COLSCAN_ROWS
    const int r0 = g * rpg;
    int v[COLSCAN_ROWS];
#pragma unroll
    for (int k = 0; k < COLSCAN_ROWS; ++k) {
        const int r = r0 + k;
        v[k] = (k < rpg && r < NB && t < T) ? matrix[(size_t)r * T + t] : 0;
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < COLSCAN_ROWS; ++k) {  // exclusive inside the group
        const int x = v[k];
        v[k] = s;
        s += x;
    }
    gsum[g][c] = s;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < g; ++k) off += gsum[k][c];
#pragma unroll
    for (int k = 0; k < COLSCAN_ROWS; ++k) {
        const int r = r0 + k;
        if (k < rpg && r < NB && t < T) matrix[(size_t)r * T + t] = off + v[k];
    }
    if (g == COLSCAN_GROUPS - 1 && t < T) tile_count[t] = off + s;
}

// ------------------------------------------------------------------ K2b: scan over tiles (single workgroup)
__global__ void __launch_bounds__(1024)
bin_tilescan_kernel(int T, const int *__restrict__ tile_count, int *__restrict__ tile_range, int *__restrict__ M_out,
                    int *__restrict__ total_scratch, int nchunk, int *__restrict__ chunk_sum, long long S) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    {
        const size_t f = blockIdx.y;
        tile_count += f * S; total_scratch += f * S; chunk_sum += f * S;
        tile_range += f * 2 * T;
        if (M_out) M_out += f;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int t = base + threadIdx.x;
        const int v = t < T ? tile_count[t] : 0;
        const int inc = wave_incl_scan_i(v, lane);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        const int start = carry + woff + inc - v;
        if (t < T) {
            // reference: tiles without pairs keep (0,0) (src/sort_gaussian.cu:44-70, zero-init)
            tile_range[2 * t] = v > 0 ? start : 0;
            tile_range[2 * t + 1] = v > 0 ? start + v : 0;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (M_out) *M_out = carry_s;
        *total_scratch = carry_s;
        carry_s = 0;
    }
    __syncthreads();
    // pairs per Gaussian chunk -> exclusive chunk offsets (K3 turns them into the Gaussian-major pair slots)
    for (int base = 0; base < nchunk; base += 1024) {
        const int c = base + threadIdx.x;
        const int v = c < nchunk ? chunk_sum[c] : 0;
        const int inc = wave_incl_scan_i(v, lane);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (c < nchunk) chunk_sum[c] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ K3
template <bool LDS>
__global__ void __launch_bounds__(BIN_BLOCK)
bin_scatter_kernel(int P, const float2 *__restrict__ uv, const float *__restrict__ depth,
                   const int *__restrict__ radius, int gx, int gy, int T, int chunk, int *__restrict__ matrix,
                   const int *__restrict__ tile_range, long long capacity, unsigned long long *__restrict__ keys,
                   int *__restrict__ overflow, int *__restrict__ goff_incl, int *__restrict__ owner,
                   const int *__restrict__ chunk_off, long long S, int kbits, const unsigned *__restrict__ reach) {
    // Pair-map mode (goff_incl != null): the kernel also produces goff_incl, the inclusive prefix of tiles per
    // Gaussian (chunk offset from K2b + a workgroup scan of the rectangle areas), and the low key word is the pair
    // slot j = goff_excl[i] + k (k-th tile of the splat's rectangle) instead of the Gaussian id: slots grow with
    // the id, so ties still order by ascending id, and the sorted keys directly give the pair map.
    extern __shared__ __attribute__((aligned(16))) int cnt[];
    __shared__ int wave_pairs[BIN_BLOCK / 64];
    const int wg = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {   // frame batch: every frame owns `capacity` key / owner slots
        const size_t f = blockIdx.y;
        uv += f * P; depth += f * P; radius += f * P; matrix += f * S; chunk_off += f * S;
        tile_range += f * 2 * T; keys += f * capacity;
        if (goff_incl) { goff_incl += f * P; owner += f * capacity; }
        if (reach) reach += f * P;
    }
    int running = goff_incl ? chunk_off[wg] : 0;
    if (LDS) {
        const int *row = matrix + (size_t)wg * T;
        for (int t = threadIdx.x; t < T; t += BIN_BLOCK) cnt[t] = tile_range[2 * t] + row[t];
        __syncthreads();
    }
    const int beg = wg * chunk, end = imin_(P, beg + chunk);
    for (int base = beg; base < end; base += BIN_BLOCK) {  // uniform trip count: the scan below needs every thread
        const int i = base + threadIdx.x;
        const int r = i < end ? radius[i] : 0;
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        if (r > 0) {
            const float2 q = uv[i];
            tile_rect(q.x, q.y, r, gx, gy, x0, y0, x1, y1);
        }
        // reach word of the count kernel: keep flags of the rectangle's tiles, or (REACH_BIG) the kept count of a large one
        const unsigned rw = (reach && r > 0) ? reach[i] : 0u;
        const bool big = (rw & REACH_BIG) != 0u;
        int j = 0;
        if (goff_incl) {
            const int area = !reach ? (x1 - x0) * (y1 - y0) : big ? reach_big_count(rw, x1 - x0, y1 - y0) : __popc(rw);
            const int inc = wave_incl_scan_i(area, lane);
            if (lane == 63) wave_pairs[w] = inc;
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < BIN_BLOCK / 64; ++k) {
                const int v = wave_pairs[k];
                woff += k < w ? v : 0;
                tot += v;
            }
            __syncthreads();
            // clamped to the capacity: after an overflowing (flagged) sort every consumer still stays inside its buffers
            if (i < end) goff_incl[i] = (int)(((long long)(running + woff + inc) < capacity) ? (long long)(running + woff + inc) : capacity);
            j = running + woff + inc - area;
            running += tot;
        }
        if (r <= 0) continue;
        const unsigned long long dkey = (unsigned long long)__float_as_uint(depth[i]) << 32;
        unsigned bit = 1u;
        const int csh = big ? reach_cell_shift(x1 - x0, y1 - y0) : 0, cnx = big ? ((x1 - x0 - 1) >> csh) + 1 : 0;
        auto kept = [&](int tx, int ty) -> bool {   // (k of the packed keys and the slots count the KEPT tiles, in row-major order)
            if (!reach) return true;
            if (big) return ((rw >> (((ty - y0) >> csh) * cnx + ((tx - x0) >> csh))) & 1u) != 0u;   // the tile's cell
            const bool k = (rw & bit) != 0u;
            bit <<= 1;
            return k;
        };
        if (goff_incl && kbits) {  // packed pair-map keys: (id << kbits) | k
            unsigned lo = (unsigned)i << kbits;
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    if (!kept(tx, ty)) continue;
                    const int t = ty * gx + tx;
                    int slot;
                    if (LDS) slot = atomicAdd(&cnt[t], 1);
                    else slot = tile_range[2 * t] + atomicAdd(&matrix[T + t], 1);
                    if ((long long)slot < capacity) keys[slot] = dkey | lo;
                    else *overflow = 1;
                    ++lo;
                }
            continue;
        }
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                if (!kept(tx, ty)) continue;
                const int t = ty * gx + tx;
                int slot;
                if (LDS) slot = atomicAdd(&cnt[t], 1);
                else slot = tile_range[2 * t] + atomicAdd(&matrix[T + t], 1);  // second row = fill counters
                if ((long long)slot < capacity) {
                    keys[slot] = dkey | (unsigned)(goff_incl ? j : i);
                    if (goff_incl && (long long)j < capacity) owner[j] = i;
                } else {
                    *overflow = 1;
                }
                ++j;
            }
    }
}

// ------------------------------------------------------------------ K4: per-tile bitonic sort
// ---- register-blocked bitonic sort: R keys per thread (element i = t R + r), 256 threads -> up to 256 R keys.
// Compare-exchange partners inside a thread are registers, partners up to 63 lanes away are fetched with VALU-only
// cross-lane moves (DPP quad_perm / row shifts / row_ror, gfx950 v_permlane{16,32}_swap), only the two largest
// strides (other waves) go through LDS: 3 of the 55 (R = 4) / 66 (R = 8) stages.  Missing elements are +inf.
typedef unsigned long long u64;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

template <int D>
__device__ __forceinline__ unsigned lane_xor_u(unsigned v, int lane) {
#if SORT_SWIZZLE_ALL
    if (D <= 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x001F | (D << 10));   // bit mode: xor D inside 32 lanes
#endif
    if (D == 1) return dpp_u<0xB1>(v);   // quad_perm:[1,0,3,2]
    if (D == 2) return dpp_u<0x4E>(v);   // quad_perm:[2,3,0,1]
    if (D == 4) {                        // inside a row of 16: lanes with bit 2 clear read lane+4, the others lane-4
#if SORT_SWIZZLE4
        return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);   // bit mode: and 0x1f, or 0, xor 4 -- the LDS crossbar, no VALU
#else
        const unsigned up = dpp_u<0x104>(v), dn = dpp_u<0x114>(v);  // row_shl:4 / row_shr:4
        return (lane & 4) ? dn : up;
#endif
    }
    if (D == 8) return dpp_u<0x128>(v);  // row_ror:8
    if (D == 16) {
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] rows: x0 x0 x2 x2, r[1]: x1 x1 x3 x3
        return (lane & 16) ? r[0] : r[1];
    }
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // r[0] rows: x0 x1 x0 x1, r[1]: x2 x3 x2 x3
    return (lane & 32) ? r[0] : r[1];
}

template <int D>
__device__ __forceinline__ u64 lane_xor_key(u64 k, int lane) {
    const unsigned lo = lane_xor_u<D>((unsigned)k, lane), hi = lane_xor_u<D>((unsigned)(k >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// one cross-thread stage at thread distance D (element stride D R): the lower thread keeps the minimum when ascending
template <int R, int D>
__device__ __forceinline__ void xthread_stage(u64 (&k)[R], int t, bool up, u64 *xbuf) {
    const bool take_min = ((t & D) == 0) == up;
    if ((D == 16 && !SORT_SWIZZLE_ALL) || D == 32) {
        // a swap of (k, k) leaves the pair's LOWER thread's key in the first result and the upper thread's in the second, in
        // both threads: the pair's minimum / maximum needs one compare of the two results and no partner select (the generic
        // form below selects the partner's words first: two selects and two register copies more per key)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned lo = (unsigned)k[r], hi = (unsigned)(k[r] >> 32);
            const u32x2_t sl = D == 16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false)
                                       : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const u32x2_t sh = D == 16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false)
                                       : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            const u64 a = ((u64)sh[0] << 32) | sl[0], b = ((u64)sh[1] << 32) | sl[1];   // lower / upper thread of the pair
            k[r] = ((a < b) == take_min) ? a : b;   // take_min ? min(a, b) : max(a, b)
        }
    } else if (D < 64) {
        const int lane = t & 63;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u64 p = lane_xor_key<D>(k[r], lane);
            const bool less = p < k[r];
            k[r] = (less == take_min) ? p : k[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) xbuf[r * SORT_BLOCK + t] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u64 p = xbuf[r * SORT_BLOCK + (t ^ D)];
            const bool less = p < k[r];
            k[r] = (less == take_min) ? p : k[r];
        }
        __syncthreads();
    }
}

// ---- the network: every comparator ASCENDS.  A merge of width K starts with the "flip" step (element i against its mirror
// inside the block of K) and goes on with the strides K/4 .. 1; the lower element keeps the minimum everywhere.  The +inf of the
// missing tail therefore never moves, and a wave whose 64 R elements all lie past the list (`live` false, wave-uniform) has
// nothing to do outside the LDS exchanges: a tile of c2 (711 keys of the 1024 a workgroup sorts) runs three waves' worth of
// compare-exchanges instead of four.
template <int M>
__device__ __forceinline__ unsigned lane_mirror_u(unsigned v, int mirror_addr) {   // lane ^ M, M = 2^s - 1
    if (M == 1) return dpp_u<0xB1>(v);    // quad_perm:[1,0,3,2]
    if (M == 3) return dpp_u<0x1B>(v);    // quad_perm:[3,2,1,0]
    if (M == 7) return dpp_u<0x141>(v);   // row_half_mirror
    if (M == 15) return dpp_u<0x140>(v);  // row_mirror
    if (M == 31) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F);   // bit mode: xor 31 inside 32 lanes
    return (unsigned)__builtin_amdgcn_ds_bpermute(mirror_addr, (int)v);          // the whole wave reversed
}

// flip step of the merge of width K = D R elements (thread t against thread t ^ (D - 1), register r against R - 1 - r)
template <int R, int D>
__device__ __forceinline__ void xthread_flip(u64 (&k)[R], int t, bool live, u64 *xbuf) {
    const bool lower = (t & (D >> 1)) == 0;
    if (D <= 64) {
        if (!live) return;
        const int mirror_addr = ((t & 63) ^ 63) << 2;
        u64 p[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u64 q = k[R - 1 - r];
            p[r] = ((u64)lane_mirror_u<D - 1>((unsigned)(q >> 32), mirror_addr) << 32) | lane_mirror_u<D - 1>((unsigned)q, mirror_addr);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) k[r] = ((p[r] < k[r]) == lower) ? p[r] : k[r];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) xbuf[r * SORT_BLOCK + t] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u64 p = xbuf[(R - 1 - r) * SORT_BLOCK + (t ^ (D - 1))];
            k[r] = ((p < k[r]) == lower) ? p : k[r];
        }
        __syncthreads();
    }
}

template <int R, int K>
__device__ __forceinline__ void flip_stage(u64 (&k)[R], int t, bool live, u64 *xbuf) {
    if (K <= R) {
        if (!live) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r & (K >> 1)) == 0) {
                const u64 a = k[r], b = k[r ^ (K - 1)];
                const bool sw = a > b;
                k[r] = sw ? b : a;
                k[r ^ (K - 1)] = sw ? a : b;
            }
        }
    } else {
        xthread_flip<R, K / R>(k, t, live, xbuf);
    }
}

template <int R, int J>
__device__ __forceinline__ void asc_stage(u64 (&k)[R], int t, bool live, u64 *xbuf) {   // stride J (elements), ascending
    if (J < R) {
        if (!live) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r & J) == 0) {
                const u64 a = k[r], b = k[r | J];
                const bool sw = a > b;
                k[r] = sw ? b : a;
                k[r | J] = sw ? a : b;
            }
        }
    } else {
        if (J / R < 64 && !live) return;
        xthread_stage<R, J / R>(k, t, true, xbuf);
    }
}

template <int R, int J>
struct BitonicJ {   // strides J, J / 2 .. 1
    static __device__ __forceinline__ void run(u64 (&k)[R], int t, bool live, u64 *xbuf) {
        asc_stage<R, J>(k, t, live, xbuf);
        BitonicJ<R, J / 2>::run(k, t, live, xbuf);
    }
};
template <int R>
struct BitonicJ<R, 0> {
    static __device__ __forceinline__ void run(u64 (&)[R], int, bool, u64 *) {}
};
template <int R, int K, int NP>
struct BitonicK {   // merges of width K, 2 K .. NP
    static __device__ __forceinline__ void run(u64 (&k)[R], int t, bool live, u64 *xbuf) {
        flip_stage<R, K>(k, t, live, xbuf);
        BitonicJ<R, K / 4>::run(k, t, live, xbuf);
        BitonicK<R, K * 2, NP>::run(k, t, live, xbuf);
    }
};
template <int R, int NP>
struct BitonicK<R, NP * 2, NP> {
    static __device__ __forceinline__ void run(u64 (&)[R], int, bool, u64 *) {}
};

// low key word -> (Gaussian id, pair slot) of a sorted entry (pair-map mode; see pair_key_kbits)
struct PairKey {
    const int *owner;   // slot keys: Gaussian of every slot
    const int *goff;    // packed keys: inclusive prefix of tiles per Gaussian
    int kbits;
    long long capacity;
    __device__ __forceinline__ void resolve(unsigned lo, int &id, int &slot) const {
        if (kbits) {
            id = (int)(lo >> kbits);
            const long long j = (long long)(id > 0 ? goff[id - 1] : 0) + (lo & ((1u << kbits) - 1u));
            slot = (int)(j < capacity ? j : capacity - 1);   // (only after a flagged overflow)
        } else {
            slot = (long long)lo < capacity ? (int)lo : (int)(capacity - 1);
            id = (long long)lo < capacity ? owner[lo] : 0;
        }
    }
};

template <int R>
__device__ __forceinline__ void tile_sort_regs(const u64 *g, int n, long long r0, const PairKey &pk, int *idx_sorted,
                                               int *slot_sorted, u64 *xbuf) {
    const int t = threadIdx.x;
    u64 k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = t * R + r;
        k[r] = i < n ? g[i] : ~0ull;
    }
    BitonicK<R, 2, R * SORT_BLOCK>::run(k, t, (t & ~63) * R < n, xbuf);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = t * R + r;
        if (i < n) {
            const unsigned lo = (unsigned)(k[r] & 0xffffffffull);
            if (slot_sorted) {
                int id, slot;
                pk.resolve(lo, id, slot);
                slot_sorted[r0 + i] = slot;
                idx_sorted[r0 + i] = id;
            } else {
                idx_sorted[r0 + i] = (int)lo;
            }
        }
    }
}

// ---- tiles above 8 SORT_BLOCK keys (crowded tiles: a foreground object on few tiles): blocks of NB = 8 SORT_BLOCK keys sort
// in registers; the merges above NB take their wide strides (>= NB elements) through the tile's global segment (the segment of
// one tile stays in L2) and finish every block's strides below NB in registers again.  All comparators of the merges ascend
// ("flip" first step), so the +inf of the missing tail never moves and is never stored.  A tile of 5600 keys: 3 block sorts,
// 3 global passes and 6 register cascades instead of the 91 global-memory stages of bitonic_any_n (BASELINE configs[1],
// clustered scene: 205 us per frame of tile_sort before).
template <int R>
__device__ __forceinline__ void tile_sort_blocks(volatile u64 *g, int n, u64 *xbuf) {
    constexpr int NB = R * SORT_BLOCK;
    const int t = threadIdx.x;
    const int nblk = (n + NB - 1) / NB;
    u64 k[R];
    auto load = [&](int b) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = b * NB + t * R + r;
            k[r] = i < n ? g[i] : ~0ull;
        }
    };
    auto store = [&](int b) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = b * NB + t * R + r;
            if (i < n) g[i] = k[r];
        }
    };
    for (int b = 0; b < nblk; ++b) {
        load(b);
        BitonicK<R, 2, NB>::run(k, t, b * NB + (t & ~63) * R < n, xbuf);
        store(b);
    }
    __syncthreads();
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    const int half = (1 << lg) >> 1;
    for (int lk = 1; lk <= lg; ++lk) {
        const int kw = 1 << lk, hk = kw >> 1;
        if (kw <= NB) continue;                      // (merges inside a block: done)
        for (int c = t; c < half; c += SORT_BLOCK) {  // flip step: i <-> mirror inside the block of kw
            const int blk = c >> (lk - 1), off = c & (hk - 1);
            const int lo = (blk << lk) + off, hi = (blk << lk) + kw - 1 - off;
            if (hi < n) {
                const u64 x = g[lo], y = g[hi];
                if (x > y) { g[lo] = y; g[hi] = x; }
            }
        }
        __syncthreads();
        for (int lj = lk - 2; (1 << lj) >= NB; --lj) {   // strides of whole blocks
            const int j = 1 << lj;
            for (int c = t; c < half; c += SORT_BLOCK) {
                const int blk = c >> lj, off = c & (j - 1);
                const int lo = (blk << (lj + 1)) + off, hi = lo + j;
                if (hi < n) {
                    const u64 x = g[lo], y = g[hi];
                    if (x > y) { g[lo] = y; g[hi] = x; }
                }
            }
            __syncthreads();
        }
        for (int b = 0; b < nblk; ++b) {                 // strides NB / 2 .. 1 of every block, ascending
            load(b);
            BitonicJ<R, NB / 2>::run(k, t, b * NB + (t & ~63) * R < n, xbuf);
            store(b);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(SORT_BLOCK)
tile_sort_kernel(int T, int *__restrict__ tile_range, long long capacity, unsigned long long *__restrict__ keys,
                 int *__restrict__ idx_sorted, const int *__restrict__ owner, int *__restrict__ slot_sorted,
                 const int *__restrict__ goff_incl, int P, int kbits) {
    __shared__ __attribute__((aligned(16))) unsigned long long sk[SORT_LDS_KEYS];
    // one workgroup per (frame, tile), XCD-aware (xcd_tile: runs of neighbouring tiles on one XCD): neighbouring tiles look
    // up largely the same sectors of `owner` (pair slots are Gaussian-major, a splat touches ~4 tiles); dealt round-robin by
    // tile every one of the 8 L2s fetched the whole array (PMC: 1208 MB per 25-frame launch for 346 MB of keys and ids)
    const int gtile = xcd_tile(blockIdx.x, gridDim.x);
    const int t = gtile % T;
    {   // frame batch: T tiles per frame
        const size_t f = gtile / T;
        tile_range += f * 2 * T; keys += f * capacity; idx_sorted += f * capacity;
        if (slot_sorted) { owner += f * capacity; slot_sorted += f * capacity; goff_incl += f * P; }
    }
    const PairKey pk{owner, goff_incl, kbits, capacity};
    long long r0 = tile_range[2 * t];
    long long r1 = tile_range[2 * t + 1];
    if (r1 > capacity) {  // overflow (already flagged by K3): the range the blend kernels will read never leaves the buffers
        r1 = capacity;
        if (r0 > capacity) r0 = capacity;
        if (threadIdx.x == 0) {
            tile_range[2 * t] = (int)r0;
            tile_range[2 * t + 1] = (int)r1;
        }
    }
    const int n = (int)(r1 - r0);
    if (n <= 0) return;
    unsigned long long *g = keys + r0;
    // low key word: Gaussian id, or (pair-map mode) the pair slot whose owner is the Gaussian id
#ifndef SORT_R2
#define SORT_R2 1   // tiles of at most 512 keys: two keys per thread (45 stages, all four waves live) instead of four (55 stages, two waves live)
#endif
    if (SORT_R2 && n <= 2 * SORT_BLOCK) {
        tile_sort_regs<2>(g, n, r0, pk, idx_sorted, slot_sorted, sk);
    } else if (n <= 4 * SORT_BLOCK) {
        tile_sort_regs<4>(g, n, r0, pk, idx_sorted, slot_sorted, sk);
    } else if (n <= 8 * SORT_BLOCK) {
        tile_sort_regs<8>(g, n, r0, pk, idx_sorted, slot_sorted, sk);
    } else {
        __syncthreads();
        tile_sort_blocks<8>((volatile u64 *)g, n, sk);
        for (int i = threadIdx.x; i < n; i += SORT_BLOCK) {
            const unsigned lo = (unsigned)(((volatile u64 *)g)[i] & 0xffffffffull);
            if (slot_sorted) {
                int id, slot;
                pk.resolve(lo, id, slot);
                slot_sorted[r0 + i] = slot;
                idx_sorted[r0 + i] = id;
            } else {
                idx_sorted[r0 + i] = (int)lo;
            }
        }
    }
}

// ================================================================== C ABI
extern "C" size_t splat_bin_scratch_bytes(int P, int W, int H) {
    if (P < 0 || W <= 0 || H <= 0) return 0;
    return make_plan(P, W, H).bytes;
}

static int bin_count_impl(int F, int P, const float *uv, const int32_t *radius, int W, int H, void *scratch,
                          int32_t *tile_range, int32_t *M_out, int32_t *gcount, hipStream_t s, const float *conic = nullptr,
                          const float *opacity = nullptr, long long opacity_fs = 0, uint32_t *reach = nullptr) {
    const BinPlan p = make_plan(P, W, H, F);
    const long long S = (long long)(p.bytes / sizeof(int));
    char *base = (char *)scratch;
    int *matrix = (int *)(base + p.off_matrix);
    int *tile_count = (int *)(base + p.off_tilecount);
    int *total = (int *)(base + p.off_total);
    int *chunk_sum = (int *)(base + p.off_chunksum);
    if (p.lds) {
        SPLAT_LAUNCH("bin_count", bin_count_kernel<true>, dim3(p.NB, F), dim3(BIN_BLOCK), (size_t)p.T * sizeof(int), s,
                     P, (const float2 *)uv, radius, p.gx, p.gy, p.T, p.chunk, matrix, gcount, chunk_sum, S, conic, opacity, opacity_fs,
                     reach);
    } else {
        // rows: [0] counts, [1] fill counters of K3
        for (int f = 0; f < F; ++f)
            SPLAT_CHECK_HIP(hipMemsetAsync(matrix + (size_t)f * S, 0, (size_t)p.T * sizeof(int), s));
        const int nblk = p.nchunk;
        const int chunk = (P + nblk - 1) / nblk;
        SPLAT_LAUNCH("bin_count", bin_count_kernel<false>, dim3(nblk, F), dim3(BIN_BLOCK), 0, s, P, (const float2 *)uv,
                     radius, p.gx, p.gy, p.T, chunk > 0 ? chunk : 1, matrix, gcount, chunk_sum, S, conic, opacity, opacity_fs, reach);
    }
    SPLAT_POST_LAUNCH();
    if (p.lds) {
        SPLAT_LAUNCH("bin_colscan", bin_colscan_kernel, dim3((p.T + COLSCAN_COLS - 1) / COLSCAN_COLS, F),
                     dim3(COLSCAN_COLS * COLSCAN_GROUPS), 0, s, p.T, p.NB, matrix, tile_count, S);
        SPLAT_POST_LAUNCH();
    } else {
        for (int f = 0; f < F; ++f)
            SPLAT_CHECK_HIP(hipMemcpyAsync(tile_count + (size_t)f * S, matrix + (size_t)f * S, (size_t)p.T * sizeof(int),
                                           hipMemcpyDeviceToDevice, s));
    }
    SPLAT_LAUNCH("bin_tilescan", bin_tilescan_kernel, dim3(1, F), dim3(1024), 0, s, p.T, tile_count, tile_range, M_out, total,
                 p.nchunk, chunk_sum, S);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

static int bin_sort_impl(int F, int P, const float *uv, const float *depth, const int32_t *radius, int W, int H,
                         void *scratch, int32_t *tile_range, int64_t capacity, uint64_t *keys, int32_t *idx_sorted,
                         int32_t *overflow_out, int32_t *goff_incl, int32_t *owner_scratch, int32_t *slot_sorted,
                         hipStream_t s, const uint32_t *reach = nullptr) {
    const BinPlan p = make_plan(P, W, H, F);
    const long long S = (long long)(p.bytes / sizeof(int));
    char *base = (char *)scratch;
    int *matrix = (int *)(base + p.off_matrix);
    const int *chunk_off = (const int *)(base + p.off_chunksum);
    const int kbits = goff_incl ? pair_key_kbits(P, p.T) : 0;
    if (p.lds) {
        SPLAT_LAUNCH("bin_scatter", bin_scatter_kernel<true>, dim3(p.NB, F), dim3(BIN_BLOCK), (size_t)p.T * sizeof(int), s,
                     P, (const float2 *)uv, depth, radius, p.gx, p.gy, p.T, p.chunk, matrix, tile_range,
                     (long long)capacity, (unsigned long long *)keys, overflow_out, goff_incl, owner_scratch, chunk_off, S, kbits,
                     reach);
    } else {
        // fill counters live in tile_count's neighbour: reuse matrix row "1" = matrix + T (allocated: NB=1 -> need 2 rows)
        for (int f = 0; f < F; ++f)
            SPLAT_CHECK_HIP(hipMemsetAsync(matrix + (size_t)f * S + p.T, 0, (size_t)p.T * sizeof(int), s));
        const int nblk = p.nchunk;
        const int chunk = (P + nblk - 1) / nblk;
        SPLAT_LAUNCH("bin_scatter", bin_scatter_kernel<false>, dim3(nblk, F), dim3(BIN_BLOCK), 0, s, P, (const float2 *)uv,
                     depth, radius, p.gx, p.gy, p.T, chunk > 0 ? chunk : 1, matrix, tile_range, (long long)capacity,
                     (unsigned long long *)keys, overflow_out, goff_incl, owner_scratch, chunk_off, S, kbits, reach);
    }
    SPLAT_POST_LAUNCH();
    SPLAT_LAUNCH("tile_sort", tile_sort_kernel, dim3((unsigned)((size_t)p.T * F)), dim3(SORT_BLOCK), 0, s, p.T, tile_range, (long long)capacity,
                 (unsigned long long *)keys, idx_sorted, owner_scratch, slot_sorted, goff_incl, P, kbits);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_bin_count(int P, const float *uv, const int32_t *radius, int W, int H, void *scratch,
                               int32_t *tile_range, int32_t *M_out, int32_t *gcount, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(scratch && tile_range, "null pointer");
    SPLAT_CHECK_ARG(P == 0 || (uv && radius), "null pointer");
    return bin_count_impl(1, P, uv, radius, W, H, scratch, tile_range, M_out, gcount, (hipStream_t)stream);
}

extern "C" int splat_bin_sort(int P, const float *uv, const float *depth, const int32_t *radius, int W, int H,
                              void *scratch, int32_t *tile_range, int64_t capacity, uint64_t *keys,
                              int32_t *idx_sorted, int32_t *overflow_out, int32_t *goff_incl,
                              int32_t *owner_scratch, int32_t *slot_sorted, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0 && capacity >= 0, "bad sizes");
    SPLAT_CHECK_ARG(scratch && tile_range && overflow_out, "null pointer");
    if (P == 0 || capacity == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(uv && depth && radius && keys && idx_sorted, "null pointer");
    {
        const int npm = (goff_incl != nullptr) + (owner_scratch != nullptr) + (slot_sorted != nullptr);
        SPLAT_CHECK_ARG(npm == 0 || npm == 3, "goff_incl, owner_scratch and slot_sorted go together");
    }
    return bin_sort_impl(1, P, uv, depth, radius, W, H, scratch, tile_range, capacity, keys, idx_sorted, overflow_out,
                         goff_incl, owner_scratch, slot_sorted, (hipStream_t)stream);
}

// ---- frame batch: F frames of the same P Gaussians in one set of launches (grid.y = frame).  Per-Gaussian arrays are
// [F,P,..], tile_range [F,T,2], M_out [F], scratch F blocks of splat_bin_scratch_bytes(P, W, H); keys / idx_sorted /
// owner / slot_sorted hold `capacity` entries per frame (tile ranges are relative to the frame's segment).
extern "C" int splat_bin_count_batch(int F, int P, const float *uv, const int32_t *radius, int W, int H, void *scratch,
                                     int32_t *tile_range, int32_t *M_out, splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && F <= 65535 && P >= 1 && W > 0 && H > 0, "bad sizes");
    SPLAT_CHECK_ARG(scratch && tile_range && uv && radius, "null pointer");
    return bin_count_impl(F, P, uv, radius, W, H, scratch, tile_range, M_out, nullptr, (hipStream_t)stream);
}

extern "C" int splat_bin_sort_batch(int F, int P, const float *uv, const float *depth, const int32_t *radius, int W, int H,
                                    void *scratch, int32_t *tile_range, int64_t capacity, uint64_t *keys,
                                    int32_t *idx_sorted, int32_t *overflow_out, int32_t *goff_incl,
                                    int32_t *owner_scratch, int32_t *slot_sorted, splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && F <= 65535 && P >= 1 && W > 0 && H > 0 && capacity >= 1, "bad sizes");
    SPLAT_CHECK_ARG(scratch && tile_range && overflow_out && uv && depth && radius && keys && idx_sorted && goff_incl &&
                        owner_scratch && slot_sorted,
                    "null pointer");
    return bin_sort_impl(F, P, uv, depth, radius, W, H, scratch, tile_range, capacity, keys, idx_sorted, overflow_out,
                         goff_incl, owner_scratch, slot_sorted, (hipStream_t)stream);
}


// ---- the same two steps with REACH masks (ABI 21; see "reach masks" above): conic [F,P,3] and opacity ([P], or [F,P] with
// opacity_frame_stride = P) let the count kernel drop the pairs whose tile the splat cannot reach with alpha >= 1/255; `reach`
// [F,P] words travel from the count step to the sort step (which reads nothing else about the decision).  F = 1 serves the per-frame operators (gcount: optional kept tiles
// per Gaussian, [F,P]).  tile_range, M_out, goff_incl and the slots count the KEPT pairs only.
extern "C" int splat_bin_count_batch_reach(int F, int P, const float *uv, const int32_t *radius, const float *conic,
                                           const float *opacity, int64_t opacity_frame_stride, int W, int H, void *scratch,
                                           int32_t *tile_range, int32_t *M_out, int32_t *gcount, uint32_t *reach,
                                           splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && F <= 65535 && P >= 1 && W > 0 && H > 0 && opacity_frame_stride >= 0, "bad sizes");
    SPLAT_CHECK_ARG(scratch && tile_range && uv && radius && conic && opacity && reach, "null pointer");
    return bin_count_impl(F, P, uv, radius, W, H, scratch, tile_range, M_out, gcount, (hipStream_t)stream, conic, opacity,
                          (long long)opacity_frame_stride, reach);
}

extern "C" int splat_bin_sort_batch_reach(int F, int P, const float *uv, const float *depth, const int32_t *radius,
                                          const uint32_t *reach, int W, int H, void *scratch, int32_t *tile_range,
                                          int64_t capacity, uint64_t *keys, int32_t *idx_sorted, int32_t *overflow_out,
                                          int32_t *goff_incl, int32_t *owner_scratch, int32_t *slot_sorted,
                                          splat_stream_t stream) {
    SPLAT_CHECK_ARG(F >= 1 && F <= 65535 && P >= 1 && W > 0 && H > 0 && capacity >= 1, "bad sizes");
    SPLAT_CHECK_ARG(scratch && tile_range && overflow_out && uv && depth && radius && keys && idx_sorted && goff_incl &&
                        owner_scratch && slot_sorted && reach,
                    "null pointer");
    return bin_sort_impl(F, P, uv, depth, radius, W, H, scratch, tile_range, capacity, keys, idx_sorted, overflow_out,
                         goff_incl, owner_scratch, slot_sorted, (hipStream_t)stream, reach);
}


// ================================================================== the reference's two sort helpers (dptr.gs._C names)
// For callers that keep the reference's own sort_gaussian.py (cumsum -> compute_gaussian_key -> torch.sort -> gather ->
// compute_tile_gaussian_range, sort_gaussian.py:42-52): same keys (tile << 32 | depth bits, src/sort_gaussian.cu:24-44)
// and the same range rule (:46-70).  The native sort above does not use them.
__global__ void __launch_bounds__(256)
gaussian_key_kernel(int P, const float2 *__restrict__ uv, const float *__restrict__ depth, const int *__restrict__ radius,
                    const int *__restrict__ tiles_cumsum, int gx, int gy, long long *__restrict__ key,
                    int *__restrict__ gidx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radius[i];
    if (r <= 0) return;
    const float2 q = uv[i];
    int x0, y0, x1, y1;
    tile_rect(q.x, q.y, r, gx, gy, x0, y0, x1, y1);
    int cur = i == 0 ? 0 : tiles_cumsum[i - 1];
    const long long d = (long long)__float_as_int(depth[i]);
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            key[cur] = ((long long)(ty * gx + tx) << 32) | d;
            gidx[cur] = i;
            ++cur;
        }
}

__global__ void __launch_bounds__(256)
tile_gaussian_range_kernel(long long M, const long long *__restrict__ key_sorted, int2 *__restrict__ tile_range) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int cur = (int)(key_sorted[i] >> 32);
    if (i == 0) tile_range[cur].x = 0;
    if (i == M - 1) tile_range[cur].y = (int)M;
    if (i == 0) return;
    const int prev = (int)(key_sorted[i - 1] >> 32);
    if (prev != cur) {
        tile_range[prev].y = (int)i;
        tile_range[cur].x = (int)i;
    }
}

extern "C" int splat_compute_gaussian_key(int P, const float *uv, const float *depth, const int32_t *radius,
                                          const int32_t *tiles_cumsum, int W, int H, int64_t *gaussian_key,
                                          int32_t *gaussian_idx, splat_stream_t stream) {
    SPLAT_CHECK_ARG(P >= 0 && W > 0 && H > 0, "bad sizes");
    if (P == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(uv && depth && radius && tiles_cumsum, "null pointer");
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    SPLAT_LAUNCH("gaussian_key", gaussian_key_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P,
                 (const float2 *)uv, depth, radius, tiles_cumsum, gx, gy, (long long *)gaussian_key, gaussian_idx);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

extern "C" int splat_compute_tile_gaussian_range(int64_t M, const int64_t *key_sorted, int32_t *tile_range /*[T,2], zero-filled*/,
                                                 splat_stream_t stream) {
    SPLAT_CHECK_ARG(M >= 0, "bad sizes");
    if (M == 0) return SPLAT_OK;
    SPLAT_CHECK_ARG(key_sorted && tile_range, "null pointer");
    SPLAT_LAUNCH("tile_gaussian_range", tile_gaussian_range_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                 (hipStream_t)stream, (long long)M, (const long long *)key_sorted, (int2 *)tile_range);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
