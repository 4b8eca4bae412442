/*
 * splat_hip.h -- C ABI of libsplat_hip.so, the MI355X (gfx950) native layer behind the
 * `dptr.gs` operator surface.
 *
 * It replaces the reference's pybind module `dptr.gs._C`
 * (reference: src/submodules/dptr/dptr/gs/src/ext.cpp:14-33, 18 functions on torch::Tensor).
 * Differences by design:
 *   - plain C: raw DEVICE pointers + int sizes + scalars + a hipStream_t; no torch / pybind types;
 *   - the CALLER owns every buffer, the library allocates nothing persistent.  Unless marked
 *     "zero-init", the kernels write EVERY element of an output (zeros for culled points, -1 for
 *     unused gs_idx slots, (0,0) for empty tiles), so outputs may be uninitialised memory;
 *     "zero-init" buffers are accumulated into (camera gradients, atomic-mode blend gradients);
 *   - every launch goes to the caller's stream (the reference uses the legacy default stream);
 *   - returns 0 or a negative SPLAT_E_* code; splat_last_error() gives the thread-local message.
 *
 * Layout conventions: row-major float32 / int32 / uint8(bool); xyz[P,3], uv[P,2], depth[P],
 * conic[P,3], cov3d[P,6], scales[P,3], uquats[P,4] (r,x,y,z), shs flat with stride (deg+1)^2
 * triplets per point, feature[P,C] (row-major; the reference transposes to [C,P] on the host),
 * out[C,H,W], final_T[H,W], ncontrib[H,W], gs_idx[H,W,K], tile_range[T,2] with
 * T = ceil(W/16)*ceil(H/16); intr = [fx,fy,cx,cy]; extr = first 12 floats of a row-major [R|T].
 * `ortho` != 0 selects the orthographic twins of
 * src/pointrix/renderer/dptr_ortho_enhanced.py:18-111,:145-202 instead of the perspective CUDA
 * semantics.
 */
#ifndef SPLAT_HIP_H
#define SPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *splat_stream_t; /* hipStream_t */

#define SPLAT_OK 0
#define SPLAT_E_ARG (-1)      /* bad argument (null pointer, negative size, unsupported value) */
#define SPLAT_E_LAUNCH (-2)   /* HIP launch / runtime error */
#define SPLAT_E_CAPACITY (-3) /* caller-provided capacity too small */

const char *splat_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int splat_abi_version(void);
/* Hash (16 hex digits) of the sources this binary was built from (csrc/Makefile); "unstamped" for a build outside the
   Makefile.  Measurement records under profiles/ carry it; bench.py only quotes PMC constants taken from the same build. */
const char *splat_build_id(void);
/* Deterministic mode (process-wide; the reference has none -- its backward is all float atomics, src/alpha_blending.cu:
   229-246): when on, no backward this library launches uses float atomics, so identical inputs give bit-identical
   gradients run to run.  The frame-batch entry points and the per-frame operators on this library's own sort (pair
   records) are deterministic with the flag off as well (tests/test_gpu_determinism.py); the flag makes the block-level
   matrix-core kernel (carried survivors: <= 15 atomic record adds per wave and super-batch) give way to the DPP pair
   kernel and makes splat_alpha_blending_backward without a pair map (foreign idx_sorted: atomic kernel) return
   SPLAT_E_ARG.  Camera gradients of project_point (one atomic per wave) stay outside the guarantee. */
void splat_set_deterministic(int on);
int splat_get_deterministic(void);
/* dst[0 .. n) = value on `stream` (gradient-bucket zeroing without a framework fill kernel). */
int splat_fill_f32(float *dst, size_t n, float value, splat_stream_t stream);
/* Process-wide options, set THROUGH THE ABI (the library reads no environment variable).  Keys (default):
     "bwd_quarters"   (1)  backward tile kernels walk one survivor list per 4x4 quarter when the forward's cull words are given;
                           0: the block-level matrix-core kernels everywhere (what a caller without cull words gets)
     "bwd_kernel_dpp" (0)  1: the DPP-reduction pair kernel instead of the matrix-core kernels (A/B measurements)
     "sets_std"       (1)  the three-set backward's float4 record stores + the forward's packed records for the renderer's own
                           plan (rgb 0-2 | depth 3 | 19 attributes 4-22); 0: the generic slot -> channel routing for every plan
     "bin_slot_keys"  (0)  1: pair-slot sort keys + owner array also where the packed (id, k) keys fit (tests: small sizes)
     "deterministic"  (0)  = splat_set_deterministic
   Unknown key: SPLAT_E_ARG.  Set them before the first launch that depends on them (they are read at launch time). */
int splat_set_option(const char *key, int value);
int splat_get_option(const char *key, int *value);

/* A feature SOURCE of a composited row: row channels [c0, c0 + cn) come from dense rows feature[P, cn]; frame f of a batch
   reads feature + f * frame_stride floats (0: one tensor shared by the frames).  d_feature (Gaussian-side backward only; NULL:
   no gradient wanted) has the same layout: a shared source receives the SUM over the frames, a per-frame source every frame's
   own gradient (both ADDED).  A set of the reference's renderer is the concatenation of its sources
   (RenderFeatures.combine, src/pointrix/utils/renderer/renderer_utils.py:31-72; e.g. ["track_gs"] + render_attributes,
   src/trainer_fragGS.py:511 -- track_gs = position(ids2) differs per frame, the attributes do not). */
typedef struct splat_feature_source_t {
    int32_t c0, cn;
    const float *feature;
    float *d_feature;
    int64_t frame_stride;
} splat_feature_source_t;
#define SPLAT_MAX_SOURCES 8

/* ---- project_point : replaces projectPointsForward/Backward (src/project_point.cu:147-227) ---- */
/* uv, depth: fully written. */
int splat_project_point_forward(int P, const float *xyz, const float *intr, const float *extr, int W, int H,
                                float nearest, float extent, int ortho, float *uv, float *depth,
                                splat_stream_t stream);
/* dL_dxyz fully written; dL_dintr[4] / dL_dextr[12] zero-init or NULL (only when requires_grad). */
int splat_project_point_backward(int P, const float *xyz, const float *intr, const float *extr, int W, int H,
                                 int ortho, const float *depth, const float *dL_duv, const float *dL_ddepth,
                                 float *dL_dxyz, float *dL_dintr, float *dL_dextr, splat_stream_t stream);

/* ---- compute_cov3d : replaces computeCov3DForward/Backward (src/compute_cov3d.cu:149-199) ---- */
int splat_compute_cov3d_forward(int P, const float *scales, const float *uquats, const uint8_t *visible,
                                float *cov3d, splat_stream_t stream);
int splat_compute_cov3d_backward(int P, const float *scales, const float *uquats, const uint8_t *visible,
                                 const float *dL_dcov3d, float *dL_dscales, float *dL_duquats,
                                 splat_stream_t stream);

/* ---- ewa_project : replaces EWAProjectForward/Backward (src/ewa_project.cu:254-344) ---- */
int splat_ewa_project_forward(int P, const float *xyz, const float *cov3d, const float *intr, const float *extr,
                              const float *uv, int W, int H, const uint8_t *visible, int ortho,
                              float *conic, int32_t *radius, int32_t *tiles,
                              splat_stream_t stream);
int splat_ewa_project_backward(int P, const float *xyz, const float *cov3d, const float *intr, const float *extr,
                               int W, int H, int ortho, const int32_t *radius, const float *dL_dconic,
                               float *dL_dxyz, float *dL_dcov3d,
                               float *dL_dintr /*[4] or NULL*/, float *dL_dextr /*[12] or NULL*/,
                               splat_stream_t stream);

/* ---- compute_sh / compute_sh_free : replaces computeSH(Free)Forward/Backward
 *      (src/compute_sh.cu:235-295, src/compute_sh_free.cu) ---- */
int splat_compute_sh_forward(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                             int free_variant, float *colors,
                             uint8_t *clamped /*[P,3], NULL when free*/, splat_stream_t stream);
int splat_compute_sh_backward(int P, const float *shs, int degree, const float *dirs, const uint8_t *visible,
                              const uint8_t *clamped /*NULL when free*/, int free_variant,
                              const float *dL_dcolors,
                              int accumulate /*1: add into dL_dshs (gradient-bucket use), 0: store*/,
                              float *dL_dshs /*rows of (deg+1)^2 triplets written*/,
                              float *dL_ddirs /*NULL: direction gradient not needed (skips reading shs)*/,
                              splat_stream_t stream);

/* ---- sort_gaussian : replaces computeGaussianKey + torch.sort + gather + computeTileGaussianRange
 *      (dptr/gs/sort_gaussian.py:42-52, src/sort_gaussian.cu:72-146) by a two-level sort:
 *      counting scatter by tile, then a per-tile bitonic sort on (depth bits, gaussian id).
 *      Result order == stable sort of the reference's 64-bit keys (ties -> ascending id).
 *   step 1  splat_bin_count : tile_range[T,2] and *M_out (device int32) from uv / radius;
 *   step 2  splat_bin_sort  : idx_sorted[0..M) (needs capacity >= M, else SPLAT_E_CAPACITY is
 *           reported through the device flag *overflow_out (no host sync inside)).
 *   scratch: splat_bin_scratch_bytes(P, W, H) bytes, shared by both steps (must stay untouched in
 *   between); keys: capacity * 8 bytes. ---- */
size_t splat_bin_scratch_bytes(int P, int W, int H);
/* gcount[P] (optional, may be NULL): number of tiles each Gaussian touches (0 when radius <= 0). */
int splat_bin_count(int P, const float *uv, const int32_t *radius, int W, int H, void *scratch,
                    int32_t *tile_range, int32_t *M_out, int32_t *gcount, splat_stream_t stream);
/* Pair map (optional; goff_incl, owner, slot_sorted all given or all NULL): the sort WRITES goff_incl[P], the
 * INCLUSIVE prefix sum of the tiles each Gaussian touches (the scan is fused into the count / scatter kernels: chunk
 * totals in splat_bin_count, chunk offsets + a workgroup scan in the scatter).  Every (Gaussian, tile) pair owns the
 * slot goff_excl[id] + k (k-th tile, row-major inside the splat's tile rectangle); the sort also emits
 * slot_sorted[M] = slot of each sorted entry (owner[M] is workspace: Gaussian id of each slot -- left untouched when the
 * Gaussian id and k fit one 32-bit key word together, bits(P) + bits(T) <= 32, which needs no such array).  The atomic-free
 * blend backward consumes goff_incl and slot_sorted. */
int splat_bin_sort(int P, const float *uv, const float *depth, const int32_t *radius, int W, int H,
                   void *scratch, int32_t *tile_range /*in; clamped to the capacity on overflow*/, int64_t capacity, uint64_t *keys,
                   int32_t *idx_sorted, int32_t *overflow_out, int32_t *goff_incl /*out*/,
                   int32_t *owner, int32_t *slot_sorted, splat_stream_t stream);

/* ---- the reference's own two sort helpers, for callers that keep its sort_gaussian.py (cumsum -> keys -> torch.sort ->
 *      gather -> ranges): replace computeGaussianKey / computeTileGaussianRange (src/sort_gaussian.cu:72-146).
 *      gaussian_key[M] = tile << 32 | depth bits, gaussian_idx[M] = Gaussian id (M = tiles_cumsum[P-1]; caller-allocated);
 *      tile_range must be zero-filled by the caller (tiles without pairs keep (0,0)). ---- */
int splat_compute_gaussian_key(int P, const float *uv, const float *depth, const int32_t *radius,
                               const int32_t *tiles_cumsum, int W, int H, int64_t *gaussian_key, int32_t *gaussian_idx,
                               splat_stream_t stream);
int splat_compute_tile_gaussian_range(int64_t M, const int64_t *key_sorted, int32_t *tile_range, splat_stream_t stream);

/* ---- alpha blending : replaces alphaBlendingForward/Backward, ...Enhanced, ...WithBias
 *      (src/alpha_blending.cu:251-582, src/alpha_blending_enhanced.cu:275-627,
 *       src/alpha_blending_with_bias.cu:266-621).
 *      opacity_bias NULL -> plain; gs_idx NULL / K<=0 -> not enhanced. Channels are processed in
 *      chunks of <=32 exactly as the reference does (matters for dL_dabs_uv only). ---- */
/* pack_scratch: P * splat_blend_pack_floats(C) floats, 64-byte aligned, uninitialised: the library first
 * packs {uv, conic, opacity, bias, id, features of the chunk} into one record per Gaussian so that the
 * tile kernels gather one contiguous record per list entry. */
size_t splat_blend_pack_floats(int C);
int splat_alpha_blending_forward(int P, int C, const float *uv, const float *conic, const float *opacity,
                                 const float *feature, const float *opacity_bias, const int32_t *idx_sorted,
                                 const int32_t *tile_range, float bg,
                                 const float *bg_channels /*NULL, or a background per channel [C] overriding bg: several
                                    feature sets of one geometry composited in ONE pass (depth with bg 1 next to rgb)*/,
                                 int W, int H, int K, int enable_truncation, float *out, float *final_T,
                                 int32_t *ncontrib,
                                 int32_t *gs_idx /*[H,W,K] (unused slots are set to -1), or NULL*/,
                                 float *pack_scratch, splat_stream_t stream);
/* dL_dfeature is [P,C]; dL_dopacity_bias NULL unless bias given; dL_dabs_uv may be NULL (the sums of
 * |d uv| are only needed when the caller's abs_ndc tap is present -- skipping them saves two of the
 * per-splat wave reductions).
 * Two modes:
 *  - atomic mode (goff_incl / slot_sorted / pair_scratch NULL): wave-reduced hardware float atomics;
 *    all gradient outputs must be zero-init.
 *  - pair mode (all three given; goff_incl / slot_sorted from splat_bin_sort for THIS idx_sorted;
 *    pair_scratch = M * splat_blend_pair_floats(C, bias != NULL) floats, 64-byte aligned,
 *    uninitialised): no global atomics, every gradient element is written (no zero-init needed). */
size_t splat_blend_pair_floats(int C, int has_bias);
int splat_alpha_blending_backward(int P, int C, const float *uv, const float *conic, const float *opacity,
                                  const float *feature, const float *opacity_bias, const int32_t *idx_sorted,
                                  const int32_t *tile_range, float bg, int W, int H, const float *final_T,
                                  const int32_t *ncontrib, const float *dL_dout, float *dL_duv, float *dL_dabs_uv,
                                  float *dL_dconic, float *dL_dopacity, float *dL_dfeature,
                                  float *dL_dopacity_bias,
                                  float *dL_dndc /*NULL, or (pair mode) the tap dL_duv * [W/2, H/2] written alongside*/,
                                  float *dL_dabs_ndc /*NULL, or (pair mode, with dL_dabs_uv) its abs twin*/,
                                  const int32_t *goff_incl, const int32_t *slot_sorted,
                                  float *pair_scratch, float *pack_scratch,
                                  int pack_is_valid /*pack_scratch still holds the forward's records (C <= 32)*/,
                                  float *dbg_T_front /*NULL, or [H,W]: the transmittance the replay reaches in front of
                                    each pixel's first splat -- 1 up to rounding iff the backward reproduced every
                                    inclusion decision of the forward (a flipped decision is off by >= 1/255)*/,
                                  splat_stream_t stream);

/* The same two calls with the forward's cull decisions handed to the backward (ABI 17): cull_flags[M] receives one word per
 * sorted entry (byte b = the 4x4 quarters of the tile's 8x8 block b the splat can reach); a backward that gets them walks one
 * survivor list per quarter instead of culling again (narrow rows without |taps|, rows of 16 .. 32 channels).  NULL: as above.
 * Reference: the same operators, src/submodules/dptr/dptr/gs/src/alpha_blending.cu:16-110 (forward), :150-249 (backward). */
int splat_alpha_blending_forward_flags(int P, int C, const float *uv, const float *conic, const float *opacity,
                                       const float *feature, const float *opacity_bias,
                                       const int32_t *idx_sorted, const int32_t *tile_range, float bg,
                                       const float *bg_channels, int W, int H, int K, int enable_truncation,
                                       float *out, float *final_T, int32_t *ncontrib, int32_t *gs_idx,
                                       float *pack_scratch, uint32_t *cull_flags, splat_stream_t stream);
int splat_alpha_blending_backward_flags(int P, int C, const float *uv, const float *conic, const float *opacity,
                                        const float *feature, const float *opacity_bias,
                                        const int32_t *idx_sorted, const int32_t *tile_range, float bg, int W,
                                        int H, const float *final_T, const int32_t *ncontrib,
                                        const float *dL_dout, float *dL_duv, float *dL_dabs_uv, float *dL_dconic,
                                        float *dL_dopacity, float *dL_dfeature, float *dL_dopacity_bias,
                                        float *dL_dndc, float *dL_dabs_ndc, const int32_t *goff_incl,
                                        const int32_t *slot_sorted, float *pair_scratch, float *pack_scratch,
                                        int pack_is_valid, float *dbg_T_front, const uint32_t *cull_flags,
                                        splat_stream_t stream);

/* ---- per-frame evaluation of the dynamic Gaussians --------------------------------------
 * Replaces the eager torch of src/dynamic_gaussian_with_base_point_cloud.py:
 *   get_position :236-250  pos_t = position + c3 + c2 d + c1 d^2 + c0 d^3, c = cubic[N,4,I,3][:, :, seg]
 *   get_rotation :184-198  rot_t = normalize(rotation + sum_k rot_poly[N,4,4][:,k] t^k + sum_m rot_fourier[N,8,4][:,m] f_m)
 *                          (both sums are detached in the reference: rot_poly / rot_fourier receive no gradient)
 *   get_opacity  :171-172  opa_t = sigmoid(opacity) ;  get_scaling :175-177  scl_t = exp(scaling)
 * seg, d and the 12 basis values (t^0..t^3, cos(t k pi) k=1..4, sin(t k pi) k=1..4) are per-frame HOST scalars
 * (basis_host is a HOST pointer, copied into the kernel arguments; no device read of it).  All other pointers are
 * device pointers.  Any output (and, in the backward, any g_* / d_* pair) may be NULL to skip it.
 * cubic_layout: SPLAT_CUBIC_GAUSSIAN_MAJOR reads the reference's table as is (4 strided 12-byte reads per Gaussian,
 * one 128-byte line each); SPLAT_CUBIC_SEGMENT_MAJOR is the native layout (4x less HBM traffic for this row).
 * backward: accumulate=0 stores, accumulate=1 adds into the d_* buffers (gradient-bucket use).  d_cubic [N,4,I,3]
 * is touched ONLY in the active segment; with accumulate=0 the caller zero-fills it once if it needs the dense
 * gradient the reference's autograd produces. */
#define SPLAT_CUBIC_GAUSSIAN_MAJOR 0 /* cubic[N,4,I,3]: the reference's parameter layout (:73-75) */
#define SPLAT_CUBIC_SEGMENT_MAJOR 1  /* cubic[I,N,4,3]: one contiguous 48-byte record per Gaussian and frame */
int splat_dynamic_eval_forward(int P, int I, int seg, float d, const float *basis_host, const float *position,
                               const float *cubic, int cubic_layout, const float *rotation, const float *rot_poly,
                               const float *rot_fourier, const float *opacity, const float *scaling, float *pos_t,
                               float *rot_t, float *opa_t, float *scl_t, splat_stream_t stream);
int splat_dynamic_eval_backward(int P, int I, int seg, float d, const float *basis_host, const float *rotation,
                                const float *rot_poly, const float *rot_fourier, const float *opacity,
                                const float *scaling, const float *g_pos, const float *g_rot, const float *g_opa,
                                const float *g_scl, int accumulate, int cubic_layout, float *d_position, float *d_cubic,
                                float *d_rotation, float *d_opacity, float *d_scaling, splat_stream_t stream);

/* Polynomial + Fourier position model of the reference's first dynamic point cloud (src/dynamic_gaussian_points.py:169-186
 * get_position): pos_t = position + sum_k pos_poly_feat[N,4,3][:,k,:] t'^k + sum_m pos_fourier_feat[N,8,3][:,m,:] basis_m(t')
 * with the same 12 host basis values as the rotation (t'^0..3, cos(t' l pi), sin(t' l pi), l = 1..4).  Both tables
 * receive gradients (the reference does not detach them here). */
int splat_position_poly_fourier_forward(int P, const float *basis_host, const float *position, const float *pos_poly_feat,
                                        const float *pos_fourier_feat, float *pos_t, splat_stream_t stream);
int splat_position_poly_fourier_backward(int P, const float *basis_host, const float *g_pos, int accumulate,
                                         float *d_position, float *d_pos_poly_feat, float *d_pos_fourier_feat,
                                         splat_stream_t stream);

/* ---- fused per-frame preprocess of the orthographic renderer (rows a2 + a5 + a3 in one pass) -------------------
 * Replaces, per rendered frame, the eager-torch orthographic projection and EWA of the reference renderer plus its
 * compute_cov3d call (src/pointrix/renderer/dptr_ortho_enhanced.py:145-202 project_point, :282-310 call sites,
 * :18-111 ewa_project_torch_impl): pos = xyz (+ offset, may be NULL) -> uv, depth ; visible = depth != 0 ;
 * cov3d(scales, uquats) ; conic, radius, tiles.  Same results as splat_project_point_forward(ortho) +
 * splat_compute_cov3d_forward + splat_ewa_project_forward(ortho) on the same inputs; visible and cov3d are not
 * materialised.  backward: dL_dxyz from dL_duv / dL_ddepth (NULL = 0), dL_dscales / dL_duquats from dL_dconic;
 * any output may be NULL; accumulate=1 adds into the outputs (gradient-bucket use), 0 stores (every element). */
int splat_preprocess_ortho_forward(int P, const float *xyz, const float *offset, const float *scales,
                                   const float *uquats, const float *extr, int W, int H, float nearest, float extent,
                                   float *uv, float *depth, float *conic, int32_t *radius, int32_t *tiles,
                                   splat_stream_t stream);
int splat_preprocess_ortho_backward(int P, const float *xyz, const float *offset, const float *scales,
                                    const float *uquats, const float *extr, int W, int H, const float *depth,
                                    const int32_t *radius, const float *dL_duv, const float *dL_ddepth,
                                    const float *dL_dconic, int accumulate, float *dL_dxyz, float *dL_dscales,
                                    float *dL_duquats, splat_stream_t stream);

/* ---- dynamic Gaussians -> screen space in one pass (row a15 fused into the preprocess) --------
 * splat_dynamic_eval_forward + splat_preprocess_ortho_forward on the same inputs (time scalars as for
 * splat_dynamic_eval_*; scaling / opacity are the raw parameters: exp / sigmoid are applied inside); the per-frame
 * position, rotation and scale stay in registers.  Outputs: uv[P,2], depth[P,1], conic[P,3], radius[P], tiles[P] and
 * opa_t[P,1] = sigmoid(opacity) for the blend.  backward: gradients w.r.t. the parameters from dL_duv, dL_ddepth,
 * dL_dconic, dL_dopa (each may be NULL = 0); d_* may be NULL; accumulate=1 adds (d_cubic: active segment only, in the
 * given cubic_layout). */
int splat_frame_preprocess_forward(int P, int I, int seg, float d, const float *basis_host, const float *position,
                                   const float *cubic, int cubic_layout, const float *rotation, const float *rot_poly,
                                   const float *rot_fourier, const float *opacity, const float *scaling,
                                   const float *extr, int W, int H, float nearest, float extent, float *uv,
                                   float *depth, float *conic, int32_t *radius, int32_t *tiles, float *opa_t,
                                   splat_stream_t stream);
int splat_frame_preprocess_backward(int P, int I, int seg, float d, const float *basis_host, const float *position,
                                    const float *cubic, int cubic_layout, const float *rotation,
                                    const float *rot_poly, const float *rot_fourier, const float *opacity,
                                    const float *scaling, const float *extr, int W, int H, const float *depth,
                                    const int32_t *radius, const float *dL_duv, const float *dL_ddepth,
                                    const float *dL_dconic, const float *dL_dopa, int accumulate, float *d_position,
                                    float *d_cubic, float *d_rotation, float *d_opacity, float *d_scaling,
                                    splat_stream_t stream);

/* Spatial (Morton / Z-curve) order of the Gaussians: keys[i] = interleaved bits of (u / W, v / H) of uv[i] quantised to
 * 15 bits each, >= 0.  A stable argsort of the keys is the order in which the binning / compositing kernels see
 * neighbouring Gaussians next to each other in memory (densify.py::spatial_order / reorder_points apply it whenever the
 * per-Gaussian arrays are rebuilt).  Not in the reference; results do not depend on the order. */
int splat_morton_keys(int P, const float *uv, int W, int H, int32_t *keys, splat_stream_t stream);

/* ---- densification statistics and structure updates -----------------------------------------
 * accumulate: one frame of a batch -- viewspace_grad[P,2] += tap[P,2] * (sx, sy) (tap = the frame's ndc / abs_ndc
 *   gradient, or dL_duv with sx = W/2, sy = H/2; tap and viewspace_grad may both be NULL), visible[P] |= radius > 0,
 *   radii[P] = max(radii, radius)            (dptr_ortho_enhanced.py:425-431, atlas_gs_optimizer.py:414-433)
 * update: once per step, for visible Gaussians -- max_radii2D = max(., radii), pos_gradient_accum += |viewspace_grad|,
 *   denom += 1                                (atlas_gs_optimizer.py:110-121)
 * masks: clone = |g| >= thr & smax <= percent_dense*extent ; split = g >= thr & smax > percent_dense*extent ;
 *   prune = sigmoid(opacity) < min_opacity | max_radii2D > size_threshold | smax > 0.1*extent (size tests skipped when
 *   size_threshold <= 0), g = accum / denom with NaN -> 0, smax = max exp(scaling_raw)   (:199-251, :363-375);
 *   any of the three outputs may be NULL.
 * compact: index[i] = number of kept rows before i (one prefix sum of the byte mask, count = rows kept), then
 *   splat_compact_rows moves row i (row_words 32-bit words) of any per-Gaussian tensor to row index[i] of dst when
 *   mask[i] != 0 (points.py:282-312 prune_optimizer: parameters, Adam moments, statistics). */
int splat_densify_accumulate(int P, const int32_t *radius, const float *tap, float sx, float sy, float *viewspace_grad,
                             uint8_t *visible, int32_t *radii, splat_stream_t stream);
int splat_densify_update(int P, const uint8_t *visible, const float *viewspace_grad, const int32_t *radii,
                         float *max_radii2D, float *pos_gradient_accum, float *denom, splat_stream_t stream);
int splat_densify_masks(int P, const float *pos_gradient_accum, const float *denom, const float *max_radii2D,
                        const float *scaling_raw, const float *opacity_raw, float grad_threshold, float percent_dense,
                        float cameras_extent, float min_opacity, float size_threshold, uint8_t *clone, uint8_t *split,
                        uint8_t *prune, splat_stream_t stream);
size_t splat_compact_scratch_bytes(int P);
int splat_compact_scan(int P, const uint8_t *mask, int32_t *index, int32_t *count, void *scratch,
                       splat_stream_t stream);
int splat_compact_rows(int P, const uint8_t *mask, const int32_t *index, int row_words, const void *src, void *dst,
                       splat_stream_t stream);
/* ---- clone / split on the device (reference: atlas_gs_optimizer.py:255-349 new_pos_scale /
 *      densify_clone / densify_split, points.py:225-395 extend / remove with the optimiser state).  With the prefix of a
 *      selection mask (splat_compact_scan: index, n_sel) every per-Gaussian tensor -- parameters and Adam moments alike --
 *      is extended by the same gather: dst row (r * n_sel + index[i]) = src row i, r < repeat (torch's
 *      x[mask].repeat(r, 1, ..)); dst = the append area behind the existing rows. ---- */
int splat_gather_rows_repeat(int P, const uint8_t *mask, const int32_t *index, int n_sel, int repeat, int row_words,
                             const void *src, void *dst, splat_stream_t stream);
/* positions and (log) scales of the split_num children of every selected Gaussian: R(rotation) (z * scale) + position,
 * log(scale / (0.8 split_num)).  z: unit normals from Philox-4x32-10 keyed by `seed` with counter (Gaussian id, replica):
 * no generator state, so all data-parallel ranks draw identical children; or, for parity tests, caller-supplied
 * unit_normals [split_num * n_sel, 3] in output-row order. */
int splat_densify_split_sample(int P, const uint8_t *mask, const int32_t *index, int n_sel, int split_num, uint64_t seed,
                               const float *position, const float *scaling_raw, const float *rotation_raw,
                               const float *unit_normals, float *new_pos, float *new_scaling, splat_stream_t stream);


/* ---- exact K nearest neighbours on a uniform grid ---------------------------------------------
 * Replaces pytorch3d.ops.knn_points(points[None], points[None], K=K+1) of src/geometry_utils.py:17-19 (un-vendored CUDA
 * dependency): squared Euclidean distances of the K nearest points per query, ascending, with indices (ties: smaller
 * index first; fewer than K points: 0 / -1 padding).
 *   budget = splat_knn_grid_cells(M)                 cell budget of the grid (the device picks an isotropic cell width
 *                                                    and per-axis counts that fit it: no host round trip)
 *   splat_knn_build   -> plan (opaque, splat_knn_plan_bytes() bytes), cell_of[M], cell_count[budget + 1] (caller zero-fills)
 *   caller: cell_start = exclusive prefix sum of cell_count (budget + 1 entries, last = M) with any device scan
 *   splat_knn_scatter -> sorted[M,4] = (x, y, z, bits of the original index), cell by cell (fill: zeroed [budget] scratch)
 *   splat_knn_search  -> dists[N,K], idx[N,K]; query_order (nullable) = order in which the queries are walked
 *                        (pass a spatially coherent permutation; results land at the original query index); K <= 16 */
int splat_knn_grid_cells(int M);
size_t splat_knn_plan_bytes(void);
int splat_knn_build(int M, const float *points, int budget, void *plan, int32_t *cell_of, int32_t *cell_count,
                    splat_stream_t stream);
int splat_knn_scatter(int M, const float *points, const int32_t *cell_of, const int32_t *cell_start, int32_t *fill,
                      float *sorted, splat_stream_t stream);
int splat_knn_search(int N, const float *query, const int32_t *query_order, int M, const float *sorted,
                     const int32_t *cell_start, const void *plan, int K, float *dists, int32_t *idx,
                     splat_stream_t stream);

/* ---- frame batch: F frames of ONE Gaussian set per launch (SURVEY 7 stage 6 / 8f: the reference renders the frames
 *      of a batch one after the other, dptr_ortho_enhanced.py:385-433, ~13 launches each; here every kernel of the
 *      per-frame path takes the frame as a grid dimension, so a batch costs the launches of one frame, the short
 *      kernels fill the chip, and the compositing kernels see F * T tiles -- no half-empty last round).
 *      Layout: per-Gaussian arrays [F,P,..]; tile_range [F,T,2] (positions relative to the frame's segment);
 *      idx_sorted / slot_sorted / keys / owner [F,capacity]; images [F,C,H,W]; pair records [F,capacity,stride]. ---- */
int splat_preprocess_ortho_forward_batch(int F, int P, const float *xyz, const float *offsets /*[F,P,3]*/,
                                         const float *scales, const float *uquats, const float *extr, int W, int H,
                                         float nearest, float extent, float *uv, float *depth, float *conic,
                                         int32_t *radius, splat_stream_t stream);
/* scratch: F * splat_bin_scratch_bytes(P, W, H) bytes; M_out[F] = pairs of each frame */
int splat_bin_count_batch(int F, int P, const float *uv, const int32_t *radius, int W, int H, void *scratch,
                          int32_t *tile_range, int32_t *M_out, splat_stream_t stream);
int splat_bin_sort_batch(int F, int P, const float *uv, const float *depth, const int32_t *radius, int W, int H,
                         void *scratch, int32_t *tile_range, int64_t capacity, uint64_t *keys, int32_t *idx_sorted,
                         int32_t *overflow_out, int32_t *goff_incl /*[F,P] out*/, int32_t *owner, int32_t *slot_sorted,
                         splat_stream_t stream);
/* The same two steps with REACH masks (ABI 21).  The reference creates a pair for every tile of a splat's bounding square
 * (include/utils.h:17-37) and lets alpha_blending skip, per pixel, what stays below alpha = 1/255 (src/alpha_blending.cu:78-95);
 * a third of those pairs reaches no pixel of its tile.  With conic [F,P,3] and opacity ([P]: opacity_frame_stride 0, or [F,P]:
 * stride P) the count step keeps only the tiles whose rectangle of pixel centres the alpha >= 1/255 ellipse can touch (the
 * compositing kernels' own conservative test; a rectangle of 32 tiles and more is tested in cells of c x c tiles, c the smallest
 * power of two that leaves at most 31 cells) and leaves one word per Gaussian in reach[F,P] for the sort step.  tile_range,
 * M_out, gcount (optional, [F,P]), goff_incl and the pair slots count the KEPT pairs.  Images, ids and gradients composited
 * from the shorter lists are those of the full lists bit for bit; list positions (ncontrib) differ.  Not for callers that
 * return idx_sorted / tile_range as the reference's sort_gaussian result.  F = 1: a single frame. */
int splat_bin_count_batch_reach(int F, int P, const float *uv, const int32_t *radius, const float *conic,
                                const float *opacity, int64_t opacity_frame_stride, int W, int H, void *scratch,
                                int32_t *tile_range, int32_t *M_out, int32_t *gcount, uint32_t *reach, splat_stream_t stream);
int splat_bin_sort_batch_reach(int F, int P, const float *uv, const float *depth, const int32_t *radius, const uint32_t *reach,
                               int W, int H, void *scratch, int32_t *tile_range, int64_t capacity, uint64_t *keys,
                               int32_t *idx_sorted, int32_t *overflow_out, int32_t *goff_incl /*[F,P] out*/, int32_t *owner,
                               int32_t *slot_sorted, splat_stream_t stream);
/* C <= 32.  opacity / feature: stride in elements between two frames' arrays, 0 = shared by all frames.
 * pack_scratch: F * P * splat_blend_pack_floats(C) floats (kept for the backward). */
int splat_alpha_blending_forward_batch(int F, int P, int C, const float *uv, const float *conic, const float *opacity,
                                       int64_t opacity_frame_stride, const float *feature, int64_t feature_frame_stride,
                                       const int32_t *idx_sorted, const int32_t *tile_range, int64_t capacity, float bg,
                                       const float *bg_channels, int W, int H, int K, int enable_truncation, float *out,
                                       float *final_T, int32_t *ncontrib, int32_t *gs_idx, float *pack_scratch,
                                       uint32_t *cull_flags /*NULL or [F,capacity]: see below*/, splat_stream_t stream);
/* exact stride (floats) of a pair record for this configuration: [ux uy ca cb cc o | ax ay (abs) | bias | features] padded
 * to whole 16-byte chunks */
size_t splat_blend_pair_stride(int C, int want_abs, int has_bias);
/* tile kernels of the backward only: one gradient record per (frame, tile, splat) pair at frame * capacity + slot
 * (pair_records: F * capacity * splat_blend_pair_stride(C, want_abs, 0) floats, uninitialised).  `pack` = the packed
 * records the forward left in its pack_scratch. */
int splat_alpha_blending_backward_batch(int F, int P, int C, const int32_t *idx_sorted, const int32_t *tile_range,
                                        int64_t capacity, float bg, int W, int H, const float *final_T,
                                        const int32_t *ncontrib, const float *dL_dout, int want_abs,
                                        const int32_t *slot_sorted, float *pair_records, const float *pack,
                                        const uint32_t *cull_flags /*NULL or the forward's*/,
                                        float *dbg_T_front /*NULL or [F,H,W]*/, splat_stream_t stream);
/* cull_flags: one 32-bit word per sorted tile entry (frame stride = capacity), byte w != 0 = the forward kept the entry for
 * the tile's 8x8 block w.  The forward writes them when the pointer is given; the backward then reads them instead of repeating the
 * cull (the same decisions: both passes skip exactly the splats that cannot reach alpha >= 1/255 in a block, and the
 * blocks that were saturated).  NULL on either side: that pass culls for itself. */
/* Gaussian-side backward of a batch of static Gaussians + per-frame offsets under the orthographic camera: sums every
 * Gaussian's pair records over all frames and runs the preprocess backward (projection, EWA, cov3d: linear in the
 * summed dL_duv / dL_dconic because conic and Jacobian do not depend on the frame) once.  Replaces, per batch, F x
 * (pair reduce + splat_preprocess_ortho_backward).  accumulate: add into the parameter gradients (gradient sinks).
 * tap / abs_tap (optional, [P,2]): sum over the frames of the densification taps dL_duv * (W/2, H/2) and its abs twin
 * (what the reference accumulates from ndc.grad / abs_ndc.grad, frag_model.py:326-343); radii_max (optional, [P], needs
 * radius [F,P]): max over the frames of the screen radius. */
int splat_frames_gauss_backward_static(int F, int P, int C, int W, int H, int64_t capacity, int want_abs,
                                       const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                       const float *xyz, const float *scales, const float *uquats, const float *extr,
                                       int accumulate, float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                       float *d_feature, float *tap, float *abs_tap, int32_t *radii_max,
                                       splat_stream_t stream);

/* One call per direction for a whole batch (static Gaussians + per-frame offsets, orthographic camera, one feature set):
 * the sequences above behind a single crossing of the ABI.  splat_frames_t holds every pointer; set struct_bytes =
 * sizeof(splat_frames_t).  splat_frames_count = preprocess + pair counts (pairs[F], device) to size `capacity` before the
 * first splat_frames_forward; forward = preprocess, binning, sort, pack, compositing; backward = tile kernels +
 * Gaussian-side reduction (accumulate: add into the parameter gradients). */
/* Camera of a frame batch.  perspective = 0: the orthographic camera of the video renderer (intr unused;
 * dptr_ortho_enhanced.py:177-202); 1: the pinhole camera of gs.rasterization / DPTRRender (project_point.cu,
 * ewa_project.cu).  Every frame may have its own camera -- render_batch gives each batch element its own
 * (dptr_ortho_enhanced.py:409-411): frame f reads extr + f * extr_frame_stride (12 floats: the first three rows of the
 * world-to-camera matrix) and intr + f * intr_frame_stride (fx fy cx cy); stride 0 = one camera for the batch.  With one
 * orthographic camera the Gaussian-side backward runs its projection chain once on the sums of all frames; otherwise once
 * per frame, and then needs the forward's `offsets` [F,P,3] (NULL: none were used). */
typedef struct splat_camera_t {
    int32_t perspective;
    const float *intr;
    int64_t intr_frame_stride;
    const float *extr;
    int64_t extr_frame_stride;
    const float *offsets;
} splat_camera_t;
int splat_preprocess_forward_batch_cam(int F, int P, const float *xyz, const float *offsets, const float *scales,
                                       const float *uquats, const splat_camera_t *cam, int W, int H, float nearest,
                                       float extent, float *uv, float *depth, float *conic, int32_t *radius,
                                       splat_stream_t stream);
/* one feature set (arguments of splat_frames_gauss_backward_static_set) / the three sets of one pass under any camera */
int splat_frames_gauss_backward_static_cam(int F, int P, int cn, int W, int H, int64_t capacity, int want_abs,
                                           const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                           const float *xyz, const float *scales, const float *uquats,
                                           const splat_camera_t *cam, int accumulate, float *d_xyz, float *d_scales,
                                           float *d_uquats, float *d_opacity, float *d_feature /*NULL: none*/,
                                           int feature_stride, int skip_opacity, int depth_channel, float *tap,
                                           float *abs_tap, int32_t *radii_max, splat_stream_t stream);
int splat_frames_gauss_backward_static_sets_cam(int F, int P, int C, int W, int H, int64_t capacity,
                                                const float *pair_records, const int32_t *goff_incl,
                                                const int32_t *radius, const float *xyz, const float *scales,
                                                const float *uquats, const splat_camera_t *cam, int accumulate,
                                                float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                                const int32_t *set_c0, const int32_t *set_cn, float *const *set_dfeature,
                                                const int32_t *set_stride, int depth_channel, float *tap, float *abs_tap,
                                                int32_t *radii_max, splat_stream_t stream);
/* The pinhole camera's operator chain of gs.rasterization (project_point -> compute_cov3d -> ewa_project:
 * src/submodules/dptr/dptr/gs/__init__.py:55-77) fused into one pass per direction, like splat_preprocess_ortho_*; the
 * backward returns the position gradient through the projection and the EWA Jacobian (camera gradients: the separate
 * operators above). */
int splat_preprocess_persp_forward(int P, const float *xyz, const float *offset, const float *scales, const float *uquats,
                                   const float *intr, const float *extr, int W, int H, float nearest, float extent,
                                   float *uv, float *depth, float *conic, int32_t *radius, int32_t *tiles,
                                   splat_stream_t stream);
int splat_preprocess_persp_backward(int P, const float *xyz, const float *offset, const float *scales, const float *uquats,
                                    const float *intr, const float *extr, int W, int H, const float *depth,
                                    const int32_t *radius, const float *dL_duv, const float *dL_ddepth,
                                    const float *dL_dconic, int accumulate, float *dL_dxyz, float *dL_dscales,
                                    float *dL_duquats, splat_stream_t stream);

typedef struct splat_frames_t {
    size_t struct_bytes;
    int32_t F, P, W, H, C;
    int32_t want_abs, accumulate;
    int64_t capacity;                 /* pairs reserved per frame */
    float nearest, extent, bg;
    splat_stream_t stream;
    /* scene (shared by the frames) + per-frame offsets [F,P,3] */
    const float *xyz, *offsets, *scales, *uquats, *opacity, *feature, *extr;
    /* per-frame geometry [F,P,..] */
    float *uv, *depth, *conic;
    int32_t *radius;
    /* binning / sort */
    void *bin_scratch;                /* F * splat_bin_scratch_bytes(P, W, H) */
    int32_t *tile_range, *pairs, *overflow, *goff_incl, *owner, *idx_sorted, *slot_sorted;
    uint64_t *keys;
    /* compositing */
    float *pack;                      /* F * P * splat_blend_pack_floats(C) */
    float *out, *final_T;             /* [F,C,H,W], [F,H,W] */
    int32_t *ncontrib;
    /* backward */
    const float *dL_dout;             /* [F,C,H,W] */
    float *pair_records;              /* F * capacity * splat_blend_pair_stride(C, want_abs, 0) */
    float *d_xyz, *d_scales, *d_uquats, *d_opacity, *d_feature;
    float *tap, *abs_tap;             /* optional [P,2] */
    int32_t *radii_max;               /* optional [P] */
    float *dbg_T_front;               /* optional [F,H,W] */
    uint32_t *cull_flags;              /* optional [F,capacity]: the forward's cull decisions, reused by the backward */
    /* camera (ABI 16): `extr` above per frame when extr_frame_stride != 0; perspective = 1 with intr (fx fy cx cy) */
    int64_t extr_frame_stride, intr_frame_stride;
    const float *intr;
    int32_t perspective;
    uint32_t *reach;                  /* optional [F,P] (ABI 21): binning with reach masks (splat_bin_count_batch_reach) */
} splat_frames_t;
int splat_frames_count(const splat_frames_t *batch);
int splat_frames_forward(const splat_frames_t *batch);
int splat_frames_backward(const splat_frames_t *batch);

/* Several feature SETS of one geometry in a frame batch -- the reference renderer's three blends (rgb through
 * alpha_blending_enhanced with the taps; depth, bg = 1; the extra attributes with opacity.detach(),
 * dptr_ortho_enhanced.py:331-375): ONE forward over the concatenated row [F,P,C] (splat_alpha_blending_forward_batch with
 * per-channel backgrounds and K ids), then per set [c0, c0 + cn) one backward pass of the tile kernels
 * (..._backward_batch_set: packs the set's own records, pair stride splat_blend_pair_stride(cn, want_abs, 0)) and one
 * Gaussian-side reduction (..._gauss_backward_static_set: accumulate = 1 after the first set; skip_opacity for a set
 * blended with a detached opacity; depth_channel >= 0 when that channel is the per-frame depth feature, whose gradient
 * goes to the position through the projection; taps only from the set that feeds them). */
int splat_alpha_blending_backward_batch_set(int F, int P, int C, int c0, int cn, const float *uv, const float *conic,
                                            const float *opacity, int64_t opacity_frame_stride, const float *feature,
                                            int64_t feature_frame_stride, const int32_t *idx_sorted,
                                            const int32_t *tile_range, int64_t capacity, float bg, int W, int H,
                                            const float *final_T, const int32_t *ncontrib, const float *dL_dout,
                                            int want_abs, const int32_t *slot_sorted, float *pair_records,
                                            float *pack_scratch, float *dbg_T_front, splat_stream_t stream);
int splat_frames_gauss_backward_static_set(int F, int P, int cn, int W, int H, int64_t capacity, int want_abs,
                                           const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                           const float *xyz, const float *scales, const float *uquats, const float *extr,
                                           int accumulate, float *d_xyz, float *d_scales, float *d_uquats,
                                           float *d_opacity, float *d_feature /*NULL: none*/, int feature_stride,
                                           int skip_opacity, int depth_channel, float *tap, float *abs_tap,
                                           int32_t *radii_max, splat_stream_t stream);

/* The same three blends' backward in ONE pass of the tile kernels (the sets share the alpha / transmittance replay; only
 * the routing of dL/dalpha differs: d uv, d conic <- every set; d opacity <- tap set + second set; taps <- tap set).
 * set 0 = the tap set (<= 4 channels), set 1 = a second set blended with the live opacity (<= 4), set 2 = the set blended
 * with opacity.detach() (<= 20); set_cn[g] = 0: no such set; the sets must tile the row's C <= 28 channels.  set_c0 /
 * set_cn / set_bg / set_stride / set_dfeature are HOST arrays of three entries.  Pair records: stride
 * splat_blend_sets_pair_stride(C) floats, layout [ux uy ca cb | cc o ax ay | tx ty | dL_dfeature[0..C-1]];
 * pack_scratch: F * P * splat_blend_sets_pack_floats() floats.  Other widths / routings: the per-set calls above. */
/* forward of a row whose feature sets live in their own tensors (no concatenated [F,P,C] row; same tables as below, in any
 * routing: set g = row channels [set_c0[g], + set_cn[g]) from set_feature[g]); otherwise splat_alpha_blending_forward_batch */
int splat_alpha_blending_forward_batch_sources(int F, int P, int C, int nsrc, const splat_feature_source_t *src,
                                               const float *uv, const float *conic, const float *opacity,
                                               int64_t opacity_frame_stride, const int32_t *idx_sorted,
                                               const int32_t *tile_range, int64_t capacity, const float *bg_channels, int W,
                                               int H, int K, int enable_truncation, float *out, float *final_T,
                                               int32_t *ncontrib, int32_t *gs_idx, float *pack_scratch, uint32_t *cull_flags,
                                               splat_stream_t stream);
int splat_alpha_blending_forward_batch_sets(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                            const float *const *set_feature, const int64_t *set_feature_fs,
                                            const float *uv, const float *conic, const float *opacity,
                                            int64_t opacity_frame_stride, const int32_t *idx_sorted,
                                            const int32_t *tile_range, int64_t capacity, const float *bg_channels, int W,
                                            int H, int K, int enable_truncation, float *out, float *final_T,
                                            int32_t *ncontrib, int32_t *gs_idx, float *pack_scratch, uint32_t *cull_flags,
                                            splat_stream_t stream);
size_t splat_blend_sets_pair_stride(int C);
/* 1 when splat_alpha_blending_backward_batch_sets_packed stages the forward's packed records for this plan (the renderer's own:
   rgb 0-2 | depth 3 | 19 attributes 4-22, with cull words, options "sets_std" and "bwd_quarters" on): no pack_scratch needed */
int splat_blend_sets_uses_forward_pack(int C, const int32_t *set_c0, const int32_t *set_cn, int has_cull_flags);
size_t splat_blend_sets_pack_floats(void);
int splat_alpha_blending_backward_batch_sets(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                             const float *set_bg, const float *uv, const float *conic,
                                             const float *opacity, int64_t opacity_frame_stride,
                                             const float *feature /* row [F,P,C], or NULL: */, int64_t feature_frame_stride,
                                             const float *const *set_feature /* [P,cn] per set */,
                                             const int64_t *set_feature_fs /* floats between frames, 0 = shared */,
                                             const int32_t *idx_sorted, const int32_t *tile_range, int64_t capacity, int W,
                                             int H, const float *final_T, const int32_t *ncontrib,
                                             const float *dL_dout /* [F,C,H,W], or NULL: */,
                                             const float *const *set_dL /* [F,cn,H,W] per set */, int want_abs,
                                             const int32_t *slot_sorted, float *pair_records,
                                             float *pack_scratch, const uint32_t *cull_flags /*NULL or the forward's*/,
                                             float *dbg_T_front, splat_stream_t stream);

/* splat_alpha_blending_backward_batch_sets with the packed records the FORWARD left for this row (forward_pack: the
   pack_scratch of splat_alpha_blending_forward_batch_sets -- or of splat_alpha_blending_forward[_flags] at F = 1 -- called
   with the same C channels, untouched since; F * P * splat_blend_pack_floats(C) floats).  For the reference renderer's own
   plan (rgb at row channels 0-2 with the taps, depth at channel 3, 19 attributes blended with opacity.detach() at channels
   4-22: dptr_ortho_enhanced.py:331-375) with the forward's cull words the tile kernel stages those records directly and no
   packing launch runs; any other plan, or forward_pack = NULL, behaves like splat_alpha_blending_backward_batch_sets
   (pack_scratch is still required then).  Round 6: with forward_pack given, `feature` NULL and a NULL set_feature pointer of a
   set that has channels (a row described by feature SOURCES: no tensor per set exists), the packing launch of such a plan takes
   the row's channels out of the forward's records -- every one-pass plan serves feature lists / per-frame tensors. */
int splat_alpha_blending_backward_batch_sets_packed(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                                    const float *set_bg, const float *uv, const float *conic,
                                                    const float *opacity, int64_t opacity_frame_stride,
                                                    const float *feature, int64_t feature_frame_stride,
                                                    const float *const *set_feature, const int64_t *set_feature_fs,
                                                    const int32_t *idx_sorted, const int32_t *tile_range, int64_t capacity,
                                                    int W, int H, const float *final_T, const int32_t *ncontrib,
                                                    const float *dL_dout, const float *const *set_dL, int want_abs,
                                                    const int32_t *slot_sorted, float *pair_records, float *pack_scratch,
                                                    const uint32_t *cull_flags, float *dbg_T_front, const float *forward_pack,
                                                    splat_stream_t stream);
/* ... with the L1 image loss FUSED into the tile kernel's hoist of the image gradient (ABI 20): set_target[g] = the target image
 * [F, cn, H, W] of set g (instead of a gradient image), pred_row = the forward's output row [F, C, H, W]; the gradient of a channel
 * of set g is l1_scale_host[g] * sign(pred - target), and l1_sum[(f * tiles + t) * 3 + g] (optional, [F, tiles, 3], tiles =
 * ceil(W/16) * ceil(H/16): every entry is written) = sum |pred - target| over tile t of frame f, set g: the caller adds them up
 * (no atomics: bit-reproducible).  Replaces three
 * splat_l1_loss_grad launches and the gradient images written and read back between them and the backward
 * (src/trainer_fragGS.py:573-600: l1_loss on the rendered images).  Needs the forward's cull words (quarter-list kernels). */
int splat_alpha_blending_backward_batch_sets_l1(int F, int P, int C, const int32_t *set_c0, const int32_t *set_cn,
                                                const float *set_bg, const float *uv, const float *conic, const float *opacity,
                                                int64_t opacity_frame_stride, const float *const *set_feature,
                                                const int64_t *set_feature_fs, const int32_t *idx_sorted,
                                                const int32_t *tile_range, int64_t capacity, int W, int H, const float *final_T,
                                                const int32_t *ncontrib, const float *pred_row, const float *const *set_target,
                                                const float *l1_scale_host, float *l1_sum, int want_abs,
                                                const int32_t *slot_sorted, float *pair_records, float *pack_scratch,
                                                const uint32_t *cull_flags, float *dbg_T_front, const float *forward_pack,
                                                splat_stream_t stream);

/* out[i, 0 .. ncp) = sum of the pair records (stride ncp floats, a multiple of 4; 16-byte aligned) in Gaussian i's slots
   [goff_incl[i-1], goff_incl[i]) -- any record layout.  With splat_alpha_blending_backward_batch_sets at F = 1 this is the
   reduction of the single-frame operator gs.alpha_blending_shared (the reference's three blends of render_iter,
   dptr_ortho_enhanced.py:331-375, as separate operators: gradients w.r.t. uv / conic / opacity / features go back to autograd);
   the caller slices the summed SETS record.  out [P, ncp] fully written; no atomics. */
int splat_pair_records_segment_sum(int P, int ncp, const float *pair_records, const int32_t *goff_incl, float *out,
                                   splat_stream_t stream);
int splat_frames_gauss_backward_static_sets(int F, int P, int C, int W, int H, int64_t capacity, const float *pair_records,
                                            const int32_t *goff_incl, const int32_t *radius, const float *xyz,
                                            const float *scales, const float *uquats, const float *extr, int accumulate,
                                            float *d_xyz, float *d_scales, float *d_uquats, float *d_opacity,
                                            const int32_t *set_c0, const int32_t *set_cn, float *const *set_dfeature,
                                            const int32_t *set_stride, int depth_channel /* row channel or -1 */,
                                            float *tap, float *abs_tap, int32_t *radii_max, splat_stream_t stream);

/* Frame batch of DYNAMIC Gaussians (rows a15 + f1: the per-frame evaluation of the reference's spline point cloud inside
 * the preprocess, for all frames of a batch at once).  tab: F entries of 64 bytes in DEVICE memory, one per frame:
 * {int32 seg; float d; float basis[12]; float pad[2]} (segment index, offset inside it, t'^0..3, cos / sin(t' l pi);
 * pad[0], optional, as int32: 1 + the index of the frame splat_dynamic_positions_batch_backward visits at this step of its walk --
 * a PERMUTATION of the frames that puts the frames of one segment next to each other -- or 0 in every entry: table order).
 * Forward: uv / depth / conic / radius [F,P,..], opa_t [P] (sigmoid(opacity): frame independent).
 * Backward (after splat_alpha_blending_backward_batch): one quad per Gaussian walks all frames -- sums the frame's pair
 * records, re-evaluates position / rotation of that frame, projection + EWA + cov3d backward, activations -- and ADDS
 * the parameter gradients into d_* (zero-filled or gradient sinks; d_cubic in the table's layout, only the segments
 * the batch touches). */
int splat_frame_preprocess_forward_batch(int F, int P, int I, const void *tab, const float *position, const float *cubic,
                                         int cubic_layout, const float *rotation, const float *rot_poly,
                                         const float *rot_fourier, const float *opacity, const float *scaling,
                                         const float *extr, int W, int H, float nearest, float extent, float *uv,
                                         float *depth, float *conic, int32_t *radius, float *opa_t, splat_stream_t stream);
int splat_frames_gauss_backward_dynamic(int F, int P, int I, int C, int W, int H, int64_t capacity, int want_abs,
                                        const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                        const void *tab, const float *position, const float *cubic, int cubic_layout,
                                        const float *rotation, const float *rot_poly, const float *rot_fourier,
                                        const float *opacity, const float *scaling, const float *extr, float *d_position,
                                        float *d_cubic, float *d_rotation, float *d_opacity, float *d_scaling,
                                        float *d_feature, float *tap, float *abs_tap, int32_t *radii_max,
                                        splat_stream_t stream);
/* The same for the SETS records of splat_alpha_blending_backward_batch_sets (the reference's real training frame: its dynamic
 * Gaussians through render_iter's three blends): taps from the tap set, feature gradients ADDED per set (set_dfeature: HOST
 * array of three device pointers, NULL entries skipped), row channel depth_channel (>= 0) = the per-frame depth, whose
 * gradient reaches the position through the projection. */
int splat_frames_gauss_backward_dynamic_sets(int F, int P, int I, int C, int W, int H, int64_t capacity,
                                             const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                             const void *tab, const float *position, const float *cubic, int cubic_layout,
                                             const float *rotation, const float *rot_poly, const float *rot_fourier,
                                             const float *opacity, const float *scaling, const float *extr,
                                             float *d_position, float *d_cubic, float *d_rotation, float *d_opacity,
                                             float *d_scaling, const int32_t *set_c0, const int32_t *set_cn,
                                             float *const *set_dfeature, const int32_t *set_stride, int depth_channel,
                                             float *tap, float *abs_tap, int32_t *radii_max, splat_stream_t stream);

/* ---- optimiser step of the frame-sharded data-parallel renderer: replaces the per-group
 *      torch.optim.Adam.step() the reference reaches through src/pointrix/optimizer/optimizer.py:70-83 (Adam built in
 *      atlas_gs_optimizer / configs with eps = 1e-15, one learning rate per parameter group), as one launch over the
 *      flat buffer the gradient all-reduce just summed.  No weight decay, no amsgrad.
 *      seg_end_host[nseg] (host, ascending, last == n) and seg_lr_host[nseg] (host): learning rate of each contiguous
 *      segment (parameter group); grad_scale multiplies the gradient first (1 / world size for a mean); step >= 1 is
 *      the 1-based step count t of the bias corrections.  All device buffers 16-byte aligned. ---- */
/* splat_frames_gauss_backward_dynamic_sets with the row described by SOURCES (nsrc <= SPLAT_MAX_SOURCES): the gradient of a shared
 * source (frame_stride 0) is the sum over the frames, ADDED to d_feature[P, cn]; a per-frame source (frame_stride != 0) gets
 * every frame's own gradient ADDED at d_feature + f * frame_stride (track_gs: the gradient then reaches the spline through
 * splat_dynamic_positions_batch_backward).  Channels no source with a d_feature covers (the depth channel) are skipped. */
int splat_frames_gauss_backward_dynamic_sources(int F, int P, int I, int C, int W, int H, int64_t capacity,
                                                const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                                const void *tab, const float *position, const float *cubic, int cubic_layout,
                                                const float *rotation, const float *rot_poly, const float *rot_fourier,
                                                const float *opacity, const float *scaling, const float *extr,
                                                float *d_position, float *d_cubic, float *d_rotation, float *d_opacity,
                                                float *d_scaling, int nsrc, const splat_feature_source_t *src,
                                                int depth_channel, float *tap, float *abs_tap, int32_t *radii_max,
                                                splat_stream_t stream);
/* the static-Gaussian counterpart (shared sources only: a per-frame source with a d_feature is SPLAT_E_ARG) */
int splat_frames_gauss_backward_static_sources_cam(int F, int P, int C, int W, int H, int64_t capacity,
                                                   const float *pair_records, const int32_t *goff_incl, const int32_t *radius,
                                                   const float *xyz, const float *scales, const float *uquats,
                                                   const splat_camera_t *cam, int accumulate, float *d_xyz, float *d_scales,
                                                   float *d_uquats, float *d_opacity, int nsrc,
                                                   const splat_feature_source_t *src, int depth_channel, float *tap,
                                                   float *abs_tap, int32_t *radii_max, splat_stream_t stream);

/* position(t_f) of every Gaussian (spline model, dynamic_gaussian_with_base_point_cloud.py:236-250) at the F frame times of
 * `tab` (the device table of splat_frame_preprocess_forward_batch): out[f * out_frame_stride + 3 n + j] -- the pair frames of a
 * training batch (track_gs = position(ids2), src/trainer_fragGS.py:486-507; the node sequence of its ARAP term, :671-675) in
 * one launch.  Backward: g (same indexing) is ADDED into d_position [P,3] and into the coefficient rows of every frame's
 * segment of d_cubic (either may be NULL); no atomics. */
int splat_dynamic_positions_batch_forward(int F, int P, int I, const void *tab, const float *position, const float *cubic,
                                          int cubic_layout, float *out, int64_t out_frame_stride, splat_stream_t stream);
int splat_dynamic_positions_batch_backward(int F, int P, int I, const void *tab, const float *g, int64_t g_frame_stride,
                                           int cubic_layout, float *d_position, float *d_cubic, splat_stream_t stream);

/* L1 image loss and its gradient in one pass (l1_loss of the reference's training step, src/trainer_fragGS.py:573-600):
 * grad[f, i] = scale * sign(pred[f * pred_frame_stride + i] - target[f, i]) for i < inner (pred may be a channel slice of a wider
 * image row), *loss_sum (zero-init, optional) += sum |pred - target| (float atomics: the sum is not bit-reproducible, the gradient is). */
int splat_l1_loss_grad(int F, int64_t inner, const float *pred, int64_t pred_frame_stride, const float *target, float scale,
                       float *grad, float *loss_sum, splat_stream_t stream);

#define SPLAT_ADAM_MAX_SEGMENTS 16
int splat_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int nseg,
                    const int64_t *seg_end_host, const float *seg_lr_host, float beta1, float beta2, float eps,
                    int step, float grad_scale, splat_stream_t stream);
/* ... with a learning-rate pattern inside segments: of every seg_period[k] consecutive elements of segment k (from its start) the
 * first seg_head[k] take seg_lr_head[k] instead of seg_lr[k] (period 0: none) -- two parameter groups interleaved in one tensor:
 * the reference's features (DC) / features_rest inside the [N, 16, 3] SH block, period 48, head 3
 * (src/configs/frag_gs_v10.yaml:44-47). */
int splat_adam_step_pattern(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int nseg,
                            const int64_t *seg_end_host, const float *seg_lr_host, const int32_t *seg_period_host,
                            const int32_t *seg_head_host, const float *seg_lr_head_host, float beta1, float beta2, float eps,
                            int step, float grad_scale, splat_stream_t stream);

/* ---- as-rigid-as-possible energy: replaces estimate_rotation + cal_arap_error of
 *      src/geometry_utils.py:50-123 (~50 eager launches per frame pair incl. torch.svd) by one launch.
 *      nodes [Nt,Nv,3]; nbr [Nv,K] neighbour ids (-1: none; slot k = the reference's `nn`); weight [Nv,K] or NULL (1 per
 *      edge); sample_idx [S] vertex ids.  energy (1 float, zero-filled by the caller) += sum over sampled vertices, frames
 *      t >= 1 and edges of w |e_tgt - R e_src|^2 (NOT yet divided by Nt); d_nodes (optional [Nt,Nv,3], zero-filled) +=
 *      d energy / d nodes with R held constant (the reference estimates it under no_grad); rotations (optional
 *      [Nt-1,S,3,3]) = the rotation of every (frame, sample).
 *      Index preconditions (device arrays, not validated on the host): 0 <= sample_idx < Nv; nbr in [0, Nv) or -1.  The
 *      kernels stay memory-safe outside them: a sample outside the set contributes nothing, a neighbour id outside it counts
 *      as "no edge". ---- */
int splat_arap_energy(int Nt, int Nv, int K, int S, const float *nodes, const int32_t *nbr, const float *weight,
                      const int64_t *sample_idx, float *energy, float *d_nodes, float *rotations, splat_stream_t stream);

/* B node sequences in one launch (the (ids1, ids2) pairs of a training batch, src/trainer_fragGS.py:671-675): sequence b =
 * nodes + b * node_batch_stride ([Nt, Nv, 3]); sample_idx [B, S]; neighbour rows BY SAMPLE nbr [B, S, K] (vertex ids, -1 = no
 * edge), weight likewise or NULL (1 per edge); energy [B] and d_nodes (strides of nodes; NULL: none) zero-init, ADDED to. */
int splat_arap_energy_batch(int B, int Nt, int Nv, int K, int S, const float *nodes, int64_t node_batch_stride,
                            const int32_t *nbr, const float *weight, const int64_t *sample_idx, float *energy, float *d_nodes,
                            float grad_scale /* d_nodes += grad_scale * gradient; the energy is not scaled */,
                            splat_stream_t stream);
/* K <= 8 nearest points of S query VERTICES (query_idx [B, S]: indices into the set itself) among the N points of each of B
 * point sets (set b at points + b * points_batch_stride), brute force in two launches: dists / idx [B, S, K] ascending, ties ->
 * smaller index, the query itself included -- knn_points(points, points)[sample] for the 512 sampled vertices of the ARAP term
 * (src/geometry_utils.py:17-19,98-101) without a grid build per point set.  Four launches: tile boxes, a distance bound per query
 * from its index neighbourhood, the scan (a wave skips the tiles whose box lies beyond all of its queries' bounds: hand the
 * queries in ascending index order when the points are spatially ordered), the merge.  Exact whatever the order.
 * Precondition (device array, not validated on the host): 0 <= query_idx < N; an id outside is clamped into the set
 * (memory-safe, the row is then that of vertex 0 / N - 1).  scratch: splat_knn_brute_scratch_bytes(B, N, S). */
size_t splat_knn_brute_scratch_bytes(int B, int N, int S);
int splat_knn_brute_batch(int B, int N, int S, int K, const float *points, int64_t points_batch_stride,
                          const int64_t *query_idx, float *dists, int32_t *idx, void *scratch, splat_stream_t stream);

/* ---- measurement hooks (bench.py: live per-kernel timing with HIP events on the launch stream) ---- */
void splat_profile_enable(int on);
void splat_profile_reset(void);
/* sums elapsed ms / launches of every kernel whose name starts with `prefix` (blocks on the events). */
int splat_profile_read(const char *prefix, double *total_ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* SPLAT_HIP_H */
