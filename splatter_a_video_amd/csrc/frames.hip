// One call per direction for a whole frame batch (SURVEY 7 stage 6): the sequences FrameBatch (frames.py) composes from the
// *_batch entry points, for callers that want a single crossing of the C ABI per gradient step -- static Gaussians +
// per-frame offsets under the orthographic camera, one feature set.  Every buffer is caller-owned (splat_frames_t holds
// the pointers); nothing is allocated, nothing synchronises with the host.
#include "common.h"

static splat_camera_t batch_camera(const splat_frames_t *b) {
    splat_camera_t cam;
    memset(&cam, 0, sizeof(cam));
    cam.perspective = b->perspective; cam.intr = b->intr; cam.intr_frame_stride = b->intr_frame_stride;
    cam.extr = b->extr; cam.extr_frame_stride = b->extr_frame_stride; cam.offsets = b->offsets;
    return cam;
}

// pairs per tile and frame; with b->reach only the pairs whose tile the splat can reach (binning.hip: "reach masks")
static int frames_bin_count(const splat_frames_t *b) {
    if (b->reach) {
        SPLAT_CHECK_ARG(b->opacity != nullptr, "reach masks need the opacity");
        return splat_bin_count_batch_reach(b->F, b->P, b->uv, b->radius, b->conic, b->opacity, 0, b->W, b->H, b->bin_scratch,
                                           b->tile_range, b->pairs, nullptr, b->reach, b->stream);
    }
    return splat_bin_count_batch(b->F, b->P, b->uv, b->radius, b->W, b->H, b->bin_scratch, b->tile_range, b->pairs, b->stream);
}

extern "C" int splat_frames_forward(const splat_frames_t *b) {
    SPLAT_CHECK_ARG(b != nullptr && b->struct_bytes == sizeof(splat_frames_t), "splat_frames_t of another ABI version");
    SPLAT_CHECK_ARG(b->capacity >= 1, "capacity (pairs reserved per frame) must be set: run splat_frames_count first");
    const splat_camera_t cam = batch_camera(b);
    int rc = splat_preprocess_forward_batch_cam(b->F, b->P, b->xyz, b->offsets, b->scales, b->uquats, &cam, b->W, b->H,
                                                b->nearest, b->extent, b->uv, b->depth, b->conic, b->radius, b->stream);
    if (rc != SPLAT_OK) return rc;
    rc = frames_bin_count(b);
    if (rc != SPLAT_OK) return rc;
    if (b->reach)
        rc = splat_bin_sort_batch_reach(b->F, b->P, b->uv, b->depth, b->radius, b->reach, b->W, b->H, b->bin_scratch,
                                        b->tile_range, b->capacity, b->keys, b->idx_sorted, b->overflow, b->goff_incl, b->owner,
                                        b->slot_sorted, b->stream);
    else
        rc = splat_bin_sort_batch(b->F, b->P, b->uv, b->depth, b->radius, b->W, b->H, b->bin_scratch, b->tile_range, b->capacity,
                                  b->keys, b->idx_sorted, b->overflow, b->goff_incl, b->owner, b->slot_sorted, b->stream);
    if (rc != SPLAT_OK) return rc;
    return splat_alpha_blending_forward_batch(b->F, b->P, b->C, b->uv, b->conic, b->opacity, 0, b->feature, 0, b->idx_sorted,
                                              b->tile_range, b->capacity, b->bg, nullptr, b->W, b->H, 0, 0, b->out, b->final_T,
                                              b->ncontrib, nullptr, b->pack, b->cull_flags, b->stream);
}

// geometry + pair counts only: pairs[F] (device) tells the caller how much capacity to reserve before the first
// splat_frames_forward
extern "C" int splat_frames_count(const splat_frames_t *b) {
    SPLAT_CHECK_ARG(b != nullptr && b->struct_bytes == sizeof(splat_frames_t), "splat_frames_t of another ABI version");
    const splat_camera_t cam = batch_camera(b);
    int rc = splat_preprocess_forward_batch_cam(b->F, b->P, b->xyz, b->offsets, b->scales, b->uquats, &cam, b->W, b->H,
                                                b->nearest, b->extent, b->uv, b->depth, b->conic, b->radius, b->stream);
    if (rc != SPLAT_OK) return rc;
    return frames_bin_count(b);
}

extern "C" int splat_frames_backward(const splat_frames_t *b) {
    SPLAT_CHECK_ARG(b != nullptr && b->struct_bytes == sizeof(splat_frames_t), "splat_frames_t of another ABI version");
    SPLAT_CHECK_ARG(b->dL_dout && b->pair_records, "null pointer");
    // everything the second launch sequence validates is checked BEFORE the tile kernels run: a bad struct must not leave
    // half a backward behind (pair records overwritten, no gradients)
    SPLAT_CHECK_ARG(b->d_opacity && b->d_feature, "null gradient pointer");
    SPLAT_CHECK_ARG(b->d_xyz && b->d_scales && b->d_uquats, "null gradient pointer");
    SPLAT_CHECK_ARG(b->xyz && b->scales && b->uquats && b->extr && b->goff_incl && b->radius, "null pointer");
    SPLAT_CHECK_ARG(!b->perspective || b->intr, "the pinhole camera needs intr");
    SPLAT_CHECK_ARG(b->F == 1 || b->offsets || b->extr_frame_stride != 0, "several frames need per-frame offsets or cameras");
    int rc = splat_alpha_blending_backward_batch(b->F, b->P, b->C, b->idx_sorted, b->tile_range, b->capacity, b->bg, b->W, b->H,
                                                 b->final_T, b->ncontrib, b->dL_dout, b->want_abs, b->slot_sorted,
                                                 b->pair_records, b->pack, b->cull_flags, b->dbg_T_front, b->stream);
    if (rc != SPLAT_OK) return rc;
    const splat_camera_t cam = batch_camera(b);
    return splat_frames_gauss_backward_static_cam(b->F, b->P, b->C, b->W, b->H, b->capacity, b->want_abs, b->pair_records,
                                                  b->goff_incl, b->radius, b->xyz, b->scales, b->uquats, &cam, b->accumulate,
                                                  b->d_xyz, b->d_scales, b->d_uquats, b->d_opacity, b->d_feature, b->C, 0, -1,
                                                  b->tap, b->abs_tap, b->radii_max, b->stream);
}
