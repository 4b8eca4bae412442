// As-rigid-as-possible energy of a node sequence (second half): replaces the eager torch of
// src/geometry_utils.py:50-123 (estimate_rotation: per-vertex covariance of source / target edge matrices, torch.svd,
// rotation W U^T with the reflection fix; cal_arap_error: sum_k w_k |e_tgt_k - R e_src_k|^2 over the sampled vertices and
// the frames t >= 1) -- ~50 small launches per frame pair -- by ONE launch: a thread per (sampled vertex, frame) builds
// the 3x3 covariance, extracts the rotation (Jacobi eigen-decomposition of S^T S -> SVD -> Kabsch rotation with
// det = +1), accumulates the energy and scatters d energy / d nodes (the rotation is a constant of the gradient: the
// reference estimates it under torch.no_grad()).
#include "common.h"

namespace {
constexpr int ARAP_MAXK = 16;

__device__ __forceinline__ void jacobi_rot(float A[3][3], float V[3][3], int p, int q) {
    if (fabsf(A[p][q]) < 1e-30f) return;
    const float theta = (A[q][q] - A[p][p]) / (2.f * A[p][q]);
    const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
    const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // A <- A J (columns p, q)
        const float akp = A[k][p], akq = A[k][q];
        A[k][p] = c * akp - s * akq;
        A[k][q] = s * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // A <- J^T A (rows p, q)
        const float apk = A[p][k], aqk = A[q][k];
        A[p][k] = c * apk - s * aqk;
        A[q][k] = s * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}

// R = W diag(1, 1, d) U^T for S = U Sigma W^T with d = sign(det(W U^T)) applied to the smallest singular value: the rotation
// estimate_rotation returns (geometry_utils.py:71-84).  S = 0 gives the identity (torch.svd of the zero matrix: U = W = I).
__device__ void kabsch_rotation(const float S[3][3], float R[3][3]) {
    float A[3][3], V[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = S[0][i] * S[0][j] + S[1][i] * S[1][j] + S[2][i] * S[2][j];  // S^T S
    for (int sweep = 0; sweep < 6; ++sweep) {
        jacobi_rot(A, V, 0, 1);
        jacobi_rot(A, V, 0, 2);
        jacobi_rot(A, V, 1, 2);
    }
    // order the eigenpairs by descending eigenvalue (columns of V = right singular vectors W)
    float lam[3] = {A[0][0], A[1][1], A[2][2]};
    int o[3] = {0, 1, 2};
    if (lam[o[0]] < lam[o[1]]) { const int t = o[0]; o[0] = o[1]; o[1] = t; }
    if (lam[o[0]] < lam[o[2]]) { const int t = o[0]; o[0] = o[2]; o[2] = t; }
    if (lam[o[1]] < lam[o[2]]) { const int t = o[1]; o[1] = o[2]; o[2] = t; }
    float W[3][3], U[3][3], sig[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int r = 0; r < 3; ++r) W[r][c] = V[r][o[c]];
        sig[c] = sqrtf(fmaxf(lam[o[c]], 0.f));
    }
    // left singular vectors u_c = S w_c / sigma_c; degenerate directions are completed to an orthonormal basis
    const float tiny = 1e-12f * fmaxf(sig[0], 1e-30f) + 1e-30f;
    float u0[3] = {1.f, 0.f, 0.f}, u1[3] = {0.f, 1.f, 0.f}, u2[3];
    if (sig[0] > tiny) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u0[r] = (S[r][0] * W[0][0] + S[r][1] * W[1][0] + S[r][2] * W[2][0]) / sig[0];
        float n = rsqrtf(fmaxf(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2], 1e-30f));
        u0[0] *= n; u0[1] *= n; u0[2] *= n;
        if (sig[1] > tiny) {
#pragma unroll
            for (int r = 0; r < 3; ++r) u1[r] = (S[r][0] * W[0][1] + S[r][1] * W[1][1] + S[r][2] * W[2][1]) / sig[1];
        } else {  // any unit vector orthogonal to u0
            const int m = fabsf(u0[0]) <= fabsf(u0[1]) ? (fabsf(u0[0]) <= fabsf(u0[2]) ? 0 : 2) : (fabsf(u0[1]) <= fabsf(u0[2]) ? 1 : 2);
            u1[0] = m == 0 ? 1.f : 0.f; u1[1] = m == 1 ? 1.f : 0.f; u1[2] = m == 2 ? 1.f : 0.f;
        }
        const float d = u1[0] * u0[0] + u1[1] * u0[1] + u1[2] * u0[2];
        u1[0] -= d * u0[0]; u1[1] -= d * u0[1]; u1[2] -= d * u0[2];
        n = rsqrtf(fmaxf(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2], 1e-30f));
        u1[0] *= n; u1[1] *= n; u1[2] *= n;
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    // det(U) = +1 by construction; the sign d of the third term makes det(R) = det(W) * d = +1
    const float detW = W[0][0] * (W[1][1] * W[2][2] - W[1][2] * W[2][1]) - W[0][1] * (W[1][0] * W[2][2] - W[1][2] * W[2][0]) +
                       W[0][2] * (W[1][0] * W[2][1] - W[1][1] * W[2][0]);
    const float d3 = detW < 0.f ? -1.f : 1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        U[r][0] = u0[r]; U[r][1] = u1[r]; U[r][2] = u2[r];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r][c] = W[r][0] * U[c][0] + W[r][1] * U[c][1] + d3 * W[r][2] * U[c][2];
}

// Batch of node sequences (grid.y = sequence b; the pairs (ids1, ids2) of a training batch, src/trainer_fragGS.py:671-675):
// sequence b's nodes / gradients at + b * node_bs floats, its sampled vertices at sample_idx + b * sample_bs, its energy at
// energy[b].  nbr_compact: the neighbour table (and `weight`) is [S, K] per sequence (row = SAMPLE s, + b * S * K) instead of
// [Nv, K] (row = vertex) -- only the sampled vertices' neighbours are ever read.
struct ArapBatch {
    long long node_bs, sample_bs;
    int nbr_compact;
    float grad_scale;   // d_nodes += grad_scale * d energy / d nodes (the loss weight of the term; the energy itself is not scaled)
};

__global__ void __launch_bounds__(128)
arap_kernel(int Nt, int Nv, int K, int S_, const float *__restrict__ nodes, const int *__restrict__ nbr,
            const float *__restrict__ weight, const long long *__restrict__ sample_idx, float *__restrict__ energy,
            float *__restrict__ d_nodes, float *__restrict__ rot_out, const ArapBatch bt) {
    const int g = blockIdx.x * 128 + threadIdx.x;
    if (g >= S_ * (Nt - 1)) return;
    const int s = g % S_, t = 1 + g / S_;
    const int b = blockIdx.y;
    nodes += (size_t)b * bt.node_bs;
    if (d_nodes) d_nodes += (size_t)b * bt.node_bs;
    sample_idx += (size_t)b * bt.sample_bs;
    energy += b;
    if (rot_out) rot_out += (size_t)b * (size_t)S_ * (Nt - 1) * 9;
    const int i = (int)sample_idx[s];
    if ((unsigned)i >= (unsigned)Nv) return;   // memory safety: a sample outside the vertex set contributes nothing
    if (bt.nbr_compact) {   // rows by sample: shift the tables so that row i below is sample s of sequence b
        const long long off = ((long long)b * S_ + s - i) * K;
        nbr += off;
        if (weight) weight += off;
    }
    const float *src = nodes, *tgt = nodes + (size_t)t * Nv * 3;
    const float pi0[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]}, pit[3] = {tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2]};
    float es[ARAP_MAXK][3], et[ARAP_MAXK][3], w[ARAP_MAXK];
    float S[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    // the reference's shortcut (:66-67): torch.where((source_edge == target_edge).all(dim=1))[0] -- the comparison is reduced
    // over the K edges only, so a vertex counts as undeformed as soon as ONE coordinate axis of all its edges is unchanged
    // (planar or axis-constant motion), not only when all K x 3 entries are
    bool same_axis[3] = {true, true, true};
    for (int k = 0; k < K; ++k) {
        int j = nbr[(size_t)i * K + k];
        j = (unsigned)j < (unsigned)Nv ? j : -1;   // memory safety: an id outside the vertex set is "no edge", like -1
        w[k] = weight ? weight[(size_t)i * K + k] : (j >= 0 ? 1.f : 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            es[k][c] = j >= 0 ? pi0[c] - src[3 * j + c] : 0.f;
            et[k][c] = j >= 0 ? pit[c] - tgt[3 * j + c] : 0.f;
            same_axis[c] = same_axis[c] && (es[k][c] == et[k][c]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) S[r][c] += es[k][r] * (w[k] * et[k][c]);  // source^T D target (:64)
    }
    if (same_axis[0] || same_axis[1] || same_axis[2]) {  // "undeformed" vertex: S = 0 so that R = I (:66-67)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) S[r][c] = 0.f;
    }
    float R[3][3];
    kabsch_rotation(S, R);
    if (rot_out) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) rot_out[(size_t)g * 9 + 3 * r + c] = R[r][c];
    }
    float e = 0.f, gi0[3] = {0.f, 0.f, 0.f}, git[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        int j = nbr[(size_t)i * K + k];
        j = (unsigned)j < (unsigned)Nv ? j : -1;   // memory safety: an id outside the vertex set is "no edge", like -1
        float st[3];  // stretch vector e_tgt - R e_src (:113-115)
#pragma unroll
        for (int r = 0; r < 3; ++r) st[r] = et[k][r] - (R[r][0] * es[k][0] + R[r][1] * es[k][1] + R[r][2] * es[k][2]);
        e += w[k] * (st[0] * st[0] + st[1] * st[1] + st[2] * st[2]);
        if (d_nodes && j >= 0) {
            float gt[3], gs[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gt[c] = bt.grad_scale * (2.f * w[k] * st[c]);
                gs[c] = bt.grad_scale * (-2.f * w[k] * (R[0][c] * st[0] + R[1][c] * st[1] + R[2][c] * st[2]));
                git[c] += gt[c]; gi0[c] += gs[c];
                atomic_add_f32(d_nodes + (size_t)t * Nv * 3 + 3 * j + c, -gt[c]);
                atomic_add_f32(d_nodes + 3 * j + c, -gs[c]);
            }
        }
    }
    if (d_nodes) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            atomic_add_f32(d_nodes + (size_t)t * Nv * 3 + 3 * i + c, git[c]);
            atomic_add_f32(d_nodes + 3 * i + c, gi0[c]);
        }
    }
    atomic_add_f32(energy, e);
}
}  // namespace

extern "C" int splat_arap_energy(int Nt, int Nv, int K, int S, const float *nodes, const int32_t *nbr, const float *weight,
                                 const int64_t *sample_idx, float *energy, float *d_nodes, float *rotations, void *stream) {
    SPLAT_CHECK_ARG(Nt >= 1 && Nv >= 1 && K >= 1 && K <= ARAP_MAXK && S >= 0, "bad sizes (K <= 16)");
    SPLAT_CHECK_ARG(nodes && nbr && energy && (S == 0 || sample_idx), "null pointer");
    if (S == 0 || Nt < 2) return SPLAT_OK;
    const int total = S * (Nt - 1);
    ArapBatch bt;
    memset(&bt, 0, sizeof(bt));
    bt.grad_scale = 1.f;
    SPLAT_LAUNCH("arap_energy", arap_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, Nt, Nv, K, S,
                 nodes, nbr, weight, (const long long *)sample_idx, energy, d_nodes, rotations, bt);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}

// B node sequences in one launch: sequence b = nodes + b * node_batch_stride ([Nt, Nv, 3]), sampled vertices sample_idx + b * S,
// neighbour rows BY SAMPLE nbr[b, s, 0..K) (vertex ids; -1: no edge; what a K-nearest-neighbour query of the sampled vertices
// alone returns), weight likewise or NULL (1 per edge); energy[b] += that sequence's energy (zero-init), d_nodes (zero-init,
// same strides as nodes) += grad_scale * its gradient (the term's loss weight; the energy is not scaled).  The (ids1, ids2) pairs of a training batch, src/trainer_fragGS.py:671-675.
extern "C" int splat_arap_energy_batch(int B, int Nt, int Nv, int K, int S, const float *nodes, int64_t node_batch_stride,
                                       const int32_t *nbr, const float *weight, const int64_t *sample_idx, float *energy,
                                       float *d_nodes, float grad_scale, void *stream) {
    SPLAT_CHECK_ARG(B >= 1 && B <= 65535 && Nt >= 1 && Nv >= 1 && K >= 1 && K <= ARAP_MAXK && S >= 0, "bad sizes (K <= 16)");
    SPLAT_CHECK_ARG(nodes && nbr && energy && (S == 0 || sample_idx), "null pointer");
    SPLAT_CHECK_ARG(node_batch_stride >= (int64_t)Nt * Nv * 3, "node_batch_stride below Nt * Nv * 3");
    if (S == 0 || Nt < 2) return SPLAT_OK;
    const int total = S * (Nt - 1);
    ArapBatch bt;
    bt.node_bs = node_batch_stride; bt.sample_bs = S; bt.nbr_compact = 1; bt.grad_scale = grad_scale;
    SPLAT_LAUNCH("arap_energy", arap_kernel, dim3((unsigned)((total + 127) / 128), (unsigned)B), dim3(128), 0, (hipStream_t)stream,
                 Nt, Nv, K, S, nodes, nbr, weight, (const long long *)sample_idx, energy, d_nodes, (float *)nullptr, bt);
    SPLAT_POST_LAUNCH();
    return SPLAT_OK;
}
