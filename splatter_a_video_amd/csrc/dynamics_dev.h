// Device helpers of the per-frame dynamic-Gaussian evaluation (shared by dynamics.hip and preprocess.hip).
// One quad (4 adjacent lanes) per Gaussian: lane j owns position / scaling component j (j < 3) and quaternion
// component j; the [N,4,4] / [N,8,4] rotation tables are read as one float4 row per lane and transposed-reduced
// inside the quad with DPP quad_perm.
#pragma once
#include "common.h"

struct DynBasis {
    float poly[4];     // t^k, k = 0..3
    float fourier[8];  // cos(t k pi) k=1..4, then sin(t k pi) k=1..4
};

// where coefficient k of Gaussian n for the active segment lives: cubic[seg_off + n*stride_n + k*stride_k + axis]
struct CubicAddr {
    size_t seg_off, stride_n, stride_k;
};

constexpr int DYN_BLOCK = 256;  // 64 Gaussians per workgroup

__device__ __forceinline__ float quad_xor1(float v) { return dpp_f<0xB1, 0xf, 0xf, true>(v); }  // quad_perm:[1,0,3,2]
__device__ __forceinline__ float quad_xor2(float v) { return dpp_f<0x4E, 0xf, 0xf, true>(v); }  // quad_perm:[2,3,0,1]
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    return v;
}

// lane j holds a float4 partial; returns in lane a the sum over the quad of component a
__device__ __forceinline__ float quad_transpose_sum(float4 v, int j) {
    const bool odd = j & 1, hi = j & 2;
    // pairs (x,y) and (z,w): keep the component whose parity matches the lane, send the other to the xor-1 partner
    const float keep_xy = odd ? v.y : v.x, send_xy = odd ? v.x : v.y;
    const float keep_zw = odd ? v.w : v.z, send_zw = odd ? v.z : v.w;
    const float xy = keep_xy + quad_xor1(send_xy);  // lanes 0,2: x over {j,j^1};  lanes 1,3: y
    const float zw = keep_zw + quad_xor1(send_zw);  // lanes 0,2: z;               lanes 1,3: w
    const float keep = hi ? zw : xy, send = hi ? xy : zw;
    return keep + quad_xor2(send);
}

// un-normalised quaternion component of this lane: rotation + detached polynomial + Fourier sums (:184-198)
__device__ __forceinline__ float quat_component(int n, int j, const DynBasis &b, const float *rotation,
                                                const float4 *rot_poly, const float4 *rot_fourier) {
    const float4 rp = rot_poly[(size_t)n * 4 + j];
    const float4 f0 = rot_fourier[(size_t)n * 8 + j];
    const float4 f1 = rot_fourier[(size_t)n * 8 + 4 + j];
    const float wp = b.poly[j], w0 = b.fourier[j], w1 = b.fourier[4 + j];
    float4 part;
    part.x = rp.x * wp + f0.x * w0 + f1.x * w1;
    part.y = rp.y * wp + f0.y * w0 + f1.y * w1;
    part.z = rp.z * wp + f0.z * w0 + f1.z * w1;
    part.w = rp.w * wp + f0.w * w0 + f1.w * w1;
    return rotation[(size_t)n * 4 + j] + quad_transpose_sum(part, j);
}


// the same with the lane's table rows already in registers (frame loops: the frozen tables are read once per Gaussian)
struct QuatRows {
    float rot;          // rotation[n][j]
    float4 rp, f0, f1;  // rot_poly[n][j][:], rot_fourier[n][j][:], rot_fourier[n][4 + j][:]
};
__device__ __forceinline__ QuatRows load_quat_rows(int n, int j, const float *rotation, const float4 *rot_poly,
                                                   const float4 *rot_fourier) {
    QuatRows r;
    r.rot = rotation[(size_t)n * 4 + j];
    r.rp = rot_poly[(size_t)n * 4 + j];
    r.f0 = rot_fourier[(size_t)n * 8 + j];
    r.f1 = rot_fourier[(size_t)n * 8 + 4 + j];
    return r;
}
// wp / w0 / w1: this lane's basis values t'^j, cos(t' (j+1) pi), sin(t' (j+1) pi) -- read per lane from the frame table
// (a register copy of the 12 values indexed by the lane's j would live in scratch memory)
__device__ __forceinline__ float quat_component_rows(const QuatRows &r, int j, float wp, float w0, float w1) {
    float4 part;
    part.x = r.rp.x * wp + r.f0.x * w0 + r.f1.x * w1;
    part.y = r.rp.y * wp + r.f0.y * w0 + r.f1.y * w1;
    part.z = r.rp.z * wp + r.f0.z * w0 + r.f1.z * w1;
    part.w = r.rp.w * wp + r.f0.w * w0 + r.f1.w * w1;
    return r.rot + quad_transpose_sum(part, j);
}

// ---- host side
// SPLAT_CUBIC_GAUSSIAN_MAJOR: the reference's [N,4,I,3]; SPLAT_CUBIC_SEGMENT_MAJOR: [I,N,4,3], one contiguous
// 48-byte record per Gaussian and frame
// per-frame scalars of a frame batch, one entry per frame in device memory (built on the host from the clip's knots:
// segment index, offset inside the segment, the 12 time-basis values)
struct DynTab {
    int seg;
    float d;
    float basis[12];
    float pad[2];  // 64-byte entries
};

__host__ __device__ static inline CubicAddr cubic_addr(int layout, int P, int I, int seg) {
    CubicAddr a;
    if (layout == SPLAT_CUBIC_SEGMENT_MAJOR) {
        a.seg_off = (size_t)seg * (size_t)P * 12;
        a.stride_n = 12;
        a.stride_k = 3;
    } else {
        a.seg_off = (size_t)seg * 3;
        a.stride_n = (size_t)4 * I * 3;
        a.stride_k = (size_t)I * 3;
    }
    return a;
}

static inline DynBasis load_basis(const float *host12) {
    DynBasis b;
    memcpy(b.poly, host12, sizeof(float) * 4);
    memcpy(b.fourier, host12 + 4, sizeof(float) * 8);
    return b;
}


// value of lane `src` of the quad in all four lanes
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
    return dpp_f<SRC | (SRC << 2) | (SRC << 4) | (SRC << 6), 0xf, 0xf, true>(v);
}
