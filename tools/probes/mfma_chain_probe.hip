// Probe: is v_mfma_f32_16x16x4_f32 bit-identical to a sequential fmaf chain over k (starting from C)?
// Decides whether the forward compositing kernel (VALU) can reproduce the backward's matrix-core power bit for bit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *A, const float *B, const float *C, float *D, int nmat) {
    const int lane = threadIdx.x & 63;
    const int mat = blockIdx.x;
    const float *a = A + mat * 64, *b = B + mat * 64, *c = C + mat * 256;
    const int n = lane & 15, kk = lane >> 4;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = c[(4 * kk + i) * 16 + n];
    // A[m][k]: lane (m = lane&15, k = lane>>4); B[k][n]: lane (n = lane&15, k = lane>>4)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(lane & 15) * 4 + kk], b[kk * 16 + n], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[mat * 256 + (4 * kk + i) * 16 + n] = acc[i];
}

int main() {
    const int nmat = 4096;
    float *hA = (float *)malloc(nmat * 64 * 4), *hB = (float *)malloc(nmat * 64 * 4), *hC = (float *)malloc(nmat * 256 * 4),
          *hD = (float *)malloc(nmat * 256 * 4);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int i = 0; i < nmat * 64; ++i) { hA[i] = rnd() * powf(2.f, (rand() % 12) - 6); hB[i] = rnd() * powf(2.f, (rand() % 12) - 6); }
    for (int i = 0; i < nmat * 256; ++i) hC[i] = (rand() % 4 == 0) ? 0.f : rnd() * powf(2.f, (rand() % 12) - 6);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, nmat * 64 * 4); hipMalloc(&dB, nmat * 64 * 4); hipMalloc(&dC, nmat * 256 * 4); hipMalloc(&dD, nmat * 256 * 4);
    hipMemcpy(dA, hA, nmat * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, nmat * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, nmat * 256 * 4, hipMemcpyHostToDevice);
    probe<<<nmat, 64>>>(dA, dB, dC, dD, nmat);
    hipMemcpy(hD, dD, nmat * 256 * 4, hipMemcpyDeviceToHost);
    long tot = 0, m_asc = 0, m_desc = 0, m_dbl = 0, m_pair = 0, m_asc_end = 0;
    for (int mat = 0; mat < nmat; ++mat)
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                const float *a = hA + mat * 64 + m * 4;
                float b[4];
                for (int k = 0; k < 4; ++k) b[k] = hB[mat * 64 + k * 16 + n];
                const float c = hC[mat * 256 + m * 16 + n], d = hD[mat * 256 + m * 16 + n];
                float x = c;
                for (int k = 0; k < 4; ++k) x = fmaf(a[k], b[k], x);
                float y = c;
                for (int k = 3; k >= 0; --k) y = fmaf(a[k], b[k], y);
                double s = (double)c;
                for (int k = 0; k < 4; ++k) s += (double)a[k] * (double)b[k];
                const float z = (float)s;
                const float p = c + (fmaf(a[0], b[0], a[1] * b[1]) + fmaf(a[2], b[2], a[3] * b[3]));
                float q = 0.f;   // products chained first, C added last
                for (int k = 0; k < 4; ++k) q = fmaf(a[k], b[k], q);
                q += c;
                ++tot;
                m_asc += memcmp(&x, &d, 4) == 0; m_desc += memcmp(&y, &d, 4) == 0; m_dbl += memcmp(&z, &d, 4) == 0;
                m_pair += memcmp(&p, &d, 4) == 0; m_asc_end += memcmp(&q, &d, 4) == 0;
            }
    printf("{\"total\": %ld, \"fma_chain_ascending\": %ld, \"fma_chain_descending\": %ld, \"single_rounding_double\": %ld, \"pairwise\": %ld, \"chain_then_c\": %ld}\n",
           tot, m_asc, m_desc, m_dbl, m_pair, m_asc_end);
    return 0;
}
