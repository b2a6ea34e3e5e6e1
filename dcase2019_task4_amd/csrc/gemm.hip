// gemm.hip - small strided fp32 GEMM on the f32 MFMA, used for the GRU input projections
// (x @ W_ih^T + b_ih, torch.nn.GRU inside baseline/models/RNN.py:12) and for the GRU weight /
// input gradients.  C[m][n] = sum_k A(m,k) * B(k,n) (+ bias[n]) (+ C), arbitrary element strides
// so the same kernel serves NN / NT / TN products.  Sizes here are tiny (M <= B*T/8, N <= 192,
// K <= B*T/8), so the kernel is a plain 64x64x16 LDS-tiled loop, 4 waves x one 32x32 tile each.
#include "common.h"
#include "kernels.h"

#define GT_M 64
#define GT_N 64
#define GT_K 16

__global__ __launch_bounds__(256) void k_gemm(GemmDesc d) {
    __shared__ float As[GT_M * (GT_K + 1)];
    __shared__ float Bs[GT_K * (GT_N + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * GT_M, n0 = blockIdx.x * GT_N;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < d.K; k0 += GT_K) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + 256 * it;
            int m, k;
            if (d.sAk == 1) { m = e >> 4; k = e & 15; } else { m = e & 63; k = e >> 6; }
            float v = 0.f;
            if (m0 + m < d.M && k0 + k < d.K) v = d.A[(int64_t)(m0 + m) * d.sAm + (int64_t)(k0 + k) * d.sAk];
            As[m * (GT_K + 1) + k] = v;
            int kb, nb;
            if (d.sBn == 1) { kb = e >> 6; nb = e & 63; } else { kb = e & 15; nb = e >> 4; }
            float w = 0.f;
            if (k0 + kb < d.K && n0 + nb < d.N) w = d.B[(int64_t)(k0 + kb) * d.sBk + (int64_t)(n0 + nb) * d.sBn];
            Bs[kb * (GT_N + 1) + nb] = w;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GT_K / 2; ++s) {
            const float a = As[(32 * wm + n) * (GT_K + 1) + 2 * s + kh];
            const float b = Bs[(2 * s + kh) * (GT_N + 1) + 32 * wn + n];
            acc = mfma32(a, b, acc);
        }
        __syncthreads();
    }
    const int col = n0 + 32 * wn + n;
    if (col < d.N) {
        const float bv = d.bias ? d.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wm + mfma32_row(r, lane);
            if (row < d.M) {
                float* c = d.C + (int64_t)row * d.ldc + col;
                float v = acc[r] + bv;
                if (d.accumulate) v += *c;
                *c = v;
            }
        }
    }
}

// out[n] = sum_m A[m*lda + n]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ A, int M, int N, int64_t lda, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float s = 0.f;
    if (col < N)
        for (int m = r; m < M; m += 4) s += A[(int64_t)m * lda + col];
    red[r][c] = s;
    __syncthreads();
    if (r == 0 && col < N) out[col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

int launch_gemm(const GemmDesc& d, hipStream_t st) {
    dim3 grid((d.N + GT_N - 1) / GT_N, (d.M + GT_M - 1) / GT_M);
    k_gemm<<<grid, 256, 0, st>>>(d);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_colsum(const float* A, int M, int N, int64_t lda, float* out, hipStream_t st) {
    k_colsum<<<(N + 63) / 64, 256, 0, st>>>(A, M, N, lda, out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
