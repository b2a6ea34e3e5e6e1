// ggemm.hip - C[m][n] = sum_k A[m][k] B[n][k] (+ bias[n]): the "NT" GEMMs of the wide BiGRU - the input projections
// gi = x W_ih^T + b_ih (baseline/models/RNN.py:12: nn.GRU's first half) and the gradient w.r.t. the layer input
// dX = [dgi_f | dgi_r] [W_ih_f ; W_ih_r] (with the stacked W_ih transposed once by k_gnt_pack_t) - exact fp32 on the f32 MFMA.
// gemm.hip's 64 x 64-tile batched kernel served the 192 x 64 .. 128 shapes of the 64-cell GRU; at H = 256 its
// K-loop (one 64-deep tile per global -> LDS -> MFMA round trip, no overlap) ran at 27 TFLOP/s and the four GEMMs on the
// critical path of a wide step cost 380 us.  Here: the same 64 x 64 output tile per workgroup (the shapes are too small for
// more: 1 872 x 512 x 1 536 is 240 such tiles, 60 at 128 x 128 - measured 123 us on a quarter of the chip), but both
// operands stream through DOUBLE-BUFFERED LDS chunks of 32 k - the next chunk's global loads are in flight during the
// current chunk's MFMAs - and 34 KB of LDS per workgroup leaves room for 4 workgroups per CU.
#include "common.h"
#include "kernels.h"
#include "gkernels.h"
#include "gpack.h"
#include "gen.h"

#define GNT_KC 32
#define GNT_S (GNT_KC + 1)
#define GNT_T 64          // output tile: 64 x 64 per workgroup, one 32 x 32 accumulator per wave

// Which output tile a workgroup computes.  In launch order (n tile fastest) the workgroups that share an A row panel - all n
// tiles of one m tile - are consecutive, i.e. dealt round-robin over the eight XCDs, and every L2 fetched every panel (35 MB
// of HBM reads per launch, six launches per wide step).  The linear workgroup number is re-read XCD-major: XCD x (= number % 8)
// works through the x-th eighth of the (problem, m tile, n tile) order, so an A panel lives under ONE L2 and only the weight
// matrix is fetched by all eight.  Same tiles, same sums: results are bit-identical.
__device__ __forceinline__ void gnt_tile_of_workgroup(int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#ifndef SED_NO_XCD_ORDER
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
    if ((total & 7) == 0) {
        const int lin = bx + gx * (by + gy * bz), l2 = (lin & 7) * (total >> 3) + (lin >> 3);
        bx = l2 % gx; by = (l2 / gx) % gy; bz = l2 / (gx * gy);
    }
#endif
}

__global__ __launch_bounds__(256) void k_gnt_gemm(GntBatch gb) {
    __shared__ float As[2][GNT_T * GNT_S];
    __shared__ float Bs[2][GNT_T * GNT_S];
    int bx, by, bz;
    gnt_tile_of_workgroup(bx, by, bz);
    const GntProb& d = gb.p[bz];
    const int m0 = by * GNT_T, n0 = bx * GNT_T;
    if (m0 >= d.M || n0 >= d.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    // staging: 64 rows x 32 k = 512 float4 per operand = 2 per thread: row = u >> 3, k4 = u & 7
    f32x4 ra[2], rb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i, row = u >> 3, k4 = u & 7;
            const int am = min(m0 + row, d.M - 1), bn = min(n0 + row, d.N - 1);      // clamped rows are never stored
            ra[i] = *(const f32x4*)(d.A + (size_t)am * d.lda + k0 + 4 * k4);
            rb[i] = *(const f32x4*)(d.B + (size_t)bn * d.ldb + k0 + 4 * k4);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i, row = u >> 3, k4 = u & 7;
            float* a = &As[buf][row * GNT_S + 4 * k4];
            float* b = &Bs[buf][row * GNT_S + 4 * k4];
            a[0] = ra[i][0]; a[1] = ra[i][1]; a[2] = ra[i][2]; a[3] = ra[i][3];
            b[0] = rb[i][0]; b[1] = rb[i][1]; b[2] = rb[i][2]; b[3] = rb[i][3];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nch = d.K / GNT_KC;
    load(0);
    store(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) load((ch + 1) * GNT_KC);
        const float* ap = &As[ch & 1][(32 * wm + n) * GNT_S + kh];
        const float* bp = &Bs[ch & 1][(32 * wn + n) * GNT_S + kh];
#pragma unroll
        for (int ks = 0; ks < GNT_KC / 2; ++ks) acc = mfma32(ap[2 * ks], bp[2 * ks], acc);
        if (ch + 1 < nch) store((ch + 1) & 1);
        __syncthreads();
    }
    const int col = n0 + 32 * wn + n;
    if (col < d.N) {
        const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wm + mfma32_row(r, lane);
            if (row < d.M) d.C[(size_t)row * d.ldc + col] = acc[r] + bias;
        }
    }
}

// The same GEMMs with bf16 MFMA operands (SED_DTYPE_BF16): fp32 in HBM, rounded to bf16 (RNE) on the way into LDS, fp32
// accumulation and output.  64 x 64 output tile, 64 k per double-buffered chunk, one 32 x 32 x 16 accumulator per wave: the
// shapes (M = 1 872, N <= 768, K <= 1 536) make this a staging-bound kernel - 4 MFMAs per chunk against 32 KB of operand
// loads - so the tile is kept small to fill the chip (60 - 720 workgroups) rather than large to feed the MFMA.
#define GNB_KC 64
#define GNB_S (GNB_KC + 8)                 // row stride in bf16 elements: 144 B, an odd multiple of 16 B (conflict-free b128)
// X3 (SED_DTYPE_BF16X3): both operands split hi + lo into two LDS planes, every k-step is hi hi + hi lo + lo hi (~2^-16 per
// product) - the projections of the wide BiGRU in the mode that holds 1e-3, in place of the exact-fp32 k_gnt_gemm above.
// X3 = 2 (SED_DTYPE_F16, forward projections): operands rounded to fp16, one v_mfma_f32_32x32x16_f16 per k-step.
template <int X3_>
__global__ __launch_bounds__(256) void k_gnt_gemm_bf16(GntBatch gb) {
    constexpr int X3 = X3_ == 1 ? 1 : 0, F16 = X3_ == 2 ? 1 : 0;
    constexpr int NPL = X3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][NPL][GNT_T * GNB_S];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][NPL][GNT_T * GNB_S];
    int bx, by, bz;
    gnt_tile_of_workgroup(bx, by, bz);
    const GntProb& d = gb.p[bz];
    const int m0 = by * GNT_T, n0 = bx * GNT_T;
    if (m0 >= d.M || n0 >= d.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    // staging: 64 rows x 64 k = 512 groups of 8 per operand = 2 per thread: row = u >> 3, k8 = u & 7
    f32x4 ra[2][2], rb[2][2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i, row = u >> 3, k8 = u & 7;
            const int am = min(m0 + row, d.M - 1), bn = min(n0 + row, d.N - 1);      // clamped rows are never stored
            const float* a = d.A + (size_t)am * d.lda + k0 + 8 * k8;
            const float* b = d.B + (size_t)bn * d.ldb + k0 + 8 * k8;
            ra[i][0] = *(const f32x4*)a; ra[i][1] = *(const f32x4*)(a + 4);
            rb[i][0] = *(const f32x4*)b; rb[i][1] = *(const f32x4*)(b + 4);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i, row = u >> 3, k8 = u & 7;
            bf16x8 a, b, al, bl;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float av = q < 4 ? ra[i][0][q] : ra[i][1][q - 4], bv = q < 4 ? rb[i][0][q] : rb[i][1][q - 4];
                if constexpr (F16 != 0) { a[q] = __builtin_bit_cast(__bf16, (_Float16)av); b[q] = __builtin_bit_cast(__bf16, (_Float16)bv); }
                else { a[q] = (__bf16)av; b[q] = (__bf16)bv; }
                if constexpr (X3 != 0) { al[q] = (__bf16)(av - (float)a[q]); bl[q] = (__bf16)(bv - (float)b[q]); }
            }
            *(bf16x8*)&As[buf][0][row * GNB_S + 8 * k8] = a;
            *(bf16x8*)&Bs[buf][0][row * GNB_S + 8 * k8] = b;
            if constexpr (X3 != 0) {
                *(bf16x8*)&As[buf][1][row * GNB_S + 8 * k8] = al;
                *(bf16x8*)&Bs[buf][1][row * GNB_S + 8 * k8] = bl;
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nch = d.K / GNB_KC;
    load(0);
    store(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) load((ch + 1) * GNB_KC);
        const __bf16* ap = &As[ch & 1][0][(32 * wm + n) * GNB_S + 8 * kh];
        const __bf16* bp = &Bs[ch & 1][0][(32 * wn + n) * GNB_S + 8 * kh];
#pragma unroll
        for (int ks = 0; ks < GNB_KC / 16; ++ks) {
            const bf16x8 a = *(const bf16x8*)(ap + 16 * ks), b = *(const bf16x8*)(bp + 16 * ks);
            if constexpr (F16 != 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            if constexpr (X3 != 0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8*)(bp + GNT_T * GNB_S + 16 * ks), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ap + GNT_T * GNB_S + 16 * ks), b, acc, 0, 0, 0);
            }
        }
        if (ch + 1 < nch) store((ch + 1) & 1);
        __syncthreads();
    }
    const int col = n0 + 32 * wn + n;
    if (col < d.N) {
        const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wm + mfma32_row(r, lane);
            if (row < d.M) d.C[(size_t)row * d.ldc + col] = acc[r] + bias;
        }
    }
}

int launch_gnt_gemm_bf16(const GntBatch& gb, hipStream_t st, int x3) {
    int maxM = 0, maxN = 0;
    for (int i = 0; i < gb.n_prob; ++i) {
        const GntProb& q = gb.p[i];
        SED_CHECK_ARG(q.K % GNB_KC == 0 && q.lda % 4 == 0 && q.ldb % 4 == 0 && ((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0,
                      "gnt gemm (bf16): K must be a multiple of 64 and the operands 16-byte aligned");
        maxM = q.M > maxM ? q.M : maxM;
        maxN = q.N > maxN ? q.N : maxN;
    }
    const dim3 grid((maxN + GNT_T - 1) / GNT_T, (maxM + GNT_T - 1) / GNT_T, gb.n_prob);
    if (x3 == 2) k_gnt_gemm_bf16<2><<<grid, 256, 0, st>>>(gb);
    else if (x3) k_gnt_gemm_bf16<1><<<grid, 256, 0, st>>>(gb);
    else k_gnt_gemm_bf16<0><<<grid, 256, 0, st>>>(gb);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_gnt_gemm(const GntBatch& gb, hipStream_t st) {
    int maxM = 0, maxN = 0;
    for (int i = 0; i < gb.n_prob; ++i) {
        const GntProb& q = gb.p[i];
        SED_CHECK_ARG(q.K % GNT_KC == 0 && q.lda % 4 == 0 && q.ldb % 4 == 0 && ((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0,
                      "gnt gemm: K must be a multiple of 32 and the operands 16-byte aligned");
        maxM = q.M > maxM ? q.M : maxM;
        maxN = q.N > maxN ? q.N : maxN;
    }
    k_gnt_gemm<<<dim3((maxN + GNT_T - 1) / GNT_T, (maxM + GNT_T - 1) / GNT_T, gb.n_prob), 256, 0, st>>>(gb);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// out[n][dir * R + k] = w_dir[k][n]  (R rows, N columns each): the two W_ih stacked along K and transposed, so that the
// dX GEMM reads it k-contiguous
__global__ __launch_bounds__(256) void k_gnt_pack_t(const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ out,
                                                     int R, int N) {
    gnt_pack_t_body(w0, w1, out, R, N, (int)blockIdx.x, (int)threadIdx.x);
}
int launch_gnt_pack_t(const float* w0, const float* w1, float* out, int R, int N, hipStream_t st) {
    SED_CHECK_ARG(R % 32 == 0 && N % 32 == 0, "gnt pack: R and N must be multiples of 32");
    k_gnt_pack_t<<<gnt_pack_t_tiles(R, N), 256, 0, st>>>(w0, w1, out, R, N);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
