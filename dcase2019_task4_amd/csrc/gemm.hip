// gemm.hip - small strided fp32 GEMMs on the f32 MFMA for the GRU: input projections
// (x @ W_ih^T + b_ih, torch.nn.GRU inside baseline/models/RNN.py:12), weight / bias gradients and the
// gradient w.r.t. the layer input.  C[m][n] = sum_k A(m,k) * B(k,n) with arbitrary element strides
// (NN / NT / TN from one kernel).
//
// The shapes are awkward for a GPU: either M = B*T/8 (1 872) with tiny N, K, or tiny M x N
// (192 x 64..128) with K = 1 872.  A one-problem-per-launch kernel leaves the chip empty (first
// profile: 3-6 workgroups, 120 us per launch, 39 % of the step).  So one launch takes a BATCH of up
// to 4 independent problems (both directions, W_ih and W_hh) and an optional split-K factor:
//   grid = (N tiles, M tiles, problems x splits); partial tiles go to a scratch buffer and
//   k_gemm_reduce sums them in a fixed order (deterministic; no float atomics) and applies
//   bias / accumulate.  A virtual all-ones column of B (n == N) yields the bias gradients
//   (column sums of A) in the same pass, and B may be the K-concatenation of two matrices
//   (B2 from row k2 on) so dX = [dgi_fwd | dgi_rev] @ [W_ih_fwd ; W_ih_rev] is one launch.
#include "common.h"
#include "kernels.h"
SED_TS_DEFINE(gemm)

#define GT_M 64
#define GT_N 64
#define GT_K 64    // these GEMMs are latency-bound (K = 64..384 per workgroup): 8 float4 loads in flight per thread and
                   // 1-6 trips through the load -> LDS -> MFMA chain (a 128-deep tile needed 256 VGPRs and measured slower)

typedef __attribute__((ext_vector_type(8))) __bf16 gemm_bf16x8;
__device__ __forceinline__ int prob_nx(const GemmProb& p) { return p.N + (p.Cones ? 1 : 0); }

// BF: operands rounded to bf16 on the way from LDS into the MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the
// SED_DTYPE_BF16 mode's GRU weight gradients (K = B T, 9-13 GFLOP at the wide model: 90 us per layer on the f32 MFMA)
// BF = 2 (SED_DTYPE_BF16X3): hi + lo split of both operands, three products per k-step (the LDS tiles hold fp32 anyway)
template <int BF>
__global__ __launch_bounds__(256) void k_gemm_batched(GemmBatch gb) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* As = gsm;                                   // [GT_M][GT_K + 1]
    float* Bs = gsm + GT_M * (GT_K + 1);               // [GT_K][GT_N + 1]
    // Which tile: the workgroups of one (problem, K split) share that split's rows of A (all n tiles) and of B (all m tiles).
    // In launch order they are consecutive, i.e. dealt round-robin over the eight XCDs, and every L2 fetched every chunk: 115 MB
    // of HBM reads per launch for 20 MB of operands at the 256-cell GRU (profiles/r05_wide-bf16_pmc_hbm_traffic.md).  The
    // linear workgroup number is re-read so that XCD x (= number % 8) works through the x-th eighth of the (z, y, x) order.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifndef SED_NO_XCD_ORDER
    {
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
        if ((total & 7) == 0) {
            const int lin = bx + gx * (by + gy * bz), l2 = (lin & 7) * (total >> 3) + (lin >> 3);
            bx = l2 % gx; by = (l2 / gx) % gy; bz = l2 / (gx * gy);
        }
    }
#endif
    const int pi = bz / gb.splits, split = bz % gb.splits;
    const GemmProb& d = gb.p[pi];
    const int Nx = prob_nx(d);
    const int m0 = by * GT_M, n0 = bx * GT_N;
    if (m0 >= d.M || n0 >= d.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    // the all-ones column of B (bias gradients = row sums of A) is NOT a GEMM column: it made a whole extra n-tile whose
    // loads took the ragged path (one branch and one wait per element - the straggler of every launch).  The n-tile-0
    // workgroups sum their A tile's rows on the VALU instead: thread (row tid >> 2, k quarter tid & 3).
    const bool ones_here = d.Cones != nullptr && bx == 0;
    float rowsum = 0.f;
    int kbeg = 0, kend = d.K;
    if (gb.splits > 1) {
        const int chunk = ((d.K + gb.splits - 1) / gb.splits + GT_K - 1) / GT_K * GT_K;
        kbeg = split * chunk;
        kend = min(d.K, kbeg + chunk);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // per-thread tile coordinates (fixed over the K loop); 8 A and 8 B elements per thread per tile, as two
    // groups of 4 consecutive elements along whichever axis is contiguous in memory, so that a group is ONE
    // float4 load when it is in range and 16-byte aligned (scalar loads with per-element bounds checks and
    // 64-bit address arithmetic made the K loop latency-bound: ~1.5 us per iteration)
    constexpr int NG = (GT_M * GT_K) / 256 / 4;          // float4 groups per thread per matrix (= 2)
    int am[NG], ak[NG], bk[NG], bn[NG];
    const bool a_kc = (d.sAk == 1), b_nc = (d.sBn == 1);
#pragma unroll
    for (int it = 0; it < NG; ++it) {
        const int e4 = (tid + 256 * it) * 4;               // first element of the group
        if (a_kc) { am[it] = e4 / GT_K; ak[it] = e4 % GT_K; } else { am[it] = e4 % GT_M; ak[it] = e4 / GT_M; }
        if (b_nc) { bk[it] = e4 / GT_N; bn[it] = e4 % GT_N; } else { bk[it] = e4 % GT_K; bn[it] = e4 / GT_K; }
    }
    float ra[NG][4], rb[NG][4];
    // interior tiles of vector-friendly problems take straight-line float4 loads: with the per-group range / alignment
    // branches below the compiler waits (s_waitcnt vmcnt(0)) behind every single load
    const bool interior = d.vec_ok && m0 + GT_M <= d.M && n0 + GT_N <= d.N;
    (void)Nx;
    auto load_tile = [&](int k0) {
        if (interior && k0 + GT_K <= kend && (d.B2 == nullptr || k0 >= d.k2 || k0 + GT_K <= d.k2)) {
            const float* bbase = (d.B2 != nullptr && k0 >= d.k2) ? d.B2 - (int64_t)d.k2 * d.sBk : d.B;
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                const f32x4 va = *(const f32x4*)(d.A + (int64_t)(m0 + am[it]) * d.sAm + (int64_t)(k0 + ak[it]) * d.sAk);
                const f32x4 vb = *(const f32x4*)(bbase + (int64_t)(k0 + bk[it]) * d.sBk + (int64_t)(n0 + bn[it]) * d.sBn);
#pragma unroll
                for (int q = 0; q < 4; ++q) { ra[it][q] = va[q]; rb[it][q] = vb[q]; }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            // ---- A group: 4 consecutive k (a_kc) or 4 consecutive m ----
            {
                const int m = m0 + am[it], k = k0 + ak[it];
                const float* p = d.A + (int64_t)m * d.sAm + (int64_t)k * d.sAk;
                const bool full = a_kc ? (m < d.M && k + 3 < kend) : (m + 3 < d.M && k < kend);
                if (full && (((uintptr_t)p & 15) == 0)) {
                    const float4 v = *(const float4*)p;
                    ra[it][0] = v.x; ra[it][1] = v.y; ra[it][2] = v.z; ra[it][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mm = a_kc ? m : m + q, kk = a_kc ? k + q : k;
                        ra[it][q] = (mm < d.M && kk < kend) ? d.A[(int64_t)mm * d.sAm + (int64_t)kk * d.sAk] : 0.f;
                    }
                }
            }
            // ---- B group: 4 consecutive n (b_nc) or 4 consecutive k ----
            {
                const int kk0 = k0 + bk[it], nn0 = n0 + bn[it];
                const bool two = d.B2 != nullptr;
                const float* base = (two && kk0 >= d.k2) ? d.B2 - (int64_t)d.k2 * d.sBk : d.B;
                const float* p = base + (int64_t)kk0 * d.sBk + (int64_t)nn0 * d.sBn;
                const bool same_src = !two || b_nc || (kk0 >= d.k2) == (kk0 + 3 >= d.k2);
                const bool full = same_src && (b_nc ? (kk0 < kend && nn0 + 3 < d.N) : (kk0 + 3 < kend && nn0 < d.N));
                if (full && (((uintptr_t)p & 15) == 0)) {
                    const float4 v = *(const float4*)p;
                    rb[it][0] = v.x; rb[it][1] = v.y; rb[it][2] = v.z; rb[it][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int kk = b_nc ? kk0 : kk0 + q, nn = b_nc ? nn0 + q : nn0;
                        float w = 0.f;
                        if (kk < kend) {
                            if (nn < d.N) {
                                w = (two && kk >= d.k2) ? d.B2[(int64_t)(kk - d.k2) * d.sBk + (int64_t)nn * d.sBn]
                                                        : d.B[(int64_t)kk * d.sBk + (int64_t)nn * d.sBn];
                            }
                        }
                        rb[it][q] = w;
                    }
                }
            }
        }
    };
    if (kbeg < kend) load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GT_K) {
#pragma unroll
        for (int it = 0; it < NG; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                As[(a_kc ? am[it] : am[it] + q) * (GT_K + 1) + (a_kc ? ak[it] + q : ak[it])] = ra[it][q];
                Bs[(b_nc ? bk[it] : bk[it] + q) * (GT_N + 1) + (b_nc ? bn[it] + q : bn[it])] = rb[it][q];
            }
        lds_barrier();
        if (k0 + GT_K < kend) load_tile(k0 + GT_K);      // next tile's loads fly under this tile's MFMAs
        if (ones_here) {
            const float* ar = As + (tid >> 2) * (GT_K + 1) + 16 * (tid & 3);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) rowsum += ar[kk];
        }
        if constexpr (BF != 0) {
#pragma unroll
            for (int s = 0; s < GT_K / 16; ++s) {                 // lane (n, kh): k = 16 s + 8 kh + e
                gemm_bf16x8 a, b, al, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float av = As[(32 * wm + n) * (GT_K + 1) + 16 * s + 8 * kh + e];
                    const float bv = Bs[(16 * s + 8 * kh + e) * (GT_N + 1) + 32 * wn + n];
                    a[e] = (__bf16)av;
                    b[e] = (__bf16)bv;
                    if constexpr (BF == 2) { al[e] = (__bf16)(av - (float)a[e]); bl[e] = (__bf16)(bv - (float)b[e]); }
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                if constexpr (BF == 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < GT_K / 2; ++s) {
                const float a = As[(32 * wm + n) * (GT_K + 1) + 2 * s + kh];
                const float b = Bs[(2 * s + kh) * (GT_N + 1) + 32 * wn + n];
                acc = mfma32(a, b, acc);
            }
        }
        lds_barrier();
    }
    if (ones_here) {
        rowsum += __shfl_xor(rowsum, 1);
        rowsum += __shfl_xor(rowsum, 2);
        const int row = m0 + (tid >> 2);
        if ((tid & 3) == 0 && row < d.M) {
            if (gb.splits > 1) gb.part[(size_t)bz * gb.part_stride + (size_t)row * Nx + d.N] = rowsum;
            else d.Cones[row] = rowsum;
        }
    }
    const int col = n0 + 32 * wn + n;
    if (col >= d.N) return;
    if (gb.splits > 1) {
        float* P = gb.part + ((size_t)bz * gb.part_stride);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wm + mfma32_row(r, lane);
            if (row < d.M) P[(size_t)row * Nx + col] = acc[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * wm + mfma32_row(r, lane);
        if (row >= d.M) continue;
        float* c = d.C + (int64_t)row * d.ldc + col;
        float v = acc[r] + (d.bias ? d.bias[col] : 0.f);
        if (d.accumulate) v += *c;
        *c = v;
    }
}

// ---- whole-K-resident variant for the GEMMs on the step's critical path ---------------------------------------
// x @ W_ih^T + b (K = 64 / 128) in front of each GRU layer and dX = dgi @ W_ih (K = 384) behind it: 0.2 GFLOP each,
// but the tiled kernel above took 15-25 us - six trips through load -> LDS -> MFMA, each paying a full memory round
// trip with 60 workgroups on the chip.  Here a workgroup owns a 32 x 64 output tile and fetches its ENTIRE K extent
// up front (up to 36 float4 per thread in flight, one round trip), then the four waves split the tile in two column
// halves x two K halves and combine through LDS.  LDS: 32 x (K+1) + K x 65 floats (146 KB at K = 384).
#define GP_M 32
#define GP_N 64
#define GP_KMAX 384
__global__ __launch_bounds__(256) void k_gemm_panel(GemmBatch gb) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const GemmProb& d = gb.p[blockIdx.z];
    const int m0 = blockIdx.y * GP_M, n0 = blockIdx.x * GP_N;
    if (m0 >= d.M || n0 >= d.N) return;
    const int K = d.K, SA = K + 1;
    float* As = gsm;                    // [GP_M][SA]
    float* Bs = gsm + GP_M * SA;        // [K][GP_N + 1]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const bool a_kc = (d.sAk == 1), b_nc = (d.sBn == 1);
    constexpr int NGA = GP_M * GP_KMAX / 4 / 256, NGB = GP_KMAX * GP_N / 4 / 256;     // 12, 24
    const int nga = GP_M * K / 4 / 256, ngb = K * GP_N / 4 / 256;                       // K is a multiple of 32
    f32x4 ra[NGA], rb[NGB];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    TS(0); TSC(14);
    // Straight-line loads, no per-group checks: the launcher only picks this kernel when every group is one aligned
    // float4 (16-byte aligned bases, unit stride along the contiguous axis, other strides multiples of 4, N a multiple
    // of 64); rows past M are clamped to the last row (loaded twice, never stored).  With per-group range/alignment
    // branches the compiler put an s_waitcnt vmcnt(0) behind every load: 36 serialized round trips, 10 us.
#pragma unroll
    for (int it = 0; it < NGA; ++it) {
        ra[it] = z4;
        if (it < nga) {
            // 8 threads per A row (k-contiguous) / per k (m-contiguous), 4 elements each
            int m = m0 + (a_kc ? (tid >> 3) : (tid & 7) * 4);
            const int k = a_kc ? ((tid & 7) + 8 * it) * 4 : (tid >> 3) + 32 * it;
            if (a_kc) m = min(m, d.M - 1); else m = min(m, d.M - 4);
            ra[it] = *(const f32x4*)(d.A + (int64_t)m * d.sAm + (int64_t)k * d.sAk);
        }
    }
#pragma unroll
    for (int it = 0; it < NGB; ++it) {
        rb[it] = z4;
        if (it < ngb) {
            // 16 threads per k row (n-contiguous) / 4 threads per n row (k-contiguous)
            const int k = b_nc ? (tid >> 4) + 16 * it : ((tid & 3) + 4 * it) * 4, nn = n0 + (b_nc ? (tid & 15) * 4 : (tid >> 2));
            const float* base = (d.B2 != nullptr && k >= d.k2) ? d.B2 - (int64_t)d.k2 * d.sBk : d.B;      // k2 % 4 == 0
            rb[it] = *(const f32x4*)(base + (int64_t)k * d.sBk + (int64_t)nn * d.sBn);
        }
    }
#pragma unroll
    for (int it = 0; it < NGA; ++it) {
        if (it < nga) {
            const int ml = a_kc ? (tid >> 3) : (tid & 7) * 4, k = a_kc ? ((tid & 7) + 8 * it) * 4 : (tid >> 3) + 32 * it;
#pragma unroll
            for (int q = 0; q < 4; ++q) As[(a_kc ? ml : ml + q) * SA + (a_kc ? k + q : k)] = ra[it][q];
        }
    }
#pragma unroll
    for (int it = 0; it < NGB; ++it) {
        if (it < ngb) {
            const int k = b_nc ? (tid >> 4) + 16 * it : ((tid & 3) + 4 * it) * 4, nl = b_nc ? (tid & 15) * 4 : (tid >> 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) Bs[(b_nc ? k : k + q) * (GP_N + 1) + (b_nc ? nl + q : nl)] = rb[it][q];
        }
    }
    TS(1);
    __syncthreads();
    TS(2);
    const int sub = wv & 1, khalf = wv >> 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int kb = khalf * (K / 2);
        const float* Ap = As + n * SA + kb + kh;
        const float* Bp = Bs + (kb + kh) * (GP_N + 1) + 32 * sub + n;
        const int steps = K / 4;                          // a multiple of 8 (K % 32 == 0)
        for (int s8 = 0; s8 < steps; s8 += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = Ap[2 * (s8 + u)]; bv[u] = Bp[2 * (s8 + u) * (GP_N + 1)]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma32(av[u], bv[u], acc);
        }
    }
    TS(3);
    __syncthreads();
    float* red = gsm;                   // [2][32][33], over the A panel (dead now)
    if (khalf == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(sub * 32 + mfma32_row(r, lane)) * 33 + n] = acc[r];
    }
    __syncthreads();
    if (khalf == 0) {
        const int col = n0 + 32 * sub + n;
        if (col < d.N) {
            const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = mfma32_row(r, lane), row = m0 + rl;
                if (row < d.M) {
                    float* c = d.C + (int64_t)row * d.ldc + col;
                    float v = acc[r] + red[(sub * 32 + rl) * 33 + n] + bias;
                    if (d.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
    TS(4); TSC(15);
}

__global__ __launch_bounds__(256) void k_gemm_reduce(GemmBatch gb) {
    const int pi = blockIdx.y;
    const GemmProb& d = gb.p[pi];
    const int Nx = prob_nx(d);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.M * Nx) return;
    const float* P = gb.part + (size_t)pi * gb.splits * gb.part_stride + i;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= gb.splits; z += 4) {
        s0 += P[(size_t)(z + 0) * gb.part_stride];
        s1 += P[(size_t)(z + 1) * gb.part_stride];
        s2 += P[(size_t)(z + 2) * gb.part_stride];
        s3 += P[(size_t)(z + 3) * gb.part_stride];
    }
    for (; z < gb.splits; ++z) s0 += P[(size_t)z * gb.part_stride];
    float v = (s0 + s1) + (s2 + s3);
    const int row = i / Nx, col = i % Nx;
    if (col < d.N) {
        float* c = d.C + (int64_t)row * d.ldc + col;
        v += d.bias ? d.bias[col] : 0.f;
        if (d.accumulate) v += *c;
        *c = v;
    } else {
        d.Cones[row] = v;
    }
}

// out[n] = sum_m A[m*lda + n]   (few rows: the per-clip partials of the head gradients)
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

size_t gemm_part_floats(int n_prob, int splits, int max_m, int max_nx) {
    return (size_t)n_prob * splits * max_m * max_nx;
}

int launch_gemm_batch(GemmBatch& gb, hipStream_t st) {
    int maxM = 0, maxNx = 0;
    for (int i = 0; i < gb.n_prob; ++i) {
        const int nx = gb.p[i].N + (gb.p[i].Cones ? 1 : 0);
        maxM = gb.p[i].M > maxM ? gb.p[i].M : maxM;
        maxNx = nx > maxNx ? nx : maxNx;
    }
    if (gb.splits < 1) gb.splits = 1;
    bool panel = (gb.splits == 1) && !gb.bf16;
    int maxK = 0, maxN = 0;
    for (int i = 0; i < gb.n_prob; ++i) {
        const GemmProb& q = gb.p[i];
        const bool a_kc = (q.sAk == 1), b_nc = (q.sBn == 1);
        panel = panel && q.K <= GP_KMAX && (q.K % 32) == 0 && q.Cones == nullptr && (q.B2 == nullptr || (q.k2 % 4) == 0) &&
                (q.N % GP_N) == 0 && q.M >= 4 && (q.M % 4) == 0 &&
                (a_kc ? (q.sAm % 4) == 0 : (q.sAm == 1 && (q.sAk % 4) == 0)) &&
                (b_nc ? (q.sBk % 4) == 0 : (q.sBk == 1 && (q.sBn % 4) == 0)) &&
                ((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0 && (q.B2 == nullptr || ((uintptr_t)q.B2 % 16) == 0);
        maxK = q.K > maxK ? q.K : maxK;
        maxN = q.N > maxN ? q.N : maxN;
    }
    if (panel) {
        const size_t lds = (size_t)(GP_M * (maxK + 1) + maxK * (GP_N + 1)) * sizeof(float);
        static size_t lds_set = 0;
        if (lds > lds_set) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_set = lds;
        }
        dim3 grid((maxN + GP_N - 1) / GP_N, (maxM + GP_M - 1) / GP_M, gb.n_prob);
        k_gemm_panel<<<grid, 256, lds, st>>>(gb);
        SED_CHECK_LAUNCH();
        return SED_OK;
    }
    for (int i = 0; i < gb.n_prob; ++i) {
        GemmProb& q = gb.p[i];
        const bool a_kc = (q.sAk == 1), b_nc = (q.sBn == 1);
        q.vec_ok = (a_kc ? (q.sAm % 4) == 0 : (q.sAm == 1 && (q.sAk % 4) == 0)) &&
                   (b_nc ? (q.sBk % 4) == 0 : (q.sBk == 1 && (q.sBn % 4) == 0)) &&
                   ((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0 && (q.B2 == nullptr || ((uintptr_t)q.B2 % 16) == 0);
    }
    if (gb.splits > 1) {
        SED_CHECK_ARG(gb.part != nullptr, "split-K gemm needs a partial buffer");
        gb.part_stride = (size_t)maxM * maxNx;
        if ((size_t)gb.n_prob * gb.splits * gb.part_stride > gb.part_floats) {
            sed_set_error("split-K gemm: partial buffer holds %zu floats, the batch needs %zu", gb.part_floats,
                          (size_t)gb.n_prob * gb.splits * gb.part_stride);
            return SED_ERR_WORKSPACE;
        }
    }
    dim3 grid((maxN + GT_N - 1) / GT_N, (maxM + GT_M - 1) / GT_M, gb.n_prob * gb.splits);
    const size_t lds = (size_t)(GT_M * (GT_K + 1) + GT_K * (GT_N + 1)) * sizeof(float);
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_batched<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_batched<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_batched<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (gb.bf16 == 2) k_gemm_batched<2><<<grid, 256, lds, st>>>(gb);
    else if (gb.bf16) k_gemm_batched<1><<<grid, 256, lds, st>>>(gb);
    else k_gemm_batched<0><<<grid, 256, lds, st>>>(gb);
    SED_CHECK_LAUNCH();
    if (gb.splits > 1) {
        dim3 g2((maxM * maxNx + 255) / 256, gb.n_prob);
        k_gemm_reduce<<<g2, 256, 0, st>>>(gb);
        SED_CHECK_LAUNCH();
    }
    return SED_OK;
}

int launch_colsum(const float* A, int M, int N, int64_t lda, float* out, hipStream_t st) {
    k_colsum<<<(N + 63) / 64, 256, 0, st>>>(A, M, N, lda, out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
