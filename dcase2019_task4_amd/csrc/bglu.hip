// bglu.hip - BatchNorm-apply + GLU + dropout + AvgPool2d((2,4)) of conv blocks 1 and 2 for SED_DTYPE_BF16 (activations
// stored as bf16), forward and backward - the round-3 replacement of gglu.hip's kernels in that mode.
//
// Reference ops (baseline/models/CNN.py:49-67, GLU CNN.py:11-16):
//   z = BatchNorm2d(y);  lin = Linear(C, C)(z over channels);  out = lin * sigmoid(z);  p = AvgPool2d((2,4))(Dropout(out))
//
// gglu.hip in this mode was 45 - 100 us forward / 115 - 150 us backward per launch at C = 128 against ~6 / ~15 us of HBM
// time: it streamed the C x C weights through LDS in chunks with a barrier per chunk, kept fp32 AND bf16 copies of every
// tile in LDS, and read the gate operand back from LDS one scalar at a time (MFMA pipe busy 0.03 - 0.04).  Here:
//   * The BatchNorm affine is folded INTO the weights in the kernel's prologue: lin = W' y + b' with W' = Wglu diag(gamma *
//     invstd), b' = bglu + Wglu (beta - mean * gamma * invstd) - the MFMA's A operand is the stored bf16 y itself.
//   * A wave owns ONE 32-channel block for the whole kernel: its rows of W' (C / 16 fragments = 32 registers at C = 128)
//     live in registers; no weight traffic, no weight barriers.
//   * The gate needs y[pixel][channel] in the accumulator layout (lane = channel, registers = pixels).  Instead of 16 scalar
//     LDS reads per 32 x 32 block it comes from TWO extra MFMAs against an identity fragment: D = y I, exact in fp32.
//   * 32-pixel row blocks (4 pooled pixels) are staged through a double-buffered padded LDS tile with 16-byte loads, the
//     next block's loads in flight during the current block's MFMAs: one barrier per row block, several workgroups per CU.
#include "gen.h"
#include "kernels.h"
#include "gkernels.h"

// BatchNorm statistics -> mean / invstd in LDS (+ running statistics and the [4][C] record for the backward, by workgroup 0)
__device__ __forceinline__ void bglu_bn_prep(const GBnArgs& a, int C, int c, bool publish, float* bn_s /* LDS [2][C] */) {
    double mean, var;
    if (a.train) {
        mean = a.stat[c] / a.N;
        var = a.stat[C + c] / a.N - mean * mean;
        if (var < 0) var = 0;
        if (a.update && publish) {
            a.run_mean[c] = (float)((1.0 - a.momentum) * a.run_mean[c] + a.momentum * mean);
            a.run_var[c] = (float)((1.0 - a.momentum) * a.run_var[c] + a.momentum * var * a.N / (a.N - 1.0));
            if (c == 0 && a.tracked) a.tracked[0] += 1;
        }
    } else {
        mean = a.run_mean[c];
        var = a.run_var[c];
    }
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const double scale = a.gamma[c] * invstd;
    bn_s[c] = (float)mean; bn_s[C + c] = (float)invstd;
    if (publish) {
        a.bn[c] = (float)mean; a.bn[C + c] = (float)invstd; a.bn[2 * C + c] = (float)scale;
        a.bn[3 * C + c] = (float)(a.beta[c] - mean * scale);
    }
}

template <int C>
struct BGluCfg {
    static constexpr int NB = C / 32, KS = C / 16, RPR = 4 / NB;          // channel blocks, k-steps, row blocks per round
    static constexpr int PS = 2 * C + 16;                                 // tile pixel stride (bytes): conflict-free b128 reads
    static constexpr int TILE = 32 * PS;                                  // one row block
    static constexpr int CJ = C / 8;                                      // 16-byte groups per pixel
};

// this wave's B fragments of the folded weights (rows 32 cb + n of W'), its folded bias and the gate's scale / shift
//   aff: LDS [2][C] = scale (gamma * invstd) | shift (beta - mean * scale) of the BatchNorm affine
template <int C>
__device__ __forceinline__ void bglu_fold(const float* __restrict__ wglu, const float* __restrict__ bglu, const float* aff, int cb,
                                          int lane, bf16x8 (&bw)[C / 16], float& b_fold, float& sc, float& sh) {
    const int n = lane & 31, kh = lane >> 5, co = 32 * cb + n;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) {
        const int k0 = 16 * ks + 8 * kh;
        const f32x4 w0 = *(const f32x4*)(wglu + (size_t)co * C + k0), w1 = *(const f32x4*)(wglu + (size_t)co * C + k0 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = e < 4 ? w0[e] : w1[e - 4];
            bw[ks][e] = (__bf16)(w * aff[k0 + e]);
            part = fmaf(w, aff[C + k0 + e], part);
        }
    }
    b_fold = bglu[co] + part + __shfl_xor(part, 32);
    sc = aff[co];
    sh = aff[C + co];
}
// identity fragments: B[k][j] = (k == j) for the two k-steps that cover a 32-channel block
__device__ __forceinline__ void bglu_identity(int lane, bf16x8 (&idf)[2]) {
    const int n = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) idf[s][e] = (__bf16)((16 * s + 8 * kh + e) == n ? 1.0f : 0.0f);
}

// 16-byte staging items of a round: (row block rl, pixel m, channel group j); 512 per round = 2 per thread
template <int C>
__device__ __forceinline__ void bglu_item(int g, int& rl, int& m, int& j) {
    constexpr int CJ = C / 8;
    j = g % CJ; m = (g / CJ) % 32; rl = g / (32 * CJ);
}

// PB: the pooled output is stored as bf16 (block 1); block 2's output p2 feeds the fp32 GRU
template <int C, int PB>
__global__ __launch_bounds__(256) void k_bglu_fwd(const __bf16* __restrict__ y, GBnArgs bnp, const float* __restrict__ wglu,
                                                   const float* __restrict__ bglu, void* __restrict__ p_v, int H, int W, int Ho,
                                                   int Wo, int Q, int block_id, int use_drop, float p_drop,
                                                   const uint64_t* __restrict__ seed_ptr, uint16_t* __restrict__ mask_out) {
    using Cfg = BGluCfg<C>;
    using PT = typename Stor<PB>::T;
    constexpr int NB = Cfg::NB, KS = Cfg::KS, RPR = Cfg::RPR, PS = Cfg::PS, TILE = Cfg::TILE;
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][RPR][TILE];
    __shared__ float bn_s[2 * C];
    __shared__ float aff[2 * C];
    PT* p = (PT*)p_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int rl = wv / NB, cb = wv % NB;
    if (tid < C) {
        bglu_bn_prep(bnp, C, tid, blockIdx.x == 0, bn_s);
        const float s = bnp.gamma[tid] * bn_s[C + tid];
        aff[tid] = s; aff[C + tid] = bnp.beta[tid] - bn_s[tid] * s;
    }
    __syncthreads();
    bf16x8 bw[KS], idf[2];
    float b_fold, sc, sh;
    bglu_fold<C>(wglu, bglu, aff, cb, lane, bw, b_fold, sc, sh);
    bglu_identity(lane, idf);
    sc *= SED_NEG_LOG2E; sh *= SED_NEG_LOG2E;                           // sigmoid(z) = 1 / (1 + exp2(-log2(e) z))
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float scp = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    const int n_rb = (Q + 3) / 4, n_round = (n_rb + RPR - 1) / RPR;
    f32x4 st[2];
    auto load = [&](int round) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + 256 * i, r2, m, j);
            const int rb = round * RPR + r2, q = rb * 4 + (m >> 3);
            st[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (round < n_round && q < Q) {
                const int pix = gen_rb_pixel(q, (m >> 2) & 1, m & 3, H, W, Ho, Wo);
                st[i] = *(const f32x4*)(y + (size_t)pix * C + 8 * j);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + 256 * i, r2, m, j);
            *(f32x4*)(&tile[buf][r2][0] + m * PS + 16 * j) = st[i];
        }
    };
    load(blockIdx.x);
    int it = 0;
    for (int round = blockIdx.x; round < n_round; round += gridDim.x, ++it) {
        const int buf = it & 1;
        store(buf);
        __syncthreads();
        load(round + gridDim.x);                                         // flies during this round's MFMAs and epilogue
        const int rb = round * RPR + rl, q0 = rb * 4;
        if (rb < n_rb) {
            const unsigned char* tp = &tile[buf][rl][0] + n * PS + 16 * kh;
            f32x16 lin, yid;
#pragma unroll
            for (int r = 0; r < 16; ++r) { lin[r] = 0.f; yid[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                lin = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tp + 32 * ks), bw[ks], lin, 0, 0, 0);
            // y[pixel][this wave's channels] in the accumulator layout: the two k-steps of the block against the identity
            yid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tp + 64 * cb), idf[0], yid, 0, 0, 0);
            yid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tp + 64 * cb + 32), idf[1], yid, 0, 0, 0);
            const int c = 32 * cb + n;
            uint32_t m16 = 0xffffu;
            if (use_drop) {
                m16 = gen_keep16(rb, cb, C, lane, block_id, seed, thr);
                if (mask_out) mask_out[((size_t)rb * NB + cb) * 64 + lane] = (uint16_t)m16;
            }
            float pooled[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = (lin[r] + b_fold) * sigmoid_from_scaled(fmaf(sc, yid[r], sh));
                pooled[r >> 2] += ((m16 >> r) & 1u) ? v : 0.f;
            }
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) pooled[jx] += __shfl_xor(pooled[jx], 32);
            const int j0 = 2 * kh;
            if (q0 + j0 < Q) st1(p + (size_t)(q0 + j0) * C + c, (kh ? pooled[2] : pooled[0]) * scp);
            if (q0 + j0 + 1 < Q) st1(p + (size_t)(q0 + j0 + 1) * C + c, (kh ? pooled[3] : pooled[1]) * scp);
        }
    }
}

template <int C, int PB>
static int bglu_fwd_launch(const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int B, int H, int W,
                           int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st) {
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo, n_rb = (Q + 3) / 4, n_round = (n_rb + BGluCfg<C>::RPR - 1) / BGluCfg<C>::RPR;
    const int grid = n_round < 512 ? n_round : 512;          // two workgroups per CU: each pays the fold of its weights once
    k_bglu_fwd<C, PB><<<grid, 256, 0, st>>>((const __bf16*)y, bn, wglu, bglu, p, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, seed, mask_out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_bglu_fwd(int C, const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int p_bf16, int B, int H,
                    int W, int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st) {
#define BGLU_CASE(CC, PP) \
    if (C == CC && p_bf16 == PP) return bglu_fwd_launch<CC, PP>(y, bn, wglu, bglu, p, B, H, W, block_id, use_drop, p_drop, seed, mask_out, st)
    BGLU_CASE(64, 0); BGLU_CASE(64, 1); BGLU_CASE(128, 0); BGLU_CASE(128, 1);
#undef BGLU_CASE
    sed_set_error("bglu forward: unsupported channels %d", C);
    return SED_ERR_UNSUPPORTED;
}
