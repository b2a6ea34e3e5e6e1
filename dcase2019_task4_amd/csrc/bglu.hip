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
#include <type_traits>
#include "gen.h"
#include "kernels.h"
#include "gkernels.h"
SED_TS_DEFINE(bglu)

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
// F16 (SED_DTYPE_F16): bw holds fp16 bit patterns (the forward's operand type); bw16 - if given - the bf16 rounding of the same
// W' for the backward kernel (wfold_out)
template <int C, int X3 = 0, int F16 = 0>
__device__ __forceinline__ void bglu_fold(const float* __restrict__ wglu, const float* __restrict__ bglu, const float* aff, int cb,
                                          int lane, bf16x8 (&bw)[C / 16], float& b_fold, float& sc, float& sh,
                                          bf16x8 (*bl)[C / 16] = nullptr /* X3: the lo parts of W' */,
                                          bf16x8 (*bw16)[C / 16] = nullptr) {
    const int n = lane & 31, kh = lane >> 5, co = 32 * cb + n;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) {
        const int k0 = 16 * ks + 8 * kh;
        const f32x4 w0 = *(const f32x4*)(wglu + (size_t)co * C + k0), w1 = *(const f32x4*)(wglu + (size_t)co * C + k0 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = e < 4 ? w0[e] : w1[e - 4];
            const float wf = w * aff[k0 + e];
            if constexpr (F16 != 0) {
                bw[ks][e] = __builtin_bit_cast(__bf16, (_Float16)wf);
                (*bw16)[ks][e] = (__bf16)wf;
            } else
            bw[ks][e] = (__bf16)wf;
            if constexpr (X3 != 0) (*bl)[ks][e] = (__bf16)(wf - (float)bw[ks][e]);
            part = fmaf(w, aff[C + k0 + e], part);
        }
    }
    b_fold = bglu[co] + part + __shfl_xor(part, 32);
    sc = aff[co];
    sh = aff[C + co];
}
// identity fragments: B[k][j] = (k == j) for the two k-steps that cover a 32-channel block
template <int F16 = 0>
__device__ __forceinline__ void bglu_identity(int lane, bf16x8 (&idf)[2]) {
    const int n = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (16 * s + 8 * kh + e) == n ? 1.0f : 0.0f;
            if constexpr (F16 != 0) idf[s][e] = __builtin_bit_cast(__bf16, (_Float16)v);
            else idf[s][e] = (__bf16)v;
        }
}

// 16-byte staging items of a round: (row block rl, pixel m, channel group j); 512 per round = 2 per thread
template <int C>
__device__ __forceinline__ void bglu_item(int g, int& rl, int& m, int& j) {
    constexpr int CJ = C / 8;
    j = g % CJ; m = (g / CJ) % 32; rl = g / (32 * CJ);
}

// PB: the pooled output is stored as bf16 (block 1); block 2's output p2 feeds the fp32 GRU
// X3 (SED_DTYPE_BF16X3): y is fp32 in HBM and split hi + lo on its way into LDS (two tile planes); lin = y_hi W'_hi + y_hi W'_lo +
// y_lo W'_hi, the gate's y = (y_hi + y_lo) I - three / four MFMAs where the bf16 mode has one / two, fp32 output.  Replaces
// gglu.hip's exact-fp32 kernel in that mode (88 us per launch at C = 128: weights streamed through LDS with a barrier per chunk).
// F16 (SED_DTYPE_F16): y arrives as fp16, W' and the identity are fp16 fragments, v_mfma_f32_32x32x16_f16; the pooled output of
// block 1 (PB) is stored as fp16 in p and - for the backward kernels of the bf16 family - as bf16 in p_b16 (may be null); y_b16
// (may be null) receives the bf16 copy of the INPUT tile y on its way into LDS (the convolution that produced it stores fp16 only)
template <int C, int PB, int X3 = 0, int F16 = 0>
__global__ __launch_bounds__(256) void k_bglu_fwd(const void* __restrict__ y_v, GBnArgs bnp, const float* __restrict__ wglu,
                                                   const float* __restrict__ bglu, void* __restrict__ p_v, int H, int W, int Ho,
                                                   int Wo, int Q, int block_id, int use_drop, float p_drop,
                                                   const uint64_t* __restrict__ seed_ptr, uint16_t* __restrict__ mask_out,
                                                   __bf16* __restrict__ wfold_out, float* __restrict__ bfold_out,
                                                   __bf16* __restrict__ p_b16, __bf16* __restrict__ y_b16) {
    using Cfg = BGluCfg<C>;
    using PT = typename std::conditional<(F16 != 0 && PB != 0), _Float16, typename Stor<PB>::T>::type;
    auto mma = [](bf16x8 a, bf16x8 b, f32x16 c) -> f32x16 {
        if constexpr (F16 != 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };
    constexpr int NB = Cfg::NB, KS = Cfg::KS, RPR = Cfg::RPR, PS = Cfg::PS, TILE = Cfg::TILE;
    constexpr int NPL = X3 ? 2 : 1;                                     // tile planes (hi | lo)
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][NPL][RPR][TILE];
    __shared__ float bn_s[2 * C];
    __shared__ float aff[2 * C];
    PT* p = (PT*)p_v;
    using YT = typename std::conditional<X3 != 0, float, __bf16>::type;
    const YT* y = (const YT*)y_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, kh = lane >> 5;
    const int rl = wv / NB, cb = wv % NB;
    if (tid < C) {
        bglu_bn_prep(bnp, C, tid, blockIdx.x == 0, bn_s);
        const float s = bnp.gamma[tid] * bn_s[C + tid];
        aff[tid] = s; aff[C + tid] = bnp.beta[tid] - bn_s[tid] * s;
    }
    __syncthreads();
    bf16x8 bw[KS], bl[X3 ? KS : 1], idf[2];
    float b_fold, sc, sh;
    const bool publish = !X3 && wfold_out != nullptr && blockIdx.x == 0 && rl == 0;
    if constexpr (X3 != 0) bglu_fold<C, 1>(wglu, bglu, aff, cb, lane, bw, b_fold, sc, sh, &bl);
    else if constexpr (F16 != 0) {
        bf16x8 bw16[KS];
        bglu_fold<C, 0, 1>(wglu, bglu, aff, cb, lane, bw, b_fold, sc, sh, nullptr, &bw16);
        if (publish) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) *(bf16x8*)(wfold_out + (size_t)(32 * cb + n) * C + 16 * ks + 8 * kh) = bw16[ks];
        }
    } else bglu_fold<C>(wglu, bglu, aff, cb, lane, bw, b_fold, sc, sh);
    bglu_identity<F16>(lane, idf);
    if (publish) {
        // workgroup 0 publishes the folded weights W' [C][C] (bf16) and bias b' [C] for the backward kernel, which then
        // starts from two 16-byte-vector copies instead of redoing the fold in every workgroup
        if constexpr (F16 == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) *(bf16x8*)(wfold_out + (size_t)(32 * cb + n) * C + 16 * ks + 8 * kh) = bw[ks];
        }
        if (kh == 0) bfold_out[32 * cb + n] = b_fold;
    }
    sc *= SED_NEG_LOG2E; sh *= SED_NEG_LOG2E;                           // sigmoid(z) = 1 / (1 + exp2(-log2(e) z))
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float scp = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    const int n_rb = (Q + 3) / 4, n_round = (n_rb + RPR - 1) / RPR;
    f32x4 st[2][X3 ? 2 : 1];                                            // an item = 8 channels of one pixel: 16 B of bf16 / 32 B of fp32
    auto load = [&](int round) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + 256 * i, r2, m, j);
            const int rb = round * RPR + r2, q = rb * 4 + (m >> 3);
            const bool ok = round < n_round && q < Q;
            const int pix = gen_rb_pixel(ok ? q : 0, (m >> 2) & 1, m & 3, H, W, Ho, Wo);
#pragma unroll
            for (int hhalf = 0; hhalf < (X3 ? 2 : 1); ++hhalf) {
                st[i][hhalf] = *(const f32x4*)((const char*)(y + (size_t)pix * C + 8 * j) + 16 * hhalf);    // unconditional (clamped)
                if (!ok) st[i][hhalf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + 256 * i, r2, m, j);
            if constexpr (X3 != 0) {
                bf16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? st[i][0][e] : st[i][1][e - 4];
                    hi[e] = (__bf16)v;
                    lo[e] = (__bf16)(v - (float)hi[e]);
                }
                *(bf16x8*)(&tile[buf][0][r2][0] + m * PS + 16 * j) = hi;
                *(bf16x8*)(&tile[buf][1][r2][0] + m * PS + 16 * j) = lo;
            } else {
                *(f32x4*)(&tile[buf][0][r2][0] + m * PS + 16 * j) = st[i][0];
            }
        }
    };
    // F16: the bf16 copy of the items this thread just loaded (same addresses as the load; items past the end are skipped)
    auto copy_y = [&](int round) {
        if constexpr (F16 != 0) {
            if (y_b16 == nullptr) return;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int r2, m, j;
                bglu_item<C>(tid + 256 * i, r2, m, j);
                const int rb = round * RPR + r2, q = rb * 4 + (m >> 3);
                if (round < n_round && q < Q) {
                    const int pix = gen_rb_pixel(q, (m >> 2) & 1, m & 3, H, W, Ho, Wo);
                    const f16x8 h = __builtin_bit_cast(f16x8, st[i][0]);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(float)h[e];
                    *(bf16x8*)(y_b16 + (size_t)pix * C + 8 * j) = o;
                }
            }
        }
    };
    if constexpr (F16 != 0) {
        // rows of y that no pooled pixel covers (H odd: AvgPool2d((2, 4)) drops the last row, e.g. H = 157 at T = 628) are never
        // staged by the loop below - but the BatchNorm backward reads EVERY pixel of y (dy = ca dz + cb y + cc): their bf16 copy
        // is made here, or the backward kernels read uninitialised memory there
        if (y_b16 != nullptr && (H & 1)) {
            const int Bn = Q / (Ho * Wo), per_row = W * C / 8;
            for (int i = blockIdx.x * 256 + tid; i < Bn * per_row; i += gridDim.x * 256) {
                const size_t e = ((size_t)((i / per_row) * H + (H - 1)) * W) * C + 8 * (i % per_row);
                const f16x8 h = *(const f16x8*)((const _Float16*)y_v + e);
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (__bf16)(float)h[q];
                *(bf16x8*)(y_b16 + e) = o;
            }
        }
    }
    load(blockIdx.x);
    int it = 0;
    for (int round = blockIdx.x; round < n_round; round += gridDim.x, ++it) {
        const int buf = it & 1;
        store(buf);
        copy_y(round);
        lds_barrier();                                                   // (LDS-only: the pooled stores of the previous round stay in flight)
        load(round + gridDim.x);                                         // flies during this round's MFMAs and epilogue
        const int rb = round * RPR + rl, q0 = rb * 4;
        if (rb < n_rb) {
            const unsigned char* tp = &tile[buf][0][rl][0] + n * PS + 16 * kh;
            f32x16 lin, yid;
#pragma unroll
            for (int r = 0; r < 16; ++r) { lin[r] = 0.f; yid[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *(const bf16x8*)(tp + 32 * ks);
                lin = mma(a, bw[ks], lin);
                if constexpr (X3 != 0) {
                    lin = mma(a, bl[ks], lin);
                    lin = mma(*(const bf16x8*)(tp + RPR * TILE + 32 * ks), bw[ks], lin);
                }
            }
            // y[pixel][this wave's channels] in the accumulator layout: the two k-steps of the block against the identity
            yid = mma(*(const bf16x8*)(tp + 64 * cb), idf[0], yid);
            yid = mma(*(const bf16x8*)(tp + 64 * cb + 32), idf[1], yid);
            if constexpr (X3 != 0) {
                yid = mma(*(const bf16x8*)(tp + RPR * TILE + 64 * cb), idf[0], yid);
                yid = mma(*(const bf16x8*)(tp + RPR * TILE + 64 * cb + 32), idf[1], yid);
            }
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
            if constexpr (F16 != 0 && PB != 0) {
                if (p_b16) {
                    if (q0 + j0 < Q) st1(p_b16 + (size_t)(q0 + j0) * C + c, (kh ? pooled[2] : pooled[0]) * scp);
                    if (q0 + j0 + 1 < Q) st1(p_b16 + (size_t)(q0 + j0 + 1) * C + c, (kh ? pooled[3] : pooled[1]) * scp);
                }
            }
        }
    }
}

template <int C, int PB, int X3 = 0, int F16 = 0>
static int bglu_fwd_launch(const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int B, int H, int W,
                           int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, void* wfold_out,
                           float* bfold_out, hipStream_t st, void* p_b16 = nullptr, void* y_b16 = nullptr) {
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo, n_rb = (Q + 3) / 4, n_round = (n_rb + BGluCfg<C>::RPR - 1) / BGluCfg<C>::RPR;
    const int grid = n_round < 512 ? n_round : 512;          // two workgroups per CU: each pays the fold of its weights once
    k_bglu_fwd<C, PB, X3, F16><<<grid, 256, 0, st>>>(y, bn, wglu, bglu, p, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, seed, mask_out,
                                                    (__bf16*)wfold_out, bfold_out, (__bf16*)p_b16, (__bf16*)y_b16);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_bglu_fwd(int C, const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int p_bf16, int B, int H,
                    int W, int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, void* wfold_out,
                    float* bfold_out, hipStream_t st, int f16, void* p_b16, void* y_b16) {
    if (f16) {      // SED_DTYPE_F16: y is fp16; p (block 1) fp16 + its bf16 copy p_b16; block 2's p stays fp32
#define BGLU_CASE(CC, PP) \
    if (C == CC && p_bf16 == PP) return bglu_fwd_launch<CC, PP, 0, 1>(y, bn, wglu, bglu, p, B, H, W, block_id, use_drop, p_drop, seed, mask_out, wfold_out, bfold_out, st, p_b16, y_b16)
        BGLU_CASE(64, 0); BGLU_CASE(64, 1); BGLU_CASE(128, 0); BGLU_CASE(128, 1);
#undef BGLU_CASE
    }
#define BGLU_CASE(CC, PP) \
    if (C == CC && p_bf16 == PP) return bglu_fwd_launch<CC, PP>(y, bn, wglu, bglu, p, B, H, W, block_id, use_drop, p_drop, seed, mask_out, wfold_out, bfold_out, st)
    BGLU_CASE(64, 0); BGLU_CASE(64, 1); BGLU_CASE(128, 0); BGLU_CASE(128, 1);
#undef BGLU_CASE
    sed_set_error("bglu forward: unsupported channels %d", C);
    return SED_ERR_UNSUPPORTED;
}
// SED_DTYPE_BF16X3: fp32 y in, fp32 p out, split operands; the backward of this mode (gglu.hip) reads what the packing pass and
// the BatchNorm record provide, nothing the forward kernel publishes
int launch_bglu_fwd_x3(int C, const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int B, int H, int W,
                       int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st) {
    if (C == 64) return bglu_fwd_launch<64, 0, 1>(y, bn, wglu, bglu, p, B, H, W, block_id, use_drop, p_drop, seed, mask_out, nullptr, nullptr, st);
    if (C == 128) return bglu_fwd_launch<128, 0, 1>(y, bn, wglu, bglu, p, B, H, W, block_id, use_drop, p_drop, seed, mask_out, nullptr, nullptr, st);
    sed_set_error("bglu forward (split operands): unsupported channels %d", C);
    return SED_ERR_UNSUPPORTED;
}

// ---- backward ------------------------------------------------------------------------------------------------------------
// Per 32-pixel row block, wave w = (row block rl, channel block cb):
//   P1  lin[:, cb] = y W'^T (register-resident fragments of W'), y[:, cb] through the identity          -> sigma, dlin, gate path
//       dlin (bf16) goes to an LDS tile [pixel][channel]: the other channel blocks' waves need it as the A operand of P2
//   P2  dz_lin[:, cb] = dlin Wglu[:, cb] (register-resident fragments of Wglu^T), dz = dz_lin + gate path -> LDS tile -> HBM
//       with 16-byte coalesced stores; sums of dz and dz * xhat for the BatchNorm backward
//   P3  dWy[cb][c] += sum_p dlin[p][cb] y[p][c] for every channel block c: BOTH operands come from accumulator-layout
//       registers - a lane of the D layout holds 8 pixels of one channel per k-step, which is exactly an A (dlin) or B (y)
//       fragment of the transposed product once both sides use the same pixel order; the other blocks' y again through
//       identity MFMAs.  No transposed LDS images (gglu.hip wrote four of them per round).
// The weight gradient is accumulated against the RAW y (exact in bf16) and converted at the end:
//   dWx[co][c] = invstd[c] (dWy[co][c] - mean[c] sum_p dlin[p][co])        (xhat = (y - mean) invstd)
// Per-workgroup partial sums in gglu.hip's layout [C * C dWx | C sdb | C sdz | C sdzx]: k_gpart_reduce / k_gbn_bwd_prep follow.
// Geometry of the backward kernel: C = 64 - 4 waves (2 row blocks x 2 channel blocks), weight fragments in registers;
// C = 128 - 8 waves (2 row blocks x 4 channel blocks, two per SIMD) with the two weight matrices as padded bf16 images in LDS:
// with them in registers (64) next to the 64 accumulators of dW the kernel needed 342 registers = one wave per SIMD, and a
// wave that waits on a barrier, an LDS transpose or a dependent MFMA chain then idles its SIMD (131 us per launch).
template <int C>
struct BGluBwdCfg {
    static constexpr int WL = (C == 128) ? 1 : 0;                       // weights in LDS
    static constexpr int NW = WL ? 8 : 4, NT = 64 * NW;
    static constexpr int NB = C / 32, RPR = NW / NB;
    static constexpr int WROW = 2 * C + 16;                             // weight image row stride (bytes)
    static constexpr int WBYTES = WL ? 2 * C * WROW : 16;
    static constexpr int NITEM = RPR * 32 * (C / 8) / NT;               // 16-byte staging items per thread
};

template <int C, int PB>
__global__ __launch_bounds__(BGluBwdCfg<C>::NT) void k_bglu_bwd(const __bf16* __restrict__ y, const float* __restrict__ bn,
                                                   const __bf16* __restrict__ wfold, const float* __restrict__ bfold,
                                                   const __bf16* __restrict__ wgT,
                                                   const void* __restrict__ dp_v, const float* __restrict__ dp2, __bf16* __restrict__ dz,
                                                   float* __restrict__ part, int H, int W, int Ho, int Wo, int Q, int use_drop,
                                                   float p_drop, const uint16_t* __restrict__ mask_in) {
    using Cfg = BGluCfg<C>;
    using BC = BGluBwdCfg<C>;
    using PT = typename Stor<PB>::T;
    constexpr int NB = Cfg::NB, KS = Cfg::KS, PS = Cfg::PS, TILE = Cfg::TILE;
    constexpr int RPR = BC::RPR, NT = BC::NT, WL = BC::WL, WROW = BC::WROW, NI = BC::NITEM;
    // one LDS arena: y tiles (double-buffered) | dlin tiles | dz tiles | [W' | Wglu^T] images (WL only); the first three are
    // contiguous on purpose - the C x C floats of the final dW exchange are laid over them
    __shared__ __attribute__((aligned(16))) unsigned char sm[4 * RPR * TILE + BC::WBYTES];
    __shared__ float aff[2 * C];
    __shared__ float red[RPR][3][C];
    auto tileY = [&](int buf, int r2) { return sm + (buf * RPR + r2) * TILE; };
    auto tileD = [&](int r2) { return sm + (2 * RPR + r2) * TILE; };
    auto tileZ = [&](int r2) { return sm + (3 * RPR + r2) * TILE; };
    unsigned char* wimg = sm + 4 * RPR * TILE;
    const PT* dp = (const PT*)dp_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, kh = lane >> 5;
    const int rl = wv / NB, cb = wv % NB, c = 32 * cb + n;
    TSC(0);
    if (H & 1) {        // the floor-mode pool drops the last row of an odd-height image: its gradient is 0
        const int per_clip = W * C, nbt = Q / (Ho * Wo);
        for (int i = blockIdx.x * NT + tid; i < nbt * per_clip; i += gridDim.x * NT) {
            const int bb = i / per_clip, r = i % per_clip;
            dz[((size_t)bb * H + (H - 1)) * W * C + r] = (__bf16)0.f;
        }
    }
    if (tid < C) { aff[tid] = bn[2 * C + tid]; aff[C + tid] = bn[3 * C + tid]; }       // scale, shift (written by the forward)
    __syncthreads();
    bf16x8 bw[KS], bt[KS], idf[2];                                       // (weight fragments: registers unless WL)
    float b_fold, sc, sh;
    // W' [C][C] and b' [C] were published by the forward kernel, Wglu^T [c][co] (raw) by the forward's packing pass
    b_fold = bfold[c];
    sc = aff[c]; sh = aff[C + c];
    if constexpr (WL != 0) {
        // both matrices as padded bf16 images in LDS: 2 x C x C / 8 16-byte vectors
        constexpr int NV = 2 * C * C / 8 / NT;
        f32x4 wv4[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + NT * i, which = v / (C * C / 8), r = (v / (C / 8)) % C, j = v % (C / 8);
            wv4[i] = *(const f32x4*)((which ? wgT : wfold) + (size_t)r * C + 8 * j);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + NT * i, which = v / (C * C / 8), r = (v / (C / 8)) % C, j = v % (C / 8);
            *(f32x4*)(wimg + (which * C + r) * WROW + 16 * j) = wv4[i];
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bw[ks] = *(const bf16x8*)(wfold + (size_t)c * C + 16 * ks + 8 * kh);
            bt[ks] = *(const bf16x8*)(wgT + (size_t)c * C + 16 * ks + 8 * kh);      // B[k = co][j = c] = Wglu[co][c]
        }
    }
    bglu_identity(lane, idf);
    const unsigned char* wp1 = wimg + c * WROW + 16 * kh;               // this lane's row of W' / Wglu^T (WL)
    const unsigned char* wp2 = wimg + (C + c) * WROW + 16 * kh;
    const float mean = bn[c], invstd = bn[C + c];
    const float sc2 = sc * SED_NEG_LOG2E, sh2 = sh * SED_NEG_LOG2E;
    const float scp = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    f32x16 dW[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[b][r] = 0.f;
    float sdb = 0.f, sdz = 0.f, sdzx = 0.f;
    const int n_rb = (Q + 3) / 4, n_round = (n_rb + RPR - 1) / RPR;
    f32x4 st[NI];
    auto load = [&](int round) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + NT * i, r2, m, j);
            const int rb = round * RPR + r2, q = rb * 4 + (m >> 3);
            const bool ok = round < n_round && q < Q;
            const int pix = gen_rb_pixel(ok ? q : 0, (m >> 2) & 1, m & 3, H, W, Ho, Wo);
            st[i] = *(const f32x4*)(y + (size_t)pix * C + 8 * j);
            if (!ok) st[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + NT * i, r2, m, j);
            *(f32x4*)(tileY(buf, r2) + m * PS + 16 * j) = st[i];
        }
    };
    // UNCONDITIONAL loads from clamped addresses (a load under a per-element condition becomes a branch with its own
    // s_waitcnt: four serialized memory round trips per round in the first version), validity applied at the use
    float gqn[4], gq2[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t mkn = 0xffffu;
    auto aux_load = [&](int round) {
        const int rbn = min(round * RPR + rl, n_rb - 1);
        size_t e[4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) e[jx] = (size_t)min(rbn * 4 + jx, Q - 1) * C + c;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) gqn[jx] = ld1(dp + e[jx]);
        if (dp2 != nullptr) {          // ONE uniform branch around all four loads of the second plane (a select per element
                                       // made four branches, each waiting out its own load - and the tile prefetch with it)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) gq2[jx] = dp2[e[jx]];
        }
        if (use_drop) mkn = (uint32_t)mask_in[((size_t)rbn * NB + cb) * 64 + lane];
    };
    TSC(1);
    load(blockIdx.x);
    aux_load(blockIdx.x);
    int it = 0;
    for (int round = blockIdx.x; round < n_round; round += gridDim.x, ++it) {
        const int buf = it & 1;
        store(buf);
        const int rb = round * RPR + rl, q0 = rb * 4;
        const bool live = rb < n_rb;
        // upstream gradient of the 4 pooled pixels and the keep bits of this wave's channels: fetched a round ahead (aux_load)
        float gq[4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) gq[jx] = (live && q0 + jx < Q) ? (gqn[jx] + gq2[jx]) * scp : 0.f;
        const uint32_t mk = (use_drop && live) ? mkn : 0xffffu;
        if (it == 1) TSC(2);
        lds_barrier();                                                   // B1: tileY[buf] staged (LDS-only barrier: __syncthreads()
                                                                         // would also drain the prefetch loads / dz stores in flight)
        if (it == 1) TSC(3);
        load(round + gridDim.x);
        aux_load(round + gridDim.x);
        const unsigned char* ty = tileY(buf, rl) + n * PS + 16 * kh;
        // ---- P1 ----------------------------------------------------------------------------------------------------------
        f32x16 lin, yid;
#pragma unroll
        for (int r = 0; r < 16; ++r) { lin[r] = 0.f; yid[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            lin = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ty + 32 * ks), WL ? *(const bf16x8*)(wp1 + 32 * ks) : bw[ks],
                                                          lin, 0, 0, 0);
        yid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ty + 64 * cb), idf[0], yid, 0, 0, 0);
        yid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ty + 64 * cb + 32), idf[1], yid, 0, 0, 0);
        float dzg[16];
        bf16x8 a3[2], yb[2];
        __bf16* td = (__bf16*)tileD(rl) + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sg = sigmoid_from_scaled(fmaf(sc2, yid[r], sh2));
            const float gg = ((mk >> r) & 1u) ? gq[r >> 2] : 0.f;
            const float dl = gg * sg;
            sdb += dl;
            dzg[r] = dl * (1.0f - sg) * (lin[r] + b_fold);
            const __bf16 dlb = (__bf16)dl;
            a3[r >> 3][r & 7] = dlb;
            yb[r >> 3][r & 7] = (__bf16)yid[r];                         // exact: y is stored as bf16
            td[(size_t)mfma32_row(r, lane) * (PS / 2)] = dlb;
        }
        if (it == 1) TSC(4);
        lds_barrier();                                                   // B2: every channel block's dlin is in tileD
        if (it == 1) TSC(5);
        // ---- P2 ----------------------------------------------------------------------------------------------------------
        f32x16 dzl;
#pragma unroll
        for (int r = 0; r < 16; ++r) dzl[r] = 0.f;
        {
            const unsigned char* tdr = tileD(rl) + n * PS + 16 * kh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                dzl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tdr + 32 * ks), WL ? *(const bf16x8*)(wp2 + 32 * ks) : bt[ks],
                                                              dzl, 0, 0, 0);
        }
        __bf16* tz = (__bf16*)tileZ(rl) + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = live ? dzl[r] + dzg[r] : 0.f;
            sdz += v;
            sdzx = fmaf(v, ((float)yb[r >> 3][r & 7] - mean) * invstd, sdzx);
            tz[(size_t)mfma32_row(r, lane) * (PS / 2)] = (__bf16)v;
        }
        if (it == 1) TSC(6);
        // ---- P3 ----------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            bf16x8 b3[2] = {yb[0], yb[1]};
            if (b != cb) {                                              // (wave-uniform)
                f32x16 yo;
#pragma unroll
                for (int r = 0; r < 16; ++r) yo[r] = 0.f;
                yo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ty + 64 * b), idf[0], yo, 0, 0, 0);
                yo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ty + 64 * b + 32), idf[1], yo, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) b3[r >> 3][r & 7] = (__bf16)yo[r];
            }
            dW[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0], dW[b], 0, 0, 0);
            dW[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1], dW[b], 0, 0, 0);
        }
        if (it == 1) TSC(7);
        lds_barrier();                                                   // B3: tileZ complete
        if (it == 1) TSC(8);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int r2, m, j;
            bglu_item<C>(tid + NT * i, r2, m, j);
            const int rb2 = round * RPR + r2, q = rb2 * 4 + (m >> 3);
            if (q < Q) {
                const int pix = gen_rb_pixel(q, (m >> 2) & 1, m & 3, H, W, Ho, Wo);
                *(f32x4*)(dz + (size_t)pix * C + 8 * j) = *(const f32x4*)(tileZ(r2) + m * PS + 16 * j);
            }
        }
        if (it == 1) TSC(9);
    }
    TSC(10);
    // ---- per-workgroup partials ------------------------------------------------------------------------------------------
    __syncthreads();
    {
        const float v0 = sdb + __shfl_xor(sdb, 32), v1 = sdz + __shfl_xor(sdz, 32), v2 = sdzx + __shfl_xor(sdzx, 32);
        if (kh == 0) { red[rl][0][c] = v0; red[rl][1][c] = v1; red[rl][2][c] = v2; }
    }
    __syncthreads();
    float* ps = part + (size_t)blockIdx.x * (C * C + 3 * C);
    for (int e = tid; e < 3 * C; e += NT) {
        float v = 0.f;
#pragma unroll
        for (int r2 = 0; r2 < RPR; ++r2) v += (&red[r2][0][0])[e];
        ps[C * C + e] = v;
    }
    // dWx[co][c'] = invstd[c'] (dWy - mean[c'] sdb_wave[co]); D layout: lane = column c', register r = row co
    float* dws = (float*)sm;                               // C x C floats: the row blocks' waves add their shares in turn
    for (int pass = 0; pass < RPR; ++pass) {
        __syncthreads();
        if (rl == pass) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int cc = 32 * b + n;
                const float is = bn[C + cc], mu = bn[cc];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = 32 * cb + mfma32_row(r, lane);
                    const float v = is * (dW[b][r] - mu * red[rl][0][co]);
                    float* d = (RPR == 1) ? &ps[(size_t)co * C + cc] : &dws[co * C + cc];
                    if (RPR == 1 || pass == 0) *d = v;
                    else *d += v;
                }
            }
        }
    }
    if (RPR > 1) {
        __syncthreads();
        for (int e = tid; e < C * C; e += NT) ps[e] = dws[e];
    }
    TSC(11);
}

int bglu_bwd_grid(int C, int B, int H, int W) {
    const int Q = B * (H / 2) * (W / 4), n_rb = (Q + 3) / 4, rpr = (C == 64) ? BGluBwdCfg<64>::RPR : BGluBwdCfg<128>::RPR;
    const int n_round = (n_rb + rpr - 1) / rpr;
    const int cap = (C == 128) ? 256 : 512;      // C = 128: 140 KB of LDS = one workgroup per CU, which then pays its prologue / partial slab once
    return n_round < cap ? n_round : cap;
}

template <int C, int PB>
static int bglu_bwd_launch(const void* y, const float* bn, const void* wfold, const float* bfold, const void* wgT,
                           const void* dp, const float* dp2, void* dz, float* part, int B, int H, int W, int use_drop, float p_drop,
                           const uint16_t* mask_in, hipStream_t st) {
    // the final dW exchange (C x C floats) reuses the y tiles and, at C = 128, the dlin / dz tiles behind them
    static_assert((size_t)C * C * 4 <= (size_t)4 * BGluBwdCfg<C>::RPR * BGluCfg<C>::TILE, "dW exchange does not fit the tiles");
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo;
    k_bglu_bwd<C, PB><<<bglu_bwd_grid(C, B, H, W), BGluBwdCfg<C>::NT, 0, st>>>((const __bf16*)y, bn, (const __bf16*)wfold, bfold, (const __bf16*)wgT, dp, dp2, (__bf16*)dz, part, H, W,
                                                                 Ho, Wo, Q, use_drop, p_drop, mask_in);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_bglu_bwd(int C, const void* y, const float* bn, const void* wfold, const float* bfold, const void* wgT,
                    const void* dp, int dp_bf16, void* dz, float* part, int B, int H, int W, int use_drop, float p_drop,
                    const uint16_t* mask_in, hipStream_t st, const float* dp2) {
#define BGLU_CASE(CC, PP) \
    if (C == CC && dp_bf16 == PP) return bglu_bwd_launch<CC, PP>(y, bn, wfold, bfold, wgT, dp, dp2, dz, part, B, H, W, use_drop, p_drop, mask_in, st)
    BGLU_CASE(64, 0); BGLU_CASE(64, 1); BGLU_CASE(128, 0); BGLU_CASE(128, 1);
#undef BGLU_CASE
    sed_set_error("bglu backward: unsupported channels %d", C);
    return SED_ERR_UNSUPPORTED;
}
