// gglu.hip - BatchNorm-apply + GLU + dropout + AvgPool2d((2,4)) for conv blocks 1 and 2, generic in the channel count
// (C in {64, 128}) and the MFMA operand type (gen.h), forward and backward.
//
// Reference ops (baseline/models/CNN.py:49-67, GLU CNN.py:11-16):
//   z = BatchNorm2d(y);  lin = Linear(C, C)(z over channels);  out = lin * sigmoid(z);  p = AvgPool2d((2,4))(Dropout(out))
// The kernels work on xhat = (y - mean) * invstd: with wg = Wglu diag(gamma) and bg = bglu + Wglu beta (k_gen_pack)
//   lin = wg xhat + bg,   z = gamma xhat + beta,
// so the LDS tile holds ONE fp32 quantity that is the MFMA operand (fp32 mode), the source of the gate, and the factor
// of the BatchNorm-backward sum (sum dz * xhat) - y itself is not needed again.
// Row block = 4 consecutive pooled pixels = 32 input pixels ordered as in bnglu.hip (MFMA row m: pooled pixel m >> 3,
// dt = (m >> 2) & 1, df = m & 3), so that a 2x4 pooling window is 4 registers of a lane pair.
#include "gen.h"
#include "kernels.h"
#include "gkernels.h"

// BatchNorm statistics -> mean / invstd (+ running statistics, published by workgroup 0), C threads
__device__ __forceinline__ void gbn_prep(const GBnArgs& a, int C, int c, bool publish, float* bn_s /* LDS [2][C] */) {
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

// Staging of one row block (32 pixels x C channels) as xhat into xf (fp32) and, in bf16 mode, xb; one wave.  Split into
// "issue all loads" and "normalise + store to LDS": a load consumed right after its issue exposes a full memory round
// trip per item on these one-workgroup-per-CU kernels (16 items per lane: ~30 us per round in the first version); the
// forward kernel issues the NEXT round's loads before the current round's MFMAs.
template <int C>
struct GGluTile { f32x4 v[32 * (C / 4) / 64]; };
template <int C, class YT>
__device__ __forceinline__ void gglu_load(GGluTile<C>& t, const YT* __restrict__ y, int q0, int Q, int H, int W, int Ho, int Wo,
                                          int lane) {
    constexpr int C4 = C / 4;
    int pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pb[j] = (q0 + j < Q) ? gen_rb_pixel(q0 + j, 0, 0, H, W, Ho, Wo) : -1;
#pragma unroll
    for (int i = 0; i < 32 * C4 / 64; ++i) {
        const int g = lane + 64 * i, m = g / C4, c4 = g % C4;
        const int j = m >> 3, dt = (m >> 2) & 1, df = m & 3;
        const int base = (j == 0) ? pb[0] : (j == 1) ? pb[1] : (j == 2) ? pb[2] : pb[3];
        // invalid pooled pixels (only past the end of the last row block) read pixel 0 and are zeroed at the store
        const size_t off = (size_t)((base >= 0 ? base : 0) + dt * W + df) * C + 4 * c4;
        t.v[i] = ld4(y + off);
        if (base < 0) t.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
template <int MODE, int C>
__device__ __forceinline__ void gglu_store(const GGluTile<C>& t, const float* bn_s, float* xf, typename MM<MODE>::E* xb, int q0, int Q,
                                           int lane) {
    using M = MM<MODE>;
    constexpr int C4 = C / 4, XS = C + 1, BS = C + M::PAD;
#pragma unroll
    for (int i = 0; i < 32 * C4 / 64; ++i) {
        const int g = lane + 64 * i, m = g / C4, c4 = g % C4;
        const bool ok = q0 + (m >> 3) < Q;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ok ? (t.v[i][q] - bn_s[4 * c4 + q]) * bn_s[C + 4 * c4 + q] : 0.f;
        float* d = xf + m * XS + 4 * c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        if (MODE == 1) M::st4(xb + m * BS + 4 * c4, v[0], v[1], v[2], v[3]);
    }
}

template <int MODE, int C>
struct GGluFwdCfg {
    using M = MM<MODE>;
    static constexpr int XS = C + 1, BS = C + M::PAD;
    static constexpr size_t XF_BYTES = (size_t)4 * 32 * XS * 4;
    static constexpr size_t XB_BYTES = MODE == 1 ? (size_t)4 * 32 * BS * 2 : 0;
    static constexpr size_t WBUF_BYTES = (size_t)2 * C * (M::KC + M::PAD) * sizeof(typename M::E);
    static constexpr size_t LDS_BYTES = XF_BYTES + XB_BYTES + WBUF_BYTES + 2 * C * 4 + 64;
};

// PB: the pooled output is stored as bf16 (SED_DTYPE_BF16, block 1; block 2's output p2 feeds the fp32 GRU)
template <int MODE, int C, int PB>
__global__ __launch_bounds__(256) void k_gglu_fwd(const void* __restrict__ y_v, GBnArgs bnp, const void* __restrict__ wg_v,
                                                   const float* __restrict__ bg, void* __restrict__ p_v, int H, int W, int Ho,
                                                   int Wo, int Q, int block_id, int use_drop, float p_drop,
                                                   const uint64_t* __restrict__ seed_ptr, uint16_t* __restrict__ mask_out) {
    using Cfg = GGluFwdCfg<MODE, C>;
    using M = MM<MODE>;
    using E = typename M::E;
    using YT = typename Stor<MODE == 1>::T;
    using PT = typename Stor<PB>::T;
    const YT* y = (const YT*)y_v;
    PT* p = (PT*)p_v;
    constexpr int NB = C / 32, XS = Cfg::XS, BS = Cfg::BS;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    float* xf_all = (float*)gsm;
    E* xb_all = (E*)(gsm + Cfg::XF_BYTES);
    E* wbuf = (E*)(gsm + Cfg::XF_BYTES + Cfg::XB_BYTES);
    float* bn_s = (float*)(gsm + Cfg::XF_BYTES + Cfg::XB_BYTES + Cfg::WBUF_BYTES);
    const E* wg = (const E*)wg_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    if (tid < C) gbn_prep(bnp, C, tid, blockIdx.x == 0, bn_s);
    __syncthreads();
    float* xf = xf_all + wv * 32 * XS;
    E* xb = xb_all + wv * 32 * BS;
    float gam[NB], bet[NB], bgl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { gam[nb] = bnp.gamma[32 * nb + n]; bet[nb] = bnp.beta[32 * nb + n]; bgl[nb] = bg[32 * nb + n]; }
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float sc = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    const int n_rb = (Q + 3) / 4;
    const int rounds = (n_rb + gridDim.x * 4 - 1) / (gridDim.x * 4);
    GGluTile<C> yt;
    {
        const int rb0 = blockIdx.x * 4 + wv;
        gglu_load<C, YT>(yt, y, (rb0 < n_rb ? rb0 : 0) * 4, Q, H, W, Ho, Wo, lane);
    }
    for (int round = 0; round < rounds; ++round) {
        const int rb = (round * gridDim.x + blockIdx.x) * 4 + wv;
        const bool live = rb < n_rb;
        const int q0 = rb * 4;
        if (live) gglu_store<MODE, C>(yt, bn_s, xf, xb, q0, Q, lane);
        {   // the next round's tile flies during this round's MFMAs and epilogue (past the end: row block 0, never used)
            const int rbn = ((round + 1) * gridDim.x + blockIdx.x) * 4 + wv;
            gglu_load<C, YT>(yt, y, (rbn < n_rb ? rbn : 0) * 4, Q, H, W, Ho, Wo, lane);
        }
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const E* a_row = (MODE == 1) ? (const E*)(xb + n * BS) : (const E*)(xf + n * XS);
        stream_gemm<MODE, NB, NB, M::KC>(a_row, [](int ch) { return ch * M::KC; }, wg, C, C, wbuf, acc, 0, tid);
        if (live) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int c = 32 * nb + n;
                uint32_t m16 = 0xffffu;
                if (use_drop) {
                    m16 = gen_keep16(rb, nb, C, lane, block_id, seed, thr);
                    if (mask_out) mask_out[((size_t)rb * NB + nb) * 64 + lane] = (uint16_t)m16;
                }
                float pooled[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xh = xf[mfma32_row(r, lane) * XS + c];
                    const float v = (acc[nb][r] + bgl[nb]) * sigmoidf_fast(fmaf(gam[nb], xh, bet[nb]));
                    pooled[r >> 2] += ((m16 >> r) & 1u) ? v : 0.f;
                }
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) pooled[jx] += __shfl_xor(pooled[jx], 32);
                const int j0 = 2 * kh;
                if (q0 + j0 < Q) st1(p + (size_t)(q0 + j0) * C + c, (kh ? pooled[2] : pooled[0]) * sc);
                if (q0 + j0 + 1 < Q) st1(p + (size_t)(q0 + j0 + 1) * C + c, (kh ? pooled[3] : pooled[1]) * sc);
            }
        }
        // (the tile is rewritten by this same wave in the next round; other waves only share wbuf, which stream_gemm guards)
    }
}

template <int MODE, int C, int PB>
static int gglu_fwd_launch(const void* y, const GBnArgs& bn, const void* wg, const float* bg, void* p, int B, int H, int W,
                           int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st) {
    using Cfg = GGluFwdCfg<MODE, C>;
    static thread_local SedAttrOnce attr;
    if (attr.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gglu_fwd<MODE, C, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
    }
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo, n_rb = (Q + 3) / 4;
    int grid = (n_rb + 3) / 4;
    if (grid > 512) grid = 512;
    k_gglu_fwd<MODE, C, PB><<<grid, 256, Cfg::LDS_BYTES, st>>>(y, bn, wg, bg, p, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, seed, mask_out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_gglu_fwd(int mode, int C, const void* y, const GBnArgs& bn, const void* wg, const float* bg, void* p, int p_bf16, int B,
                    int H, int W, int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st) {
#define GGLU_CASE(MD, CC, PP) \
    if (mode == MD && C == CC && p_bf16 == PP) return gglu_fwd_launch<MD, CC, PP>(y, bn, wg, bg, p, B, H, W, block_id, use_drop, p_drop, seed, mask_out, st)
    GGLU_CASE(0, 64, 0); GGLU_CASE(0, 128, 0); GGLU_CASE(1, 64, 0); GGLU_CASE(1, 128, 0); GGLU_CASE(1, 64, 1); GGLU_CASE(1, 128, 1);
#undef GGLU_CASE
    sed_set_error("gglu forward: unsupported mode %d / channels %d", mode, C);
    return SED_ERR_UNSUPPORTED;
}

// ---- backward ----------------------------------------------------------------------------------------------------------
// A workgroup round covers TWO row blocks (64 pixels); the four waves are (row block g = wave >> 1) x (channel half
// hf = wave & 1): wherever a channel is an OUTPUT index (lin / dz columns) a wave owns C / 2 of them, which halves the
// accumulators and - above all - the LDS tiles (fp32 xhat + fp32 dlin (+ their bf16 operand copies) for 4 row blocks
// would not fit next to the weight chunks at C = 128).
//   P1  lin[:, half] = xhat @ wg^T (recomputed)        epilogue: sigma, dlin = g sigma -> LDS, gate path dzg
//   P2  dzl[:, half] = dlin @ Wglu                      epilogue: dz = dzl + dzg -> HBM, sums of dz and dz * xhat
//   P3  dWx[co][c] += sum_p dlin[p][co] xhat[p][c]      contraction over the round's 64 PIXELS; wave w owns (C / 32)^2 / 4
//       tiles of 32 x 32.  fp32 mode: f32 MFMA on the natural [pixel][channel] tiles.  bf16 mode: both operands are
//       needed pixel-contiguous - the P1 epilogue holds 4 consecutive pixels of one channel per lane (D layout), so it
//       writes dlinT / xhatT [channel][pixel] as 8-byte bf16 groups for free (the fp32 P3 was 8 192 of the 10 240 MFMA
//       cycles of a round)
// Per-workgroup partial sums (dWx, sum dlin, sum dz, sum dz xhat) go to `part`; k_gbn_bwd_prep adds them in fixed order.
template <int MODE, int C>
struct GGluBwdCfg {
    using M = MM<MODE>;
    static constexpr int KC = (MODE == 0 && C == 128) ? 16 : M::KC;        // fp32 at C = 128: smaller weight chunks, LDS budget
    static constexpr int XS = C + 1, BS = C + M::PAD;
    static constexpr int TS = 64 + 8;                                      // bf16 mode: row stride of the pixel-contiguous tiles
    static constexpr size_t XF_BYTES = (size_t)2 * 32 * XS * 4;            // xhat fp32, 2 row blocks
    static constexpr size_t DF_BYTES = MODE == 0 ? XF_BYTES : 0;           // dlin fp32 (P3 operand of the fp32 mode)
    static constexpr size_t XB_BYTES = MODE == 1 ? (size_t)2 * 32 * BS * 2 : 0;
    static constexpr size_t DB_BYTES = XB_BYTES;
    static constexpr size_t TT_BYTES = MODE == 1 ? (size_t)2 * C * TS * 2 : 0;    // dlinT | xhatT: [C][64 pixels] bf16 (P3 operands)
    static constexpr size_t WBUF_BYTES = (size_t)2 * C * (KC + M::PAD) * sizeof(typename M::E);
    static constexpr size_t LDS_BYTES = XF_BYTES + DF_BYTES + XB_BYTES + DB_BYTES + TT_BYTES + WBUF_BYTES + 2 * C * 4 + 64;
};

// PB: dp (the gradient w.r.t. the pooled output) arrives as bf16 (SED_DTYPE_BF16, block 1: written by block 2's dgrad);
// y is read and dz written in the mode's storage type
template <int MODE, int C, int PB>
__global__ __launch_bounds__(256) void k_gglu_bwd(const void* __restrict__ y_v, const float* __restrict__ bn,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const void* __restrict__ wg_v, const void* __restrict__ wgT_v,
                                                   const float* __restrict__ bg, const void* __restrict__ dp_v, const float* __restrict__ dp2,
                                                   void* __restrict__ dz_v, float* __restrict__ part, int H, int W, int Ho, int Wo,
                                                   int Q, int use_drop, float p_drop, const uint16_t* __restrict__ mask_in) {
    using Cfg = GGluBwdCfg<MODE, C>;
    using M = MM<MODE>;
    using E = typename M::E;
    using YT = typename Stor<MODE == 1>::T;
    using PT = typename Stor<PB>::T;
    const YT* y = (const YT*)y_v;
    YT* dz = (YT*)dz_v;
    const PT* dp = (const PT*)dp_v;
    constexpr int NB = C / 32, NBW = NB / 2, XS = Cfg::XS, BS = Cfg::BS, KC = Cfg::KC;
    constexpr int TPW = NB * NB / 4;                       // dWx tiles (32 x 32) per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    float* xf_all = (float*)gsm;
    float* df_all = (float*)(gsm + Cfg::XF_BYTES);
    E* xb_all = (E*)(gsm + Cfg::XF_BYTES + Cfg::DF_BYTES);
    E* db_all = (E*)(gsm + Cfg::XF_BYTES + Cfg::DF_BYTES + Cfg::XB_BYTES);
    __bf16* dlT = (__bf16*)(gsm + Cfg::XF_BYTES + Cfg::DF_BYTES + Cfg::XB_BYTES + Cfg::DB_BYTES);
    __bf16* xhT = dlT + C * Cfg::TS;
    E* wbuf = (E*)(gsm + Cfg::XF_BYTES + Cfg::DF_BYTES + Cfg::XB_BYTES + Cfg::DB_BYTES + Cfg::TT_BYTES);
    constexpr int TS = Cfg::TS;
    float* bn_s = (float*)((unsigned char*)wbuf + Cfg::WBUF_BYTES);
    const E* wg = (const E*)wg_v;
    const E* wgT = (const E*)wgT_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int g = wv >> 1, hf = wv & 1, nb0 = hf * NBW;
    if (H & 1) {        // the floor-mode pool drops the last row of an odd-height image: its gradient is 0
        const int per_clip = W * C, nbt = Q / (Ho * Wo);
        for (int i = blockIdx.x * 256 + tid; i < nbt * per_clip; i += gridDim.x * 256) {
            const int bb = i / per_clip, r = i % per_clip;
            st1(dz + ((size_t)bb * H + (H - 1)) * W * C + r, 0.f);
        }
    }
    if (tid < C) { bn_s[tid] = bn[tid]; bn_s[C + tid] = bn[C + tid]; }       // mean, invstd
    __syncthreads();
    float* xf = xf_all + g * 32 * XS;
    float* dfl = df_all + g * 32 * XS;
    E* xb = xb_all + g * 32 * BS;
    E* dbl = db_all + g * 32 * BS;
    float gam[NBW], bet[NBW], bgl[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = 32 * (nb0 + nb) + n;
        gam[nb] = gamma[c]; bet[nb] = beta[c]; bgl[nb] = bg[c];
    }
    const float sc = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    f32x16 dW[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[t][r] = 0.f;
    float sdb[NBW], sdz[NBW], sdzx[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) { sdb[nb] = 0.f; sdz[nb] = 0.f; sdzx[nb] = 0.f; }
    // dWx tiles of this wave: C = 64: one tile (co block wv >> 1, c block wv & 1); C = 128: co block wv, c blocks 0..3
    const int cob = (C == 64) ? (wv >> 1) : wv;
    const int n_rb = (Q + 3) / 4;
    const int rounds = (n_rb + gridDim.x * 2 - 1) / (gridDim.x * 2);
    for (int round = 0; round < rounds; ++round) {
        const int rb = (round * gridDim.x + blockIdx.x) * 2 + g;
        const bool live = rb < n_rb;
        const int q0 = rb * 4;
        // ---- stage: each wave of the pair stages half of the row block's 32 rows (all channels); all loads first ----------
        {
            constexpr int C4 = C / 4, NI = 16 * C4 / 64;
            int pb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pb[j] = (live && q0 + j < Q) ? gen_rb_pixel(q0 + j, 0, 0, H, W, Ho, Wo) : -1;
            f32x4 yv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int gg = lane + 64 * i, m = 16 * hf + gg / C4, c4 = gg % C4;
                const int j = m >> 3, dt = (m >> 2) & 1, df = m & 3;
                const int base = (j == 0) ? pb[0] : (j == 1) ? pb[1] : (j == 2) ? pb[2] : pb[3];
                yv[i] = ld4(y + (size_t)((base >= 0 ? base : 0) + dt * W + df) * C + 4 * c4);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int gg = lane + 64 * i, m = 16 * hf + gg / C4, c4 = gg % C4;
                const int j = m >> 3;
                const int base = (j == 0) ? pb[0] : (j == 1) ? pb[1] : (j == 2) ? pb[2] : pb[3];
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (base >= 0) ? (yv[i][q] - bn_s[4 * c4 + q]) * bn_s[C + 4 * c4 + q] : 0.f;
                float* d = xf + m * XS + 4 * c4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
                if (MODE == 1) M::st4(xb + m * BS + 4 * c4, v[0], v[1], v[2], v[3]);
            }
        }
        // pooled gradients and keep bits of this wave's channels
        float gq[NBW][4];
        uint32_t mk[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int c = 32 * (nb0 + nb) + n;
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const size_t e = (size_t)(q0 + jx) * C + c;
                // dp2: the second direction plane of gru4.hip's dX (H = 64), added while loading
                gq[nb][jx] = (live && q0 + jx < Q) ? (dp2 ? ld1(dp + e) + dp2[e] : ld1(dp + e)) * sc : 0.f;
            }
            mk[nb] = (use_drop && live) ? (uint32_t)mask_in[((size_t)rb * NB + nb0 + nb) * 64 + lane] : 0xffffu;
        }
        // ---- P1: lin = xhat @ wg^T (this wave's channel half) -------------------------------------------------------------
        f32x16 acc[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        {
            const E* a_row = (MODE == 1) ? (const E*)(xb + n * BS) : (const E*)(xf + n * XS);
            stream_gemm<MODE, NB, NBW, KC>(a_row, [](int ch) { return ch * KC; }, wg, C, C, wbuf, acc, nb0, tid);
        }
        float dzg[NBW][16];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int c = 32 * (nb0 + nb) + n;
            float dl4[4], xh4[4];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, lane);
                const float xh = xf[row * XS + c];
                const float sg = sigmoidf_fast(fmaf(gam[nb], xh, bet[nb]));
                const float gg = ((mk[nb] >> r) & 1u) ? gq[nb][r >> 2] : 0.f;
                const float dl = gg * sg;
                if (MODE == 0) dfl[row * XS + c] = dl;
                if (MODE == 1) {
                    dbl[row * BS + c] = M::cvt(dl);
                    dl4[r & 3] = dl; xh4[r & 3] = xh;
                    if ((r & 3) == 3) {     // rows row - 3 .. row are 4 consecutive pixels of this row block: one 8-byte group
                        const int p0 = 32 * g + row - 3;
                        bf16x4 vd = {(__bf16)dl4[0], (__bf16)dl4[1], (__bf16)dl4[2], (__bf16)dl4[3]};
                        bf16x4 vx = {(__bf16)xh4[0], (__bf16)xh4[1], (__bf16)xh4[2], (__bf16)xh4[3]};
                        *(bf16x4*)(dlT + c * TS + p0) = vd;
                        *(bf16x4*)(xhT + c * TS + p0) = vx;
                    }
                }
                sdb[nb] += dl;
                dzg[nb][r] = dl * (1.0f - sg) * (acc[nb][r] + bgl[nb]);
            }
        }
        // ---- P2: dz_lin = dlin @ Wglu (this wave's channel half); the call's first barrier publishes both halves' dlin --
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        {
            const E* a_row = (MODE == 1) ? (const E*)(dbl + n * BS) : (const E*)(dfl + n * XS);
            stream_gemm<MODE, NB, NBW, KC>(a_row, [](int ch) { return ch * KC; }, wgT, C, C, wbuf, acc, nb0, tid);
        }
        if (live) {
            int pb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pb[j] = (q0 + j < Q) ? gen_rb_pixel(q0 + j, 0, 0, H, W, Ho, Wo) : -1;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int c = 32 * (nb0 + nb) + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = r >> 2, df = r & 3;
                    const int base = (j == 0) ? pb[0] : (j == 1) ? pb[1] : (j == 2) ? pb[2] : pb[3];
                    if (base >= 0) {
                        const float v = acc[nb][r] + dzg[nb][r];
                        st1(dz + (size_t)(base + kh * W + df) * C + c, v);
                        sdz[nb] += v;
                        sdzx[nb] += v * xf[mfma32_row(r, lane) * XS + c];
                    }
                }
            }
        }
        // ---- P3: dWx[co][c] += sum over the round's 64 pixels of dlin[p][co] xhat[p][c] ----------------------------------
        if (MODE == 1) {
            const __bf16* Ap = dlT + (32 * cob + n) * TS + 8 * kh;       // A[i = co][k = p .. p + 7]
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = *(const bf16x8*)(Ap + 16 * ks);
                if (C == 64) {
                    dW[0] = MM<1>::mma(a, *(const bf16x8*)(xhT + (32 * (wv & 1) + n) * TS + 16 * ks + 8 * kh), dW[0]);
                } else {
#pragma unroll
                    for (int t = 0; t < TPW; ++t) dW[t] = MM<1>::mma(a, *(const bf16x8*)(xhT + (32 * t + n) * TS + 16 * ks + 8 * kh), dW[t]);
                }
            }
        } else {
            const float* Ap = df_all + 32 * cob + n;          // A[i = co][k = p]
#pragma unroll 4
            for (int s = 0; s < 32; ++s) {
                const int pix = 2 * s + kh;                    // 0..63: row block pix >> 5, row pix & 31 -> contiguous tiles
                const float a = Ap[pix * XS];
                const float* bx = xf_all + pix * XS + n;
                if (C == 64) {
                    dW[0] = mfma32(a, bx[32 * (wv & 1)], dW[0]);
                } else {
#pragma unroll
                    for (int t = 0; t < TPW; ++t) dW[t] = mfma32(a, bx[32 * t], dW[t]);
                }
            }
        }
        __syncthreads();                                       // tiles free for the next round
    }
    // ---- per-workgroup partials: [C * C dWx | C sdb | C sdz | C sdzx] ----------------------------------------------------
    float* ps = part + (size_t)blockIdx.x * (C * C + 3 * C);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int cb = (C == 64) ? (wv & 1) : t;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps[(size_t)(32 * cob + mfma32_row(r, lane)) * C + 32 * cb + n] = dW[t][r];
    }
    // the two waves with the same channel half (g = 0, 1) add their sums through LDS
    float* red = xf_all;                                        // [2 g][3][C]
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const float v0 = sdb[nb] + __shfl_xor(sdb[nb], 32), v1 = sdz[nb] + __shfl_xor(sdz[nb], 32),
                    v2 = sdzx[nb] + __shfl_xor(sdzx[nb], 32);
        if (kh == 0) {
            const int c = 32 * (nb0 + nb) + n;
            red[(g * 3 + 0) * C + c] = v0; red[(g * 3 + 1) * C + c] = v1; red[(g * 3 + 2) * C + c] = v2;
        }
    }
    __syncthreads();
    for (int e = tid; e < 3 * C; e += 256) ps[C * C + e] = red[e] + red[3 * C + e];
}

int gglu_bwd_grid(int B, int H, int W) {
    const int Q = B * (H / 2) * (W / 4), n_rb = (Q + 3) / 4;
    int grid = (n_rb + 1) / 2;
    return grid > 256 ? 256 : grid;
}

template <int MODE, int C, int PB>
static int gglu_bwd_launch(const void* y, const float* bn, const float* gamma, const float* beta, const void* wg, const void* wgT,
                           const float* bg, const void* dp, const float* dp2, void* dz, float* part, int B, int H, int W, int use_drop,
                           float p_drop, const uint16_t* mask_in, hipStream_t st) {
    using Cfg = GGluBwdCfg<MODE, C>;
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "GLU backward tiles exceed the LDS");
    static thread_local SedAttrOnce attr;
    if (attr.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gglu_bwd<MODE, C, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
    }
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo;
    k_gglu_bwd<MODE, C, PB><<<gglu_bwd_grid(B, H, W), 256, Cfg::LDS_BYTES, st>>>(y, bn, gamma, beta, wg, wgT, bg, dp, dp2, dz, part, H, W, Ho,
                                                                             Wo, Q, use_drop, p_drop, mask_in);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_gglu_bwd(int mode, int C, const void* y, const float* bn, const float* gamma, const float* beta, const void* wg,
                    const void* wgT, const float* bg, const void* dp, int dp_bf16, void* dz, float* part, int B, int H, int W,
                    int use_drop, float p_drop, const uint16_t* mask_in, hipStream_t st, const float* dp2) {
#define GGLU_CASE(MD, CC, PP) \
    if (mode == MD && C == CC && dp_bf16 == PP) return gglu_bwd_launch<MD, CC, PP>(y, bn, gamma, beta, wg, wgT, bg, dp, dp2, dz, part, B, H, W, use_drop, p_drop, mask_in, st)
    GGLU_CASE(0, 64, 0); GGLU_CASE(0, 128, 0); GGLU_CASE(1, 64, 0); GGLU_CASE(1, 128, 0); GGLU_CASE(1, 64, 1); GGLU_CASE(1, 128, 1);
#undef GGLU_CASE
    sed_set_error("gglu backward: unsupported mode %d / channels %d", mode, C);
    return SED_ERR_UNSUPPORTED;
}

// ---- BatchNorm-backward coefficients + the block's parameter gradients from the per-workgroup partials ----------------
// grid = C * C / 256 + 1 workgroups: the first C * C / 256 each finish 256 entries of dWglu, the last one the vectors.
//   g_wglu[co][c] = gamma[c] dWx[co][c] + beta[c] sdb[co]        (dlin^T z with z = gamma xhat + beta)
//   g_bglu = sdb;  g_beta = sum dz;  g_gamma = sum dz xhat;  conv bias: exactly 0 in front of a train-mode BatchNorm
//   dy = scale (dz - m1 - xhat m2) = ca dz + cb y + cc  for the conv dgrad / wgrad loaders
__global__ __launch_bounds__(256) void k_gbn_bwd_prep(GBnBwdArgs a) {
    const int C = a.C, tid = threadIdx.x, stride = C * C + 3 * C;
    const int nw = C * C / 256;
    if ((int)blockIdx.x < nw) {
        const int e = blockIdx.x * 256 + tid, co = e / C, c = e % C;
        double s = 0, sb = 0;
        for (int k = 0; k < a.n_part; ++k) {
            s += (double)a.part[(size_t)k * stride + e];
            sb += (double)a.part[(size_t)k * stride + C * C + co];
        }
        a.g_wglu[e] = (float)((double)a.gamma[c] * s + (double)a.beta[c] * sb);
        return;
    }
    for (int c = tid; c < C; c += 256) {
        double sb = 0, sz = 0, szx = 0;
        for (int k = 0; k < a.n_part; ++k) {
            const float* p = a.part + (size_t)k * stride + C * C;
            sb += (double)p[c]; sz += (double)p[C + c]; szx += (double)p[2 * C + c];
        }
        const double mean = a.bn[c], invstd = a.bn[C + c], scale = a.bn[2 * C + c];
        a.g_bglu[c] = (float)sb;
        a.g_beta[c] = (float)sz;
        a.g_gamma[c] = (float)szx;
        a.g_convb[c] = 0.f;
        const double m1 = sz / a.N, m2 = szx / a.N;
        a.coef[c] = (float)scale;
        a.coef[C + c] = (float)(-scale * m2 * invstd);
        a.coef[2 * C + c] = (float)(scale * (m2 * invstd * mean - m1));
    }
}

// First stage of the partial-sum reduction: the up-to-256 per-workgroup slabs of k_gglu_bwd are folded into GPART_SLICES
// slabs by the whole chip (slice s adds slabs s, s + S, s + 2S, ... in that fixed order, 8 loads in flight per thread);
// k_gbn_bwd_prep then adds the GPART_SLICES survivors.  One 65-workgroup kernel walking 256 slabs serially took 75 us -
// the slowest kernel of the bf16 step.
__global__ __launch_bounds__(256) void k_gpart_reduce(const float* __restrict__ part, int n_part, int n_el, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x, sl = blockIdx.y;
    if (e >= n_el) return;
    float s = 0.f;
    int k = sl;
    for (; k + 7 * GPART_SLICES < n_part; k += 8 * GPART_SLICES) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u * GPART_SLICES) * n_el + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < n_part; k += GPART_SLICES) s += part[(size_t)k * n_el + e];
    out[(size_t)sl * n_el + e] = s;
}

int launch_gbn_bwd_prep(const GBnBwdArgs& a0, hipStream_t st) {
    GBnBwdArgs a = a0;
    const int n_el = a.C * a.C + 3 * a.C;
    if (a.n_part > GPART_SLICES) {
        k_gpart_reduce<<<dim3((n_el + 255) / 256, GPART_SLICES), 256, 0, st>>>(a.part, a.n_part, n_el, a.part2);
        SED_CHECK_LAUNCH();
        a.part = a.part2;
        a.n_part = GPART_SLICES;
    }
    k_gbn_bwd_prep<<<a.C * a.C / 256 + 1, 256, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
