// bnglu.hip - BatchNorm-apply + GLU + dropout + AvgPool2d((2,4)) fused, forward and backward,
// for conv blocks 1 and 2.
//
// Reference ops (baseline/models/CNN.py:49-67, GLU CNN.py:11-16):
//   z = BatchNorm2d(y);  lin = Linear(64,64)(z over channels);  out = lin * sigmoid(z)
//   p = AvgPool2d((2,4))(Dropout(out))
//
// One wave owns a "row block" = 4 consecutive pooled pixels = 32 input pixels, ordered so that
// the 32x32x2 MFMA's D fragment leaves a whole 2x4 pooling window inside a lane pair: MFMA row
// m <-> pooled pixel j = m>>3, dt = (m>>2)&1, df = m&3; D register r of lane l holds
// (j = r>>2, dt = l>>5, df = r&3), so pooling is 3 adds in-lane plus one cross-half add.
// The GLU weight (64x64) lives in registers as MFMA B fragments for the whole kernel; the BN
// affine is applied while the y tile is staged into LDS (65-float pixel stride: conflict-free
// for lane = pixel A-fragment reads and for lane = channel epilogue reads).
//
// Backward recomputes z, lin and the dropout mask (Philox, no stored masks), then
//   dlin = g*sig(z);  dz = dlin @ Wglu + g*lin*sig'(z);  dWglu += dlin^T z   (all on the MFMA)
// and accumulates the BatchNorm-backward sums (sum dz, sum dz*y) so that the conv dgrad / wgrad
// kernels can form dy = ca*dz + cb*y + cc on the fly (conv.hip).
#include "common.h"
#include "philox.h"
#include "kernels.h"

#define ZS 65   // pixel stride of LDS tiles (floats)

struct BnPrepArgs {
    const double* stat; double N;
    const float *gamma, *beta;
    float *run_mean, *run_var; int64_t* tracked;
    int train, update; float eps, momentum;
    float* bn;
};
__global__ __launch_bounds__(64) void k_bn_prep(BnPrepArgs a) {
    const int c = threadIdx.x;
    double mean, var;
    if (a.train) {
        mean = a.stat[c] / a.N;
        var = a.stat[64 + c] / a.N - mean * mean;
        if (var < 0) var = 0;
        if (a.update) {
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
    a.bn[c] = (float)mean; a.bn[64 + c] = (float)invstd; a.bn[128 + c] = (float)scale;
    a.bn[192 + c] = (float)(a.beta[c] - mean * scale);
}

// pixel index (into [B][H][W]) of MFMA row m of the row block starting at pooled pixel q0
__device__ __forceinline__ int rb_pixel(int q, int dt, int df, int H, int W, int Ho, int Wo) {
    const int wo = q % Wo, t = q / Wo;
    const int ho = t % Ho, b = t / Ho;
    return (b * H + 2 * ho + dt) * W + 4 * wo + df;
}

// A row block's conv output y (32 pixels x 64 channels) goes global -> registers -> LDS in two separately
// callable halves, so that the loads of row block k+1 are issued BEFORE the MFMAs of row block k: these kernels
// run one or two waves per SIMD, and a load consumed right after its issue exposes a full memory round trip
// (~2 us per row block in the first version; MFMA pipe busy 21 % in k_glu_pool_bwd).
struct YTile { float4 v[8]; };
__device__ __forceinline__ void tile_load(YTile& t, const float* __restrict__ y, int q0, int Q, int H, int W, int Ho,
                                          int Wo, int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int m = (lane >> 4) + 4 * it, c4 = (lane & 15) * 4;
        const int q = q0 + (m >> 3);
        t.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < Q) {
            const int pix = rb_pixel(q, (m >> 2) & 1, m & 3, H, W, Ho, Wo);
            t.v[it] = *(const float4*)(y + (size_t)pix * 64 + c4);
        }
    }
}
template <bool KEEP_Y>
__device__ __forceinline__ void tile_store(const YTile& t, const float4& sc, const float4& sh, float* zt, float* yt,
                                           int q0, int Q, int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int m = (lane >> 4) + 4 * it, c4 = (lane & 15) * 4;
        const float4 v = t.v[it];
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + (m >> 3) < Q) {
            z.x = fmaf(v.x, sc.x, sh.x); z.y = fmaf(v.y, sc.y, sh.y);
            z.z = fmaf(v.z, sc.z, sh.z); z.w = fmaf(v.w, sc.w, sh.w);
        }
        float* d = zt + m * ZS + c4;
        d[0] = z.x; d[1] = z.y; d[2] = z.z; d[3] = z.w;
        if (KEEP_Y) {
            float* e = yt + m * ZS + c4;
            e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
        }
    }
}

__global__ __launch_bounds__(256) void k_glu_pool_fwd(const float* __restrict__ y, const float* __restrict__ bn,
                                                       const float* __restrict__ wglu, const float* __restrict__ bglu,
                                                       float* __restrict__ p, int H, int W, int Ho, int Wo, int Q,
                                                       int block_id, int use_drop, float p_drop,
                                                       const uint64_t* __restrict__ seed_ptr, uint16_t* __restrict__ mask_out) {
    __shared__ float zts[4][32 * ZS];
    __shared__ float WsT[64 * ZS];     // Wglu transposed [c][co], stride 65: coalesced global read, conflict-free both ways
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    float* zt = zts[wv];
    for (int e = tid; e < 4096; e += 256) WsT[(e & 63) * ZS + (e >> 6)] = wglu[e];
    __syncthreads();
    float bw[32][2];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        bw[s][0] = WsT[(2 * s + kh) * ZS + n];            // B[k = c][j = co] = Wglu[co][c]
        bw[s][1] = WsT[(2 * s + kh) * ZS + 32 + n];
    }
    const float bg[2] = {bglu[n], bglu[32 + n]};
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const bool one_bit = (thr == 128u);
    const float sc = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    const int n_rb = (Q + 3) / 4;
    const float4 bsc = *(const float4*)(bn + 128 + (lane & 15) * 4), bsh = *(const float4*)(bn + 192 + (lane & 15) * 4);
    YTile yt_n;
    if (blockIdx.x * 4 + wv < n_rb) tile_load(yt_n, y, (blockIdx.x * 4 + wv) * 4, Q, H, W, Ho, Wo, lane);
    for (int rb = blockIdx.x * 4 + wv; rb < n_rb; rb += gridDim.x * 4) {
        const int q0 = rb * 4;
        tile_store<false>(yt_n, bsc, bsh, zt, nullptr, q0, Q, lane);
        {
            const int rbn = rb + gridDim.x * 4;
            if (rbn < n_rb) tile_load(yt_n, y, rbn * 4, Q, H, W, Ho, Wo, lane);
        }
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        const float* A = zt + n * ZS + kh;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a = A[2 * s];
            acc[0] = mfma32(a, bw[s][0], acc[0]);
            acc[1] = mfma32(a, bw[s][1], acc[1]);
        }
        u32x4 o1 = {0u, 0u, 0u, 0u};
        if (use_drop && one_bit) o1 = philox_stream_1bit((uint32_t)rb, lane, block_id, seed);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 32 * h + n;
            float pooled[4] = {0.f, 0.f, 0.f, 0.f};
            uint32_t m16 = 0xffffu;
            if (use_drop) {
                if (one_bit) {
                    m16 = philox_field16(o1, 2 * (rb & 3) + h);
                } else {
                    const u32x4 o = philox_stream((uint32_t)(rb * 64 + c), (uint32_t)(2 * block_id + kh), seed);
                    m16 = philox_keep16(o, thr);
                }
                if (mask_out) mask_out[((size_t)rb * 2 + h) * 64 + lane] = (uint16_t)m16;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float zr = zt[mfma32_row(r, lane) * ZS + c];
                const float v = (acc[h][r] + bg[h]) * sigmoidf_fast(zr);
                pooled[r >> 2] += ((m16 >> r) & 1u) ? v : 0.f;
            }
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) pooled[jx] += __shfl_xor(pooled[jx], 32);
            const int j0 = 2 * kh;
            if (q0 + j0 < Q) p[(size_t)(q0 + j0) * 64 + c] = (kh ? pooled[2] : pooled[0]) * sc;
            if (q0 + j0 + 1 < Q) p[(size_t)(q0 + j0 + 1) * 64 + c] = (kh ? pooled[3] : pooled[1]) * sc;
        }
    }
}

// acc layout (doubles): [0,4096) dWglu[co][c]; [4096,4160) dbglu; [4160,4224) sum dz; [4224,4288) sum dz*y
#define GLUACC_N 4288

__global__ __launch_bounds__(256) void k_glu_pool_bwd(const float* __restrict__ y, const float* __restrict__ bn,
                                                       const float* __restrict__ wglu, const float* __restrict__ bglu,
                                                       const float* __restrict__ dp, float* __restrict__ dz,
                                                       double* __restrict__ accg, int H, int W, int Ho, int Wo, int Q,
                                                       int block_id, int use_drop, float p_drop,
                                                       const uint16_t* __restrict__ mask_in, int no_atomic) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    float* WsT = smem + 4 * (3 * 32 * ZS);               // Wglu transposed [c][co], stride 65
    float* zt = smem + wv * (3 * 32 * ZS);
    float* yt = zt + 32 * ZS;
    float* dlt = yt + 32 * ZS;
    for (int e = tid; e < 4096; e += 256) WsT[(e & 63) * ZS + (e >> 6)] = wglu[e];
    __syncthreads();
    // forward operand B[k=c][j=co] = Wglu[co][c] lives in registers; the transposed one for dz = dlin @ Wglu,
    // B[k=co][j=c] = WsT[c][co], is read from LDS per use (conflict-free at stride 65): its 64 registers hold
    // the next row block's prefetched tile instead
    float bw[32][2];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        bw[s][0] = WsT[(2 * s + kh) * ZS + n];
        bw[s][1] = WsT[(2 * s + kh) * ZS + 32 + n];
    }
    const float* BT = WsT + n * ZS + kh;
    const float bg[2] = {bglu[n], bglu[32 + n]};
    const float sc = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    (void)block_id;
    f32x16 dW[2][2];   // [co block][c block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW[a][b2][r] = 0.f;
    float sdb[2] = {0.f, 0.f}, sdz[2] = {0.f, 0.f}, sdzy[2] = {0.f, 0.f};
    const int n_rb = (Q + 3) / 4;
    const float4 bsc = *(const float4*)(bn + 128 + (lane & 15) * 4), bsh = *(const float4*)(bn + 192 + (lane & 15) * 4);
    YTile yt_n;
    float gq_n[2][4];
    uint32_t m_n[2];
    auto prefetch = [&](int rbn) {
        tile_load(yt_n, y, rbn * 4, Q, H, W, Ho, Wo, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) gq_n[h][jx] = (rbn * 4 + jx < Q) ? dp[(size_t)(rbn * 4 + jx) * 64 + 32 * h + n] : 0.f;
            m_n[h] = use_drop ? (uint32_t)mask_in[((size_t)rbn * 2 + h) * 64 + lane] : 0xffffu;
        }
    };
    if (blockIdx.x * 4 + wv < n_rb) prefetch(blockIdx.x * 4 + wv);
    for (int rb = blockIdx.x * 4 + wv; rb < n_rb; rb += gridDim.x * 4) {
        const int q0 = rb * 4;
        tile_store<true>(yt_n, bsc, bsh, zt, yt, q0, Q, lane);
        float gq_c[2][4];
        uint32_t m_c[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            m_c[h] = m_n[h];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) gq_c[h][jx] = gq_n[h][jx] * sc;
        }
        if (rb + gridDim.x * 4 < n_rb) prefetch(rb + gridDim.x * 4);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        {
            const float* A = zt + n * ZS + kh;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float a = A[2 * s];
                acc[0] = mfma32(a, bw[s][0], acc[0]);
                acc[1] = mfma32(a, bw[s][1], acc[1]);
            }
        }
        float dzg[2][16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 32 * h + n;
            const uint32_t m16 = m_c[h];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = mfma32_row(r, lane);
                const float gg = ((m16 >> r) & 1u) ? gq_c[h][r >> 2] : 0.f;
                const float sg = sigmoidf_fast(zt[i * ZS + c]);
                const float dl = gg * sg;
                dlt[i * ZS + c] = dl;
                sdb[h] += dl;
                dzg[h][r] = gg * (acc[h][r] + bg[h]) * sg * (1.0f - sg);
            }
        }
        // dz = dlin @ Wglu (+ gate path)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        {
            const float* A = dlt + n * ZS + kh;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float a = A[2 * s];
                acc[0] = mfma32(a, BT[2 * s], acc[0]);
                acc[1] = mfma32(a, BT[32 * ZS + 2 * s], acc[1]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 32 * h + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = mfma32_row(r, lane);
                const int q = q0 + (i >> 3);
                const float v = acc[h][r] + dzg[h][r];
                if (q < Q) {
                    const int pix = rb_pixel(q, (i >> 2) & 1, i & 3, H, W, Ho, Wo);
                    dz[(size_t)pix * 64 + c] = v;
                    sdz[h] += v;
                    sdzy[h] += v * yt[i * ZS + c];
                }
            }
        }
        // dWglu[co][c] += sum_m dlin[m][co] z[m][c]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int mrow = 2 * s + kh;
            const float a0 = dlt[mrow * ZS + n], a1 = dlt[mrow * ZS + 32 + n];
            const float b0 = zt[mrow * ZS + n], b1 = zt[mrow * ZS + 32 + n];
            dW[0][0] = mfma32(a0, b0, dW[0][0]);
            dW[0][1] = mfma32(a0, b1, dW[0][1]);
            dW[1][0] = mfma32(a1, b0, dW[1][0]);
            dW[1][1] = mfma32(a1, b1, dW[1][1]);
        }
    }
    // ---- reduce across the 4 waves through LDS, then fp64 atomics ------------------------------
    __syncthreads();
    float* red = smem;   // needs 4 * 4096 floats = 64 KB <= 4 * 3 * 32 * 65 * 4 = 99,840 B
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wv * 4096 + (32 * a + mfma32_row(r, lane)) * 64 + 32 * b2 + n] = dW[a][b2][r];
    __syncthreads();
    for (int i = tid; i < 4096 && !no_atomic; i += 256)
        atomicAdd(&accg[i], (double)red[i] + (double)red[4096 + i] + (double)red[8192 + i] + (double)red[12288 + i]);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float v0 = sdb[h] + __shfl_xor(sdb[h], 32);
        const float v1 = sdz[h] + __shfl_xor(sdz[h], 32);
        const float v2 = sdzy[h] + __shfl_xor(sdzy[h], 32);
        if (kh == 0) {
            red[(wv * 3 + 0) * 64 + 32 * h + n] = v0;
            red[(wv * 3 + 1) * 64 + 32 * h + n] = v1;
            red[(wv * 3 + 2) * 64 + 32 * h + n] = v2;
        }
    }
    __syncthreads();
    if (tid < 192) {
        const int which = tid >> 6, c = tid & 63;
        double v = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) v += (double)red[(w2 * 3 + which) * 64 + c];
        if (!no_atomic) atomicAdd(&accg[4096 + which * 64 + c], v);
    }
}

struct BnBwdPrepArgs {
    const double* acc; double N;
    const float *gamma, *bn;
    float *coef, *g_gamma, *g_beta, *g_wglu, *g_bglu, *g_convb;
};
__global__ __launch_bounds__(64) void k_bn_bwd_prep(BnBwdPrepArgs a) {
    const int c = threadIdx.x;
    const double mean = a.bn[c], invstd = a.bn[64 + c], scale = a.bn[128 + c];
    const double Sdz = a.acc[4160 + c], Sdzy = a.acc[4224 + c];
    const double Sdzxhat = invstd * (Sdzy - mean * Sdz);
    a.g_beta[c] = (float)Sdz;
    a.g_gamma[c] = (float)Sdzxhat;
    const double m1 = Sdz / a.N, m2 = Sdzxhat / a.N;
    // dy = scale * (dz - m1 - xhat*m2),  xhat = (y - mean) * invstd
    a.coef[c] = (float)scale;
    a.coef[64 + c] = (float)(-scale * m2 * invstd);
    a.coef[128 + c] = (float)(scale * (m2 * invstd * mean - m1));
    for (int k = 0; k < 64; ++k) a.g_wglu[c * 64 + k] = (float)a.acc[c * 64 + k];
    a.g_bglu[c] = (float)a.acc[4096 + c];
    a.g_convb[c] = 0.f;   // sum_p dy == 0: a conv bias in front of a train-mode BatchNorm has zero gradient
}

// ---- host launchers -------------------------------------------------------------------------------
int launch_bn_prep(const double* stat, double N, const float* gamma, const float* beta, float* run_mean, float* run_var,
                   int64_t* tracked, int train, int update, float eps, float momentum, float* bn, hipStream_t st) {
    BnPrepArgs a;
    a.stat = stat; a.N = N; a.gamma = gamma; a.beta = beta; a.run_mean = run_mean; a.run_var = run_var;
    a.tracked = tracked; a.train = train; a.update = update; a.eps = eps; a.momentum = momentum; a.bn = bn;
    k_bn_prep<<<1, 64, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_glu_pool_fwd(const float* y, const float* bn, const float* wglu, const float* bglu, float* p, int B, int H,
                        int W, int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out,
                        hipStream_t st) {
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo;
    const int n_rb = (Q + 3) / 4;
    int grid = (n_rb + 3) / 4;
    if (grid > 512) grid = 512;
    k_glu_pool_fwd<<<grid, 256, 0, st>>>(y, bn, wglu, bglu, p, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, seed, mask_out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_glu_pool_bwd(const float* y, const float* bn, const float* wglu, const float* bglu, const float* dp, float* dz,
                        double* acc, int zero_acc, int B, int H, int W, int block_id, int use_drop, float p_drop,
                        const uint16_t* mask_in, hipStream_t st) {
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo;
    const size_t lds = (size_t)(4 * 3 * 32 * ZS + 64 * ZS) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_glu_pool_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    if (zero_acc) SED_CHECK_HIP(hipMemsetAsync(acc, 0, GLUACC_N * sizeof(double), st));
    if (H & 1) SED_CHECK_HIP(hipMemsetAsync(dz, 0, (size_t)B * H * W * 64 * sizeof(float), st));
    const int n_rb = (Q + 3) / 4;
    int grid = (n_rb + 3) / 4;
    if (grid > 256) grid = 256;
    k_glu_pool_bwd<<<grid, 256, lds, st>>>(y, bn, wglu, bglu, dp, dz, acc, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, mask_in, g_sed_debug & 1);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_bn_bwd_prep(const double* acc, double N, const float* gamma, const float* bn, float* coef, float* g_gamma,
                       float* g_beta, float* g_wglu, float* g_bglu, float* g_convb, hipStream_t st) {
    BnBwdPrepArgs a;
    a.acc = acc; a.N = N; a.gamma = gamma; a.bn = bn; a.coef = coef; a.g_gamma = g_gamma; a.g_beta = g_beta;
    a.g_wglu = g_wglu; a.g_bglu = g_bglu; a.g_convb = g_convb;
    k_bn_bwd_prep<<<1, 64, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
