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
#include <type_traits>

#define ZS 65   // pixel stride of LDS tiles (floats)

SED_TS_DEFINE(bnglu)

struct BnPrepArgs {
    const double* stat; double N;
    const float *gamma, *beta;
    float *run_mean, *run_var; int64_t* tracked;
    int train, update; float eps, momentum;
    float* bn;
};
__device__ __forceinline__ void bn_prep_body(const BnPrepArgs& a, int c, bool publish, float* bn_s);

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

// The BatchNorm statistics -> (mean, invstd, scale, shift) step of k_bn_prep runs in the prologue of every workgroup
// (64 threads, a few fp64 operations) instead of as a 1-workgroup kernel of its own in front: one launch and one
// dependency gap less on the forward chain per conv block.  Workgroup 0 also publishes bn[] for the backward and
// updates the running statistics.
__device__ __forceinline__ void bn_prep_body(const BnPrepArgs& a, int c, bool publish, float* bn_s /* LDS [256] */) {
    double mean, var;
    if (a.train) {
        mean = a.stat[c] / a.N;
        var = a.stat[64 + c] / a.N - mean * mean;
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
    const float v0 = (float)mean, v1 = (float)invstd, v2 = (float)scale, v3 = (float)(a.beta[c] - mean * scale);
    bn_s[c] = v0; bn_s[64 + c] = v1; bn_s[128 + c] = v2; bn_s[192 + c] = v3;
    if (publish) { a.bn[c] = v0; a.bn[64 + c] = v1; a.bn[128 + c] = v2; a.bn[192 + c] = v3; }
}

__global__ __launch_bounds__(256, 2) void k_glu_pool_fwd(const float* __restrict__ y, BnPrepArgs bnp,
                                                       const float* __restrict__ wglu, const float* __restrict__ bglu,
                                                       float* __restrict__ p, int H, int W, int Ho, int Wo, int Q,
                                                       int block_id, int use_drop, float p_drop,
                                                       const uint64_t* __restrict__ seed_ptr, uint16_t* __restrict__ mask_out) {
    __shared__ float zts[4][32 * ZS];
    __shared__ float WsT[64 * ZS];     // Wglu transposed [c][co], stride 65: coalesced global read, conflict-free both ways
    __shared__ __attribute__((aligned(16))) float bn_s[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    float* zt = zts[wv];
    if (tid < 64) bn_prep_body(bnp, tid, blockIdx.x == 0, bn_s);
    const float* bn = bn_s;
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
        // A operands two k-steps (four MFMAs) AHEAD, in rotating registers: written as `a = A[2 s]; mfma; mfma` the compiler
        // issued each ds_read right in front of the MFMAs that need it and the wave sat out one LDS latency per k-step pair
        // (ds_read2_b32 | s_waitcnt lgkmcnt(0) | 4 MFMAs, sixteen times per row block)
        float av[4];
        av[0] = A[0]; av[1] = A[2];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (s + 2 < 32) av[(s + 2) & 3] = A[2 * (s + 2)];
            acc[0] = mfma32(av[s & 3], bw[s][0], acc[0]);
            acc[1] = mfma32(av[s & 3], bw[s][1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
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

#define GLUACC_N SED_GLUACC_N     // layout: common.h

// (BnBwdPrepArgs, bn_bwd_coef, bn_bwd_prep_body: kernels.h - shared with the conv kernels that consume the result)

__global__ __launch_bounds__(256) void k_glu_pool_bwd(const float* __restrict__ y, const float* __restrict__ bn,
                                                       const float* __restrict__ wglu, const float* __restrict__ bglu,
                                                       const float* __restrict__ dp, const float* __restrict__ dp_b,
                                                       float* __restrict__ dz, double* __restrict__ accg, int H, int W, int Ho, int Wo, int Q,
                                                       int block_id, int use_drop, float p_drop,
                                                       const uint16_t* __restrict__ mask_in, int no_atomic) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    float* WsT = smem + 4 * (3 * 32 * ZS);               // Wglu transposed [c][co], stride 65
    float* zt = smem + wv * (3 * 32 * ZS);
    float* yt = zt + 32 * ZS;
    float* dlt = yt + 32 * ZS;
    TS(0); TSC(14);
    if (H & 1) {        // the floor-mode pool drops the last row of an odd-height image: its gradient is 0
        const int per_clip = W * 64, nb = Q / (Ho * Wo);
        for (int i = blockIdx.x * 256 + tid; i < nb * per_clip; i += gridDim.x * 256) {
            const int bb = i / per_clip, r = i % per_clip;
            dz[((size_t)bb * H + (H - 1)) * W * 64 + r] = 0.f;
        }
    }
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
    // Addressing without per-element divisions or 64-bit arithmetic: po[j] = BYTE offset (into y / dz, both
    // [pixels][64] fp32 < 4 GB) of the image pixel of pooled pixel q0+j at (dt, df) = (0, 0); MFMA row i of the
    // row block is pixel po[i>>3]/256 + ((i>>2)&1)*W + (i&3).  Pooled pixels past Q (only in the last row block,
    // only if Q % 4 != 0) get offset 0 - a harmless load - and are masked where it matters.
    // (integer division by a run-time value is ~50 instructions with quarter-rate multiplies - 4 of them per row
    // block were 0.8 us of an in-order wave's time; q < 2^24, so a float reciprocal plus one correction is exact)
    const float inv_wo = 1.0f / (float)Wo, inv_ho = 1.0f / (float)Ho;
    auto divmod = [](int a, int d, float inv, int& rem) {
        int qd = (int)((float)a * inv);
        int r = a - qd * d;
        if (r < 0) { r += d; --qd; }
        if (r >= d) { r -= d; ++qd; }
        rem = r;
        return qd;
    };
    auto pixel_offsets = [&](int q0, uint32_t (&po)[4]) {
        int wo, ho;
        const int t = divmod(q0, Wo, inv_wo, wo);
        int bb = divmod(t, Ho, inv_ho, ho);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            po[j] = (q0 + j < Q) ? (uint32_t)((bb * H + 2 * ho) * W + 4 * wo) * 256u : 0u;
            if (++wo == Wo) { wo = 0; if (++ho == Ho) { ho = 0; ++bb; } }
        }
    };
    const uint32_t ld_off = (uint32_t)((lane >> 4) * 64 + (lane & 15) * 4) * 4u;   // tile_load lane: pixel df, 4 channels
    const uint32_t st_off = (uint32_t)(kh * W * 64 + n) * 4u;                      // D-fragment lane: row dt = kh, channel n
    YTile yt_n;
    float gq_n[2][4];
    uint32_t m_n[2];
    uint32_t po_n[4];
    // The next row block's operands are fetched in 8 slices spread over the groups of phase 1 rather than in one
    // burst: 41 KB per CU issued at once (and by all 256 CUs at the same moment) exceeds what a CU can keep in
    // flight, and the in-order wave sat ~1.5 us per row block in the ISSUE of those loads.
    int q_n = 0;                       // first pooled pixel of the row block being prefetched
    auto prefetch_begin = [&](int rbn, bool valid) {
        q_n = valid ? rbn * 4 : 0;     // past the end: re-read row block 0 (harmless, never consumed)
        pixel_offsets(q_n, po_n);
    };
    auto prefetch_slice = [&](int k) {
        yt_n.v[k] = *(const float4*)((const char*)y + (po_n[k >> 1] + (uint32_t)((k & 1) * W) * 256u + ld_off));
        const int h = k >> 2, jx = k & 3;
        const int q = q_n + jx;
        const uint32_t goff = (uint32_t)((q < Q ? q : 0) * 64 + 32 * h + n) * 4u;
        gq_n[h][jx] = *(const float*)((const char*)dp + goff);
        if (dp_b) gq_n[h][jx] += *(const float*)((const char*)dp_b + goff);     // second direction plane of the GRU's dX
        if (q >= Q) gq_n[h][jx] = 0.f;
        if (jx == 0) m_n[h] = use_drop ? (uint32_t)mask_in[((size_t)(q_n >> 2) * 2 + h) * 64 + lane] : 0xffffu;
    };
    auto prefetch = [&](int rbn) {
        prefetch_begin(rbn, true);
#pragma unroll
        for (int k = 0; k < 8; ++k) prefetch_slice(k);
    };
    if (blockIdx.x * 4 + wv < n_rb) prefetch(blockIdx.x * 4 + wv);
    TS(1);
    int ts_k = 2;
    (void)ts_k;
    // one row block; FULL = all four pooled pixels exist (always, unless it is the last row block and Q % 4 != 0)
    auto body = [&](int rb, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int q0 = rb * 4;
        tile_store<true>(yt_n, bsc, bsh, zt, yt, q0, FULL ? q0 + 4 : Q, lane);
        TS(ts_k); ++ts_k;
        float gq_c[2][4];
        uint32_t m_c[2];
        uint32_t po[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) po[j] = po_n[j] + st_off;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            m_c[h] = m_n[h];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) gq_c[h][jx] = gq_n[h][jx] * sc;
        }
        prefetch_begin(rb + gridDim.x * 4, rb + gridDim.x * 4 < n_rb);
        if (ts_k == 3) TS(7);
        // The three MFMA phases (64 x v_mfma_f32_32x32x2 each) carry the element-wise work of the row block in
        // their shadow: one wave per SIMD issues in order, so VALU / LDS work placed BETWEEN independent MFMAs is
        // free, while the same work in a phase of its own leaves the MFMA pipe idle (21 % busy in the first version).
        // ---- phase 1: lin = z @ Wglu^T   ||   sigma(z), dlin = g*sigma -> LDS, t = g*sigma*(1-sigma) -------------
        f32x16 lin[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { lin[0][r] = 0.f; lin[1][r] = 0.f; }
        float dzg[2][16];
        {
            // Groups of 4 K-steps (8 MFMAs = 512 MFMA-pipe cycles) carry 4 elements' VALU work side by side: one
            // element's mul -> exp -> add -> rcp -> mul -> ds_write is a dependent chain of ~100 cycles on a lone
            // in-order wave, four of them interleave.  The LDS operands of group g+1 are read during group g.
            const float* A = zt + n * ZS + kh;
            float a_c[4], z_c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a_c[u] = A[2 * u]; z_c[u] = zt[mfma32_row(u, lane) * ZS + n]; }
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                float a_n[4] = {0.f, 0.f, 0.f, 0.f}, z_n[4] = {0.f, 0.f, 0.f, 0.f};
                if (g4 + 1 < 8) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s1 = 4 * (g4 + 1) + u;
                        a_n[u] = A[2 * s1];
                        z_n[u] = zt[mfma32_row(s1 & 15, lane) * ZS + 32 * (s1 >> 4) + n];
                    }
                }
                prefetch_slice(g4);
                float sg[4], gg[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = 4 * g4 + u, h = s >> 4, r = s & 15;
                    lin[0] = mfma32(a_c[u], bw[s][0], lin[0]);
                    lin[1] = mfma32(a_c[u], bw[s][1], lin[1]);
                    gg[u] = ((m_c[h] >> r) & 1u) ? gq_c[h][r >> 2] : 0.f;
                    sg[u] = sigmoidf_fast(z_c[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = 4 * g4 + u, h = s >> 4, r = s & 15;
                    const float dl = gg[u] * sg[u];
                    dlt[mfma32_row(r, lane) * ZS + 32 * h + n] = dl;
                    sdb[h] += dl;
                    dzg[h][r] = dl * (1.0f - sg[u]);
                    a_c[u] = a_n[u]; z_c[u] = z_n[u];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ts_k == 3) TS(11);
        // ---- phase 2: dz_lin = dlin @ Wglu   ||   gate path dzg = t * (lin + b) -----------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        {
            const float* A = dlt + n * ZS + kh;
            float a_c = A[0], b0_c = BT[0], b1_c = BT[32 * ZS];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int h = s >> 4, r = s & 15;
                float a_n = 0.f, b0_n = 0.f, b1_n = 0.f;
                if (s + 1 < 32) { a_n = A[2 * (s + 1)]; b0_n = BT[2 * (s + 1)]; b1_n = BT[32 * ZS + 2 * (s + 1)]; }
                acc[0] = mfma32(a_c, b0_c, acc[0]);
                acc[1] = mfma32(a_c, b1_c, acc[1]);
                dzg[h][r] *= lin[h][r] + bg[h];
                a_c = a_n; b0_c = b0_n; b1_c = b1_n;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ts_k == 3) TS(12);
        // vmcnt counts loads and stores alike and cannot tell them apart: claim the prefetched registers HERE, while
        // only the (long issued) loads are outstanding - at the top of the next row block the same wait would also
        // drain the 32 dz stores per lane that phase 3 is about to issue (a full store round trip per row block)
#pragma unroll
        for (int it = 0; it < 8; ++it)
            asm volatile("" : "+v"(yt_n.v[it].x), "+v"(yt_n.v[it].y), "+v"(yt_n.v[it].z), "+v"(yt_n.v[it].w));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile("" : "+v"(gq_n[h][0]), "+v"(gq_n[h][1]), "+v"(gq_n[h][2]), "+v"(gq_n[h][3]), "+v"(m_n[h]));
        }
        // ---- phase 3: dWglu += dlin^T z   ||   dz = dz_lin + dzg -> global, BatchNorm-backward sums ----------------
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int mrow = 2 * s + kh;
            const float a0 = dlt[mrow * ZS + n], a1 = dlt[mrow * ZS + 32 + n];
            const float b0 = zt[mrow * ZS + n], b1 = zt[mrow * ZS + 32 + n];
            dW[0][0] = mfma32(a0, b0, dW[0][0]);
            dW[0][1] = mfma32(a0, b1, dW[0][1]);
            dW[1][0] = mfma32(a1, b0, dW[1][0]);
            dW[1][1] = mfma32(a1, b1, dW[1][1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = s, c = 32 * h + n;
                const int i = mfma32_row(r, lane);
                const float v = acc[h][r] + dzg[h][r];
                if (FULL || q0 + (r >> 2) < Q) {
                    *(float*)((char*)dz + (po[r >> 2] + (uint32_t)(((r & 3) * 64 + 32 * h) * 4))) = v;
                    sdz[h] += v;
                    sdzy[h] += v * yt[i * ZS + c];
                }
            }
        }
        if (ts_k == 3) TS(13);
    };
    for (int rb = blockIdx.x * 4 + wv; rb < n_rb; rb += gridDim.x * 4) {
        if (rb * 4 + 3 < Q) body(rb, std::true_type{});
        else body(rb, std::false_type{});
    }
    // ---- reduce across the 4 waves through LDS, then fp64 atomics ------------------------------
    TS(8);
    __syncthreads();
    TS(9);
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
    TS(10); TSC(15);
}

// ---- 8-wave variant: two waves per SIMD share one row block, split by channel half ---------------------------------
// k_glu_pool_bwd above runs one wave per SIMD (415 registers) and that wave issues in order: whatever VALU / LDS /
// global work is not perfectly interleaved with its MFMAs leaves the MFMA pipe idle (phase timestamps: 9.5 us per row
// block for 5.7 us of MFMAs).  Here waves w and w+4 - same SIMD - work on the SAME row block: wave half h owns the 32
// channels [32h, 32h+32) everywhere a channel index is an OUTPUT (lin columns, dz columns, dW rows), so each wave has
// half the accumulators (<= 256 registers: two waves per SIMD) and, while one wave of the pair is in VALU / LDS /
// memory instructions, the other's MFMAs keep the pipe busy.  The tile (z, y, dlin) is shared through LDS; three
// LDS-only workgroup barriers per row block (tile staged | dlin complete | tile free).
__global__ __launch_bounds__(512, 1) void k_glu_pool_bwd8(const float* __restrict__ y, const float* __restrict__ bn,
                                                          const float* __restrict__ wglu, const float* __restrict__ bglu,
                                                          const float* __restrict__ dp, const float* __restrict__ dp_b,
                                                          float* __restrict__ dz, double* __restrict__ accg, int H, int W, int Ho,
                                                          int Wo, int Q, int use_drop, float p_drop,
                                                          const uint16_t* __restrict__ mask_in) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pr = wave & 3, hf = wave >> 2;                 // pair (row-block slot), channel half
    const int n = lane & 31, kh = lane >> 5;
    const int c_own = 32 * hf + n;                           // the channel this lane owns as an output column
    float* WsT = smem + 4 * (3 * 32 * ZS);                   // Wglu transposed [c][co], stride 65
    float* zt = smem + pr * (3 * 32 * ZS);
    float* yt = zt + 32 * ZS;
    float* dlt = yt + 32 * ZS;
    if (H & 1) {        // the floor-mode pool drops the last row of an odd-height image: its gradient is 0
        const int per_clip = W * 64, nb = Q / (Ho * Wo);
        for (int i = blockIdx.x * 512 + tid; i < nb * per_clip; i += gridDim.x * 512) {
            const int bb = i / per_clip, r = i % per_clip;
            dz[((size_t)bb * H + (H - 1)) * W * 64 + r] = 0.f;
        }
    }
    for (int e = tid; e < 4096; e += 512) WsT[(e & 63) * ZS + (e >> 6)] = wglu[e];
    __syncthreads();
    float bw[32];                                            // phase 1: B[k = c][j = co own half] = Wglu[co][c]
#pragma unroll
    for (int s = 0; s < 32; ++s) bw[s] = WsT[(2 * s + kh) * ZS + c_own];
    const float* BT = WsT + c_own * ZS + kh;                 // phase 2: B[k = co][j = c own half] = WsT[c][co]
    const float bgl = bglu[c_own];
    const float sc = 0.125f * (use_drop ? drop_scale8(p_drop) : 1.0f);
    f32x16 dW[2];                                            // dW[co own half][c block b2]
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[b2][r] = 0.f;
    float sdb = 0.f, sdz = 0.f, sdzy = 0.f;
    const int n_rb = (Q + 3) / 4;
    const float4 bsc = *(const float4*)(bn + 128 + (lane & 15) * 4), bsh = *(const float4*)(bn + 192 + (lane & 15) * 4);
    const float inv_wo = 1.0f / (float)Wo, inv_ho = 1.0f / (float)Ho;
    auto divmod = [](int a, int d, float inv, int& rem) {
        int qd = (int)((float)a * inv);
        int r = a - qd * d;
        if (r < 0) { r += d; --qd; }
        if (r >= d) { r -= d; ++qd; }
        rem = r;
        return qd;
    };
    auto pixel_offsets = [&](int q0, uint32_t (&po)[4]) {     // byte offsets of the 4 pooled pixels' image pixels (dt = df = 0)
        int wo, ho;
        const int t = divmod(q0, Wo, inv_wo, wo);
        int bb = divmod(t, Ho, inv_ho, ho);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            po[j] = (q0 + j < Q) ? (uint32_t)((bb * H + 2 * ho) * W + 4 * wo) * 256u : 0u;
            if (++wo == Wo) { wo = 0; if (++ho == Ho) { ho = 0; ++bb; } }
        }
    };
    const uint32_t ld_off = (uint32_t)((lane >> 4) * 64 + (lane & 15) * 4) * 4u;
    const uint32_t st_off = (uint32_t)(kh * W * 64 + c_own) * 4u;
    // this wave stages rows m = (lane >> 4) + 4 * it for it in [4 hf, 4 hf + 4) of the pair's tile
    f32x4 yv[4];
    float gq_n[4];
    uint32_t m_n = 0xffffu, po_n[4];
    int q_n = 0;
    auto prefetch_begin = [&](int rbn, bool valid) {
        q_n = valid ? rbn * 4 : 0;
        pixel_offsets(q_n, po_n);
    };
    auto prefetch_slice = [&](int k) {                         // k in 0..3
        // rows it = 4 hf + k: pooled pixel 2 hf + (k >> 1) (selected without indexing po_n by a run-time value, which
        // would put the array in scratch), image row dt = k & 1
        const uint32_t pbase = (k >> 1) ? (hf ? po_n[3] : po_n[1]) : (hf ? po_n[2] : po_n[0]);
        yv[k] = *(const f32x4*)((const char*)y + (pbase + (uint32_t)((k & 1) * W) * 256u + ld_off));
        const int q = q_n + k;
        const uint32_t goff = (uint32_t)((q < Q ? q : 0) * 64 + c_own) * 4u;
        gq_n[k] = *(const float*)((const char*)dp + goff);
        if (dp_b) gq_n[k] += *(const float*)((const char*)dp_b + goff);
        if (q >= Q) gq_n[k] = 0.f;
        if (k == 0) m_n = use_drop ? (uint32_t)mask_in[((size_t)(q_n >> 2) * 2 + hf) * 64 + lane] : 0xffffu;
    };
    const int rb0 = blockIdx.x * 4 + pr;
    prefetch_begin(rb0, rb0 < n_rb);
#pragma unroll
    for (int k = 0; k < 4; ++k) prefetch_slice(k);
    // every pair runs the same number of trips (the barriers are workgroup-wide); pairs past the end idle through them
    const int trips = (n_rb - blockIdx.x * 4 + gridDim.x * 4 - 1) / (gridDim.x * 4);
    for (int trip = 0; trip < trips; ++trip) {
        const int rb = rb0 + trip * gridDim.x * 4;
        const bool live = rb < n_rb;
        const int q0 = rb * 4;
        if (live) {
            // ---- stage this wave's 4 of the 8 row groups: z = scale * y + shift, and y itself --------------------
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int it = 4 * hf + k, m = (lane >> 4) + 4 * it, c4 = (lane & 15) * 4;
                const f32x4 v = yv[k];
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (q0 + (m >> 3) < Q) { z.x = fmaf(v.x, bsc.x, bsh.x); z.y = fmaf(v.y, bsc.y, bsh.y); z.z = fmaf(v.z, bsc.z, bsh.z); z.w = fmaf(v.w, bsc.w, bsh.w); }
                float* d = zt + m * ZS + c4;
                d[0] = z.x; d[1] = z.y; d[2] = z.z; d[3] = z.w;
                float* e = yt + m * ZS + c4;
                e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
            }
        }
        float gq_c[4];
        uint32_t po[4];
        const uint32_t m_c = m_n;
#pragma unroll
        for (int j = 0; j < 4; ++j) { po[j] = po_n[j] + st_off; gq_c[j] = gq_n[j] * sc; }
        prefetch_begin(rb + gridDim.x * 4, rb + gridDim.x * 4 < n_rb);
        lds_barrier();                                        // A: the pair's tile is complete
        f32x16 lin, acc;
        float dzg[16];
        if (live) {
            // ---- phase 1: lin[:, own half] = z @ Wglu^T   ||   sigma(z), dlin = g * sigma -> LDS -------------------
#pragma unroll
            for (int r = 0; r < 16; ++r) lin[r] = 0.f;
            const float* A = zt + n * ZS + kh;
            float a_c[4], z_c[2];
#pragma unroll
            for (int u = 0; u < 4; ++u) a_c[u] = A[2 * u];
#pragma unroll
            for (int u = 0; u < 2; ++u) z_c[u] = zt[mfma32_row(u, lane) * ZS + c_own];
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {                  // 4 K-steps + 2 elements per group
                float a_n[4] = {0.f, 0.f, 0.f, 0.f}, z_n[2] = {0.f, 0.f};
                if (g4 + 1 < 8) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) a_n[u] = A[2 * (4 * (g4 + 1) + u)];
#pragma unroll
                    for (int u = 0; u < 2; ++u) z_n[u] = zt[mfma32_row(2 * (g4 + 1) + u, lane) * ZS + c_own];
                }
                if (g4 < 4) prefetch_slice(g4);
#pragma unroll
                for (int u = 0; u < 4; ++u) lin = mfma32(a_c[u], bw[4 * g4 + u], lin);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = 2 * g4 + u;
                    const float gg = ((m_c >> r) & 1u) ? gq_c[r >> 2] : 0.f;
                    const float sg = sigmoidf_fast(z_c[u]);
                    const float dl = gg * sg;
                    dlt[mfma32_row(r, lane) * ZS + c_own] = dl;
                    sdb += dl;
                    dzg[r] = dl * (1.0f - sg);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) a_c[u] = a_n[u];
                z_c[0] = z_n[0]; z_c[1] = z_n[1];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();                                        // B: dlin of both halves is in LDS
        if (live) {
            // ---- phase 2: dz_lin[:, own half] = dlin @ Wglu   ||   gate path dzg = t * (lin + b) -------------------
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* A = dlt + n * ZS + kh;
            // operands two MFMAs ahead in rotating registers (a `_c = _n` copy per step made the wave wait for the prefetch
            // it had just issued: ds_read x2 | s_waitcnt lgkmcnt(0) | one MFMA, thirty-two times)
            float a_r[4], b_r[4];
            a_r[0] = A[0]; b_r[0] = BT[0]; a_r[1] = A[2]; b_r[1] = BT[2];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                if (s + 2 < 32) { a_r[(s + 2) & 3] = A[2 * (s + 2)]; b_r[(s + 2) & 3] = BT[2 * (s + 2)]; }
                acc = mfma32(a_r[s & 3], b_r[s & 3], acc);
                if (s < 16) dzg[s] *= lin[s] + bgl;
                __builtin_amdgcn_sched_barrier(0);
            }
            // claim the prefetched registers while only loads are outstanding (vmcnt cannot tell loads from stores)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(yv[k]), "+v"(gq_n[k]));
            asm volatile("" : "+v"(m_n));
            // ---- phase 3: dW[own co half][:] += dlin^T z   ||   dz = dz_lin + dzg -> global, BN-backward sums ------
            float a3[2], b3[2][2];
            a3[0] = dlt[kh * ZS + c_own]; b3[0][0] = zt[kh * ZS + n]; b3[0][1] = zt[kh * ZS + 32 + n];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {          // the next step's three operands now, one pair of MFMAs ahead
                    const int mn = 2 * (s + 1) + kh;
                    a3[(s + 1) & 1] = dlt[mn * ZS + c_own];
                    b3[(s + 1) & 1][0] = zt[mn * ZS + n]; b3[(s + 1) & 1][1] = zt[mn * ZS + 32 + n];
                }
                dW[0] = mfma32(a3[s & 1], b3[s & 1][0], dW[0]);
                dW[1] = mfma32(a3[s & 1], b3[s & 1][1], dW[1]);
                const int r = s, i = mfma32_row(r, lane);
                const float v = acc[r] + dzg[r];
                if (q0 + (r >> 2) < Q) {
                    *(float*)((char*)dz + (po[r >> 2] + (uint32_t)((r & 3) * 64 * 4))) = v;
                    sdz += v;
                    sdzy += v * yt[i * ZS + c_own];
                }
            }
        }
        lds_barrier();                                        // C: the tile may be overwritten
    }
    // ---- reduce over the 4 pairs through LDS (each (co, c) lives in exactly one half), then fp64 atomics ---------------
    float* red = smem;   // [4 pairs][64][64]: 64 KB of the 100 KB of tiles
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[pr * 4096 + (32 * hf + mfma32_row(r, lane)) * 64 + 32 * b2 + n] = dW[b2][r];
    __syncthreads();
    for (int i = tid; i < 4096; i += 512)
        atomicAdd(&accg[i], (double)red[i] + (double)red[4096 + i] + (double)red[8192 + i] + (double)red[12288 + i]);
    __syncthreads();
    {
        const float v0 = sdb + __shfl_xor(sdb, 32), v1 = sdz + __shfl_xor(sdz, 32), v2 = sdzy + __shfl_xor(sdzy, 32);
        if (kh == 0) {
            red[(pr * 3 + 0) * 64 + c_own] = v0;
            red[(pr * 3 + 1) * 64 + c_own] = v1;
            red[(pr * 3 + 2) * 64 + c_own] = v2;
        }
    }
    __syncthreads();
    if (tid < 192) {
        const int which = tid >> 6, c = tid & 63;
        double v = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) v += (double)red[(w2 * 3 + which) * 64 + c];
        atomicAdd(&accg[4096 + which * 64 + c], v);
    }
}

// BatchNorm-backward coefficients + the block's parameter gradients: a kernel of its own again.  Folding it into the
// last workgroup of k_glu_pool_bwd (ticket after a workgroup-scope release) was a RACE: about 1 step in 10 the last
// workgroup read the Sdz / Sdzy accumulators before every other workgroup's fp64 atomics had been performed
// (tools/determinism.py: bn1 / conv1 / block-0 gradients off by up to 7e-3); ordering those atomics device-wide needs
// __threadfence(), whose L2 write-back of the 31 MB of dz just produced costs more (17 us) than this launch (5 us).
__global__ __launch_bounds__(256) void k_bn_bwd_prep(BnBwdPrepArgs a) { bn_bwd_prep_body(a, threadIdx.x, 256); }

// ---- host launchers -------------------------------------------------------------------------------
int launch_glu_pool_fwd(const float* y, const double* stat, double N, const float* gamma, const float* beta, float* run_mean,
                        float* run_var, int64_t* tracked, int train, int update, float eps, float momentum, float* bn,
                        const float* wglu, const float* bglu, float* p, int B, int H, int W, int block_id, int use_drop,
                        float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st) {
    BnPrepArgs a;
    a.stat = stat; a.N = N; a.gamma = gamma; a.beta = beta; a.run_mean = run_mean; a.run_var = run_var;
    a.tracked = tracked; a.train = train; a.update = update; a.eps = eps; a.momentum = momentum; a.bn = bn;
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo;
    const int n_rb = (Q + 3) / 4;
    int grid = (n_rb + 3) / 4;
    if (grid > 512) grid = 512;
    k_glu_pool_fwd<<<grid, 256, 0, st>>>(y, a, wglu, bglu, p, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, seed, mask_out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_glu_pool_bwd(const float* y, const float* bn, const float* wglu, const float* bglu, const float* dp,
                        const float* dp_b, float* dz, double* acc, int zero_acc, int B, int H, int W, int block_id, int use_drop, float p_drop,
                        const uint16_t* mask_in, const float* gamma, float* coef, float* g_gamma, float* g_beta, float* g_wglu,
                        float* g_bglu, float* g_convb, BnBwdPrepArgs* prep_out, hipStream_t st) {
    const int Ho = H / 2, Wo = W / 4, Q = B * Ho * Wo;
    const size_t lds = (size_t)(4 * 3 * 32 * ZS + 64 * ZS) * sizeof(float);
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_glu_pool_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (zero_acc) SED_CHECK_HIP(hipMemsetAsync(acc, 0, GLUACC_N * sizeof(double), st));
    BnBwdPrepArgs a;
    a.acc = acc; a.N = (double)B * H * W; a.gamma = gamma; a.bn = bn; a.coef = coef; a.g_gamma = g_gamma; a.g_beta = g_beta;
    a.g_wglu = g_wglu; a.g_bglu = g_bglu; a.g_convb = g_convb;
    const int n_rb = (Q + 3) / 4;
    int grid = (n_rb + 3) / 4;
    if (grid > 256) grid = 256;
    if (g_sed_debug & 16) {          // single-wave-per-SIMD variant (A/B timing)
        k_glu_pool_bwd<<<grid, 256, lds, st>>>(y, bn, wglu, bglu, dp, dp_b, dz, acc, H, W, Ho, Wo, Q, block_id, use_drop, p_drop, mask_in, g_sed_debug & 1);
    } else {
        static thread_local SedAttrOnce attr8;
        if (attr8.need()) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_glu_pool_bwd8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        k_glu_pool_bwd8<<<grid, 512, lds, st>>>(y, bn, wglu, bglu, dp, dp_b, dz, acc, H, W, Ho, Wo, Q, use_drop, p_drop, mask_in);
    }
    SED_CHECK_LAUNCH();
    if (prep_out) {             // the conv dgrad / wgrad kernels derive the coefficients themselves
        *prep_out = a;
        return SED_OK;
    }
    k_bn_bwd_prep<<<1, 256, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

