// gen.h - building blocks of the GENERIC kernel family (gconv.hip, gglu.hip, ggru.hip, gcrnn.hip): any conv width
// C in {64, 128}, any GRU width H that is a multiple of 64 up to 256, MFMA operands in fp32 (MODE 0: exact f32,
// v_mfma_f32_32x32x2_f32) or bf16 (MODE 1: v_mfma_f32_32x32x16_bf16, fp32 accumulate).  The C = 64 / H = 64 / fp32
// configuration of baseline/config.py:53-58 keeps its own specialised kernels (conv.hip, bnglu.hip, gru.hip); this
// family serves BASELINE.json configs[2] (bf16 operands) and configs[4] (nb_filters 3 x 128, n_RNN_cell 256).
//
// Common shape of every GEMM here: C[m][n] += sum_k A[m][k] B[n][k] with BOTH operands k-contiguous - what the bf16
// MFMA wants (a lane supplies 8 consecutive k of one row) and what channels-last activations / [out][in] weights give
// for free.  A lives in an LDS tile owned by the workgroup (a halo tile of the image, a row block of pixels); B (weights,
// packed once per forward by k_gen_pack) streams from L2 through a double-buffered LDS chunk.  All HBM tensors stay fp32:
// operands are rounded to bf16 when they are staged into LDS, accumulators / BatchNorm statistics / gates are fp32.
#pragma once
#include "common.h"
#include "philox.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// ---- storage of the conv-block activations / gradients in HBM: fp32, or bf16 in SED_DTYPE_BF16 mode ----------------------
// (p0, y1, p1, y2 in ctx; dz1, dz2, dp0, dp1 in the backward workspace.  p2 - the GRU input - and dp2 stay fp32.)
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
template <int BF> struct Stor { using T = float; };
template <> struct Stor<1> { using T = __bf16; };

// ---- SED_DTYPE_F16: the forward chain's 16-bit flavour.  H16<0> = bf16 (the bf16 mode), H16<1> = fp16. ----------------------
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ f32x4 ld4(const _Float16* p) {
    const f16x4 v = *(const f16x4*)p;
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ float ld1(const _Float16* p) { return (float)*p; }
// fp16 stores SATURATE at +-65504: a value beyond fp16's range (none of the forward tensors of this model comes near it - they sit
// behind BatchNorms or are convolutions of O(1) inputs - but nothing in the arithmetic forbids conv weights from drifting there
// over a long run) becomes the largest finite fp16, which the BatchNorm behind it absorbs, instead of an Inf that turns the step
// into NaNs
__device__ __forceinline__ _Float16 f16_sat(float v) { return (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f); }
__device__ __forceinline__ void st1(_Float16* p, float v) { *p = f16_sat(v); }
template <int F16> struct H16 {
    using T = __bf16;
    using V8 = bf16x8;
    static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct H16<1> {
    using T = _Float16;
    using V8 = f16x8;
    static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <int MODE> struct MM;
template <> struct MM<0> {
    using E = float;
    using Frag = float;
    static constexpr int KPL = 1;       // k elements one lane supplies per MFMA operand
    static constexpr int KSTEP = 2;     // k per MFMA (32x32x2)
    static constexpr int PAD = 1;       // LDS row padding (elements): odd stride -> lane = row reads hit 32 distinct banks
    static constexpr int KC = 32;       // k-chunk of the streamed B operand
    static __device__ __forceinline__ Frag ld(const E* p) { return *p; }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ E cvt(float v) { return v; }
    static __device__ __forceinline__ void st4(E* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
};
template <> struct MM<1> {
    using E = __bf16;
    using Frag = bf16x8;
    static constexpr int KPL = 8;
    static constexpr int KSTEP = 16;    // 32x32x16
    static constexpr int PAD = 8;       // row stride (K + 8) * 2 B = odd multiple of 16 B: ds_read_b128 lane groups conflict-free
    static constexpr int KC = 64;
    static __device__ __forceinline__ Frag ld(const E* p) { return *(const Frag*)p; }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ E cvt(float v) { return (__bf16)v; }     // v_cvt_pk_bf16_f32: round to nearest even
    static __device__ __forceinline__ void st4(E* p, float a, float b, float c, float d) {
        bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        *(bf16x4*)p = v;                // 8-byte store (row strides are multiples of 8 B)
    }
};

// ---- B operand streamed through LDS -----------------------------------------------------------------------------------
// acc[nb][.] += sum_{k < K} A(row of this lane, k) * Bg[(nb0 + nb) * 32 + (lane & 31)][k],  nb < NBW
//   Bg     global, element type E, row-major [32 * NBT][ldb] (k contiguous), 16-byte aligned rows; ALL 32 * NBT rows are
//          staged (the waves of a workgroup may each take a different slice nb0 .. nb0 + NBW of them)
//   a_row  per-lane LDS pointer to the first k of this lane's A row; the A elements of chunk ch start at a_row + aoff(ch)
//          (a functor: the 3x3 convolution moves to another halo pixel every C / KC chunks)
//   wbuf   LDS, 2 x [32 * NBT][KC + PAD] elements; every thread of the 256-thread workgroup must call (staging + barriers:
//          one barrier before the first MFMA - so LDS tiles written before the call are visible to all waves - and one
//          after the last, so they may be overwritten right after the call)
template <int MODE, int NBT, int NBW, int KC, class AOff>
__device__ __forceinline__ void stream_gemm(const typename MM<MODE>::E* a_row, AOff aoff, const typename MM<MODE>::E* __restrict__ Bg,
                                            int ldb, int K, typename MM<MODE>::E* wbuf, f32x16 (&acc)[NBW], int nb0, int tid) {
    using M = MM<MODE>;
    using E = typename M::E;
    constexpr int N = 32 * NBT, WS = KC + M::PAD;
    constexpr int EPV = 16 / (int)sizeof(E);                 // elements per 16-byte vector
    constexpr int VPR = KC / EPV;                            // vectors per row
    constexpr int NV = N * VPR / 256;                        // vectors per thread per chunk
    static_assert(N * VPR % 256 == 0 && NV >= 1, "chunk must divide over 256 threads");
    const int lane = tid & 63, kh = lane >> 5, n = lane & 31;
    const int nch = K / KC;
    f32x4 st[NV];
    auto load = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + 256 * i, row = u / VPR, cv = u % VPR;
            st[i] = *(const f32x4*)((const char*)(Bg + (size_t)row * ldb + (size_t)ch * KC) + 16 * cv);
        }
    };
    auto store = [&](int buf) {
        E* wb = wbuf + buf * N * WS;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + 256 * i, row = u / VPR, cv = u % VPR;
            E* d = wb + row * WS + cv * EPV;
            if (MODE == 1) {
                *(f32x4*)d = st[i];                          // 16-byte aligned: WS * 2 B is a multiple of 16
            } else {
                float* df = (float*)d;
                df[0] = st[i][0]; df[1] = st[i][1]; df[2] = st[i][2]; df[3] = st[i][3];
            }
        }
    };
    load(0);
    store(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) load(ch + 1);
        const E* ap = a_row + aoff(ch) + M::KPL * kh;
        const E* bp = wbuf + (ch & 1) * N * WS + (nb0 * 32 + n) * WS + M::KPL * kh;
#pragma unroll
        for (int ks = 0; ks < KC / M::KSTEP; ++ks) {
            const typename M::Frag a = M::ld(ap + ks * M::KSTEP);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = M::mma(a, M::ld(bp + nb * 32 * WS + ks * M::KSTEP), acc[nb]);
        }
        if (ch + 1 < nch) store((ch + 1) & 1);
        __syncthreads();
    }
}

// ---- dropout keep bits of one (row block, 32-channel slice) unit, generic in C ---------------------------------------
// unit u = rb * (C / 32) + nb.  p == 0.5: ONE Philox draw (index (u >> 3) * 64 + lane, stream 32 + block) carries the
// 16-bit fields of 8 consecutive units - for C = 64 this is exactly the stream of the specialised kernels (philox.h).
// Other rates: the 8-bit stream, one draw per (row block, channel, dt) as in philox.h.  Mirrored by oracle/philox.py.
__device__ __forceinline__ uint32_t gen_keep16(int rb, int nb, int C, int lane, int block, uint64_t seed, uint32_t thr) {
    if (thr == 128u) {
        const uint32_t u = (uint32_t)rb * (uint32_t)(C >> 5) + (uint32_t)nb;
        const u32x4 o = philox_stream((u >> 3) * 64u + (uint32_t)lane, PHILOX_STREAM_1BIT + (uint32_t)block, seed);
        return philox_field16(o, (int)(u & 7u));
    }
    const int c = 32 * nb + (lane & 31);
    return philox_keep16(philox_stream((uint32_t)(rb * C + c), (uint32_t)(2 * block + (lane >> 5)), seed), thr);
}

// pixel index (into [B][H][W]) of pooled pixel q at (dt, df); Ho = H / 2, Wo = W / 4
__device__ __forceinline__ int gen_rb_pixel(int q, int dt, int df, int H, int W, int Ho, int Wo) {
    const int wo = q % Wo, t = q / Wo;
    const int ho = t % Ho, b = t / Ho;
    return (b * H + 2 * ho + dt) * W + 4 * wo + df;
}
