// conv.hip - 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, channels-last fp32, as an
// implicit GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
// Reference op: nn.Conv2d(64, 64, 3, 1, 1) of conv blocks 1 and 2 (baseline/models/CNN.py:46-47)
// and its autograd (dgrad / wgrad).  Images are [B][H][W][64] with W = 16 (block 1) or 4 (block 2),
// so one workgroup tile spans the full width: 128 pixels = TH rows x TW cols.
//
//  * forward / dgrad (k_conv3x3): GEMM M = pixels, N = 64, K = 9 taps x 64 channels.  The halo tile
//    ((TH+2) x (TW+2) pixels x 64 ch) is staged ONCE in LDS with a 65-float pixel stride and a row
//    stride == TW (mod 32) so the A-fragment reads (lane = pixel) are bank-conflict free; the
//    per-tap 64x64 weight slab is double-buffered in LDS.  Each of the 4 waves owns 32 pixels x 64
//    outputs (2 independent accumulators -> back-to-back MFMA issue).  The forward epilogue adds
//    the bias and accumulates the BatchNorm batch statistics (sum, sum of squares per channel) so
//    the conv output is read only once more, by the fused BN+GLU+pool kernel.  In dgrad mode the
//    loader applies the BatchNorm-backward affine dy = ca*dz + cb*y + cc on the fly, so dy is
//    never materialised.
//  * wgrad (k_conv3x3_wgrad): GEMM M = co, N = ci, K = pixels per tap.  Persistent workgroups keep
//    all 9 taps' 32x32 accumulators of their quadrant in registers (144 VGPRs), walk many tiles,
//    write one partial each, and k_wgrad_reduce sums the partials (deterministic, no atomics).
#include <type_traits>
#include "common.h"
#include "kernels.h"
SED_TS_DEFINE(conv)

template <int TW>
struct ConvCfg {
    static constexpr int TH = 128 / TW;
    static constexpr int HW = TW + 2;
    static constexpr int HH = TH + 2;
    static constexpr int PS = 65;
    static constexpr int RS = HW * PS + (((TW - HW * PS) % 32) + 32) % 32;
    static constexpr int HALO_FLOATS = HH * RS;
    static constexpr int W_FLOATS = 64 * 64;
    static constexpr size_t LDS_BYTES = (size_t)(HALO_FLOATS + 2 * W_FLOATS) * 4;
};

// (wino_u, conv_pack_body: kernels.h - the packing also rides along in the k_x_moments launch of a training forward)
__global__ __launch_bounds__(256) void k_conv_pack(ConvPackArgs a) { conv_pack_body(a, blockIdx.x * 256 + threadIdx.x); }

// MODE 0: forward (in = activations, epilogue bias + stats).  MODE 1: dgrad (in = affine(dz, y)).
// NS = 2 splits the 64 output channels over two workgroups (blockIdx.y): used for the narrow block-2
// images, whose B*ceil(H/32) tiles alone would leave half the chip idle.
template <int TW, int MODE, int NS>
__global__ __launch_bounds__(256) void k_conv3x3(const float* __restrict__ in0, const float* __restrict__ in1,
                                                  const float* __restrict__ coef, const float* __restrict__ wpk,
                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                  double* __restrict__ stat, int B, int H, int tiles_per_clip) {
    using Cfg = ConvCfg<TW>;
    constexpr int TH = Cfg::TH, HW = Cfg::HW, HH = Cfg::HH, PS = Cfg::PS, RS = Cfg::RS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;
    float* Ws = smem + Cfg::HALO_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int b = blockIdx.x / tiles_per_clip, y0 = (blockIdx.x % tiles_per_clip) * TH;
    const int W = TW;
    const int nh = (NS == 2) ? blockIdx.y : 0;      // which half of the output channels

    // ---- stage halo -------------------------------------------------------------------------
    for (int f = tid; f < HH * HW * 16; f += 256) {
        const int pix = f >> 4, c4 = (f & 15) * 4;
        const int hy = pix / HW, hx = pix % HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const size_t g = (((size_t)b * H + iy) * W + ix) * 64 + c4;
            v = *(const float4*)(in0 + g);
            if (MODE == 1) {
                const float4 yv = *(const float4*)(in1 + g);
                const float4 ca = *(const float4*)(coef + c4);
                const float4 cb = *(const float4*)(coef + 64 + c4);
                const float4 cc = *(const float4*)(coef + 128 + c4);
                v.x = ca.x * v.x + cb.x * yv.x + cc.x;
                v.y = ca.y * v.y + cb.y * yv.y + cc.y;
                v.z = ca.z * v.z + cb.z * yv.z + cc.z;
                v.w = ca.w * v.w + cb.w * yv.w + cc.w;
            }
        }
        float* dst = halo + hy * RS + hx * PS + c4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    // ---- stage tap 0 weights ----------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = tid + 256 * it;
        *(float4*)(Ws + f * 4) = *(const float4*)(wpk + f * 4);
    }
    __syncthreads();

    const int m = 32 * wv + n;
    const int ty = m / TW, tx = m % TW;
    const float* Ab = halo + (ty + 1) * RS + (tx + 1) * PS + kh;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    for (int tap = 0; tap < 9; ++tap) {
        float4 pre[4];
        if (tap < 8) {
#pragma unroll
            for (int it = 0; it < 4; ++it) pre[it] = *(const float4*)(wpk + (tap + 1) * 4096 + (tid + 256 * it) * 4);
        }
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const float* A = Ab + dy * RS + dx * PS;
        const float* Bw = Ws + (tap & 1) * 4096 + kh * 64 + n + 32 * nh;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a = A[2 * s];
            const float b0 = Bw[2 * s * 64];
            acc0 = mfma32(a, b0, acc0);
            if (NS == 1) {
                const float b1 = Bw[2 * s * 64 + 32];
                acc1 = mfma32(a, b1, acc1);
            }
        }
        if (tap < 8) {
            float* Wn = Ws + ((tap + 1) & 1) * 4096;
#pragma unroll
            for (int it = 0; it < 4; ++it) *(float4*)(Wn + (tid + 256 * it) * 4) = pre[it];
        }
        lds_barrier();
    }

    // ---- epilogue -----------------------------------------------------------------------------
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const float bia0 = (MODE == 0) ? bias[32 * nh + n] : 0.f, bia1 = (MODE == 0 && NS == 1) ? bias[32 + n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = 32 * wv + mfma32_row(r, lane);
        const int yy = y0 + mm / TW, xx = mm % TW;
        if (yy < H) {
            const size_t g = (((size_t)b * H + yy) * W + xx) * 64;
            const float v0 = acc0[r] + bia0, v1 = acc1[r] + bia1;
            out[g + 32 * nh + n] = v0;
            if (NS == 1) out[g + 32 + n] = v1;
            if (MODE == 0) { s1[0] += v0; s2[0] += v0 * v0; s1[1] += v1; s2[1] += v1 * v1; }
        }
    }
    if (MODE == 0 && stat != nullptr) {
        float* red = smem;   // reuse (all waves are past the last barrier of the main loop)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float a = s1[h] + __shfl_xor(s1[h], 32);
            float q = s2[h] + __shfl_xor(s2[h], 32);
            if (kh == 0) { red[(wv * 2 + 0) * 64 + 32 * h + n] = a; red[(wv * 2 + 1) * 64 + 32 * h + n] = q; }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;     // c = local channel slot: [0,32) first block, [32,64) second
            double v = 0;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) v += (double)red[(w2 * 2 + which) * 64 + c];
            if (NS == 1) atomicAdd(&stat[which * 64 + c], v);
            else if (c < 32) atomicAdd(&stat[which * 64 + 32 * nh + c], v);
        }
    }
}


// ---- direct (9-tap) weight-stationary persistent kernel for W = 16 -------------------------------------------
// Kept as the A/B baseline of the Winograd kernel below (debug bit 6); it was the block-1 forward / dgrad kernel until
// r01_i.  PMC on the tile kernel above (k_conv3x3<16,..>): MFMA pipe busy 55 % of the kernel, waves parked in
// s_waitcnt / s_barrier 38 % of their cycles - the per-tap weight slab hand-over (LDS double buffer + barrier) and the
// un-overlapped halo staging.  Here the WEIGHTS stay in registers for the whole life of a persistent workgroup:
//   * a wave owns 16 output channels and holds their full K = 9 x 64 weight panel as v_mfma_f32_16x16x4_f32 B
//     fragments: 144 VGPRs, loaded once; no LDS weight traffic and no barrier inside a tile;
//   * a tile is 8 image rows x 16 pixels; the next tile's halo is prefetched into registers while the current tile
//     computes and is written to the other LDS halo buffer afterwards (one barrier per tile); halo pixel stride 66
//     floats makes the A-fragment reads (lane = pixel x 4 channels) bank-conflict free;
//   * BatchNorm sums are accumulated in registers across all tiles of the workgroup: 32 atomics per wave per launch;
//   * every SIMD hosts TWO waves with the same 16-channel panel (<= 256 registers per wave): the "early" wave owns
//     image rows 0-3 of the tile and stores its outputs right after its MFMAs, the "late" wave owns rows 4-7 and
//     stores its outputs at the START of the next tile - whenever one of the two is in its epilogue the other one
//     is issuing MFMAs (a 4-wave version with one wave per SIMD spent 2.8 us of every 21.4 us tile outside MFMAs;
//     moving that work into the MFMA groups lost: the waits the compiler attaches to interleaved memory operations
//     stall the in-order MFMA stream).  Halo staging is shared by all 512 threads (6 items each).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct Ws16 {
    static constexpr int TH = 8, TW = 16, HW = 18, HH = 10, PS = 66, RS = HW * PS;
    static constexpr int HALO_FLOATS = HH * RS;
    static constexpr size_t LDS_BYTES = (size_t)2 * HALO_FLOATS * 4;
};
struct Ws16x2 {
    static constexpr int NLD = (Ws16::HH * Ws16::HW * 16 + 511) / 512;     // float4 loads per thread per halo (6)
};
template <int MODE>
__global__ __launch_bounds__(512, 1) void k_conv16_ws2(const float* __restrict__ in0, const float* __restrict__ in1,
                                                       const float* __restrict__ coef, const float* __restrict__ wpk,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       double* __restrict__ stat, int H, int tiles_per_clip, int n_tiles) {
    using C = Ws16;
    constexpr int NLD = Ws16x2::NLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = wave & 3, rh = wave >> 2;                 // channel group (16 channels), row half (4 image rows)
    const int p16 = lane & 15, kq = lane >> 4;
    float st1 = 0.f, st2 = 0.f;
    // dgrad stages two tensors (dz, y) per halo: it fetches the halo in two halves (items [0, NLD/2) before the MFMA loop,
    // the rest from its middle) so that only NLD/2 float4 pairs are live at a time - all at once spilled (80 B of scratch)
    constexpr int NH = (MODE == 1) ? NLD / 2 : NLD;
    f32x4_t pre0[NH], pre1[MODE == 1 ? NH : 1];
    auto load_halo = [&](int tile, int part) {
        const int b = tile / tiles_per_clip, y0 = (tile % tiles_per_clip) * C::TH;
        const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < NH; ++k2) {
            const int it = part * NH + k2;
            const int f = tid + 512 * it;
            const int pix = f >> 4, c4 = (f & 15) * 4;
            const int hy = pix / C::HW, hx = pix % C::HW;
            const int iy = y0 - 1 + hy, ix = hx - 1;
            const bool ok = (f < C::HH * C::HW * 16) && iy >= 0 && iy < H && ix >= 0 && ix < C::TW;
            const size_t g = ok ? (((size_t)b * H + iy) * C::TW + ix) * 64 + c4 : 0;
            const f32x4_t v = *(const f32x4_t*)(in0 + g);
            pre0[k2] = ok ? v : z4;
            if (MODE == 1) { const f32x4_t w = *(const f32x4_t*)(in1 + g); pre1[k2] = ok ? w : z4; }
        }
    };
    auto store_halo = [&](float* halo, int tile, int part) {
        const int y0 = (tile % tiles_per_clip) * C::TH;
#pragma unroll
        for (int k2 = 0; k2 < NH; ++k2) {
            const int it = part * NH + k2;
            const int f = tid + 512 * it;
            if (f >= C::HH * C::HW * 16) continue;
            const int pix = f >> 4, c4 = (f & 15) * 4;
            const int hy = pix / C::HW, hx = pix % C::HW;
            f32x4_t v = pre0[k2];
            if (MODE == 1) {
                const int iy = y0 - 1 + hy, ix = hx - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < C::TW) {      // padding stays exactly 0
                    const f32x4_t ca = *(const f32x4_t*)(coef + c4), cb = *(const f32x4_t*)(coef + 64 + c4), cc = *(const f32x4_t*)(coef + 128 + c4);
                    v = ca * v + cb * pre1[k2] + cc;
                }
            }
            float* d = halo + hy * C::RS + hx * C::PS + c4;          // 8-byte aligned (PS even, c4 % 4 == 0)
            *(float2*)d = make_float2(v[0], v[1]);
            *(float2*)(d + 2) = make_float2(v[2], v[3]);
        }
    };
    int tile = blockIdx.x;
    if (tile < n_tiles) load_halo(tile, 0);        // first halo in flight while the weight panel is fetched
    float bw[9][16];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) bw[t][s4] = wpk[t * 4096 + (4 * s4 + kq) * 64 + 16 * cg + p16];
    const float bia = (MODE == 0) ? bias[16 * cg + p16] : 0.f;
    if (tile < n_tiles) {
        store_halo(smem, tile, 0);
        if (MODE == 1) { load_halo(tile, 1); store_halo(smem, tile, 1); }
    }
    __syncthreads();
    // output rows 4 rh .. 4 rh + 3 of a tile; D[i][j]: j = lane & 15 -> channel 16 cg + j, i = 4 (lane >> 4) + r -> pixel x
    auto epilogue = [&](const f32x4_t (&a)[4], int t2) {
        const int b = t2 / tiles_per_clip, y0 = (t2 % tiles_per_clip) * C::TH + 4 * rh;
#pragma unroll
        for (int rbl = 0; rbl < 4; ++rbl) {
            const int yy = y0 + rbl;
            if (yy < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = a[rbl][r] + bia;
                    out[(((size_t)b * H + yy) * C::TW + 4 * kq + r) * 64 + 16 * cg + p16] = v;
                    if (MODE == 0) { st1 += v; st2 += v * v; }
                }
            }
        }
    };
    f32x4_t acc[4];                                          // lives across iterations: the late wave stores it one tile later
    int ptile = -1;
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const int nxt_tile = tile + gridDim.x;
        if (rh == 1 && ptile >= 0) epilogue(acc, ptile);     // late wave: previous tile's outputs, under the early wave's MFMAs
        if (nxt_tile < n_tiles) load_halo(nxt_tile, 0);
        const float* halo = smem + cur * C::HALO_FLOATS;
        const float* Ab = halo + (1 + 4 * rh) * C::RS + (1 + p16) * C::PS + kq;   // pixel (row 4 rh, x = p16), channel kq
#pragma unroll
        for (int rbl = 0; rbl < 4; ++rbl) acc[rbl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // 72 groups of 8 MFMAs (one tap, two 4-channel K steps, 4 image rows); A fragments of group g+1 are read from
        // LDS before the MFMAs of group g are issued (register double buffer pinned with sched_barrier)
        auto load_a = [&](float (&a)[8], int gi) {
            const int t = gi / 8, sp = gi % 8;
            const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rbl = 0; rbl < 4; ++rbl) a[q * 4 + rbl] = Ab[(rbl + dy) * C::RS + dx * C::PS + 4 * (2 * sp + q)];
        };
        auto mma = [&](const float (&a)[8], int gi) {
            const int t = gi / 8, sp = gi % 8;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rbl = 0; rbl < 4; ++rbl)
                    acc[rbl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * 4 + rbl], bw[t][2 * sp + q], acc[rbl], 0, 0, 0);
        };
        float a0[8], a1[8];
        load_a(a0, 0);
#pragma unroll
        for (int gi = 0; gi < 72; gi += 2) {
            load_a(a1, gi + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, gi);
            if (MODE == 1 && gi == 36 && nxt_tile < n_tiles) {      // first half of the next halo -> LDS, second half in flight
                store_halo(smem + (cur ^ 1) * C::HALO_FLOATS, nxt_tile, 0);
                load_halo(nxt_tile, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < 72) load_a(a0, gi + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, gi + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nxt_tile < n_tiles) store_halo(smem + (cur ^ 1) * C::HALO_FLOATS, nxt_tile, MODE == 1 ? 1 : 0);
        if (rh == 0) epilogue(acc, tile);                     // early wave: now, under the late wave's MFMAs
        ptile = tile;
        lds_barrier();
        cur ^= 1;
    }
    if (rh == 1 && ptile >= 0) epilogue(acc, ptile);
    if (MODE == 0 && stat != nullptr) {
        st1 += __shfl_xor(st1, 16); st1 += __shfl_xor(st1, 32);
        st2 += __shfl_xor(st2, 16); st2 += __shfl_xor(st2, 32);
        if (kq == 0) {
            atomicAdd(&stat[16 * cg + p16], (double)st1);
            atomicAdd(&stat[64 + 16 * cg + p16], (double)st2);
        }
    }
}

// ---- Winograd F(2x2, 3x3) variant of the 8-wave weight-stationary kernel -------------------------------------------------
// The MFMA pipe is what bounds the block-1 convolutions, so do fewer multiplies: Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A
// needs 16 multiplies per 2x2 output block and (ci, co) pair instead of 36 - 2.25x less MFMA work, at the price of 12 LDS
// reads + ~30 adds per 8 MFMAs for the input transform (done on the fly from the raw halo tile in LDS) and a 4-value
// inverse transform per block in the epilogue.  Layout of the work:
//   * tile = 8 image rows x 16 columns = 32 blocks of 2x2; an MFMA tile (16x16x4) is 16 blocks x 16 output channels,
//     K = 4 input channels; there is one such product per transform position (16 of them);
//   * 8 waves: wave (cg, ph) owns output channels [16 cg, 16 cg + 16) and the 8 positions of transform rows 2 ph, 2 ph + 1
//     - its share of the transformed weights is 8 x 64 x 16 floats = 128 VGPRs, loaded once (weight-stationary);
//   * the two waves of a channel group (same SIMD) each inverse-transform their own positions into a PARTIAL 2x2 output
//     (the transform is linear); the ph = 1 wave hands its partial over through LDS and moves on to the next tile, the
//     ph = 0 wave adds, applies the bias, accumulates the BatchNorm sums and stores.
// Rounding: the transforms use only additions and halvings; results differ from the direct kernel at the 1e-7 level
// (parity tests unchanged).
// Tried: wave = one transform row (4 positions) x 32 output channels instead of 8 positions x 16 channels - half the
// input-transform work per wave (8 VALU + 8 LDS reads per 8 MFMAs; tools/ubench/wino_loop.cpp: 105 -> 119 TFLOP/s for the
// bare loop) and the epilogue split evenly over the 8 waves, at the price of a 4-way exchange of the inverse-transform
// partials (12 values out, 12 in per lane and MFMA tile).  Bit-correct, but no faster: 58.9 / 57.0 us vs 59.3 / 56.8 us.
//   * TW = 4 (block 2: 157 x 4 images): one MFMA tile per workgroup tile (16 image rows x 4 columns), 240 tiles at
//     B = 24 - one per workgroup, so the launch is mostly the weight prologue; still 2.5x faster than the 9-tap tile kernel.
template <int TW_>
struct Wino {
    static constexpr int TW = TW_, BC = TW / 2;                          // blocks per image row
    static constexpr int NMT = (TW == 16) ? 2 : 1;                       // MFMA tiles (16 blocks) per workgroup tile
    static constexpr int TH = 2 * (16 * NMT / BC);                       // image rows per tile: 8 (TW 16), 16 (TW 4)
    static constexpr int HW = TW + 2, HH = TH + 2, PS = 66, RS = HW * PS;
    static constexpr int HALO_FLOATS = HH * RS, HALO_F4 = HH * HW * 16;
    // (The A-operand reads are 2-way bank-conflicted within a half-wave: bank = kq + 4 bc + 8 br.  A 2-float skew on
    // alternate row pairs makes them conflict-free - PMC 198 k -> 14 k conflict cycles - but the kernel is not LDS-bound:
    // same duration, and the extra address arithmetic spilled the dgrad variant.  Not kept.)
    __host__ __device__ static constexpr int row_off(int hy) { return hy * RS; }
    static constexpr int XCH_FLOATS = 4 * NMT * 16 * 64;                 // [cg][mtile][16 partial outputs][lane]
    static constexpr size_t LDS_BYTES = (size_t)(2 * HALO_FLOATS + XCH_FLOATS + 192 + 4) * 4;   // + coefficients + dump slot
    static constexpr int STEPS = 16 * NMT;                               // k-steps (4 input channels each) per tile
};
// piece p of the halo prefetch is issued at k-step p * STEPS / PARTS of the tile and stored to LDS right before the next one
__host__ __device__ constexpr int wino_piece_at(int gs, int steps, int parts) {
    for (int p2 = 1; p2 < parts; ++p2)
        if (p2 * steps / parts == gs) return p2;
    return 0;
}
template <int TW, int MODE>
__global__ __launch_bounds__(512, 1) void k_conv_wino(const float* __restrict__ in0, const float* __restrict__ in1,
                                                        const float* __restrict__ coef, const float* __restrict__ U,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        double* __restrict__ stat, int H, int tiles_per_clip, int n_tiles,
                                                        BnBwdPrepArgs prep) {
    using C = Wino<TW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xch = smem + 2 * C::HALO_FLOATS;
    float* dump = xch + C::XCH_FLOATS + 192;
    float* cfs = xch + C::XCH_FLOATS;                   // dgrad: the three BatchNorm-backward coefficient rows
    if (MODE == 1) {
        if (prep.acc != nullptr) {     // BatchNorm-backward coefficients straight from the reduction sums (no k_bn_bwd_prep)
            if (threadIdx.x < 64) {
                float ca, cb, cc;
                bn_bwd_coef(prep, threadIdx.x, ca, cb, cc);
                cfs[threadIdx.x] = ca; cfs[64 + threadIdx.x] = cb; cfs[128 + threadIdx.x] = cc;
            }
            if (blockIdx.x == 0) bn_bwd_prep_body(prep, threadIdx.x, 512);      // the block's parameter gradients (+ coef)
        } else if (threadIdx.x < 192) {
            cfs[threadIdx.x] = coef[threadIdx.x];
        }
        __syncthreads();
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, ph = wave >> 2;
    const int i16 = lane & 15, kq = lane >> 4;
    float st1 = 0.f, st2 = 0.f;
    // the next tile's halo is fetched in PARTS pieces of NH float4 per thread (registers are the scarce resource here;
    // dgrad fetches two arrays): 2880 float4 = 5.6 per thread for TW = 16, 1728 = 3.4 for TW = 4
    constexpr int NLD = (C::HALO_F4 + 511) / 512;
    constexpr int NH = (MODE == 1) ? 1 : (NLD + 1) / 2, PARTS = (NLD + NH - 1) / NH;
    f32x4_t pre0[NH], pre1[MODE == 1 ? NH : 1];
    // (Row-wise staging as in k_wgrad_wino - thread = column x 4 channels, ~10 instead of ~25 VALU per item, scalar-base
    // loads, but 10 one-row items in 5 pieces instead of 6 in 2 - measured no better here: 60 / 58.5 vs 59 / 56.5 us.)
    // raw loads only (padding is applied when the values go to LDS: a select here would make the compiler wait for the
    // loads on the spot); item it = float4 number tid + 512 it of the halo
    auto load_item = [&](int tile, int it, f32x4_t& d0, f32x4_t& d1) {
        const int b = tile / tiles_per_clip, y0 = (tile % tiles_per_clip) * C::TH;
        const int f = tid + 512 * it;
        const int pix = f >> 4, c4 = (f & 15) * 4;
        const int hy = pix / C::HW, hx = pix % C::HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = (f < C::HALO_F4) && iy >= 0 && iy < H && ix >= 0 && ix < C::TW;
        const uint32_t g = ok ? (uint32_t)(((b * H + iy) * C::TW + ix) * 64 + c4) : 0u;   // < 2^31 floats (checked by the launcher)
        d0 = *(const f32x4_t*)(in0 + g);
        if (MODE == 1) d1 = *(const f32x4_t*)(in1 + g);
    };
    auto store_item = [&](float* halo, int tile, int it, const f32x4_t& s0, const f32x4_t& s1) {
        if (512 * it >= C::HALO_F4) return;                            // (compile-time) nothing left
        const int y0 = (tile % tiles_per_clip) * C::TH;
        const int f = tid + 512 * it;
        const int pix = f >> 4, c4 = (f & 15) * 4;
        const int hy = pix / C::HW, hx = pix % C::HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < C::TW;
        f32x4_t v = s0;
        if (MODE == 1) {
            const f32x4_t ca = *(const f32x4_t*)(cfs + c4), cb = *(const f32x4_t*)(cfs + 64 + c4), cc = *(const f32x4_t*)(cfs + 128 + c4);
            v = ca * v + cb * s1 + cc;
        }
        const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
        v = ok ? v : z4;                                               // padding stays exactly 0
        // lanes past the end of the halo write to a dump slot: a conditional store would let the compiler sink the
        // global load into the branch, i.e. issue it here and wait for it on the spot
        float* d = (f < C::HALO_F4) ? halo + C::row_off(hy) + hx * C::PS + c4 : dump;
        *(float2*)d = make_float2(v[0], v[1]);
        *(float2*)(d + 2) = make_float2(v[2], v[3]);
    };
    auto load_halo = [&](int tile, int part) {
#pragma unroll
        for (int k2 = 0; k2 < NH; ++k2) load_item(tile, part * NH + k2, pre0[k2], pre1[MODE == 1 ? k2 : 0]);
    };
    auto store_halo = [&](float* halo, int tile, int part) {
#pragma unroll
        for (int k2 = 0; k2 < NH; ++k2) store_item(halo, tile, part * NH + k2, pre0[k2], pre1[MODE == 1 ? k2 : 0]);
    };
    const TileWalk walk = xcd_walk(n_tiles);
    int tile = walk.first;
    TS(0);
    // the first halo is fetched whole, together with the weights (the accumulators are not live yet, so there are
    // registers for all of it; piecewise it cost PARTS load round trips - 7 us of prologue for dgrad)
    // (dgrad, two arrays: in two rounds - all 12 float4 at once pushed loop-invariant values into scratch)
    constexpr int FR = (MODE == 1) ? 2 : 1, FN = (NLD + FR - 1) / FR;
    f32x4_t first0[FN], first1[MODE == 1 ? FN : 1];
    if (tile < walk.end) {
#pragma unroll
        for (int it = 0; it < FN; ++it) load_item(tile, it, first0[it], first1[MODE == 1 ? it : 0]);
    }
    // this wave's transformed weights, B[k = ci][n = co]: p8 = 0..3 <-> transform row 3 ph (the "X" row below), column p8;
    // p8 = 4..7 <-> row 1 + ph (the "Y" row), column p8 - 4; k_conv_pack wrote them in register order (float4 per lane)
    float uw[8][16];
    {
        const f32x4_t* Uf = (const f32x4_t*)U + (size_t)(ph * 4 + cg) * 8 * 4 * 64 + lane;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8)
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                const f32x4_t u4 = Uf[(p8 * 4 + sq) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) uw[p8][4 * sq + e] = u4[e];
            }
    }
    const float bia = (MODE == 0) ? bias[16 * cg + i16] : 0.f;
    if (tile < walk.end) {
#pragma unroll
        for (int fr = 0; fr < FR; ++fr) {
            if (fr > 0) {
#pragma unroll
                for (int it = 0; it < FN; ++it) load_item(tile, fr * FN + it, first0[it], first1[MODE == 1 ? it : 0]);
            }
#pragma unroll
            for (int it = 0; it < FN; ++it) store_item(smem, tile, fr * FN + it, first0[it], first1[MODE == 1 ? it : 0]);
        }
    }
    TS(1);
    __syncthreads();
    TS(2);
    TSC(14);
    int it_ts = 0;
    // input transform, rows: with patch rows (A, B, C) = (d0, d1, d2) for ph = 0 and (d3, d2, d1) for ph = 1
    //   X = A - C      = T0 = d0 - d2            |  d3 - d1 = -T3
    //   Y = B + sg C   = T1 = d1 + d2  (sg = 1)  |  d2 - d1 =  T2  (sg = -1)
    // the sign of T3 is undone in the inverse transform (rows of A^T = [1 1 1 0; 0 1 -1 -1]):
    //   ph = 0: out row 0 = q0 + q1 = qx + qy, out row 1 = q1 = qy;   ph = 1: row 0 = q2 = qy, row 1 = -q2 - q3 = qx - qy
    const float sg = ph ? -1.f : 1.f;
    int cur = 0;
    for (; tile < walk.end; tile += walk.step) {
        // the prefetch is unconditional (the last iteration re-fetches its own tile into the idle buffer): with the loads and
        // the LDS stores under separate `if (has next)` the compiler must assume a load may still be in flight when its
        // registers are written again and waits there - behind the epilogue's global stores
        const int nxt_tile = (tile + walk.step < walk.end) ? tile + walk.step : tile;
        const float* halo = smem + cur * C::HALO_FLOATS;
        float* halo_nxt = smem + (cur ^ 1) * C::HALO_FLOATS;
        const int b = tile / tiles_per_clip, y0 = (tile % tiles_per_clip) * C::TH;
        load_halo(nxt_tile, 0);
#pragma unroll
        for (int mt = 0; mt < C::NMT; ++mt) {
            // block of this lane as an MFMA row: blk = 16 mt + i16 -> (br, bc)
            const int blk = 16 * mt + i16, br = blk / C::BC, bc = blk % C::BC;
            const float* Pa = halo + C::row_off(2 * br + 3 * ph) + (2 * bc) * C::PS + kq;
            const float* Pm = halo + C::row_off(2 * br + 1 + ph) + (2 * bc) * C::PS + kq;
            const float* Pc = halo + C::row_off(2 * br + 2 - ph) + (2 * bc) * C::PS + kq;
            f32x4_t acc[8];
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) acc[p8] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            float r[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) { r[q] = Pa[q * C::PS]; r[4 + q] = Pm[q * C::PS]; r[8 + q] = Pc[q * C::PS]; }
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                float va[4], vb[4];
                {
                    float X[4], Y[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        X[q] = r[q] - r[8 + q];
                        Y[q] = fmaf(sg, r[8 + q], r[4 + q]);
                    }
                    va[0] = X[0] - X[2]; va[1] = X[1] + X[2]; va[2] = X[2] - X[1]; va[3] = X[1] - X[3];
                    vb[0] = Y[0] - Y[2]; vb[1] = Y[1] + Y[2]; vb[2] = Y[2] - Y[1]; vb[3] = Y[1] - Y[3];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s4 + 1 < 16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        r[q] = Pa[q * C::PS + 4 * (s4 + 1)]; r[4 + q] = Pm[q * C::PS + 4 * (s4 + 1)]; r[8 + q] = Pc[q * C::PS + 4 * (s4 + 1)];
                    }
                }
                {
                    const int piece = wino_piece_at(16 * mt + s4 + 1, C::STEPS, PARTS);   // (folds: mt, s4 are unrolled)
                    if (piece > 0) {
                        store_halo(halo_nxt, nxt_tile, piece - 1);
                        load_halo(nxt_tile, piece);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[j], uw[j][s4], acc[j], 0, 0, 0);
                    acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[j], uw[4 + j][s4], acc[4 + j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it_ts == 0) TS(3 + 4 * mt);
            if (it_ts == 0 && mt == 0) TSC(15);
            if (mt == C::NMT - 1) store_halo(halo_nxt, nxt_tile, PARTS - 1);
            // partial inverse transform of this wave's 8 positions: D register q <-> block 4 kq + q of the MFMA tile
            float yp[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float qx0 = acc[0][q] + acc[1][q] + acc[2][q], qx1 = acc[1][q] - acc[2][q] - acc[3][q];
                const float qy0 = acc[4][q] + acc[5][q] + acc[6][q], qy1 = acc[5][q] - acc[6][q] - acc[7][q];
                yp[q][0] = ph ? qy0 : qx0 + qy0;                // out[0][0]
                yp[q][1] = ph ? qy1 : qx1 + qy1;                // out[0][1]
                yp[q][2] = ph ? qx0 - qy0 : qy0;                // out[1][0]
                yp[q][3] = ph ? qx1 - qy1 : qy1;                // out[1][1]
            }
            float* xw = xch + ((cg * C::NMT + mt) * 16) * 64 + lane;
            if (ph == 1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) xw[q * 64] = yp[q >> 2][q & 3];
            }
            if (it_ts == 0) TS(4 + 4 * mt);
            lds_barrier();            // partials of the ph = 1 waves visible (mt = 1: next halo complete as well)
            if (it_ts == 0) TS(5 + 4 * mt);
            if (ph == 0) {
                auto emit = [&](auto checked) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ob = 16 * mt + 4 * kq + q, obr = ob / C::BC, obc = ob % C::BC;
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const int yy = y0 + 2 * obr + (o >> 1), xx = 2 * obc + (o & 1);
                            if (!decltype(checked)::value || yy < H) {
                                const float v = yp[q][o] + xw[(q * 4 + o) * 64] + bia;
                                out[(((size_t)b * H + yy) * C::TW + xx) * 64 + 16 * cg + i16] = v;
                                if (MODE == 0) { st1 += v; st2 += v * v; }
                            }
                        }
                    }
                };
                if (y0 + C::TH <= H) emit(std::false_type{});      // whole tile inside the image: straight-line stores
                else emit(std::true_type{});
            }
        }
        if (it_ts == 0) TS(6 + 4);
        if (it_ts == 1) TS(11);
        ++it_ts;
        cur ^= 1;
    }
    TS(12);
    if (MODE == 0 && stat != nullptr && ph == 0) {
        st1 += __shfl_xor(st1, 16); st1 += __shfl_xor(st1, 32);
        st2 += __shfl_xor(st2, 16); st2 += __shfl_xor(st2, 32);
        if (kq == 0) {
            atomicAdd(&stat[16 * cg + i16], (double)st1);
            atomicAdd(&stat[64 + 16 * cg + i16], (double)st2);
        }
    }
    TS(13);
}

template <int MODE>
static int conv16_ws_launch(const float* in0, const float* in1, const float* coef, const float* wpk, const float* bias,
                            float* out, double* stat, int B, int H, hipStream_t st) {
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv16_ws2<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)Ws16::LDS_BYTES));
    }
    const int tpc = (H + Ws16::TH - 1) / Ws16::TH, nt = B * tpc;
    const int grid = nt < 256 ? nt : 256;          // one persistent workgroup per CU
    k_conv16_ws2<MODE><<<grid, 512, Ws16::LDS_BYTES, st>>>(in0, in1, coef, wpk, bias, out, stat, H, tpc, nt);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// Winograd F(2x2, 3x3): the default for both 64 -> 64 convolutions.  Block 1: 59 / 57 us per launch (forward / dgrad)
// against 91 / 90 us for the direct 8-wave kernel; bit 6 of the debug knob selects the direct kernels.
template <int TW, int MODE>
static int conv_wino_launch(const float* in0, const float* in1, const float* coef, const float* wpk, const float* bias,
                            float* out, double* stat, int B, int H, const BnBwdPrepArgs* prep, hipStream_t st) {
    BnBwdPrepArgs pa = {};
    if (prep) pa = *prep;
    using C = Wino<TW>;
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_wino<TW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)C::LDS_BYTES));
    }
    SED_CHECK_ARG((size_t)B * H * TW * 64 < ((size_t)1 << 31), "conv: image too large for 32-bit offsets");
    const int tpc = (H + C::TH - 1) / C::TH, nt = B * tpc;
    const int grid = nt < 256 ? nt : 256;          // one persistent workgroup per CU
    k_conv_wino<TW, MODE><<<grid, 512, C::LDS_BYTES, st>>>(in0, in1, coef, wpk + SED_WINO_OFF, bias, out, stat, H, tpc, nt, pa);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// ---- wgrad ------------------------------------------------------------------------------------
template <int TW>
struct WgCfg {
    static constexpr int TH = 128 / TW;
    static constexpr int HW = TW + 2;
    static constexpr int HH = TH + 2;
    static constexpr int XH_FLOATS = HH * HW * 64;
    static constexpr int DY_FLOATS = 128 * 64;
    static constexpr size_t LDS_BYTES = (size_t)(XH_FLOATS + DY_FLOATS) * 4;
};

// TS = 3 splits the 9 taps (by kernel row) over three workgroups (blockIdx.y) writing disjoint tap
// slices of the same partial slab: 3x the parallelism for the narrow block-2 images.
template <int TW, int TS>
__global__ __launch_bounds__(256) void k_conv3x3_wgrad(const float* __restrict__ dz, const float* __restrict__ yin,
                                                        const float* __restrict__ coef, const float* __restrict__ xin,
                                                        float* __restrict__ part, int B, int H, int tiles_per_clip,
                                                        int n_tiles, BnBwdPrepArgs prep) {
    using Cfg = WgCfg<TW>;
    constexpr int TH = Cfg::TH, HW = Cfg::HW, HH = Cfg::HH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ __attribute__((aligned(16))) float cfl[192];       // BatchNorm-backward coefficients (from `coef` or derived here)
    if (prep.acc != nullptr) {
        if (threadIdx.x < 64) {
            float ca, cb, cc;
            bn_bwd_coef(prep, threadIdx.x, ca, cb, cc);
            cfl[threadIdx.x] = ca; cfl[64 + threadIdx.x] = cb; cfl[128 + threadIdx.x] = cc;
        }
    } else if (threadIdx.x < 192) {
        cfl[threadIdx.x] = coef[threadIdx.x];
    }
    float* xh = smem;
    float* dyt = smem + Cfg::XH_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int cob = wv >> 1, cib = wv & 1;
    const int W = TW;
    constexpr int NT = 9 / TS;                       // taps per workgroup
    const int tap0 = (TS == 3) ? 3 * blockIdx.y : 0;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    TS(0); TSC(14);
    int ts_k = 1;
    (void)ts_k;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip, y0 = (tile % tiles_per_clip) * TH;
        __syncthreads();
        if (ts_k < 12) { TS(ts_k); ++ts_k; }
        for (int f = tid; f < HH * HW * 16; f += 256) {
            const int pix = f >> 4, c4 = (f & 15) * 4;
            const int hy = pix / HW, hx = pix % HW;
            const int iy = y0 - 1 + hy, ix = hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const float4*)(xin + (((size_t)b * H + iy) * W + ix) * 64 + c4);
            *(float4*)(xh + pix * 64 + c4) = v;
        }
        for (int f = tid; f < 128 * 16; f += 256) {
            const int pix = f >> 4, c4 = (f & 15) * 4;
            const int yy = y0 + pix / TW, xx = pix % TW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy < H) {
                const size_t g = (((size_t)b * H + yy) * W + xx) * 64 + c4;
                const float4 d = *(const float4*)(dz + g);
                const float4 yv = *(const float4*)(yin + g);
                const float4 ca = *(const float4*)(cfl + c4);
                const float4 cb = *(const float4*)(cfl + 64 + c4);
                const float4 cc = *(const float4*)(cfl + 128 + c4);
                v.x = ca.x * d.x + cb.x * yv.x + cc.x;
                v.y = ca.y * d.y + cb.y * yv.y + cc.y;
                v.z = ca.z * d.z + cb.z * yv.z + cc.z;
                v.w = ca.w * d.w + cb.w * yv.w + cc.w;
            }
            *(float4*)(dyt + pix * 64 + c4) = v;
        }
        __syncthreads();
        if (ts_k < 12) { TS(ts_k); ++ts_k; }
        const float* Ab = dyt + kh * 64 + 32 * cob + n;
        const float* Bb = xh + kh * 64 + 32 * cib + n;
        for (int ty = 0; ty < TH; ++ty) {
#pragma unroll
            for (int txp = 0; txp < TW / 2; ++txp) {
                const float a = Ab[(ty * TW + 2 * txp) * 64];
#pragma unroll
                for (int tl = 0; tl < NT; ++tl) {
                    const int dy = (TS == 3) ? (int)blockIdx.y - 1 : tl / 3 - 1, dx = tl % 3 - 1;
                    const float bv = Bb[((ty + 1 + dy) * HW + 2 * txp + 1 + dx) * 64];
                    acc[tl] = mfma32(a, bv, acc[tl]);
                }
            }
        }
    }
    TS(12);
    float* dst = part + (size_t)blockIdx.x * 9 * 4096;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * cob + mfma32_row(r, lane);
            dst[(tap0 + tl) * 4096 + co * 64 + 32 * cib + n] = acc[tl][r];
        }
    TS(13); TSC(15);
}

#ifdef SED_AB   // A/B baseline kernels: only in `make EXTRA=-DSED_AB` builds (the shipped library carries the product path only)
// ---- block-1 wgrad with the staging hidden under the MFMAs ---------------------------------------------------
// Phase timestamps of the kernel above at B = 24 (tools/ts_kernel.py): 18 us of MFMAs per tile (95 % of the
// pipe's rate) but 7.7 us of staging in front of every tile - all 256 workgroups stage at the same moment, so the
// 28 MB burst runs at HBM speed while the memory system idles during the MFMA phases.  Here both LDS buffers
// exist twice (2 x 77 KB = 154 KB of the CU's 160 KB) and the next tile arrives in 8 register-staged slices, one
// per image row of MFMAs: slice c is LOADED during row c and written to the other LDS buffer (with the
// BatchNorm-backward affine applied) at the start of row c+1, i.e. 2 us later; one LDS-only barrier per tile.
// (An 8-wave variant - two waves per SIMD splitting the 9 taps 5 + 4, the recipe that helped the forward / dgrad kernels -
// measured exactly the same, alone and in the step, and was dropped.)
struct Wg16 {
    static constexpr int TH = 8, TW = 16, HW = 18, HH = 10;
    static constexpr int XH_FLOATS = HH * HW * 64, DY_FLOATS = 128 * 64, BUF_FLOATS = XH_FLOATS + DY_FLOATS;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * 4;
    static constexpr int HALO_F4 = HH * HW * 16, HALO_SLICE = HALO_F4 / 8;     // 2880 float4, 360 per slice
};
struct Wg16Slice { f32x4 h0, h1, d, y; };     // clang ext vectors: HIP's float4 struct did not stay in registers here
__device__ __forceinline__ Wg16Slice wg16_load(const float* __restrict__ dz, const float* __restrict__ yin,
                                               const float* __restrict__ xin, int H, int tiles_per_clip, int tile, int c, int tid) {
    using C = Wg16;
    Wg16Slice s;
    const int b = tile / tiles_per_clip, y0 = (tile % tiles_per_clip) * C::TH;
    const int c4 = (tid & 15) * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    {
        const int f = c * C::HALO_SLICE + tid;
        const int pix = f >> 4, hy = pix / C::HW, hx = pix % C::HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < C::TW;
        const f32x4 v = *(const f32x4*)(xin + (ok ? (((size_t)b * H + iy) * C::TW + ix) * 64 + (f & 15) * 4 : (size_t)0));
        s.h0 = ok ? v : z4;
    }
    {   // threads past the slice re-load its last element (never stored): no divergent assignment
        const int f = c * C::HALO_SLICE + 256 + (tid < C::HALO_SLICE - 256 ? tid : C::HALO_SLICE - 257);
        const int pix = f >> 4, hy = pix / C::HW, hx = pix % C::HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < C::TW;
        const f32x4 v = *(const f32x4*)(xin + (ok ? (((size_t)b * H + iy) * C::TW + ix) * 64 + (f & 15) * 4 : (size_t)0));
        s.h1 = ok ? v : z4;
    }
    {
        const int yy = y0 + c, xx = tid >> 4;
        const size_t g = (yy < H) ? (((size_t)b * H + yy) * C::TW + xx) * 64 + c4 : (size_t)0;   // masked in wg16_store
        s.d = *(const f32x4*)(dz + g);
        s.y = *(const f32x4*)(yin + g);
    }
    return s;
}
__device__ __forceinline__ void wg16_store(const Wg16Slice& s, float* buf, int H, int tiles_per_clip, int tile, int c, int tid,
                                           const f32x4& ca, const f32x4& cb, const f32x4& cc) {
    using C = Wg16;
    const int y0 = (tile % tiles_per_clip) * C::TH;
    *(f32x4*)(buf + (size_t)(c * C::HALO_SLICE + tid) * 4) = s.h0;          // xh[pix*64 + c4] with f = pix*16 + c4/4
    if (tid < C::HALO_SLICE - 256) *(f32x4*)(buf + (size_t)(c * C::HALO_SLICE + 256 + tid) * 4) = s.h1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y0 + c < H) v = ca * s.d + cb * s.y + cc;                          // rows past the image stay exactly 0
    *(f32x4*)(buf + C::XH_FLOATS + (size_t)(c * 256 + tid) * 4) = v;        // dyt[pix*64 + c4], pix = 16c + tid/16
}
__global__ __launch_bounds__(256) void k_wgrad16_db(const float* __restrict__ dz, const float* __restrict__ yin,
                                                     const float* __restrict__ coef, const float* __restrict__ xin,
                                                     float* __restrict__ part, int H, int tiles_per_clip, int n_tiles) {
    using C = Wg16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int cob = wv >> 1, cib = wv & 1;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int c4 = (tid & 15) * 4;
    const f32x4 ca = *(const f32x4*)(coef + c4), cb = *(const f32x4*)(coef + 64 + c4), cc = *(const f32x4*)(coef + 128 + c4);

    Wg16Slice sl;                     // one slice in flight: 2 halo float4 (the second only for tid < 104), dz, y
    TS(0); TSC(14);
    int ts_k = 1;
    (void)ts_k;
    int tile = blockIdx.x;
    if (tile < n_tiles) {
        // first tile: all 8 slices in flight at once (one memory round trip instead of eight)
        Wg16Slice s8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) s8[c] = wg16_load(dz, yin, xin, H, tiles_per_clip, tile, c, tid);
#pragma unroll
        for (int c = 0; c < 8; ++c) wg16_store(s8[c], smem, H, tiles_per_clip, tile, c, tid, ca, cb, cc);
    }
    __syncthreads();
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        if (ts_k < 12) { TS(ts_k); ++ts_k; }
        const int nxt = tile + gridDim.x;
        const bool has = nxt < n_tiles;
        const float* xh = smem + cur * C::BUF_FLOATS;
        float* other = smem + (cur ^ 1) * C::BUF_FLOATS;
        const float* Ab = xh + C::XH_FLOATS + kh * 64 + 32 * cob + n;
        const float* Bb = xh + kh * 64 + 32 * cib + n;
#pragma unroll
        for (int ty = 0; ty < C::TH; ++ty) {          // unrolled: the slice index is a constant in the address arithmetic
            if (has) {
                if (ty > 0) wg16_store(sl, other, H, tiles_per_clip, nxt, ty - 1, tid, ca, cb, cc);
                sl = wg16_load(dz, yin, xin, H, tiles_per_clip, nxt, ty, tid);
            }
#pragma unroll
            for (int txp = 0; txp < C::TW / 2; ++txp) {
                const float a = Ab[(ty * C::TW + 2 * txp) * 64];
#pragma unroll
                for (int tl = 0; tl < 9; ++tl) {
                    const int dy = tl / 3 - 1, dx = tl % 3 - 1;
                    const float bv = Bb[((ty + 1 + dy) * C::HW + 2 * txp + 1 + dx) * 64];
                    acc[tl] = mfma32(a, bv, acc[tl]);
                }
            }
        }
        if (has) wg16_store(sl, other, H, tiles_per_clip, nxt, C::TH - 1, tid, ca, cb, cc);
        lds_barrier();
        cur ^= 1;
    }
    TS(12);
    float* dst = part + (size_t)blockIdx.x * 9 * 4096;
#pragma unroll
    for (int tl = 0; tl < 9; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * cob + mfma32_row(r, lane);
            dst[tl * 4096 + co * 64 + 32 * cib + n] = acc[tl][r];
        }
    TS(13); TSC(15);
}

#endif  // SED_AB

// ---- block-1 wgrad in the Winograd domain -----------------------------------------------------------------------------
// dW = G^T [ sum over 2x2 output blocks of (B^T d B) (.) (A dY A^T) ] G: 16 multiplies per block and (ci, co) pair instead
// of 36, like the forward kernel.  Here BOTH operands need a transform, so they are transformed once into LDS (no
// redundancy between waves) and the MFMAs read ready-made fragments:
//   * tile = 8 image rows x 16 columns = 32 blocks = 8 k-steps of 4 blocks (K of v_mfma_f32_16x16x4_f32); block 2 (TW = 4):
//     16 rows x 4 columns = 16 blocks = 4 k-steps, one tile per workgroup at B = 24;
//   * wave (cg, ph) accumulates dU[pos][co][ci] for its 8 positions (transform rows 3 ph and 1 + ph), all 64 co (4 M
//     tiles, one ds_read_b128 per position thanks to the co' = 4 (co & 15) + (co >> 4) order) and its 16 ci (N tile cg):
//     128 accumulator registers, 32 MFMAs per k-step for 8 + 8 LDS reads;
//   * the operands of k-step s + 1 are produced while k-step s is multiplied (double-buffered Vs / Ms): the ph = 0 waves
//     transform the inputs of its 4 blocks (lane = channel: V = B^T d B from the LDS halo of x -> Vs[blk][pos][ci]), the
//     ph = 1 waves the output gradients (dY = ca dz + cb y + cc straight from global memory - 2x2 blocks do not overlap,
//     so no staging, prefetched one k-step ahead; dM = A dY A^T -> Ms[blk][pos][co']).  The two waves of a SIMD do
//     these in OPPOSITE order (ph = 0: transform, then MFMAs; ph = 1: MFMAs, then transform), so one of them always
//     has MFMAs to issue; one barrier per k-step;
//   * epilogue: G^T dU G per wave for its two transform rows (linear, so the two partial results just add up), summed in
//     LDS into the same [tap][co][ci] slab format as the direct kernel - k_wgrad_reduce is shared.
// (A first version transformed 8 blocks at a time in a phase of its own: 0.8 us of every 3.35 us with the MFMA pipe idle.)
#ifdef SED_TS
#define TSW4(k) do { if (threadIdx.x == 256 && blockIdx.x < 1024) g_ts[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define TSW4(k) do { } while (0)
#endif
template <int TW_>
struct WgW {
    static constexpr int TW = TW_, BC = TW / 2;                        // block columns per image row
    static constexpr int NKS = (TW == 16) ? 8 : 4;                     // k-steps (4 blocks each) per tile
    static constexpr int TH = 2 * (4 * NKS / BC);                      // image rows per tile: 8 (TW 16), 16 (TW 4)
    static constexpr int HW = TW + 2, HH = TH + 2, PS = 66, RS = HW * PS;
    static constexpr int HALO_FLOATS = HH * RS;
    static constexpr int RW = HW * 16, RPI = 512 / RW, NITEMS = (HH + RPI - 1) / RPI;   // row staging: TW 16: 288, 1, 10; TW 4: 96, 5, 4
    static constexpr int SV = 1040, SM = 1024;      // block strides: the b32 B reads of a half-wave (2 blocks) need SV = 16 mod 32
    static constexpr int OPS_FLOATS = 4 * SV + 4 * SM;                 // one k-step's operands
    static constexpr int OUT_STRIDE = 68;           // epilogue staging [tap][co][68]: 4 co rows apart = 16 banks apart
    static constexpr size_t MAIN_BYTES = (size_t)(2 * HALO_FLOATS + 2 * OPS_FLOATS) * 4;
    static constexpr size_t OUT_BYTES = (size_t)9 * 64 * OUT_STRIDE * 4;
    static constexpr size_t LDS_BYTES = MAIN_BYTES > OUT_BYTES ? MAIN_BYTES : OUT_BYTES;
    // block cg (0..3) of k-step ks: blk = 4 ks + cg -> block row / column; split into a compile-time part (ks) and a
    // wave-uniform part (cg)
    __host__ __device__ static constexpr int br_ks(int ks) { return (4 * ks) / BC; }
    __host__ __device__ static constexpr int bc_ks(int ks) { return (4 * ks) % BC; }
};
template <int TW>
__global__ __launch_bounds__(512, 1) void k_wgrad_wino(const float* __restrict__ dz, const float* __restrict__ yin,
                                                         const float* __restrict__ coef, const float* __restrict__ xin,
                                                         float* __restrict__ part, int H, int tiles_per_clip, int n_tiles,
                                                         BnBwdPrepArgs prep) {
    using C = WgW<TW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ops = smem + 2 * C::HALO_FLOATS;          // [2 buffers][Vs: 4 blocks x SV | Ms: 4 blocks x SM]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, ph = wave >> 2;
    // block cg of a k-step relative to the k-step's first block: (row, column) offsets (wave-uniform)
    const int cg_br = (TW == 16) ? 0 : (cg >> 1), cg_bc = (TW == 16) ? cg : (cg & 1);
    const int i16 = lane & 15, kq = lane >> 4;
    float ca, cb, cc;                                                              // transform role: lane = channel
    if (prep.acc != nullptr) bn_bwd_coef(prep, lane, ca, cb, cc);                  // (no k_bn_bwd_prep: see BnBwdPrepArgs)
    else { ca = coef[lane]; cb = coef[64 + lane]; cc = coef[128 + lane]; }
    // halo staging by image rows: thread -> (column hx = tid >> 4, channels 4 (tid & 15) ..), one float4 per halo row and
    // thread, so everything but the row is a per-lane constant (a flat "item = tid + 512 i" split cost ~25 VALU per item
    // in div / mod / bounds arithmetic - 0.35 us per k-step that carried one).  Threads 288..511 duplicate columns 2..15
    // (same data to the same LDS address) instead of being masked: no divergent store for the compiler to sink the load into.
    constexpr int NITEMS = C::NITEMS, NH = 2;                                      // items (1 or 5 halo rows each), fetched two at a time
    f32x4_t pre[NH];
    const int td = tid < C::RPI * C::RW ? tid : tid - (TW == 16 ? 256 : 480);      // left-over threads duplicate others
    const int r_in = (C::RPI == 1) ? 0 : td / C::RW, hx = ((C::RPI == 1) ? td : td % C::RW) >> 4, c4 = (td & 15) * 4;
    const int ixc = hx - 1 < 0 ? 0 : (hx - 1 > C::TW - 1 ? C::TW - 1 : hx - 1);
    const uint32_t goff4 = (uint32_t)(ixc * 64 + c4) * 4u;                         // byte offset inside an image row
    const bool colok = hx >= 1 && hx <= C::TW;
    const int lds_off = hx * C::PS + c4;
    auto load_row = [&](int b, int y0, int it, f32x4_t& d0) {
        const int hy = it * C::RPI + r_in < C::HH ? it * C::RPI + r_in : C::HH - 1;  // rows past the halo: its last row again
        const int iy = y0 - 1 + hy;
        const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);                    // always a valid row; masked when stored
        const float* rowp = xin + (size_t)(b * H + iyc) * (C::TW * 64);
        d0 = *(const f32x4_t*)((const char*)rowp + goff4);
    };
    auto store_row = [&](float* halo, int y0, int it, const f32x4_t& s0) {
        const int hy = it * C::RPI + r_in < C::HH ? it * C::RPI + r_in : C::HH - 1;
        const int iy = y0 - 1 + hy;
        const bool ok = colok && iy >= 0 && iy < H;
        const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t v = ok ? s0 : z4;
        float* d = halo + hy * C::RS + lds_off;
        *(float2*)d = make_float2(v[0], v[1]);
        *(float2*)(d + 2) = make_float2(v[2], v[3]);
    };
    // k-step ks of a tile = block row ks >> 1, block columns 4 (ks & 1) .. + 3; transform role: block (wave & 3) of them
    f32x4_t acc[8][4];
    // ph = 1 waves: raw 2x2 output-gradient blocks, fetched TWO k-steps ahead (one k-step is ~1 us: not enough to cover the
    // load latency under load - the first version waited ~1 us per k-step here); slot = k-step parity
    float dzr[2][4], yr[2][4];
    // buffer loads: descriptor + uniform pixel offset (soffset) + 4 * lane (voffset) - no per-load address arithmetic
    const int img_bytes = (n_tiles / tiles_per_clip) * H * (C::TW * 64 * 4);
    const auto rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)dz, (short)0, img_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)yin, (short)0, img_bytes, 0x00020000);
    const int lane4 = 4 * lane;
    auto load_dy = [&](int b, int y0, int ks, int slot) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y0 + 2 * (C::br_ks(ks) + cg_br) + (q >> 1), xx = 2 * (C::bc_ks(ks) + cg_bc) + (q & 1);
            // rows past the image re-read the last row and are zeroed in the transform - by a multiplication: with a
            // select there the compiler threads the (uniform) condition back to here and branches around the loads, and
            // the control flow inside the unrolled loop made it shuffle the accumulators between registers
            const int yc = yy < H ? yy : H - 1;
            const int so = ((b * H + yc) * C::TW + xx) * 256;
            dzr[slot][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dz, lane4, so, 0));
            yr[slot][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_y, lane4, so, 0));
        }
    };
    // input transform of block cg of k-step ks, channel lane: V = B^T d B -> Vs[cg][pos][lane]
    auto transform_v = [&](const float* halo, int ks, float* Vs) {
        const float* P = halo + (2 * cg_br) * C::RS + (2 * cg_bc) * C::PS + lane + ((2 * C::br_ks(ks)) * C::RS + (2 * C::bc_ks(ks)) * C::PS);   // (constant part -> instruction offset)
        float T[4][4];
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            const float d0 = P[b2 * C::PS], d1 = P[C::RS + b2 * C::PS], d2 = P[2 * C::RS + b2 * C::PS], d3 = P[3 * C::RS + b2 * C::PS];
            T[0][b2] = d0 - d2; T[1][b2] = d1 + d2; T[2][b2] = d2 - d1; T[3][b2] = d1 - d3;
        }
        float* Vd = Vs + cg * C::SV + lane;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            Vd[(4 * i2 + 0) * 64] = T[i2][0] - T[i2][2];
            Vd[(4 * i2 + 1) * 64] = T[i2][1] + T[i2][2];
            Vd[(4 * i2 + 2) * 64] = T[i2][2] - T[i2][1];
            Vd[(4 * i2 + 3) * 64] = T[i2][1] - T[i2][3];
        }
    };
    // output-gradient transform of the same block, channel lane: dM = A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1] -> Ms[cg][pos][co']
    auto transform_m = [&](int y0, int ks, int slot, float* Ms) {
        float dy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = fmaf(ca, dzr[slot][q], fmaf(cb, yr[slot][q], cc));
            dy[q] = v * ((y0 + 2 * (C::br_ks(ks) + cg_br) + (q >> 1) < H) ? 1.f : 0.f);     // rows past the image contribute nothing
        }
        // R[i][x] = sum_y A[i][y] dY[y][x]
        const float R[4][2] = {{dy[0], dy[1]}, {dy[0] + dy[2], dy[1] + dy[3]}, {dy[0] - dy[2], dy[1] - dy[3]}, {-dy[2], -dy[3]}};
        float* Md = Ms + cg * C::SM + 4 * (lane & 15) + (lane >> 4);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            Md[(4 * i2 + 0) * 64] = R[i2][0];
            Md[(4 * i2 + 1) * 64] = R[i2][0] + R[i2][1];
            Md[(4 * i2 + 2) * 64] = R[i2][0] - R[i2][1];
            Md[(4 * i2 + 3) * 64] = -R[i2][1];
        }
    };
    // the 32 MFMAs of one k-step: A = dM (M = co, 4 tiles per b128), B = V (N = ci)
    auto mma = [&](const float* Vs, const float* Ms) {
        const float* Ab = Ms + kq * C::SM + 4 * i16 + 256 * (1 + ph);       // this wave's Y row (1 + ph); X row = 3 ph
        const float* Bb = Vs + kq * C::SV + 16 * cg + i16 + 256 * (1 + ph);
        const int xo = ph ? 256 : -256;                                     // (uniform) X row (3 ph) relative to the Y row
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int po = (p8 < 4 ? xo : 0) + (p8 & 3) * 64;
            const f32x4_t a4 = *(const f32x4_t*)(Ab + po);
            const float bv = Bb[po];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[p8][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m], bv, acc[p8][m], 0, 0, 0);
        }
    };
    const TileWalk walk = xcd_walk(n_tiles);
    int tile = walk.first;
    TS(0);
    int it_ts = 0;
    int tb = tile / tiles_per_clip, ty0 = (tile % tiles_per_clip) * C::TH;
    if (tile < walk.end) {
        f32x4_t first[NITEMS];
#pragma unroll
        for (int hy = 0; hy < NITEMS; ++hy) load_row(tb, ty0, hy, first[hy]);
        if (ph) { load_dy(tb, ty0, 0, 0); load_dy(tb, ty0, 1, 1); }
#pragma unroll
        for (int hy = 0; hy < NITEMS; ++hy) store_row(smem, ty0, hy, first[hy]);
    }
    __syncthreads();
    if (tile < walk.end) {                                   // operands of k-step 0 -> buffer 0
        if (ph == 0) transform_v(smem, 0, ops);
        else transform_m(ty0, 0, 0, ops + 4 * C::SV);
    }
    __syncthreads();
    TS(1);
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[p8][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (; tile < walk.end; tile += walk.step) {
        const int nxt = (tile + walk.step < walk.end) ? tile + walk.step : tile;             // (unconditional prefetch, see k_conv_wino)
        const int nb_ = nxt / tiles_per_clip, ny0 = (nxt % tiles_per_clip) * C::TH;
        const float* halo = smem + cur * C::HALO_FLOATS;
        float* halo_nxt = smem + (cur ^ 1) * C::HALO_FLOATS;
        constexpr int NKS = C::NKS, NPIECE = (NITEMS + NH - 1) / NH;       // NPIECE <= NKS - 2
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            // next tile's halo, two rows per k-step: rows 2 ks, 2 ks + 1 are fetched at the start of k-step ks (0..4) and stored
            // at the end of k-step ks + 1; complete before the barrier of k-step 6, because k-step 7 already transforms the
            // next tile's first blocks
            f32x4_t held[NH];
            if (ks >= 1 && ks <= NPIECE) {
#pragma unroll
                for (int k2 = 0; k2 < NH; ++k2) held[k2] = pre[k2];
            }
            if (ks < NPIECE) {
#pragma unroll
                for (int k2 = 0; k2 < NH; ++k2)
                    if (NH * ks + k2 < NITEMS) load_row(nb_, ny0, NH * ks + k2, pre[k2]);
            }
            float* nb = ops + ((ks + 1) & 1) * C::OPS_FLOATS;                   // operands of the next k-step
            const float* cb2 = ops + (ks & 1) * C::OPS_FLOATS;
            // (one copy of the MFMA code between two uniform branches: with it inside both arms of an if / else the register
            // allocator spilled the accumulators)
            if (ph == 0) {
                if (ks < NKS - 1) transform_v(halo, ks + 1, nb); else transform_v(halo_nxt, 0, nb);
            }
            if (it_ts == 0 && ks == 2) { TS(3); TSW4(7); }
            mma(cb2, cb2 + 4 * C::SV);
            if (it_ts == 0 && ks == 2) { TS(4); TSW4(8); }
            if (ph == 1) {                                     // slot ks & 1 was consumed by the previous k-step's transform
                if (ks < NKS - 2) load_dy(tb, ty0, ks + 2, ks & 1); else load_dy(nb_, ny0, ks - (NKS - 2), ks & 1);
                if (ks < NKS - 1) transform_m(ty0, ks + 1, (ks + 1) & 1, nb + 4 * C::SV); else transform_m(ny0, 0, 0, nb + 4 * C::SV);
            }
            if (ks >= 1 && ks <= NPIECE) {
#pragma unroll
                for (int k2 = 0; k2 < NH; ++k2)
                    if (NH * (ks - 1) + k2 < NITEMS) store_row(halo_nxt, ny0, NH * (ks - 1) + k2, held[k2]);
            }
            if (it_ts == 0 && ks == 2) { TS(5); TSW4(9); }
            lds_barrier();
            if (it_ts == 0 && ks == 1) { TS(2); TSW4(6); }
        }
        if (it_ts == 0) TS(10);
        ++it_ts;
        cur ^= 1;
        tb = nb_; ty0 = ny0;
    }
    TS(11);
    // ---- epilogue: dg = G^T dU G for this wave's two transform rows, the two wave rows summed in LDS ----------------
    // per row: W[b] = sum_j dU[.][j] G[j][b] = (u0 + (u1 + u2)/2, (u1 - u2)/2, (u1 + u2)/2 + u3); then over the rows
    // i with G[i][a]: rows 0, 1 (ph = 0): dg[0][b] = Wx + Wy/2, dg[1][b] = dg[2][b] = Wy/2;
    //                 rows 3, 2 (ph = 1): dg[0][b] = Wy/2, dg[1][b] = -Wy/2, dg[2][b] = Wx + Wy/2   (x = row 3 ph, y = row 1 + ph)
    __syncthreads();
    float* ob = smem;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (ph == round) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float wx[3], wy[3];
                    {
                        const float u0 = acc[0][m][r], u1 = acc[1][m][r], u2 = acc[2][m][r], u3 = acc[3][m][r];
                        const float h = 0.5f * (u1 + u2);
                        wx[0] = u0 + h; wx[1] = 0.5f * (u1 - u2); wx[2] = h + u3;
                    }
                    {
                        const float u0 = acc[4][m][r], u1 = acc[5][m][r], u2 = acc[6][m][r], u3 = acc[7][m][r];
                        const float h = 0.5f * (u1 + u2);
                        wy[0] = 0.5f * (u0 + h); wy[1] = 0.25f * (u1 - u2); wy[2] = 0.5f * (h + u3);     // Wy / 2
                    }
                    float* o = ob + (16 * m + 4 * kq + r) * C::OUT_STRIDE + 16 * cg + i16;          // + tap * 64 * OUT_STRIDE
#pragma unroll
                    for (int b2 = 0; b2 < 3; ++b2) {
                        const float g0 = ph ? wy[b2] : wx[b2] + wy[b2];
                        const float g1 = ph ? -wy[b2] : wy[b2];
                        const float g2 = ph ? wx[b2] + wy[b2] : wy[b2];
                        if (round == 0) {
                            o[(0 + b2) * 64 * C::OUT_STRIDE] = g0; o[(3 + b2) * 64 * C::OUT_STRIDE] = g1; o[(6 + b2) * 64 * C::OUT_STRIDE] = g2;
                        } else {
                            o[(0 + b2) * 64 * C::OUT_STRIDE] += g0; o[(3 + b2) * 64 * C::OUT_STRIDE] += g1; o[(6 + b2) * 64 * C::OUT_STRIDE] += g2;
                        }
                    }
                }
        }
        __syncthreads();
    }
    TS(12);
    float* dst = part + (size_t)blockIdx.x * 9 * 4096;
    for (int idx = tid; idx < 9 * 64 * 16; idx += 512) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        *(f32x4_t*)(dst + row * 64 + c4) = *(const f32x4_t*)(ob + row * C::OUT_STRIDE + c4);
    }
    TS(13);
}

// ---- block-2 wgrad (W = 4), output-stationary over the Winograd transform rows --------------------------------------------------
// k_wgrad_wino<4> gives every one of its 240 one-tile workgroups a full [9][64][64] partial slab: 35 MB written and read
// again for 11.6 MB of operands - the slabs, not the multiplies, are that launch (profiles/r05_mt-f32_pmc_hbm_traffic.md; MFMA
// busy 0.10).  Here a workgroup owns ONE transform row i of the Winograd-domain sum dU[i][j][co][ci] (4 positions x 64 x 64 =
// 32 accumulator registers per lane instead of 128) over a RUN of tiles (run r: tiles r, r + R, ...; R = 64 runs x 4 rows = 256
// workgroups, the four rows of a run under one XCD's L2: workgroup w = run (w & 7) + 8 (w >> 5), row (w >> 3) & 3).  Per
// k-step (4 blocks) a workgroup transforms only its row - V[i][.] = (B^T d B)[i][.] needs two of the four patch rows, dM[i][.]
// = (A dY A^T)[i][.] is a combination of the block's two dY rows - so the four rows of a run split the transform work without
// repeating it; what they repeat is the halo staging (L2 reads).  Wave (cg, jh): positions (i, 2 jh), (i, 2 jh + 1), all 64 co,
// ci tile cg: 8 MFMAs per k-step; transform role as in k_wgrad_wino (jh = 0: V of block cg, jh = 1: dM of block cg).
// Epilogue: the column half of G^T dU G (j -> kernel column b: W_i[b] = sum_j dU[i][j] G[j][b]) in registers, the two jh
// waves' halves added through LDS; 3 x 64 x 64 floats per workgroup = 12.6 MB for the launch.  k_wgrad4_os_reduce sums the
// runs in a fixed order and applies the row half (dg[a][b] = sum_i G[i][a] W_i[b]).  Deterministic, no atomics.
#define WG4_RUNS 64
struct Wg4Os {
    static constexpr int SV4 = 4 * 64 + 16, SM4 = 4 * 64, OPS4 = 4 * SV4 + 4 * SM4;      // one k-step's operands: [blk][j][64]
    static constexpr size_t LDS_BYTES = (size_t)(2 * WgW<4>::HALO_FLOATS + 2 * OPS4) * 4; // 74 KB (the epilogue's 48 KB reuse it)
};
__global__ __launch_bounds__(512, 1) void k_wgrad4_os(const float* __restrict__ dz, const float* __restrict__ yin,
                                                       const float* __restrict__ coef, const float* __restrict__ xin,
                                                       float* __restrict__ part, int H, int tiles_per_clip, int n_tiles, int n_runs,
                                                       BnBwdPrepArgs prep) {
    using C = WgW<4>;
    constexpr int SV4 = Wg4Os::SV4, SM4 = Wg4Os::SM4, OPS4 = Wg4Os::OPS4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ops = smem + 2 * C::HALO_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, jh = wave >> 2;
    const int w = blockIdx.x, run = (w & 7) + 8 * (w >> 5), ti = (w >> 3) & 3;     // (wave-uniform: SGPRs)
    const int cg_br = cg >> 1, cg_bc = cg & 1;
    const int i16 = lane & 15, kq = lane >> 4;
    // row ti of B^T d: d[p] + sv d[q]; row ti of A dY: ma dY[0] + mb dY[1]
    const int vp = (ti == 0) ? 0 : (ti == 2 ? 2 : 1), vq = (ti == 3) ? 3 : (ti == 2 ? 1 : 2);
    const float sv = (ti == 1) ? 1.f : -1.f;
    const float ma = (ti == 3) ? 0.f : 1.f, mb = (ti == 0) ? 0.f : (ti == 1 ? 1.f : -1.f);
    float ca, cb, cc;
    if (prep.acc != nullptr) bn_bwd_coef(prep, lane, ca, cb, cc);
    else { ca = coef[lane]; cb = coef[64 + lane]; cc = coef[128 + lane]; }
    // halo staging by image rows, as in k_wgrad_wino<4>: thread -> (row r_in of 5, column hx, channels c4 ..)
    constexpr int NITEMS = C::NITEMS, NH = 2;
    f32x4_t pre[NH];
    const int td = tid < C::RPI * C::RW ? tid : tid - 480;
    const int r_in = td / C::RW, hx = (td % C::RW) >> 4, c4 = (td & 15) * 4;
    const int ixc = hx - 1 < 0 ? 0 : (hx - 1 > C::TW - 1 ? C::TW - 1 : hx - 1);
    const uint32_t goff4 = (uint32_t)(ixc * 64 + c4) * 4u;
    const bool colok = hx >= 1 && hx <= C::TW;
    const int lds_off = hx * C::PS + c4;
    auto load_row = [&](int b, int y0, int it, f32x4_t& d0) {
        const int hy = it * C::RPI + r_in < C::HH ? it * C::RPI + r_in : C::HH - 1;
        const int iy = y0 - 1 + hy;
        const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
        const float* rowp = xin + (size_t)(b * H + iyc) * (C::TW * 64);
        d0 = *(const f32x4_t*)((const char*)rowp + goff4);
    };
    auto store_row = [&](float* halo, int y0, int it, const f32x4_t& s0) {
        const int hy = it * C::RPI + r_in < C::HH ? it * C::RPI + r_in : C::HH - 1;
        const int iy = y0 - 1 + hy;
        const bool ok = colok && iy >= 0 && iy < H;
        const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t v = ok ? s0 : z4;
        float* d = halo + hy * C::RS + lds_off;
        *(float2*)d = make_float2(v[0], v[1]);
        *(float2*)(d + 2) = make_float2(v[2], v[3]);
    };
    f32x4_t acc[2][4];
    float dzr[2][4], yr[2][4];
    const int img_bytes = (n_tiles / tiles_per_clip) * H * (C::TW * 64 * 4);
    const auto rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)dz, (short)0, img_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)yin, (short)0, img_bytes, 0x00020000);
    const int lane4 = 4 * lane;
    auto load_dy = [&](int b, int y0, int ks, int slot) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y0 + 2 * (C::br_ks(ks) + cg_br) + (q >> 1), xx = 2 * (C::bc_ks(ks) + cg_bc) + (q & 1);
            const int yc = yy < H ? yy : H - 1;                          // (masked by multiplication in the transform)
            const int so = ((b * H + yc) * C::TW + xx) * 256;
            dzr[slot][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dz, lane4, so, 0));
            yr[slot][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_y, lane4, so, 0));
        }
    };
    // V[ti][j] of block cg of k-step ks, channel lane -> Vs[cg][j][lane]
    auto transform_v = [&](const float* halo, int ks, float* Vs) {
        const float* P = halo + (2 * cg_br) * C::RS + (2 * cg_bc) * C::PS + lane + ((2 * C::br_ks(ks)) * C::RS + (2 * C::bc_ks(ks)) * C::PS);
        const float* Pp = P + vp * C::RS;
        const float* Pq = P + vq * C::RS;
        float T[4];
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) T[b2] = fmaf(sv, Pq[b2 * C::PS], Pp[b2 * C::PS]);
        float* Vd = Vs + cg * SV4 + lane;
        Vd[0] = T[0] - T[2]; Vd[64] = T[1] + T[2]; Vd[128] = T[2] - T[1]; Vd[192] = T[1] - T[3];
    };
    // dM[ti][j] of the same block, channel lane -> Ms[cg][j][co']
    auto transform_m = [&](int y0, int ks, int slot, float* Ms) {
        float dy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = fmaf(ca, dzr[slot][q], fmaf(cb, yr[slot][q], cc));
            dy[q] = v * ((y0 + 2 * (C::br_ks(ks) + cg_br) + (q >> 1) < H) ? 1.f : 0.f);
        }
        const float R0 = ma * dy[0] + mb * dy[2], R1 = ma * dy[1] + mb * dy[3];
        float* Md = Ms + cg * SM4 + 4 * (lane & 15) + (lane >> 4);
        Md[0] = R0; Md[64] = R0 + R1; Md[128] = R0 - R1; Md[192] = -R1;
    };
    auto mma = [&](const float* Vs, const float* Ms) {
        const float* Ab = Ms + kq * SM4 + 4 * i16 + 128 * jh;
        const float* Bb = Vs + kq * SV4 + 16 * cg + i16 + 128 * jh;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const f32x4_t a4 = *(const f32x4_t*)(Ab + 64 * j2);
            const float bv = Bb[64 * j2];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[j2][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m], bv, acc[j2][m], 0, 0, 0);
        }
    };
    int tile = run;
    int tb = tile / tiles_per_clip, ty0 = (tile % tiles_per_clip) * C::TH;
    if (tile < n_tiles) {
        f32x4_t first[NITEMS];
#pragma unroll
        for (int hy = 0; hy < NITEMS; ++hy) load_row(tb, ty0, hy, first[hy]);
        if (jh) { load_dy(tb, ty0, 0, 0); load_dy(tb, ty0, 1, 1); }
#pragma unroll
        for (int hy = 0; hy < NITEMS; ++hy) store_row(smem, ty0, hy, first[hy]);
    }
    __syncthreads();
    if (tile < n_tiles) {
        if (jh == 0) transform_v(smem, 0, ops);
        else transform_m(ty0, 0, 0, ops + 4 * SV4);
    }
    __syncthreads();
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[j2][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (; tile < n_tiles; tile += n_runs) {
        const int nxt = (tile + n_runs < n_tiles) ? tile + n_runs : tile;          // (unconditional prefetch, see k_conv_wino)
        const int nb_ = nxt / tiles_per_clip, ny0 = (nxt % tiles_per_clip) * C::TH;
        const float* halo = smem + cur * C::HALO_FLOATS;
        float* halo_nxt = smem + (cur ^ 1) * C::HALO_FLOATS;
        constexpr int NKS = C::NKS, NPIECE = (NITEMS + NH - 1) / NH;              // 4 k-steps, 2 pieces
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            f32x4_t held[NH];
            if (ks >= 1 && ks <= NPIECE) {
#pragma unroll
                for (int k2 = 0; k2 < NH; ++k2) held[k2] = pre[k2];
            }
            if (ks < NPIECE) {
#pragma unroll
                for (int k2 = 0; k2 < NH; ++k2)
                    if (NH * ks + k2 < NITEMS) load_row(nb_, ny0, NH * ks + k2, pre[k2]);
            }
            float* nb = ops + ((ks + 1) & 1) * OPS4;
            const float* cb2 = ops + (ks & 1) * OPS4;
            if (jh == 0) {
                if (ks < NKS - 1) transform_v(halo, ks + 1, nb); else transform_v(halo_nxt, 0, nb);
            }
            mma(cb2, cb2 + 4 * SV4);
            if (jh == 1) {
                if (ks < NKS - 2) load_dy(tb, ty0, ks + 2, ks & 1); else load_dy(nb_, ny0, ks - (NKS - 2), ks & 1);
                if (ks < NKS - 1) transform_m(ty0, ks + 1, (ks + 1) & 1, nb + 4 * SV4); else transform_m(ny0, 0, 0, nb + 4 * SV4);
            }
            if (ks >= 1 && ks <= NPIECE) {
#pragma unroll
                for (int k2 = 0; k2 < NH; ++k2)
                    if (NH * (ks - 1) + k2 < NITEMS) store_row(halo_nxt, ny0, NH * (ks - 1) + k2, held[k2]);
            }
            lds_barrier();
        }
        cur ^= 1;
        tb = nb_; ty0 = ny0;
    }
    // ---- epilogue: W[b] = sum_j dU[ti][j] G[j][b], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; jh = 0 holds j = 0, 1, jh = 1 j = 2, 3 ----
    __syncthreads();
    float* xb = smem + (size_t)cg * 48 * 64 + lane;           // [cg][(m, r, b)][lane]
    if (jh == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u2 = acc[0][m][r], u3 = acc[1][m][r];
                xb[((m * 4 + r) * 3 + 0) * 64] = 0.5f * u2;
                xb[((m * 4 + r) * 3 + 1) * 64] = -0.5f * u2;
                xb[((m * 4 + r) * 3 + 2) * 64] = 0.5f * u2 + u3;
            }
    }
    __syncthreads();
    if (jh == 0) {
        float* dst = part + ((size_t)run * 4 + ti) * 3 * 4096 + 16 * cg + i16;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u0 = acc[0][m][r], u1 = acc[1][m][r];
                const int co = 16 * m + 4 * kq + r;
                dst[(0 * 64 + co) * 64] = (u0 + 0.5f * u1) + xb[((m * 4 + r) * 3 + 0) * 64];
                dst[(1 * 64 + co) * 64] = 0.5f * u1 + xb[((m * 4 + r) * 3 + 1) * 64];
                dst[(2 * 64 + co) * 64] = 0.5f * u1 + xb[((m * 4 + r) * 3 + 2) * 64];
            }
    }
}

// part [run][row i][b][co][ci] -> g_w[co][ci][3 a + b] = sum_i G[i][a] sum_run part.  A workgroup owns 64 consecutive (co, ci)
// as 16 float4 columns; its 16 thread groups take every 16th run for all 12 (i, b) planes, then combine through LDS.
__global__ __launch_bounds__(256) void k_wgrad4_os_reduce(const float* __restrict__ part, int n_runs, float* __restrict__ g_w) {
    __shared__ __attribute__((aligned(16))) float red[16][12][64];
    __shared__ float S[12][64];
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const f32x4_t* P = (const f32x4_t*)part + blockIdx.x * 16 + c;
    f32x4_t s[12];
#pragma unroll
    for (int pl = 0; pl < 12; ++pl) s[pl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int r = grp; r < n_runs; r += 16) {
#pragma unroll
        for (int pl = 0; pl < 12; ++pl) s[pl] += P[((size_t)r * 12 + pl) * 1024];
    }
#pragma unroll
    for (int pl = 0; pl < 12; ++pl) *(f32x4_t*)&red[grp][pl][4 * c] = s[pl];
    __syncthreads();
    for (int o = threadIdx.x; o < 12 * 64; o += 256) {
        const int pl = o >> 6, e = o & 63;
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) v += red[g2][pl][e];
        S[pl][e] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * 9; o += 256) {
        const int e = o / 9, tap = o % 9, a = tap / 3, b = tap % 3;
        const float w0 = S[0 + b][e], w1 = S[3 + b][e], w2 = S[6 + b][e], w3 = S[9 + b][e];
        const float v = a == 0 ? w0 + 0.5f * (w1 + w2) : (a == 1 ? 0.5f * (w1 - w2) : 0.5f * (w1 + w2) + w3);
        g_w[(size_t)blockIdx.x * 64 * 9 + o] = v;
    }
}

// Sum of the per-workgroup partial slabs, in a fixed order (bit-reproducible).  A workgroup owns 64 consecutive outputs
// as 16 float4 columns; its 16 thread groups take every 16th slab (8 independent float4 loads in flight per thread - the
// scalar version with 4 groups had 32 B per thread in flight and ran at 3.1 TB/s), then combine through LDS.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int n_blocks, float* __restrict__ g_w) {
    __shared__ __attribute__((aligned(16))) float red[16][64];
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i4 = blockIdx.x * 16 + c;                      // float4 index over [tap][co][ci]
    const f32x4_t* P = (const f32x4_t*)part + i4;
    constexpr int SLAB4 = 9 * 4096 / 4;
    f32x4_t s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int k = grp;
    for (; k + 16 * 7 < n_blocks; k += 16 * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += P[(size_t)(k + 16 * u) * SLAB4];
    }
    for (; k < n_blocks; k += 16) s[0] += P[(size_t)k * SLAB4];
    *(f32x4_t*)&red[grp][4 * c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) v += red[g2][threadIdx.x];
        const int i = blockIdx.x * 64 + threadIdx.x;
        const int tap = i / 4096, co = (i / 64) % 64, ci = i % 64;
        g_w[(co * 64 + ci) * 9 + tap] = v;
    }
}

// ---- host launchers ---------------------------------------------------------------------------
int launch_conv_pack(const ConvPackArgs& a, hipStream_t st) {
    // every panel is followed by its Winograd-transformed form (SED_WINO_OFF floats in)
    k_conv_pack<<<SED_PACK_BLOCKS, 256, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

template <int TW, int MODE, int NS>
static int conv_launch_t(const float* in0, const float* in1, const float* coef, const float* wpk, const float* bias,
                         float* out, double* stat, int B, int H, hipStream_t st) {
    using Cfg = ConvCfg<TW>;
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3<TW, MODE, NS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)Cfg::LDS_BYTES));
    }
    const int tpc = (H + Cfg::TH - 1) / Cfg::TH;
    k_conv3x3<TW, MODE, NS><<<dim3(B * tpc, NS), 256, Cfg::LDS_BYTES, st>>>(in0, in1, coef, wpk, bias, out, stat, B, H, tpc);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

#ifdef SED_AB
#define SED_AB_FLAGS(mask) (g_sed_debug & (mask))
#else
#define SED_AB_FLAGS(mask) 0      // the direct (9-tap) A/B kernels are not compiled into the product library
#endif
extern "C" int sed_build_flags(void) {
#ifdef SED_AB
    return 1;
#else
    return 0;
#endif
}

int launch_conv_fwd(const float* in, const float* wpk, const float* bias, float* y, double* stat, int zero_stat, int B,
                    int H, int W, hipStream_t st) {
    if (stat && zero_stat) SED_CHECK_HIP(hipMemsetAsync(stat, 0, 128 * sizeof(double), st));
#ifdef SED_AB
    const bool direct = (g_sed_debug & (2 | 64)) != 0;          // A/B timing only: the 9-tap kernels
    if (direct && W == 16) return (g_sed_debug & 2) ? conv_launch_t<16, 0, 1>(in, nullptr, nullptr, wpk, bias, y, stat, B, H, st)
                                                    : conv16_ws_launch<0>(in, nullptr, nullptr, wpk, bias, y, stat, B, H, st);
    if (direct && W == 4) return conv_launch_t<4, 0, 2>(in, nullptr, nullptr, wpk, bias, y, stat, B, H, st);
#endif
    if (W == 16) return conv_wino_launch<16, 0>(in, nullptr, nullptr, wpk, bias, y, stat, B, H, nullptr, st);
    if (W == 4) return conv_wino_launch<4, 0>(in, nullptr, nullptr, wpk, bias, y, stat, B, H, nullptr, st);
    sed_set_error("conv: unsupported width %d", W);
    return SED_ERR_UNSUPPORTED;
}

int launch_conv_dgrad(const float* dz, const float* yin, const float* coef, const float* wpkT, float* dx, int B, int H,
                      int W, const BnBwdPrepArgs* prep, hipStream_t st) {
#ifdef SED_AB
    const bool direct = (g_sed_debug & (4 | 64)) != 0;          // A/B timing only: the 9-tap kernels
    SED_CHECK_ARG(!(direct && prep), "conv dgrad: in-kernel BatchNorm-backward coefficients need the Winograd kernel");
    if (direct && W == 16) return (g_sed_debug & 4) ? conv_launch_t<16, 1, 1>(dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, st)
                                                    : conv16_ws_launch<1>(dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, st);
    if (direct && W == 4) return conv_launch_t<4, 1, 2>(dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, st);
#endif
    if (W == 16) return conv_wino_launch<16, 1>(dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, prep, st);
    if (W == 4) return conv_wino_launch<4, 1>(dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, prep, st);
    sed_set_error("conv dgrad: unsupported width %d", W);
    return SED_ERR_UNSUPPORTED;
}

template <int TW, int TS>
static int wgrad_launch_t(const float* dz, const float* yin, const float* coef, const float* xin, float* part,
                          int n_blocks, float* g_w, int B, int H, const BnBwdPrepArgs* prep, hipStream_t st) {
    BnBwdPrepArgs pa = {};
    if (prep) pa = *prep;
#ifdef SED_AB
    using Cfg = WgCfg<TW>;
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wgrad<TW, TS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)Cfg::LDS_BYTES));
    }
    const int tpc = (H + Cfg::TH - 1) / Cfg::TH, nt = B * tpc;
    int nb = nt < n_blocks ? nt : n_blocks;
#else
    int nb = 0;
#endif
    // block 1: the double-buffered kernel (15 % faster alone; in the step, next to dgrad on the other stream, 1.143 vs
    // 1.162 ms per step although its 154 KB of LDS keep any other workgroup off its CU); bit 3 of the debug knob = old
    // default: Winograd-domain kernel (block 1: operator 65 us against 105 us for the direct double-buffered kernel); bits 3 / 7
    // of the debug knob bring the direct kernels back (bit 7 = k_wgrad16_db for block 1, bit 3 = the tile kernel)
    if (TW == 4 && !SED_AB_FLAGS(8 | 128) && !(g_sed_debug & 67108864)) {      // (debug bit 26: k_wgrad_wino<4>, the slab form, A/B)
        using CW = WgW<4>;
        static thread_local SedAttrOnce attro;
        if (attro.need()) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad4_os, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg4Os::LDS_BYTES));
        }
        SED_CHECK_ARG((size_t)B * H * TW * 64 < ((size_t)1 << 31), "wgrad: image too large for 32-bit offsets");
        const int tpcw = (H + CW::TH - 1) / CW::TH, ntw = B * tpcw;
        int n_runs = (ntw + 7) & ~7;
        if (n_runs > WG4_RUNS) n_runs = WG4_RUNS;
        SED_CHECK_ARG(n_runs * 4 <= n_blocks * 3, "wgrad: partial buffer too small");          // 12 x 4096 floats per run against 9 x 4096 per block
        k_wgrad4_os<<<4 * n_runs, 512, Wg4Os::LDS_BYTES, st>>>(dz, yin, coef, xin, part, H, tpcw, ntw, n_runs, pa);
        SED_CHECK_LAUNCH();
        k_wgrad4_os_reduce<<<4096 / 64, 256, 0, st>>>(part, n_runs, g_w);
        SED_CHECK_LAUNCH();
        return SED_OK;
    }
    if (!SED_AB_FLAGS(8 | 128)) {
        using CW = WgW<TW>;
        static thread_local SedAttrOnce attrw;
        if (attrw.need()) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad_wino<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CW::LDS_BYTES));
        }
        SED_CHECK_ARG((size_t)B * H * TW * 64 < ((size_t)1 << 31), "wgrad: image too large for 32-bit offsets");
        const int tpcw = (H + CW::TH - 1) / CW::TH, ntw = B * tpcw;
        nb = ntw < n_blocks ? ntw : n_blocks;
        k_wgrad_wino<TW><<<nb, 512, CW::LDS_BYTES, st>>>(dz, yin, coef, xin, part, H, tpcw, ntw, pa);
    }
#ifdef SED_AB
    else if (TW == 16 && TS == 1 && !(g_sed_debug & 8)) {
        static thread_local SedAttrOnce attr16;
        if (attr16.need()) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad16_db, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg16::LDS_BYTES));
        }
        k_wgrad16_db<<<nb, 256, Wg16::LDS_BYTES, st>>>(dz, yin, coef, xin, part, H, tpc, nt);
    } else {
        k_conv3x3_wgrad<TW, TS><<<dim3(nb, TS), 256, Cfg::LDS_BYTES, st>>>(dz, yin, coef, xin, part, B, H, tpc, nt, pa);
    }
#endif
    SED_CHECK_LAUNCH();
    k_wgrad_reduce<<<9 * 4096 / 64, 256, 0, st>>>(part, nb, g_w);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_conv_wgrad(const float* dz, const float* yin, const float* coef, const float* xin, float* part, int n_blocks,
                      float* g_w, int B, int H, int W, const BnBwdPrepArgs* prep, hipStream_t st) {
    SED_CHECK_ARG(!(prep && W == 16 && SED_AB_FLAGS(8 | 128)), "conv wgrad: in-kernel BatchNorm-backward coefficients need the default kernels");
    if (W == 16) return wgrad_launch_t<16, 1>(dz, yin, coef, xin, part, n_blocks, g_w, B, H, prep, st);
    if (W == 4) return wgrad_launch_t<4, 3>(dz, yin, coef, xin, part, n_blocks, g_w, B, H, prep, st);
    sed_set_error("conv wgrad: unsupported width %d", W);
    return SED_ERR_UNSUPPORTED;
}
