// bconv.hip - 3x3 convolution C -> C (C in {64, 128}; images [B][H][W][C] with W in {16, 4}), forward and dgrad, on the
// bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulation) with register-blocked waves - the round-3 replacement of
// gconv.hip's k_gconv for the two bf16 arithmetic modes:
//   SED_DTYPE_BF16    (X3 = 0)  operands rounded to bf16 (RNE); activations are STORED as bf16 in HBM
//   SED_DTYPE_F16     (X3 = 2)  FORWARD only: activations and the weight panel are fp16 in HBM, v_mfma_f32_32x32x16_f16, the
//                               output is stored as fp16 (for the next forward operator; its bf16 copy for the backward kernels
//                               of the bf16 family is written by k_bglu_fwd, which has the tile staged anyway - a second store
//                               stream here cost 27 - 85 spilled registers); the dgrad of this mode is X3 = 0's
//   SED_DTYPE_BF16X3  (X3 = 1)  split operands: a = a_hi + a_lo, b = b_hi + b_lo (both halves bf16),
//                               a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi - three MFMAs per K = 16 (96 cycles against 512 for the
//                               exact-fp32 MFMA), products exact, relative error ~2^-16 per product; fp32 storage
//
// Reference ops: Conv2d(C, C, 3, 1, 1) of conv blocks 1 and 2 (baseline/models/CNN.py:46-47) and its autograd
// (gconv.hip states the three contractions).
//
// Why a new kernel (profiles/r02_b_wide-bf16_pmc_mfma_busy.md): k_gconv ran ONE wave per SIMD with a
// 32-pixel x C wave tile - 1.25 LDS fragment reads (1 KB each) per 32-cycle MFMA issued in order by a single wave, a
// barrier per 64-wide weight chunk with nothing else resident to cover it: MFMA pipe busy 0.17 (forward) / 0.12 (dgrad).
// Here:
//   * 8 waves per workgroup = TWO per SIMD: one wave's LDS waits / barrier arrivals sit under the other's MFMAs;
//   * wave tile 64 pixels x 64 channels = 2 x 2 MFMA tiles: 4 fragment reads per 4 MFMAs (X3: 8 per 12);
//   * workgroup tile 256 pixels x 128 channels (C = 128; 512 x 64 at C = 64): the 9C x C weight panel streams from L2
//     once per 256 / 512 output pixels instead of once per 128;
//   * padded LDS images whose fragment reads are bank-conflict-free by construction AND addressable with immediate
//     offsets: pixel stride 2C + 16 bytes (consecutive pixels shift by one 16-byte bank group), row pitch chosen so that
//     the rows a 16-lane ds_read_b128 group touches land on distinct groups too (tools/lds_layout_check.py enumerates
//     every tap);
//   * the next tile's halo (forward) is fetched into registers before the k-loop and lands in LDS after it.
#include <type_traits>
#include "gen.h"
#include "kernels.h"
#include "gkernels.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4b;

template <int X3> struct BStore { using T = __bf16; };
template <> struct BStore<1> { using T = float; };
template <> struct BStore<2> { using T = _Float16; };

// hi / lo split of 8 consecutive fp32 values into two bf16x8 vectors (RNE both times: v - hi is exact in fp32)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const __bf16 h0 = (__bf16)a[q], h1 = (__bf16)b[q];
        hi[q] = h0; hi[4 + q] = h1;
        lo[q] = (__bf16)(a[q] - (float)h0); lo[4 + q] = (__bf16)(b[q] - (float)h1);
    }
}
__device__ __forceinline__ bf16x8 round8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r[q] = (__bf16)a[q]; r[4 + q] = (__bf16)b[q]; }
    return r;
}

SED_TS_DEFINE(bconv)
#ifndef TS_DIR
#define TS_DIR 0
#endif
// ---- geometry of one instantiation ---------------------------------------------------------------------------------------
//   TW  image width = tile width (16 or 4)          TH  tile height (rows)
//   WM x WN waves: wave (wm, wn) owns pixels [wm * PXW, +PXW) of the tile (row-major) and channels [wn * CHW, +CHW)
template <int X3_, int C, int TW, int TH, int WM, int WN>
struct BConvCfg {
    static constexpr int X3 = (X3_ == 1) ? 1 : 0;                         // (the fp16 forward has the bf16 mode's geometry)
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int M = TH * TW, PXW = M / WM, CHW = C / WN, MB = PXW / 32, NB = CHW / 32;
    static constexpr int HW = TW + 2, HH = TH + 2;
    static constexpr int PS = 2 * C + 16;                                 // halo pixel stride (bytes)
    // row pitch (bytes), a multiple of 16: TW = 16 -> multiple of 256 (slot = hx mod 16 decides);
    // TW = 4 -> == 64 (mod 256) so that the 4 rows of a read group land 4 slots apart
    static constexpr int RP_RAW = HW * PS;
    static constexpr int RP = (TW == 16) ? ((RP_RAW + 255) / 256) * 256 : ((RP_RAW - 64 + 255) / 256) * 256 + 64;
    static constexpr int HALO_PLANE = HH * RP;
    static constexpr int PLANES = X3 ? 2 : 1;
    // k per streamed weight chunk (the split mode at C = 128, W = 16 holds two halo planes + two weight planes: 32)
    static constexpr int KC = (X3 && C == 128 && TW == 16) ? 32 : 64;
    static constexpr int BROW = KC * 2 + 16;                              // weight-chunk row stride (bytes): 144 / 80
    static constexpr int BBUF = C * BROW;                                 // one plane of one chunk
    static constexpr int HALO_BYTES = PLANES * HALO_PLANE;
    // bf16 mode (one plane), W = 4: THREE chunk buffers filled by LDS-DMA two chunks ahead (+ a 1 KB dump for the surplus
    // wave-instructions).  Measured per kernel, LDS-DMA against register staging: W = 4 forward 19.1 -> 18.2 / 12.3 -> 12.1 us
    // (C = 128 / 64), dgrad 16.6 -> 12.7 / 9.3 -> 7.8; W = 16 forward 48.0 -> 50.9 / 20.7 -> 22.4, dgrad 49.4 -> 54.3 / 20.4 ->
    // 19.5: the wide tiles are bound by LDS bandwidth (2 x 2 register blocking = 1 KB of fragments per MFMA = the LDS peak), not
    // by the weight prefetch, and pay for the in-order vmcnt waits (the previous tile's epilogue stores) - they keep the two
    // register-staged buffers, and so does the split mode (its tiles leave no room for a third buffer)
    static constexpr bool DMA = !X3 && TW == 4;
    static constexpr int NBUF = DMA ? 3 : 2;
    static constexpr int WB_BYTES = NBUF * PLANES * BBUF + (DMA ? 1024 : 0);
    static constexpr int RED_BYTES = WM * 2 * C * 4;                      // BatchNorm partial sums per M-row of waves
    static constexpr size_t LDS_BYTES = (size_t)HALO_BYTES + WB_BYTES + RED_BYTES;
    static_assert(M % (32 * WM) == 0 && C % (32 * WN) == 0, "wave tile must be whole MFMA tiles");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
};

// DIR 0: forward   out[p][co] = bias[co] + sum x[p + tap][ci] W[co][ci][tap]          (+ BatchNorm sum / sum^2 of out)
// DIR 1: dgrad     out[p][ci] = sum dy[p - tap][co] W[co][ci][tap], dy = ca * dz + cb * y + cc inside the image
//                  (wpk then holds the flipped / transposed panel, so the kernel body is the same correlation)
//   wpk: [plane (hi | lo)][n][9 C] bf16, k = tap * C + c contiguous (k_gen_pack)
template <int X3_, int C, int TW, int TH, int WM, int WN, int DIR>
__global__ __launch_bounds__(64 * WM * WN) void k_bconv(const void* __restrict__ in0_v, const void* __restrict__ in1_v,
                                                          const float* __restrict__ coef, const __bf16* __restrict__ wpk,
                                                          const float* __restrict__ bias, void* __restrict__ out_v,
                                                          double* __restrict__ stat, int H, int tiles_per_clip, int n_tiles,
                                                          __bf16* __restrict__ out2) {
    using Cfg = BConvCfg<X3_, C, TW, TH, WM, WN>;
    using S = typename BStore<X3_>::T;
    constexpr int X3 = (X3_ == 1) ? 1 : 0;
    constexpr int F16 = (X3_ == 2) ? 1 : 0;
    static_assert(!(F16 && DIR == 1), "the fp16 flavour is forward only");
    constexpr int NT = Cfg::NT, MB = Cfg::MB, NB = Cfg::NB, HH = Cfg::HH, PS = Cfg::PS, RP = Cfg::RP;
    constexpr int KC = Cfg::KC, BROW = Cfg::BROW, BBUF = Cfg::BBUF, PL = Cfg::PLANES, HPL = Cfg::HALO_PLANE;
    constexpr int K = 9 * C, NCH = K / KC, CPT = C / KC;                  // chunks per tap
    constexpr int CJ = C / 8;                                             // 16-byte channel groups per pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    unsigned char* halo = bsm;
    unsigned char* wb = bsm + Cfg::HALO_BYTES;
    float* red = (float*)(bsm + Cfg::HALO_BYTES + Cfg::WB_BYTES);
    const S* in0 = (const S*)in0_v;
    const S* in1 = (const S*)in1_v;
    S* out = (S*)out_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int wm = wv / WN, wn = wv % WN;

    // ---- one-time: zero the whole halo (the columns left and right of the image stay zero for good) --------------------
    for (int e = tid; e < Cfg::HALO_BYTES / 16; e += NT) *(u32x4b*)(halo + 16 * e) = (u32x4b){0u, 0u, 0u, 0u};

    // ---- halo staging items: (halo row hy, pixel px, channel group j); j is constant per thread (NT % CJ == 0) ---------
    constexpr int NITEM = HH * TW * CJ, NL = (NITEM + NT - 1) / NT;
    static_assert(NT % CJ == 0, "channel group must be constant per thread");
    const int sj = tid % CJ;
    float ca[8], cb[8], cc[8];
    if (DIR == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { ca[q] = coef[8 * sj + q]; cb[q] = coef[C + 8 * sj + q]; cc[q] = coef[2 * C + 8 * sj + q]; }
    }
    // registers of one staged tile: X3 (fp32 storage) two float4 per item, bf16 storage one 16-byte vector
    constexpr int RV = X3 ? 2 : 1;
    f32x4 hv[NL][RV], hw2[DIR == 1 ? NL : 1][RV];
    auto halo_load = [&](int tile) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int g = tid + NT * i, hy = g / (TW * CJ), px = (g / CJ) % TW;
            const int row = r0 - 1 + hy;
            const bool ok = (g < NITEM) && row >= 0 && row < H;
            const size_t off = ((size_t)(b * H + (ok ? row : 0)) * TW + (g < NITEM ? px : 0)) * C + 8 * sj;
            // unconditional loads (clamped address), zeroed afterwards: a load under a per-item condition becomes a branch with
            // its own s_waitcnt - serialized memory round trips
#pragma unroll
            for (int v = 0; v < RV; ++v) {
                hv[i][v] = *(const f32x4*)((const char*)(in0 + off) + 16 * v);
                if (DIR == 1) hw2[i][v] = *(const f32x4*)((const char*)(in1 + off) + 16 * v);
            }
            if (!ok) {
#pragma unroll
                for (int v = 0; v < RV; ++v) {
                    hv[i][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (DIR == 1) hw2[i][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };
    auto halo_store = [&](int tile) {
        const int r0 = (tile % tiles_per_clip) * TH;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int g = tid + NT * i, hy = g / (TW * CJ), px = (g / CJ) % TW;
            if (g >= NITEM) continue;
            unsigned char* d = halo + hy * RP + (px + 1) * PS + 16 * sj;
            if (X3) {
                f32x4 a = hv[i][0], bq = hv[i][RV - 1];
                if (DIR == 1) {
                    const int row = r0 - 1 + hy;
                    const bool in = row >= 0 && row < H;                 // outside the image dy is 0, not cc
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a[q] = in ? ca[q] * a[q] + cb[q] * hw2[i][0][q] + cc[q] : 0.f;
                        bq[q] = in ? ca[4 + q] * bq[q] + cb[4 + q] * hw2[i][RV - 1][q] + cc[4 + q] : 0.f;
                    }
                }
                bf16x8 hi, lo;
                split8(a, bq, hi, lo);
                *(bf16x8*)d = hi;
                *(bf16x8*)(d + HPL) = lo;
            } else {
                if (DIR == 1) {
                    const int row = r0 - 1 + hy;
                    const bool in = row >= 0 && row < H;
                    const bf16x8 z = *(const bf16x8*)&hv[i][0], y = *(const bf16x8*)&hw2[i][0];
                    bf16x8 o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = (__bf16)(in ? ca[q] * (float)z[q] + cb[q] * (float)y[q] + cc[q] : 0.f);
                    *(bf16x8*)d = o;
                } else {
                    *(f32x4*)d = hv[i][0];
                }
            }
        }
    };

    // ---- weight chunk staging -------------------------------------------------------------------------------------------
    // DMA configurations: global_load_lds_dwordx4 - no registers, no ds_write - issued TWO chunks ahead into a ring of three buffers.
    // (The first version staged the next chunk through registers and waited for it at the end of the same iteration: ~500
    // MFMA cycles after the loads were issued, less than an L2 round trip - MFMA busy 0.30.)  One wave-instruction fills 1 KB
    // of LDS linearly (lane l -> byte 16 l), i.e. 64 of the chunk's 16-byte pieces: the rows are BROW = 16 (GJ + 1) bytes, every
    // (GJ + 1)-th piece is row padding (its lane re-reads piece 0: the bytes are never used).  Every wave issues the SAME number
    // of instructions per chunk (NQW = ceil(NQ / NW), the surplus ones land in a dump area) so that the s_waitcnt vmcnt
    // immediates below are compile-time constants.
    constexpr int GJ = KC / 8;                                            // 16-byte groups per chunk row
    constexpr bool DMA = Cfg::DMA;
    constexpr int NQ = (BBUF + 1023) / 1024, NQW = (NQ + Cfg::NW - 1) / Cfg::NW;
    unsigned char* wdump = wb + Cfg::NBUF * PL * BBUF;
    int dsrc[DMA ? NQW : 1];                                              // byte offset into wpk of this lane's piece (chunk 0)
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < NQW; ++i) {
            const int q = wv + Cfg::NW * i, o = 1024 * q + 16 * lane;
            const int row = (o / BROW) < C ? (o / BROW) : C - 1, piece = (o % BROW) / 16;
            dsrc[i] = (row * K + 8 * (piece < GJ ? piece : 0)) * 2;
        }
    }
    auto b_dma = [&](int ch, int buf) {
#pragma unroll
        for (int i = 0; i < NQW; ++i) {
            const int q = wv + Cfg::NW * i;                               // (wave-uniform)
            unsigned char* dst = q < NQ ? wb + buf * BBUF + 1024 * q : wdump;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)wpk + dsrc[i] + ch * (KC * 2)),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    constexpr int BITEM = C * GJ, BL = (BITEM + NT - 1) / NT;
    f32x4 bst[DMA ? 1 : BL][PL];
    auto b_load = [&](int ch) {
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int g = tid + NT * i, row = g / GJ, j = g % GJ;
            if (BITEM % NT == 0 || g < BITEM) {
#pragma unroll
                for (int p = 0; p < PL; ++p)
                    bst[DMA ? 0 : i][p] = *(const f32x4*)((const char*)(wpk + (size_t)p * C * K + (size_t)row * K + (size_t)ch * KC) + 16 * j);
            }
        }
    };
    auto b_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int g = tid + NT * i, row = g / GJ, j = g % GJ;
            if (BITEM % NT == 0 || g < BITEM) {
#pragma unroll
                for (int p = 0; p < PL; ++p) *(f32x4*)(wb + (buf * PL + p) * BBUF + row * BROW + 16 * j) = bst[DMA ? 0 : i][p];
            }
        }
    };

    // ---- this lane's fragment bases -------------------------------------------------------------------------------------
    int a_off[MB];                                                        // halo byte offset of (pixel, tap 0, channel 8 kh)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int q = wm * Cfg::PXW + mb * 32 + n, pr = q / TW, pc = q % TW;
        a_off[mb] = pr * RP + pc * PS + 16 * kh;
    }
    const int b_off = (wn * Cfg::CHW + n) * BROW + 16 * kh;               // + nb * 32 * BROW

    float s1[NB], s2[NB], bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; bv[nb] = (DIR == 0) ? bias[wn * Cfg::CHW + 32 * nb + n] : 0.f; }

    __syncthreads();                                                      // halo zeroed
    const TileWalk walk = xcd_walk(n_tiles);                              // (common.h: consecutive tiles under one L2)
    if (DIR == 0 && walk.first < walk.end) halo_load(walk.first);
    for (int tile = walk.first; tile < walk.end; tile += walk.step) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        if (DIR == 1) halo_load(tile);
        if constexpr (DMA) {
            b_dma(0, 0);
            b_dma(1, 1);
            halo_store(tile);
            // chunk 0 has landed when at most chunk 1's NQW instructions are outstanding (in-order counter: this also waits
            // out the previous tile's epilogue stores and, DIR 1, this tile's halo loads - which halo_store consumed anyway)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQW) : "memory");
        } else {
            b_load(0);
            halo_store(tile);
            b_store(0);
        }
        lds_barrier();                                                    // LDS-only barriers inside the tile loop: __syncthreads()
                                                                          // would drain the halo prefetch / the epilogue's stores
        if (DIR == 0) {       // the next tile's halo flies during this tile's MFMAs (past the end: this tile again, unused)
            const int nt = tile + walk.step;
            halo_load(nt < walk.end ? nt : tile);
        }
        f32x16 acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
        int cbuf = 0;                                                     // ring position of chunk ch (DMA)
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            if constexpr (DMA) {
                // (buffer (ch + 2) % 3 was read in iteration ch - 1: every wave is past that iteration's barrier)
                if (ch + 2 < NCH) b_dma(ch + 2, cbuf == 0 ? 2 : cbuf - 1);
            } else {
                if (ch + 1 < NCH) b_load(ch + 1);
            }
            const int tap = ch / CPT, dr = tap / 3, dc = tap - 3 * dr;
            const int t_off = dr * RP + dc * PS + (ch % CPT) * (KC * 2);
            const unsigned char* bp = wb + (DMA ? cbuf : (ch & 1)) * PL * BBUF + b_off;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                bf16x8 af[MB][PL], bf[NB][PL];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int p = 0; p < PL; ++p) af[mb][p] = *(const bf16x8*)(halo + p * HPL + a_off[mb] + t_off + 32 * ks);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int p = 0; p < PL; ++p) bf[nb][p] = *(const bf16x8*)(bp + p * BBUF + nb * 32 * BROW + 32 * ks);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (X3) {       // small cross terms first, the leading product last
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb][PL - 1], bf[nb][0], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb][0], bf[nb][PL - 1], acc[mb][nb], 0, 0, 0);
                        }
                        if constexpr (F16)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[mb][0]), __builtin_bit_cast(f16x8, bf[nb][0]), acc[mb][nb], 0, 0, 0);
                        else
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb][0], bf[nb][0], acc[mb][nb], 0, 0, 0);
                    }
            }
            if constexpr (DMA) {
                // chunk ch + 1 must have landed before the barrier publishes it: everything issued AFTER its instructions may
                // stay in flight - chunk ch + 2's (this iteration) and, in iteration 0 of a forward tile, the next tile's halo loads
                constexpr int NHL = (DIR == 0) ? NL * RV : 0;
                if (ch == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQW + NHL < 63 ? NQW + NHL : 63) : "memory");
                else if (ch + 2 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                cbuf = cbuf == 2 ? 0 : cbuf + 1;
            } else {
                if (ch + 1 < NCH) b_store((ch + 1) & 1);
            }
            lds_barrier();
        }
        // ---- epilogue: D register r of lane (n, kh) is MFMA row (r & 3) + 8 (r >> 2) + 4 kh, column n ------------------
        // (a tile that lies inside the image - all but the last of a clip - stores without per-row conditions: 64 stores
        // each under its own exec-mask branch were a fifth of the kernel's instructions)
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = wm * Cfg::PXW + mb * 32 + mfma32_row(r, lane);
                    const int row = r0 + q / TW, col = q % TW;
                    if (FULL || row < H) {
                        S* o = out + ((size_t)(b * H + row) * TW + col) * C + wn * Cfg::CHW + n;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const float v = acc[mb][nb][r] + bv[nb];
                            if constexpr (F16) o[32 * nb] = f16_sat(v);
                            else o[32 * nb] = (S)v;
                            if (DIR == 0) { s1[nb] += v; s2[nb] += v * v; }
                        }
                    }
                }
        };
        if (r0 + TH <= H) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    if (DIR == 0 && stat != nullptr) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float a1 = s1[nb] + __shfl_xor(s1[nb], 32), a2 = s2[nb] + __shfl_xor(s2[nb], 32);
            if (kh == 0) {
                const int c = wn * Cfg::CHW + 32 * nb + n;
                red[(wm * 2 + 0) * C + c] = a1; red[(wm * 2 + 1) * C + c] = a2;
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * C; e += NT) {
            const int which = e / C, c = e % C;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) v += (double)red[(w * 2 + which) * C + c];
            atomicAdd(&stat[which * C + c], v);
        }
    }
}


template <int X3, int C, int TW, int TH, int WM, int WN, int DIR>
static int bconv_launch(const void* in0, const void* in1, const float* coef, const void* wpk, const float* bias, void* out,
                        double* stat, int B, int H, hipStream_t st, void* out2 = nullptr) {
    using Cfg = BConvCfg<X3, C, TW, TH, WM, WN>;
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_bconv<X3, C, TW, TH, WM, WN, DIR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)Cfg::LDS_BYTES));
    }
    SED_CHECK_ARG((size_t)B * H * TW * C < ((size_t)1 << 31), "bconv: image too large for 32-bit offsets");
    const int tpc = (H + TH - 1) / TH, nt = B * tpc;
    const int grid = nt < 256 ? nt : 256;
    k_bconv<X3, C, TW, TH, WM, WN, DIR><<<grid, Cfg::NT, Cfg::LDS_BYTES, st>>>(in0, in1, coef, (const __bf16*)wpk, bias, out, stat, H, tpc, nt, (__bf16*)out2);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

template <int DIR>
static int bconv_dispatch(int x3, int C, int W, const void* in0, const void* in1, const float* coef, const void* wpk,
                          const float* bias, void* out, double* stat, int B, int H, hipStream_t st, void* out2 = nullptr) {
    //            X3  C    TW  TH  WM WN
#define BCONV_CASE(XX, CC, WW, HH, MM, NN) \
    if (x3 == XX && C == CC && W == WW) return bconv_launch<XX, CC, WW, HH, MM, NN, DIR>(in0, in1, coef, wpk, bias, out, stat, B, H, st)
    BCONV_CASE(0, 128, 16, 16, 4, 2); BCONV_CASE(0, 64, 16, 32, 8, 1); BCONV_CASE(0, 128, 4, 16, 2, 4); BCONV_CASE(0, 64, 4, 16, 2, 2);
    BCONV_CASE(1, 128, 16, 8, 2, 4);  BCONV_CASE(1, 64, 16, 16, 4, 2); BCONV_CASE(1, 128, 4, 16, 2, 4); BCONV_CASE(1, 64, 4, 16, 2, 2);
#undef BCONV_CASE
    if constexpr (DIR == 0) {
#define BCONV_CASE(XX, CC, WW, HH, MM, NN) \
    if (x3 == XX && C == CC && W == WW) return bconv_launch<XX, CC, WW, HH, MM, NN, DIR>(in0, in1, coef, wpk, bias, out, stat, B, H, st, out2)
        BCONV_CASE(2, 128, 16, 16, 4, 2); BCONV_CASE(2, 64, 16, 32, 8, 1); BCONV_CASE(2, 128, 4, 16, 2, 4); BCONV_CASE(2, 64, 4, 16, 2, 2);
#undef BCONV_CASE
    }
    sed_set_error("bconv: unsupported x3 %d / channels %d / width %d", x3, C, W);
    return SED_ERR_UNSUPPORTED;
}

int launch_bconv_fwd(int x3, int C, const void* in, const void* wpk, const float* bias, void* y, double* stat, int B, int H, int W,
                     hipStream_t st, void* y_bf16_copy) {
    return bconv_dispatch<0>(x3, C, W, in, nullptr, nullptr, wpk, bias, y, stat, B, H, st, y_bf16_copy);
}
int launch_bconv_dgrad(int x3, int C, const void* dz, const void* yin, const float* coef, const void* wpkT, void* dx, int B, int H,
                       int W, hipStream_t st) {
    return bconv_dispatch<1>(x3, C, W, dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, st);
}
