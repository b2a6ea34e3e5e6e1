// gconv.hip - generic 3x3 convolution family (C -> C channels, C in {64, 128}, images [B][H][W][C] with W in {16, 4}),
// forward / dgrad / wgrad, MFMA operands in fp32 (MODE 0) or bf16 (MODE 1).  See gen.h for the shared structure.
//
// Reference ops: Conv2d(C, C, 3, 1, 1) of conv blocks 1 and 2 (baseline/models/CNN.py:46-47) with
// nb_filters = [C, C, C] (CNN.py:35-38 takes any list; BASELINE.json configs[4] uses 128) and its autograd.
//   forward : y[p][co]  = bias[co] + sum_{tap, ci} x[p + tap][ci] W[co][ci][tap]          (+ BatchNorm sum / sum^2)
//   dgrad   : dx[p][ci] = sum_{tap, co} dy[p - tap][co] W[co][ci][tap],  dy = ca*dz + cb*y + cc inside the image
//   wgrad   : dW[co][ci][tap] = sum_p dy[p][co] x[p + tap][ci]
// Forward and dgrad are ONE implicit GEMM kernel (M = 128 output pixels per workgroup, N = C, K = 9 C): the halo tile
// of the input lives in LDS, k-contiguous per pixel; the packed weights [n][tap * C + k] stream through LDS.
#include "gen.h"
#include "kernels.h"
#include "gkernels.h"
#include "gpack.h"

// ---- weight packing (once per forward) ----------------------------------------------------------------------------------
// conv i (1, 2):  wpk [co][tap * C + ci] = W[co][ci][tap]                 (forward B operand, n = co)
//                 wpkT[ci][tap * C + co] = W[co][ci][8 - tap]             (dgrad   B operand, n = ci: flipped kernel)
// GLU i (1, 2), folded with the BatchNorm affine so that the kernels work on xhat = (y - mean) * invstd:
//                 wg  [co][c] = Wglu[co][c] * gamma[c]      bg[co] = bglu[co] + sum_c Wglu[co][c] beta[c]   (fp32)
//                 wgT [c][co] = Wglu[co][c]                               (dz_lin = dlin @ Wglu, n = c)
template <int MODE>
__global__ __launch_bounds__(256) void k_gen_pack(GenPackArgs a) { gen_pack_body<MODE>(a, blockIdx.x * 256 + threadIdx.x); }
__global__ __launch_bounds__(256) void k_gen_pack_bias(GenPackArgs a) { gen_pack_bias_body(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63); }

int launch_gen_pack(const GenPackArgs& a, int mode, hipStream_t st) {
    const int n = 2 * 9 * a.C * a.C;
    const int blocks = ((n > a.n_zero ? n : a.n_zero) + 255) / 256;
    if (mode == 1) k_gen_pack<1><<<blocks, 256, 0, st>>>(a);
    else if (mode == 2) k_gen_pack<2><<<blocks, 256, 0, st>>>(a);
    else k_gen_pack<0><<<blocks, 256, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    k_gen_pack_bias<<<(2 * a.C + 3) / 4, 256, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// ---- forward / dgrad ---------------------------------------------------------------------------------------------------
template <int MODE, int C, int TW>
struct GConvCfg {
    using M = MM<MODE>;
    static constexpr int TH = 128 / TW, HW = TW + 2, HH = TH + 2;
    static constexpr int CS = C + M::PAD;                                   // halo pixel stride (elements)
    static constexpr int HALO_E = HH * HW * CS;
    static constexpr int WBUF_E = 2 * C * (M::KC + M::PAD);
    static constexpr size_t HALO_BYTES = ((size_t)HALO_E * sizeof(typename M::E) + 15) & ~(size_t)15;
    static constexpr size_t LDS_BYTES = HALO_BYTES + (size_t)WBUF_E * sizeof(typename M::E) + 3 * C * 4 + 4 * 2 * C * 4;
};

// DIR 0: forward (in0 = activations; epilogue: bias, BatchNorm sums).  DIR 1: dgrad (in0 = dz, in1 = y, coef = ca | cb | cc).
template <int MODE, int C, int TW, int DIR>
__global__ __launch_bounds__(256) void k_gconv(const float* __restrict__ in0, const float* __restrict__ in1,
                                                const float* __restrict__ coef, const void* __restrict__ wpk_v,
                                                const float* __restrict__ bias, float* __restrict__ out,
                                                double* __restrict__ stat, int H, int tiles_per_clip, int n_tiles) {
    using Cfg = GConvCfg<MODE, C, TW>;
    using M = MM<MODE>;
    using E = typename M::E;
    constexpr int NB = C / 32, TH = Cfg::TH, HW = Cfg::HW, HH = Cfg::HH, CS = Cfg::CS, C4 = C / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    E* halo = (E*)gsm;
    E* wbuf = (E*)(gsm + Cfg::HALO_BYTES);
    float* cf = (float*)(wbuf + Cfg::WBUF_E);            // [3][C] dgrad affine
    float* red = cf + 3 * C;                             // [4 waves][2][C] BatchNorm partial sums
    const E* wpk = (const E*)wpk_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    if (DIR == 1) {
        for (int e = tid; e < 3 * C; e += 256) cf[e] = coef[e];
    }
    // this lane's MFMA row m = n: pixel (r, c) of the tile
    const int pr = (TW == 16) ? 2 * wv + (n >> 4) : 8 * wv + (n >> 2);
    const int pc = (TW == 16) ? (n & 15) : (n & 3);
    const E* a_row = halo + (pr * HW + pc) * CS;
    float s1[NB], s2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; }
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = (DIR == 0) ? bias[32 * nb + n] : 0.f;
    __syncthreads();
    // ---- halo staging: rows r0 - 1 .. r0 + TH, columns -1 .. TW, all C channels; zero outside the image ----------------
    // Split into "issue every load of the tile" and "convert + store to LDS": a load consumed right after its issue costs
    // a full memory round trip per item on these one-workgroup-per-CU kernels (23 items per thread: 30 us per tile against
    // 4 us of bf16 MFMAs in the first version).  The forward kernel goes further and issues the NEXT tile's loads before the
    // current tile's MFMAs, so they are in flight while it computes; dgrad (two source tensors: twice the registers) issues
    // its loads in one batch at the top of the tile.
    constexpr int NL = (HH * HW * C4 + 255) / 256;         // float4 items per thread
    auto halo_load = [&](int tile, f32x4 (&v)[NL], f32x4 (&w)[DIR == 1 ? NL : 1]) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int g = tid + 256 * i, hp = g / C4, c4 = g % C4;
            const int hy = hp / HW, hx = hp % HW;
            const int row = r0 - 1 + hy, col = hx - 1;
            if (DIR == 1) {
                // dgrad: unconditional loads from clamped coordinates - the select happens in halo_store, which needs the
                // in-image predicate anyway (23 predicated loads kept 23 lane masks alive from here to the store:
                // 26 - 38 scalar registers spilled at C = 128)
                const int gc = g < HH * HW * C4 ? g : HH * HW * C4 - 1, hpc = gc / C4, c4c = gc % C4;
                const int rowc = min(max(r0 - 1 + hpc / HW, 0), H - 1), colc = min(max(hpc % HW - 1, 0), TW - 1);
                const size_t off = ((size_t)(b * H + rowc) * TW + colc) * C + 4 * c4c;
                v[i] = *(const f32x4*)(in0 + off);
                w[i] = *(const f32x4*)(in1 + off);
            } else {
                v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (g < HH * HW * C4 && row >= 0 && row < H && col >= 0 && col < TW) {
                    const size_t off = ((size_t)(b * H + row) * TW + col) * C + 4 * c4;
                    v[i] = *(const f32x4*)(in0 + off);
                }
            }
        }
    };
    auto halo_store = [&](int tile, const f32x4 (&v)[NL], const f32x4 (&w)[DIR == 1 ? NL : 1]) {
        const int r0 = (tile % tiles_per_clip) * TH;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int g = tid + 256 * i, hp = g / C4, c4 = g % C4;
            if (g < HH * HW * C4) {
                f32x4 o = v[i];
                if (DIR == 1) {
                    const int hy = hp / HW, hx = hp % HW;
                    const int row = r0 - 1 + hy, col = hx - 1;
                    const bool in = row >= 0 && row < H && col >= 0 && col < TW;      // outside the image dy is 0, not cc
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = in ? cf[4 * c4 + q] * v[i][q] + cf[C + 4 * c4 + q] * w[i][q] + cf[2 * C + 4 * c4 + q] : 0.f;
                }
                M::st4(halo + hp * CS + 4 * c4, o[0], o[1], o[2], o[3]);
            }
        }
    };
    f32x4 hv[NL], hw[DIR == 1 ? NL : 1];
    const TileWalk walk = xcd_walk(n_tiles);                              // (common.h: consecutive tiles under one L2)
    if (DIR == 0 && walk.first < walk.end) halo_load(walk.first, hv, hw);
    for (int tile = walk.first; tile < walk.end; tile += walk.step) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        if (DIR == 1) halo_load(tile, hv, hw);
        halo_store(tile, hv, hw);
        __syncthreads();
        if (DIR == 0) {       // next tile's loads fly during this tile's MFMAs (past the end: re-read this tile, harmless)
            const int nt = tile + walk.step;
            halo_load(nt < walk.end ? nt : tile, hv, hw);
        }
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        constexpr int CPT = C / M::KC;                    // chunks per tap
        auto aoff = [&](int ch) {
            const int tap = ch / CPT, dr = tap / 3, dc = tap % 3;
            return (dr * HW + dc) * CS + (ch % CPT) * M::KC;
        };
        stream_gemm<MODE, NB, NB, M::KC>(a_row, aoff, wpk, 9 * C, 9 * C, wbuf, acc, 0, tid);
        // (stream_gemm ends with a barrier: the halo may be overwritten by the next tile's staging)
        // ---- epilogue: D register r of lane (n, kh) is MFMA row m = (r & 3) + 8 (r >> 2) + 4 kh, column n ------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mfma32_row(r, lane);
            const int rr = (TW == 16) ? 2 * wv + (m >> 4) : 8 * wv + (m >> 2);
            const int cc = (TW == 16) ? (m & 15) : (m & 3);
            const int row = r0 + rr;
            if (row < H) {
                float* o = out + ((size_t)(b * H + row) * TW + cc) * C + n;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v = acc[nb][r] + bv[nb];
                    o[32 * nb] = v;
                    if (DIR == 0) { s1[nb] += v; s2[nb] += v * v; }
                }
            }
        }
    }
    if (DIR == 0 && stat != nullptr) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float a1 = s1[nb] + __shfl_xor(s1[nb], 32), a2 = s2[nb] + __shfl_xor(s2[nb], 32);
            if (kh == 0) { red[(wv * 2 + 0) * C + 32 * nb + n] = a1; red[(wv * 2 + 1) * C + 32 * nb + n] = a2; }
        }
        __syncthreads();
        for (int e = tid; e < 2 * C; e += 256) {
            const int which = e / C, c = e % C;
            const double v = (double)red[(0 * 2 + which) * C + c] + (double)red[(1 * 2 + which) * C + c] +
                             (double)red[(2 * 2 + which) * C + c] + (double)red[(3 * 2 + which) * C + c];
            atomicAdd(&stat[which * C + c], v);
        }
    }
}

template <int MODE, int C, int TW, int DIR>
static int gconv_launch(const float* in0, const float* in1, const float* coef, const void* wpk, const float* bias, float* out,
                        double* stat, int B, int H, hipStream_t st) {
    using Cfg = GConvCfg<MODE, C, TW>;
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gconv<MODE, C, TW, DIR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)Cfg::LDS_BYTES));
    }
    SED_CHECK_ARG((size_t)B * H * TW * C < ((size_t)1 << 31), "gconv: image too large for 32-bit offsets");
    const int tpc = (H + Cfg::TH - 1) / Cfg::TH, nt = B * tpc;
    const int grid = nt < 256 ? nt : 256;
    k_gconv<MODE, C, TW, DIR><<<grid, 256, Cfg::LDS_BYTES, st>>>(in0, in1, coef, wpk, bias, out, stat, H, tpc, nt);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

template <int DIR>
static int gconv_dispatch(int mode, int C, int W, const float* in0, const float* in1, const float* coef, const void* wpk,
                          const float* bias, float* out, double* stat, int B, int H, hipStream_t st) {
#define GCONV_CASE(MD, CC, WW) \
    if (mode == MD && C == CC && W == WW) return gconv_launch<MD, CC, WW, DIR>(in0, in1, coef, wpk, bias, out, stat, B, H, st)
    GCONV_CASE(0, 64, 16); GCONV_CASE(0, 64, 4); GCONV_CASE(0, 128, 16); GCONV_CASE(0, 128, 4);
    GCONV_CASE(1, 64, 16); GCONV_CASE(1, 64, 4); GCONV_CASE(1, 128, 16); GCONV_CASE(1, 128, 4);
#undef GCONV_CASE
    sed_set_error("gconv: unsupported mode %d / channels %d / width %d", mode, C, W);
    return SED_ERR_UNSUPPORTED;
}

int launch_gconv_fwd(int mode, int C, const float* in, const void* wpk, const float* bias, float* y, double* stat, int B, int H,
                     int W, hipStream_t st) {
    return gconv_dispatch<0>(mode, C, W, in, nullptr, nullptr, wpk, bias, y, stat, B, H, st);
}
int launch_gconv_dgrad(int mode, int C, const float* dz, const float* yin, const float* coef, const void* wpkT, float* dx, int B,
                       int H, int W, hipStream_t st) {
    return gconv_dispatch<1>(mode, C, W, dz, yin, coef, wpkT, nullptr, dx, nullptr, B, H, st);
}

// ---- wgrad ---------------------------------------------------------------------------------------------------------------
// dW[co][ci][tap] = sum_p dy[p][co] x[p + tap][ci]: the contraction runs over PIXELS, which is the slow axis of both
// channels-last operands.  The f32 MFMA takes one float per lane per operand, so lane = channel reads of the natural
// [pixel][channel] tiles are already the right fragments (the bf16 MFMA would need both tiles transposed on the way into
// LDS); this kernel therefore computes in fp32 in BOTH modes.  A workgroup owns a 64 x 64 (co, ci) quadrant for all 9 taps
// (9 accumulators of 32 x 32 per wave, 144 registers) over a slab of 128-pixel tiles and writes ONE partial
// [9][64][64] slab; k_gwgrad_reduce adds the slabs in fixed order (bit-reproducible, no float atomics).
template <int TW>
struct GWgCfg {
    static constexpr int TH = 128 / TW, HW = TW + 2, HH = TH + 2, PS = 65;
    static constexpr int X_F = HH * HW * PS, DY_F = 128 * PS;
    static constexpr size_t LDS_BYTES = (size_t)(X_F + DY_F + 3 * 64) * 4;
};
// BF: dz, yin and xin are stored as bf16 (SED_DTYPE_BF16); the products stay fp32 (this kernel serves block 2 in that mode)
template <int TW, int BF>
__global__ __launch_bounds__(256) void k_gwgrad(const void* __restrict__ dz_v, const void* __restrict__ yin_v,
                                                 const float* __restrict__ coef, const void* __restrict__ xin_v,
                                                 float* __restrict__ part, int C, int H, int tiles_per_clip, int n_tiles) {
    using Cfg = GWgCfg<TW>;
    using ST = typename Stor<BF>::T;
    const ST* dz = (const ST*)dz_v;
    const ST* yin = (const ST*)yin_v;
    const ST* xin = (const ST*)xin_v;
    constexpr int TH = Cfg::TH, HW = Cfg::HW, HH = Cfg::HH, PS = Cfg::PS;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* xs = wsm;                       // halo of x: [HH * HW][PS] (this quadrant's 64 input channels)
    float* dys = xs + Cfg::X_F;            // dy: [128][PS] (this quadrant's 64 output channels)
    float* cf = dys + Cfg::DY_F;           // ca | cb | cc of this quadrant's output channels
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int nq = C / 64, quad = blockIdx.y, co0 = (quad / nq) * 64, ci0 = (quad % nq) * 64;
    const int wa = wv >> 1, wb = wv & 1;   // wave's 32 x 32 sub-quadrant: co 32 wa.., ci 32 wb..
    if (tid < 192) cf[tid] = coef[(tid / 64) * C + co0 + (tid % 64)];
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();
    const TileWalk walk = xcd_walk(n_tiles);
    for (int tile = walk.first; tile < walk.end; tile += walk.step) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        __syncthreads();
        for (int g = tid; g < HH * HW * 16; g += 256) {
            const int hp = g >> 4, c4 = g & 15;
            const int hy = hp / HW, hx = hp % HW, row = r0 - 1 + hy, col = hx - 1;
            // unconditional load from a clamped address + select: a load under a per-item condition becomes a branch that
            // waits out its own load (and everything else in flight)
            const bool okx = row >= 0 && row < H && col >= 0 && col < TW;
            f32x4 v = ld4(xin + ((size_t)(b * H + (okx ? row : 0)) * TW + (okx ? col : 0)) * C + ci0 + 4 * c4);
            if (!okx) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            float* d = xs + hp * PS + 4 * c4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        for (int g = tid; g < 128 * 16; g += 256) {
            const int p = g >> 4, c4 = g & 15;
            const int rr = p / TW, cc = p % TW, row = r0 + rr;
            f32x4 v;
            {
                const bool oky = row < H;
                const size_t off = ((size_t)(b * H + (oky ? row : 0)) * TW + cc) * C + co0 + 4 * c4;
                const f32x4 z = ld4(dz + off), y = ld4(yin + off);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = oky ? cf[4 * c4 + q] * z[q] + cf[64 + 4 * c4 + q] * y[q] + cf[128 + 4 * c4 + q] : 0.f;
            }
            float* d = dys + p * PS + 4 * c4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
        // K = 128 pixels, 2 per MFMA: A[i = co][k = pixel] = dy[pixel][co], B[k = pixel][j = ci] = x[pixel + tap][ci]
        const float* Ap = dys + 32 * wa + n;
        const float* Bp = xs + 32 * wb + n;
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
            const int p = 2 * s + kh, rr = p / TW, cc = p % TW;
            const float a = Ap[p * PS];
            const float* bx = Bp + (rr * HW + cc) * PS;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = mfma32(a, bx[((t / 3) * HW + (t % 3)) * PS], acc[t]);
        }
    }
    // partial slab [tap][co (C)][ci (C)] of this workgroup's slab index
    float* ps = part + (size_t)blockIdx.x * 9 * C * C;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ps[((size_t)t * C + co0 + 32 * wa + mfma32_row(r, lane)) * C + ci0 + 32 * wb + n] = acc[t][r];
}

// ---- wgrad on the bf16 MFMA (block 1, W = 16) ----------------------------------------------------------------------------
// The contraction index is the PIXEL, so both operands must reach the MFMA pixel-contiguous: they are transposed on their
// way into LDS.  A thread loads consecutive pixels x 4 channels (8-byte bf16 quads, kept packed) and writes 4 x (8 pixels of
// one channel) as 16-byte bf16 vectors - a register transpose.  With k = r * 16 + c a row shift of the 3x3 tap is an offset of
// 16 elements (32 B: aligned) and is served by ONE copy of the x tile with halo rows; the column shift moves to the OTHER
// operand (summing over x's column c' = c + dc - 1 instead of dy's column c): three copies of dy, each shifted by dc - 1 and
// zero-filled at the tile edge - dy has no halo rows, so its copies are the smaller ones.
//   A[i = co][k = (r, c')] = dyT[dc][co][k] = dy[r][c' - dc + 1]      3 x 64 x (128 + 8) bf16
//   B[k = (r, c')][j = ci] = xT[ci][dr * 16 + k] = x[r + dr - 1][c']      64 x (160 + 8) bf16
// 74 KB of LDS and <= 256 registers (the first version - three copies of the haloed x tile, fp32 staging registers - had one
// wave per SIMD and every load / LDS latency of its in-order stream exposed: MFMA busy 0.135).
// Same decomposition as k_gwgrad: a workgroup owns a 64 x 64 (co, ci) quadrant for all 9 taps over a slab of tiles
// (9 accumulators of 32 x 32 per wave) and writes one partial slab.  The next tile's loads are issued before the
// current tile's MFMAs.
struct GWgB {
    static constexpr int TH = 8, TW = 16, HH = 10, DS = 128 + 8, XS = HH * 16 + 8;
    static constexpr int DY_E = 3 * 64 * DS, X_E = 64 * XS;
    static constexpr size_t STAGE_BYTES = (size_t)(DY_E + X_E) * 2 + 3 * 64 * 4;
    // split-operand mode: a hi and a lo plane of both operands
    static constexpr size_t STAGE_BYTES_X3 = (size_t)(DY_E + X_E) * 4 + 3 * 64 * 4;
};
typedef __attribute__((ext_vector_type(2))) unsigned int gw_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4g;
__device__ __forceinline__ float gw_lo(unsigned int v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float gw_hi(unsigned int v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
// four fp32 values -> packed bf16 pairs of their hi parts and of their lo parts (v - hi, exact in fp32; RNE both times)
__device__ __forceinline__ void gw_split4(const float (&v)[4], gw_u32x2& hi, gw_u32x2& lo) {
    const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    const bf16x4 l = {(__bf16)(v[0] - (float)h[0]), (__bf16)(v[1] - (float)h[1]), (__bf16)(v[2] - (float)h[2]), (__bf16)(v[3] - (float)h[3])};
    hi = __builtin_bit_cast(gw_u32x2, h);
    lo = __builtin_bit_cast(gw_u32x2, l);
}
// EIGHT waves, two groups of four that split the nine TAPS (group 0: taps 0 - 4, group 1: taps 5 - 8) - two waves per SIMD, ONE
// staged copy of the tile and ONE partial slab per workgroup, 80 accumulator registers per wave.  Round 3 split every tile's K
// between the groups instead: 144 accumulator registers + 56 of prefetched tile + fragments did not fit the 256 a wave has
// at two waves per SIMD - 27 registers spilled to scratch - and group 1 handed its sums to group 0 through 144 KB of LDS
// at the end.  A four-wave form with two workgroups per CU (C = 128, round 3) spilled as well and measures the same
// (113 against 112 us in-step at C = 128) with twice the partial slabs for the reduce: removed.
//   X3 = 0  SED_DTYPE_BF16: dz, y, x are stored as bf16; single products; the next tile's loads fly during the MFMAs
//   X3 = 1  SED_DTYPE_BF16X3: fp32 storage; dy and x are split hi + lo on their way into LDS (two planes each, 148 KB) and
//           every tap is hi hi + hi lo + lo hi - the weight gradient of conv block 1 on the bf16 MFMA at ~2^-16 per product
//           (k_gwgrad<16, 0>, the exact-fp32 MFMA kernel it replaces in this mode, was 28 % of the wide step).  The tile
//           is loaded at the top of its iteration (fp32 staging registers for a tile in flight do not fit beside the
//           accumulators).
template <int X3>
__global__ __launch_bounds__(512, 1) void k_gwgrad_bf16(const void* __restrict__ dz_v, const void* __restrict__ yin_v,
                                                       const float* __restrict__ coef, const void* __restrict__ xin_v,
                                                       float* __restrict__ part, int C, int H, int tiles_per_clip, int n_tiles) {
    using M = MM<1>;
    using S = typename std::conditional<X3 != 0, float, __bf16>::type;
    using LV = typename std::conditional<X3 != 0, f32x4, gw_u32x2>::type;        // four channels of one pixel as loaded
    constexpr int NG = 2;                                          // wave groups (the four-wave NG = 1 form is gone, see above)
    constexpr int TH = GWgB::TH, DS = GWgB::DS, XS = GWgB::XS;
    constexpr int NP = X3 ? 2 : 1;                                 // operand planes in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm2[];
    __bf16* dyT = (__bf16*)wsm2;                                   // [plane][3 copies][64][DS]
    __bf16* xT = dyT + NP * GWgB::DY_E;                            // [plane][64][XS]
    float* cf = (float*)(xT + NP * GWgB::X_E);
    const S* dz = (const S*)dz_v;
    const S* yin = (const S*)yin_v;
    const S* xin = (const S*)xin_v;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int nq = C / 64, quad = blockIdx.y, co0 = (quad / nq) * 64, ci0 = (quad % nq) * 64;
    const int grp = wv >> 2, wa = (wv >> 1) & 1, wb = wv & 1;
    if (tid < 192) cf[tid] = coef[(tid / 64) * C + co0 + (tid % 64)];
    constexpr int NT = NG == 2 ? 5 : 9;                            // taps per wave (NG = 2: group 1 uses four of its five)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging items: dy - 256 of (pixel group pg of 8 + one pixel either side, channel quad cq): threads 0 .. 255; x - 320 of
    // (halo row hr, column group cg, channel quad cq): threads 256 .. 511, and 0 .. 63 a second one.  Loads are unconditional
    // from clamped addresses, the selects follow (see k_gwgrad).
    const bool dy_thread = tid < 256;
    const int cq = tid & 15, pg = (tid >> 4) & 15;
    const int x_item = tid >= 256 ? tid - 256 : 256 + tid;
    const bool x_valid = tid >= 256 || tid < 64;
    LV dzv[10], yv[10], xv[8];
    auto load_dy = [&](int tile) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        if (dy_thread) {
            const int r = pg >> 1, c0 = (pg & 1) * 8, row = r0 + r;
            const size_t rbase = (size_t)(b * H + (row < H ? row : 0)) * 16;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int col = c0 - 1 + i, cc = col < 0 ? 0 : (col > 15 ? 15 : col);
                const size_t off = (rbase + cc) * C + co0 + 4 * cq;
                dzv[i] = *(const LV*)(dz + off);
                yv[i] = *(const LV*)(yin + off);
            }
        }
    };
    auto load_x = [&](int tile) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        if (x_valid) {
            const int cg = (x_item >> 4) & 1, hr = (x_item >> 5) < 10 ? (x_item >> 5) : 9;
            const int row = r0 - 1 + hr, rc = row < 0 ? 0 : (row >= H ? H - 1 : row);
            const size_t base = ((size_t)(b * H + rc) * 16 + 8 * cg) * C + ci0 + 4 * cq;
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = *(const LV*)(xin + base + (size_t)i * C);
        }
    };
    // channel q of the four a loaded item holds, as fp32
    auto chan = [&](const LV& v, int q) -> float {
        if constexpr (X3 != 0) return v[q];
        else return (q & 1) ? gw_hi(v[q >> 1]) : gw_lo(v[q >> 1]);
    };
    // dyp[i] = packed bf16 quad of pixel i (10 pixels: the 8 of this group and one either side) -> the three column-shifted,
    // pixel-contiguous copies of one plane.  Copy dc holds dy[r][c' - dc + 1] at column c' = c0 + e: staged pixel index e - dc + 2
    auto put_dy = [&](const gw_u32x2 (&dyp)[10], __bf16* plane, int r, int c0) {
#pragma unroll
        for (int dc = 0; dc < 3; ++dc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned int w[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const unsigned int p0 = dyp[2 * e2 - dc + 2][q >> 1], p1 = dyp[2 * e2 + 1 - dc + 2][q >> 1];
                    w[e2] = (q & 1) ? ((p0 >> 16) | (p1 & 0xffff0000u)) : ((p0 & 0xffffu) | (p1 << 16));
                }
                *(u32x4g*)(plane + ((size_t)(dc * 64 + 4 * cq + q)) * DS + r * 16 + c0) = (u32x4g){w[0], w[1], w[2], w[3]};
            }
    };
    auto put_x = [&](const gw_u32x2 (&xp)[8], __bf16* plane, int hr, int cg, bool ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned int w[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const unsigned int p0 = xp[2 * e2][q >> 1], p1 = xp[2 * e2 + 1][q >> 1];
                const unsigned int v = (q & 1) ? ((p0 >> 16) | (p1 & 0xffff0000u)) : ((p0 & 0xffffu) | (p1 << 16));
                w[e2] = ok ? v : 0u;
            }
            *(u32x4g*)(plane + ((size_t)(4 * cq + q)) * XS + hr * 16 + 8 * cg) = (u32x4g){w[0], w[1], w[2], w[3]};
        }
    };
    // `next`: the tile whose dz / y this thread fetches as soon as the current tile's are converted (bf16 path only) - i.e. BEFORE
    // the 12 transposing LDS writes, the x staging, the barrier and the MFMA phase, not behind the barrier: the kernel moves 123 MB
    // at B = 64 and the loads of a tile (3.1 us per round of 256 workgroups at ~4 TB/s) used to have only the 1.3 us MFMA phase to
    // hide behind (round 6 experiment: without the staging 37 us, without the MFMAs 37 us, with both 50 us)
    auto store_dy = [&](int tile, int next) {
        const int r0 = (tile % tiles_per_clip) * TH;
#ifdef GW_EXP_NOSTAGE      // timing experiment only (tools/build_variant.sh): results are garbage
        if (dzv[0][0] == 0x12345678u && dy_thread) dyT[tid] = (__bf16)1.0f;
        return;
#endif
        if (dy_thread) {
            const int r = pg >> 1, c0 = (pg & 1) * 8;
            const bool rok = r0 + r < H;
            // dy = cf0 dz + cf1 y + cf2 (the BatchNorm backward, per channel) for the 10 pixels, as packed bf16 pairs.
            // X3: one plane at a time (hi, then lo recomputed from the same loads): both planes' packed pixels at once do not fit
            float k0[4], k1[4], k2[4];                             // (read up front: an LDS read under the `ok` select becomes a branch)
#pragma unroll
            for (int q = 0; q < 4; ++q) { k0[q] = cf[4 * cq + q]; k1[q] = cf[64 + 4 * cq + q]; k2[q] = cf[128 + 4 * cq + q]; }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                gw_u32x2 dyp[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int col = c0 - 1 + i;
                    const bool ok = rok && col >= 0 && col < 16;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float val = k0[q] * chan(dzv[i], q) + k1[q] * chan(yv[i], q) + k2[q];
                        v[q] = ok ? val : 0.f;
                    }
                    if constexpr (X3 != 0) {
                        gw_u32x2 h, l;
                        gw_split4(v, h, l);
                        dyp[i] = pl ? l : h;
                    } else {
                        const bf16x4 pk = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        dyp[i] = __builtin_bit_cast(gw_u32x2, pk);
                    }
                }
                if constexpr (X3 == 0) load_dy(next);                 // (dzv / yv are dead from here on)
                put_dy(dyp, dyT + pl * GWgB::DY_E, r, c0);
            }
        }
    };
    auto store_x = [&](int tile, int next) {
        const int r0 = (tile % tiles_per_clip) * TH;
        if (x_valid) {
            const int cg = (x_item >> 4) & 1, hr = x_item >> 5;
            const int row = r0 - 1 + hr;
            const bool ok = row >= 0 && row < H;
            if constexpr (X3 != 0) {
                gw_u32x2 xh[8], xl[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v[4] = {xv[i][0], xv[i][1], xv[i][2], xv[i][3]};
                    gw_split4(v, xh[i], xl[i]);
                }
                put_x(xh, xT, hr, cg, ok);
                put_x(xl, xT + GWgB::X_E, hr, cg, ok);
            } else {
                put_x(xv, xT, hr, cg, ok);
                load_x(next);
            }
        }
    };
    // X3 staging: fp32 sources.  dy: ALL 512 threads take one item of (pixel group pg, channel PAIR cp) - 10 pixels x 2 channels
    // of dz and y as 8-byte loads (40 registers; the 4-channel items of the bf16 path would be 80 here), split into hi / lo
    // packed pairs (one register per pixel and plane) and written as three column-shifted copies per plane.  x: threads
    // 0 .. 319 take one (halo row, column group, channel quad) item each, after the dy phase.
    auto x3_stage = [&](int tile) {
        typedef __attribute__((ext_vector_type(2))) float f32x2g;
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        {
            const int cp = tid & 31, pgx = tid >> 5, r = pgx >> 1, c0 = (pgx & 1) * 8, row = r0 + r;
            const size_t rbase = (size_t)(b * H + (row < H ? row : 0)) * 16;
            f32x2g zv[10], wv2[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int col = c0 - 1 + i, cc = col < 0 ? 0 : (col > 15 ? 15 : col);
                const size_t off = (rbase + cc) * C + co0 + 2 * cp;
                zv[i] = *(const f32x2g*)((const float*)dz_v + off);
                wv2[i] = *(const f32x2g*)((const float*)yin_v + off);
            }
            const bool rok = row < H;
            const float ka0 = cf[2 * cp], ka1 = cf[2 * cp + 1], kb0 = cf[64 + 2 * cp], kb1 = cf[64 + 2 * cp + 1],
                        kc0 = cf[128 + 2 * cp], kc1 = cf[128 + 2 * cp + 1];
            unsigned int ph[10], pl[10];                           // (channel 2 cp | channel 2 cp + 1) as packed bf16, hi and lo parts
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int col = c0 - 1 + i;
                const bool ok = rok && col >= 0 && col < 16;
                const float v0 = ok ? ka0 * zv[i][0] + kb0 * wv2[i][0] + kc0 : 0.f;
                const float v1 = ok ? ka1 * zv[i][1] + kb1 * wv2[i][1] + kc1 : 0.f;
                const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
                ph[i] = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
                pl[i] = (unsigned int)__builtin_bit_cast(unsigned short, l0) | ((unsigned int)__builtin_bit_cast(unsigned short, l1) << 16);
            }
#pragma unroll
            for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned int w[4];
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            const unsigned int p0 = plane ? pl[2 * e2 - dc + 2] : ph[2 * e2 - dc + 2];
                            const unsigned int p1 = plane ? pl[2 * e2 + 1 - dc + 2] : ph[2 * e2 + 1 - dc + 2];
                            w[e2] = q ? ((p0 >> 16) | (p1 & 0xffff0000u)) : ((p0 & 0xffffu) | (p1 << 16));
                        }
                        *(u32x4g*)(dyT + plane * GWgB::DY_E + ((size_t)(dc * 64 + 2 * cp + q)) * DS + r * 16 + c0) = (u32x4g){w[0], w[1], w[2], w[3]};
                    }
        }
        asm volatile("" ::: "memory");                             // (the x loads are not to be hoisted above the dy conversion)
        if (tid < 320) {
            const int cg = (tid >> 4) & 1, hr = tid >> 5, cqx = tid & 15;
            const int row = r0 - 1 + hr, rc = row < 0 ? 0 : (row >= H ? H - 1 : row);
            const bool ok = row >= 0 && row < H;
            const float* base = (const float*)xin_v + ((size_t)(b * H + rc) * 16 + 8 * cg) * C + ci0 + 4 * cqx;
            f32x4 xf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[i] = *(const f32x4*)(base + (size_t)i * C);
            gw_u32x2 xh[8], xl[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v[4] = {xf[i][0], xf[i][1], xf[i][2], xf[i][3]};
                gw_split4(v, xh[i], xl[i]);
            }
#pragma unroll
            for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned int w[4];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const unsigned int p0 = plane ? xl[2 * e2][q >> 1] : xh[2 * e2][q >> 1];
                        const unsigned int p1 = plane ? xl[2 * e2 + 1][q >> 1] : xh[2 * e2 + 1][q >> 1];
                        const unsigned int v = (q & 1) ? ((p0 >> 16) | (p1 & 0xffff0000u)) : ((p0 & 0xffffu) | (p1 << 16));
                        w[e2] = ok ? v : 0u;
                    }
                    *(u32x4g*)(xT + plane * GWgB::X_E + ((size_t)(4 * cqx + q)) * XS + hr * 16 + 8 * cg) = (u32x4g){w[0], w[1], w[2], w[3]};
                }
        }
    };
    __syncthreads();
    const TileWalk walk = xcd_walk(n_tiles);
    if (!X3 && walk.first < walk.end) { load_dy(walk.first); load_x(walk.first); }
    for (int tile = walk.first; tile < walk.end; tile += walk.step) {
        if constexpr (X3 != 0) {
            x3_stage(tile);
        } else {
            const int nt = tile + walk.step, nxt = nt < walk.end ? nt : tile;
            store_dy(tile, nxt);
            store_x(tile, nxt);
        }
        __syncthreads();
        const __bf16* Ap = dyT + (size_t)(32 * wa + n) * DS + 8 * kh;
        const __bf16* Bp = xT + (size_t)(32 * wb + n) * XS + 8 * kh;
        // tap t = 3 dr + dc multiplies dy copy dc (A) with x rows shifted by dr (B)
        auto taps = [&](auto first, auto count) {
            constexpr int T0 = decltype(first)::value, TN = decltype(count)::value;
            constexpr int DR0 = T0 / 3, DR1 = (T0 + TN - 1) / 3;
#ifndef GW_KS_UNROLL
#define GW_KS_UNROLL 4      // (1 -> 4: 52.0 -> 49.5 us solo at B = 64, C = 64; the fragment reads of the next k-steps fly under the MFMAs)
#endif
#pragma unroll GW_KS_UNROLL
            for (int ks = 0; ks < 8; ++ks) {
                bf16x8 a[3], bx[3], al[X3 ? 3 : 1], bl[X3 ? 3 : 1];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    a[d] = *(const bf16x8*)(Ap + (size_t)d * 64 * DS + 16 * ks);
                    if constexpr (X3 != 0) al[d] = *(const bf16x8*)(Ap + GWgB::DY_E + (size_t)d * 64 * DS + 16 * ks);
                }
#pragma unroll
                for (int d = DR0; d <= DR1; ++d) {
                    bx[d] = *(const bf16x8*)(Bp + d * 16 + 16 * ks);
                    if constexpr (X3 != 0) bl[d] = *(const bf16x8*)(Bp + GWgB::X_E + d * 16 + 16 * ks);
                }
#pragma unroll
                for (int t = T0; t < T0 + TN; ++t) {
                    acc[t - T0] = M::mma(a[t % 3], bx[t / 3], acc[t - T0]);
                    if constexpr (X3 != 0) {
                        acc[t - T0] = M::mma(a[t % 3], bl[t / 3], acc[t - T0]);
                        acc[t - T0] = M::mma(al[t % 3], bx[t / 3], acc[t - T0]);
                    }
                }
            }
        };
#ifndef GW_EXP_NOMMA       // timing experiment only
        if (grp == 0) taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
        else taps(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
#endif
        __syncthreads();
    }
    // every wave writes its own taps of the partial slab [tap][co (C)][ci (C)]
    float* ps = part + (size_t)blockIdx.x * 9 * C * C;
    const int t0 = 5 * grp, tn = grp ? 4 : 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= tn) break;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ps[((size_t)(t0 + t) * C + co0 + 32 * wa + mfma32_row(r, lane)) * C + ci0 + 32 * wb + n] = acc[t][r];
    }
}

// ---- wgrad on the bf16 MFMA, W = 4 (block 2) ------------------------------------------------------------------------------
// 15 000 pixels (B = 24) x C x 9 C: 1 - 4 GFLOP, a latency problem.  The pixel-slab decomposition above costs more in partial
// slabs than in work here (every workgroup writes all 9 C^2 sums: 37 MB of partials + the reduce = 48 us at C = 64), so this
// kernel is OUTPUT-stationary: a workgroup owns one 32 x 32 (co, ci) block for the three taps of ONE kernel row dr over a long
// run of tiles (every n_slab-th tile of 32 rows x 4 columns = 128 pixels = K), its four waves split K (two k-steps each) and
// meet in LDS at the end - 12 KB of partials per workgroup, 20 slabs.  With dr fixed the x tile is loaded already shifted by
// dr - 1 rows (no halo, aligned fragment reads); the column shift sits on dy as in k_gwgrad_bf16: three copies, shifted
// inside each 4-pixel row and zero-filled at its ends (a row never needs its neighbours' pixels).
struct GWg4 {
    static constexpr int TH = 32, DS = 128 + 8;
    static constexpr int DY_E = 3 * 32 * DS, X_E = 32 * DS;
    static constexpr size_t STAGE_BYTES = (size_t)(DY_E + X_E) * 2 + 3 * 32 * 4, RED_BYTES = (size_t)3 * 3 * 16 * 64 * 4;
    static constexpr size_t LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
};
__global__ __launch_bounds__(256) void k_gwgrad4_bf16(const __bf16* __restrict__ dz, const __bf16* __restrict__ yin,
                                                       const float* __restrict__ coef, const __bf16* __restrict__ xin,
                                                       float* __restrict__ part, int C, int H, int tiles_per_clip, int n_tiles) {
    using M = MM<1>;
    constexpr int TH = GWg4::TH, DS = GWg4::DS;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm4[];
    __bf16* dyT = (__bf16*)wsm4;
    __bf16* xT = dyT + GWg4::DY_E;
    float* cf = (float*)(xT + GWg4::X_E);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, kh = lane >> 5;
    const int ncb = C / 32, dr = blockIdx.y % 3, quad = blockIdx.y / 3, co0 = (quad / ncb) * 32, ci0 = (quad % ncb) * 32;
    if (tid < 96) cf[tid] = coef[(tid / 32) * C + co0 + (tid % 32)];
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging items (8 consecutive pixels = two image rows, 4 channels): threads 0 .. 127 dy (dz and y), 128 .. 255 x
    const bool dy_thread = tid < 128;
    const int it = tid & 127, cq = it & 7, pg = it >> 3;
    gw_u32x2 va[8], vb[8];                                         // dy threads: dz, y;  x threads: x (va)
    auto load = [&](int tile) {
        const int b = tile / tiles_per_clip, r0 = (tile % tiles_per_clip) * TH;
        const int sh = dy_thread ? 0 : dr - 1, c0 = dy_thread ? co0 : ci0;
        const __bf16* src = dy_thread ? dz : xin;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                               // the item's two rows (unconditional loads, clamped rows)
            const int row = r0 + 2 * pg + hf + sh, rc = row < 0 ? 0 : (row >= H ? H - 1 : row);
            const size_t off = ((size_t)(b * H + rc) * 4) * C + c0 + 4 * cq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                va[4 * hf + i] = *(const gw_u32x2*)(src + off + (size_t)i * C);
                if (dy_thread) vb[4 * hf + i] = *(const gw_u32x2*)(yin + off + (size_t)i * C);
            }
        }
    };
    auto pack8 = [&](const unsigned int (&pw)[8][2], int q, const int (&idx)[8]) {       // channel q of 8 pixels -> 16 bytes
        unsigned int w[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int i0 = idx[2 * e2], i1 = idx[2 * e2 + 1];
            const unsigned int p0 = i0 < 0 ? 0u : pw[i0 < 0 ? 0 : i0][q >> 1], p1 = i1 < 0 ? 0u : pw[i1 < 0 ? 0 : i1][q >> 1];
            w[e2] = (q & 1) ? ((p0 >> 16) | (p1 & 0xffff0000u)) : ((p0 & 0xffffu) | (p1 << 16));
        }
        return (u32x4g){w[0], w[1], w[2], w[3]};
    };
    auto store = [&](int tile) {
        const int r0 = (tile % tiles_per_clip) * TH;
        unsigned int pw[8][2];
        if (dy_thread) {
            float k0[4], k1[4], k2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { k0[q] = cf[4 * cq + q]; k1[q] = cf[32 + 4 * cq + q]; k2[q] = cf[64 + 4 * cq + q]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = r0 + 2 * pg + (i >> 2) < H;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned int zw = va[i][q >> 1], yw = vb[i][q >> 1];
                    const float z = (q & 1) ? gw_hi(zw) : gw_lo(zw), y = (q & 1) ? gw_hi(yw) : gw_lo(yw);
                    const float val = k0[q] * z + k1[q] * y + k2[q];
                    v[q] = ok ? val : 0.f;
                }
                const bf16x4 pk = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                const gw_u32x2 t2 = __builtin_bit_cast(gw_u32x2, pk);
                pw[i][0] = t2.x; pw[i][1] = t2.y;
            }
            // copy dc holds dy[r][c' - dc + 1] at column c' (zero outside the row): x's column is the summation index
#pragma unroll
            for (int dc = 0; dc < 3; ++dc) {
                int idx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int sc = (e & 3) - dc + 1; idx[e] = (sc >= 0 && sc < 4) ? (e & 4) + sc : -1; }
#pragma unroll
                for (int q = 0; q < 4; ++q) *(u32x4g*)(dyT + ((size_t)(dc * 32 + 4 * cq + q)) * DS + 8 * pg) = pack8(pw, q, idx);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = r0 + 2 * pg + (i >> 2) + dr - 1;
                const bool ok = row >= 0 && row < H;
                pw[i][0] = ok ? va[i].x : 0u; pw[i][1] = ok ? va[i].y : 0u;
            }
            const int idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4g*)(xT + ((size_t)(4 * cq + q)) * DS + 8 * pg) = pack8(pw, q, idx);
        }
    };
    __syncthreads();
    const TileWalk walk = xcd_walk(n_tiles);
    if (walk.first < walk.end) load(walk.first);
    for (int tile = walk.first; tile < walk.end; tile += walk.step) {
        store(tile);
        __syncthreads();
        {
            const int nt = tile + walk.step;
            load(nt < walk.end ? nt : tile);
        }
        const __bf16* Ap = dyT + (size_t)n * DS + 8 * kh + 32 * wv;
        const __bf16* Bp = xT + (size_t)n * DS + 8 * kh + 32 * wv;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bx = *(const bf16x8*)(Bp + 16 * ks);
#pragma unroll
            for (int dc = 0; dc < 3; ++dc) acc[dc] = M::mma(*(const bf16x8*)(Ap + (size_t)dc * 32 * DS + 16 * ks), bx, acc[dc]);
        }
        __syncthreads();
    }
    // waves 1 .. 3 -> LDS, wave 0 adds and writes the partial block
    float* red = (float*)wsm4;
    if (wv > 0) {
#pragma unroll
        for (int dc = 0; dc < 3; ++dc)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wv - 1) * 3 + dc) * 16 + r) * 64 + lane] = acc[dc][r];
    }
    __syncthreads();
    if (wv == 0) {
        float* ps = part + (size_t)blockIdx.x * 9 * C * C;
#pragma unroll
        for (int dc = 0; dc < 3; ++dc)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[dc][r] + red[((0 * 3 + dc) * 16 + r) * 64 + lane] + red[((1 * 3 + dc) * 16 + r) * 64 + lane] +
                                red[((2 * 3 + dc) * 16 + r) * 64 + lane];
                ps[((size_t)(dr * 3 + dc) * C + co0 + mfma32_row(r, lane)) * C + ci0 + n] = v;
            }
    }
}

// g_w[co][ci][tap] = sum over slabs, fixed order
__global__ __launch_bounds__(256) void k_gwgrad_reduce(const float* __restrict__ part, int n_slabs, int C, float* __restrict__ g_w) {
    const int e = blockIdx.x * 256 + threadIdx.x;        // e = (tap * C + co) * C + ci
    const int n = 9 * C * C;
    if (e >= n) return;
    float s = 0.f;
    int k = 0;
    // 16 slabs per trip, every load issued before the first add (4 per trip left the kernel at one L2 / HBM round trip per 4
    // slabs: 27 us for 256 slabs of 147 KB); the additions keep the slab order
    for (; k + 16 <= n_slabs; k += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = part[(size_t)(k + i) * n + e];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    for (; k + 4 <= n_slabs; k += 4) {
        const float v0 = part[(size_t)k * n + e], v1 = part[(size_t)(k + 1) * n + e], v2 = part[(size_t)(k + 2) * n + e],
                    v3 = part[(size_t)(k + 3) * n + e];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < n_slabs; ++k) s += part[(size_t)k * n + e];
    const int tap = e / (C * C), co = (e / C) % C, ci = e % C;
    g_w[((size_t)co * C + ci) * 9 + tap] = s;
}

// partial slabs (one workgroup per CU and (co, ci) quadrant)
int gwgrad_slabs(int C) { return 256 / ((C / 64) * (C / 64)); }

int launch_gwgrad(int mode, int C, const void* dz, const void* yin, const float* coef, const void* xin, float* part, float* g_w,
                  int B, int H, int W, hipStream_t st) {
    SED_CHECK_ARG(C == 64 || C == 128, "gwgrad: C must be 64 or 128");
    const int nq = (C / 64) * (C / 64);
    int slabs = gwgrad_slabs(C);
    int nt, tpc;
    if (W == 16 && (mode == SED_DTYPE_BF16 || (mode == SED_DTYPE_BF16X3 && !(g_sed_debug & 4194304)))) {
        // (debug bit 22: the split-operand mode's weight gradient on the exact-fp32 MFMA kernel, as in round 3 - A/B)
        static thread_local SedAttrOnce attr;
        if (attr.need()) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gwgrad_bf16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GWgB::STAGE_BYTES));
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gwgrad_bf16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GWgB::STAGE_BYTES_X3));
        }
        tpc = (H + GWgB::TH - 1) / GWgB::TH; nt = B * tpc;
        if (slabs > nt) slabs = nt;
        if (mode == SED_DTYPE_BF16) k_gwgrad_bf16<0><<<dim3(slabs, nq), 512, GWgB::STAGE_BYTES, st>>>(dz, yin, coef, xin, part, C, H, tpc, nt);
        else k_gwgrad_bf16<1><<<dim3(slabs, nq), 512, GWgB::STAGE_BYTES_X3, st>>>(dz, yin, coef, xin, part, C, H, tpc, nt);
    } else if (W == 16) {
        using Cfg = GWgCfg<16>;
        static thread_local SedAttrOnce attr;
        if (attr.need()) { SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gwgrad<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES)); }
        tpc = (H + Cfg::TH - 1) / Cfg::TH; nt = B * tpc;
        if (slabs > nt) slabs = nt;
        k_gwgrad<16, 0><<<dim3(slabs, nq), 256, Cfg::LDS_BYTES, st>>>(dz, yin, coef, xin, part, C, H, tpc, nt);
    } else if (W == 4 && mode == SED_DTYPE_BF16) {
        static thread_local SedAttrOnce attr;
        if (attr.need()) { SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gwgrad4_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GWg4::LDS_BYTES)); }
        tpc = (H + GWg4::TH - 1) / GWg4::TH; nt = B * tpc;
        const int blocks = (C / 32) * (C / 32) * 3;                        // (co block, ci block, kernel row) per slab
        slabs = (480 + blocks - 1) / blocks;                               // ~two workgroups per CU (37 KB of LDS, 144 registers)
        if (slabs > nt) slabs = nt;
        if (slabs > gwgrad_slabs(C)) slabs = gwgrad_slabs(C);
        k_gwgrad4_bf16<<<dim3(slabs, blocks), 256, GWg4::LDS_BYTES, st>>>((const __bf16*)dz, (const __bf16*)yin, coef, (const __bf16*)xin, part, C, H, tpc, nt);
    } else if (W == 4) {
        using Cfg = GWgCfg<4>;
        static thread_local SedAttrOnce attr;
        if (attr.need()) {
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gwgrad<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
            SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gwgrad<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
        }
        tpc = (H + Cfg::TH - 1) / Cfg::TH; nt = B * tpc;
        if (slabs > nt) slabs = nt;
        if (mode == SED_DTYPE_BF16) k_gwgrad<4, 1><<<dim3(slabs, nq), 256, Cfg::LDS_BYTES, st>>>(dz, yin, coef, xin, part, C, H, tpc, nt);
        else k_gwgrad<4, 0><<<dim3(slabs, nq), 256, Cfg::LDS_BYTES, st>>>(dz, yin, coef, xin, part, C, H, tpc, nt);
    } else {
        sed_set_error("gwgrad: unsupported width %d", W);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    k_gwgrad_reduce<<<(9 * C * C + 255) / 256, 256, 0, st>>>(part, slabs, C, g_w);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
