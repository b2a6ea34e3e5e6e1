// blk0.hip - conv block 0 (1 -> 64 channels) fused end to end.
//
// Reference ops (baseline/models/CNN.py:46-67, GLU CNN.py:11-16):
//   u = Conv2d(1,64,3,1,1)(x);  z = BatchNorm2d(u);  out = AvgPool2d((2,4))(Dropout(GLU(z)))
//   GLU(z) = (Wglu z + bglu) * sigmoid(z)
//
// MI355X design: everything up to the GLU's Linear is affine in the 3x3 input patch P (9 taps +
// a constant 1), so with the BN scale/shift folded in
//     z[c]    = sum_t wz[c][t] * P[t]                       (wz = scale*w0, wz[9] = scale*b0+shift)
//     lin[co] = sum_t wl[co][t] * P[t]                      (wl = Wglu @ wz, wl[9] += bglu)
// i.e. block 0 is ONE 1->128-channel 3x3 conv (K = 10) followed by lin*sigmoid(z), dropout and
// pooling.  The 64x64 per-pixel GLU GEMM (38% of the reference's forward FLOPs) disappears, and
// the full-resolution tensors [B,64,T,64] (75% of the reference's activation bytes) are never
// written: forward reads x and writes the pooled p0; the train-mode BatchNorm statistics come
// from the 9+45 first/second moments of the patch (k_x_moments), and backward reduces to the
// sums D = dlin^T P, E = dzgate^T P (two 64x10 matrices) from which k_blk0_bwd_finalize derives
// every parameter gradient of the block in fp64.
#include <type_traits>
#include "common.h"
#include "philox.h"
#include "kernels.h"
#include "gpack.h"

SED_TS_DEFINE(blk0)
#define XS_W 66
#define XS_H 10
#define FXS_H 6       // forward tile: 2 pooled rows = 4 input rows + halo

__device__ __forceinline__ int gidx(int a, int b) { return 9 + a * 9 - (a * (a - 1)) / 2 + (b - a); }  // a <= b

// ---- patch moments: s[t] = sum_p P[p][t], G[a][b] = sum_p P[p][a] P[p][b] (upper triangle) ----
// The 10x10 Gram matrix of the patch matrix P~ = [9 taps | 1] holds both (s = G~[.][9]).  It is a rank-4
// update per MFMA: v_mfma_f32_16x16x4_f32 with A[i][k] = B[k][i] = P~[pixel k][tap i], i.e. the SAME
// register feeds both operands - one LDS read per 4 pixels per lane, no cross-lane reduction at all
// (the first version did 54 FMAs per pixel on the VALU and then 54 six-step shuffle reductions: 22-32 us).
// fp32 accumulation runs over 1 024 pixels per wave; the cross-wave sums are fp64 and go to a per-workgroup
// partial row which k_blk0_prep (the next kernel anyway) adds up in fixed order: no memset, no same-address
// fp64 atomics (240 of them per address cost 7.5 us), bit-reproducible statistics.
#define MOM_ROWS 64       // (32 rows = two workgroups per CU: the moments 14.4 -> 11 us, but twice the partials cost k_blk0_prep more)
// Workgroups with blockIdx.x >= nx do the conv1 / conv2 weight packing of the same step instead (conv_pack_body,
// kernels.h): independent work that used to be a 6 us launch of its own on the forward chain.
__device__ __forceinline__ void x_moments_body(const float* __restrict__ x, int T, double* __restrict__ part, int nx);
__global__ __launch_bounds__(256) void k_x_moments(const float* __restrict__ x, int T, double* __restrict__ part, int nx,
                                                    ConvPackArgs pack) {
    if ((int)blockIdx.x >= nx) {
        const int pb = ((int)blockIdx.x - nx) * (int)gridDim.y + (int)blockIdx.y;
        if (pb < SED_PACK_BLOCKS) conv_pack_body(pack, pb * 256 + (int)threadIdx.x);
        return;
    }
    x_moments_body(x, T, part, nx);
}
// the generic kernel set's packing work in the spare workgroups (gpack.h)
__global__ __launch_bounds__(256) void k_x_moments_aux(const float* __restrict__ x, int T, double* __restrict__ part, int nx,
                                                        GenAuxPack aux, int n_aux) {
    if ((int)blockIdx.x >= nx) {
        const int pb = ((int)blockIdx.x - nx) * (int)gridDim.y + (int)blockIdx.y;
        if (pb < n_aux) gen_aux_body(aux, pb, (int)threadIdx.x);
        return;
    }
    x_moments_body(x, T, part, nx);
}
__device__ __forceinline__ void x_moments_body(const float* __restrict__ x, int T, double* __restrict__ part, int nx) {
    __shared__ float xs[(MOM_ROWS + 2) * XS_W];
    __shared__ float red[4][10][10];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * MOM_ROWS;
    // the tile (rows t0-1 .. t0+MOM_ROWS of 64 floats) is contiguous in x: float4 loads, zero rows outside the clip
#pragma unroll
    for (int k = 0; k < ((MOM_ROWS + 2) * 16 + 255) / 256; ++k) {
        const int e = tid + 256 * k;
        if (e < (MOM_ROWS + 2) * 16) {
            const int r = e >> 4, c4 = e & 15, t = t0 - 1 + r;
            // unconditional load from a clamped row, zeroed afterwards: under `if (in range)` every one of the five loads of a
            // thread was a branch with its own s_waitcnt - five serialized HBM round trips, half of this kernel's 15 us
            const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
            float4 v = *(const float4*)&x[((size_t)b * T + tc) * 64 + 4 * c4];
            if (t < 0 || t >= T) v = float4{0.f, 0.f, 0.f, 0.f};
            float* d = &xs[r * XS_W + 1 + 4 * c4];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    if (tid < 2 * (MOM_ROWS + 2)) xs[(tid >> 1) * XS_W + (tid & 1) * 65] = 0.f;
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int toff = (i < 9) ? (i / 3) * XS_W + (i % 3) : 0;
    const float tap_mul = (i < 9) ? 1.0f : 0.f, tap_add = (i == 9) ? 1.0f : 0.f;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < MOM_ROWS / 4; ++r) {
        const int row = (MOM_ROWS / 4) * wv + r;
        if (t0 + row >= T) break;                      // wave-uniform
        const float* base = xs + row * XS_W + kq + toff;
#pragma unroll
        for (int c4 = 0; c4 < 16; c4 += 2) {
            const float a0 = fmaf(base[4 * c4], tap_mul, tap_add);
            const float a1 = fmaf(base[4 * c4 + 4], tap_mul, tap_add);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc1, 0, 0, 0);
        }
    }
    // D layout of the 16x16 tile: lane (j = lane & 15, q = lane >> 4), register r -> row 4q + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        if (row < 10 && i < 10) red[wv][row][i] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (tid < 100) {
        const int a = tid / 10, c = tid % 10;
        if (a <= c && a < 9) {
            const double v = (double)red[0][a][c] + (double)red[1][a][c] + (double)red[2][a][c] + (double)red[3][a][c];
            part[(size_t)(blockIdx.y * nx + blockIdx.x) * 54 + (c == 9 ? a : gidx(a, c))] = v;
        }
    }
}

// ---- fold BN into the conv weights; update running stats --------------------------------------
struct Blk0PrepArgs {
    const float *w0, *b0, *gamma, *beta, *wglu, *bglu;
    float *run_mean, *run_var;
    int64_t* tracked;
    double* mom;             // [54] s | G, written here from the partials (train) for the backward finalize
    const double* mompart;   // [n_part][54] per-workgroup partial moments of k_x_moments
    int n_part;
    double N;
    int train, update;
    float eps, momentum;
    float *wz, *wl, *bn;   // wz/wl [C][12], bn [4][C] = mean, invstd, scale, shift
    int C;                 // conv filters of block 0 (64 on the reference's configuration, 128 for the wide one)
};
#define PREP_THREADS 896      // 54 moments x 16 partial-sum lanes = 864 threads for the reduction; >= 640 for the GLU fold
__device__ __forceinline__ void blk0_prep_body(const Blk0PrepArgs& a);
__global__ __launch_bounds__(PREP_THREADS) void k_blk0_prep(Blk0PrepArgs a) { blk0_prep_body(a); }
// Round 6: a training forward whose patch moments are ALREADY in ctx (sed_crnn_moments ran for this batch during the previous
// step, off the critical chain) has no moments launch for the per-forward weight packing to ride in; it rides here instead:
// workgroup 0 is the prep, workgroups 1 .. do the packing with their first 256 threads (the bodies are written for 256-thread
// blocks; the other waves leave at once - s_barrier does not wait for terminated waves).
__global__ __launch_bounds__(PREP_THREADS) void k_blk0_prep_pack(Blk0PrepArgs a, ConvPackArgs pack) {
    if (blockIdx.x == 0) { blk0_prep_body(a); return; }
    if (threadIdx.x >= 256) return;
    const int pb = (int)blockIdx.x - 1;
    if (pb < SED_PACK_BLOCKS) conv_pack_body(pack, pb * 256 + (int)threadIdx.x);
}
__global__ __launch_bounds__(PREP_THREADS) void k_blk0_prep_aux(Blk0PrepArgs a, GenAuxPack aux, int n_aux) {
    if (blockIdx.x == 0) { blk0_prep_body(a); return; }
    if (threadIdx.x >= 256) return;
    const int pb = (int)blockIdx.x - 1;
    if (pb < n_aux) gen_aux_body(aux, pb, (int)threadIdx.x);
}
__device__ __forceinline__ void blk0_prep_body(const Blk0PrepArgs& a) {
    __shared__ double wzs[128][10];
    const int C = a.C;
    __shared__ double mred[54][16];
    __shared__ double moms[54];
    __shared__ float wgl[128 * 129];          // Wglu, staged for the fold at the end; row stride C + 1: the fold's lanes read one k of
                                              // ~7 different rows at a time - at stride C = 64 / 128 all of them in one bank
    const int tid = threadIdx.x;
    // Everything this one-workgroup kernel reads is requested up front, so that its ~10 us are ONE memory round trip instead
    // of three in a row (partials -> per-channel parameters -> Wglu): Wglu as float4 into registers (<= 5 per thread), the
    // per-channel conv / BatchNorm parameters, then the moment partials below.
    constexpr int NWG = (128 * 128 / 4 + PREP_THREADS - 1) / PREP_THREADS;
    f32x4 wgv[NWG];
    const int nwg4 = C * C / 4;
#pragma unroll
    for (int u = 0; u < NWG; ++u) {
        const int e = tid + PREP_THREADS * u;
        wgv[u] = *(const f32x4*)(a.wglu + 4 * (e < nwg4 ? e : 0));
    }
    float w0r[9], b0r = 0.f, gamr = 0.f, betr = 0.f, rmr = 0.f, rvr = 0.f;
    {
        const int c = tid < C ? tid : 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) w0r[t] = a.w0[c * 9 + t];
        b0r = a.b0[c]; gamr = a.gamma[c]; betr = a.beta[c];
        if (a.run_mean != nullptr && a.run_var != nullptr) { rmr = a.run_mean[c]; rvr = a.run_var[c]; }
    }
    if (a.train) {   // patch moments = fixed-order fp64 sum of the per-workgroup partials
        if (tid < 864) {
            const int k = tid / 16, j = tid % 16;
            // thread (k, j) takes partials j, j + 16, ...: up to 16 independent loads in flight, so the 240 partials of the
            // BASELINE shape are ONE memory round trip (10 lanes x 8 loads were three: 6 of this kernel's 10 us)
            double acc = 0;
            int w = j;
            for (; w + 16 * 15 < a.n_part; w += 16 * 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = a.mompart[(size_t)(w + 16 * u) * 54 + k];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u];
            }
            {   // the remaining (< 16) partials of this lane, still issued together
                // (clamped index, then select: written as `in range ? load : 0` every load was a branch with its own s_waitcnt -
                // up to fifteen serialized round trips in this one-workgroup kernel on the head of every forward)
                double v[15];
#pragma unroll
                for (int u = 0; u < 15; ++u) {
                    const int wi = w + 16 * u;
                    v[u] = a.mompart[(size_t)(wi < a.n_part ? wi : a.n_part - 1) * 54 + k];
                }
#pragma unroll
                for (int u = 0; u < 15; ++u) acc += (w + 16 * u < a.n_part) ? v[u] : 0.0;
            }
            mred[k][j] = acc;
        }
        __syncthreads();
        if (tid < 54) {
            double acc = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += mred[tid][j];
            moms[tid] = acc;
            a.mom[tid] = acc;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NWG; ++u) {
        const int e = tid + PREP_THREADS * u;
        if (e < nwg4) {
            float* d = &wgl[(4 * e / C) * (C + 1) + (4 * e % C)];
            d[0] = wgv[u][0]; d[1] = wgv[u][1]; d[2] = wgv[u][2]; d[3] = wgv[u][3];
        }
    }
    if (tid < C) {
        const int c = tid;
        double w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = w0r[t];
        const double b = b0r;
        double mean, var;
        if (a.train) {
            double ws = 0, wGw = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) ws += w[t] * moms[t];
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = 0; j < 9; ++j) wGw += w[i] * w[j] * moms[i <= j ? gidx(i, j) : gidx(j, i)];
            const double mu = ws / a.N;
            mean = mu + b;
            var = wGw / a.N - mu * mu;
            if (var < 0) var = 0;
            if (a.update) {
                a.run_mean[c] = (float)((1.0 - a.momentum) * rmr + a.momentum * mean);
                a.run_var[c] = (float)((1.0 - a.momentum) * rvr + a.momentum * var * a.N / (a.N - 1.0));
                if (c == 0 && a.tracked) a.tracked[0] += 1;
            }
        } else {
            mean = rmr;
            var = rvr;
        }
        const double invstd = 1.0 / sqrt(var + (double)a.eps);
        const double scale = gamr * invstd;
        const double shift = betr - mean * scale;
#pragma unroll
        for (int t = 0; t < 9; ++t) wzs[c][t] = scale * w[t];
        wzs[c][9] = scale * b + shift;
#pragma unroll
        for (int t = 0; t < 10; ++t) a.wz[c * 12 + t] = (float)wzs[c][t];
        a.wz[c * 12 + 10] = 0.f; a.wz[c * 12 + 11] = 0.f;
        a.wl[c * 12 + 10] = 0.f; a.wl[c * 12 + 11] = 0.f;
        a.bn[c] = (float)mean; a.bn[C + c] = (float)invstd; a.bn[2 * C + c] = (float)scale; a.bn[3 * C + c] = (float)shift;
    }
    __syncthreads();
    for (int e = tid; e < C * 10; e += PREP_THREADS) {   // wl[c][t] = sum_k Wglu[c][k] wz[k][t] (+ bglu at t = 9): one thread per (c, t)
        const int c = e / 10, t = e % 10;
        double acc = (t == 9) ? (double)a.bglu[c] : 0.0;
        for (int k = 0; k < C; ++k) acc += (double)wgl[c * (C + 1) + k] * wzs[k][t];
        a.wl[c * 12 + t] = (float)acc;
    }
}

// ---- shared tile machinery ----------------------------------------------------------------------
// A workgroup (4 waves) owns 4 pooled rows x 16 pooled cols of one clip = 8 x 64 input pixels.
// Wave w owns pooled row w; it walks 4 "row blocks" g of 32 pixels = pooled cols 4g..4g+3.
// MFMA row m of a row block: j = m>>3 (pooled col 4g+j), dt = (m>>2)&1, df = m&3, so that
// D-fragment register r of lane l (row (r&3)+8(r>>2)+4(l>>5)) is pooled col r>>2, dt = l>>5, df = r&3.
// MODE 0: exact fp32 operands, K = 10 as five v_mfma_f32_32x32x2_f32 per 32 x 32 tile.  MODE 1 (sed_dims.dtype = bf16): the
// patch and the folded weights rounded to bf16, K = 10 padded to 16 = ONE v_mfma_f32_32x32x16_bf16 per tile (fp32
// accumulation): the MFMA part of the tile drops from 640 to 2 x 16 cycles; everything downstream (sigmoid, dropout,
// pooling, the backward's D / E sums) stays fp32.  Same D layout for both.
typedef __attribute__((ext_vector_type(8))) __bf16 blk0_bf16x8;
template <int NH, int MODE>      // NH = C / 32 channel slices
struct Blk0W;
template <int NH>
struct Blk0W<NH, 0> {
    float bw[5][2 * NH];   // B fragments: [k-step][col block]; col blocks 0 .. NH-1 = lin, NH .. 2NH-1 = z
};
template <int NH>
struct Blk0W<NH, 1> {
    blk0_bf16x8 bw[2 * NH];   // B fragments, k = 8 (lane >> 5) + 0..7 (taps >= 10 are zero)
};
// MODE 2 (SED_DTYPE_BF16X3): patch and folded weights split hi + lo (two bf16 each), the tile is hi hi + hi lo + lo hi:
// three K = 16 bf16 MFMAs (48 cycles) instead of the five exact-fp32 ones (640) at ~2^-16 per product; everything downstream
// as in MODE 0 (fp32 storage, the backward's D / E sums on split operands).
template <int NH>
struct Blk0W<NH, 2> {
    blk0_bf16x8 bw[2 * NH], bl[2 * NH];
};
// MODE 3 (SED_DTYPE_F16, forward only): MODE 1 with fp16 operands (v_mfma_f32_32x32x16_f16; the registers hold fp16 bit patterns)
// and the pooled output stored as fp16 + a bf16 copy for the backward (which is MODE 1's)
template <int NH>
struct Blk0W<NH, 3> {
    blk0_bf16x8 bw[2 * NH];
};
typedef __attribute__((ext_vector_type(8))) _Float16 blk0_f16x8;
__device__ __forceinline__ __bf16 blk0_h16(float v) { return __builtin_bit_cast(__bf16, (_Float16)v); }
template <int NH, int MODE>
__device__ __forceinline__ void blk0_load_w(Blk0W<NH, MODE>& W, const float* __restrict__ wz, const float* __restrict__ wl, int lane) {
    const int n = lane & 31, kh = lane >> 5;
    // the z columns carry -log2(e): the MFMA then delivers the argument of exp2 in sigmoid(z) = 1 / (1 + 2^(-log2e z))
    // directly (these kernels are VALU-bound; z itself is never needed, only sigmoid(z))
    if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int k = 2 * s + kh;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                W.bw[s][h] = wl[(32 * h + n) * 12 + k];
                W.bw[s][NH + h] = wz[(32 * h + n) * 12 + k] * SED_NEG_LOG2E;
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 8 * kh + i, kc = k < 10 ? k : 0;
                const float l = wl[(32 * h + n) * 12 + kc], z = wz[(32 * h + n) * 12 + kc] * SED_NEG_LOG2E;
                const float lv = k < 10 ? l : 0.f, zv = k < 10 ? z : 0.f;
                W.bw[h][i] = MODE == 3 ? blk0_h16(lv) : (__bf16)lv;
                W.bw[NH + h][i] = MODE == 3 ? blk0_h16(zv) : (__bf16)zv;
                if constexpr (MODE == 2) {
                    W.bl[h][i] = (__bf16)(lv - (float)W.bw[h][i]);
                    W.bl[NH + h][i] = (__bf16)(zv - (float)W.bw[NH + h][i]);
                }
            }
    }
}
// MFMA A operand of a row block: this lane's pixel (LDS offset `base` of its top-left tap)
template <int MODE> struct Blk0A;
template <> struct Blk0A<0> { float v[5]; };           // taps 2 s + kh
template <> struct Blk0A<1> { blk0_bf16x8 v; };        // taps 8 kh + 0..7 (tap 9 = the constant 1, taps >= 10 zero)
template <> struct Blk0A<2> { blk0_bf16x8 v, lo; };    // the same taps, hi and lo parts
template <> struct Blk0A<3> { blk0_bf16x8 v; };        // the same taps as fp16 bit patterns
template <int MODE>
__device__ __forceinline__ void blk0_load_a(Blk0A<MODE>& A, const float* xb, int base, int kh) {
    // Every LDS read below is UNCONDITIONAL (clamped address) and the constant / zero taps are selected afterwards: written as
    // `cond ? constant : xb[...]` each read became a divergent branch with its own s_waitcnt lgkmcnt(0) - one (fp32) or seven
    // (bf16 family) serialized LDS round trips per 32-pixel row block in every block-0 kernel (round 6).
    if constexpr (MODE == 0) {
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {      // (only k = 9 - s5 = 4 of the kh = 1 lanes - is the constant: one conditional read)
            const int k = 2 * s5 + kh;
            A.v[s5] = (k == 9) ? 1.0f : xb[base + (k / 3) * XS_W + (k % 3)];
        }
    } else {
        // kh = 0: taps 0 .. 7; kh = 1: tap 8, the constant, six zeros
        float t[8];
        t[0] = xb[base + (kh ? 2 * XS_W + 2 : 0)];
#pragma unroll
        for (int i = 1; i < 8; ++i) t[i] = xb[base + (i / 3) * XS_W + (i % 3)];
        t[1] = kh ? 1.0f : t[1];
#pragma unroll
        for (int i = 2; i < 8; ++i) t[i] = kh ? 0.f : t[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            A.v[i] = MODE == 3 ? blk0_h16(t[i]) : (__bf16)t[i];
            if constexpr (MODE == 2) A.lo[i] = (__bf16)(t[i] - (float)A.v[i]);
        }
    }
}
template <int NH, int MODE>
__device__ __forceinline__ void blk0_mma(const Blk0A<MODE>& A, const Blk0W<NH, MODE>& W, int h, f32x16& al, f32x16& az) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { al[r] = 0.f; az[r] = 0.f; }
    if constexpr (MODE == 0) {
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            al = mfma32(A.v[s5], W.bw[s5][h], al);
            az = mfma32(A.v[s5], W.bw[s5][NH + h], az);
        }
    } else if constexpr (MODE == 3) {
        al = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(blk0_f16x8, A.v), __builtin_bit_cast(blk0_f16x8, W.bw[h]), al, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(blk0_f16x8, A.v), __builtin_bit_cast(blk0_f16x8, W.bw[NH + h]), az, 0, 0, 0);
    } else {
        al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, W.bw[h], al, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, W.bw[NH + h], az, 0, 0, 0);
        if constexpr (MODE == 2) {
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, W.bl[h], al, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, W.bl[NH + h], az, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.lo, W.bw[h], al, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.lo, W.bw[NH + h], az, 0, 0, 0);
        }
    }
}
// the lin half only (k_blk0_bwd<..., SG = 1>: the gate comes from the forward's saved bytes); bf16 family
template <int NH, int MODE>
__device__ __forceinline__ void blk0_mma_lin(const Blk0A<MODE>& A, const Blk0W<NH, MODE>& W, int h, f32x16& al) {
    static_assert(MODE == 1, "saved gates: SED_DTYPE_BF16 / F16 backward only");
#pragma unroll
    for (int r = 0; r < 16; ++r) al[r] = 0.f;
    al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, W.bw[h], al, 0, 0, 0);
}
__device__ __forceinline__ void blk0_load_xs(float* xs, const float* __restrict__ x, int b, int T, int t0, int tid) {
    // XS_H rows of 64 contiguous floats: one float4 per thread (160 of the 256), zero rows outside the clip
    if (tid < XS_H * 16) {
        const int r = tid >> 4, c4 = tid & 15, t = t0 - 1 + r;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *(const float4*)&x[((size_t)b * T + t) * 64 + 4 * c4];
        float* d = &xs[r * XS_W + 1 + 4 * c4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if (tid < XS_H * 16 + 2 * XS_H) {
        const int e = tid - XS_H * 16;
        xs[(e >> 1) * XS_W + (e & 1) * 65] = 0.f;
    }
}
// ---- forward --------------------------------------------------------------------------------
// Per 32-pixel x 32-channel tile: 10 v_mfma_f32_32x32x2_f32 (lin and z, K = 10: 640 MFMA cycles) and then ~110 VALU
// instructions on their 2 x 16 results (exp2, rcp, keep bit, pooled FMA).  The two pipes are about equally loaded
// (rocprof round 2: VALU-pipe time 17.6 us, MFMA-pipe time 14.6 us of 48 us solo) but inside ONE wave they were strictly
// serial - the epilogue consumes what the MFMAs just produced - so the overlap was left to chance between waves.  Now the
// wave is software-pipelined: the MFMAs of tile t+1 are issued BEFORE the epilogue of tile t (two accumulator sets), the
// input tile is double-buffered in LDS (next tile's global loads in flight during the compute, one barrier per tile), and
// a workgroup tile is 2 pooled rows (two waves per row, two row blocks each: 3 768 tiles at the baseline shape, five full
// rounds of the persistent grid - 4-row tiles were 2.5 rounds, a 16 % idle tail), and the dropout mode is a template parameter (the run-time flags split the loop body into a dozen basic blocks, which kept
// the scheduler from interleaving anything).  DROP: 0 none, 1 one keep bit per element (p = 0.5), 2 byte threshold.
__device__ __forceinline__ float blk0_half_sum(float x) {      // x + (the other half-wave's x), in every lane
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// SG (bf16 family, round 6): the forward that WILL be differentiated also stores the GLU gate sigmoid(z) of every element as one
// byte (round(255 s)), 16 bytes per lane and (row block, channel slice) beside the 16 keep bits - k_blk0_bwd<..., SG = 1> then
// reads the gate instead of recomputing z on the MFMA and taking exp2 + rcp per element (8 of its 15 VALU issue slots per element;
// it is the tail of every bf16-family step).  1 byte per element of B x T x 64 x C: written and read once, by VALU-bound kernels
// whose memory pipes idle.  |error| <= 1 / 510 per gate, unbiased, entering only the block's parameter-gradient SUMS over
// ~10^6 pixels (block 0 has no data gradient).
template <int NH, int DROP, bool SAVE, int MODE, int SG = 0>
__global__ __launch_bounds__(256, (NH == 2 ? 3 : 2)) void k_blk0_fwd(const float* __restrict__ x, const float* __restrict__ wz,
                                                   const float* __restrict__ wl, float* __restrict__ p0, int B, int T,
                                                   int H1, int tiles_per_clip, int n_tiles, float p_drop,
                                                   const uint64_t* __restrict__ seed_ptr, uint16_t* __restrict__ mask_out,
                                                   void* __restrict__ p0_b16 /* MODE 3: bf16 copy of the output, may be null */,
                                                   uint4* __restrict__ sg_out = nullptr /* SG: [unit][lane] 16 gate bytes */) {
    __shared__ float xs[2][FXS_H * XS_W];
    constexpr int C = 32 * NH;
    constexpr int NT = 2 * NH;                       // MFMA tiles of a wave: (row block g0 + {0, 1}, channel slice h)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    TS(0);
    Blk0W<NH, MODE> W;
    blk0_load_w<NH, MODE>(W, wz, wl, lane);
    const uint64_t seed = DROP ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float sc = 0.125f * (DROP ? drop_scale8(p_drop) : 1.0f);
    // input tile: FXS_H rows of 64 contiguous floats, one float4 per thread (96 of the 256); zero rows outside the clip
    const int xr = tid >> 4, xc4 = tid & 15;
    auto xs_fetch = [&](int tile) -> float4 {
        const int b = tile / tiles_per_clip, t = 2 * ((tile % tiles_per_clip) * 2) - 1 + xr;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (tid < FXS_H * 16 && t >= 0 && t < T) v = *(const float4*)&x[((size_t)b * T + t) * 64 + 4 * xc4];
        return v;
    };
    auto xs_put = [&](float* dst, const float4& v) {
        if (tid < FXS_H * 16) {
            float* d = &dst[xr * XS_W + 1 + 4 * xc4];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    };
    if (tid < 2 * 2 * FXS_H) xs[tid / (2 * FXS_H)][((tid % (2 * FXS_H)) >> 1) * XS_W + (tid & 1) * 65] = 0.f;   // left / right halo columns
    int tile = blockIdx.x;
    if (tile < n_tiles) xs_put(xs[0], xs_fetch(tile));
    __syncthreads();
#ifdef SED_TS2
    TSC(1);
#else
    TS(1);
#endif
    int ts_k = 2;
    const int mj = n >> 3, mdt = (n >> 2) & 1, mdf = n & 3;     // this lane's pixel m = n of a row block
    for (int buf = 0; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        const int nxt = tile + gridDim.x;
        float4 vn = {0.f, 0.f, 0.f, 0.f};
        if (nxt < n_tiles) vn = xs_fetch(nxt);
        const int b = tile / tiles_per_clip, to = (tile % tiles_per_clip) * 2 + (wv >> 1), g0 = 2 * (wv & 1);
        if (to < H1) {
            const float* xb = xs[buf];
            // p = 0.5: one Philox draw carries the 16-bit keep fields of 8 consecutive (row block, channel slice) units
            // u = rb * NH + h (philox.h / gen.h): the 4 row blocks of this pooled row are units [4 NH rb0, 4 NH rb0 + 4 NH),
            // i.e. NH / 2 whole draws, made here once (C = 64: one draw, exactly philox_stream_1bit of the first version)
            // (this wave's two row blocks g0, g0 + 1 sit in ONE of them: draw d = 0 at C = 64, d = g0 >> 1 at C = 128)
            static_assert(NH == 2 || NH == 4, "draw selection below assumes 8 or 16 units per pooled row");
            const int dsel = (NH == 2) ? 0 : (g0 >> 1);
            u32x4 o1 = {0u, 0u, 0u, 0u};
            if (DROP == 1) o1 = philox_stream(((uint32_t)(b * H1 + to) * (NH / 2) + (uint32_t)dsel) * 64u + (uint32_t)lane, PHILOX_STREAM_1BIT, seed);
            Blk0A<MODE> av;
            // MFMA A operand of row block g: this lane's pixel (shared by the channel slices)
            auto load_av = [&](int g) { blk0_load_a<MODE>(av, xb, (2 * (wv >> 1) + mdt) * XS_W + 16 * g + 4 * mj + mdf, kh); };
            f32x16 al[2], az[2];
            auto mma = [&](int h, int sl) {
#ifdef BLK0_EXP_NOMMA       // timing experiments only (tools/build_variant.sh): results are garbage
#pragma unroll
                for (int r = 0; r < 16; ++r) { al[sl][r] = (float)r * xb[r]; az[sl][r] = (float)(r + h) * xb[r + 1]; }
#else
                blk0_mma<NH, MODE>(av, W, h, al[sl], az[sl]);
#endif
            };
            auto epilogue = [&](int g, int h, const f32x16& l16, const f32x16& z16) {
                const int c = 32 * h + n;
                const int q0 = (b * H1 + to) * 16 + 4 * g;
                float pooled[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef BLK0_EXP_NOEPI
#pragma unroll
                for (int r = 0; r < 16; r += 4) pooled[r >> 2] = l16[r] + z16[r] + l16[r + 1] + z16[r + 1] + l16[r + 2] + z16[r + 2] + l16[r + 3] + z16[r + 3];
#else
                if (DROP) {
                    uint32_t m16;
                    if (DROP == 1) {
                        // local unit g * NH + h: draw (g * NH + h) >> 3, field (g * NH + h) & 7
                        m16 = philox_field16(o1, (NH == 2) ? 2 * g + h : 4 * (g & 1) + h);
                    } else {
                        // one Philox draw = 16 bytes = this lane's 16 elements (4 pooled pixels x 4 df) of channel c
                        const u32x4 o = philox_stream((uint32_t)((q0 >> 2) * C + c), (uint32_t)kh, seed);
                        m16 = philox_keep16(o, thr);
                    }
                    if (SAVE) mask_out[((size_t)(q0 >> 2) * NH + h) * 64 + lane] = (uint16_t)m16;
                    uint32_t sgw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // keep bit r as an all-ones / all-zeros word (v_bfe_i32) ANDed onto lin: two plain VALU instructions
                        const int keep = __builtin_amdgcn_sbfe((int)m16, r, 1);
                        const float lm = __int_as_float(__float_as_int(l16[r]) & keep);
                        const float sgv = sigmoid_from_scaled(z16[r]);
                        pooled[r >> 2] = fmaf(lm, sgv, pooled[r >> 2]);
                        if constexpr (SG != 0) sgw[r >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(sgv * 255.0f, r & 3, sgw[r >> 2]);
                    }
                    if constexpr (SG != 0) sg_out[((size_t)(q0 >> 2) * NH + h) * 64 + lane] = uint4{sgw[0], sgw[1], sgw[2], sgw[3]};
                } else {
                    uint32_t sgw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sgv = sigmoid_from_scaled(z16[r]);
                        pooled[r >> 2] = fmaf(l16[r], sgv, pooled[r >> 2]);
                        if constexpr (SG != 0) sgw[r >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(sgv * 255.0f, r & 3, sgw[r >> 2]);
                    }
                    if constexpr (SG != 0) sg_out[((size_t)(q0 >> 2) * NH + h) * 64 + lane] = uint4{sgw[0], sgw[1], sgw[2], sgw[3]};
                }
#endif
                // the two half-waves hold dt = 0 / 1 of the same pooled pixels
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) pooled[jx] = blk0_half_sum(pooled[jx]);
                const int j0 = 2 * kh;
#ifdef BLK0_EXP_NOST
                if (pooled[0] == 12345.678f)
#endif
                {
                    // (MODE 1 = SED_DTYPE_BF16: the pooled output is stored as bf16, gen.h)
                    using PT = typename std::conditional<MODE == 3, _Float16, typename Stor<MODE == 1>::T>::type;
                    st1((PT*)p0 + (size_t)(q0 + j0) * C + c, (kh ? pooled[2] : pooled[0]) * sc);
                    st1((PT*)p0 + (size_t)(q0 + j0 + 1) * C + c, (kh ? pooled[3] : pooled[1]) * sc);
                    if constexpr (MODE == 3) {
                        if (p0_b16) {
                            st1((__bf16*)p0_b16 + (size_t)(q0 + j0) * C + c, (kh ? pooled[2] : pooled[0]) * sc);
                            st1((__bf16*)p0_b16 + (size_t)(q0 + j0 + 1) * C + c, (kh ? pooled[3] : pooled[1]) * sc);
                        }
                    }
                }
            };
            load_av(g0);
            mma(0, 0);
#ifdef SED_TS2
            const bool first = ts_k == 2;
            if (first) TSC(2);
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + 1 < NT) {
                    if ((t + 1) % NH == 0) load_av(g0 + (t + 1) / NH);
                    mma((t + 1) % NH, (t + 1) & 1);
                }
#ifdef SED_TS2
                if (first) { __builtin_amdgcn_sched_barrier(0); TSC(3 + 2 * t); __builtin_amdgcn_sched_barrier(0); }
#endif
                epilogue(g0 + t / NH, t % NH, al[t & 1], az[t & 1]);
#ifdef SED_TS2
                if (first) { __builtin_amdgcn_sched_barrier(0); TSC(4 + 2 * t); __builtin_amdgcn_sched_barrier(0); }
#endif
            }
        }
#ifndef SED_TS2
        if (ts_k < 12) { TS(ts_k); ++ts_k; }
#endif
        if (nxt < n_tiles) xs_put(xs[buf ^ 1], vn);
        __syncthreads();
#ifndef SED_TS2
        if (ts_k < 12) { TS(ts_k); ++ts_k; }
#else
        ts_k = 3;
#endif
    }
    TS(13);
}

// ---- backward: D[co][t] = sum_p dlin[p][co] P[p][t],  E[c][t] = sum_p dzgate[p][c] P[p][t] ------
// NHT: channel slices of the whole block; a workgroup handles NH of them, those from blockIdx.y * NH on (C = 128 runs as two
// 64-channel halves: the accumulators of all four slices need 290 registers = ONE wave per SIMD with every LDS / MFMA
// latency of its in-order stream exposed - 197 us against 2 x 53 for the halves at two waves per SIMD)
// STRICT (fp32 only, debug bit 27 / SED_STRICT_F32=1): the 2 x 10 sums per channel as plain fp32 FMAs on the VALU instead of
// split-bf16 MFMA products - the all-fp32 twin of the `dtype: f32` headline (bench.py extra_configs["mt-f32-strict"]).
#ifndef BLK0_BWD_OCC16
#define BLK0_BWD_OCC16 2
#endif
#ifndef BLK0_BWD_GUNROLL
#define BLK0_BWD_GUNROLL 4      // (row-block loop fully unrolled: 97 -> 90 us solo at B = 64 in the bf16 family; 1: 103)
#endif
template <int NH, int MODE, int NHT, int STRICT = 0, int SG = 0>
__global__ __launch_bounds__(256, (NH == 2 ? (MODE == 1 ? BLK0_BWD_OCC16 : 2) : 1)) void k_blk0_bwd(const float* __restrict__ x, const float* __restrict__ wz,
                                                   const float* __restrict__ wl, const float* __restrict__ dp0, int B,
                                                   int T, int H1, int tiles_per_clip, int n_tiles, int use_drop,
                                                   float p_drop, const uint16_t* __restrict__ mask_in,
                                                   double* __restrict__ de /* [2][C][10] */, int no_atomic,
                                                   const uint4* __restrict__ sg_in = nullptr /* SG: the forward's gate bytes */) {
    __shared__ float xs[XS_H * XS_W];
    __shared__ __attribute__((aligned(16))) float P[4][32 * 12];
    __shared__ float red[4][2][NH][32][10];
    constexpr int C = 32 * NHT, CS = 32 * NH;
    const int h0 = blockIdx.y * NH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    Blk0W<NH, MODE> W;
    blk0_load_w<NH, MODE>(W, wz + 32 * h0 * 12, wl + 32 * h0 * 12, lane);
    const float keep_scale = use_drop ? drop_scale8(p_drop) : 1.0f;
    // The first version formed the 2 x 10 sums per channel on the VALU (fp32 FMAs, lane = channel).  The sums are the GEMMs
    // D = P^T dlin, E = P^T dzgate contracted over pixels - on the MFMA pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulation over
    // the whole grid-stride loop): the forward tile's D layout (lane = channel column, register r = pixel row) IS the B
    // fragment of the transposed product, the A fragment is P^T (lane = tap row, 10 of 32 rows used) read straight from the
    // input tile.  20 FMAs per (pixel, channel) - 2/3 of this kernel's VALU work - become 4 MFMAs per 32 x 32 tile.
    constexpr bool VALU_SUMS = STRICT != 0;      // (the round-1/2 path: 2 x 10 fp32 FMAs per pixel and channel on the VALU)
    float aD[VALU_SUMS ? NH : 1][10], aE[VALU_SUMS ? NH : 1][10];
    f32x16 accD[VALU_SUMS ? 1 : NH], accE[VALU_SUMS ? 1 : NH];
    if constexpr (VALU_SUMS) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int t = 0; t < 10; ++t) { aD[h][t] = 0.f; aE[h][t] = 0.f; }
    } else {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accD[h][r] = 0.f; accE[h][r] = 0.f; }
    }
    float* Pw = P[wv];
    // the next tile's rows are fetched (one float4 per thread, clamped row + select: no branch) before this tile's compute
    // and land in LDS after it - the load used to sit between two barriers, a full HBM round trip per tile in the open
    auto xs_fetch = [&](int tile) -> float4 {
        const int b = tile / tiles_per_clip, t0 = 2 * ((tile % tiles_per_clip) * 4);
        const int r = tid >> 4, c4 = tid & 15, t = t0 - 1 + r, tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (tid < XS_H * 16) {                                            // (wave-uniform up to the last wave)
            v = *(const float4*)&x[((size_t)b * T + tc) * 64 + 4 * c4];
            if (t < 0 || t >= T) v = float4{0.f, 0.f, 0.f, 0.f};
        }
        return v;
    };
    auto xs_put = [&](const float4& v) {
        if (tid < XS_H * 16) {
            float* d = &xs[(tid >> 4) * XS_W + 1 + 4 * (tid & 15)];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else if (tid < XS_H * 16 + 2 * XS_H) {
            const int e = tid - XS_H * 16;
            xs[(e >> 1) * XS_W + (e & 1) * 65] = 0.f;
        }
    };
    float4 xnext = {0.f, 0.f, 0.f, 0.f};
    if ((int)blockIdx.x < n_tiles) xnext = xs_fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip, to0 = (tile % tiles_per_clip) * 4;
        __syncthreads();
        xs_put(xnext);
        __syncthreads();
        {
            const int nxt = tile + (int)gridDim.x;
            if (nxt < n_tiles) xnext = xs_fetch(nxt);
        }
        const int to = to0 + wv;
        if (to >= H1) continue;
        // pooled gradients + keep bits of a row block are fetched one row block AHEAD: this kernel runs one
        // wave per SIMD (256 VGPRs), so a load issued right before its use stalls the whole SIMD for a full
        // HBM/L2 round trip (~1 us per row block in the first version)
        float gq_n[NH][4];
        uint32_t m_n[NH];
        uint4 sg_n[SG ? NH : 1];
        auto fetch = [&](int g) {
            const int q0 = (b * H1 + to) * 16 + 4 * g;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int c = 32 * (h0 + h) + n;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) gq_n[h][jx] = ld1((const typename Stor<MODE == 1>::T*)dp0 + (size_t)(q0 + jx) * C + c);
                m_n[h] = use_drop ? (uint32_t)mask_in[((size_t)(q0 >> 2) * NHT + h0 + h) * 64 + lane] : 0xffffu;
                if constexpr (SG != 0) sg_n[h] = sg_in[((size_t)(q0 >> 2) * NHT + h0 + h) * 64 + lane];
            }
        };
        fetch(0);
        constexpr int GUNROLL = STRICT != 0 ? 1 : BLK0_BWD_GUNROLL;      // (the strict-fp32 form spills when unrolled)
#pragma unroll GUNROLL
        for (int g = 0; g < 4; ++g) {
            float gq_c[NH][4];
            uint32_t m_c[NH];
            uint4 sg_c[SG ? NH : 1];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                m_c[h] = m_n[h];
                if constexpr (SG != 0) sg_c[h] = sg_n[h];
                // (SG: the gate arrives as a byte u = round(255 s): the 1 / 255 rides in the pooled gradient's scale)
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) gq_c[h][jx] = gq_n[h][jx] * (0.125f * keep_scale * (SG ? (1.0f / 255.0f) : 1.0f));
            }
            if (g < 3) fetch(g + 1);
            // patch values: MFMA A operand (this lane's pixel m, taps 2s+kh) and the im2col rows for the reductions
            Blk0A<MODE> av;
            {
                const int m = n, j = m >> 3, dt = (m >> 2) & 1, df = m & 3;
                const int base = (2 * wv + dt) * XS_W + 16 * g + 4 * j + df;
                blk0_load_a<MODE>(av, xs, base, kh);
                if constexpr (VALU_SUMS) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int k = kh * 6 + i;
                        Pw[m * 12 + k] = (k < 9) ? xs[base + (k / 3) * XS_W + (k % 3)] : (k == 9 ? 1.0f : 0.f);
                    }
                }
            }
            if constexpr (!VALU_SUMS) {
                // A fragments of the pixel contraction: lane (tap = n, half kh) holds P[pixel(r, kh)][tap] for its 16 pixel rows
                // r - pixel(r, kh) is pooled col r >> 2, dt = kh, df = r & 3, i.e. 16 CONSECUTIVE floats of input row
                // 2 wv + kh + tap / 3 (tap 9 = the constant 1, taps >= 10 zero: clamped address, then select).
                // MODE 0 (fp32 arithmetic): every operand split v = hi + lo into two bf16 (lo = bf16(v - hi)), three products
                // hi hi + hi lo + lo hi - the dropped lo lo term is 2^-16 of a product, below the fp32 rounding of the sums.
                constexpr int NP = MODE != 1 ? 2 : 1;
                blk0_bf16x8 pT[NP][2];
                {
                    const int tc = n < 9 ? n : 0;
                    const float* src = xs + (2 * wv + kh + tc / 3) * XS_W + 16 * g + tc % 3;
                    const float other = n == 9 ? 1.0f : 0.f;
                    // (all sixteen reads unconditional - the address is clamped - and selected afterwards: written as
                    // `n < 9 ? src[r] : other` every read became a branch with its own s_waitcnt, sixteen serialized LDS round
                    // trips per row block)
                    float pv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) pv[r] = src[r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const __bf16 hi = (__bf16)pv[r];
                        pT[0][r >> 3][r & 7] = hi;
                        if constexpr (MODE != 1) pT[1][r >> 3][r & 7] = (__bf16)(pv[r] - (float)hi);
                    }
                    // the constant / zero taps (lanes n >= 9) are selected on the PACKED pairs: 8 selects per plane instead of 16
                    typedef __attribute__((ext_vector_type(4))) unsigned int blk0_u32x4;
                    const unsigned int oth = __builtin_bit_cast(unsigned short, (__bf16)other) * 0x10001u;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            blk0_u32x4 w = __builtin_bit_cast(blk0_u32x4, pT[pl][hf]);
#pragma unroll
                            for (int q = 0; q < 4; ++q) w[q] = n < 9 ? w[q] : (pl ? 0u : oth);
                            pT[pl][hf] = __builtin_bit_cast(blk0_bf16x8, w);
                        }
                }
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    f32x16 al, az;
                    if constexpr (SG != 0) blk0_mma_lin<NH, MODE>(av, W, h, al);      // (z is not needed: its gate was saved)
                    else blk0_mma<NH, MODE>(av, W, h, al, az);
                    blk0_bf16x8 fl[NP][2], fz[NP][2];
                    const uint32_t sgw[4] = {SG ? sg_c[SG ? h : 0].x : 0u, SG ? sg_c[SG ? h : 0].y : 0u, SG ? sg_c[SG ? h : 0].z : 0u,
                                             SG ? sg_c[SG ? h : 0].w : 0u};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // keep bit r as an all-ones / all-zeros word (v_bfe_i32) ANDed onto the pooled gradient: two plain VALU
                        // instructions, as in the forward (`bit ? g : 0` compiled to and + compare + select)
                        // (the v_bfe_i32 is assembly: LLVM folds `g & sext(bit)` back into the three-instruction select)
                        int keepw;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keepw) : "v"(m_c[h]), "n"(r));
                        const float gg = __int_as_float(__float_as_int(gq_c[h][r >> 2]) & keepw);
                        float dl, dzg;
                        if constexpr (SG != 0) {
                            // u = 255 s as a float (v_cvt_f32_ubyte<r & 3>); gg carries the 1 / 255: dl = gg' u, 1 - s = 1 - u / 255
                            const float u = (float)((sgw[r >> 2] >> (8 * (r & 3))) & 0xffu);
                            dl = gg * u;
                            dzg = dl * al[r] * fmaf(u, -1.0f / 255.0f, 1.0f);
                        } else {
                            const float sg = sigmoid_from_scaled(az[r]);
                            dl = gg * sg;
                            dzg = dl * al[r] * (1.0f - sg);
                        }
                        const __bf16 lh = (__bf16)dl, zh = (__bf16)dzg;
                        fl[0][r >> 3][r & 7] = lh;
                        fz[0][r >> 3][r & 7] = zh;
                        if constexpr (MODE != 1) {
                            fl[1][r >> 3][r & 7] = (__bf16)(dl - (float)lh);
                            fz[1][r >> 3][r & 7] = (__bf16)(dzg - (float)zh);
                        }
                    }
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        accD[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pT[0][hf], fl[0][hf], accD[h], 0, 0, 0);
                        accE[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pT[0][hf], fz[0][hf], accE[h], 0, 0, 0);
                        if constexpr (MODE != 1) {
                            accD[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pT[0][hf], fl[1][hf], accD[h], 0, 0, 0);
                            accE[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pT[0][hf], fz[1][hf], accE[h], 0, 0, 0);
                            accD[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pT[1][hf], fl[0][hf], accD[h], 0, 0, 0);
                            accE[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pT[1][hf], fz[0][hf], accE[h], 0, 0, 0);
                        }
                    }
                }
            } else {
            __builtin_amdgcn_wave_barrier();
            // the two 32-channel halves one after the other: half the live accumulators / gradients, so the
            // kernel fits 2 waves per SIMD (the all-at-once version needed 331 registers = 1 wave, and every
            // LDS / MFMA latency of its in-order instruction stream was exposed)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                f32x16 al, az;
                blk0_mma<NH, MODE>(av, W, h, al, az);
                // one pass over the lane's 16 pixel rows: dlin / dz_gate of row r are formed on the fly (no 2 x 16-register
                // arrays live across the loop) and the NEXT row's patch (3 ds_read_b128) is fetched before this row's 20
                // accumulation FMAs - with the reads issued right in front of their FMAs the wave sat out one LDS latency per
                // row (SQ_WAIT_ANY 23 % of the wave cycles, profiles/r02_b)
                f32x4 pa = *(const f32x4*)&Pw[mfma32_row(0, lane) * 12 + 0];
                f32x4 pb = *(const f32x4*)&Pw[mfma32_row(0, lane) * 12 + 4];
                f32x4 pc = *(const f32x4*)&Pw[mfma32_row(0, lane) * 12 + 8];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv[10] = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3], pc[0], pc[1]};
                    if (r + 1 < 16) {
                        const int i1 = mfma32_row(r + 1, lane);
                        pa = *(const f32x4*)&Pw[i1 * 12 + 0];
                        pb = *(const f32x4*)&Pw[i1 * 12 + 4];
                        pc = *(const f32x4*)&Pw[i1 * 12 + 8];
                    }
                    const float gg = ((m_c[h] >> r) & 1u) ? gq_c[h][r >> 2] : 0.f;
                    const float sg = sigmoid_from_scaled(az[r]);
                    const float dl = gg * sg;
                    const float dzg = dl * al[r] * (1.0f - sg);
#pragma unroll
                    for (int t = 0; t < 10; ++t) {
                        aD[h][t] = fmaf(dl, pv[t], aD[h][t]);
                        aE[h][t] = fmaf(dzg, pv[t], aE[h][t]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // reduce: half-waves share the channel; then waves; then fp64 atomics
    if constexpr (VALU_SUMS) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                float vD = aD[h][t] + __shfl_xor(aD[h][t], 32);
                float vE = aE[h][t] + __shfl_xor(aE[h][t], 32);
                if (kh == 0) { red[wv][0][h][n][t] = vD; red[wv][1][h][n][t] = vE; }
            }
    } else {
        // accumulator (tap row (r & 3) + 8 (r >> 2) + 4 kh, channel column n): taps 0 .. 7 are r < 4 of both halves, taps 8, 9 are
        // r = 4, 5 of half 0 (the contraction already ran over both halves' pixels: no cross-lane sum)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { red[wv][0][h][n][r + 4 * kh] = accD[h][r]; red[wv][1][h][n][r + 4 * kh] = accE[h][r]; }
            if (kh == 0) {
                red[wv][0][h][n][8] = accD[h][4]; red[wv][0][h][n][9] = accD[h][5];
                red[wv][1][h][n][8] = accE[h][4]; red[wv][1][h][n][9] = accE[h][5];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * CS * 10; i += 256) {
        const int which = i / (CS * 10), c = (i % (CS * 10)) / 10, t = i % 10;
        const int h = c >> 5, nn = c & 31;
        double v = (double)red[0][which][h][nn][t] + (double)red[1][which][h][nn][t] + (double)red[2][which][h][nn][t] +
                   (double)red[3][which][h][nn][t];
        if (!no_atomic) atomicAdd(&de[(which * C + 32 * h0 + c) * 10 + t], v);
    }
}

// ---- backward finalize (fp64 algebra, one workgroup) ---------------------------------------------
struct Blk0BwdFinArgs {
    const float *w0, *b0, *gamma, *beta, *wglu;
    const float* bn;       // mean, invstd, scale, shift
    const double* mom;     // s, G
    const double* de;      // D[64][10], E[64][10]
    double N;
    float *g_w0, *g_b0, *g_gamma, *g_beta, *g_wglu, *g_bglu;
    int C;
};
// grid = C / 16 workgroups; workgroup w owns channels [16 w, 16 w + 16): their rows of dWglu, their rows of S and their
// per-channel results (one workgroup doing all of it took 104 us at C = 128, at the very end of the step)
__global__ __launch_bounds__(640) void k_blk0_bwd_finalize(Blk0BwdFinArgs a) {
    __shared__ double wzs[128][10];
    __shared__ double Ds[128][10];
    __shared__ double Ss[16][10];
    __shared__ double moms[54];
    __shared__ float Wc[128][16];               // W_glu[co][c0 .. c0 + 16): this workgroup's columns of the S sum below
    const int tid = threadIdx.x, C = a.C, c0 = blockIdx.x * 16;
    // (requested up front, 64 contiguous bytes per row: as `a.wglu[co * C + c]` inside the S loop every one of its 16 - 32 trips was a
    // strided global load behind an fp64 accumulation - 20 us solo at C = 128, at the very end of the wide step)
    for (int e = tid; e < C * 16; e += 640) Wc[e >> 4][e & 15] = a.wglu[(size_t)(e >> 4) * C + c0 + (e & 15)];
    for (int e = tid; e < C * 10; e += 640) {   // (c, t)
        const int c = e / 10, t = e % 10;
        const double scale = a.bn[2 * C + c], shift = a.bn[3 * C + c];
        wzs[c][t] = (t < 9) ? scale * (double)a.w0[c * 9 + t] : scale * (double)a.b0[c] + shift;
        Ds[c][t] = a.de[c * 10 + t];
    }
    if (tid < 54) moms[tid] = a.mom[tid];
    __syncthreads();
    // GLU linear: dWglu[co][k] = sum_t D[co][t] wz[k][t] for co in this workgroup's 16 rows;  dbglu[co] = D[co][9]
    for (int e = tid; e < 16 * C; e += 640) {
        const int co = c0 + e / C, k = e % C;
        double acc = 0;
#pragma unroll
        for (int t = 0; t < 10; ++t) acc += Ds[co][t] * wzs[k][t];
        a.g_wglu[(size_t)co * C + k] = (float)acc;
    }
    // total dz against the patch: S[c][t] = sum_co Wglu[co][c] D[co][t] + E[c][t], 4 threads per (c, t) over co quarters
    {
        const int e = tid >> 2, part = tid & 3;        // e < 160
        const int c = c0 + e / 10, t = e % 10;
        double acc = 0;
        for (int co = part; co < C; co += 4) acc += (double)Wc[co][e / 10] * Ds[co][t];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) Ss[e / 10][t] = acc + a.de[C * 10 + c * 10 + t];
    }
    __syncthreads();
    if (tid < 16) {
        const int cl = tid, c = c0 + cl;
        const double mean = a.bn[c], invstd = a.bn[C + c], scale = a.bn[2 * C + c];
        double w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = a.w0[c * 9 + t];
        const double b = a.b0[c];
        a.g_bglu[c] = (float)Ds[c][9];
        const double Sdz = Ss[cl][9];
        double Sdzu = b * Sdz;
#pragma unroll
        for (int t = 0; t < 9; ++t) Sdzu += w[t] * Ss[cl][t];
        const double Sdzxhat = invstd * (Sdzu - mean * Sdz);
        a.g_beta[c] = (float)Sdz;
        a.g_gamma[c] = (float)Sdzxhat;
        const double m1 = Sdz / a.N, m2 = Sdzxhat / a.N;
        // du = scale * (dz - m1 - xhat * m2);  dW0[c][t] = sum_p du P[t]
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            double xhP = (b - mean) * moms[t];
#pragma unroll
            for (int t2 = 0; t2 < 9; ++t2) xhP += w[t2] * moms[t2 <= t ? gidx(t2, t) : gidx(t, t2)];
            xhP *= invstd;
            a.g_w0[c * 9 + t] = (float)(scale * (Ss[cl][t] - m1 * moms[t] - m2 * xhP));
        }
        // sum_p du = scale * (Sdz - N m1 - m2 * sum xhat) = 0: a conv bias in front of a train-mode BN
        a.g_b0[c] = 0.f;
    }
}

// ---- host launchers -------------------------------------------------------------------------------
int x_moments_parts(const Geo& g) { return ((g.T + MOM_ROWS - 1) / MOM_ROWS) * g.B; }
int launch_x_moments(const Geo& g, const float* x, double* mompart, const ConvPackArgs* pack, hipStream_t st, const GenAuxPack* aux) {
    const int nx = (g.T + MOM_ROWS - 1) / MOM_ROWS;
    if (aux) {
        const int n_aux = gen_aux_blocks(*aux);
        dim3 grid(nx + (n_aux + g.B - 1) / g.B, g.B);
        k_x_moments_aux<<<grid, 256, 0, st>>>(x, g.T, mompart, nx, *aux, n_aux);
        SED_CHECK_LAUNCH();
        return SED_OK;
    }
    ConvPackArgs pa = {};
    if (pack) pa = *pack;
    dim3 grid(nx + (pack ? (SED_PACK_BLOCKS + g.B - 1) / g.B : 0), g.B);
    k_x_moments<<<grid, 256, 0, st>>>(x, g.T, mompart, nx, pa);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
// workgroups of `threads` threads of kernel `fn` that are resident on the device at once (occupancy x CUs), cached per kernel:
// the first call of every instantiation happens in the eager warm-up steps, never inside a stream capture
#include <map>
#include <mutex>
static int resident_workgroups(const void* fn, int threads) {
    static std::mutex mu;
    static std::map<const void*, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(fn);
    if (it != cache.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (g_sed_debug & 8192) fprintf(stderr, "[sed] occupancy: %d workgroups per CU, %d CUs\n", per_cu, cus);
    return cache[fn] = per_cu * cus;
}
int launch_blk0_forward(const Geo& g, const float* x, const float* w0, const float* b0, const float* gamma,
                        const float* beta, const float* wglu, const float* bglu, float* run_mean, float* run_var,
                        int64_t* tracked, int train, int update, const uint64_t* seed, double* mom, double* mompart,
                        float* wz, float* wl, float* bn, float* p0, uint16_t* mask_out, const ConvPackArgs* pack, hipStream_t st,
                        int main_kernel_only, const GenAuxPack* aux, void* p0_b16, int mom_ready, void* sg_out) {
    // main_kernel_only (sed_kernel_replay): the folded weights of the last real forward are still in wz / wl
    // mom_ready: mompart already holds this batch's patch moments (sed_crnn_moments) - no moments launch; the packing that would
    // have ridden in it rides in the prep launch
    if (train && !main_kernel_only && !mom_ready) {
        const int rc = launch_x_moments(g, x, mompart, pack, st, aux);
        if (rc != SED_OK) return rc;
    }
    Blk0PrepArgs a;
    a.w0 = w0; a.b0 = b0; a.gamma = gamma; a.beta = beta; a.wglu = wglu; a.bglu = bglu;
    a.run_mean = run_mean; a.run_var = run_var; a.tracked = tracked; a.mom = mom;
    a.mompart = mompart; a.n_part = x_moments_parts(g);
    a.N = (double)g.B * g.T * g.F; a.train = train; a.update = update; a.eps = g.eps; a.momentum = g.mom;
    a.wz = wz; a.wl = wl; a.bn = bn; a.C = g.C;
    if (!main_kernel_only) {
        if (train && mom_ready && aux) {
            const int n_aux = gen_aux_blocks(*aux);
            k_blk0_prep_aux<<<1 + n_aux, PREP_THREADS, 0, st>>>(a, *aux, n_aux);
        } else if (train && mom_ready && pack) {
            k_blk0_prep_pack<<<1 + SED_PACK_BLOCKS, PREP_THREADS, 0, st>>>(a, *pack);
        } else {
            k_blk0_prep<<<1, PREP_THREADS, 0, st>>>(a);
        }
        SED_CHECK_LAUNCH();
    }
    const int tpc = (g.H1 + 1) / 2, nt = tpc * g.B;
    const bool use_drop = train && g.p > 0.f;
    const int drop = !use_drop ? 0 : (drop_thresh8(g.p) == 128u ? 1 : 2);
    const bool save = use_drop && mask_out != nullptr;
    // persistent grid: R full rounds of the workgroups that are resident at once (asked of the runtime per instantiation)
    auto grid_for = [&](const void* fn) {
        const int slots = resident_workgroups(fn, 256), rounds = (nt + slots - 1) / slots;
        const int per_cu = 0, cus = slots;
        if (g_sed_debug & 8192) fprintf(stderr, "[sed] blk0 forward: %d resident workgroups, %d tiles, %d rounds\n", per_cu + cus, nt, rounds);
        return (nt + rounds - 1) / rounds;
    };
#define BLK0_FWD_M(NH, DROP, SAVE, MODE) \
    k_blk0_fwd<NH, DROP, SAVE, MODE><<<grid_for((const void*)k_blk0_fwd<NH, DROP, SAVE, MODE>), 256, 0, st>>>(x, wz, wl, p0, g.B, g.T, g.H1, tpc, nt, g.p, seed, mask_out, p0_b16)
#define BLK0_FWD_SG(NH, DROP, SAVE, MODE) \
    k_blk0_fwd<NH, DROP, SAVE, MODE, 1><<<grid_for((const void*)k_blk0_fwd<NH, DROP, SAVE, MODE, 1>), 256, 0, st>>>(x, wz, wl, p0, g.B, g.T, g.H1, tpc, nt, g.p, seed, mask_out, p0_b16, (uint4*)sg_out)
#define BLK0_FWD(NH, DROP, SAVE)                          \
    do {                                                  \
        if (g.f16 && sg_out) BLK0_FWD_SG(NH, DROP, SAVE, 3);   \
        else if (g.f16) BLK0_FWD_M(NH, DROP, SAVE, 3);    \
        else if (g.mode == 1 && sg_out) BLK0_FWD_SG(NH, DROP, SAVE, 1);   \
        else if (g.mode == 1) BLK0_FWD_M(NH, DROP, SAVE, 1);   \
        else if (g.mode == 2) BLK0_FWD_M(NH, DROP, SAVE, 2);   \
        else BLK0_FWD_M(NH, DROP, SAVE, 0);               \
    } while (0)
#define BLK0_FWD_NH(NH)                                              \
    do {                                                             \
        if (drop == 0) BLK0_FWD(NH, 0, false);                       \
        else if (drop == 1 && save) BLK0_FWD(NH, 1, true);           \
        else if (drop == 1) BLK0_FWD(NH, 1, false);                  \
        else if (save) BLK0_FWD(NH, 2, true);                        \
        else BLK0_FWD(NH, 2, false);                                 \
    } while (0)
    if (g.C == 64) BLK0_FWD_NH(2);
    else if (g.C == 128) BLK0_FWD_NH(4);
    else { sed_set_error("block 0: unsupported filter count %d", g.C); return SED_ERR_UNSUPPORTED; }
#undef BLK0_FWD_NH
#undef BLK0_FWD
#undef BLK0_FWD_M
#undef BLK0_FWD_SG
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_blk0_backward(const Geo& g, const float* x, const float* w0, const float* b0, const float* gamma,
                         const float* beta, const float* wglu, const uint16_t* mask_in, const double* mom,
                         const float* wz, const float* wl, const float* bn, const float* dp0, double* de, int zero_de,
                         float* g_w0, float* g_b0, float* g_gamma, float* g_beta, float* g_wglu, float* g_bglu,
                         hipStream_t st, const void* sg_in) {
    if (zero_de) SED_CHECK_HIP(hipMemsetAsync(de, 0, 2 * g.C * 10 * sizeof(double), st));
    const int tpc = (g.H1 + 3) / 4, nt = tpc * g.B;
    const int use_drop = (g.p > 0.f) ? 1 : 0;
#ifndef BLK0_GRID128
#define BLK0_GRID128 256
#endif
#ifdef BLK0_HALF_ONLY
#define BLK0_YGRID(y) 1
#else
#define BLK0_YGRID(y) (y)
#endif
#define BLK0_BWD(NH, MODE, NHT, GRID) \
    k_blk0_bwd<NH, MODE, NHT><<<dim3(nt < (GRID) ? nt : (GRID), BLK0_YGRID((NHT) / (NH))), 256, 0, st>>>(x, wz, wl, dp0, g.B, g.T, g.H1, tpc, nt, use_drop, g.p, mask_in, de, g_sed_debug & 1)
    if (g.C == 64 && g.mode == 1 && sg_in)
        k_blk0_bwd<2, 1, 2, 0, 1><<<dim3(nt < 512 ? nt : 512, 1), 256, 0, st>>>(x, wz, wl, dp0, g.B, g.T, g.H1, tpc, nt, use_drop, g.p, mask_in, de, g_sed_debug & 1, (const uint4*)sg_in);
    else if (g.C == 128 && g.mode == 1 && sg_in)
        k_blk0_bwd<2, 1, 4, 0, 1><<<dim3(nt < (BLK0_GRID128) ? nt : (BLK0_GRID128), BLK0_YGRID(2)), 256, 0, st>>>(x, wz, wl, dp0, g.B, g.T, g.H1, tpc, nt, use_drop, g.p, mask_in, de, g_sed_debug & 1, (const uint4*)sg_in);
    else if (g.C == 64 && g.mode == 1) BLK0_BWD(2, 1, 2, 512);
    else if (g.C == 64 && g.mode == 2) BLK0_BWD(2, 2, 2, 512);
    else if (g.C == 128 && g.mode == 2) BLK0_BWD(2, 2, 4, BLK0_GRID128);
    else if (g.C == 64 && (g_sed_debug & 134217728))      // debug bit 27: strict fp32 (no split-bf16 products anywhere in the step)
        k_blk0_bwd<2, 0, 2, 1><<<dim3(nt < 512 ? nt : 512, 1), 256, 0, st>>>(x, wz, wl, dp0, g.B, g.T, g.H1, tpc, nt, use_drop, g.p, mask_in, de, g_sed_debug & 1);
    else if (g.C == 64) BLK0_BWD(2, 0, 2, 512);
    else if (g.C == 128 && g.mode == 1) BLK0_BWD(2, 1, 4, BLK0_GRID128);
    else if (g.C == 128) BLK0_BWD(2, 0, 4, BLK0_GRID128);
#undef BLK0_BWD
    else { sed_set_error("block 0: unsupported filter count %d", g.C); return SED_ERR_UNSUPPORTED; }
    SED_CHECK_LAUNCH();
    Blk0BwdFinArgs a;
    a.w0 = w0; a.b0 = b0; a.gamma = gamma; a.beta = beta; a.wglu = wglu; a.bn = bn; a.mom = mom; a.de = de;
    a.N = (double)g.B * g.T * g.F;
    a.g_w0 = g_w0; a.g_b0 = g_b0; a.g_gamma = g_gamma; a.g_beta = g_beta; a.g_wglu = g_wglu; a.g_bglu = g_bglu; a.C = g.C;
    k_blk0_bwd_finalize<<<g.C / 16, 640, 0, st>>>(a);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
