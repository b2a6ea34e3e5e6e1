// gpack.h - per-forward packing work of the generic kernel family as device functions, so that a training forward can run
// it in spare workgroups of the k_x_moments launch (blk0.hip) instead of as launches of its own in front of block 0:
// k_gen_pack + k_gen_pack_bias (+ two k_gnt_pack_t for H != 64) were 2 - 4 tiny kernels at the head of each model's chain,
// and on the second model's chain - next to the first model's persistent one-workgroup-per-CU kernels, which leave no
// registers for anything else on any CU - they were what made the teacher's chain finish ~50 us behind the student's
// (profiles/r02_b_mt-bf16_step_timeline.txt).  Eval-mode forwards (no moments launch) keep the stand-alone kernels.
#pragma once
#include "gen.h"
#include "gkernels.h"

// MODE 0 / 1: every packed operand in fp32 / bf16.  MODE 2 (SED_DTYPE_BF16X3): the conv panels as TWO bf16 planes
// [hi | lo][n][9 C] (w = hi + lo to ~2^-17 relative; bconv.hip), the GLU operands as in MODE 0.
template <int MODE>
__device__ __forceinline__ void gen_pack_body(const GenPackArgs& a, int i) {
    using M = MM<MODE == 2 ? 0 : MODE>;
    using E = typename M::E;
    const int C = a.C, CC = C * C;
    if (i < a.n_zero) a.zero[i] = 0.0;
    if (i < 2 * 9 * CC) {
        const int layer = i / (9 * CC), e = i % (9 * CC);
        const int n = e / (9 * C), r = e % (9 * C), tap = r / C, k = r % C;
        const float* w = layer ? a.w2 : a.w1;
        const float wf = w[((size_t)n * C + k) * 9 + tap], wt = w[((size_t)k * C + n) * 9 + (8 - tap)];
        if (MODE == 2) {
            __bf16* wpk = (__bf16*)(layer ? a.wpk2 : a.wpk1);
            __bf16* wpkT = (__bf16*)(layer ? a.wpkT2 : a.wpkT1);
            const __bf16 h = (__bf16)wf;
            wpk[e] = h; wpk[9 * CC + e] = (__bf16)(wf - (float)h);
            if (wpkT) { const __bf16 ht = (__bf16)wt; wpkT[e] = ht; wpkT[9 * CC + e] = (__bf16)(wt - (float)ht); }
        } else {
            E* wpk = (E*)(layer ? a.wpk2 : a.wpk1);
            E* wpkT = (E*)(layer ? a.wpkT2 : a.wpkT1);
            if (MODE == 1 && a.f16) ((_Float16*)wpk)[e] = (_Float16)wf;      // the forward's operand type; the dgrad panel stays bf16
            else wpk[e] = M::cvt(wf);
            if (wpkT) wpkT[e] = M::cvt(wt);
        }
    }
    if (i < 2 * CC) {
        const int layer = i / CC, e = i % CC, co = e / C, c = e % C;
        const float* wg = layer ? a.glu_w2 : a.glu_w1;
        const float* gam = layer ? a.gamma2 : a.gamma1;
        E* o = (E*)(layer ? a.wg2 : a.wg1);
        E* oT = (E*)(layer ? a.wgT2 : a.wgT1);
        o[e] = M::cvt(wg[e] * gam[c]);
        if (oT) oT[(size_t)c * C + co] = M::cvt(wg[e]);
    }
}
// bg[co] = bglu[co] + sum_c Wglu[co][c] beta[c]: one wave per (layer, co) row, coalesced reads, fp64 butterfly
__device__ __forceinline__ void gen_pack_bias_body(const GenPackArgs& a, int row, int lane) {
    const int C = a.C;
    if (row >= 2 * C) return;
    const int layer = row / C, co = row % C;
    const float* wg = layer ? a.glu_w2 : a.glu_w1;
    const float* bet = layer ? a.beta2 : a.beta1;
    double acc = 0;
    for (int c = lane; c < C; c += 64) acc += (double)wg[(size_t)co * C + c] * (double)bet[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) (layer ? a.bg2 : a.bg1)[co] = (float)(acc + (double)(layer ? a.glu_b2 : a.glu_b1)[co]);
}
// out[n][dir * R + k] = w_dir[k][n]  (R rows, N columns each): the two W_ih stacked along K and transposed.  One 256-thread block
// per 32 x 32 tile through LDS, so that the reads (rows of w) AND the writes (rows of out) are 128-byte runs: the per-element
// version wrote 4 bytes per lane at a stride of 2R floats - 983 k scattered stores per model at H = 256, which made the packing
// workgroups outlast the moments they ride with (43 us instead of 15 at the head of the wide step).  R and N are multiples of 32.
__host__ __device__ inline int gnt_pack_t_tiles(int R, int N) { return 2 * (R / 32) * (N / 32); }
__device__ __forceinline__ void gnt_pack_t_body(const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ out,
                                                int R, int N, int tile, int tid) {
    __shared__ float tl[32][33];
    const int per_dir = (R / 32) * (N / 32);
    if (tile >= 2 * per_dir) return;                                  // (uniform per block)
    const int dir = tile / per_dir, t2 = tile % per_dir, tk = t2 / (N / 32), tn = t2 % (N / 32);
    const float* w = dir ? w1 : w0;
    const int c = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) tl[r0 + 8 * i][c] = w[(size_t)(32 * tk + r0 + 8 * i) * N + 32 * tn + c];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(size_t)(32 * tn + r0 + 8 * i) * 2 * R + dir * R + 32 * tk + c] = tl[c][r0 + 8 * i];
}

// W_hh (H = 256) -> bf16 in the order the one-CU recurrence kernels load it (grec.hip): one 8-vector per thread
//   wp [dir][g][kc][u][8]  = W[g H + u][8 kc + e]          wpT[dir][gc][j][8] = W[8 gc + e][j]
__host__ __device__ inline int grec_pack_blocks() { return (2 * 3 * 256 * 256 / 8 + 255) / 256; }
// f16 (SED_DTYPE_F16): the FORWARD layout wp as fp16 (the forward recurrence's operand type); wpT stays bf16
__device__ __forceinline__ void grec_pack_body(const float* __restrict__ w_f, const float* __restrict__ w_r, __bf16* __restrict__ wp,
                                               __bf16* __restrict__ wpT, int i, int f16 = 0) {
    constexpr int H = 256;
    const int per_dir = 3 * H * H / 8;
    if (i >= 2 * per_dir) return;
    const int dir = i / per_dir, v = i % per_dir;
    const float* w = dir ? w_r : w_f;
    {
        const int u = v % H, kc = (v / H) % (H / 8), g = v / (H * (H / 8));
        const float* s = w + (size_t)(g * H + u) * H + 8 * kc;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f16 ? __builtin_bit_cast(__bf16, (_Float16)s[e]) : (__bf16)s[e];
        *(bf16x8*)(wp + ((size_t)dir * per_dir + v) * 8) = o;
    }
    if (wpT != nullptr) {
        const int j = v % H, gc = v / H;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)w[(size_t)(8 * gc + e) * H + j];
        *(bf16x8*)(wpT + ((size_t)dir * per_dir + v) * 8) = o;
    }
}

// everything a training forward packs, as a list of 256-thread blocks: [pack][bias][gnt layer 0][gnt layer 1][grec layer 0][grec layer 1]
struct GenAuxPack {
    GenPackArgs pk;
    int mode;
    int n_gnt;                                  // 0 (H = 64: the recurrence kernels read W_ih themselves) or the number of GRU layers
    const float *gw0[2], *gw1[2];
    float* gout[2];
    int gR[2], gN[2];
    int n_grec;                                 // GRU layers whose W_hh goes to the bf16 recurrence layout here (0: own launches)
    const float *rw0[2], *rw1[2];
    void *rwp[2], *rwpT[2];
};
__host__ __device__ inline int gen_aux_pack_blocks(const GenAuxPack& a) {
    const int n = 2 * 9 * a.pk.C * a.pk.C;
    return ((n > a.pk.n_zero ? n : a.pk.n_zero) + 255) / 256;
}
__host__ __device__ inline int gen_aux_bias_blocks(const GenAuxPack& a) { return (2 * a.pk.C + 3) / 4; }
__host__ __device__ inline int gen_aux_gnt_blocks(const GenAuxPack& a, int l) { return l < a.n_gnt ? gnt_pack_t_tiles(a.gR[l], a.gN[l]) : 0; }
__host__ __device__ inline int gen_aux_blocks(const GenAuxPack& a) {
    return gen_aux_pack_blocks(a) + gen_aux_bias_blocks(a) + gen_aux_gnt_blocks(a, 0) + gen_aux_gnt_blocks(a, 1) + a.n_grec * grec_pack_blocks();
}
__device__ __forceinline__ void gen_aux_body(const GenAuxPack& a, int pb, int tid) {
    int b = pb;
    const int np = gen_aux_pack_blocks(a);
    if (b < np) {
        if (a.mode == 1) gen_pack_body<1>(a.pk, b * 256 + tid);
        else if (a.mode == 2) gen_pack_body<2>(a.pk, b * 256 + tid);
        else gen_pack_body<0>(a.pk, b * 256 + tid);
        return;
    }
    b -= np;
    const int nbias = gen_aux_bias_blocks(a);
    if (b < nbias) { gen_pack_bias_body(a.pk, b * 4 + (tid >> 6), tid & 63); return; }
    b -= nbias;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int ng = gen_aux_gnt_blocks(a, l);
        if (b < ng) { gnt_pack_t_body(a.gw0[l], a.gw1[l], a.gout[l], a.gR[l], a.gN[l], b, tid); return; }
        b -= ng;
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int nr = l < a.n_grec ? grec_pack_blocks() : 0;
        if (b < nr) { grec_pack_body(a.rw0[l], a.rw1[l], (__bf16*)a.rwp[l], (__bf16*)a.rwpT[l], b * 256 + tid, a.pk.f16); return; }
        b -= nr;
    }
}
