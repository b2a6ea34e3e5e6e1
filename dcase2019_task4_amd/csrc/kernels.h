// kernels.h - host-side launchers of the individual kernels (internal; the public surface is
// include/dcase_sed.h).
#pragma once
#include "common.h"

// blk0.hip
struct ConvPackArgs;
int x_moments_parts(const Geo& g);
// pack != null: the conv1 / conv2 weight packing (independent work of the same step) rides along as extra workgroups
struct GenAuxPack;   // gpack.h: the generic kernel set's per-forward packing, run in spare workgroups of the moments launch
int launch_x_moments(const Geo& g, const float* x, double* mompart, const ConvPackArgs* pack, hipStream_t st,
                     const GenAuxPack* aux = nullptr);
int launch_blk0_forward(const Geo& g, const float* x, const float* w0, const float* b0, const float* gamma,
                        const float* beta, const float* wglu, const float* bglu, float* run_mean, float* run_var,
                        int64_t* tracked, int train, int update, const uint64_t* seed, double* mom, double* mompart,
                        float* wz, float* wl, float* bn, float* p0, uint16_t* mask_out, const ConvPackArgs* pack /* train only */,
                        hipStream_t st,
                        int main_kernel_only = 0, const GenAuxPack* aux = nullptr,
                        void* p0_b16 = nullptr /* g.f16: p0 is fp16 and this (may be null) receives its bf16 copy */,
                        int mom_ready = 0 /* the batch's patch moments are already in mompart (sed_crnn_moments) */,
                        void* sg_out = nullptr /* bf16 family, differentiated forward: receives the GLU gate bytes (blk0.hip SG) */);
int launch_blk0_backward(const Geo& g, const float* x, const float* w0, const float* b0, const float* gamma,
                         const float* beta, const float* wglu, const uint16_t* mask_in, const double* mom,
                         const float* wz, const float* wl, const float* bn, const float* dp0, double* de, int zero_de,
                         float* g_w0, float* g_b0, float* g_gamma, float* g_beta, float* g_wglu, float* g_bglu,
                         hipStream_t st, const void* sg_in = nullptr /* the forward's saved gate bytes (bf16 family) */);

// conv.hip : 3x3, 64 -> 64 channels, channels-last, image [B][H][W][64] with W in {16, 4}
// ---- conv weight packing (conv.hip; also called from k_x_moments, blk0.hip) -----------------------------------------
struct ConvPackArgs {
    const float *w1, *w2;            // conv1 / conv2 weights [co][ci][3][3]
    float *wpk1, *wpk2;              // [tap][ci][co] | Winograd panel (SED_WINO_OFF floats in)
    float *wpkT1, *wpkT2;            // the same for dgrad (flipped / transposed), may be null
    double* zero; int n_zero;        // fp64 accumulators to clear (the forward's BatchNorm sums)
};
#define SED_PACK_BLOCKS ((2 * 9 * 4096 + 255) / 256)
#ifdef __HIPCC__
// U = G g G^T of one 3x3 kernel (Winograd F(2x2, 3x3)) for the pair (k, n) of the B operand [k][n], written in the order
// the 8 waves of k_conv16_wino hold it in registers: Uf[ph][cg][p8][sq][lane][e] with lane = 16 kq + i16,
// k = 4 (4 sq + e) + kq, n = 16 cg + i16; wave row ph: p8 < 4 <-> transform row 3 ph, p8 >= 4 <-> row 1 + ph; column p8 & 3
__device__ __forceinline__ void wino_u(const float (&g)[3][3], float* __restrict__ U, int k, int n) {
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int kq = k & 3, s4 = k >> 2, sq = s4 >> 2, e = s4 & 3, cg = n >> 4, i16 = n & 15;
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
        // transform row i2 lives in wave row ph = (i2 >= 2), as its X row (i2 = 0, 3) or its Y row (i2 = 1, 2)
        const int ph = i2 >> 1, p8b = (i2 == 0 || i2 == 3) ? 0 : 4;
        float* d = U + ((((size_t)(ph * 4 + cg) * 8 + p8b) * 4 + sq) * 64 + 16 * kq + i16) * 4 + e;
        const size_t pstride = 4 * 64 * 4;          // p8 -> p8 + 1
        d[0 * pstride] = t[i2][0];
        d[1 * pstride] = 0.5f * (t[i2][0] + t[i2][1] + t[i2][2]);
        d[2 * pstride] = 0.5f * (t[i2][0] - t[i2][1] + t[i2][2]);
        d[3 * pstride] = t[i2][2];
    }
}
// thread i of SED_PACK_BLOCKS x 256: one element of [layer][tap][ci][co] (+ one (k, n) pair of the Winograd panels)
__device__ __forceinline__ void conv_pack_body(const ConvPackArgs& a, int i) {
    if (i < a.n_zero) a.zero[i] = 0.0;                   // the forward's fp64 BatchNorm accumulators (saves a memset node)
    if (i < 4 * 4096) {                              // Winograd panels: [layer][forward | dgrad]
        const int which = i >> 12, kn = i & 4095, k = kn >> 6, n2 = kn & 63;
        const float* w = (which >> 1) ? a.w2 : a.w1;
        float* dst = (which >> 1) ? ((which & 1) ? a.wpkT2 : a.wpk2) : ((which & 1) ? a.wpkT1 : a.wpk1);
        if (dst != nullptr) {
            float g[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    // forward: B[k = ci][n = co] from g[a][b] = W[co][ci][a][b]; dgrad: B[k = co][n = ci] from the flipped kernel
                    g[a][b] = (which & 1) ? w[(k * 64 + n2) * 9 + 3 * (2 - a) + (2 - b)] : w[(n2 * 64 + k) * 9 + 3 * a + b];
            wino_u(g, dst + SED_WINO_OFF, k, n2);
        }
    }
    const int layer = i / (9 * 4096);
    i -= layer * 9 * 4096;
    const float* w = layer ? a.w2 : a.w1;
    float* wpk = layer ? a.wpk2 : a.wpk1;
    float* wpkT = layer ? a.wpkT2 : a.wpkT1;
    const int tap = i / 4096, ci = (i / 64) % 64, co = i % 64;
    wpk[i] = w[(co * 64 + ci) * 9 + tap];
    if (wpkT) {
        // transposed conv for dgrad: wpkT[tap'][k = co][n = ci] = W[co][ci][8 - tap']
        const int k = ci, n = co;      // reuse the index split: (i/64)%64 -> k, i%64 -> n
        wpkT[i] = w[(k * 64 + n) * 9 + (8 - tap)];
    }
}
#endif

// packs conv1 and conv2 weights [co][ci][3][3] -> wpk [tap][ci][co] (+ flipped/transposed wpkT for dgrad) + Winograd panels
int launch_conv_pack(const ConvPackArgs& a, hipStream_t st);
// forward: y = conv(in) + bias; optional per-channel sum / sum-of-squares (fp64 atomics into stat[128])
int launch_conv_fwd(const float* in, const float* wpk, const float* bias, float* y, double* stat, int zero_stat, int B,
                    int H, int W, hipStream_t st);
// BatchNorm-backward reduction results of one conv block (filled by launch_glu_pool_bwd) -> per-channel affine
// dy = ca*dz + cb*y + cc for the conv dgrad / wgrad loaders, and the block's parameter gradients.  Either a 1-workgroup
// kernel of its own turns them into `coef` (k_bn_bwd_prep: 5 us + a launch gap on the critical chain, twice per step),
// or - default - the consumers do it themselves in their prologue: they get this struct (`prep`, acc != null), every
// workgroup derives the 192 coefficients from the fp64 sums (a few double operations per channel), and workgroup 0 of
// the dgrad kernel also writes the parameter gradients and `coef`.
struct BnBwdPrepArgs {
    const double* acc; double N;
    const float *gamma, *bn;
    float *coef, *g_gamma, *g_beta, *g_wglu, *g_bglu, *g_convb;
};
#ifdef __HIPCC__
__device__ __forceinline__ void bn_bwd_coef(const BnBwdPrepArgs& a, int c, float& ca, float& cb, float& cc) {
    const double mean = a.bn[c], invstd = a.bn[64 + c], scale = a.bn[128 + c];
    const double Sdz = a.acc[4160 + c], Sdzy = a.acc[4224 + c];
    const double Sdzxhat = invstd * (Sdzy - mean * Sdz);
    const double m1 = Sdz / a.N, m2 = Sdzxhat / a.N;
    // dy = scale * (dz - m1 - xhat*m2),  xhat = (y - mean) * invstd
    ca = (float)scale;
    cb = (float)(-scale * m2 * invstd);
    cc = (float)(scale * (m2 * invstd * mean - m1));
}
__device__ __forceinline__ void bn_bwd_prep_body(const BnBwdPrepArgs& a, int tid, int nthreads) {
    if (tid < 64) {
        const int c = tid;
        const double mean = a.bn[c], invstd = a.bn[64 + c];
        const double Sdz = a.acc[4160 + c], Sdzy = a.acc[4224 + c];
        a.g_beta[c] = (float)Sdz;
        a.g_gamma[c] = (float)(invstd * (Sdzy - mean * Sdz));
        float ca, cb, cc;
        bn_bwd_coef(a, c, ca, cb, cc);
        a.coef[c] = ca; a.coef[64 + c] = cb; a.coef[128 + c] = cc;
        a.g_bglu[c] = (float)a.acc[4096 + c];
        a.g_convb[c] = 0.f;   // sum_p dy == 0: a conv bias in front of a train-mode BatchNorm has zero gradient
    }
    for (int e = tid; e < 4096; e += nthreads) a.g_wglu[e] = (float)a.acc[e];
}
#endif
// dgrad: dx = conv_flipped(dy), dy = ca*dz + cb*yin + cc (per channel) inside the image, 0 outside
//   prep != null (Winograd kernels only): the coefficients are derived in the kernel (see BnBwdPrepArgs)
int launch_conv_dgrad(const float* dz, const float* yin, const float* coef /*[3][64]*/, const float* wpkT, float* dx,
                      int B, int H, int W, const BnBwdPrepArgs* prep, hipStream_t st);
// wgrad: dW[co][ci][tap] = sum_p dy[p][co] * xin[p+tap][ci]; dy as above
int launch_conv_wgrad(const float* dz, const float* yin, const float* coef, const float* xin, float* part, int n_blocks,
                      float* g_w /*[co][ci][3][3]*/, int B, int H, int W, const BnBwdPrepArgs* prep, hipStream_t st);

// bnglu.hip
// BN statistics -> (mean, invstd, scale, shift) (+ running-stat update) happens in the kernel's prologue
int launch_glu_pool_fwd(const float* y, const double* stat, double N, const float* gamma, const float* beta, float* run_mean,
                        float* run_var, int64_t* tracked, int train, int update, float eps, float momentum, float* bn /*[4][64]*/,
                        const float* wglu, const float* bglu, float* p, int B, int H, int W, int block_id, int use_drop,
                        float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st);
// backward pass 1: dz (full-res grad wrt BN output), GLU weight grads and BN reduction sums
//   acc: double [64*64 (dWglu) + 64 (dbglu) + 64 (sum dz) + 64 (sum dz*y)]
int launch_glu_pool_bwd(const float* y, const float* bn, const float* wglu, const float* bglu, const float* dp, const float* dp_b, float* dz,
                        double* acc, int zero_acc, int B, int H, int W, int block_id, int use_drop, float p_drop,
                        const uint16_t* mask_in, const float* gamma, float* coef, float* g_gamma, float* g_beta, float* g_wglu,
                        float* g_bglu, float* g_convb, BnBwdPrepArgs* prep_out /* non-null: no k_bn_bwd_prep, see BnBwdPrepArgs */,
                        hipStream_t st);

// gemm.hip : batched / split-K strided GEMM  C[m][n] = sum_k A(m,k) B(k,n) (+ bias[n]) (+ C)
struct GemmProb {
    const float* A; int64_t sAm, sAk;
    const float* B; int64_t sBk, sBn;
    const float* B2; int k2;      // optional: rows k >= k2 of B come from B2[(k - k2)]
    float* C; int64_t ldc;
    const float* bias;            // per output column (may be null)
    float* Cones;                 // optional: receives sum_k A(m,k) (virtual all-ones column n == N)
    int M, N, K;
    int accumulate;
    int vec_ok;                   // set by launch_gemm_batch: every in-range group of 4 is one aligned float4
};
struct GemmBatch {
    GemmProb p[4];
    int n_prob;
    int splits;                   // split-K factor (>1 needs part)
    float* part;                  // scratch: n_prob * splits * max(M) * max(N+1) floats
    size_t part_floats;           // capacity of `part` in floats (checked by launch_gemm_batch)
    size_t part_stride;           // filled by launch_gemm_batch
    int bf16 = 0;                 // operands rounded to bf16 inside the kernel (bf16 MFMA, fp32 accumulation)
};
static inline GemmProb gemm_prob(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBk, int64_t sBn, float* C,
                                 int64_t ldc, int M, int N, int K) {
    GemmProb q;
    q.A = A; q.sAm = sAm; q.sAk = sAk; q.B = B; q.sBk = sBk; q.sBn = sBn; q.B2 = nullptr; q.k2 = 0;
    q.C = C; q.ldc = ldc; q.bias = nullptr; q.Cones = nullptr; q.M = M; q.N = N; q.K = K; q.accumulate = 0; q.vec_ok = 0;
    return q;
}
size_t gemm_part_floats(int n_prob, int splits, int max_m, int max_nx);
int launch_gemm_batch(GemmBatch& gb, hipStream_t st);
int launch_colsum(const float* A, int M, int N, int64_t lda, float* out, hipStream_t st);

// gru.hip
// crnn.hip: runs the host callback registered by sed_crnn_fork_callback (if any) for st - called by both forwards between the conv
// stack and the recurrence
int sed_fork_point(hipStream_t st);
int launch_gru_fwd(const float* x, int nin, const float* w_ih_f, const float* w_ih_r, const float* b_ih_f, const float* b_ih_r,
                   const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r, float* out, float* gates,
                   int B, int T, hipStream_t st);
int launch_gru_bwd(const float* d_out, const float* d_out2, const float* out, const float* gates, const float* w_hh_f,
                   const float* w_hh_r, const float* w_ih_f, const float* w_ih_r, int nin, float* dgi, float* dgh, float* hprev,
                   float* dx_planes, int B, int T, hipStream_t st);

// heads.hip
int launch_heads_fwd(const float* h /*[B][T][128]*/, const float* wd, const float* bd, const float* ws, const float* bs,
                     float* strong, float* weak, float* strong_sv, float* weak_sv, float* logits_s, float* den, int B, int T,
                     int NC, int use_drop, float p_drop, const uint64_t* seed, hipStream_t st, int HF = 128);
int launch_heads_colsum(const float* part, float* g_wd, int B, int NC, hipStream_t st, int HF = 128);
// ---- device-side step state (sed_step_state): derived fields ------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// what the forward passes and the loss of step `global_step` read
__device__ __forceinline__ void step_state_derive_forward(sed_step_state* s) {
    const int64_t gs = s->global_step;
    // sigmoid_rampup (utils/ramps.py:20-27) behind the `global_step < rampup_length` test of main.py:74-78
    double r = 1.0;
    if (gs < s->rampup_length && s->rampup_length > 0) {
        double cur = (double)gs;
        if (cur < 0) cur = 0;
        const double phase = 1.0 - cur / (double)s->rampup_length;
        r = exp(-5.0 * phase * phase);
    }
    s->cons_weight = (float)(s->max_cons_cost * r);
    s->seed_student = splitmix64(s->base_seed + 2ull * (uint64_t)gs * 0x9E3779B97F4A7C15ull + 1ull);
    s->seed_teacher = splitmix64(s->base_seed + (2ull * (uint64_t)gs + 1ull) * 0x9E3779B97F4A7C15ull + 1ull);
}
// what the Adam + EMA update of step `global_step` (Adam step `opt_step`) reads
__device__ __forceinline__ void step_state_derive_update(sed_step_state* s) {
    const int64_t gs = s->global_step;
    // update_ema_variables is called with global_step already incremented (main.py:155-157)
    const double a = 1.0 - 1.0 / ((double)(gs + 1) + 1.0);
    s->ema_alpha = (float)(a < s->ema_decay ? a : s->ema_decay);
    const double bc1 = 1.0 - pow(s->beta1, (double)s->opt_step);
    const double bc2 = 1.0 - pow(s->beta2, (double)s->opt_step);
    s->adam_step_size = (float)(s->lr / bc1);
    s->adam_sqrt_bc2 = (float)sqrt(bc2);
}
__device__ __forceinline__ void step_state_derive(sed_step_state* s) {
    step_state_derive_forward(s);
    step_state_derive_update(s);
}
// The advance of the fused step, done by the last workgroup of the loss / heads-backward kernel instead of a 1-thread
// kernel at the end of the step (4.6 us on the tail): at that point every forward kernel and the loss have read their
// fields of this step, and the update has not run yet - so the update's fields are (re)derived for THIS step, then the
// counters move on and the forward's fields are derived for the NEXT step.  At step boundaries the counters and the
// forward fields are the same as with sed_step_state_advance; the update fields lag by one step until they are needed.
__device__ __forceinline__ void step_state_advance_early(sed_step_state* s) {
    step_state_derive_update(s);
    s->global_step += 1;
    s->opt_step += 1;
    step_state_derive_forward(s);
}
#endif

// loss inputs of the fused loss + heads backward (sed_mt_loss_backward); strong_ema == null: gradients come from the caller
struct HeadsLoss {
    const float* strong_ema; const float* weak_ema; const float* target;
    int wlo, whi, slo, shi;
    const sed_step_state* state;
    float* losses; float* d_strong_out; float* d_weak_out;
    sed_step_state* advance;      // non-null: the last workgroup also advances the step state (step_state_advance_early)
};
int launch_heads_bwd(const float* h, const float* wd, const float* ws, const float* strong, const float* weak,
                     const float* logits_s, const float* den, const float* d_strong, const float* d_weak, float* dh,
                     float* part, float* g_wd, float* g_bd, float* g_ws, float* g_bs, int B, int T, int NC, int use_drop,
                     float p_drop, const uint64_t* seed, double* zero, int n_zero, int defer_colsum, const HeadsLoss* hl, hipStream_t st,
                     int HF = 128);

// ---- round 5: heads forward + loss + heads backward as the prologue of the top layer's backward recurrence (hfuse.h, gru4.hip) ----
struct HeadsFuse {
    const float* wd;                    // dense.weight; dense.bias, dense_softmax.weight, dense_softmax.bias follow it (flat layout)
    float *strong, *weak;               // posteriors out: [B][T'][NC], [B][NC]
    float* part;                        // per-clip partial weight gradients [B][2 (NC * 128 + NC)]
    int NC, use_drop;
    float p_drop;
    const uint64_t* seed;
    double* zero; int n_zero;           // fp64 accumulators of the conv-block backward to clear (may be null / 0)
    HeadsLoss hl;                       // strong_ema != null always (the fused form IS the loss); d_strong_out / d_weak_out unused
};
// k_heads_bwd at HF = 512 (the 256-cell BiGRU) works through a clip in 32-frame chunks; they are independent of each other (the
// per-clip terms come from the forward's saved sums), so a clip's chunks go to separate workgroups: grid (B, chunks), one
// weight-gradient partial slab and one loss partial per workgroup (the loss partials of a multi-chunk launch live behind the
// slabs in the workspace: sed_mt_loss's SED_LOSS_FLOATS(B) contract is unchanged).  HF = 128: one 128-frame chunk, one workgroup.
static inline int heads_bwd_chunks(int HF, int T3) {
    if (HF != 512) return 1;
    const int n = (T3 + 31) / 32;
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}
static inline size_t heads_part_floats(int B, int T3, int NC, int HF) {
    const int ny = heads_bwd_chunks(HF, T3);
    return (size_t)B * ny * 2 * (NC * HF + NC) + (size_t)8 * B * ny;
}
struct HeadsOut { float *strong, *weak; };       // sed_mt_step_backward: where the deferred heads' posteriors go
// the frames (T / 8) and hidden size the fused form serves; everything else takes k_heads_fwd + k_heads_bwd
static inline bool heads_fusable(int H, int T3) { return H == 64 && T3 <= 128; }
// top BiGRU layer's backward recurrence with the heads phase in front (d_out comes out of LDS); otherwise as launch_gru_bwd
int launch_gru_bwd_heads(const float* out, const float* gates, const float* w_hh_f, const float* w_hh_r, const float* w_ih_f,
                         const float* w_ih_r, int nin, float* dgi, float* dgh, float* hprev, float* dx_planes, int B, int T,
                         const HeadsFuse& hf, hipStream_t st);
// column sum of the per-clip head weight-gradient partials (n_cols == 0: skipped) + the loss meters' clip sums + (hl.advance)
// the step-state advance that k_heads_bwd's last workgroup does in the two-kernel form
int launch_heads_fin(const float* part, float* g_wd, int B, int T, int NC, int n_cols, const HeadsLoss& hl, hipStream_t st);

