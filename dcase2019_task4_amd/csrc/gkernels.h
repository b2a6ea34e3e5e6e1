// gkernels.h - host-side launchers of the generic kernel family (gen.h); internal.
#pragma once
#include "common.h"

// gconv.hip ---------------------------------------------------------------------------------------------------------------
struct GenPackArgs {
    int C;
    const float *w1, *w2;                       // conv1 / conv2 weights [co][ci][3][3]
    void *wpk1, *wpk2, *wpkT1, *wpkT2;          // packed (element type by mode) [n][9 C]; wpkT may be null (eval)
    const float *glu_w1, *glu_w2, *glu_b1, *glu_b2, *gamma1, *gamma2, *beta1, *beta2;
    void *wg1, *wg2, *wgT1, *wgT2;              // GLU weights folded with gamma [co][c]; transposed raw [c][co] (may be null)
    float *bg1, *bg2;                           // GLU bias folded with beta [C]
    double* zero; int n_zero;                   // fp64 accumulators to clear
    int* err;                                   // unused (the spin-timeout counter is sticky: cleared by sed_crnn_buffers_init only)
    int f16;                                    // SED_DTYPE_F16 (mode 1 family): the FORWARD panels wpk1 / wpk2 as fp16; everything else bf16
};
int launch_gen_pack(const GenPackArgs& a, int mode, hipStream_t st);
int launch_gconv_fwd(int mode, int C, const float* in, const void* wpk, const float* bias, float* y, double* stat, int B, int H,
                     int W, hipStream_t st);
int launch_gconv_dgrad(int mode, int C, const float* dz, const float* yin, const float* coef, const void* wpkT, float* dx, int B,
                       int H, int W, hipStream_t st);
int gwgrad_slabs(int C);
// mode 1: dz, yin, xin are bf16
int launch_gwgrad(int mode, int C, const void* dz, const void* yin, const float* coef, const void* xin, float* part, float* g_w, int B,
                  int H, int W, hipStream_t st);

// bconv.hip: register-blocked bf16 convolution (x3 = 0: bf16 storage, single products; x3 = 1: fp32 storage, split operands)
// x3: 0 bf16 (bf16 storage), 1 split operands (fp32 storage), 2 fp16 forward (SED_DTYPE_F16: fp16 in / panel / out; y_bf16_copy
// is unused - launch_bglu_fwd writes the bf16 copy)
int launch_bconv_fwd(int x3, int C, const void* in, const void* wpk, const float* bias, void* y, double* stat, int B, int H, int W,
                     hipStream_t st, void* y_bf16_copy = nullptr);
int launch_bconv_dgrad(int x3, int C, const void* dz, const void* yin, const float* coef, const void* wpkT, void* dx, int B, int H,
                       int W, hipStream_t st);

// gglu.hip ----------------------------------------------------------------------------------------------------------------
struct GBnArgs {
    const double* stat; double N;               // [2][C] sum, sum of squares of the conv output; element count
    const float *gamma, *beta;
    float *run_mean, *run_var; int64_t* tracked;
    int train, update; float eps, momentum;
    float* bn;                                  // out [4][C]: mean, invstd, scale, shift
};
// mode 1 (SED_DTYPE_BF16): y is bf16; p is bf16 when p_bf16 (block 1) and fp32 otherwise (block 2: the GRU input)
int launch_gglu_fwd(int mode, int C, const void* y, const GBnArgs& bn, const void* wg, const float* bg, void* p, int p_bf16, int B,
                    int H, int W, int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st);
int gglu_bwd_grid(int B, int H, int W);
// part: [grid][C * C + 3 * C] floats (per-workgroup partial sums: dWx | sdb | sdz | sdzx)
// mode 1: y is read and dz written as bf16; dp is bf16 when dp_bf16 (block 1) - block 2's dp comes from the GRU in fp32
int launch_gglu_bwd(int mode, int C, const void* y, const float* bn, const float* gamma, const float* beta, const void* wg,
                    const void* wgT, const float* bg, const void* dp, int dp_bf16, void* dz, float* part, int B, int H, int W,
                    int use_drop, float p_drop, const uint16_t* mask_in, hipStream_t st, const float* dp2 = nullptr);
#define GPART_SLICES 8
struct GBnBwdArgs {
    const float* part; int n_part; int C; double N;
    float* part2;                               // scratch: [GPART_SLICES][C * C + 3 * C] floats (first reduction stage)
    const float *gamma, *beta, *bn;
    float *coef, *g_gamma, *g_beta, *g_wglu, *g_bglu, *g_convb;
};
int launch_gbn_bwd_prep(const GBnBwdArgs& a, hipStream_t st);

// ggru.hip ----------------------------------------------------------------------------------------------------------------
// W_hh re-laid for coalesced streaming: fwd [k / 4][3H rows][4], bwd (transposed) [g / 4][H columns][4]
int launch_ggru_pack(const float* w_hh_f, const float* w_hh_r, float* wp /*[2][3H*H]*/, float* wpT /*[2][3H*H] or null*/, int H,
                     hipStream_t st);
// gi: [B*T][2][3H] (input projection incl. b_ih, both directions); out [B*T][2H]; gates [B*T][2][4H] (r, z, n, gh_n) or null
int launch_ggru_fwd(int H, const float* gi, const float* wp, const float* b_hh_f, const float* b_hh_r, float* out, float* gates,
                    int B, int T, hipStream_t st);
// d_out [B*T][2H]; dgi / dgh [B*T][2][3H]; hprev [B*T][2][H]
int launch_ggru_bwd(int H, const float* d_out, const float* out, const float* gates, const float* wpT, float* dgi, float* dgh,
                    float* hprev, int B, int T, hipStream_t st);

// cluster recurrence (4 workgroups per chain, W_hh in registers, per-step granule exchange through L2); H = 256 only
size_t gclu_xch_bytes(int B, int H, int bwd);
int launch_gclu_fwd(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r, float* out,
                    float* gates, void* xch, unsigned int* epoch, int* err, int B, int T, hipStream_t st);
int launch_gclu_bwd(const float* d_out, const float* out, const float* gates, const float* w_hh_f, const float* w_hh_r, float* dgi,
                    float* dgh, float* hprev, void* xch, unsigned int* epoch, int* err, int B, int T, hipStream_t st);

// bglu.hip: SED_DTYPE_BF16 (bf16 storage) GLU kernels; wglu / bglu are the RAW parameters (the BatchNorm affine is folded in-kernel)
//   wfold_out [C][C] bf16 / bfold_out [C]: the folded weights and bias, published for launch_bglu_bwd (may be null)
int launch_bglu_fwd(int C, const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int p_bf16, int B, int H,
                    int W, int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, void* wfold_out,
                    float* bfold_out, hipStream_t st, int f16 = 0 /* SED_DTYPE_F16: fp16 y / p */, void* p_b16 = nullptr /* + bf16 copy of p */,
                    void* y_b16 = nullptr /* + bf16 copy of the input y */);

int launch_bglu_fwd_x3(int C, const void* y, const GBnArgs& bn, const float* wglu, const float* bglu, void* p, int B, int H, int W,
                       int block_id, int use_drop, float p_drop, const uint64_t* seed, uint16_t* mask_out, hipStream_t st);
int bglu_bwd_grid(int C, int B, int H, int W);
//   wfold / bfold: as published by the forward; wgT: Wglu^T [c][co] raw bf16 (k_gen_pack)
int launch_bglu_bwd(int C, const void* y, const float* bn, const void* wfold, const float* bfold, const void* wgT,
                    const void* dp, int dp_bf16, void* dz, float* part, int B, int H, int W, int use_drop, float p_drop,
                    const uint16_t* mask_in, hipStream_t st, const float* dp2 = nullptr);

// grec.hip: H = 256 recurrence of SED_DTYPE_BF16 - one workgroup per chain, W_hh as bf16 in registers
int launch_grec_pack(const float* w_hh_f, const float* w_hh_r, void* wp, void* wpT /* may be null */, hipStream_t st, int f16 = 0);
int launch_grec_fwd(const float* gi, const void* wp, const float* b_hh_f, const float* b_hh_r, float* out, float* gates, int B, int T,
                    hipStream_t st, int f16 = 0);
int launch_grec_bwd(const float* d_out, const float* out, const float* gates, const void* wpT, float* dgi, float* dgh, float* hprev,
                    int B, int T, hipStream_t st);

// ggemm.hip: C[m][n] = sum_k A[m][k] B[n][k] + bias[n], exact fp32, 128 x 128 tiles; K % 32 == 0
struct GntProb { const float* A; int lda; const float* B; int ldb; float* C; int ldc; const float* bias; int M, N, K; };
struct GntBatch { GntProb p[2]; int n_prob; };
int launch_gnt_gemm(const GntBatch& gb, hipStream_t st);
// bf16 MFMA operands: x3 = 0 SED_DTYPE_BF16 (single products), x3 = 1 SED_DTYPE_BF16X3 (split operands); K % 64 == 0
int launch_gnt_gemm_bf16(const GntBatch& gb, hipStream_t st, int x3 = 0);
int launch_gnt_pack_t(const float* w0, const float* w1, float* out, int R, int N, hipStream_t st);

// gcrnn.hip ---------------------------------------------------------------------------------------------------------------
struct HeadsLoss;
size_t gen_ctx_bytes(const Geo& g);
size_t gen_ws_bytes(const Geo& g);
int gen_buffers_init(const Geo& g, void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, hipStream_t st);
int gen_ctx_view(const Geo& g, const char* name, size_t* offset, size_t* bytes);
int gen_mompart(const Geo& g, void* ctx, size_t ctx_bytes, double** out);      // where the patch-moment partials live in a generic ctx
int gen_forward(const Geo& g, const ParamOff& P, const float* params, float* bn_running, int64_t* bn_tracked, const float* x,
                int train, int update_bn, const uint64_t* seed_dev, void* ctx, size_t ctx_bytes, float* strong, float* weak,
                hipStream_t st, hipStream_t ss, hipEvent_t ev_fork, hipEvent_t ev_join);
int gen_backward(const Geo& g, const ParamOff& P, const float* params, const float* x, const uint64_t* seed_dev, void* ctx,
                 size_t ctx_bytes, const float* d_strong, const float* d_weak, float* grads, void* ws, size_t ws_bytes, int parts,
                 hipStream_t st, hipStream_t ss, hipEvent_t ev_fork, hipEvent_t ev_join, hipStream_t ss2, hipEvent_t ev_join2,
                 const HeadsLoss* hl, const HeadsOut* ho);
