// heads.hip - output heads of the CRNN and the mean-teacher loss.
//
// Reference ops (baseline/models/CRNN.py:74-81):
//   x = Dropout(p)(gru_out)
//   strong = sigmoid(dense(x))                                   [B, T', nclass]
//   sof    = clamp(softmax(dense_softmax(x), dim=-1), 1e-7, 1)   (softmax over CLASSES)
//   weak   = (strong * sof).sum(1) / sof.sum(1)                  [B, nclass]
// and the loss block of main.train (baseline/main.py:93-145).  One workgroup per clip; the work is
// tiny (0.4 MFLOP / clip) so everything is fused: dropout mask (Philox; backward re-draws it, one draw per 16 features),
// both Linear layers, sigmoid, softmax, clamp and the attention pooling.
#include "common.h"
#include "philox.h"
#include "kernels.h"
#include "hfuse.h"

#define HD_THREADS 1024
// Templated on HF = 2 * n_RNN_cell (128 for the reference's 64 cells, 512 for BASELINE.json configs[4]'s 256):
//   HD_TC  frames per chunk: 1024 threads = HD_TC frames x HF / 16 feature groups (128 frames at HF = 128, 32 at HF = 512)
//   HD_SF  forward row stride of x and W (both read as (row = lane & 15, col = k)): HF + 4
//   HD_SB  backward row stride of x and W (both read as (row = k, col = lane & 15)): HF + 16
#define HD_CONSTS(HF) constexpr int HD_F = (HF), HD_TC = ((HF) == 128 ? 128 : 32), HD_SF = (HF) + 4, HD_SB = (HF) + 16, HD_GPF = (HF) / 16; \
    (void)HD_F; (void)HD_TC; (void)HD_SF; (void)HD_SB; (void)HD_GPF
#define HD_MAXO 32    // 2 * max nclass
// The first version did the three small matrix products with scalar FMAs fed from LDS (2 ds_reads per FMA):
// 2-4 MB of LDS traffic per workgroup, 25 us forward / 37 us backward for 0.4 MFLOP per clip, all of it on the
// critical path of the step.  Now they are v_mfma_f32_16x16x4_f32 tiles (one or a few per wave); row strides are
// chosen per use so that the fragment reads are bank-conflict free: a fragment indexed (row = lane & 15,
// col = 4s + (lane >> 4)) wants stride = 4 (mod 64), one indexed (row = 4s + (lane >> 4), col = lane & 15) wants
// stride = 16 (mod 64).
#define HD_SD 33      // backward: dl rows (read both ways; small residual conflicts)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Stages one chunk of <= 128 frames of the GRU output with the recurrent-output dropout applied (CRNN.py:74):
// each thread owns 16 consecutive features of one frame = ONE Philox draw (flat stream 8: index = element >> 4,
// byte = element & 15).  Returns the 16 keep bits (all ones without dropout, 0 for frames past T).
template <int HF, int STRIDE>
__device__ __forceinline__ uint32_t heads_stage(const float* __restrict__ h, float* xs, int b, int T, int t0, int use_drop,
                                                uint64_t seed, uint32_t thr, float ks, int tid) {
    HD_CONSTS(HF);
    const int tl = tid / HD_GPF, f0 = (tid % HD_GPF) * 16, t = t0 + tl;
    float v[16];
    uint32_t keep = 0;
    if (t < T) {
        const size_t ge = (size_t)(b * T + t) * HD_F + f0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 u = *(const float4*)(h + ge + 4 * q);
            v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
        }
        keep = 0xffffu;
        if (use_drop) keep = philox_keep16(philox_stream((uint32_t)(ge >> 4), 8u, seed), thr);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) xs[tl * STRIDE + f0 + i] = ((keep >> i) & 1u) ? v[i] * ks : 0.f;
    return keep;
}
// both weight matrices as rows [0, NC) = dense, [NC, 2NC) = dense_softmax, zero rows up to HD_MAXO
template <int HF, int STRIDE>
__device__ __forceinline__ void heads_stage_w(const float* __restrict__ wd, const float* __restrict__ ws, float* wsm, int NC, int tid) {
    HD_CONSTS(HF);
    // one float4 per trip from a SELECTED row pointer (never a load under the row condition: as `o < NC ? wd[..] : ...` every
    // trip was a branch with its own s_waitcnt - 4 (HF = 128) or 16 (HF = 512) serialized round trips at the head of both
    // heads kernels), all trips issued before the first store
    constexpr int N4 = HD_MAXO * HD_F / 4, TRIPS = (N4 + HD_THREADS - 1) / HD_THREADS;
    f32x4 v[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int e4 = tid + HD_THREADS * i, o = (4 * e4) / HD_F, f = (4 * e4) % HD_F;
        const int oc = o < 2 * NC ? o : 2 * NC - 1;
        const float* row = oc < NC ? wd + (size_t)oc * HD_F : ws + (size_t)(oc - NC) * HD_F;
        v[i] = *(const f32x4*)(row + (e4 < N4 ? f : 0));
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int e4 = tid + HD_THREADS * i, o = (4 * e4) / HD_F, f = (4 * e4) % HD_F;
        if (e4 < N4) *(f32x4*)&wsm[o * STRIDE + f] = o < 2 * NC ? v[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

template <int HF>
__global__ __launch_bounds__(HD_THREADS) void k_heads_fwd(const float* __restrict__ h, const float* __restrict__ wd,
                                                    const float* __restrict__ bd, const float* __restrict__ ws,
                                                    const float* __restrict__ bs, float* __restrict__ strong,
                                                    float* __restrict__ weak, float* __restrict__ strong_sv,
                                                    float* __restrict__ weak_sv, float* __restrict__ logits_s,
                                                    float* __restrict__ den_out, int T, int NC, int use_drop, float p_drop,
                                                    const uint64_t* __restrict__ seed_ptr) {
    HD_CONSTS(HF);
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* xs = hsm;                                   // [HD_TC][HD_SF]
    float* wsm = xs + HD_TC * HD_SF;                   // [HD_MAXO][HD_SF]
    float* lg = wsm + HD_MAXO * HD_SF;                 // [HD_TC][HD_MAXO]
    float (*nums)[16] = (float (*)[16])(lg + HD_TC * HD_MAXO);
    float (*dens)[16] = (float (*)[16])(lg + HD_TC * HD_MAXO + HD_TC * 16);
    float* num_acc = lg + HD_TC * HD_MAXO + 2 * HD_TC * 16;
    float* den_acc = num_acc + 16;
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float ks = use_drop ? drop_scale8(p_drop) : 1.0f;
    heads_stage_w<HF, HD_SF>(wd, ws, wsm, NC, tid);
    if (tid < 16) { num_acc[tid] = 0.f; den_acc[tid] = 0.f; }
    for (int t0 = 0; t0 < T; t0 += HD_TC) {
        __syncthreads();
        heads_stage<HF, HD_SF>(h, xs, b, T, t0, use_drop, seed, thr, ks, tid);
        __syncthreads();
        if (wv < 2 * (HD_TC / 16)) {   // logits[t][o] = x[t][:] . W[o][:] + bias: (HD_TC / 16) x 2 tiles of 16 x 16, one per wave, K = HF
            const int rt = wv >> 1, ct = wv & 1, i = lane & 15, kq = lane >> 4;
            const float* A = xs + (16 * rt + i) * HD_SF + kq;
            const float* Bp = wsm + (16 * ct + i) * HD_SF + kq;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
            for (int s4 = 0; s4 < HD_F / 4; s4 += 2) {
                acc0 = mfma16(A[4 * s4], Bp[4 * s4], acc0);
                acc1 = mfma16(A[4 * s4 + 4], Bp[4 * s4 + 4], acc1);
            }
            const int o = 16 * ct + i;
            const float bias = (o < NC) ? bd[o] : (o < 2 * NC ? bs[o - NC] : 0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) lg[(16 * rt + 4 * kq + r) * HD_MAXO + o] = acc0[r] + acc1[r] + bias;
        }
        __syncthreads();
        {   // softmax over classes + sigmoid, 8 threads per frame (thread `sub`: classes sub, sub + 8; DPP reductions inside
            // the 8-lane group) - one thread per frame left 7/8 of the workgroup idle here
            const int tl = tid >> 3, sub = tid & 7, t = t0 + tl;
            auto red8_sum = [](float v) {
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
                return v;
            };
            auto red8_max = [](float v) {
                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
                return v;
            };
            const bool has[2] = {sub < NC, sub + 8 < NC};
            if (tl >= HD_TC) {
                // (HF = 512: only the first HD_TC * 8 threads own a frame of the chunk)
            } else if (t < T) {
                float ls[2], ex[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) ls[q] = has[q] ? lg[tl * HD_MAXO + NC + sub + 8 * q] : -3.0e38f;
                const float mx = red8_max(fmaxf(ls[0], ls[1]));
#pragma unroll
                for (int q = 0; q < 2; ++q) ex[q] = has[q] ? __expf(ls[q] - mx) : 0.f;
                const float inv = rcp_fast(red8_sum(ex[0] + ex[1]));
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (has[q]) {
                        const int c = sub + 8 * q;
                        const float sof = fminf(fmaxf(ex[q] * inv, 1e-7f), 1.0f);
                        const float sv = sigmoidf_fast(lg[tl * HD_MAXO + c]);
                        strong[(size_t)(b * T + t) * NC + c] = sv;
                        if (strong_sv) strong_sv[(size_t)(b * T + t) * NC + c] = sv;
                        logits_s[(size_t)(b * T + t) * NC + c] = ls[q];
                        nums[tl][c] = sv * sof;
                        dens[tl][c] = sof;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (has[q]) { nums[tl][sub + 8 * q] = 0.f; dens[tl][sub + 8 * q] = 0.f; }
            }
        }
        __syncthreads();
        if (wv < NC) {          // wave c sums class c over the chunk's frames
            float a = 0.f, d2 = 0.f;
#pragma unroll
            for (int tl2 = lane; tl2 < HD_TC; tl2 += 64) { a += nums[tl2][wv]; d2 += dens[tl2][wv]; }
            a = wave_sum(a); d2 = wave_sum(d2);
            if (lane == 0) { num_acc[wv] += a; den_acc[wv] += d2; }
        }
    }
    __syncthreads();
    if (tid < NC) {
        const float wk = num_acc[tid] / den_acc[tid];
        weak[b * NC + tid] = wk;
        if (weak_sv) weak_sv[b * NC + tid] = wk;
        den_out[b * NC + tid] = den_acc[tid];
    }
}

// ---- mean-teacher loss terms (main.py:93-145): bce_term / bce_grad live in hfuse.h (shared with the fused form) ----------

// part row layout (matches the flat parameter order dense.weight, dense.bias, dense_softmax.weight,
// dense_softmax.bias): [NC*128 dWd][NC dbd][NC*128 dWs][NC dbs]
// (22-24 us whether the clips have 20 or 128 frames: a chain of memory round trips on B workgroups.  Tried without gain:
// every front-part load issued before the first barrier (23.1 vs 23.3 us); the loss partials / ticket / last-workgroup
// tail moved in front of the MFMA + store phase (30 us: then every workgroup waits for the fence and the ticket mid-way).)
// With hl.strong_ema != null the kernel also IS the loss (sed_mt_loss_backward): the gradient of the mean-teacher loss
// w.r.t. the student's posteriors needs no reduction over the batch (the BCE / MSE normalisers are known constants), so
// each clip's workgroup forms it on the fly instead of reading it from a separate kernel's output - k_mt_loss was 12 us
// on the critical path between the forward and the backward.  The six loss sums go the same way as there: per-clip
// partials, the last workgroup (device-scope ticket) adds them up in clip order.
template <int HF>
__global__ __launch_bounds__(HD_THREADS) void k_heads_bwd(const float* __restrict__ h, const float* __restrict__ wd,
                                                    const float* __restrict__ ws, const float* __restrict__ strong,
                                                    const float* __restrict__ weak, const float* __restrict__ logits_s,
                                                    const float* __restrict__ den, const float* __restrict__ d_strong,
                                                    const float* __restrict__ d_weak, float* __restrict__ dh,
                                                    float* __restrict__ part, int T, int NC, int use_drop, float p_drop,
                                                    const uint64_t* __restrict__ seed_ptr, double* __restrict__ zero, int n_zero,
                                                    HeadsLoss hl) {
    HD_CONSTS(HF);
    constexpr int TPW = HF / 128;                      // dW tiles (16 outputs x 16 features) per wave: 2 x HF / 16 tiles, 16 waves
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    // the fp64 accumulators of the conv-block backward that follows (saves a memset node on the critical path)
    if (blockIdx.y == 0)
        for (int i = blockIdx.x * HD_THREADS + threadIdx.x; i < n_zero; i += gridDim.x * HD_THREADS) zero[i] = 0.0;
    float* xs = hsm;                                   // [HD_TC][HD_SB]
    float* wsm = xs + HD_TC * HD_SB;                   // [HD_MAXO][HD_SB]
    float* dl = wsm + HD_MAXO * HD_SB;                 // [HD_TC][HD_SD]
    float* dnum = dl + HD_TC * HD_SD;
    float* dden = dnum + 16;
    uint32_t* mk = (uint32_t*)(dden + 16);             // [HD_TC][HF / 16] keep bits of the staged chunk
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NY = gridDim.y, cy = blockIdx.y;         // this workgroup's share of the clip: chunks cy, cy + NY, ... (kernels.h heads_bwd_chunks)
    const int i16 = lane & 15, kq = lane >> 4;
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float ks = use_drop ? drop_scale8(p_drop) : 1.0f;
    const int NO = 2 * NC;
    heads_stage_w<HF, HD_SB>(wd, ws, wsm, NC, tid);
    const bool fused = hl.strong_ema != nullptr;
    const int B = gridDim.x;
    float lacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // weak_bce, strong_bce, mse_strong, mse_weak, weak_ema_bce, strong_ema_bce
    const float cw = fused ? hl.state->cons_weight : 0.f;
    const float inv_nS = 1.0f / (float)(B * T * NC), inv_nW = 1.0f / (float)(B * NC);
    const float inv_sb = (hl.shi > hl.slo) ? 1.0f / (float)((hl.shi - hl.slo) * T * NC) : 0.f;
    const float inv_wb = (hl.whi > hl.wlo) ? 1.0f / (float)((hl.whi - hl.wlo) * NC) : 0.f;
    const bool in_s = fused && b >= hl.slo && b < hl.shi, in_w = fused && b >= hl.wlo && b < hl.whi;
    float* tmaxs = dden + 16;                          // (aliases the mask words, which are first written inside the loop)
    if (in_w && wv < NC) {                             // target_weak = target.max(-2) (main.py:95): wave c takes class c
        float t = -3.0e38f;
        for (int tt = lane; tt < T; tt += 64) t = fmaxf(t, hl.target[((size_t)b * T + tt) * NC + wv]);
        t = wave_max(t);
        if (lane == 0) tmaxs[wv] = t;
    }
    if (fused) __syncthreads();
    if (tid < NC) {
        float dw;
        if (fused) {
            const float p = weak[b * NC + tid], pe = hl.weak_ema[b * NC + tid];
            const float diff = p - pe;
            const float once = cy == 0 ? 1.f : 0.f;          // the clip-level terms enter the sums through ONE of the clip's workgroups
            lacc[3] = once * diff * diff;
            dw = cw * 2.0f * diff * inv_nW;
            if (in_w) {
                const float t = tmaxs[tid];
                lacc[0] = once * bce_term(p, t);
                lacc[4] = once * bce_term(pe, t);
                dw += bce_grad(p, t) * inv_wb;
            }
            if (hl.d_weak_out) hl.d_weak_out[b * NC + tid] = dw;
        } else {
            dw = d_weak[b * NC + tid];
        }
        const float dn = den[b * NC + tid];
        dnum[tid] = dw / dn;
        dden[tid] = -dw * weak[b * NC + tid] / dn;
    }
    // dW[o][f] accumulates over chunks in the MFMA accumulator of the wave that owns the (o, f) tile
    f32x4 wacc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) wacc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bacc = 0.f;    // thread o < NO
    for (int t0 = cy * HD_TC; t0 < T; t0 += NY * HD_TC) {
        __syncthreads();
        mk[tid] = heads_stage<HF, HD_SB>(h, xs, b, T, t0, use_drop, seed, thr, ks, tid);
        {   // softmax / sigmoid backward of the chunk's frames: 8 threads per frame, thread `sub` takes classes sub and sub + 8
            // (one thread per frame walking all classes left 7/8 of the workgroup idle for the longest serial stretch of
            // the kernel); the three per-frame reductions (max, sum of exponentials, dot) are DPP butterflies inside the
            // 8-lane group: quad_perm xor 1, xor 2, then row_half_mirror
            const int tl = tid >> 3, sub = tid & 7, t = t0 + tl;
            auto red8_sum = [](float v) {
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
                return v;
            };
            auto red8_max = [](float v) {
                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
                return v;
            };
            float* dlr = dl + tl * HD_SD;
            if (tl >= HD_TC) {
                // (HF = 512: only the first HD_TC * 8 threads own a frame of the chunk)
            } else if (t < T) {
                const size_t e0 = (size_t)(b * T + t) * NC;
                const bool has[2] = {sub < NC, sub + 8 < NC};
                float lg[2], ex[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) lg[q] = has[q] ? logits_s[e0 + sub + 8 * q] : -3.0e38f;
                const float mx = red8_max(fmaxf(lg[0], lg[1]));
#pragma unroll
                for (int q = 0; q < 2; ++q) ex[q] = has[q] ? __expf(lg[q] - mx) : 0.f;
                const float inv = rcp_fast(red8_sum(ex[0] + ex[1]));
                float raw[2], ds[2], dsig[2], dpart = 0.f;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    raw[q] = 0.f; ds[q] = 0.f; dsig[q] = 0.f;
                    if (has[q]) {
                        const int c = sub + 8 * q;
                        const float sv = strong[e0 + c];
                        raw[q] = ex[q] * inv;
                        const float sof = fminf(fmaxf(raw[q], 1e-7f), 1.0f);
                        const float pass = (raw[q] >= 1e-7f && raw[q] <= 1.0f) ? 1.f : 0.f;
                        ds[q] = (dnum[c] * sv + dden[c]) * pass;
                        dpart += raw[q] * ds[q];
                        float gin;
                        if (fused) {
                            const float pe = hl.strong_ema[e0 + c];
                            const float diff = sv - pe;
                            lacc[2] += diff * diff;
                            gin = cw * 2.0f * diff * inv_nS;
                            if (in_s) {
                                const float tg = hl.target[e0 + c];
                                lacc[1] += bce_term(sv, tg);
                                lacc[5] += bce_term(pe, tg);
                                gin += bce_grad(sv, tg) * inv_sb;
                            }
                            if (hl.d_strong_out) hl.d_strong_out[e0 + c] = gin;
                        } else {
                            gin = d_strong[e0 + c];
                        }
                        dsig[q] = (gin + dnum[c] * sof) * sv * (1.0f - sv);
                    }
                }
                const float dot = red8_sum(dpart);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (has[q]) {
                        dlr[sub + 8 * q] = dsig[q];
                        dlr[NC + sub + 8 * q] = raw[q] * (ds[q] - dot);
                    }
                for (int o = NO + sub; o < HD_MAXO; o += 8) dlr[o] = 0.f;
            } else {
                for (int o = sub; o < HD_MAXO; o += 8) dlr[o] = 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TPW; ++q) {   // dW[o][f] += sum_t dl[t][o] x[t][f]: 2 x HF / 16 tiles, TPW per wave, K = HD_TC frames
            const int tile = wv * TPW + q, ot = tile / HD_GPF, ft = tile % HD_GPF;
            const float* A = dl + kq * HD_SD + 16 * ot + i16;          // A[i = o][k = t]
            const float* Bp = xs + kq * HD_SB + 16 * ft + i16;         // B[k = t][j = f]
#pragma unroll 8
            for (int s4 = 0; s4 < HD_TC / 4; ++s4) wacc[q] = mfma16(A[4 * s4 * HD_SD], Bp[4 * s4 * HD_SB], wacc[q]);
        }
        if (tid < NO) {
            float a = 0.f;
            for (int tl = 0; tl < HD_TC; ++tl) a += dl[tl * HD_SD + tid];
            bacc += a;
        }
        // dx[t][f] = (sum_o dl[t][o] W[o][f]) * mask: (HD_TC / 16) x (HF / 16) = 64 tiles, four per wave, K = 32
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = wv * 4 + q, tt = tile / HD_GPF, ft = tile % HD_GPF;
            const float* A = dl + (16 * tt + i16) * HD_SD + kq;        // A[i = t][k = o]
            const float* Bp = wsm + kq * HD_SB + 16 * ft + i16;        // B[k = o][j = f]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) acc = mfma16(A[4 * s4], Bp[4 * s4 * HD_SB], acc);
            const int f = 16 * ft + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tl = 16 * tt + 4 * kq + r, t = t0 + tl;
                if (t < T) {
                    const uint32_t keep = (mk[tl * HD_GPF + (f >> 4)] >> (f & 15)) & 1u;
                    dh[(size_t)(b * T + t) * HD_F + f] = keep ? acc[r] * ks : 0.f;
                }
            }
        }
    }
    float* pr = part + (size_t)(b * NY + cy) * (2 * (NC * HD_F + NC));
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int tile = wv * TPW + q, ot = tile / HD_GPF, ft = tile % HD_GPF, f = 16 * ft + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * ot + 4 * kq + r;
            if (o < NC) pr[o * HD_F + f] = wacc[q][r];
            else if (o < NO) pr[NC * HD_F + NC + (o - NC) * HD_F + f] = wacc[q][r];
        }
    }
    if (tid < NC) pr[NC * HD_F + tid] = bacc;
    else if (tid < NO) pr[2 * NC * HD_F + NC + (tid - NC)] = bacc;
    if (!fused) return;
    // ---- the loss values: workgroup sums -> per-clip partials -> the last workgroup adds them in clip order ----
    __syncthreads();
    float* red = dl;                                    // [16 waves][8]
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float v = wave_sum(lacc[k]);
        if (lane == 0) red[wv * 8 + k] = v;
    }
    __syncthreads();
    const int NP = B * NY;                              // loss partials: one per workgroup, summed in (clip, chunk) order
    float* lpart = NY == 1 ? hl.losses + 8 : part + (size_t)NP * (2 * (NC * HD_F + NC));
    unsigned int* ticket = (unsigned int*)(hl.losses + 8 + 8 * B);
    __shared__ int is_last;
    if (tid < 6) {
        float s2 = 0.f;
        for (int w2 = 0; w2 < HD_THREADS / 64; ++w2) s2 += red[w2 * 8 + tid];
        lpart[8 * (b * NY + cy) + tid] = s2;
        __threadfence();
    }
    __syncthreads();
    if (tid == 0) is_last = (atomicAdd(ticket, 1u) == (unsigned int)(NP - 1));
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    {   // all per-clip partials in ONE round trip (a 6-thread loop over the clips paid one memory latency per clip), then
        // six threads add them up in clip order
        float* stage = dl + 128;                             // [8 * B] behind the wave partials (B <= 512; larger batches: loop)
        const bool staged = 8 * NP <= HD_TC * HD_SD - 128;
        if (staged)
            for (int e = tid; e < 8 * NP; e += HD_THREADS) stage[e] = __builtin_nontemporal_load(&lpart[e]);
        __syncthreads();
        if (tid < 6) {
            float s2 = 0.f;
            if (staged) for (int bb = 0; bb < NP; ++bb) s2 += stage[8 * bb + tid];
            else for (int bb = 0; bb < NP; ++bb) s2 += __builtin_nontemporal_load(&lpart[8 * bb + tid]);
            red[tid] = s2;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float wb = red[0] * inv_wb, sb = red[1] * inv_sb;
        const float cs = cw * red[2] * inv_nS, cwk = cw * red[3] * inv_nW;
        hl.losses[0] = wb + sb + cs + cwk;
        hl.losses[1] = wb; hl.losses[2] = sb; hl.losses[3] = cs; hl.losses[4] = cwk;
        hl.losses[5] = red[4] * inv_wb; hl.losses[6] = red[5] * inv_sb; hl.losses[7] = cw;
        *ticket = 0u;
        if (hl.advance) step_state_advance_early(hl.advance);   // every workgroup has read its fields of this step by now
    }
}

// ---- mean-teacher loss as a kernel of its own (sed_mt_loss) --------------------------------------------

#define LOSS_THREADS 1024
// One workgroup per clip (the single-workgroup version walked 19 elements per thread, one exposed memory round
// trip each: 20 us on the critical path between the forward and the backward).  Every workgroup writes its six
// partial sums to the scratch part of `losses`; the LAST one to finish (device-scope ticket) adds them up in clip
// order - the meters stay bit-reproducible - and re-arms the ticket for the next launch.
// losses layout: [0,8) meters | [8, 8+8B) per-clip partials | [8+8B] ticket (uint32, zero before first use)
__global__ __launch_bounds__(LOSS_THREADS) void k_mt_loss(const float* __restrict__ strong, const float* __restrict__ weak,
                                                  const float* __restrict__ strong_ema, const float* __restrict__ weak_ema,
                                                  const float* __restrict__ target, int B, int T, int NC, int wlo, int whi,
                                                  int slo, int shi, const sed_step_state* __restrict__ state,
                                                  float* __restrict__ losses, float* __restrict__ d_strong,
                                                  float* __restrict__ d_weak) {
    __shared__ float red[LOSS_THREADS / 64][8];
    __shared__ float tmax[16];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    const float cw = state->cons_weight;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // weak_bce, strong_bce, mse_strong, mse_weak, weak_ema_bce, strong_ema_bce
    const float inv_nS = 1.0f / (float)(B * T * NC), inv_nW = 1.0f / (float)(B * NC);
    const float inv_sb = (shi > slo) ? 1.0f / (float)((shi - slo) * T * NC) : 0.f;
    const float inv_wb = (whi > wlo) ? 1.0f / (float)((whi - wlo) * NC) : 0.f;
    const bool in_s = (b >= slo && b < shi), in_w = (b >= wlo && b < whi);
    const size_t base = (size_t)b * T * NC;
    if (in_w && wv < NC) {        // target_weak = target.max(-2) (main.py:95): wave c takes class c
        float t = -3.0e38f;
        for (int tt = lane; tt < T; tt += 64) t = fmaxf(t, target[base + (size_t)tt * NC + wv]);
        t = wave_max(t);
        if (lane == 0) tmax[wv] = t;
    }
    for (int e = tid; e < T * NC; e += LOSS_THREADS) {
        const float p = strong[base + e], pe = strong_ema[base + e];
        const float diff = p - pe;
        acc[2] += diff * diff;
        float g = cw * 2.0f * diff * inv_nS;
        if (in_s) {
            const float t = target[base + e];
            acc[1] += bce_term(p, t);
            acc[5] += bce_term(pe, t);
            g += bce_grad(p, t) * inv_sb;
        }
        d_strong[base + e] = g;
    }
    __syncthreads();
    if (tid < NC) {
        const int e = b * NC + tid;
        const float p = weak[e], pe = weak_ema[e];
        const float diff = p - pe;
        acc[3] += diff * diff;
        float g = cw * 2.0f * diff * inv_nW;
        if (in_w) {
            const float t = tmax[tid];
            acc[0] += bce_term(p, t);
            acc[4] += bce_term(pe, t);
            g += bce_grad(p, t) * inv_wb;
        }
        d_weak[e] = g;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    float* part = losses + 8;
    unsigned int* ticket = (unsigned int*)(losses + 8 + 8 * B);
    if (tid < 6) {
        float s2 = 0.f;
        for (int w2 = 0; w2 < LOSS_THREADS / 64; ++w2) s2 += red[w2][tid];
        part[8 * b + tid] = s2;
        __threadfence();
    }
    __syncthreads();
    if (tid == 0) is_last = (atomicAdd(ticket, 1u) == (unsigned int)(B - 1));
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (tid < 6) {
        float s2 = 0.f;
        for (int bb = 0; bb < B; ++bb) s2 += __builtin_nontemporal_load(&part[8 * bb + tid]);
        red[0][tid] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        const float* s6 = red[0];
        const float wb = s6[0] * inv_wb, sb = s6[1] * inv_sb;
        const float cs = cw * s6[2] * inv_nS, cwk = cw * s6[3] * inv_nW;
        losses[0] = wb + sb + cs + cwk;
        losses[1] = wb; losses[2] = sb; losses[3] = cs; losses[4] = cwk;
        losses[5] = s6[4] * inv_wb; losses[6] = s6[5] * inv_sb; losses[7] = cw;
        *ticket = 0u;
    }
}

// ---- the tail of the fused form (hfuse.h): what k_heads_bwd's last workgroup and k_colsum do in the two-kernel form --------
// blocks [0, gridDim.x - 1): out[n] = sum_b part[b * N + n], four row groups as k_colsum (bit-identical sums);
// last block: the six loss sums in clip order -> the meters; step-state advance (every reader of this step's forward / loss
// fields ran in kernels that precede this one in stream order; the update runs after the side stream's join).
__global__ __launch_bounds__(256) void k_heads_fin(const float* __restrict__ part, int B, int N, float* __restrict__ out, HeadsLoss hl,
                                                   int T, int NC) {
    __shared__ float red[4][64];
    if (blockIdx.x + 1 < gridDim.x) {
        const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
        const int col = blockIdx.x * 64 + c;
        float s = 0.f;
        if (col < N)
            for (int m = r; m < B; m += 4) s += part[(int64_t)m * N + col];
        red[r][c] = s;
        __syncthreads();
        if (r == 0 && col < N) out[col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
        return;
    }
    const int tid = threadIdx.x;
    const float* lpart = hl.losses + 8;
    if (tid < 6) {
        float s2 = 0.f;
        for (int bb = 0; bb < B; ++bb) s2 += lpart[8 * bb + tid];
        red[0][tid] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        const float cw = hl.state->cons_weight;
        const float inv_nS = 1.0f / (float)(B * T * NC), inv_nW = 1.0f / (float)(B * NC);
        const float inv_sb = (hl.shi > hl.slo) ? 1.0f / (float)((hl.shi - hl.slo) * T * NC) : 0.f;
        const float inv_wb = (hl.whi > hl.wlo) ? 1.0f / (float)((hl.whi - hl.wlo) * NC) : 0.f;
        const float* s6 = red[0];
        const float wb = s6[0] * inv_wb, sb = s6[1] * inv_sb;
        const float cs = cw * s6[2] * inv_nS, cwk = cw * s6[3] * inv_nW;
        hl.losses[0] = wb + sb + cs + cwk;
        hl.losses[1] = wb; hl.losses[2] = sb; hl.losses[3] = cs; hl.losses[4] = cwk;
        hl.losses[5] = s6[4] * inv_wb; hl.losses[6] = s6[5] * inv_sb; hl.losses[7] = cw;
        if (hl.advance) step_state_advance_early(hl.advance);
    }
}
int launch_heads_fin(const float* part, float* g_wd, int B, int T, int NC, int n_cols, const HeadsLoss& hl, hipStream_t st) {
    k_heads_fin<<<(n_cols + 63) / 64 + 1, 256, 0, st>>>(part, B, n_cols, g_wd, hl, T, NC);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// ---- host launchers -------------------------------------------------------------------------------
template <int HF>
static int heads_fwd_launch(const float* h, const float* wd, const float* bd, const float* ws, const float* bs, float* strong,
                            float* weak, float* strong_sv, float* weak_sv, float* logits_s, float* den, int B, int T, int NC,
                            int use_drop, float p_drop, const uint64_t* seed, hipStream_t st) {
    HD_CONSTS(HF);
    const size_t lds = (size_t)(HD_TC * HD_SF + HD_MAXO * HD_SF + HD_TC * HD_MAXO + 2 * HD_TC * 16 + 32) * sizeof(float);
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_heads_fwd<HF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    k_heads_fwd<HF><<<B, HD_THREADS, lds, st>>>(h, wd, bd, ws, bs, strong, weak, strong_sv, weak_sv, logits_s, den, T, NC, use_drop,
                                                p_drop, seed);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
int launch_heads_fwd(const float* h, const float* wd, const float* bd, const float* ws, const float* bs, float* strong,
                     float* weak, float* strong_sv, float* weak_sv, float* logits_s, float* den, int B, int T, int NC,
                     int use_drop, float p_drop, const uint64_t* seed, hipStream_t st, int HF) {
    if (HF == 128) return heads_fwd_launch<128>(h, wd, bd, ws, bs, strong, weak, strong_sv, weak_sv, logits_s, den, B, T, NC, use_drop, p_drop, seed, st);
    if (HF == 512) return heads_fwd_launch<512>(h, wd, bd, ws, bs, strong, weak, strong_sv, weak_sv, logits_s, den, B, T, NC, use_drop, p_drop, seed, st);
    sed_set_error("heads: unsupported feature width %d (2 x n_RNN_cell must be 128 or 512)", HF);
    return SED_ERR_UNSUPPORTED;
}

// B = number of partial slabs (clips x heads_bwd_chunks)
int launch_heads_colsum(const float* part, float* g_wd, int B, int NC, hipStream_t st, int HF) {
    // dense.weight, dense.bias, dense_softmax.weight, dense_softmax.bias are contiguous in the flat layout
    return launch_colsum(part, B, 2 * (NC * HF + NC), 2 * (NC * HF + NC), g_wd, st);
}

template <int HF>
static int heads_bwd_launch(const float* h, const float* wd, const float* ws, const float* strong, const float* weak,
                            const float* logits_s, const float* den, const float* d_strong, const float* d_weak, float* dh,
                            float* part, int B, int T, int NC, int use_drop, float p_drop, const uint64_t* seed, double* zero,
                            int n_zero, const HeadsLoss& hl0, hipStream_t st) {
    HD_CONSTS(HF);
    const size_t lds = (size_t)(HD_TC * HD_SB + HD_MAXO * HD_SB + HD_TC * HD_SD + 32 + HD_THREADS) * sizeof(float);
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_heads_bwd<HF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    k_heads_bwd<HF><<<dim3(B, heads_bwd_chunks(HF, T)), HD_THREADS, lds, st>>>(h, wd, ws, strong, weak, logits_s, den, d_strong, d_weak, dh,
                                                                               part, T, NC, use_drop, p_drop, seed, zero, zero ? n_zero : 0, hl0);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// defer_colsum: the caller sums the per-clip partial weight gradients later (launch_heads_colsum, off the critical path)
int launch_heads_bwd(const float* h, const float* wd, const float* ws, const float* strong, const float* weak,
                     const float* logits_s, const float* den, const float* d_strong, const float* d_weak, float* dh,
                     float* part, float* g_wd, float* g_bd, float* g_ws, float* g_bs, int B, int T, int NC, int use_drop,
                     float p_drop, const uint64_t* seed, double* zero, int n_zero, int defer_colsum, const HeadsLoss* hl,
                     hipStream_t st, int HF) {
    (void)g_bd; (void)g_ws; (void)g_bs;
    HeadsLoss hl0 = {};
    if (hl) hl0 = *hl;
    int rc;
    if (HF == 128) rc = heads_bwd_launch<128>(h, wd, ws, strong, weak, logits_s, den, d_strong, d_weak, dh, part, B, T, NC, use_drop, p_drop, seed, zero, n_zero, hl0, st);
    else if (HF == 512) rc = heads_bwd_launch<512>(h, wd, ws, strong, weak, logits_s, den, d_strong, d_weak, dh, part, B, T, NC, use_drop, p_drop, seed, zero, n_zero, hl0, st);
    else {
        sed_set_error("heads: unsupported feature width %d (2 x n_RNN_cell must be 128 or 512)", HF);
        return SED_ERR_UNSUPPORTED;
    }
    if (rc != SED_OK || defer_colsum) return rc;
    return launch_heads_colsum(part, g_wd, B * heads_bwd_chunks(HF, T), NC, st, HF);
}

extern "C" int sed_mt_loss(const sed_dims* d, const float* strong, const float* weak, const float* strong_ema,
                           const float* weak_ema, const float* target, int weak_lo, int weak_hi, int strong_lo,
                           int strong_hi, const sed_step_state* state_dev, float* losses, float* d_strong, float* d_weak,
                           void* stream) {
    SED_CHECK_ARG(d && strong && weak && strong_ema && weak_ema && target && state_dev && losses && d_strong && d_weak,
                  "sed_mt_loss: null argument");
    SED_TRY(sed_validate_dims(d));
    const Geo g = make_geo(d);
    SED_CHECK_ARG(weak_lo >= 0 && weak_hi <= g.B && weak_lo <= weak_hi && strong_lo >= 0 && strong_hi <= g.B &&
                      strong_lo <= strong_hi, "sed_mt_loss: bad mask range");
    k_mt_loss<<<g.B, LOSS_THREADS, 0, (hipStream_t)stream>>>(strong, weak, strong_ema, weak_ema, target, g.B, g.T3, g.NC, weak_lo,
                                                  weak_hi, strong_lo, strong_hi, state_dev, losses, d_strong, d_weak);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
