// heads.hip - output heads of the CRNN and the mean-teacher loss.
//
// Reference ops (baseline/models/CRNN.py:74-81):
//   x = Dropout(p)(gru_out)
//   strong = sigmoid(dense(x))                                   [B, T', nclass]
//   sof    = clamp(softmax(dense_softmax(x), dim=-1), 1e-7, 1)   (softmax over CLASSES)
//   weak   = (strong * sof).sum(1) / sof.sum(1)                  [B, nclass]
// and the loss block of main.train (baseline/main.py:93-145).  One workgroup per clip; the work is
// tiny (0.4 MFLOP / clip) so everything is fused: dropout mask (Philox, recomputed in backward),
// both Linear layers, sigmoid, softmax, clamp and the attention pooling.
#include "common.h"
#include "philox.h"
#include "kernels.h"

#define HD_TC 128     // frames per chunk (one chunk covers T/8 <= 128, i.e. clips up to 1024 frames)
#define HD_THREADS 1024
#define HD_F 128      // 2 * hidden
#define HD_FS 129     // padded row stride
#define HD_MAXO 32    // 2 * max nclass

__device__ __forceinline__ float rnn_drop(float v, int use_drop, size_t e, uint64_t seed, uint32_t thr, float ks) {
    if (!use_drop) return v;
    const u32x4 o = philox_stream((uint32_t)(e >> 4), 8u, seed);
    return (philox_byte(o, (int)(e & 15)) >= thr) ? v * ks : 0.f;
}

// loads a chunk of frames (dropout applied) and both weight matrices into LDS
__device__ __forceinline__ void heads_stage(const float* __restrict__ h, float* xs, int b, int T, int t0, int use_drop,
                                            uint64_t seed, uint32_t thr, float ks, int tid) {
    for (int e = tid; e < HD_TC * HD_F; e += HD_THREADS) {
        const int tl = e >> 7, f = e & 127, t = t0 + tl;
        float v = 0.f;
        if (t < T) {
            const size_t ge = (size_t)(b * T + t) * HD_F + f;
            v = rnn_drop(h[ge], use_drop, ge, seed, thr, ks);
        }
        xs[tl * HD_FS + f] = v;
    }
}

__global__ __launch_bounds__(HD_THREADS) void k_heads_fwd(const float* __restrict__ h, const float* __restrict__ wd,
                                                    const float* __restrict__ bd, const float* __restrict__ ws,
                                                    const float* __restrict__ bs, float* __restrict__ strong,
                                                    float* __restrict__ weak, float* __restrict__ strong_sv,
                                                    float* __restrict__ weak_sv, float* __restrict__ logits_s,
                                                    float* __restrict__ den_out, int T, int NC, int use_drop, float p_drop,
                                                    const uint64_t* __restrict__ seed_ptr) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* xs = hsm;                                   // [HD_TC][HD_FS]
    float* wsm = xs + HD_TC * HD_FS;                   // [HD_MAXO][HD_FS]
    float* lg = wsm + HD_MAXO * HD_FS;                 // [HD_TC][HD_MAXO]
    float (*nums)[16] = (float (*)[16])(lg + HD_TC * HD_MAXO);
    float (*dens)[16] = (float (*)[16])(lg + HD_TC * HD_MAXO + HD_TC * 16);
    float* num_acc = lg + HD_TC * HD_MAXO + 2 * HD_TC * 16;
    float* den_acc = num_acc + 16;
    const int tid = threadIdx.x, b = blockIdx.x;
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float ks = use_drop ? drop_scale8(p_drop) : 1.0f;
    const int NO = 2 * NC;
    for (int e = tid; e < NO * HD_F; e += HD_THREADS) {
        const int o = e >> 7, f = e & 127;
        wsm[o * HD_FS + f] = (o < NC) ? wd[o * HD_F + f] : ws[(o - NC) * HD_F + f];
    }
    if (tid < 16) { num_acc[tid] = 0.f; den_acc[tid] = 0.f; }
    for (int t0 = 0; t0 < T; t0 += HD_TC) {
        __syncthreads();
        heads_stage(h, xs, b, T, t0, use_drop, seed, thr, ks, tid);
        __syncthreads();
        for (int e = tid; e < HD_TC * NO; e += HD_THREADS) {
            const int tl = e / NO, o = e % NO;
            float a = (o < NC) ? bd[o] : bs[o - NC];
            const float* xr = xs + tl * HD_FS;
            const float* wr = wsm + o * HD_FS;
#pragma unroll 8
            for (int f = 0; f < HD_F; ++f) a = fmaf(xr[f], wr[f], a);
            lg[tl * HD_MAXO + o] = a;
        }
        __syncthreads();
        if (tid < HD_TC) {
            const int t = t0 + tid;
            if (t < T) {
                float mx = -3.0e38f;
                for (int c = 0; c < NC; ++c) mx = fmaxf(mx, lg[tid * HD_MAXO + NC + c]);
                float se = 0.f;
                for (int c = 0; c < NC; ++c) se += __expf(lg[tid * HD_MAXO + NC + c] - mx);
                const float inv = rcp_fast(se);
                for (int c = 0; c < NC; ++c) {
                    const float ls = lg[tid * HD_MAXO + NC + c];
                    float sof = __expf(ls - mx) * inv;
                    sof = fminf(fmaxf(sof, 1e-7f), 1.0f);
                    const float sv = sigmoidf_fast(lg[tid * HD_MAXO + c]);
                    strong[(size_t)(b * T + t) * NC + c] = sv;
                    if (strong_sv) strong_sv[(size_t)(b * T + t) * NC + c] = sv;
                    logits_s[(size_t)(b * T + t) * NC + c] = ls;
                    nums[tid][c] = sv * sof;
                    dens[tid][c] = sof;
                }
            } else {
                for (int c = 0; c < NC; ++c) { nums[tid][c] = 0.f; dens[tid][c] = 0.f; }
            }
        }
        __syncthreads();
        if (tid < NC) {
            float a = 0.f, d2 = 0.f;
            for (int tl = 0; tl < HD_TC; ++tl) { a += nums[tl][tid]; d2 += dens[tl][tid]; }
            num_acc[tid] += a; den_acc[tid] += d2;
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

// part row layout (matches the flat parameter order dense.weight, dense.bias, dense_softmax.weight,
// dense_softmax.bias): [NC*128 dWd][NC dbd][NC*128 dWs][NC dbs]
__global__ __launch_bounds__(HD_THREADS) void k_heads_bwd(const float* __restrict__ h, const float* __restrict__ wd,
                                                    const float* __restrict__ ws, const float* __restrict__ strong,
                                                    const float* __restrict__ weak, const float* __restrict__ logits_s,
                                                    const float* __restrict__ den, const float* __restrict__ d_strong,
                                                    const float* __restrict__ d_weak, float* __restrict__ dh,
                                                    float* __restrict__ part, int T, int NC, int use_drop, float p_drop,
                                                    const uint64_t* __restrict__ seed_ptr) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* xs = hsm;                                   // [HD_TC][HD_FS]
    float* wsm = xs + HD_TC * HD_FS;                   // [HD_MAXO][HD_FS]
    float* dl = wsm + HD_MAXO * HD_FS;                 // [HD_TC][HD_MAXO]
    float* dnum = dl + HD_TC * HD_MAXO;
    float* dden = dnum + 16;
    const int tid = threadIdx.x, b = blockIdx.x;
    const uint64_t seed = use_drop ? seed_ptr[0] : 0ull;
    const uint32_t thr = drop_thresh8(p_drop);
    const float ks = use_drop ? drop_scale8(p_drop) : 1.0f;
    const int NO = 2 * NC;
    for (int e = tid; e < NO * HD_F; e += HD_THREADS) {
        const int o = e >> 7, f = e & 127;
        wsm[o * HD_FS + f] = (o < NC) ? wd[o * HD_F + f] : ws[(o - NC) * HD_F + f];
    }
    if (tid < NC) {
        const float dw = d_weak[b * NC + tid], dn = den[b * NC + tid];
        dnum[tid] = dw / dn;
        dden[tid] = -dw * weak[b * NC + tid] / dn;
    }
    // weight-gradient accumulators: outputs e = tid + 256*i over [NO][128]
    float wacc[(HD_MAXO * HD_F) / HD_THREADS];
#pragma unroll
    for (int i = 0; i < (HD_MAXO * HD_F) / HD_THREADS; ++i) wacc[i] = 0.f;
    float bacc = 0.f;    // thread o < NO
    for (int t0 = 0; t0 < T; t0 += HD_TC) {
        __syncthreads();
        heads_stage(h, xs, b, T, t0, use_drop, seed, thr, ks, tid);
        if (tid < HD_TC) {
            const int t = t0 + tid;
            if (t < T) {
                float mx = -3.0e38f;
                for (int c = 0; c < NC; ++c) mx = fmaxf(mx, logits_s[(size_t)(b * T + t) * NC + c]);
                float se = 0.f;
                for (int c = 0; c < NC; ++c) se += __expf(logits_s[(size_t)(b * T + t) * NC + c] - mx);
                const float inv = rcp_fast(se);
                float dot = 0.f;
                float sraw[16], dsof[16];
                for (int c = 0; c < NC; ++c) {
                    const float sv = strong[(size_t)(b * T + t) * NC + c];
                    const float raw = __expf(logits_s[(size_t)(b * T + t) * NC + c] - mx) * inv;
                    const float sof = fminf(fmaxf(raw, 1e-7f), 1.0f);
                    const float pass = (raw >= 1e-7f && raw <= 1.0f) ? 1.f : 0.f;
                    const float ds = (dnum[c] * sv + dden[c]) * pass;
                    sraw[c] = raw; dsof[c] = ds;
                    dot += raw * ds;
                    const float dst = d_strong[(size_t)(b * T + t) * NC + c] + dnum[c] * sof;
                    dl[tid * HD_MAXO + c] = dst * sv * (1.0f - sv);
                }
                for (int c = 0; c < NC; ++c) dl[tid * HD_MAXO + NC + c] = sraw[c] * (dsof[c] - dot);
            } else {
                for (int o = 0; o < NO; ++o) dl[tid * HD_MAXO + o] = 0.f;
            }
        }
        __syncthreads();
        // dW[o][f] += sum_t dl[t][o] * x[t][f]
#pragma unroll
        for (int i = 0; i < (HD_MAXO * HD_F) / HD_THREADS; ++i) {
            const int e = tid + HD_THREADS * i, o = e >> 7, f = e & 127;
            if (o < NO) {
                float a = 0.f;
                for (int tl = 0; tl < HD_TC; ++tl) a = fmaf(dl[tl * HD_MAXO + o], xs[tl * HD_FS + f], a);
                wacc[i] += a;
            }
        }
        if (tid < NO) {
            float a = 0.f;
            for (int tl = 0; tl < HD_TC; ++tl) a += dl[tl * HD_MAXO + tid];
            bacc += a;
        }
        // dx[t][f] = (sum_o dl[t][o] W[o][f]) * mask
        for (int e = tid; e < HD_TC * HD_F; e += HD_THREADS) {
            const int tl = e >> 7, f = e & 127, t = t0 + tl;
            if (t < T) {
                float a = 0.f;
                for (int o = 0; o < NO; ++o) a = fmaf(dl[tl * HD_MAXO + o], wsm[o * HD_FS + f], a);
                const size_t ge = (size_t)(b * T + t) * HD_F + f;
                dh[ge] = rnn_drop(a, use_drop, ge, seed, thr, ks);
            }
        }
    }
    float* pr = part + (size_t)b * (2 * (NC * HD_F + NC));
#pragma unroll
    for (int i = 0; i < (HD_MAXO * HD_F) / HD_THREADS; ++i) {
        const int e = tid + HD_THREADS * i, o = e >> 7, f = e & 127;
        if (o < NC) pr[o * HD_F + f] = wacc[i];
        else if (o < NO) pr[NC * HD_F + NC + (o - NC) * HD_F + f] = wacc[i];
    }
    if (tid < NC) pr[NC * HD_F + tid] = bacc;
    else if (tid < NO) pr[2 * NC * HD_F + NC + (tid - NC)] = bacc;
}

// ---- mean-teacher loss (main.py:93-145) -----------------------------------------------------------
__device__ __forceinline__ float bce_term(float p, float t) {
    const float lp = fmaxf(logf(p), -100.0f), l1p = fmaxf(logf(1.0f - p), -100.0f);
    return -(t * lp + (1.0f - t) * l1p);
}
__device__ __forceinline__ float bce_grad(float p, float t) { return (p - t) / fmaxf((1.0f - p) * p, 1e-12f); }

#define LOSS_THREADS 1024
__global__ __launch_bounds__(LOSS_THREADS) void k_mt_loss(const float* __restrict__ strong, const float* __restrict__ weak,
                                                  const float* __restrict__ strong_ema, const float* __restrict__ weak_ema,
                                                  const float* __restrict__ target, int B, int T, int NC, int wlo, int whi,
                                                  int slo, int shi, const sed_step_state* __restrict__ state,
                                                  float* __restrict__ losses, float* __restrict__ d_strong,
                                                  float* __restrict__ d_weak) {
    __shared__ float red[LOSS_THREADS / 64][8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float cw = state->cons_weight;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // weak_bce, strong_bce, mse_strong, mse_weak, weak_ema_bce, strong_ema_bce
    const int nS = B * T * NC, nW = B * NC;
    const float inv_nS = 1.0f / (float)nS, inv_nW = 1.0f / (float)nW;
    const float inv_sb = (shi > slo) ? 1.0f / (float)((shi - slo) * T * NC) : 0.f;
    const float inv_wb = (whi > wlo) ? 1.0f / (float)((whi - wlo) * NC) : 0.f;
    for (int e = tid; e < nS; e += LOSS_THREADS) {
        const int b = e / (T * NC);
        const float p = strong[e], pe = strong_ema[e];
        const float diff = p - pe;
        acc[2] += diff * diff;
        float g = cw * 2.0f * diff * inv_nS;
        if (b >= slo && b < shi) {
            const float t = target[e];
            acc[1] += bce_term(p, t);
            acc[5] += bce_term(pe, t);
            g += bce_grad(p, t) * inv_sb;
        }
        d_strong[e] = g;
    }
    for (int e = tid; e < nW; e += LOSS_THREADS) {
        const int b = e / NC, c = e % NC;
        const float p = weak[e], pe = weak_ema[e];
        const float diff = p - pe;
        acc[3] += diff * diff;
        float g = cw * 2.0f * diff * inv_nW;
        if (b >= wlo && b < whi) {
            float t = -3.0e38f;       // target_weak = target.max(-2)  (main.py:95)
            int tt = 0;
            for (; tt + 16 <= T; tt += 16) {        // 16 independent loads in flight (a rolled loop waits per load)
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = target[(size_t)(b * T + tt + i) * NC + c];
#pragma unroll
                for (int i = 0; i < 16; ++i) t = fmaxf(t, v[i]);
            }
            for (; tt < T; ++tt) t = fmaxf(t, target[(size_t)(b * T + tt) * NC + c]);
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
    if (tid == 0) {
        float s[6];
        for (int k = 0; k < 6; ++k) {
            s[k] = 0.f;
            for (int w2 = 0; w2 < LOSS_THREADS / 64; ++w2) s[k] += red[w2][k];
        }
        const float wb = s[0] * inv_wb, sb = s[1] * inv_sb;
        const float cs = cw * s[2] * inv_nS, cwk = cw * s[3] * inv_nW;
        losses[0] = wb + sb + cs + cwk;
        losses[1] = wb; losses[2] = sb; losses[3] = cs; losses[4] = cwk;
        losses[5] = s[4] * inv_wb; losses[6] = s[5] * inv_sb; losses[7] = cw;
    }
}

// ---- host launchers -------------------------------------------------------------------------------
int launch_heads_fwd(const float* h, const float* wd, const float* bd, const float* ws, const float* bs, float* strong,
                     float* weak, float* strong_sv, float* weak_sv, float* logits_s, float* den, int B, int T, int NC,
                     int use_drop, float p_drop, const uint64_t* seed, hipStream_t st) {
    const size_t lds = (size_t)(HD_TC * HD_FS + HD_MAXO * HD_FS + HD_TC * HD_MAXO + 2 * HD_TC * 16 + 32) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_heads_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    k_heads_fwd<<<B, HD_THREADS, lds, st>>>(h, wd, bd, ws, bs, strong, weak, strong_sv, weak_sv, logits_s, den, T, NC, use_drop,
                                            p_drop, seed);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_heads_bwd(const float* h, const float* wd, const float* ws, const float* strong, const float* weak,
                     const float* logits_s, const float* den, const float* d_strong, const float* d_weak, float* dh,
                     float* part, float* g_wd, float* g_bd, float* g_ws, float* g_bs, int B, int T, int NC, int use_drop,
                     float p_drop, const uint64_t* seed, hipStream_t st) {
    (void)g_bd; (void)g_ws; (void)g_bs;
    const size_t lds = (size_t)(HD_TC * HD_FS + HD_MAXO * HD_FS + HD_TC * HD_MAXO + 32) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_heads_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    k_heads_bwd<<<B, HD_THREADS, lds, st>>>(h, wd, ws, strong, weak, logits_s, den, d_strong, d_weak, dh, part, T, NC, use_drop,
                                            p_drop, seed);
    SED_CHECK_LAUNCH();
    // dense.weight, dense.bias, dense_softmax.weight, dense_softmax.bias are contiguous in the flat layout
    return launch_colsum(part, B, 2 * (NC * HD_F + NC), 2 * (NC * HD_F + NC), g_wd, st);
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
    k_mt_loss<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(strong, weak, strong_ema, weak_ema, target, g.B, g.T3, g.NC, weak_lo,
                                                  weak_hi, strong_lo, strong_hi, state_dev, losses, d_strong, d_weak);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
