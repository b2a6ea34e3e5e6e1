// hfuse.h - the output heads of the CRNN (forward), the mean-teacher loss and the heads' backward as a PROLOGUE PHASE of the
// top GRU layer's backward-recurrence kernel (gru4.hip k_gru4_bwd<NIN, true>), round 5.
//
// Reference ops: baseline/models/CRNN.py:74-81 (dropout, dense + sigmoid, dense_softmax + softmax + clamp, attention
// pooling) and the loss block of baseline/main.py:93-145 - the same arithmetic as heads.hip's k_heads_fwd / k_heads_bwd, in the
// same order (the posteriors, the gradient w.r.t. the GRU output and the per-clip weight-gradient partials come out
// bit-identical to the two-kernel path; tests/test_gpu_parity.py asserts it).
//
// Why: on the critical chain of every step the student's recurrence forward was followed by k_heads_fwd (8 us), a cross-queue
// join with the teacher (10 us idle), k_heads_bwd (22 - 24 us: a chain of ~5 dependent memory round trips on B workgroups)
// and only then the backward recurrence - 40 us in which 24 of 256 CUs do 0.8 MFLOP per clip.  All of it is local to a clip
// once the teacher's posteriors exist: the two workgroups (clip, direction) of the backward-recurrence launch each redo the
// clip's heads forward (K = 128 logits GEMM, 5 x 2 MFMA tiles), form the loss gradient, and produce THEIR 64-column half of
// dL/dh straight into LDS, where the recurrence's I/O waves pick it up - d_out never goes through HBM, and the posteriors
// `strong` / `weak` are written on the way (by the direction-0 workgroup).  The loss meters' cross-clip sums, the step-state
// advance and the column sum of the per-clip weight-gradient partials move to k_heads_fin on the weight-gradient side stream
// (no device-scope ticket / fence in front of the recurrence).
#pragma once
#include "common.h"
#include "philox.h"
#include "kernels.h"

#define HF_S 132          // row stride of the staged GRU output and of the staged weights (4 mod 64: conflict-free A fragments)
#define HF_SD 33          // row stride of the logits / dlogits tile
#define HF_MAXO 32        // 2 x max nclass
#define HF_TMAX 128       // frames (T / 8) a fused launch supports: one 128-frame chunk, as k_heads_* use per pass
#define HF_MISC 192

// floats of LDS scratch the phase needs for T frames (aliased over the recurrence's ops / history rings by the caller)
static inline int hfuse_scratch_floats(int T) {
    const int TP = (T + 15) & ~15;
    return TP * HF_S + HF_MAXO * HF_S + TP * HF_SD + TP * 8 + HF_MISC;
}

#ifdef __HIPCC__
__device__ __forceinline__ float bce_term(float p, float t) {
    const float lp = fmaxf(logf(p), -100.0f), l1p = fmaxf(logf(1.0f - p), -100.0f);
    return -(t * lp + (1.0f - t) * l1p);
}
__device__ __forceinline__ float bce_grad(float p, float t) { return (p - t) / fmaxf((1.0f - p) * p, 1e-12f); }

__device__ __forceinline__ float hf_red8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float hf_red8_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    return v;
}
__device__ __forceinline__ f32x4 hf_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One workgroup (clip b, direction dir) of NT threads; h = the top GRU layer's output [B][T][128] (H = 64).
//   R       scratch, hfuse_scratch_floats(T) floats
//   dout_s  [TP][64] result: this direction's half of dL/dh (dropout mask and scale applied); used as scratch before
// Ends with a workgroup barrier: on return dout_s is complete and R is free.  Every barrier of the phase is LDS-only
// (lds_barrier): __syncthreads() would also drain the wave's outstanding global stores (the posteriors, the partials) and the
// caller's prefetch loads at every phase boundary.
template <int NT>
__device__ __forceinline__ void heads_fused_phase(const HeadsFuse& hf, const float* __restrict__ h, int b, int dir, int nwg, int wg,
                                                  int T, float* __restrict__ R, float* __restrict__ dout_s) {
    constexpr int NW = NT / 64;
    constexpr int TRIPS = (HF_TMAX * 8 + NT - 1) / NT;          // (frame, 8-lane group) items per thread
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const int TP = (T + 15) & ~15, NC = hf.NC, NO = 2 * NC;
    float* xs = R;                                   // [TP][HF_S]   dropped GRU output
    float* wsm = xs + TP * HF_S;                     // [32][HF_S]   rows [0, NC) dense, [NC, 2 NC) dense_softmax, rest zero
    float* lg = wsm + HF_MAXO * HF_S;                // [TP][HF_SD]  logits, later dlogits
    uint32_t* mk = (uint32_t*)(lg + TP * HF_SD);     // [TP][8]      keep bits per (frame, 16 features)
    float* misc = (float*)(mk + TP * 8);
    float* dnum = misc, *dden = misc + 16, *tmaxs = misc + 32, *lw = misc + 48 /* [16][3] */, *red = misc + 96 /* [NW][8] */;
    float (*nums)[16] = (float (*)[16])dout_s;       // [TP][16]  (dead before dout_s is written)
    float (*dens)[16] = (float (*)[16])(dout_s + TP * 16);
    const HeadsLoss& hl = hf.hl;
    const uint64_t seed = hf.use_drop ? hf.seed[0] : 0ull;
    const uint32_t thr = drop_thresh8(hf.p_drop);
    const float ks = hf.use_drop ? drop_scale8(hf.p_drop) : 1.0f;
    // the fp64 accumulators of the conv-block backward that follows (saves a memset node on the critical path)
    for (int i = wg * NT + tid; i < hf.n_zero; i += nwg * NT) hf.zero[i] = 0.0;
    const int B = nwg / 2;
    const float cw = hl.state->cons_weight;
    const float inv_nS = 1.0f / (float)(B * T * NC), inv_nW = 1.0f / (float)(B * NC);
    const float inv_sb = (hl.shi > hl.slo) ? 1.0f / (float)((hl.shi - hl.slo) * T * NC) : 0.f;
    const float inv_wb = (hl.whi > hl.wlo) ? 1.0f / (float)((hl.whi - hl.wlo) * NC) : 0.f;
    // (wave-uniform scalars that live through the whole phase: kept in VGPRs - next to the recurrence kernel's own dozen
    // pointers they cost SGPR spills otherwise, and the phase has vector registers to spare)
    float cw_v = cw, inv_nS_v = inv_nS, inv_nW_v = inv_nW, inv_sb_v = inv_sb, inv_wb_v = inv_wb, ks_v = ks;
    asm volatile("" : "+v"(cw_v), "+v"(inv_nS_v), "+v"(inv_nW_v), "+v"(inv_sb_v), "+v"(inv_wb_v), "+v"(ks_v));
    float* pr = hf.part + (size_t)b * (2 * (NC * 128 + NC));       // (the two pointers of the last phase: the same, as addresses)
    float* lossp = hl.losses + 8 + 8 * b;
    asm volatile("" : "+v"(pr), "+v"(lossp));
    const bool in_s = b >= hl.slo && b < hl.shi, in_w = b >= hl.wlo && b < hl.whi;
    // supervised (main_simple_CRNN.py): the "teacher" posteriors ARE this kernel's own outputs - both consistency terms are zero
    const bool self_t = (hl.strong_ema == hf.strong);
    const bool store = (dir == 0);
    // ---- loads that are only needed two phases on: the teacher's posteriors and the targets of this thread's (frame, classes) ----
    float pe_r[TRIPS][2], tg_r[TRIPS][2];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
        const int it = tid + NT * k, tl = it >> 3, sub = it & 7;
        const size_t e0 = (size_t)(b * T + min(tl, T - 1)) * NC;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = min(sub + 8 * q, NC - 1);
            pe_r[k][q] = self_t ? 0.f : hl.strong_ema[e0 + c];
            tg_r[k][q] = in_s ? hl.target[e0 + c] : 0.f;
        }
    }
    // ---- H0: weights and the dropped GRU output into LDS ----------------------------------------------------------------
    {
        constexpr int N4 = HF_MAXO * 128 / 4, WT = (N4 + NT - 1) / NT;
        f32x4 v[WT];
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int e4 = tid + NT * i, o = (4 * e4) / 128, f = (4 * e4) % 128;
            const int oc = o < NO ? o : NO - 1;
            // dense.weight | dense.bias | dense_softmax.weight | dense_softmax.bias are contiguous in the flat parameter
            // layout (one base pointer: the kernel is short of scalar registers, not of adds)
            const float* row = hf.wd + (size_t)oc * 128 + (oc < NC ? 0 : NC);
            v[i] = *(const f32x4*)(row + (e4 < N4 ? f : 0));
        }
        float xv[TRIPS][16];
        uint32_t keep[TRIPS];
#pragma unroll
        for (int k = 0; k < TRIPS; ++k) {
            const int it = tid + NT * k, tl = it >> 3, f0 = (it & 7) * 16;
            keep[k] = 0;
            if (tl < T) {
                const size_t ge = (size_t)(b * T + tl) * 128 + f0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 u = *(const float4*)(h + ge + 4 * q);
                    xv[k][4 * q] = u.x; xv[k][4 * q + 1] = u.y; xv[k][4 * q + 2] = u.z; xv[k][4 * q + 3] = u.w;
                }
                keep[k] = 0xffffu;
                if (hf.use_drop) keep[k] = philox_keep16(philox_stream((uint32_t)(ge >> 4), 8u, seed), thr);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) xv[k][i] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int e4 = tid + NT * i, o = (4 * e4) / 128, f = (4 * e4) % 128;
            if (e4 < N4) *(f32x4*)&wsm[o * HF_S + f] = o < NO ? v[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < TRIPS; ++k) {
            const int it = tid + NT * k, tl = it >> 3, f0 = (it & 7) * 16;
            if (tl < TP) {
                mk[it] = keep[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = ((keep[k] >> (4 * q + e)) & 1u) ? xv[k][4 * q + e] * ks_v : 0.f;
                    *(f32x4*)&xs[tl * HF_S + f0 + 4 * q] = o4;
                }
            }
        }
        if (in_w)                                          // target_weak = target.max(-2) (main.py:95): a wave per class
            for (int c = wv; c < NC; c += NW) {
                float t = -3.0e38f;
                for (int tt = lane; tt < T; tt += 64) t = fmaxf(t, hl.target[((size_t)b * T + tt) * NC + c]);
                t = wave_max(t);
                if (lane == 0) tmaxs[c] = t;
            }
    }
    lds_barrier();
    // ---- H1: logits[t][o] = x[t][:] . W[o][:] + bias: (TP / 16) x 2 tiles of 16 x 16, K = 128 (same MFMA order as k_heads_fwd) ----
    for (int tile = wv; tile < (TP / 16) * 2; tile += NW) {
        const int rt = tile >> 1, ct = tile & 1;
        const float* A = xs + (16 * rt + i16) * HF_S + kq;
        const float* Bp = wsm + (16 * ct + i16) * HF_S + kq;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
        for (int s4 = 0; s4 < 32; s4 += 2) {
            acc0 = hf_mfma16(A[4 * s4], Bp[4 * s4], acc0);
            acc1 = hf_mfma16(A[4 * s4 + 4], Bp[4 * s4 + 4], acc1);
        }
        const int o = 16 * ct + i16;
        const float bias = (o < NC) ? hf.wd[NC * 128 + o] : (o < NO ? hf.wd[NC * 128 + NC + NC * 128 + (o - NC)] : 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[(16 * rt + 4 * kq + r) * HF_SD + o] = acc0[r] + acc1[r] + bias;
    }
    lds_barrier();
    // ---- H2: softmax over classes + sigmoid, 8 threads per frame (thread `sub`: classes sub, sub + 8) ------------------------
    float sv_r[TRIPS][2], raw_r[TRIPS][2];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
        const int it = tid + NT * k, tl = it >> 3, sub = it & 7;
        const bool has[2] = {sub < NC, sub + 8 < NC};
#pragma unroll
        for (int q = 0; q < 2; ++q) { sv_r[k][q] = 0.f; raw_r[k][q] = 0.f; }
        if (tl >= TP) {
        } else if (tl < T) {
            float ls[2], ex[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) ls[q] = has[q] ? lg[tl * HF_SD + NC + sub + 8 * q] : -3.0e38f;
            const float mx = hf_red8_max(fmaxf(ls[0], ls[1]));
#pragma unroll
            for (int q = 0; q < 2; ++q) ex[q] = has[q] ? __expf(ls[q] - mx) : 0.f;
            const float inv = rcp_fast(hf_red8_sum(ex[0] + ex[1]));
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (has[q]) {
                    const int c = sub + 8 * q;
                    raw_r[k][q] = ex[q] * inv;
                    const float sof = fminf(fmaxf(raw_r[k][q], 1e-7f), 1.0f);
                    const float sv = sigmoidf_fast(lg[tl * HF_SD + c]);
                    sv_r[k][q] = sv;
                    if (store) hf.strong[(size_t)(b * T + tl) * NC + c] = sv;
                    nums[tl][c] = sv * sof;
                    dens[tl][c] = sof;
                }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (has[q]) { nums[tl][sub + 8 * q] = 0.f; dens[tl][sub + 8 * q] = 0.f; }
        }
    }
    lds_barrier();
    // ---- H3: attention pooling over time (a wave per class), the weak posterior, its loss terms and gradient ------------------
    for (int c = wv; c < NC; c += NW) {
        float a = 0.f, d2 = 0.f;
#pragma unroll
        for (int tl2 = lane; tl2 < HF_TMAX; tl2 += 64)
            if (tl2 < TP) { a += nums[tl2][c]; d2 += dens[tl2][c]; }
        a = wave_sum(a); d2 = wave_sum(d2);
        if (lane == 0) {
            const float num = 0.f + a, den = 0.f + d2;
            const float wk = num / den;
            if (store) hf.weak[b * NC + c] = wk;
            const float pe = self_t ? wk : hl.weak_ema[b * NC + c];
            const float diff = wk - pe;
            float l3 = diff * diff, l0 = 0.f, l4 = 0.f;
            float dw = cw_v * 2.0f * diff * inv_nW_v;
            if (in_w) {
                const float t = tmaxs[c];
                l0 = bce_term(wk, t);
                l4 = bce_term(pe, t);
                dw += bce_grad(wk, t) * inv_wb_v;
            }
            dnum[c] = dw / den;
            dden[c] = -dw * wk / den;
            lw[3 * c] = l0; lw[3 * c + 1] = l3; lw[3 * c + 2] = l4;
        }
    }
    lds_barrier();
    // ---- H4: softmax / sigmoid backward per frame -> dlogits (over the logits tile) -------------------------------------------
    float lacc1 = 0.f, lacc2 = 0.f, lacc5 = 0.f;       // strong_bce, mse_strong, strong_ema_bce
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
        const int it = tid + NT * k, tl = it >> 3, sub = it & 7;
        float* dlr = lg + tl * HF_SD;
        if (tl >= TP) {
        } else if (tl < T) {
            const bool has[2] = {sub < NC, sub + 8 < NC};
            float ds[2], dsig[2], dpart = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ds[q] = 0.f; dsig[q] = 0.f;
                if (has[q]) {
                    const int c = sub + 8 * q;
                    const float sv = sv_r[k][q], raw = raw_r[k][q];
                    const float sof = fminf(fmaxf(raw, 1e-7f), 1.0f);
                    const float pass = (raw >= 1e-7f && raw <= 1.0f) ? 1.f : 0.f;
                    ds[q] = (dnum[c] * sv + dden[c]) * pass;
                    dpart += raw * ds[q];
                    const float pe = self_t ? sv : pe_r[k][q];
                    const float diff = sv - pe;
                    lacc2 += diff * diff;
                    float gin = cw_v * 2.0f * diff * inv_nS_v;
                    if (in_s) {
                        const float tg = tg_r[k][q];
                        lacc1 += bce_term(sv, tg);
                        lacc5 += bce_term(pe, tg);
                        gin += bce_grad(sv, tg) * inv_sb_v;
                    }
                    dsig[q] = (gin + dnum[c] * sof) * sv * (1.0f - sv);
                }
            }
            const float dot = hf_red8_sum(dpart);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (has[q]) {
                    dlr[sub + 8 * q] = dsig[q];
                    dlr[NC + sub + 8 * q] = raw_r[k][q] * (ds[q] - dot);
                }
            for (int o = NO + sub; o < HF_MAXO; o += 8) dlr[o] = 0.f;
        } else {
            for (int o = sub; o < HF_MAXO; o += 8) dlr[o] = 0.f;
        }
    }
    {
        const float v1 = wave_sum(lacc1), v2 = wave_sum(lacc2), v5 = wave_sum(lacc5);
        if (lane == 0) { red[wv * 8 + 1] = v1; red[wv * 8 + 2] = v2; red[wv * 8 + 5] = v5; }
    }
    lds_barrier();
    // ---- H5: the MFMA jobs, round-robin over the waves ---------------------------------------------------------------------------
    //   jobs 0..7: dW[o][f] = sum_t dl[t][o] x[t][f] for this direction's 64 features (2 x 4 tiles, K = TP frames)
    //   jobs 8.. : dx[t][f] = (sum_o dl[t][o] W[o][f]) * mask for this direction's 64 features ((TP / 16) x 4 tiles, K = 32)
    const int njobs = 8 + (TP / 16) * 4;
    for (int j = wv; j < njobs; j += NW) {
        if (j < 8) {
            const int ot = j >> 2, ft = 4 * dir + (j & 3);
            const float* A = lg + kq * HF_SD + 16 * ot + i16;           // A[i = o][k = t]
            const float* Bp = xs + kq * HF_S + 16 * ft + i16;           // B[k = t][j = f]
            f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
            for (int s4 = 0; s4 < TP / 4; ++s4) wacc = hf_mfma16(A[4 * s4 * HF_SD], Bp[4 * s4 * HF_S], wacc);
            const int f = 16 * ft + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * ot + 4 * kq + r;
                if (o < NC) pr[o * 128 + f] = wacc[r];
                else if (o < NO) pr[NC * 128 + NC + (o - NC) * 128 + f] = wacc[r];
            }
        } else {
            const int tile = j - 8, tt = tile >> 2, fl = tile & 3, ft = 4 * dir + fl;
            const float* A = lg + (16 * tt + i16) * HF_SD + kq;         // A[i = t][k = o]
            const float* Bp = wsm + kq * HF_S + 16 * ft + i16;          // B[k = o][j = f]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) acc = hf_mfma16(A[4 * s4], Bp[4 * s4 * HF_S], acc);
            const int f = 16 * ft + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tl = 16 * tt + 4 * kq + r;
                const uint32_t keep = (mk[tl * 8 + (f >> 4)] >> (f & 15)) & 1u;
                dout_s[tl * 64 + 16 * fl + i16] = keep ? acc[r] * ks_v : 0.f;          // (rows >= T: keep == 0)
            }
        }
    }
    if (store && wv == NW - 1 && lane < NO) {            // bias gradients: the frame sums of dlogits, in frame order
        float a = 0.f;
        for (int tl = 0; tl < TP; ++tl) a += lg[tl * HF_SD + lane];
        const float bacc = 0.f + a;
        if (lane < NC) pr[NC * 128 + lane] = bacc;
        else pr[2 * NC * 128 + NC + (lane - NC)] = bacc;
    }
    if (store && tid < 6) {                              // the clip's six loss sums -> per-clip partials (k_heads_fin adds them up)
        float s2 = 0.f;
        if (tid == 1 || tid == 2 || tid == 5)
            for (int w2 = 0; w2 < NW; ++w2) s2 += red[w2 * 8 + tid];
        else {
            const int sel = tid == 0 ? 0 : (tid == 3 ? 1 : 2);
            for (int c = 0; c < NC; ++c) s2 += lw[3 * c + sel];
        }
        lossp[tid] = s2;
    }
    lds_barrier();
}
#endif
