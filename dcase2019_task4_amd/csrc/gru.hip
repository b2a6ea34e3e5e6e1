// gru.hip - bidirectional GRU recurrence (hidden 64), forward and backward through time.
//
// Reference op: nn.GRU(n_in, 64, bidirectional=True, batch_first=True) inside BidirectionalGRU
// (baseline/models/RNN.py:12-16); torch gate order (r, z, n), h0 = 0:
//   r = sig(gi_r + gh_r); z = sig(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1-z) n + z h
// with gi = W_ih x + b_ih (one batched GEMM over all time steps, gemm.hip) and
// gh = W_hh h + b_hh (this file).
//
// The recurrence is a serial chain of T/8 steps: latency-, not throughput-bound.  One workgroup per
// (clip, direction) keeps its W_hh slice in REGISTERS for all steps (192 threads x 64 weights) and
// broadcasts h through LDS; all B x 2 chains run concurrently on separate CUs.
//
// The per-step loop touches NO global memory.  On CDNA4 `vmcnt` counts loads and stores alike and they
// retire out of order with respect to each other, so a wave that needs a loaded value has to drain all
// its outstanding stores first - a full HBM/L2 write latency per step when every step both loads its
// inputs and stores its outputs (first version: 0.72 us/step).  Instead the inputs of GRU_SB steps are
// pulled into LDS and the outputs of GRU_SB steps are pushed out of LDS at block boundaries, with the
// next block's loads issued GRU_SB steps before they are needed; inside a block the only
// synchronisation is an LDS-only barrier (lds_barrier: no vmcnt drain).
#include "common.h"
#include "kernels.h"

#define GRU_SB 8

__device__ __forceinline__ float tanhf_fast(float x) { return 1.0f - 2.0f * rcp_fast(1.0f + __expf(2.0f * x)); }

__global__ __launch_bounds__(192) void k_gru_fwd(const float* __restrict__ gi, const float* __restrict__ w_hh_f,
                                                  const float* __restrict__ w_hh_r, const float* __restrict__ b_hh_f,
                                                  const float* __restrict__ b_hh_r, float* __restrict__ out,
                                                  float* __restrict__ gates, int T) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* hs = gsm;                                  // [64]
    float* ghs = hs + 64;                             // [192]
    float* gi_s = ghs + 192;                          // [2][GRU_SB][192]
    float* hist = gi_s + 2 * GRU_SB * 192;            // [2][GRU_SB][320] : h, r, z, n, gh_n
    float* Wl = hist + 2 * GRU_SB * 320;              // [192][68] staging of W_hh (coalesced global read)
    const int b = blockIdx.x, dir = blockIdx.y, g = threadIdx.x;
    const float* whh = dir ? w_hh_r : w_hh_f;
    for (int e = g; e < 192 * 64; e += 192) Wl[(e >> 6) * 68 + (e & 63)] = whh[e];
    __syncthreads();
    float w[64];
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
        const float4 v = *(const float4*)(Wl + g * 68 + j);
        w[j] = v.x; w[j + 1] = v.y; w[j + 2] = v.z; w[j + 3] = v.w;
    }
    float bh = (dir ? b_hh_r : b_hh_f)[g];
    // pin the wait for this load HERE: left to the compiler it lands at the first use inside the step loop,
    // where `s_waitcnt vmcnt(0)` also drains the block-boundary stores on every re-entry
    asm volatile("" : "+v"(bh));
    if (g < 64) hs[g] = 0.f;
    float hprev = 0.f;
    const int nblk = (T + GRU_SB - 1) / GRU_SB;
    auto t_of = [&](int step) { return dir ? (T - 1 - step) : step; };
    // block 0 inputs
    {
        float first[GRU_SB];
#pragma unroll
        for (int s = 0; s < GRU_SB; ++s) first[s] = (s < T) ? gi[((size_t)(b * T + t_of(s)) * 2 + dir) * 192 + g] : 0.f;
#pragma unroll
        for (int s = 0; s < GRU_SB; ++s) gi_s[s * 192 + g] = first[s];
    }
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, s0 = blk * GRU_SB;
        const int sb = min(GRU_SB, T - s0);
        // ---- block boundary: push the previous block's outputs, pull the next block's inputs ----------
        if (blk > 0) {
            const float* hp = hist + (cur ^ 1) * GRU_SB * 320;
            for (int e = g; e < GRU_SB * 320; e += 192) {
                const int s = e / 320, k = e % 320;
                const int t = t_of(s0 - GRU_SB + s);
                if (k < 64) out[(size_t)(b * T + t) * 128 + dir * 64 + k] = hp[e];
                else if (gates) gates[((size_t)(b * T + t) * 2 + dir) * 256 + (k - 64)] = hp[e];
            }
        }
        float nxt[GRU_SB];
#pragma unroll
        for (int s = 0; s < GRU_SB; ++s) {
            const int st = s0 + GRU_SB + s;
            nxt[s] = (st < T) ? gi[((size_t)(b * T + t_of(st)) * 2 + dir) * 192 + g] : 0.f;
        }
        // ---- GRU_SB steps on LDS only ---------------------------------------------------------------------
        const float* gib = gi_s + cur * GRU_SB * 192;
        float* hb = hist + cur * GRU_SB * 320;
        for (int s = 0; s < sb; ++s) {
            float a0 = bh, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                const float4 h4 = *(const float4*)(hs + j);
                a0 = fmaf(w[j], h4.x, a0);
                a1 = fmaf(w[j + 1], h4.y, a1);
                a2 = fmaf(w[j + 2], h4.z, a2);
                a3 = fmaf(w[j + 3], h4.w, a3);
            }
            ghs[g] = (a0 + a1) + (a2 + a3);
            lds_barrier();
            if (g < 64) {
                const float* gr = gib + s * 192;
                const float r = sigmoidf_fast(gr[g] + ghs[g]);
                const float z = sigmoidf_fast(gr[64 + g] + ghs[64 + g]);
                const float ghn = ghs[128 + g];
                const float nn = tanhf_fast(gr[128 + g] + r * ghn);
                const float h = (1.0f - z) * nn + z * hprev;
                float* ho = hb + s * 320;
                ho[g] = h; ho[64 + g] = r; ho[128 + g] = z; ho[192 + g] = nn; ho[256 + g] = ghn;
                hs[g] = h;
                hprev = h;
            }
            lds_barrier();
        }
        // ---- hand the prefetched inputs of the next block to LDS (loads were issued GRU_SB steps ago) ----
        float* gin = gi_s + (cur ^ 1) * GRU_SB * 192;
#pragma unroll
        for (int s = 0; s < GRU_SB; ++s) gin[s * 192 + g] = nxt[s];
        lds_barrier();
    }
    {   // last block's outputs
        const int blk = nblk - 1, s0 = blk * GRU_SB, sb = T - s0;
        const float* hp = hist + (blk & 1) * GRU_SB * 320;
        for (int e = g; e < sb * 320; e += 192) {
            const int s = e / 320, k = e % 320;
            const int t = t_of(s0 + s);
            if (k < 64) out[(size_t)(b * T + t) * 128 + dir * 64 + k] = hp[e];
            else if (gates) gates[((size_t)(b * T + t) * 2 + dir) * 256 + (k - 64)] = hp[e];
        }
    }
}

// Backward through time.  Thread = (gate block part = tid>>6, hidden unit j = tid&63) holds column j of
// W_hh's gate block `part` (W_hh^T mat-vec: dh_prev = W_hh^T dgh).  Per step inputs: d_out, r, z, n, gh_n,
// h_prev (6 x 64); outputs: dgi (192), dgh (192), h_prev (64).
__global__ __launch_bounds__(192) void k_gru_bwd(const float* __restrict__ d_out, const float* __restrict__ out,
                                                  const float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                  const float* __restrict__ w_hh_r, float* __restrict__ dgi,
                                                  float* __restrict__ dgh, float* __restrict__ hprev_out, int T) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* dghs = gsm;                                // [192]
    float* parts = dghs + 192;                        // [3][64]
    float* ops = parts + 192;                         // [2][GRU_SB][384] : d_out, r, z, n, gh_n, h_prev
    float* hist = ops + 2 * GRU_SB * 384;             // [2][GRU_SB][448] : dgi(192), dgh(192), h_prev(64)
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int part = tid >> 6, j = tid & 63;
    const float* whh = dir ? w_hh_r : w_hh_f;
    float wt[64];   // W_hh[part*64 + i][j], i = 0..63
#pragma unroll
    for (int i = 0; i < 64; ++i) wt[i] = whh[(part * 64 + i) * 64 + j];
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(wt[i]));      // same reason as `bh` in k_gru_fwd
    float dh_carry = 0.f, dh_z = 0.f;
    const int nblk = (T + GRU_SB - 1) / GRU_SB;
    auto t_of = [&](int step) { return dir ? step : (T - 1 - step); };     // reverse of the forward order
    // address of operand k (0..383: d_out, r, z, n, gh_n, h_prev rows of 64) of recurrence step `step`,
    // always a valid address (clamped) so that the loads can be issued unconditionally, back to back;
    // `ok` says whether the value is real or must read as 0 (past the end / h_prev before the first step)
    auto op_addr = [&](int step, int k, bool& ok) -> const float* {
        const int t = t_of(min(step, T - 1)), a = k >> 6, jj = k & 63;
        const int tp = dir ? t + 1 : t - 1;
        const int tpc = min(max(tp, 0), T - 1);
        ok = (step < T) && (a != 5 || (tp >= 0 && tp < T));
        const float* p0 = d_out + (size_t)(b * T + t) * 128 + dir * 64 + jj;
        const float* p1 = gates + ((size_t)(b * T + t) * 2 + dir) * 256 + (max(a, 1) - 1) * 64 + jj;
        const float* p2 = out + (size_t)(b * T + tpc) * 128 + dir * 64 + jj;
        return a == 0 ? p0 : (a < 5 ? p1 : p2);
    };
    constexpr int NOP = (GRU_SB * 384) / 192;       // 16 operand values per thread per block
    {
        float first[NOP];
        unsigned okm = 0;
#pragma unroll
        for (int i = 0; i < NOP; ++i) {
            const int e = tid + 192 * i;
            bool ok;
            first[i] = *op_addr(e / 384, e % 384, ok);
            okm |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < NOP; ++i) ops[tid + 192 * i] = ((okm >> i) & 1u) ? first[i] : 0.f;
    }
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, s0 = blk * GRU_SB;
        const int sb = min(GRU_SB, T - s0);
        if (blk > 0) {
            const float* hp = hist + (cur ^ 1) * GRU_SB * 448;
            for (int e = tid; e < GRU_SB * 448; e += 192) {
                const int s = e / 448, k = e % 448;
                const size_t bt = (size_t)(b * T + t_of(s0 - GRU_SB + s)) * 2 + dir;
                if (k < 192) dgi[bt * 192 + k] = hp[e];
                else if (k < 384) dgh[bt * 192 + (k - 192)] = hp[e];
                else hprev_out[bt * 64 + (k - 384)] = hp[e];
            }
        }
        float nxt[NOP];
        unsigned okm = 0;
#pragma unroll
        for (int i = 0; i < NOP; ++i) {
            const int e = tid + 192 * i;
            bool ok;
            nxt[i] = *op_addr(s0 + GRU_SB + e / 384, e % 384, ok);
            okm |= (ok ? 1u : 0u) << i;
        }
        const float* ob = ops + cur * GRU_SB * 384;
        float* hb = hist + cur * GRU_SB * 448;
        for (int s = 0; s < sb; ++s) {
            if (tid < 64) {
                const float* o = ob + s * 384;
                const float dh = o[j] + dh_carry;
                const float r = o[64 + j], z = o[128 + j], nn = o[192 + j], ghn = o[256 + j], hp = o[320 + j];
                const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
                const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
                const float dr_pre = dn_pre * ghn * r * (1.0f - r);
                const float dghn = dn_pre * r;
                float* ho = hb + s * 448;
                ho[j] = dr_pre; ho[64 + j] = dz_pre; ho[128 + j] = dn_pre;
                ho[192 + j] = dr_pre; ho[256 + j] = dz_pre; ho[320 + j] = dghn;
                ho[384 + j] = hp;
                dghs[j] = dr_pre; dghs[64 + j] = dz_pre; dghs[128 + j] = dghn;
                dh_z = dh * z;
            }
            lds_barrier();
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < 64; i += 4) {
                const float4 d4 = *(const float4*)(dghs + part * 64 + i);
                a0 = fmaf(wt[i], d4.x, a0);
                a1 = fmaf(wt[i + 1], d4.y, a1);
                a2 = fmaf(wt[i + 2], d4.z, a2);
                a3 = fmaf(wt[i + 3], d4.w, a3);
            }
            parts[part * 64 + j] = (a0 + a1) + (a2 + a3);
            lds_barrier();
            if (tid < 64) dh_carry = dh_z + parts[j] + parts[64 + j] + parts[128 + j];
        }
        float* on = ops + (cur ^ 1) * GRU_SB * 384;
#pragma unroll
        for (int i = 0; i < NOP; ++i) on[tid + 192 * i] = ((okm >> i) & 1u) ? nxt[i] : 0.f;
        lds_barrier();
    }
    {
        const int blk = nblk - 1, s0 = blk * GRU_SB, sb = T - s0;
        const float* hp = hist + (blk & 1) * GRU_SB * 448;
        for (int e = tid; e < sb * 448; e += 192) {
            const int s = e / 448, k = e % 448;
            const size_t bt = (size_t)(b * T + t_of(s0 + s)) * 2 + dir;
            if (k < 192) dgi[bt * 192 + k] = hp[e];
            else if (k < 384) dgh[bt * 192 + (k - 192)] = hp[e];
            else hprev_out[bt * 64 + (k - 384)] = hp[e];
        }
    }
}

static const size_t GRU_FWD_LDS = (size_t)(64 + 192 + 2 * GRU_SB * 192 + 2 * GRU_SB * 320 + 192 * 68) * sizeof(float);
static const size_t GRU_BWD_LDS = (size_t)(192 + 192 + 2 * GRU_SB * 384 + 2 * GRU_SB * 448) * sizeof(float);

int launch_gru_fwd(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r,
                   float* out, float* gates, int B, int T, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU_FWD_LDS));
        attr_done = true;
    }
    k_gru_fwd<<<dim3(B, 2), 192, GRU_FWD_LDS, st>>>(gi, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_gru_bwd(const float* d_out, const float* out, const float* gates, const float* w_hh_f, const float* w_hh_r,
                   float* dgi, float* dgh, float* hprev, int B, int T, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU_BWD_LDS));
        attr_done = true;
    }
    k_gru_bwd<<<dim3(B, 2), 192, GRU_BWD_LDS, st>>>(d_out, out, gates, w_hh_f, w_hh_r, dgi, dgh, hprev, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
