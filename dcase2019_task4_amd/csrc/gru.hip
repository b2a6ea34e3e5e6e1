// gru.hip - bidirectional GRU recurrence (hidden 64), forward and backward through time.
//
// Reference op: nn.GRU(n_in, 64, bidirectional=True, batch_first=True) inside BidirectionalGRU
// (baseline/models/RNN.py:12-16); torch gate order (r, z, n), h0 = 0:
//   r = sig(gi_r + gh_r); z = sig(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1-z) n + z h
// with gi = W_ih x + b_ih (one batched GEMM over all time steps, gemm.hip) and
// gh = W_hh h + b_hh (this file).
//
// The recurrence is a serial chain of T/8 steps, so it is latency- not throughput-bound: one
// workgroup per (clip, direction) keeps its W_hh slice in REGISTERS for all steps (192 threads x
// 64 weights), broadcasts h through LDS, and all B x 2 chains run concurrently on separate CUs.
// Backward keeps W_hh^T the same way (thread = (gate block, hidden unit j)).
#include "common.h"
#include "kernels.h"

__device__ __forceinline__ float tanhf_fast(float x) { return 1.0f - 2.0f * rcp_fast(1.0f + __expf(2.0f * x)); }

__global__ __launch_bounds__(192) void k_gru_fwd(const float* __restrict__ gi, const float* __restrict__ w_hh_f,
                                                  const float* __restrict__ w_hh_r, const float* __restrict__ b_hh_f,
                                                  const float* __restrict__ b_hh_r, float* __restrict__ out,
                                                  float* __restrict__ gates, int T) {
    __shared__ __attribute__((aligned(16))) float hs[64];
    __shared__ float ghs[192];
    __shared__ float gis[192];
    __shared__ __attribute__((aligned(16))) float Wl[192 * 68];   // W_hh staged coalesced; row stride 68: conflict-free b128 row reads
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
    const float bh = (dir ? b_hh_r : b_hh_f)[g];
    if (g < 64) hs[g] = 0.f;
    float hprev = 0.f;
    // Wave roles keep global LOADS and global STORES in different waves.  vmcnt counts both and they
    // complete out of order with respect to each other, so a wave that does both must drain its stores
    // (full write latency) every time it needs a loaded value.  Wave 0 (gate math) only stores; waves 1-2
    // only load - gi of the NEXT step, one iteration ahead, for all 192 gate rows - and hand it over in LDS.
    float gi_a = 0.f, gi_b = 0.f;
    auto fetch = [&](int t) {
        const float* src = gi + ((size_t)(b * T + t) * 2 + dir) * 192;
        gi_a = src[g];
        if (g < 128) gi_b = src[g - 64];
    };
    if (g >= 64) fetch(dir ? T - 1 : 0);
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        float cur_a = 0.f, cur_b = 0.f;
        if (g >= 64) {
            cur_a = gi_a; cur_b = gi_b;
            if (step + 1 < T) fetch(dir ? (T - 2 - step) : (step + 1));
        }
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
        if (g >= 64) {
            gis[g] = cur_a;
            if (g < 128) gis[g - 64] = cur_b;
        }
        lds_barrier();
        if (g < 64) {
            const float r = sigmoidf_fast(gis[g] + ghs[g]);
            const float z = sigmoidf_fast(gis[64 + g] + ghs[64 + g]);
            const float ghn = ghs[128 + g];
            const float nn = tanhf_fast(gis[128 + g] + r * ghn);
            const float h = (1.0f - z) * nn + z * hprev;
            out[(size_t)(b * T + t) * 128 + dir * 64 + g] = h;
            if (gates) {
                float* gs = gates + ((size_t)(b * T + t) * 2 + dir) * 256;
                gs[g] = r; gs[64 + g] = z; gs[128 + g] = nn; gs[192 + g] = ghn;
            }
            hs[g] = h;
            hprev = h;
        }
        lds_barrier();
    }
}

__global__ __launch_bounds__(192) void k_gru_bwd(const float* __restrict__ d_out, const float* __restrict__ out,
                                                  const float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                  const float* __restrict__ w_hh_r, float* __restrict__ dgi,
                                                  float* __restrict__ dgh, float* __restrict__ hprev_out, int T) {
    __shared__ __attribute__((aligned(16))) float dghs[192];
    __shared__ float parts[3][64];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int part = tid >> 6, j = tid & 63;
    const float* whh = dir ? w_hh_r : w_hh_f;
    float wt[64];   // W_hh[part*64 + i][j], i = 0..63
#pragma unroll
    for (int i = 0; i < 64; ++i) wt[i] = whh[(part * 64 + i) * 64 + j];
    float dh_carry = 0.f, dh_z = 0.f;
    // Same load/store wave split as the forward: wave 0 computes and stores, waves 1-2 prefetch the six
    // operand rows (d_out, r, z, n, gh_n, h_prev) of the NEXT step and pass them through LDS.
    __shared__ float ops[2][6][64];
    float pre[3] = {0.f, 0.f, 0.f};
    const int u = tid - 64;     // 0..127 for the loader waves
    auto fetch = [&](int t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = u + 128 * k, a = idx >> 6, jj = idx & 63;
            float v;
            if (a == 0) v = d_out[(size_t)(b * T + t) * 128 + dir * 64 + jj];
            else if (a < 5) v = gates[((size_t)(b * T + t) * 2 + dir) * 256 + (a - 1) * 64 + jj];
            else {
                const int tp = dir ? t + 1 : t - 1;
                v = (tp >= 0 && tp < T) ? out[(size_t)(b * T + tp) * 128 + dir * 64 + jj] : 0.f;
            }
            pre[k] = v;
        }
    };
    auto publish = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = u + 128 * k;
            ops[buf][idx >> 6][idx & 63] = pre[k];
        }
    };
    if (tid >= 64) {
        fetch(dir ? 0 : T - 1);
        publish(0);
        if (T > 1) fetch(dir ? 1 : T - 2);
    }
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : (T - 1 - step);
        if (tid >= 64) {
            if (step + 1 < T) publish((step + 1) & 1);                  // operands of step+1 (loaded a step ago)
            if (step + 2 < T) fetch(dir ? step + 2 : T - 3 - step);     // start loading step+2
        } else {
            const float (*o)[64] = ops[step & 1];
            const float dh = o[0][j] + dh_carry;
            const float r = o[1][j], z = o[2][j], nn = o[3][j], ghn = o[4][j], hp = o[5][j];
            const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
            const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const size_t base = ((size_t)(b * T + t) * 2 + dir) * 192;
            dgi[base + j] = dr_pre; dgi[base + 64 + j] = dz_pre; dgi[base + 128 + j] = dn_pre;
            const float dghn = dn_pre * r;
            dgh[base + j] = dr_pre; dgh[base + 64 + j] = dz_pre; dgh[base + 128 + j] = dghn;
            dghs[j] = dr_pre; dghs[64 + j] = dz_pre; dghs[128 + j] = dghn;
            hprev_out[((size_t)(b * T + t) * 2 + dir) * 64 + j] = hp;
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
        parts[part][j] = (a0 + a1) + (a2 + a3);
        lds_barrier();
        if (tid < 64) dh_carry = dh_z + parts[0][j] + parts[1][j] + parts[2][j];
    }
}

int launch_gru_fwd(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r,
                   float* out, float* gates, int B, int T, hipStream_t st) {
    k_gru_fwd<<<dim3(B, 2), 192, 0, st>>>(gi, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

int launch_gru_bwd(const float* d_out, const float* out, const float* gates, const float* w_hh_f, const float* w_hh_r,
                   float* dgi, float* dgh, float* hprev, int B, int T, hipStream_t st) {
    k_gru_bwd<<<dim3(B, 2), 192, 0, st>>>(d_out, out, gates, w_hh_f, w_hh_r, dgi, dgh, hprev, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
