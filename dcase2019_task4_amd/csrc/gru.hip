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
    // gi of the NEXT step is fetched one iteration ahead: its HBM/L2 latency hides under this step
    float gi_next = gi[((size_t)(b * T + (dir ? T - 1 : 0)) * 2 + dir) * 192 + g];
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const float giv = gi_next;
        if (step + 1 < T) {
            const int tn = dir ? (T - 2 - step) : (step + 1);
            gi_next = gi[((size_t)(b * T + tn) * 2 + dir) * 192 + g];
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
        gis[g] = giv;
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
    // operands of the NEXT step are fetched one iteration ahead (tid < 64 only)
    float n_do = 0.f, n_r = 0.f, n_z = 0.f, n_n = 0.f, n_g = 0.f, n_hp = 0.f;
    auto fetch = [&](int t) {
        n_do = d_out[(size_t)(b * T + t) * 128 + dir * 64 + j];
        const float* gs = gates + ((size_t)(b * T + t) * 2 + dir) * 256;
        n_r = gs[j]; n_z = gs[64 + j]; n_n = gs[128 + j]; n_g = gs[192 + j];
        const int tp = dir ? t + 1 : t - 1;
        n_hp = (tp >= 0 && tp < T) ? out[(size_t)(b * T + tp) * 128 + dir * 64 + j] : 0.f;
    };
    if (tid < 64) fetch(dir ? 0 : T - 1);
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : (T - 1 - step);
        if (tid < 64) {
            const float dh = n_do + dh_carry;
            const float r = n_r, z = n_z, nn = n_n, ghn = n_g, hp = n_hp;
            if (step + 1 < T) fetch(dir ? step + 1 : T - 2 - step);
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
