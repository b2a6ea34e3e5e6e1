// ggru.hip - bidirectional GRU recurrence for hidden sizes H in {128, 256} (n_RNN_cell of baseline/models/CRNN.py:12-31;
// BASELINE.json configs[4] uses 256), forward and backward through time.  H = 64 keeps gru.hip.
//
// Reference op: nn.GRU(n_in, H, bidirectional=True, batch_first=True) (baseline/models/RNN.py:12-16), gate order (r, z, n):
//   r = sig(gi_r + gh_r); z = sig(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1 - z) n + z h;  gh = W_hh h + b_hh
// gi = W_ih x + b_ih for all time steps is one batched GEMM in front (gemm.hip).
//
// W_hh no longer fits in registers: 3H x H fp32 = 768 KB at H = 256 against 512 KB of VGPRs + 160 KB of LDS per CU.
// One workgroup of H threads per (clip, direction) STREAMS it from L2 every time step, re-laid by k_ggru_pack so that
// the stream is perfectly coalesced: thread j (hidden unit j) reads its three gate rows as [k / 4][row][4] float4 groups,
// 16 consecutive bytes per lane, while h is broadcast from LDS.  The step time is what a CU can pull from L2
// (768 KB per step), not the arithmetic; all B x 2 chains run concurrently on separate CUs.
#include "common.h"
#include "kernels.h"
#include "gkernels.h"

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float gtanhf_fast(float x) { return 1.0f - 2.0f * rcp_fast(1.0f + __expf(2.0f * x)); }

// wp [dir][(k / 4) * 3H + row][4] = W_hh[dir][row][k .. k + 3];  wpT[dir][(g / 4) * H + j][4] = W_hh[dir][g .. g + 3][j]
__global__ __launch_bounds__(256) void k_ggru_pack(const float* __restrict__ w_f, const float* __restrict__ w_r, float* __restrict__ wp,
                                                    float* __restrict__ wpT, int H) {
    const int n = 3 * H * H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n) return;
    const int dir = i / n, e = i % n;
    const float* w = dir ? w_r : w_f;
    {
        const int q = e & 3, row = (e >> 2) % (3 * H), k4 = (e >> 2) / (3 * H);
        wp[(size_t)dir * n + e] = w[(size_t)row * H + 4 * k4 + q];
    }
    if (wpT) {
        const int q = e & 3, j = (e >> 2) % H, g4 = (e >> 2) / H;
        wpT[(size_t)dir * n + e] = w[(size_t)(4 * g4 + q) * H + j];
    }
}

int launch_ggru_pack(const float* w_hh_f, const float* w_hh_r, float* wp, float* wpT, int H, hipStream_t st) {
    k_ggru_pack<<<(2 * 3 * H * H + 255) / 256, 256, 0, st>>>(w_hh_f, w_hh_r, wp, wpT, H);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

template <int H>
__global__ __launch_bounds__(H) void k_ggru_fwd(const float* __restrict__ gi, const float* __restrict__ wp,
                                                 const float* __restrict__ b_hh_f, const float* __restrict__ b_hh_r,
                                                 float* __restrict__ out, float* __restrict__ gates, int T) {
    __shared__ __attribute__((aligned(16))) float hs[2][H];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const float* w = wp + (size_t)dir * 3 * H * H;
    const float* bhh = dir ? b_hh_r : b_hh_f;
    const float bh_r = bhh[j], bh_z = bhh[H + j], bh_n = bhh[2 * H + j];
    hs[0][j] = 0.f;
    float hprev = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        const int cur = s & 1;
        const float* g = gi + ((size_t)(b * T + t) * 2 + dir) * 3 * H;
        const float gi_r = g[j], gi_z = g[H + j], gi_n = g[2 * H + j];
        float ar = bh_r, az = bh_z, an = bh_n;
        const v4f* w4 = (const v4f*)w;
#pragma unroll 8
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const v4f h4 = *(const v4f*)&hs[cur][4 * k4];
            const v4f wr = w4[(size_t)k4 * 3 * H + j], wz = w4[(size_t)k4 * 3 * H + H + j], wn = w4[(size_t)k4 * 3 * H + 2 * H + j];
            ar = fmaf(wr.x, h4.x, ar); ar = fmaf(wr.y, h4.y, ar); ar = fmaf(wr.z, h4.z, ar); ar = fmaf(wr.w, h4.w, ar);
            az = fmaf(wz.x, h4.x, az); az = fmaf(wz.y, h4.y, az); az = fmaf(wz.z, h4.z, az); az = fmaf(wz.w, h4.w, az);
            an = fmaf(wn.x, h4.x, an); an = fmaf(wn.y, h4.y, an); an = fmaf(wn.z, h4.z, an); an = fmaf(wn.w, h4.w, an);
        }
        const float r = sigmoidf_fast(gi_r + ar);
        const float z = sigmoidf_fast(gi_z + az);
        const float nn = gtanhf_fast(gi_n + r * an);
        const float h = (1.0f - z) * nn + z * hprev;
        hs[cur ^ 1][j] = h;
        hprev = h;
        out[(size_t)(b * T + t) * 2 * H + dir * H + j] = h;
        if (gates) {
            float* gt = gates + ((size_t)(b * T + t) * 2 + dir) * 4 * H;
            gt[j] = r; gt[H + j] = z; gt[2 * H + j] = nn; gt[3 * H + j] = an;
        }
        __syncthreads();
    }
}

int launch_ggru_fwd(int H, const float* gi, const float* wp, const float* b_hh_f, const float* b_hh_r, float* out, float* gates,
                    int B, int T, hipStream_t st) {
    if (H == 256) k_ggru_fwd<256><<<dim3(B, 2), 256, 0, st>>>(gi, wp, b_hh_f, b_hh_r, out, gates, T);
    else if (H == 128) k_ggru_fwd<128><<<dim3(B, 2), 128, 0, st>>>(gi, wp, b_hh_f, b_hh_r, out, gates, T);
    else {
        sed_set_error("generic GRU forward: unsupported hidden size %d", H);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// Backward through time.  Thread j: gate gradients of unit j -> dgi / dgh rows, then the carry
//   dh_prev[j] = dh[j] z[j] + sum_g dgh[g] W_hh[g][j]   (column j of W_hh, streamed from the transposed packing)
template <int H>
__global__ __launch_bounds__(H) void k_ggru_bwd(const float* __restrict__ d_out, const float* __restrict__ out,
                                                 const float* __restrict__ gates, const float* __restrict__ wpT,
                                                 float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ hprev_out,
                                                 int T) {
    __shared__ __attribute__((aligned(16))) float ds[2][3 * H];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const float* w = wpT + (size_t)dir * 3 * H * H;
    float carry = 0.f;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : (T - 1 - s);             // reverse of the forward order
        const int tp = dir ? t + 1 : t - 1;              // the step whose output was this step's h_prev
        const int cur = s & 1;
        const size_t bt = (size_t)(b * T + t);
        const float* gt = gates + (bt * 2 + dir) * 4 * H;
        const float r = gt[j], z = gt[H + j], nn = gt[2 * H + j], ghn = gt[3 * H + j];
        const float hp = (tp >= 0 && tp < T) ? out[(size_t)(b * T + tp) * 2 * H + dir * H + j] : 0.f;
        const float dh = d_out[bt * 2 * H + dir * H + j] + carry;
        const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
        const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
        const float dr_pre = dn_pre * ghn * r * (1.0f - r);
        const float dghn = dn_pre * r;
        float* gi_o = dgi + (bt * 2 + dir) * 3 * H;
        float* gh_o = dgh + (bt * 2 + dir) * 3 * H;
        gi_o[j] = dr_pre; gi_o[H + j] = dz_pre; gi_o[2 * H + j] = dn_pre;
        gh_o[j] = dr_pre; gh_o[H + j] = dz_pre; gh_o[2 * H + j] = dghn;
        hprev_out[(bt * 2 + dir) * H + j] = hp;
        ds[cur][j] = dr_pre; ds[cur][H + j] = dz_pre; ds[cur][2 * H + j] = dghn;
        __syncthreads();
        float a0 = 0.f, a1 = 0.f;
        const v4f* w4 = (const v4f*)w;
#pragma unroll 8
        for (int g4 = 0; g4 < 3 * H / 4; g4 += 2) {
            const v4f d0 = *(const v4f*)&ds[cur][4 * g4], d1 = *(const v4f*)&ds[cur][4 * g4 + 4];
            const v4f w0 = w4[(size_t)g4 * H + j], w1 = w4[(size_t)(g4 + 1) * H + j];
            a0 = fmaf(w0.x, d0.x, a0); a0 = fmaf(w0.y, d0.y, a0); a0 = fmaf(w0.z, d0.z, a0); a0 = fmaf(w0.w, d0.w, a0);
            a1 = fmaf(w1.x, d1.x, a1); a1 = fmaf(w1.y, d1.y, a1); a1 = fmaf(w1.z, d1.z, a1); a1 = fmaf(w1.w, d1.w, a1);
        }
        carry = dh * z + (a0 + a1);
        // (ds is double-buffered: the next step writes the other half, and its barrier orders it against this step's reads)
    }
}

int launch_ggru_bwd(int H, const float* d_out, const float* out, const float* gates, const float* wpT, float* dgi, float* dgh,
                    float* hprev, int B, int T, hipStream_t st) {
    if (H == 256) k_ggru_bwd<256><<<dim3(B, 2), 256, 0, st>>>(d_out, out, gates, wpT, dgi, dgh, hprev, T);
    else if (H == 128) k_ggru_bwd<128><<<dim3(B, 2), 128, 0, st>>>(d_out, out, gates, wpT, dgi, dgh, hprev, T);
    else {
        sed_set_error("generic GRU backward: unsupported hidden size %d", H);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}
