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

// =====================================================================================================================
// Cluster recurrence (the default for H = 256): FOUR workgroups per (clip, direction) chain, each keeping a quarter of
// W_hh in REGISTERS for the whole sequence, exchanging h (forward) / the gate gradients (backward) once per time step
// through L2.
//
// Why: the streaming kernels above pull 768 KB through one CU every time step - 10 us (forward) / 14 us (backward) per
// step measured, 5.4 of the 9.4 ms of kernel time of a wide mean-teacher step.  3H x H fp32 does not fit one CU
// (512 KB of VGPRs), but a quarter does: workgroup w of a chain owns hidden units [64 w, 64 w + 64); its thread
// (unit u = t & 63, k-quarter kq = t >> 6) holds the 3 x 64 weights W_hh[gate][64 w + u][64 kq .. 64 kq + 64) - 192
// registers, like gru.hip's lane.  Per step: every wave forms its k-quarter of the mat-vec from h in LDS (96
// v_pk_fma_f32), the four partial sums meet in LDS, wave 0 does the gate math for the workgroup's 64 units and
// PUBLISHES the 64 new h values, waves 1-3 collect the other three workgroups' values, wave 4 does all bulk global
// I/O through LDS rings (vmcnt counts loads and stores alike: a wave that waits for a load also drains its stores).
//
// Exchange = MI355X_MICROARCH.md's "granule" hand-off: one naturally aligned 8-byte {value, tag} written by ONE
// relaxed agent-scope (sc1, write-through) store and polled with relaxed agent-scope loads - no flag, no fence; the
// tag is (launch epoch << 16) | (step + 1), two slots per value so that a producer one step ahead never overwrites
// what a consumer still has to read (it cannot be two steps ahead: it needs the consumer's own value of the step in
// between).  The epoch is a per-chain device word read by all four workgroups at start and bumped by workgroup 0 at
// the end, so a replayed hipGraph never mistakes the previous launch's granules for fresh ones.
// Residency: the four workgroups of a chain spin on each other, so they must become co-resident.  They are
// consecutive in dispatch order modulo the XCD interleave (block b -> XCD b % 8: a chain's blocks are b, b + 8,
// b + 16, b + 24 of a group of 32, i.e. the SAME XCD - a speed choice only), at most one chain per concurrently
// running launch is ever partially resident, every other resident chain is complete and finishes, so the frontier
// always advances; every spin is bounded (GCL_SPIN_LIMIT) and raises *err instead of hanging the GPU.
// =====================================================================================================================
#define GCL_WG 4
#define GCL_UNITS 64              // hidden units per workgroup (H / GCL_WG at H = 256)
#define GCL_THREADS 320           // 4 mat-vec waves + 1 I/O wave
#define GCL_RING 4                // depth of the I/O rings (time steps)
#define GCL_SPIN_LIMIT (1 << 22)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f gpkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void gcl_publish(unsigned long long* p, float v, uint32_t tag) {
    const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float gcl_collect(const unsigned long long* p, uint32_t tag, int* err) {
    unsigned long long g;
    int it = 0;
    do {
        g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(g >> 32) == tag) break;
        __builtin_amdgcn_s_sleep(1);
    } while (++it < GCL_SPIN_LIMIT);
    if (it >= GCL_SPIN_LIMIT) atomicAdd(err, 1);          // sticky counter: read by the host (train.py health check)
    return __uint_as_float((uint32_t)g);
}
// chain / member of this block: blocks b, b + 8, b + 16, b + 24 of a group of 32 form one chain (same XCD)
__device__ __forceinline__ void gcl_ids(int& chain, int& wg) {
    const int b = blockIdx.x, g32 = b >> 5, r = b & 31;
    wg = r >> 3;
    chain = g32 * 8 + (r & 7);
}

// xch: [chains][2 slots][H] granules; epoch: [chains]
__global__ __launch_bounds__(GCL_THREADS) void k_gclu_fwd(const float* __restrict__ gi, const float* __restrict__ w_hh_f,
                                                           const float* __restrict__ w_hh_r, const float* __restrict__ b_hh_f,
                                                           const float* __restrict__ b_hh_r, float* __restrict__ out,
                                                           float* __restrict__ gates, unsigned long long* __restrict__ xch,
                                                           unsigned int* __restrict__ epoch, int* __restrict__ err, int B, int T) {
    constexpr int H = 256, U = GCL_UNITS;
    __shared__ __attribute__((aligned(16))) float hs[H];
    __shared__ float red[4][3][U];
    __shared__ float gis[GCL_RING][3][U];
    __shared__ float outs[GCL_RING][5][U];
    int chain, wg;
    gcl_ids(chain, wg);
    if (chain >= 2 * B) return;
    const int b = chain >> 1, dir = chain & 1, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* whh = dir ? w_hh_r : w_hh_f;
    const float* bhh = dir ? b_hh_r : b_hh_f;
    unsigned long long* xc = xch + (size_t)chain * 2 * H;
    const uint32_t ep = __hip_atomic_load(&epoch[chain], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 16;
    auto t_of = [&](int s) { return dir ? (T - 1 - s) : s; };
    // ---- prologue -------------------------------------------------------------------------------------------------
    v2f wr[32], wz[32], wn[32];
    float bh_r = 0.f, bh_z = 0.f, bh_n = 0.f;
    const int u = lane, kq = wave;                         // mat-vec waves: unit u of this workgroup, k-quarter kq
    if (wave < 4) {
        const int row = U * wg + u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v4f a = *(const v4f*)(whh + (size_t)row * H + 64 * kq + 4 * q);
            const v4f c = *(const v4f*)(whh + (size_t)(H + row) * H + 64 * kq + 4 * q);
            const v4f d = *(const v4f*)(whh + (size_t)(2 * H + row) * H + 64 * kq + 4 * q);
            wr[2 * q] = a.xy; wr[2 * q + 1] = a.zw;
            wz[2 * q] = c.xy; wz[2 * q + 1] = c.zw;
            wn[2 * q] = d.xy; wn[2 * q + 1] = d.zw;
        }
        if (wave == 0) { bh_r = bhh[row]; bh_z = bhh[H + row]; bh_n = bhh[2 * H + row]; }
        hs[tid] = 0.f;
    }
    // I/O wave: gi of steps 0 .. GCL_RING - 2 straight into the ring, step GCL_RING - 1 in flight in registers
    float gnext[3] = {0.f, 0.f, 0.f};
    auto gi_load = [&](int s, float (&v)[3]) {
        const int sc = min(s, T - 1);
        const float* g = gi + ((size_t)(b * T + t_of(sc)) * 2 + dir) * 3 * H + U * wg + lane;
        v[0] = g[0]; v[1] = g[H]; v[2] = g[2 * H];
    };
    if (wave == 4) {
        for (int s = 0; s < GCL_RING - 1; ++s) {
            float v[3];
            gi_load(s, v);
            gis[s][0][lane] = v[0]; gis[s][1][lane] = v[1]; gis[s][2][lane] = v[2];
        }
        gi_load(GCL_RING - 1, gnext);
    }
    float hprev = 0.f;
    __syncthreads();
    // ---- time steps -------------------------------------------------------------------------------------------------
    for (int s = 0; s < T; ++s) {
        const uint32_t tag = ep | (uint32_t)(s + 1);
        const int slot = s & 1;
        if (wave < 4) {
            // (a) this wave's k-quarter of gh = W_hh h for the workgroup's 64 units
            v2f ar0 = {0.f, 0.f}, ar1 = {0.f, 0.f}, az0 = {0.f, 0.f}, az1 = {0.f, 0.f}, an0 = {0.f, 0.f}, an1 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v4f h4 = *(const v4f*)(hs + 64 * kq + 4 * q);
                ar0 = gpkfma(wr[2 * q], h4.xy, ar0); ar1 = gpkfma(wr[2 * q + 1], h4.zw, ar1);
                az0 = gpkfma(wz[2 * q], h4.xy, az0); az1 = gpkfma(wz[2 * q + 1], h4.zw, az1);
                an0 = gpkfma(wn[2 * q], h4.xy, an0); an1 = gpkfma(wn[2 * q + 1], h4.zw, an1);
            }
            red[kq][0][u] = (ar0.x + ar0.y) + (ar1.x + ar1.y);
            red[kq][1][u] = (az0.x + az0.y) + (az1.x + az1.y);
            red[kq][2][u] = (an0.x + an0.y) + (an1.x + an1.y);
        }
        lds_barrier();                                     // partial sums complete; every wave has read hs
        if (wave == 0) {
            // (b) gates of this workgroup's units; publish h first, everything else afterwards
            const float gh_r = bh_r + ((red[0][0][u] + red[1][0][u]) + (red[2][0][u] + red[3][0][u]));
            const float gh_z = bh_z + ((red[0][1][u] + red[1][1][u]) + (red[2][1][u] + red[3][1][u]));
            const float ghn = bh_n + ((red[0][2][u] + red[1][2][u]) + (red[2][2][u] + red[3][2][u]));
            const int rs = s % GCL_RING;
            const float r = sigmoidf_fast(gis[rs][0][u] + gh_r);
            const float z = sigmoidf_fast(gis[rs][1][u] + gh_z);
            const float nn = gtanhf_fast(gis[rs][2][u] + r * ghn);
            const float h = (1.0f - z) * nn + z * hprev;
            if (s + 1 < T) gcl_publish(xc + (size_t)slot * H + U * wg + u, h, tag);
            hs[U * wg + u] = h;
            hprev = h;
            outs[rs][0][u] = h; outs[rs][1][u] = r; outs[rs][2][u] = z; outs[rs][3][u] = nn; outs[rs][4][u] = ghn;
        } else if (wave < 4) {
            // (c) the other three workgroups' units of h(s + 1)
            if (s + 1 < T) {
                const int src = (wg + wave) & 3;
                hs[U * src + lane] = gcl_collect(xc + (size_t)slot * H + U * src + lane, tag, err);
            }
        } else {
            // (d) I/O wave: outputs of step s - 1 (its ring slot was filled before the barrier above), gi of step s + RING - 1
            if (s > 0) {
                const int ps = (s - 1) % GCL_RING;
                const size_t bt = (size_t)(b * T + t_of(s - 1));
                out[bt * 2 * H + dir * H + U * wg + lane] = outs[ps][0][lane];
                if (gates) {
                    float* gt = gates + (bt * 2 + dir) * 4 * H + U * wg + lane;
                    gt[0] = outs[ps][1][lane]; gt[H] = outs[ps][2][lane]; gt[2 * H] = outs[ps][3][lane]; gt[3 * H] = outs[ps][4][lane];
                }
            }
            const int fs = (s + GCL_RING - 1) % GCL_RING;  // == (s - 1) % RING: the slot step s - 1 just vacated
            gis[fs][0][lane] = gnext[0]; gis[fs][1][lane] = gnext[1]; gis[fs][2][lane] = gnext[2];
            gi_load(s + GCL_RING, gnext);
        }
        lds_barrier();                                     // hs holds h(s + 1); rings advanced
    }
    if (wave == 4) {
        const int ps = (T - 1) % GCL_RING;
        const size_t bt = (size_t)(b * T + t_of(T - 1));
        out[bt * 2 * H + dir * H + U * wg + lane] = outs[ps][0][lane];
        if (gates) {
            float* gt = gates + (bt * 2 + dir) * 4 * H + U * wg + lane;
            gt[0] = outs[ps][1][lane]; gt[H] = outs[ps][2][lane]; gt[2 * H] = outs[ps][3][lane]; gt[3 * H] = outs[ps][4][lane];
        }
    }
    if (wg == 0 && tid == 0) epoch[chain] = (ep >> 16) + 1;
}

// Backward through time, same cluster: workgroup w owns units (columns) [64 w, 64 w + 64).  Per step wave 0 turns
// dh = d_out + carry into the gate gradients of its 64 units and publishes (dr, dz, dgh_n) - the vector every workgroup
// needs in full - waves 1-3 collect the other 3 x 192 values, then every wave forms its g-quarter of
//   dh_prev[j] = dh[j] z[j] + sum_g dgh[g] W_hh[g][j]
// with column j of W_hh (its 192-row quarter) in registers.  xch: [chains][2 slots][3H] granules.
__global__ __launch_bounds__(GCL_THREADS) void k_gclu_bwd(const float* __restrict__ d_out, const float* __restrict__ out,
                                                           const float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                           const float* __restrict__ w_hh_r, float* __restrict__ dgi,
                                                           float* __restrict__ dgh, float* __restrict__ hprev_out,
                                                           unsigned long long* __restrict__ xch, unsigned int* __restrict__ epoch,
                                                           int* __restrict__ err, int B, int T) {
    constexpr int H = 256, U = GCL_UNITS;
    __shared__ __attribute__((aligned(16))) float ds[3 * H];
    __shared__ float red[4][U];
    __shared__ float ins[GCL_RING][6][U];                  // d_out, r, z, n, gh_n, h_prev of the workgroup's units
    __shared__ float outs[GCL_RING][5][U];                 // dr, dz, dn, dgh_n, h_prev
    int chain, wg;
    gcl_ids(chain, wg);
    if (chain >= 2 * B) return;
    const int b = chain >> 1, dir = chain & 1, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* whh = dir ? w_hh_r : w_hh_f;
    unsigned long long* xc = xch + (size_t)chain * 2 * 3 * H;
    const uint32_t ep = __hip_atomic_load(&epoch[chain], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 16;
    auto t_of = [&](int s) { return dir ? s : (T - 1 - s); };          // reverse of the forward order
    v2f wt[96];                                            // W_hh[192 gq + 2 i + {0, 1}][column]
    const int col = U * wg + lane, gq = wave;
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            wt[i].x = whh[(size_t)(192 * gq + 2 * i) * H + col];
            wt[i].y = whh[(size_t)(192 * gq + 2 * i + 1) * H + col];
        }
    }
    float inext[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto in_load = [&](int s, float (&v)[6]) {
        const int sc = min(s, T - 1), t = t_of(sc);
        const int tp = dir ? t + 1 : t - 1;
        const size_t bt = (size_t)(b * T + t);
        const float* gt = gates + (bt * 2 + dir) * 4 * H + col;
        v[0] = d_out[bt * 2 * H + dir * H + col];
        v[1] = gt[0]; v[2] = gt[H]; v[3] = gt[2 * H]; v[4] = gt[3 * H];
        v[5] = (tp >= 0 && tp < T) ? out[(size_t)(b * T + tp) * 2 * H + dir * H + col] : 0.f;
    };
    if (wave == 4) {
        for (int s = 0; s < GCL_RING - 1; ++s) {
            float v[6];
            in_load(s, v);
#pragma unroll
            for (int a = 0; a < 6; ++a) ins[s][a][lane] = v[a];
        }
        in_load(GCL_RING - 1, inext);
    }
    float carry = 0.f, carry_base = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const uint32_t tag = ep | (uint32_t)(s + 1);
        const int slot = s & 1, rs = s % GCL_RING;
        if (wave == 0) {
            const float* iv = &ins[rs][0][lane];
            const float dh = iv[0] + carry;
            const float r = iv[U], z = iv[2 * U], nn = iv[3 * U], ghn = iv[4 * U], hp = iv[5 * U];
            const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
            const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dghn = dn_pre * r;
            if (s + 1 < T) {
                unsigned long long* px = xc + (size_t)slot * 3 * H + col;
                gcl_publish(px, dr_pre, tag); gcl_publish(px + H, dz_pre, tag); gcl_publish(px + 2 * H, dghn, tag);
            }
            ds[col] = dr_pre; ds[H + col] = dz_pre; ds[2 * H + col] = dghn;
            carry_base = dh * z;
            outs[rs][0][lane] = dr_pre; outs[rs][1][lane] = dz_pre; outs[rs][2][lane] = dn_pre; outs[rs][3][lane] = dghn;
            outs[rs][4][lane] = hp;
        } else if (wave < 4) {
            if (s + 1 < T) {
                const int src = (wg + wave) & 3;
                const unsigned long long* px = xc + (size_t)slot * 3 * H + U * src + lane;
                // the three granules of a unit are polled TOGETHER (three loads in flight per round trip, not three round trips)
                unsigned long long g0, g1, g2;
                int it = 0;
                do {
                    g0 = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g1 = __hip_atomic_load(px + H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g2 = __hip_atomic_load(px + 2 * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag && (uint32_t)(g2 >> 32) == tag) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++it < GCL_SPIN_LIMIT);
                if (it >= GCL_SPIN_LIMIT) atomicAdd(err, 1);
                ds[U * src + lane] = __uint_as_float((uint32_t)g0); ds[H + U * src + lane] = __uint_as_float((uint32_t)g1);
                ds[2 * H + U * src + lane] = __uint_as_float((uint32_t)g2);
            }
        } else {
            if (s > 0) {
                const int ps = (s - 1) % GCL_RING;
                const size_t bt = (size_t)(b * T + t_of(s - 1)) * 2 + dir;
                float* gi_o = dgi + bt * 3 * H + col;
                float* gh_o = dgh + bt * 3 * H + col;
                const float dr = outs[ps][0][lane], dz = outs[ps][1][lane], dn = outs[ps][2][lane], dg = outs[ps][3][lane];
                gi_o[0] = dr; gi_o[H] = dz; gi_o[2 * H] = dn;
                gh_o[0] = dr; gh_o[H] = dz; gh_o[2 * H] = dg;
                hprev_out[bt * H + col] = outs[ps][4][lane];
            }
            const int fs = (s + GCL_RING - 1) % GCL_RING;
#pragma unroll
            for (int a = 0; a < 6; ++a) ins[fs][a][lane] = inext[a];
            in_load(s + GCL_RING, inext);
        }
        lds_barrier();                                     // ds holds all 3H gate gradients of this step
        if (wave < 4 && s + 1 < T) {
            v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 48; q += 2) {
                const v4f d0 = *(const v4f*)(ds + 192 * gq + 4 * q);
                const v4f d1 = *(const v4f*)(ds + 192 * gq + 4 * q + 4);
                a0 = gpkfma(wt[2 * q], d0.xy, a0);
                a1 = gpkfma(wt[2 * q + 1], d0.zw, a1);
                a2 = gpkfma(wt[2 * q + 2], d1.xy, a2);
                a3 = gpkfma(wt[2 * q + 3], d1.zw, a3);
            }
            red[gq][lane] = ((a0.x + a0.y) + (a1.x + a1.y)) + ((a2.x + a2.y) + (a3.x + a3.y));
        }
        lds_barrier();
        if (wave == 0) carry = carry_base + ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    }
    if (wave == 4) {
        const int ps = (T - 1) % GCL_RING;
        const size_t bt = (size_t)(b * T + t_of(T - 1)) * 2 + dir;
        float* gi_o = dgi + bt * 3 * H + col;
        float* gh_o = dgh + bt * 3 * H + col;
        const float dr = outs[ps][0][lane], dz = outs[ps][1][lane], dn = outs[ps][2][lane], dg = outs[ps][3][lane];
        gi_o[0] = dr; gi_o[H] = dz; gi_o[2 * H] = dn;
        gh_o[0] = dr; gh_o[H] = dz; gh_o[2 * H] = dg;
        hprev_out[bt * H + col] = outs[ps][4][lane];
    }
    if (wg == 0 && tid == 0) epoch[chain] = (ep >> 16) + 1;
}

size_t gclu_xch_bytes(int B, int H, int bwd) { return (size_t)2 * B * 2 * (bwd ? 3 * H : H) * sizeof(unsigned long long); }

int launch_gclu_fwd(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r, float* out,
                    float* gates, void* xch, unsigned int* epoch, int* err, int B, int T, hipStream_t st) {
    const int chains = 2 * B, blocks = ((chains + 7) / 8) * 32;
    k_gclu_fwd<<<blocks, GCL_THREADS, 0, st>>>(gi, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, (unsigned long long*)xch, epoch, err, B, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
int launch_gclu_bwd(const float* d_out, const float* out, const float* gates, const float* w_hh_f, const float* w_hh_r, float* dgi,
                    float* dgh, float* hprev, void* xch, unsigned int* epoch, int* err, int B, int T, hipStream_t st) {
    const int chains = 2 * B, blocks = ((chains + 7) / 8) * 32;
    k_gclu_bwd<<<blocks, GCL_THREADS, 0, st>>>(d_out, out, gates, w_hh_f, w_hh_r, dgi, dgh, hprev, (unsigned long long*)xch, epoch, err, B, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
