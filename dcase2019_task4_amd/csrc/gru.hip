// gru.hip - bidirectional GRU recurrence (hidden 64), forward and backward through time.
//
// Reference op: nn.GRU(n_in, 64, bidirectional=True, batch_first=True) inside BidirectionalGRU
// (baseline/models/RNN.py:12-16); torch gate order (r, z, n), h0 = 0:
//   r = sig(gi_r + gh_r); z = sig(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1-z) n + z h
// with gi = W_ih x + b_ih (one batched GEMM over all time steps, gemm.hip) and
// gh = W_hh h + b_hh (this file).
//
// The recurrence is a serial chain of T/8 steps: latency-, not throughput-bound (measured: the step time is
// instruction latency + barrier skew, not memory).  Design, one workgroup = 2 waves per (clip, direction),
// all B x 2 chains concurrently on separate CUs:
//   * wave 0 COMPUTES ALONE: lane j owns hidden unit j and keeps the three W_hh rows (r, z, n) of that unit
//     - 192 weights - in registers for all steps.  Everything a step needs from other lanes is the 64-float
//     h vector, broadcast through LDS inside the wave: in-order LDS, NO workgroup barrier in the step loop
//     (the first version split the rows over 3 waves and paid 2 barriers + cross-wave skew per step).
//     The mat-vec is written on float2 so it compiles to v_pk_fma_f32 on natural register pairs.
//   * wave 1 does ALL global memory traffic, GRU_SB steps at a time: prefetches the next block's inputs,
//     writes the previous block's outputs from an LDS ring, hands the inputs over in LDS.  On CDNA4 `vmcnt`
//     counts loads and stores alike and they retire out of order w.r.t. each other, so a wave that both
//     loads and stores per step would drain its stores (full write latency) before every use of a load.
//   The two waves meet at ONE LDS-only barrier per block of GRU_SB steps.
// Backward is the same structure with lane j owning COLUMN j of W_hh (dh_prev = W_hh^T dgh).
// Where a step's ~1500 cycles go (forward, measured by leaving parts out, tools/kbench.py gru0_fwd): the mat-vec (96
// v_pk_fma_f32 + 16 broadcast LDS reads issued by a lone wave, ~6 cycles per instruction) 19 of 42 us, the gate
// transcendentals < 2 us, the rest is the LDS write -> read turn-around of h, the gi / history traffic and the block
// hand-over.  Tried and rejected: scalar v_fmac_f32 instead of the packed form (55 us); the mat-vec split over four
// waves on the four SIMDs (quad-lane partial dot products, h exchanged through LDS, the four waves meeting at an LDS
// step counter with release / acquire instead of s_barrier): correct, but the per-step rendezvous costs more than
// the split saves - 52 / 61 us (layer 0 / 1) against 42 / 44 us.
#include "common.h"
#include "kernels.h"
#ifdef SED_AB      // superseded by gru4.hip; kept for A/B timing builds (make EXTRA=-DSED_AB)

#define GRU_SB 8
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tanhf_fast(float x) { return 1.0f - 2.0f * rcp_fast(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// ---------------------------------------------------------------------------------------------------------
// Forward.  Waves 2 and 3 compute the input projection gi = x W_ih^T + b_ih of the NEXT block of GRU_SBF time steps on
// the MFMA (16 steps = one 16-row tile, W_ih fragments resident in their registers) while wave 0 runs the recurrence
// of the current block: the projection GEMM used to be a kernel of its own in front of the recurrence (9-16 us per
// layer on the critical path) and gi a 2.9 MB round trip through HBM per layer; now gi only ever exists in LDS.
#define GRU_SBF 16
template <int NIN>
__global__ __launch_bounds__(256) void k_gru_fwd(const float* __restrict__ x, const float* __restrict__ w_ih_f,
                                                  const float* __restrict__ w_ih_r, const float* __restrict__ b_ih_f,
                                                  const float* __restrict__ b_ih_r, const float* __restrict__ w_hh_f,
                                                  const float* __restrict__ w_hh_r, const float* __restrict__ b_hh_f,
                                                  const float* __restrict__ b_hh_r, float* __restrict__ out,
                                                  float* __restrict__ gates, int T) {
    constexpr int XS = NIN + 4;                        // x row stride: (row = lane & 15, col = 4s + lane >> 4) reads conflict free
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* hs = gsm;                                  // [64]
    float* gi_s = hs + 64;                            // [2][GRU_SBF][192]
    float* hist = gi_s + 2 * GRU_SBF * 192;           // [2][GRU_SBF][320] : h, r, z, n, gh_n
    float* xs = hist + 2 * GRU_SBF * 320;             // [2][GRU_SBF][XS]
    float* Wl = xs + 2 * GRU_SBF * XS;                // [192][68] staging of W_hh (coalesced global read)
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int role = tid >> 6;                        // 0 compute, 1 I/O, 2-3 projection GEMM
    const int l = tid & 63;
    const float* whh = dir ? w_hh_r : w_hh_f;
    const float* bhh = dir ? b_hh_r : b_hh_f;
    const int nblk = (T + GRU_SBF - 1) / GRU_SBF;
    auto t_of = [&](int step) { return dir ? (T - 1 - step) : step; };
    constexpr int XPL = NIN / 64;                     // x values per lane per step
    // x rows of block `blk` -> registers / registers -> xs[blk & 1]  (I/O wave)
    auto x_load = [&](int blk, float (&v)[GRU_SBF * XPL]) {
#pragma unroll
        for (int i = 0; i < GRU_SBF * XPL; ++i) {
            const int st = min(blk * GRU_SBF + i / XPL, T - 1);
            v[i] = x[(size_t)(b * T + t_of(st)) * NIN + 64 * (i % XPL) + l];
        }
    };
    auto x_store = [&](int blk, const float (&v)[GRU_SBF * XPL]) {
        float* d = xs + (blk & 1) * GRU_SBF * XS;
#pragma unroll
        for (int i = 0; i < GRU_SBF * XPL; ++i) d[(i / XPL) * XS + 64 * (i % XPL) + l] = v[i];
    };
    // gi of block `blk` from xs[blk & 1] -> gi_s[blk & 1]  (GEMM waves; 6 column tiles of 16 gates each)
    constexpr int KS = NIN / 4;
    const int gw = role - 2, i16 = l & 15, kq = l >> 4;
    float bw[6][KS];
    float bi[6];
    if (role >= 2) {
        const float* wih = dir ? w_ih_r : w_ih_f;
        const float* bih = dir ? b_ih_r : b_ih_f;
        // B[k = input feature][j = gate] = W_ih[gate][feature].  Which feature a lane supplies at which MFMA step is
        // free as long as A and B agree: lane group kq takes features 16s + 4kq + u at step 4s + u, so that its
        // fragments are aligned float4 loads (16 gate rows x 64 contiguous bytes per instruction) - the natural
        // 4s + kq assignment made this a scatter of 192 scalar loads per lane, ~5 us of prologue.
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int gcol = 16 * (gw * 6 + c) + i16;
#pragma unroll
            for (int s16 = 0; s16 < KS / 4; ++s16) {
                const v4f w4 = *(const v4f*)(wih + (size_t)gcol * NIN + 16 * s16 + 4 * kq);
                bw[c][4 * s16] = w4.x; bw[c][4 * s16 + 1] = w4.y; bw[c][4 * s16 + 2] = w4.z; bw[c][4 * s16 + 3] = w4.w;
            }
            bi[c] = bih[gcol];
        }
    }
    auto proj_block = [&](int blk) {
        const float* A = xs + (blk & 1) * GRU_SBF * XS + i16 * XS + 4 * kq;
        v4f acc[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s16 = 0; s16 < KS / 4; ++s16) {          // fully unrolled: bw[][] must stay in registers
            const v4f a4 = *(const v4f*)(A + 16 * s16);   // features 16s + 4kq + (0..3), same assignment as bw
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u], bw[c][4 * s16 + u], acc[c], 0, 0, 0);
        }
        float* gd = gi_s + (blk & 1) * GRU_SBF * 192;
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) gd[(4 * kq + r) * 192 + 16 * (gw * 6 + c) + i16] = acc[c][r] + bi[c];
    };

    // Prologue (10 us of fixed cost per launch, tools/gru_prologue.py, on the step's critical path four times): every wave
    // issues ITS loads first - the I/O wave the x rows of blocks 0 and 1 (both before it stores either), the projection
    // waves their W_ih fragments (above), the recurrence wave its biases - and only then do all of them stage W_hh
    // (float4 loads); the recurrence wave gathers its weight rows from LDS while the projection waves compute block 0.
    float xv0[GRU_SBF * XPL], xv1[GRU_SBF * XPL];
    float bh_r = 0.f, bh_z = 0.f, bh_n = 0.f;
    if (role == 1) {
        x_load(0, xv0);
        if (nblk > 1) x_load(1, xv1);
    } else if (role == 0) {
        bh_r = bhh[l]; bh_z = bhh[64 + l]; bh_n = bhh[128 + l];
    }
    for (int e4 = tid; e4 < 192 * 16; e4 += 256) *(v4f*)(Wl + (e4 >> 4) * 68 + 4 * (e4 & 15)) = *(const v4f*)(whh + 4 * e4);
    if (role == 1) {
        x_store(0, xv0);
        if (nblk > 1) x_store(1, xv1);
    } else if (role == 0) {
        hs[l] = 0.f;
    }
    __syncthreads();
    v2f wr[32], wz[32], wn[32];       // recurrence wave: rows l, 64+l, 128+l of W_hh
    if (role >= 2) {
        proj_block(0);
    } else if (role == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v4f a = *(const v4f*)(Wl + l * 68 + 4 * q);
            const v4f c = *(const v4f*)(Wl + (64 + l) * 68 + 4 * q);
            const v4f d = *(const v4f*)(Wl + (128 + l) * 68 + 4 * q);
            wr[2 * q] = a.xy; wr[2 * q + 1] = a.zw;
            wz[2 * q] = c.xy; wz[2 * q + 1] = c.zw;
            wn[2 * q] = d.xy; wn[2 * q + 1] = d.zw;
        }
    }
    __syncthreads();

    if (role == 1) {
        // ================================ I/O wave ==========================================================
        for (int blk = 0; blk < nblk; ++blk) {
            const int cur = blk & 1, s0 = blk * GRU_SBF;
            float nxt[GRU_SBF * XPL];
            const bool more = blk + 2 < nblk;            // x of block blk+2 (the GEMM waves work on blk+1 meanwhile)
            if (more) x_load(blk + 2, nxt);
            if (blk > 0) {      // outputs of the previous block
                const float* hp = hist + (cur ^ 1) * GRU_SBF * 320;
#pragma unroll
                for (int i = 0; i < 5 * GRU_SBF; ++i) {
                    const int s = i / 5, a = i % 5;
                    const int t = t_of(s0 - GRU_SBF + s);
                    const float v = hp[s * 320 + 64 * a + l];
                    if (a == 0) out[(size_t)(b * T + t) * 128 + dir * 64 + l] = v;
                    else if (gates) gates[((size_t)(b * T + t) * 2 + dir) * 256 + 64 * (a - 1) + l] = v;
                }
            }
            if (more) x_store(blk + 2, nxt);             // into xs[blk & 1]: block blk's rows were consumed a block ago
            lds_barrier();      // block boundary
        }
        {   // outputs of the last block
            const int blk = nblk - 1, s0 = blk * GRU_SBF, sb = T - s0;
            const float* hp = hist + (blk & 1) * GRU_SBF * 320;
            for (int i = 0; i < 5 * sb; ++i) {
                const int s = i / 5, a = i % 5;
                const int t = t_of(s0 + s);
                const float v = hp[s * 320 + 64 * a + l];
                if (a == 0) out[(size_t)(b * T + t) * 128 + dir * 64 + l] = v;
                else if (gates) gates[((size_t)(b * T + t) * 2 + dir) * 256 + 64 * (a - 1) + l] = v;
            }
        }
        return;
    }
    if (role >= 2) {
        // ================================ projection waves ==================================================
        for (int blk = 0; blk < nblk; ++blk) {
            if (blk + 1 < nblk) proj_block(blk + 1);
            lds_barrier();
        }
        return;
    }
    // ==================================== compute wave =======================================================
    asm volatile("" : "+v"(bh_r), "+v"(bh_z), "+v"(bh_n));      // pin the waits for these loads before the loop
    float hprev = 0.f;
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, s0 = blk * GRU_SBF, sb = min(GRU_SBF, T - s0);
        const float* gib = gi_s + cur * GRU_SBF * 192;
        float* hb = hist + cur * GRU_SBF * 320;
        for (int s = 0; s < sb; ++s) {
            v2f ar0 = {bh_r, 0.f}, ar1 = {0.f, 0.f}, az0 = {bh_z, 0.f}, az1 = {0.f, 0.f}, an0 = {bh_n, 0.f}, an1 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v4f h4 = *(const v4f*)(hs + 4 * q);
                ar0 = pkfma(wr[2 * q], h4.xy, ar0); ar1 = pkfma(wr[2 * q + 1], h4.zw, ar1);
                az0 = pkfma(wz[2 * q], h4.xy, az0); az1 = pkfma(wz[2 * q + 1], h4.zw, az1);
                an0 = pkfma(wn[2 * q], h4.xy, an0); an1 = pkfma(wn[2 * q + 1], h4.zw, an1);
            }
            const float gh_r = (ar0.x + ar0.y) + (ar1.x + ar1.y);
            const float gh_z = (az0.x + az0.y) + (az1.x + az1.y);
            const float ghn = (an0.x + an0.y) + (an1.x + an1.y);
            const float* gr = gib + s * 192;
            const float r = sigmoidf_fast(gr[l] + gh_r);
            const float z = sigmoidf_fast(gr[64 + l] + gh_z);
            const float nn = tanhf_fast(gr[128 + l] + r * ghn);
            const float h = (1.0f - z) * nn + z * hprev;
            float* ho = hb + s * 320;
            ho[l] = h; ho[64 + l] = r; ho[128 + l] = z; ho[192 + l] = nn; ho[256 + l] = ghn;
            hs[l] = h;          // same wave reads it back next step: in-order LDS, no barrier
            hprev = h;
        }
        lds_barrier();          // block boundary: next block's inputs are in LDS, this block's outputs may be read
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward through time.  Per step inputs: d_out, r, z, n, gh_n, h_prev (6 rows of 64); outputs: dgi (192),
// dgh (192), h_prev (64) = 7 rows of 64.
// Waves 2 and 3 compute this direction's share of the gradient w.r.t. the layer input,
//   dX_dir[t][i] = sum_g dgi[t][g] W_ih[dir][g][i],
// one block of GRU_SB time steps behind the recurrence, straight from the LDS history ring the compute wave fills
// (v_mfma_f32_16x16x4_f32, the W_ih fragments - 96 KB per direction - live in the two waves' registers).  That GEMM
// used to be a kernel of its own after the recurrence (16 us per layer on the critical path of the step); the two
// directions write separate planes which the consumer adds while loading.
#define GRU_HS 452        // history row stride: 448 + 4, so that the (row = lane & 15, col = 4s + lane >> 4) reads are conflict free
template <int NIN>
__global__ __launch_bounds__(256) void k_gru_bwd(const float* __restrict__ d_out, const float* __restrict__ d_out2,
                                                  const float* __restrict__ out, const float* __restrict__ gates,
                                                  const float* __restrict__ w_hh_f, const float* __restrict__ w_hh_r,
                                                  const float* __restrict__ w_ih_f, const float* __restrict__ w_ih_r,
                                                  float* __restrict__ dgi, float* __restrict__ dgh,
                                                  float* __restrict__ hprev_out, float* __restrict__ dx_planes, int B, int T) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* dghs = gsm;                                // [192]
    float* ops = dghs + 192;                          // [2][GRU_SB][384] : d_out, r, z, n, gh_n, h_prev
    float* hist = ops + 2 * GRU_SB * 384;             // [2][GRU_SB][GRU_HS] : dgi(192), dgh(192), h_prev(64)
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int role = tid >> 6;                        // 0 compute, 1 I/O, 2-3 dX GEMM
    const int l = tid & 63;
    const int nblk = (T + GRU_SB - 1) / GRU_SB;
    auto t_of = [&](int step) { return dir ? step : (T - 1 - step); };     // reverse of the forward order
    // operand row a (0: d_out, 1-4: r z n gh_n, 5: h_prev) of recurrence step `step`, lane l; clamped address +
    // validity, so that the loads issue unconditionally back to back
    auto op_load = [&](int step, int a, bool& ok) -> float {
        const int t = t_of(min(step, T - 1));
        const int tp = dir ? t + 1 : t - 1;
        const int tpc = min(max(tp, 0), T - 1);
        ok = (step < T) && (a != 5 || (tp >= 0 && tp < T));
        if (a == 0) {
            const size_t e = (size_t)(b * T + t) * 128 + dir * 64 + l;
            return d_out2 ? d_out[e] + d_out2[e] : d_out[e];          // upstream gradient = sum of the two direction planes
        }
        const float* p = (a < 5) ? gates + ((size_t)(b * T + t) * 2 + dir) * 256 + (a - 1) * 64 + l
                                 : out + (size_t)(b * T + tpc) * 128 + dir * 64 + l;
        return *p;
    };
    auto store_hist = [&](const float* hp, int s0, int n_steps) {
        for (int i = 0; i < 7 * n_steps; ++i) {
            const int s = i / 7, a = i % 7;
            const size_t bt = (size_t)(b * T + t_of(s0 + s)) * 2 + dir;
            const float v = hp[s * GRU_HS + 64 * a + l];
            if (a < 3) dgi[bt * 192 + 64 * a + l] = v;
            else if (a < 6) dgh[bt * 192 + 64 * (a - 3) + l] = v;
            else hprev_out[bt * 64 + l] = v;
        }
    };
    if (role == 1) {
        float first[6 * GRU_SB];
        unsigned long long okm = 0;
#pragma unroll
        for (int i = 0; i < 6 * GRU_SB; ++i) {
            bool ok;
            first[i] = op_load(i / 6, i % 6, ok);
            okm |= (ok ? 1ull : 0ull) << i;
        }
#pragma unroll
        for (int i = 0; i < 6 * GRU_SB; ++i) ops[(i / 6) * 384 + 64 * (i % 6) + l] = ((okm >> i) & 1ull) ? first[i] : 0.f;
    }
    __syncthreads();
    // (Moving this barrier behind each role's own prologue loads - the W_hh column of the recurrence wave, the W_ih
    // fragments of the dX waves - so that all of them are in flight together did not pay: fixed cost 9.0 -> 9.0 us,
    // 10-17 ns more per step; tools/gru_prologue.py.  The fixed ~9 us per launch are launch + one load round trip +
    // the fill and drain of the one-block-behind I/O and dX pipelines.)

    if (role == 1) {
        // ================================ I/O wave ==========================================================
        for (int blk = 0; blk < nblk; ++blk) {
            const int cur = blk & 1, s0 = blk * GRU_SB;
            float nxt[6 * GRU_SB];
            unsigned long long okm = 0;
            const bool more = s0 + GRU_SB < T;
            if (more) {
#pragma unroll
                for (int i = 0; i < 6 * GRU_SB; ++i) {
                    bool ok;
                    nxt[i] = op_load(s0 + GRU_SB + i / 6, i % 6, ok);
                    okm |= (ok ? 1ull : 0ull) << i;
                }
            }
            if (blk > 0) {
                const float* hp = hist + (cur ^ 1) * GRU_SB * GRU_HS;
#pragma unroll
                for (int i = 0; i < 7 * GRU_SB; ++i) {
                    const int s = i / 7, a = i % 7;
                    const size_t bt = (size_t)(b * T + t_of(s0 - GRU_SB + s)) * 2 + dir;
                    const float v = hp[s * GRU_HS + 64 * a + l];
                    if (a < 3) dgi[bt * 192 + 64 * a + l] = v;
                    else if (a < 6) dgh[bt * 192 + 64 * (a - 3) + l] = v;
                    else hprev_out[bt * 64 + l] = v;
                }
            }
            if (more) {
                float* on = ops + (cur ^ 1) * GRU_SB * 384;
#pragma unroll
                for (int i = 0; i < 6 * GRU_SB; ++i) on[(i / 6) * 384 + 64 * (i % 6) + l] = ((okm >> i) & 1ull) ? nxt[i] : 0.f;
            }
            lds_barrier();
        }
        store_hist(hist + ((nblk - 1) & 1) * GRU_SB * GRU_HS, (nblk - 1) * GRU_SB, T - (nblk - 1) * GRU_SB);
        return;
    }
    if (role >= 2) {
        // ================================ dX GEMM waves ========================================================
        // block of GRU_SB (= 8) steps x 192 gates times W_ih[dir] (192 x NIN): rows 8..15 of the 16-row MFMA tile are
        // zero; each wave owns NIN/32 column tiles and keeps their B fragments (k = gate, j = input feature) resident
        constexpr int CT = NIN / 32;                      // column tiles per wave
        const int gw = role - 2, i16 = l & 15, kq = l >> 4;
        const float* wih = dir ? w_ih_r : w_ih_f;
        float bw[CT][48];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int s4 = 0; s4 < 48; ++s4) bw[c][s4] = wih[(size_t)(4 * s4 + kq) * NIN + 16 * (gw * CT + c) + i16];
        float* plane = dx_planes + (size_t)dir * B * T * NIN;
        auto gemm_block = [&](int blk) {
            const int s0 = blk * GRU_SB, sb = min(GRU_SB, T - s0);
            const float* hp = hist + (blk & 1) * GRU_SB * GRU_HS + (i16 & (GRU_SB - 1)) * GRU_HS + kq;
            const float rowmask = (i16 < sb) ? 1.0f : 0.f;
            v4f acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 48; ++s4) {             // fully unrolled: bw[][] must stay in registers
                const float a = hp[4 * s4] * rowmask;
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[c][s4], acc[c], 0, 0, 0);
            }
            // D: lane (j = i16, rows 4*kq + r) -> time step s0 + 4*kq + r of the block (kq < 2)
            if (kq < GRU_SB / 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * kq + r;
                    if (s < sb) {
                        float* dst = plane + (size_t)(b * T + t_of(s0 + s)) * NIN + 16 * gw * CT + i16;
#pragma unroll
                        for (int c = 0; c < CT; ++c) dst[16 * c] = acc[c][r];
                    }
                }
            }
        };
        for (int blk = 0; blk < nblk; ++blk) {
            if (blk > 0) gemm_block(blk - 1);
            lds_barrier();
        }
        gemm_block(nblk - 1);
        return;
    }
    // ==================================== compute wave =======================================================
    const float* whh = dir ? w_hh_r : w_hh_f;
    v2f wt[96];   // column l of W_hh: wt[i/2][i&1] = W_hh[i][l], i = 0..191
#pragma unroll
    for (int i = 0; i < 96; ++i) {
        wt[i].x = whh[(2 * i) * 64 + l];
        wt[i].y = whh[(2 * i + 1) * 64 + l];
    }
#pragma unroll
    for (int i = 0; i < 96; ++i) asm volatile("" : "+v"(wt[i]));      // pin the load waits before the loop
    float dh_carry = 0.f;
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, s0 = blk * GRU_SB, sb = min(GRU_SB, T - s0);
        const float* ob = ops + cur * GRU_SB * 384;
        float* hb = hist + cur * GRU_SB * GRU_HS;
        for (int s = 0; s < sb; ++s) {
            const float* o = ob + s * 384;
            const float dh = o[l] + dh_carry;
            const float r = o[64 + l], z = o[128 + l], nn = o[192 + l], ghn = o[256 + l], hp = o[320 + l];
            const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
            const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dghn = dn_pre * r;
            float* ho = hb + s * GRU_HS;
            ho[l] = dr_pre; ho[64 + l] = dz_pre; ho[128 + l] = dn_pre;
            ho[192 + l] = dr_pre; ho[256 + l] = dz_pre; ho[320 + l] = dghn;
            ho[384 + l] = hp;
            dghs[l] = dr_pre; dghs[64 + l] = dz_pre; dghs[128 + l] = dghn;     // read back by this same wave
            v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 48; q += 2) {
                const v4f d0 = *(const v4f*)(dghs + 4 * q);
                const v4f d1 = *(const v4f*)(dghs + 4 * q + 4);
                a0 = pkfma(wt[2 * q], d0.xy, a0);
                a1 = pkfma(wt[2 * q + 1], d0.zw, a1);
                a2 = pkfma(wt[2 * q + 2], d1.xy, a2);
                a3 = pkfma(wt[2 * q + 3], d1.zw, a3);
            }
            dh_carry = dh * z + ((a0.x + a0.y) + (a1.x + a1.y)) + ((a2.x + a2.y) + (a3.x + a3.y));
        }
        lds_barrier();
    }
}

template <int NIN> static constexpr size_t gru_fwd_lds() { return (size_t)(64 + 2 * GRU_SBF * 192 + 2 * GRU_SBF * 320 + 2 * GRU_SBF * (NIN + 4) + 192 * 68) * sizeof(float); }
static const size_t GRU_BWD_LDS = (size_t)(192 + 2 * GRU_SB * 384 + 2 * GRU_SB * GRU_HS) * sizeof(float);

// x: the layer input [B*T][nin] (the input projection runs inside the kernel)
int launch_gru_fwd_v1(const float* x, int nin, const float* w_ih_f, const float* w_ih_r, const float* b_ih_f, const float* b_ih_r,
                   const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r, float* out, float* gates,
                   int B, int T, hipStream_t st) {
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru_fwd<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gru_fwd_lds<64>()));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru_fwd<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gru_fwd_lds<128>()));
    }
    if (nin == 128)
        k_gru_fwd<128><<<dim3(B, 2), 256, gru_fwd_lds<128>(), st>>>(x, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, T);
    else if (nin == 64)
        k_gru_fwd<64><<<dim3(B, 2), 256, gru_fwd_lds<64>(), st>>>(x, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, T);
    else {
        sed_set_error("gru forward: unsupported input width %d", nin);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// dx_planes: [2][B*T][nin] - the two directions' shares of the gradient w.r.t. the layer input (the consumer adds
// them); d_out2: optional second plane of the upstream gradient (the layer above's dx_planes + B*T*128), or null
int launch_gru_bwd_v1(const float* d_out, const float* d_out2, const float* out, const float* gates, const float* w_hh_f,
                   const float* w_hh_r, const float* w_ih_f, const float* w_ih_r, int nin, float* dgi, float* dgh, float* hprev,
                   float* dx_planes, int B, int T, hipStream_t st) {
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru_bwd<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU_BWD_LDS));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru_bwd<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU_BWD_LDS));
    }
    if (nin == 128)
        k_gru_bwd<128><<<dim3(B, 2), 256, GRU_BWD_LDS, st>>>(d_out, d_out2, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, dgi, dgh,
                                                             hprev, dx_planes, B, T);
    else if (nin == 64)
        k_gru_bwd<64><<<dim3(B, 2), 256, GRU_BWD_LDS, st>>>(d_out, d_out2, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, dgi, dgh,
                                                            hprev, dx_planes, B, T);
    else {
        sed_set_error("gru backward: unsupported input width %d", nin);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}
#endif  // SED_AB
