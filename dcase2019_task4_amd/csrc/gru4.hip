// gru4.hip - bidirectional GRU recurrence (hidden 64): the mat-vec of one chain split over the FOUR SIMDs of a CU.
//
// Reference op: nn.GRU(n_in, 64, bidirectional=True, batch_first=True) inside BidirectionalGRU
// (baseline/models/RNN.py:12-16); torch gate order (r, z, n), h0 = 0:
//   r = sig(gi_r + gh_r); z = sig(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1-z) n + z h
//
// A chain (clip, direction) is T/8 = 78 serial steps of a 192 x 64 mat-vec + gate math: latency-, not throughput-bound.
// Round 1 (gru.hip, now only built with SED_AB) gave the whole mat-vec to ONE wave: 394 / 409 ns per step forward (layer
// 0 / 1), 563 / 570 ns backward (tools/gru_prologue.py: time = fixed + per-step x steps).  Here one workgroup = 9 (forward)
// or 10 (backward) waves per chain:
//   * waves 0-3, one per SIMD, "recurrence".  Forward: wave w owns hidden units 16w .. 16w+15; lane (u = lane >> 2,
//     kq = lane & 3) holds the three W_hh rows (r, z, n) of unit j = 16w + u restricted to the k-quarter [16 kq, 16 kq + 16):
//     48 weights, 24 v_pk_fma_f32 per step.  The quarter sums meet inside the quad with two DPP adds (quad_perm
//     butterflies, every lane of the quad ends with the bit-identical sum), all four lanes do the gate math of their unit,
//     and the quad writes h, r, z, n (one ds_write: lane kq picks its value) and gh_n into the step's row of the LDS history
//     ring.  The next step reads h straight from that row (4 ds_read_b128 per lane).
//   * ONE s_barrier per time step is the rendezvous of all waves (the LDS step counter with release / acquire polling
//     that round 1 tried for a four-wave split cost more than it saved; a hardware barrier is 19 - 24 ns).
//   * "I/O" wave(s): all global memory traffic, one block of 16 steps ahead (loads) / behind (stores), a slice per step so
//     that they reach every barrier early.  (vmcnt counts loads and stores alike: a wave that loads and stores per step
//     would drain its stores before every use of a load.)
//   * four "GEMM" waves.  Forward: gi = x W_ih^T + b_ih of the NEXT block on the MFMA (16 steps = one 16-row tile, the
//     W_ih fragments resident in registers), a slice of k-steps per time step; backward: this direction's share of the
//     gradient w.r.t. the layer input, dX_dir[t][i] = sum_g dgi[t][g] W_ih[dir][g][i], one block BEHIND the recurrence,
//     straight from the LDS history ring.  gi / dgi never make an extra HBM round trip for these GEMMs.
// Backward through time: lane (kq = lane >> 4, u = lane & 15) holds column j = 16w + u of W_hh restricted to the gate
// quarter [48 kq, 48 kq + 48): dh_prev[j] = dh[j] z[j] + sum_i W_hh[i][j] dg[i], dg = (dr_pre, dz_pre, dgh_n) of the
// previous step.  The quarter's 48 dg values sit in the ROW's registers (3 per lane, 3 ds_read_b32) and reach the lanes
// through DPP row rotations (v_fmac_f32_dpp row_ror); the four rows meet through v_permlane32_swap / v_permlane16_swap.
//
// Where a step goes (tools/ubench/step_latency.cpp on this box, 48 workgroups; ns): ds_write + s_waitcnt lgkmcnt(0) 75-95,
// s_barrier 19-24, 4 ds_read_b128 + wait 55, 24 v_pk_fma_f32 + pair / quad sums 100 (v_pk_fma_f32 issues at HALF the rate of
// v_fma_f32 here: 2.0 vs 1.1 ns - same flops), exp / rcp gate chain 58: sum 317, measured 320 - 360 forward.  The LDS
// publish -> barrier -> read turn-around (160 ns) is what a multi-wave split pays per step and what bounds this design;
// leave-one-out builds (tools/build_variant.sh, the G4_EXP_* knobs below - results are garbage with any of them set) put
// the co-resident GEMM waves at 35 - 65 ns per step (their MFMAs share the SIMDs' issue with the recurrence waves) and the
// I/O slices at 15 - 20 ns.  Measured now: forward 320 / 360 ns per step + 6.0 / 7.3 us fixed, backward 390 / 420 ns + 9 us.
#include "common.h"
#include "kernels.h"
#include "hfuse.h"
#include <type_traits>
SED_TS_DEFINE(gru4)
#ifdef SED_TS
#define TSW(k, t) do { if (threadIdx.x == (t) && blockIdx.y == 0 && blockIdx.x < 1024) g_ts[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define TSW(k, t) do { } while (0)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define G4_SB 16                 // time steps per block = rows of one 16x16x4 MFMA tile
#define G4_THREADS 576           // forward: 4 recurrence + 1 I/O + 4 GEMM waves
#define G4B_THREADS 640          // backward: 4 recurrence + 2 I/O + 4 GEMM waves
#define G4_HS 452                // backward history row stride: 448 + 4 (conflict-free (row = lane & 15, col = 4s + lane >> 4) reads)

__device__ __forceinline__ float g4_tanh(float x) { return 1.0f - 2.0f * rcp_fast(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ v2f g4_pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// sum over the four lanes of a quad; every lane gets the same bits (each level adds the same two numbers, commuted)
__device__ __forceinline__ float g4_quad_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    return x;
}

// sum over the four 16-lane rows (lanes l, l ^ 16, l ^ 32, l ^ 48) with the gfx950 row swaps; every lane gets the same bits
__device__ __forceinline__ float g4_row4_sum(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// acc += rot_S(v) * w, rot_S = DPP row_ror:S (the 16 lanes of a row see each other's registers without LDS).  Inline asm:
// the compiler does not fold a DPP move into v_fmac_f32 here (it emits v_mov_b32_dpp + v_fmac_f32, twice the instructions).
// The first rotation of a chain carries the two wait states a DPP read needs after a VALU write of its source (the
// hazard recognizer does not look inside inline asm).
template <int S> __device__ __forceinline__ void g4_fmac_ror(float& acc, float v, float w) {
    if constexpr (S == 0) acc = __builtin_fmaf(v, w, acc);
    else if constexpr (S == 1) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w));
    else asm("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w), "n"(S));
}
// which lane of the row a lane reads under row_ror:S - asked of the hardware, not assumed
template <int S> __device__ __forceinline__ int g4_ror_src(int u) {
    if constexpr (S == 0) return u;
    else return __builtin_amdgcn_update_dpp(0, u, 0x120 + S, 0xF, 0xF, true);
}
template <int S> struct G4Rot {
    // backward: wt[16 m + S] = W_hh[48 kq + 16 m + src_S(u)][j]
    static __device__ __forceinline__ void load_bwd(float (&wt)[48], const float* __restrict__ whh, int kq, int u, int j) {
        const int su = g4_ror_src<S>(u);
#pragma unroll
        for (int m = 0; m < 3; ++m) wt[16 * m + S] = whh[(size_t)(48 * kq + 16 * m + su) * 64 + j];
        if constexpr (S < 15) G4Rot<S + 1>::load_bwd(wt, whh, kq, u, j);
    }
    static __device__ __forceinline__ void dot3(float (&acc)[3], const float (&v)[3], const float (&wt)[48]) {
        g4_fmac_ror<S>(acc[0], v[0], wt[S]);
        g4_fmac_ror<S>(acc[1], v[1], wt[16 + S]);
        g4_fmac_ror<S>(acc[2], v[2], wt[32 + S]);
        if constexpr (S < 15) G4Rot<S + 1>::dot3(acc, v, wt);
    }
};

// ---------------------------------------------------------------------------------------------------------
// Forward.
template <int NIN>
__global__ __launch_bounds__(G4_THREADS) void k_gru4_fwd(const float* __restrict__ x, const float* __restrict__ w_ih_f,
                                                          const float* __restrict__ w_ih_r, const float* __restrict__ b_ih_f,
                                                          const float* __restrict__ b_ih_r, const float* __restrict__ w_hh_f,
                                                          const float* __restrict__ w_hh_r, const float* __restrict__ b_hh_f,
                                                          const float* __restrict__ b_hh_r, float* __restrict__ out,
                                                          float* __restrict__ gates, int T) {
    constexpr int XS = NIN + 4;                        // x row stride: (row = lane & 15, col = 4s + lane >> 4) reads conflict free
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* zero = gsm;                                // [64] h before the first step
    float* gi_s = zero + 64;                          // [2][G4_SB][192]
    float* hist = gi_s + 2 * G4_SB * 192;             // [2][G4_SB][320] : h, r, z, n, gh_n
    float* xs = hist + 2 * G4_SB * 320;               // [2][G4_SB][XS]
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int role = tid >> 6;                        // 0-3 recurrence, 4 I/O, 5-8 projection GEMM
    const int l = tid & 63;
    const int nblk = (T + G4_SB - 1) / G4_SB;
    auto t_of = [&](int step) { return dir ? (T - 1 - step) : step; };
    constexpr int XPL = NIN / 64;                     // x values per lane per step

    if (role == 4) {
        // ================================ I/O wave ==========================================================
        auto x_load = [&](int blk, float (&v)[G4_SB * XPL]) {
#pragma unroll
            for (int i = 0; i < G4_SB * XPL; ++i) {
                const int st = min(blk * G4_SB + i / XPL, T - 1);
                v[i] = x[(size_t)(b * T + t_of(st)) * NIN + 64 * (i % XPL) + l];
            }
        };
        auto x_store = [&](int blk, const float (&v)[G4_SB * XPL]) {
            float* d = xs + (blk & 1) * G4_SB * XS;
#pragma unroll
            for (int i = 0; i < G4_SB * XPL; ++i) d[(i / XPL) * XS + 64 * (i % XPL) + l] = v[i];
        };
        // history row s of block `blk` -> out / gates
        auto put_row = [&](int blk, int s) {
            const float* hp = hist + (blk & 1) * G4_SB * 320 + s * 320;
            const int t = t_of(blk * G4_SB + s);
            float v[5];
#pragma unroll
            for (int a = 0; a < 5; ++a) v[a] = hp[64 * a + l];
            out[(size_t)(b * T + t) * 128 + dir * 64 + l] = v[0];
            if (gates) {
#pragma unroll
                for (int a = 1; a < 5; ++a) gates[((size_t)(b * T + t) * 2 + dir) * 256 + 64 * (a - 1) + l] = v[a];
            }
        };
        {
            float xv0[G4_SB * XPL], xv1[G4_SB * XPL];
            x_load(0, xv0);
            if (nblk > 1) x_load(1, xv1);
            x_store(0, xv0);
            if (nblk > 1) x_store(1, xv1);
        }
        __syncthreads();
        __syncthreads();
        for (int blk = 0; blk < nblk; ++blk) {
            const int sb = min(G4_SB, T - blk * G4_SB);
            float nxt[G4_SB * XPL];
            const bool more = blk + 2 < nblk;            // x of block blk+2 (the GEMM waves work on blk+1 meanwhile)
            if (more) x_load(blk + 2, nxt);
            for (int s = 0; s < sb; ++s) {
#ifndef G4_EXP_NOIO
                if (blk > 0) put_row(blk - 1, s);
#endif
                if (s == sb - 1 && more) x_store(blk + 2, nxt);   // into xs[blk & 1]: block blk's rows were consumed a block ago
                lds_barrier();
            }
        }
        {   // what the per-step slices did not reach: the rest of the block before the last, and the last block
            const int blk = nblk - 1, sb = T - blk * G4_SB;
            if (blk > 0)
                for (int s = sb; s < G4_SB; ++s) put_row(blk - 1, s);
            for (int s = 0; s < sb; ++s) put_row(blk, s);
        }
        return;
    }
    if (role >= 5) {
        // ================================ projection waves ==================================================
        // 12 column tiles of 16 gates; wave pw owns tiles 3 pw .. 3 pw + 2 and keeps their B fragments resident.
        // B[k = input feature][j = gate] = W_ih[gate][feature].  Which feature a lane supplies at which MFMA step is free
        // as long as A and B agree: lane group kq takes features 16 s + 4 kq + u at step 4 s + u, so that its fragments are
        // aligned float4 loads (16 gate rows x 64 contiguous bytes per instruction).
        constexpr int KS = NIN / 4;                       // MFMA k-steps per block
        constexpr int KPS = KS / G4_SB;                   // k-steps per time step
        const int pw = role - 5, i16 = l & 15, kq = l >> 4;
        const float* wih = dir ? w_ih_r : w_ih_f;
        const float* bih = dir ? b_ih_r : b_ih_f;
        float bw[3][KS];
        float bi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int gcol = 16 * (pw * 3 + c) + i16;
#pragma unroll
            for (int s16 = 0; s16 < KS / 4; ++s16) {
                const v4f w4 = *(const v4f*)(wih + (size_t)gcol * NIN + 16 * s16 + 4 * kq);
                bw[c][4 * s16] = w4.x; bw[c][4 * s16 + 1] = w4.y; bw[c][4 * s16 + 2] = w4.z; bw[c][4 * s16 + 3] = w4.w;
            }
            bi[c] = bih[gcol];
        }
        // gi of block `blk` from xs[blk & 1] -> gi_s[blk & 1]; with sync: 16 barriers, one per time step of the block that
        // is running meanwhile - the k-steps are spread one slice AHEAD of the time steps so that the last slice is only the
        // write of the result
        auto proj_block = [&](int blk, bool sync) {
            const float* A = xs + (blk & 1) * G4_SB * XS + i16 * XS + 4 * kq;
            v4f acc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
            v4f a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                 // fully unrolled: bw[][] must stay in registers
                if ((ks & 3) == 0) a4 = *(const v4f*)(A + 4 * ks);   // features 16 (ks/4) + 4 kq + (0..3), as bw
#ifndef G4_EXP_NOPROJ
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[ks & 3], bw[c][ks], acc[c], 0, 0, 0);
#endif
                if (sync && (ks + 1) % KPS == 0 && ks + 1 >= 2 * KPS) lds_barrier();          // 15 of them
            }
            float* gd = gi_s + (blk & 1) * G4_SB * 192;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) gd[(4 * kq + r) * 192 + 16 * (pw * 3 + c) + i16] = acc[c][r] + bi[c];
            if (sync) lds_barrier();                                                        // the 16th
        };
        __syncthreads();
        proj_block(0, false);
        __syncthreads();
        for (int blk = 0; blk < nblk; ++blk) {
            const int sb = min(G4_SB, T - blk * G4_SB);
            if (blk + 1 < nblk) proj_block(blk + 1, true);      // blk is a full block here
            else
                for (int s = 0; s < sb; ++s) lds_barrier();
        }
        return;
    }
    // ==================================== recurrence waves ===================================================
    const int w = role, u = l >> 2, kq = l & 3, j = 16 * w + u;
    const float* whh = dir ? w_hh_r : w_hh_f;
    const float* bhh = dir ? b_hh_r : b_hh_f;
    v2f wr[8], wz[8], wn[8];          // rows j, 64 + j, 128 + j of W_hh, columns 16 kq .. 16 kq + 15
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const v4f a = *(const v4f*)(whh + (size_t)j * 64 + 16 * kq + 4 * q);
        const v4f c = *(const v4f*)(whh + (size_t)(64 + j) * 64 + 16 * kq + 4 * q);
        const v4f d = *(const v4f*)(whh + (size_t)(128 + j) * 64 + 16 * kq + 4 * q);
        wr[2 * q] = a.xy; wr[2 * q + 1] = a.zw;
        wz[2 * q] = c.xy; wz[2 * q + 1] = c.zw;
        wn[2 * q] = d.xy; wn[2 * q + 1] = d.zw;
    }
    float bh_r = bhh[j], bh_z = bhh[64 + j], bh_n = bhh[128 + j];
    if (w == 0) zero[l] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(wr[q]), "+v"(wz[q]), "+v"(wn[q]));      // pin the load waits before the loop
    asm volatile("" : "+v"(bh_r), "+v"(bh_z), "+v"(bh_n));
    __syncthreads();
#ifndef G4_EXP_NOPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const float* hrow = zero;
    float hprev = 0.f;
    const int wofs = 64 * kq + j;     // lane kq writes field kq of the row: h, r, z, n
    const bool is0 = kq == 0, is1 = kq == 1, is2 = kq == 2, is3 = kq == 3;
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, sb = min(G4_SB, T - blk * G4_SB);
        const float* gib = gi_s + cur * G4_SB * 192;
        float* hb = hist + cur * G4_SB * 320;
        float gi_r = gib[j], gi_z = gib[64 + j], gi_n = gib[128 + j];
        for (int s = 0; s < sb; ++s) {
            v4f h4[4];
#ifdef G4_EXP_NOREAD
#pragma unroll
            for (int q = 0; q < 4; ++q) h4[q] = (v4f){hprev, hprev, hprev, hprev};
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) h4[q] = *(const v4f*)(hrow + 16 * kq + 4 * q);
#endif
            // the input projections of the NEXT step now (row 15's successor is a clamped re-read): their LDS latency must
            // not sit between the mat-vec and the gate math
            const float* gr = gib + min(s + 1, G4_SB - 1) * 192;
            const float gn_r = gr[j], gn_z = gr[64 + j], gn_n = gr[128 + j];
            __builtin_amdgcn_sched_barrier(0);
            // b_hh (+ the input projection for r and z) ride in the accumulators of the quad's first lane: nothing but the
            // quad sum stands between the last FMA and the sigmoid
            v2f ar0 = {is0 ? gi_r + bh_r : 0.f, 0.f}, ar1 = {0.f, 0.f}, az0 = {is0 ? gi_z + bh_z : 0.f, 0.f}, az1 = {0.f, 0.f};
            v2f an0 = {is0 ? bh_n : 0.f, 0.f}, an1 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ar0 = g4_pkfma(wr[2 * q], h4[q].xy, ar0); ar1 = g4_pkfma(wr[2 * q + 1], h4[q].zw, ar1);
                az0 = g4_pkfma(wz[2 * q], h4[q].xy, az0); az1 = g4_pkfma(wz[2 * q + 1], h4[q].zw, az1);
                an0 = g4_pkfma(wn[2 * q], h4[q].xy, an0); an1 = g4_pkfma(wn[2 * q + 1], h4[q].zw, an1);
            }
            const v2f sr = ar0 + ar1, sz = az0 + az1, sn = an0 + an1;
            const float pre_r = g4_quad_sum(sr.x + sr.y);
            const float pre_z = g4_quad_sum(sz.x + sz.y);
            const float ghn = g4_quad_sum(sn.x + sn.y);
#ifdef G4_EXP_NOGATE
            const float r = pre_r, z = pre_z, nn = gi_n + r * ghn;
#else
            const float r = sigmoidf_fast(pre_r);
            const float z = sigmoidf_fast(pre_z);
            const float nn = g4_tanh(gi_n + r * ghn);
#endif
            const float h = (1.0f - z) * nn + z * hprev;
            float* ho = hb + s * 320;
            float v = h;
            v = is1 ? r : v;
            v = is2 ? z : v;
            v = is3 ? nn : v;
            ho[wofs] = v;
            ho[256 + j] = ghn;        // the four lanes of the quad write the same value
            hrow = ho;
            hprev = h;
            gi_r = gn_r; gi_z = gn_z; gi_n = gn_n;
            lds_barrier();            // the step's rendezvous (the last one of a block is the block boundary too)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward through time.  Per step inputs: d_out, r, z, n, gh_n, h_prev (6 rows of 64); outputs: dgi (192),
// dgh (192), h_prev (64) = 7 rows of 64.
// HEADS (round 5, hfuse.h): the launch is the TOP layer's; the output heads' forward, the mean-teacher loss and the heads'
// backward of the clip run as a prologue phase of every workgroup, and the upstream gradient d_out [T][64] of this direction
// is read from LDS (dout_s) instead of global memory.  d_out / d_out2 are unused then.
template <int NIN, bool HEADS>
__global__ __launch_bounds__(G4B_THREADS) void k_gru4_bwd(const float* __restrict__ d_out, const float* __restrict__ d_out2,
                                                          const float* __restrict__ out, const float* __restrict__ gates,
                                                          const float* __restrict__ w_hh_f, const float* __restrict__ w_hh_r,
                                                          const float* __restrict__ w_ih_f, const float* __restrict__ w_ih_r,
                                                          float* __restrict__ dgi, float* __restrict__ dgh,
                                                          float* __restrict__ hprev_out, float* __restrict__ dx_planes, int B, int T,
                                                          HeadsFuse hf) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* zero = gsm;                                // [192] dg before the first step
    float* ops = zero + 192;                          // [2][G4_SB][384] : d_out, r, z, n, gh_n, h_prev
    float* hist = ops + 2 * G4_SB * 384;              // [2][G4_SB][G4_HS] : dgi(192), dgh(192), h_prev(64)
    float* dout_s = hist + 2 * G4_SB * G4_HS;         // HEADS: [round16(T)][64] this direction's half of dL/dh
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int role = tid >> 6;                        // 0-3 recurrence, 4-5 I/O, 6-9 dX GEMM
    const int l = tid & 63;
    const int nblk = (T + G4_SB - 1) / G4_SB;
    auto t_of = [&](int step) { return dir ? step : (T - 1 - step); };     // reverse of the forward order
    TSW(0, 0);
    // operand row a (0: d_out, 1-4: r z n gh_n, 5: h_prev) of recurrence step `step`, lane l; clamped address +
    // validity, so that the loads issue unconditionally back to back
    auto op_load = [&](int step, int a, bool& ok) -> float {
            const int t = t_of(min(step, T - 1));
            const int tp = dir ? t + 1 : t - 1;
            const int tpc = min(max(tp, 0), T - 1);
            ok = (step < T) && (a != 5 || (tp >= 0 && tp < T));
            if constexpr (HEADS) {
                if (a == 0) return dout_s[t * 64 + l];
            }
            const float* p = (a == 0) ? d_out + (size_t)(b * T + t) * 128 + dir * 64 + l
                           : (a < 5) ? gates + ((size_t)(b * T + t) * 2 + dir) * 256 + (a - 1) * 64 + l
                                     : out + (size_t)(b * T + tpc) * 128 + dir * 64 + l;
            return *p;
    };
    // HEADS: what the recurrence needs before its first step - the W_hh columns of the four recurrence waves (48 registers)
    // and the saved gates / previous states of block 0 of the two I/O waves (40 of 48 slots; the d_out slots come out of the
    // heads phase) - is REQUESTED here and lands while the heads phase runs (3.4 us of strided weight loads otherwise sat
    // between the phase and step 0).  The dX GEMM waves need their W_ih fragments only from block 1 on and load them later.
    float pre[48];
    unsigned long long pre_ok = 0;
    if constexpr (HEADS) {
        if (role < 4) {
            G4Rot<0>::load_bwd(pre, dir ? w_hh_r : w_hh_f, l >> 4, l & 15, 16 * role + (l & 15));
        } else if (role < 6) {
#pragma unroll
            for (int i = 0; i < 6 * G4_SB / 2; ++i) {
                bool ok = false;
                pre[i] = (i % 6 == 0) ? 0.f : op_load(2 * (i / 6) + (role - 4), i % 6, ok);
                pre_ok |= (ok ? 1ull : 0ull) << i;
            }
        }
        // (scratch = the ops / history rings, which nothing has touched yet; ends with a workgroup barrier)
        heads_fused_phase<G4B_THREADS>(hf, out, b, dir, 2 * (int)gridDim.x, (int)(blockIdx.y * gridDim.x + blockIdx.x), T, ops, dout_s);
        TSW(8, 0);
    }

    if (role == 4 || role == 5) {
        // ================================ I/O waves =========================================================
        // two of them (96 operand values per lane and block do not fit one wave's registers next to the output slices):
        // wave io takes the steps of its parity
        const int io = role - 4;
        // the wave's 8 steps of block `blk`: global -> registers / registers -> ops[blk & 1].  The upstream gradient is the
        // sum of two direction planes below the top layer: the second plane travels in registers of its own and is added
        // on the way into LDS (an add at load time would make the wave wait for the loads before its first barrier)
        auto ops_load = [&](int blk, float (&v)[6 * G4_SB / 2], float (&v2)[G4_SB / 2], unsigned long long& okm) {
            okm = 0;
#pragma unroll
            for (int i = 0; i < 6 * G4_SB / 2; ++i) {
                bool ok;
                v[i] = op_load(blk * G4_SB + 2 * (i / 6) + io, i % 6, ok);
                okm |= (ok ? 1ull : 0ull) << i;
            }
#pragma unroll
            for (int i = 0; i < G4_SB / 2; ++i) {
                const int t = t_of(min(blk * G4_SB + 2 * i + io, T - 1));
                v2[i] = (!HEADS && d_out2) ? d_out2[(size_t)(b * T + t) * 128 + dir * 64 + l] : 0.f;
            }
        };
        auto ops_store = [&](int blk, const float (&v)[6 * G4_SB / 2], const float (&v2)[G4_SB / 2], unsigned long long okm) {
            float* on = ops + (blk & 1) * G4_SB * 384;
#pragma unroll
            for (int i = 0; i < 6 * G4_SB / 2; ++i) {
                const float x = (i % 6 == 0) ? v[i] + v2[i / 6] : v[i];
                on[(2 * (i / 6) + io) * 384 + 64 * (i % 6) + l] = ((okm >> i) & 1ull) ? x : 0.f;
            }
        };
        auto put_row = [&](int blk, int s) {
            const float* hp = hist + (blk & 1) * G4_SB * G4_HS + s * G4_HS;
            const size_t bt = (size_t)(b * T + t_of(blk * G4_SB + s)) * 2 + dir;
            float v[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) v[a] = hp[64 * a + l];
#pragma unroll
            for (int a = 0; a < 3; ++a) dgi[bt * 192 + 64 * a + l] = v[a];
#pragma unroll
            for (int a = 3; a < 6; ++a) dgh[bt * 192 + 64 * (a - 3) + l] = v[a];
            hprev_out[bt * 64 + l] = v[6];
        };
        if constexpr (HEADS) {
            float first2[G4_SB / 2];
#pragma unroll
            for (int i = 0; i < G4_SB / 2; ++i) {          // the d_out slots: out of LDS now
                bool ok = false;
                pre[6 * i] = op_load(2 * i + io, 0, ok);
                pre_ok |= (ok ? 1ull : 0ull) << (6 * i);
                first2[i] = 0.f;
            }
            ops_store(0, pre, first2, pre_ok);
        } else {
            float first[6 * G4_SB / 2], first2[G4_SB / 2];
            unsigned long long okm;
            ops_load(0, first, first2, okm);
            ops_store(0, first, first2, okm);
        }
        __syncthreads();
        for (int blk = 0; blk < nblk; ++blk) {
            const int s0 = blk * G4_SB, sb = min(G4_SB, T - s0);
            float nxt[6 * G4_SB / 2], nxt2[G4_SB / 2];
            unsigned long long okm = 0;
            const bool more = s0 + G4_SB < T;
            if (more) ops_load(blk + 1, nxt, nxt2, okm);
            for (int s = 0; s < sb; ++s) {
#ifndef G4_EXP_NOIO
                if (blk > 0 && (s & 1) == io) put_row(blk - 1, s);
#endif
                if (s == sb - 1 && more) ops_store(blk + 1, nxt, nxt2, okm);
                lds_barrier();
            }
        }
        {
            const int blk = nblk - 1, sb = T - blk * G4_SB;
            if (blk > 0)
                for (int s = sb + ((sb ^ io) & 1); s < G4_SB; s += 2) put_row(blk - 1, s);
            for (int s = io; s < sb; s += 2) put_row(blk, s);
        }
        TSW(4, 256);
        return;
    }
    if (role >= 6) {
        // ================================ dX GEMM waves ========================================================
        // block of 16 steps x 192 gates times W_ih[dir] (192 x NIN); each wave owns NIN/64 column tiles and keeps their
        // B fragments (k = gate, j = input feature) resident; the 48 k-steps of block blk-1 are spread over the time steps
        // of block blk, three per step
        constexpr int CT = NIN / 64;                      // column tiles per wave
        const int gw = role - 6, i16 = l & 15, kq = l >> 4;
        const float* wih = dir ? w_ih_r : w_ih_f;
        float bw[CT][48];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int s4 = 0; s4 < 48; ++s4) bw[c][s4] = wih[(size_t)(4 * s4 + kq) * NIN + 16 * (gw * CT + c) + i16];
        float* plane = dx_planes + (size_t)dir * B * T * NIN;
        // (the prologue rendezvous: LDS-only barrier - these waves have written nothing, and a __syncthreads() here made the
        // whole workgroup wait for their 96 fragment loads, which are first used a block of 16 steps later)
        lds_barrier();
        // n_sync: how many of the 16 slices end with a barrier (the time steps of the block running meanwhile)
        auto gemm_block = [&](int blk, int n_sync, auto sync_tag) {
            constexpr bool SYNC = decltype(sync_tag)::value;      // false: the tail call - no barrier code in the loop, so the 48 LDS
                                                                  // reads are hoisted ahead of the MFMAs instead of one exposed read each
            const int s0 = blk * G4_SB, sb = min(G4_SB, T - s0);
            const float* hp = hist + (blk & 1) * G4_SB * G4_HS + i16 * G4_HS + kq;
            const bool rowok = i16 < sb;
            v4f acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 48; ++s4) {             // fully unrolled: bw[][] must stay in registers
                const float a = rowok ? hp[4 * s4] : 0.f;
#ifndef G4_EXP_NODX
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[c][s4], acc[c], 0, 0, 0);
#endif
                if (SYNC && (s4 + 1) % 3 == 0 && (s4 + 1) / 3 <= n_sync) lds_barrier();
            }
            // D: lane (j = i16, rows 4 kq + r) -> time step s0 + 4 kq + r of the block
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * kq + r;
                if (s < sb) {
                    float* dst = plane + (size_t)(b * T + t_of(s0 + s)) * NIN + 16 * gw * CT + i16;
#pragma unroll
                    for (int c = 0; c < CT; ++c) dst[16 * c] = acc[c][r];
                }
            }
        };
        for (int blk = 0; blk < nblk; ++blk) {
            const int sb = min(G4_SB, T - blk * G4_SB);
            if (blk > 0) gemm_block(blk - 1, sb, std::true_type{});
            else
                for (int s = 0; s < sb; ++s) lds_barrier();
        }
        TSW(5, 384);
        gemm_block(nblk - 1, 0, std::false_type{});
        TSW(6, 384);
        return;
    }
    // ==================================== recurrence waves ===================================================
    // lane (kq = lane >> 4, u = lane & 15): column j = 16 w + u of W_hh, rows 48 kq .. 48 kq + 47.  The 48 dg values of the
    // quarter live in the ROW's registers (3 per lane, 3 ds_read_b32 - not 12 replicated ds_read_b128, which made the step
    // LDS-bandwidth bound: 48 KB per step and workgroup) and reach the lanes through 16 DPP row rotations.
    const int w = role, kq = l >> 4, u = l & 15, j = 16 * w + u;
    const float* whh = dir ? w_hh_r : w_hh_f;
    float (&wt)[48] = pre;
    if constexpr (!HEADS) G4Rot<0>::load_bwd(wt, whh, kq, u, j);
    for (int e = tid; e < 192; e += 256) zero[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) asm volatile("" : "+v"(wt[i]));      // pin the load waits before the loop
    TSW(7, 0);
    __syncthreads();
    TSW(1, 0);
    __builtin_amdgcn_s_setprio(3);
    // the unit's seven values of a step go out in two ds_writes: row kq writes (dr, dr) / (dz, dz) / (dn, dgh_n) / (hp, hp)
    const int wofs1 = (kq < 3) ? 64 * kq + j : 384 + j;
    const int wofs2 = (kq < 3) ? 192 + 64 * kq + j : 384 + j;
    const int rofs = 48 * kq + u;
    const float* dgrow = zero;
    float dhz = 0.f;                  // dh * z of the previous step
    const bool is1 = kq == 1, is2 = kq == 2, is3 = kq == 3;
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, sb = min(G4_SB, T - blk * G4_SB);
        const float* ob = ops + cur * G4_SB * 384;
        float* hb = hist + cur * G4_SB * G4_HS;
        float dout = ob[j], r = ob[64 + j], z = ob[128 + j], nn = ob[192 + j], ghn = ob[256 + j], hp = ob[320 + j];
        for (int s = 0; s < sb; ++s) {
            float dgv[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) dgv[m] = dgrow[rofs + 16 * m];
            // the NEXT step's operands now (their LDS latency must not sit between the mat-vec and the gate math)
            const float* o = ob + min(s + 1, G4_SB - 1) * 384;
            const float n_dout = o[j], n_r = o[64 + j], n_z = o[128 + j], n_nn = o[192 + j], n_ghn = o[256 + j], n_hp = o[320 + j];
            __builtin_amdgcn_sched_barrier(0);
            float acc[3] = {0.f, 0.f, 0.f};
#ifdef G4_EXP_NOMV
            acc[0] = dgv[0] * wt[0]; acc[1] = dgv[1] * wt[1]; acc[2] = dgv[2] * wt[2];
#else
            G4Rot<0>::dot3(acc, dgv, wt);
#endif
#ifdef G4_EXP_NOSUM
            const float dh = dout + (dhz + ((acc[0] + acc[1]) + acc[2]));
#else
            const float dh = dout + (dhz + g4_row4_sum((acc[0] + acc[1]) + acc[2]));
#endif
            const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
            const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dghn = dn_pre * r;
            float* ho = hb + s * G4_HS;
            float v1 = dr_pre, v2 = dr_pre;
            v1 = is1 ? dz_pre : v1; v2 = is1 ? dz_pre : v2;
            v1 = is2 ? dn_pre : v1; v2 = is2 ? dghn : v2;
            v1 = is3 ? hp : v1;     v2 = is3 ? hp : v2;
            ho[wofs1] = v1;
            ho[wofs2] = v2;
            dgrow = ho + 192;
            dhz = dh * z;
            dout = n_dout; r = n_r; z = n_z; nn = n_nn; ghn = n_ghn; hp = n_hp;
            lds_barrier();
        }
        if (blk == 0) TSW(2, 0);
    }
    TSW(3, 0);
}

template <int NIN> static constexpr size_t gru4_fwd_lds() { return (size_t)(64 + 2 * G4_SB * 192 + 2 * G4_SB * 320 + 2 * G4_SB * (NIN + 4)) * sizeof(float); }
static const size_t GRU4_BWD_LDS = (size_t)(192 + 2 * G4_SB * 384 + 2 * G4_SB * G4_HS) * sizeof(float);
static_assert(2 * G4_SB * 384 + 2 * G4_SB * G4_HS >= HF_TMAX * HF_S + HF_MAXO * HF_S + HF_TMAX * HF_SD + HF_TMAX * 8 + HF_MISC,
              "the heads phase's scratch must fit the ops / history rings it aliases");
static const size_t GRU4_BWD_HEADS_LDS_MAX = GRU4_BWD_LDS + (size_t)HF_TMAX * 64 * sizeof(float);

int launch_gru_fwd_v1(const float* x, int nin, const float* w_ih_f, const float* w_ih_r, const float* b_ih_f, const float* b_ih_r,
                      const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r, float* out, float* gates,
                      int B, int T, hipStream_t st);
int launch_gru_bwd_v1(const float* d_out, const float* d_out2, const float* out, const float* gates, const float* w_hh_f,
                      const float* w_hh_r, const float* w_ih_f, const float* w_ih_r, int nin, float* dgi, float* dgh, float* hprev,
                      float* dx_planes, int B, int T, hipStream_t st);

// x: the layer input [B*T][nin] (the input projection runs inside the kernel)
int launch_gru_fwd(const float* x, int nin, const float* w_ih_f, const float* w_ih_r, const float* b_ih_f, const float* b_ih_r,
                   const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r, float* out, float* gates,
                   int B, int T, hipStream_t st) {
#ifdef SED_AB
    if (g_sed_debug & 4096)       // round 1's one-wave recurrence, for A/B timing
        return launch_gru_fwd_v1(x, nin, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, B, T, st);
#endif
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru4_fwd<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gru4_fwd_lds<64>()));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru4_fwd<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gru4_fwd_lds<128>()));
    }
    if (nin == 128)
        k_gru4_fwd<128><<<dim3(B, 2), G4_THREADS, gru4_fwd_lds<128>(), st>>>(x, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, T);
    else if (nin == 64)
        k_gru4_fwd<64><<<dim3(B, 2), G4_THREADS, gru4_fwd_lds<64>(), st>>>(x, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, w_hh_r, b_hh_f, b_hh_r, out, gates, T);
    else {
        sed_set_error("gru forward: unsupported input width %d", nin);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// dx_planes: [2][B*T][nin] - the two directions' shares of the gradient w.r.t. the layer input (the consumer adds
// them); d_out2: optional second plane of the upstream gradient (the layer above's dx_planes + B*T*128), or null
int launch_gru_bwd(const float* d_out, const float* d_out2, const float* out, const float* gates, const float* w_hh_f,
                   const float* w_hh_r, const float* w_ih_f, const float* w_ih_r, int nin, float* dgi, float* dgh, float* hprev,
                   float* dx_planes, int B, int T, hipStream_t st) {
#ifdef SED_AB
    if (g_sed_debug & 4096)
        return launch_gru_bwd_v1(d_out, d_out2, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, nin, dgi, dgh, hprev, dx_planes, B, T, st);
#endif
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru4_bwd<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU4_BWD_LDS));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru4_bwd<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU4_BWD_LDS));
    }
    if (nin == 128)
        k_gru4_bwd<128, false><<<dim3(B, 2), G4B_THREADS, GRU4_BWD_LDS, st>>>(d_out, d_out2, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, dgi, dgh,
                                                                             hprev, dx_planes, B, T, HeadsFuse{});
    else if (nin == 64)
        k_gru4_bwd<64, false><<<dim3(B, 2), G4B_THREADS, GRU4_BWD_LDS, st>>>(d_out, d_out2, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, dgi, dgh,
                                                                            hprev, dx_planes, B, T, HeadsFuse{});
    else {
        sed_set_error("gru backward: unsupported input width %d", nin);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// The top layer's backward recurrence with the heads phase in front (hfuse.h); T <= HF_TMAX frames
int launch_gru_bwd_heads(const float* out, const float* gates, const float* w_hh_f, const float* w_hh_r, const float* w_ih_f,
                         const float* w_ih_r, int nin, float* dgi, float* dgh, float* hprev, float* dx_planes, int B, int T,
                         const HeadsFuse& hf, hipStream_t st) {
    if (T > HF_TMAX || hf.NC < 1 || hf.NC > 16 || hf.hl.strong_ema == nullptr) {
        sed_set_error("fused heads + gru backward: needs T/8 <= %d, nclass <= 16 and the loss inputs (got T/8 = %d)", HF_TMAX, T);
        return SED_ERR_UNSUPPORTED;
    }
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need()) {
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru4_bwd<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU4_BWD_HEADS_LDS_MAX));
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_gru4_bwd<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU4_BWD_HEADS_LDS_MAX));
    }
    const size_t lds = GRU4_BWD_LDS + (size_t)((T + 15) & ~15) * 64 * sizeof(float);
    if (nin == 128)
        k_gru4_bwd<128, true><<<dim3(B, 2), G4B_THREADS, lds, st>>>(nullptr, nullptr, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, dgi, dgh,
                                                                    hprev, dx_planes, B, T, hf);
    else if (nin == 64)
        k_gru4_bwd<64, true><<<dim3(B, 2), G4B_THREADS, lds, st>>>(nullptr, nullptr, out, gates, w_hh_f, w_hh_r, w_ih_f, w_ih_r, dgi, dgh,
                                                                   hprev, dx_planes, B, T, hf);
    else {
        sed_set_error("gru backward: unsupported input width %d", nin);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}

