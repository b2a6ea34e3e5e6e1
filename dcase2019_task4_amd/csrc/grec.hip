// grec.hip - H = 256 GRU recurrence for SED_DTYPE_BF16: ONE workgroup per (clip, direction) chain with the whole W_hh in
// registers as bf16, no cross-workgroup exchange.
//
// Reference op: nn.GRU(bidirectional=True) inside BidirectionalGRU (baseline/models/RNN.py:12-16), gate order r, z, n:
//   r = s(gi_r + W_hr h + b_hr)   z = s(gi_z + W_hz h + b_hz)   n = tanh(gi_n + r (W_hn h + b_hn))   h' = (1 - z) n + z h
//
// Why: 3H x H fp32 = 768 KB does not fit one CU, so the fp32 kernels (ggru.hip k_gclu_*) split a chain over FOUR workgroups
// that hand h over through L2 every time step - 1.6 - 2.4 us per step, ~1 us of it the publish -> poll round trip, on 192
// CUs per launch.  In bf16 the matrix is 384 KB and fits the register file of one CU: 512 threads, thread
// (unit pair up = t >> 2, k quarter kq = t & 3) keeps W_hh[gate][2 up + {0, 1}][64 kq .. +64) as 192 packed registers (two
// units per thread halve the LDS reads of h per dot product; all registers are ARCH VGPRs:
// a VALU instruction cannot address the AGPR half of the file, and parking weights there costs a v_accvgpr_read per use -
// tools/ubench/dot2_matvec.cpp: 1.78 us per step with 256 threads x 384 registers, 1.0 us with 512 x 192).  Per step: the
// 256-vector h (bf16, LDS, double-buffered: one barrier per step) is read with broadcast ds_read_b128, each thread forms its
// three half dot products with v_dot2c_f32_bf16 (fp32 accumulation; 2.0 ns per instruction per wave = the fp32 FMA rate
// for two MACs), the four quarters meet through two DPP quad_perm adds, lane kq of a quad does the gate math of unit kq & 1.
// Arithmetic: W_hh and the h that enters the mat-vec are rounded to bf16 (RNE) - the GEMM operands, as sed_dims.dtype =
// SED_DTYPE_BF16 states for every GEMM-shaped operator; gi, the biases, the gates and the carried state h are fp32.
// The backward kernel is the transpose: thread (column pair, gate-row quarter) keeps 2 x 192 entries and reads a quarter of
// the step's 768 gate gradients (bf16, LDS) - in its first form (one column x half the rows per thread) the broadcast reads
// alone were 384 KB per step and CU = 0.64 us of LDS time.
#include "gen.h"
#include "kernels.h"
#include "gkernels.h"
#include "gpack.h"

SED_TS_DEFINE(grec)
// tuning aid (make EXTRA=-DSED_TS; tools/ts_generic.py grec): shader-clock stamps of ONE time step (s = 20) of workgroup 0's
// wave 0 - forward stamps 0 .. 6, backward 8 .. 14.  sched_barriers pin the phases; compiles to nothing in the product build.
#ifdef SED_TS
#define GREC_TS(k) do { if (s == 20) { __builtin_amdgcn_sched_barrier(0); TSC(k); __builtin_amdgcn_sched_barrier(0); } } while (0)
// the values a phase produces are pinned in front of its stamp (every step, so that the stamped step runs the same code as the others)
#define GREC_PIN6(a) asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]) :: "memory")
#define GREC_PIN4(a) asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) :: "memory")
#define GREC_PIN1(x) asm volatile("" : "+v"(x) :: "memory")
#else
#define GREC_TS(k) do { } while (0)
#define GREC_PIN6(a) do { } while (0)
#define GREC_PIN4(a) do { } while (0)
#define GREC_PIN1(x) do { } while (0)
#endif
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4r;
#define GREC_H 256
#define GREC_T 512

__device__ __forceinline__ float dot2(unsigned int a, unsigned int b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
// the same with fp16 pairs (SED_DTYPE_F16's forward recurrence): v_dot2_f32_f16, fp32 accumulation
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2r;
__device__ __forceinline__ float dot2h(unsigned int a, unsigned int b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2r, a), __builtin_bit_cast(f16x2r, b), c, false);
}
// sum over the four lanes of a quad: DPP quad_perm [1, 0, 3, 2] then [2, 3, 0, 1]; every lane ends with the same value
__device__ __forceinline__ float quad_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
    return x;
}
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * rcp_fast(1.0f + __expf(-2.0f * x)) - 1.0f; }

// ---- packing (once per forward): fp32 W_hh [3H][H] of both directions -> bf16 in the order the threads load it ----------
//   wp [dir][g][kc][u][8]  = W[g H + u][8 kc + e]          (forward:  16-byte vector (g, kc) of unit u; coalesced over u)
//   wpT[dir][gc][j][8]     = W[8 gc + e][j]                (backward: 16-byte vector gc of column j; gc < 3H / 8)
__global__ __launch_bounds__(256) void k_grec_pack(const float* __restrict__ w_f, const float* __restrict__ w_r,
                                                    __bf16* __restrict__ wp, __bf16* __restrict__ wpT, int f16) {
    grec_pack_body(w_f, w_r, wp, wpT, blockIdx.x * 256 + threadIdx.x, f16);      // (gpack.h: training forwards run it in the moments launch)
}
int launch_grec_pack(const float* w_hh_f, const float* w_hh_r, void* wp, void* wpT, hipStream_t st, int f16) {
    const int n = 2 * 3 * GREC_H * GREC_H / 8;
    k_grec_pack<<<(n + 255) / 256, 256, 0, st>>>(w_hh_f, w_hh_r, (__bf16*)wp, (__bf16*)wpT, f16);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// Global I/O is BLOCKED over time: vmcnt counts loads and stores alike and in order, so a wave that stores its step's
// outputs and then waits for the next step's inputs drains its own stores - an HBM / L2 write latency per time step (the
// first version: 1.2 us per step forward, 2.0 backward, against 1.0 for the arithmetic alone).  A ninth wave for the I/O
// (ggru.hip's way) does not fit: nine waves put three on one SIMD, and 3 x 232 registers exceed its file.  Instead every
// GREC_TBF / GREC_TBB steps ALL waves (a) flush the previous block's outputs from an LDS ring with 16-byte coalesced
// stores and (b) start the LDS-DMA of the block after next's inputs into the other half of a double-buffered input ring
// (no registers: the kernel sits at 256) - the one vmcnt(0) per block then falls on transfers issued microseconds
// earlier - and the steps in between touch LDS only.
#define GREC_TBF 8
#define GREC_TBB 8                       // (4 until the end of round 3: a block boundary costs ~2 600 cycles - 127 -> 123 us)
// one wave-instruction: 64 lanes x 16 bytes from global straight into 1 KB of LDS (global_load_lds_dwordx4: no VGPRs; the
// destination is the wave-uniform `lds` + 16 * lane, the source address is per lane).  Completion: s_waitcnt vmcnt + barrier.
// The source is given as a wave-uniform base + 16 * lane so that the instruction takes its scalar-base form (saddr + 32-bit
// voffset): per-lane 64-bit pointers for three source tensors were six loop-invariant VGPRs the backward kernel does not have.
__device__ __forceinline__ void dma16(const float* gsrc_wave, int lane, float* lds_wave) {
    const char* p = (const char*)gsrc_wave + (unsigned)(16 * lane);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)lds_wave, 16, 0, 0);
}

// gi: [B*T][2][3H] (input projection incl. b_ih); out [B*T][2H]; gates [B*T][2][4H] (r, z, n, gh_n) or null
// F16: wp holds fp16 (grec_pack_body f16), the h that enters the mat-vec is rounded to fp16, v_dot2_f32_f16
template <int F16>
__global__ __launch_bounds__(GREC_T) void k_grec_fwd(const float* __restrict__ gi, const __bf16* __restrict__ wp,
                                                      const float* __restrict__ b_hh_f, const float* __restrict__ b_hh_r,
                                                      float* __restrict__ out, float* __restrict__ gates, int B, int T) {
    constexpr int H = GREC_H, TB = GREC_TBF;
    __shared__ __attribute__((aligned(16))) unsigned int hs[2][H / 2];
    __shared__ __attribute__((aligned(16))) float gis[2][TB][3][H];
    __shared__ __attribute__((aligned(16))) float outs[TB][5][H];            // r, z, n, gh_n, h
    const int chain = blockIdx.x, b = chain >> 1, dir = chain & 1;
    const int t = threadIdx.x, up = t >> 2, kq = t & 3, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: the block I/O addressing below stays scalar
    const int u = 2 * up + (kq & 1);                              // the unit whose gates this lane forms (kq >= 2: duplicates)
    const bool writer = kq < 2;
    const float* bhh = dir ? b_hh_r : b_hh_f;
    unsigned int wr[2][3][32];
    {
        const u32x4r* src = (const u32x4r*)(wp + (size_t)dir * 3 * H * H);
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32x4r v = src[(size_t)(g * (H / 8) + 8 * kq + c) * H + 2 * up + uu];
                    wr[uu][g][4 * c] = v.x; wr[uu][g][4 * c + 1] = v.y; wr[uu][g][4 * c + 2] = v.z; wr[uu][g][4 * c + 3] = v.w;
                }
    }
    const float bh_r = bhh[u], bh_z = bhh[H + u], bh_n = bhh[2 * H + u];
    if (t < H / 2) { hs[0][t] = 0u; hs[1][t] = 0u; }
    auto t_of = [&](int s) { return dir ? (T - 1 - s) : s; };
    // block I/O in 16-byte items: inputs TB x 192 (3 wave-instructions per wave; 192 = 3 x 64: a wave-instruction never
    // straddles two time steps), outputs TB x 320 (5 per thread)
    auto in_dma = [&](int s0, int buf) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i0 = 64 * (wv + 8 * k), st = i0 / 192, rem0 = i0 % 192, s = s0 + st;
            if (s < T) dma16(gi + ((size_t)(b * T + t_of(s)) * 2 + dir) * 3 * H + 4 * rem0, lane, &gis[buf][0][0][0] + 4 * i0);
        }
    };
    auto out_flush = [&](int s0) {
#pragma unroll 1
        for (int k = 0; k < 5; ++k) {
            const int i = t + GREC_T * k, st = i / 320, rem = i % 320, s = s0 + st;
            if (s < T) {
                const size_t bt = (size_t)(b * T + t_of(s));
                if (rem < 256) {
                    if (gates) *(f32x4*)(gates + (bt * 2 + dir) * 4 * H + 4 * rem) = *(const f32x4*)(&outs[st][0][0] + 4 * rem);
                } else {
                    *(f32x4*)(out + bt * 2 * H + dir * H + 4 * (rem - 256)) = *(const f32x4*)(&outs[st][4][0] + 4 * (rem - 256));
                }
            }
        }
    };
    float hprev = 0.f;
    in_dma(0, 0);
    int buf = 0;
    for (int s0 = 0; s0 < T; s0 += TB, buf ^= 1) {
        // this block's inputs were requested a whole block ago (the prologue for block 0) and the stores still counted were
        // issued then too: the wait is over transfers that have long landed.  It comes BEFORE this iteration issues anything
        // (a counted wait would have to know how many of the flush's stores a wave really issues: none of the gate stores in
        // eval mode).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (s0 > 0) out_flush(s0 - TB);                                // (the previous block's last barrier published outs)
        in_dma(s0 + TB, buf ^ 1);
        lds_barrier();                                                 // flush reads done; every wave's share of gis[buf] landed
        const int ns = (T - s0 < TB) ? (T - s0) : TB;
        for (int st = 0; st < ns; ++st) {
            const int s = s0 + st;
            GREC_TS(0);                                                // behind the previous step's barrier
            const float gi_r = gis[buf][st][0][u], gi_z = gis[buf][st][1][u], gi_n = gis[buf][st][2][u];
            float a[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            const unsigned int* hb = hs[s & 1] + 32 * kq;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const u32x4r h4 = *(const u32x4r*)(hb + 4 * c);
                const unsigned int hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                        for (int g = 0; g < 3; ++g) a[uu][g] = F16 ? dot2h(wr[uu][g][4 * c + q], hv[q], a[uu][g]) : dot2(wr[uu][g][4 * c + q], hv[q], a[uu][g]);
            }
            GREC_PIN6(a);
            GREC_TS(1);                                                // h reads + 192 dot2
#pragma unroll
            for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                for (int g = 0; g < 3; ++g) a[uu][g] = quad_sum(a[uu][g]);
            GREC_PIN6(a);
            GREC_TS(2);                                                // quad sums (12 DPP adds)
            const bool odd = (kq & 1) != 0;
            const float gh_r = (odd ? a[1][0] : a[0][0]) + bh_r, gh_z = (odd ? a[1][1] : a[0][1]) + bh_z,
                        ghn = (odd ? a[1][2] : a[0][2]) + bh_n;
            const float r = sigmoidf_fast(gi_r + gh_r);
            const float z = sigmoidf_fast(gi_z + gh_z);
            const float nn = tanh_fast(gi_n + r * ghn);
            const float h = (1.0f - z) * nn + z * hprev;
            hprev = h;
            GREC_PIN1(hprev);
            GREC_TS(3);                                                // gate chain (2 sigmoid + tanh)
            if (writer) {
                if constexpr (F16 != 0) ((_Float16*)hs[(s + 1) & 1])[u] = (_Float16)h;
                else ((__bf16*)hs[(s + 1) & 1])[u] = (__bf16)h;
                outs[st][0][u] = r; outs[st][1][u] = z; outs[st][2][u] = nn; outs[st][3][u] = ghn; outs[st][4][u] = h;
            }
            GREC_TS(4);                                                // publish: 6 LDS writes issued
            lds_barrier();
            GREC_TS(5);                                                // barrier (incl. the writes' completion)
        }
    }
    out_flush(((T - 1) / TB) * TB);
}

// Backward through time.  d_out [B*T][2H]; out / gates as written by the forward; dgi / dgh [B*T][2][3H]; hprev [B*T][2][H].
//   dh = d_out + carry;  dn = dh (1 - z)(1 - n^2);  dz = dh (h_prev - n) z (1 - z);  dr = dn gh_n r (1 - r);  dghn = dn r
//   carry'[j] = dh[j] z[j] + sum_g d[g] W_hh[g][j],  d = (dr | dz | dghn)
__global__ __launch_bounds__(GREC_T) void k_grec_bwd(const float* __restrict__ d_out, const float* __restrict__ out,
                                                      const float* __restrict__ gates, const __bf16* __restrict__ wpT,
                                                      float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ hprev_out,
                                                      int B, int T) {
    constexpr int H = GREC_H, TB = GREC_TBB;
    __shared__ __attribute__((aligned(16))) unsigned int ds[2][3 * H / 2];
    __shared__ __attribute__((aligned(16))) float ins[2][TB][6][H];          // d_out, r, z, n, gh_n, h_prev
    __shared__ __attribute__((aligned(16))) float outs[TB][5][H];            // dr, dz, dn, dgh_n, h_prev
    const int chain = blockIdx.x, b = chain >> 1, dir = chain & 1;
    const int t = threadIdx.x, up = t >> 2, gq = t & 3, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: the block I/O addressing below stays scalar
    const int j = 2 * up + (gq & 1);                               // the column whose gate gradients this lane forms
    const bool writer = gq < 2;
    unsigned int wc[2][96];                                        // columns 2 up + {0, 1}, gate rows [192 gq, 192 gq + 192)
    {
        const u32x4r* src = (const u32x4r*)(wpT + (size_t)dir * 3 * H * H);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int c = 0; c < 24; ++c) {
                const u32x4r v = src[(size_t)(24 * gq + c) * H + 2 * up + jj];
                wc[jj][4 * c] = v.x; wc[jj][4 * c + 1] = v.y; wc[jj][4 * c + 2] = v.z; wc[jj][4 * c + 3] = v.w;
            }
    }
    auto t_of = [&](int s) { return dir ? s : (T - 1 - s); };     // reverse of the forward order
    // block I/O: inputs TB x 384 16-byte items (3 wave-instructions per wave; per step 64 of d_out | 256 of the gates | 64 of
    // h_prev - every wave-instruction has ONE kind and ONE step), outputs TB x 448 (3.5 per thread)
    auto in_dma = [&](int s0, int buf) {
#pragma unroll
        for (int k = 0; k < 3 * TB / 4; ++k) {
            const int i0 = 64 * (wv + 8 * k), st = i0 / 384, rem0 = i0 % 384, s = s0 + st;
            if (s < T) {
                const int tt = t_of(s);
                const size_t bt = (size_t)(b * T + tt);
                float* dst = &ins[buf][0][0][0] + 4 * i0;
                if (rem0 < 64) {
                    dma16(d_out + bt * 2 * H + dir * H, lane, dst);
                } else if (rem0 < 320) {
                    dma16(gates + (bt * 2 + dir) * 4 * H + 4 * (rem0 - 64), lane, dst);
                } else {
                    const int tp = dir ? tt + 1 : tt - 1;
                    if (tp >= 0 && tp < T) dma16(out + (size_t)(b * T + tp) * 2 * H + dir * H, lane, dst);
                    else *(f32x4*)(dst + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};      // h before the first step
                }
            }
        }
    };
    auto out_flush = [&](int s0) {
#pragma unroll 1
        for (int k = 0; k < (TB * 448 + GREC_T - 1) / GREC_T; ++k) {
            const int i = t + GREC_T * k, st = i / 448, rem = i % 448, s = s0 + st;
            if (i < TB * 448 && s < T) {
                const size_t bt2 = (size_t)(b * T + t_of(s)) * 2 + dir;
                if (rem < 192) {                                       // dgi = (dr, dz, dn)
                    *(f32x4*)(dgi + bt2 * 3 * H + 4 * rem) = *(const f32x4*)(&outs[st][0][0] + 4 * rem);
                } else if (rem < 384) {                                // dgh = (dr, dz, dgh_n)
                    const int q = rem - 192, kind = q / 64;
                    *(f32x4*)(dgh + bt2 * 3 * H + 4 * q) = *(const f32x4*)(&outs[st][kind == 2 ? 3 : kind][0] + 4 * (q % 64));
                } else {
                    *(f32x4*)(hprev_out + bt2 * H + 4 * (rem - 384)) = *(const f32x4*)(&outs[st][4][0] + 4 * (rem - 384));
                }
            }
        }
    };
    float carry = 0.f;
    in_dma(0, 0);
    int buf = 0;
    for (int s0 = 0; s0 < T; s0 += TB, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (see k_grec_fwd)
        if (s0 > 0) out_flush(s0 - TB);
        in_dma(s0 + TB, buf ^ 1);
        lds_barrier();
        const int ns = (T - s0 < TB) ? (T - s0) : TB;
        for (int st = 0; st < ns; ++st) {
            const int s = s0 + st;
            GREC_TS(8);
            const float dh = ins[buf][st][0][j] + carry;
            const float r = ins[buf][st][1][j], z = ins[buf][st][2][j], nn = ins[buf][st][3][j], ghn = ins[buf][st][4][j],
                        hp = ins[buf][st][5][j];
            const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
            const float dz_pre = dh * (hp - nn) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dghn = dn_pre * r;
            if (writer) {
                __bf16* dd = (__bf16*)ds[s & 1];
                dd[j] = (__bf16)dr_pre; dd[H + j] = (__bf16)dz_pre; dd[2 * H + j] = (__bf16)dghn;
                outs[st][0][j] = dr_pre; outs[st][1][j] = dz_pre; outs[st][2][j] = dn_pre; outs[st][3][j] = dghn; outs[st][4][j] = hp;
            }
            GREC_TS(9);                                                // 6 input reads + gate gradients + 8 LDS writes issued
            lds_barrier();                                             // this step's 3H gate gradients are in ds[s & 1]
            GREC_TS(10);                                               // barrier
            float a[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            const unsigned int* db = ds[s & 1] + 96 * gq;
#pragma unroll
            for (int c = 0; c < 24; ++c) {
                const u32x4r d4 = *(const u32x4r*)(db + 4 * c);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    a[jj][0] = dot2(wc[jj][4 * c], d4.x, a[jj][0]);
                    a[jj][1] = dot2(wc[jj][4 * c + 1], d4.y, a[jj][1]);
                    a[jj][0] = dot2(wc[jj][4 * c + 2], d4.z, a[jj][0]);
                    a[jj][1] = dot2(wc[jj][4 * c + 3], d4.w, a[jj][1]);
                }
            }
            GREC_PIN4(a);
            GREC_TS(11);                                               // 24 ds_read_b128 of gate gradients + 192 dot2
            const float c0 = quad_sum(a[0][0] + a[0][1]), c1 = quad_sum(a[1][0] + a[1][1]);
            carry = dh * z + ((gq & 1) ? c1 : c0);
            GREC_PIN1(carry);
            GREC_TS(12);                                               // quad sums + carry
            // (ds is double-buffered: the next step writes the other buffer, and every reader of this one has passed the
            // next barrier before it is written again)
        }
    }
    out_flush(((T - 1) / TB) * TB);
}

int launch_grec_fwd(const float* gi, const void* wp, const float* b_hh_f, const float* b_hh_r, float* out, float* gates, int B, int T,
                    hipStream_t st, int f16) {
    if (f16) k_grec_fwd<1><<<2 * B, GREC_T, 0, st>>>(gi, (const __bf16*)wp, b_hh_f, b_hh_r, out, gates, B, T);
    else k_grec_fwd<0><<<2 * B, GREC_T, 0, st>>>(gi, (const __bf16*)wp, b_hh_f, b_hh_r, out, gates, B, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
int launch_grec_bwd(const float* d_out, const float* out, const float* gates, const void* wpT, float* dgi, float* dgh, float* hprev,
                    int B, int T, hipStream_t st) {
    k_grec_bwd<<<2 * B, GREC_T, 0, st>>>(d_out, out, gates, (const __bf16*)wpT, dgi, dgh, hprev, B, T);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
