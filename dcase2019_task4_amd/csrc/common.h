// common.h - shared device helpers, geometry and buffer layouts (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dcase_sed.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SED_C 64          // conv channels on the hot path
#define SED_HID 64        // GRU hidden size on the hot path
#define SED_WAVE 64

// ---- error plumbing (host) ---------------------------------------------------------------
void sed_set_error(const char* fmt, ...);
#define SED_CHECK_ARG(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            sed_set_error("%s:%d: %s", __FILE__, __LINE__, msg);   \
            return SED_ERR_BAD_ARG;                                \
        }                                                          \
    } while (0)
#define SED_CHECK_HIP(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            sed_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SED_ERR_LAUNCH;                                                          \
        }                                                                                   \
    } while (0)
#define SED_CHECK_LAUNCH() SED_CHECK_HIP(hipGetLastError())
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: every launcher raises it once per
// device it is used on AND per host thread: `static thread_local SedAttrOnce once; if (once.need()) { ...set attributes... }`.
// The flag is thread-local on purpose: a process-wide flag has to be set either before the attribute call (a second thread
// then sees it, skips the call and can launch with the default 64 KB limit while the first thread is still inside
// hipFuncSetAttribute) or after it (which needs a hook behind every call site's error returns).  The call is idempotent and
// thread-safe, so each thread simply makes it once itself; a thread only skips what it has itself completed.  If the call
// fails the launcher returns SED_ERR_LAUNCH and the launches that follow fail loudly on the LDS size.
struct SedAttrOnce {
    bool done[64] = {};
    bool need() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};
#define SED_TRY(expr)               \
    do {                            \
        int _s = (expr);            \
        if (_s != SED_OK) return _s; \
    } while (0)

// ---- geometry ------------------------------------------------------------------------------
struct Geo {
    int B, T, F, C, H, NC, L;
    int mode;     // SED_DTYPE_F32 / SED_DTYPE_BF16 / SED_DTYPE_BF16X3: the arithmetic FAMILY of the generic kernel set.
                  // SED_DTYPE_F16 runs in the BF16 family (bf16 backward, bf16 copies of the activations) with f16 set:
    bool f16;     // the forward chain's operands and hand-over activations are fp16 (SED_DTYPE_F16)
    bool generic; // anything but C = 64, H = 64, fp32: served by the generic kernel set (gen.h)
    int H1, W1;   // after block 0 pooling: T/2 x 16
    int H2, W2;   // after block 1 pooling: H1/2 x 4
    int T3;       // after block 2 pooling: H2/2 (x 1)
    float p, eps, mom;
};
static inline Geo make_geo(const sed_dims* d) {
    Geo g;
    g.B = d->B; g.T = d->T; g.F = d->F; g.C = d->C; g.H = d->H; g.NC = d->nclass; g.L = d->n_layers_rnn;
    g.f16 = d->dtype == SED_DTYPE_F16;
    g.mode = g.f16 ? SED_DTYPE_BF16 : d->dtype;
    g.generic = !(d->C == 64 && d->H == 64 && d->dtype == SED_DTYPE_F32);
    g.H1 = g.T / 2; g.W1 = g.F / 4;
    g.H2 = g.H1 / 2; g.W2 = g.W1 / 4;
    g.T3 = g.H2 / 2;
    g.p = d->p_drop; g.eps = d->bn_eps; g.mom = d->bn_momentum;
    return g;
}
int sed_validate_dims(const sed_dims* d);

// flat parameter offsets (elements), reference named_parameters() order
struct ParamOff {
    int64_t conv_w[3], conv_b[3], bn_g[3], bn_b[3], glu_w[3], glu_b[3];
    int64_t w_ih[2][2], w_hh[2][2], b_ih[2][2], b_hh[2][2];   // [layer][dir]
    int64_t dense_w, dense_b, soft_w, soft_b;
    int64_t total;
    int count;
};
ParamOff make_param_off(const Geo& g, int64_t* offsets_out /* may be null */);

// saved-context layout (byte offsets)
struct CtxLayout {
    size_t acc0;                       // mom0 | stat1 | stat2 contiguous (320 doubles, zeroed by ONE memset)
    size_t mom0, wz0, wl0, bn0;        // bn0: mean,invstd,scale,shift [4][64]
    size_t mompart;                    // per-workgroup partial patch moments [parts][54] fp64
    size_t wpkT1, wpkT2;               // flipped/transposed conv weights for dgrad (train only)
    size_t p0;
    size_t wpk1, y1, stat1, bn1, p1;
    size_t wpk2, y2, stat2, bn2, p2;
    size_t gates[2], out[2];
    size_t logits_s, strong_sv, weak_sv, den_sv;
    size_t mask0, mask1, mask2;        // 16 dropout keep bits per (row block, half, lane), written by forward
    size_t total;
};
CtxLayout make_ctx_layout(const Geo& g);

struct WsLayout {
    size_t d_out, dgi[2], dgh[2], hprev[2], d_in, heads_part;   // dgi/dgh/hprev per GRU layer (read by the side stream)
    size_t dz2, dp1, dz1, dp0, dp2;
    size_t bwd_acc;                    // gluacc1 | gluacc2 | de0 contiguous doubles, zeroed by ONE memset
    size_t bnb, gluacc1, gluacc2, coef[3], wg_part, de0, gemm_part;   // coef per conv block (1, 2)
    size_t total;
    int wgrad_blocks;
};
WsLayout make_ws_layout(const Geo& g);
#define SED_WGRAD_MAX_BLOCKS 256
#ifndef SED_GRU_SPLITK
#define SED_GRU_SPLITK 16
#endif
// fp64 accumulators of k_glu_pool_bwd: [0,4096) dWglu[co][c]; [4096,4160) dbglu; [4160,4224) sum dz; [4224,4288) sum dz*y;
// [4288] ticket of the last-workgroup epilogue (uint32 in a double slot); padded to a multiple of 8
#define SED_GLUACC_N 4296
#define SED_WINO_OFF (9 * 4096)   // conv weight panels: [9 taps][64][64] followed by the 16 x 64 x 64 Winograd-domain weights

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains every outstanding
// global load/store of the wave (s_waitcnt vmcnt(0), CDNA4 counts stores too), which puts the full
// HBM/L2 write latency on the critical path of loops that store per iteration (GRU steps) or keep
// prefetch loads in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// v_exp_f32 + v_rcp_f32 (1 ulp each): a plain `1.0f / x` compiles to the ~10-instruction IEEE division
// sequence, which made the sigmoid the dominant VALU cost of the fused GLU epilogues.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_fast(float x) { return rcp_fast(1.0f + __expf(-x)); }
// sigmoid of z given e = -log2(e) * z (the scale folded into the producer's weights): one multiply less per element
#define SED_NEG_LOG2E (-1.4426950408889634f)
__device__ __forceinline__ float sigmoid_from_scaled(float e) { return rcp_fast(1.0f + __builtin_amdgcn_exp2f(e)); }
// Persistent convolution kernels walk their tiles as tile = first, first + step, ... < end.  Workgroup w runs on XCD w % 8
// (round-robin dispatch, one L2 per XCD): with the plain walk (first = w, step = grid) image tiles that are neighbours - and
// share two halo rows - sit under eight different L2s and every halo row comes from HBM twice.  Here XCD x owns the x-th
// eighth of the tile range (consecutive tiles = consecutive image rows of a few clips) and its grid / 8 workgroups stride
// through it, so the eight XCDs stay as evenly loaded as with the plain walk.  Tile ORDER only: the same sums over the same
// tiles (the wgrad kernels' partial slabs group other tiles: rounding-level differences, run-to-run deterministic as before).
struct TileWalk { int first, end, step; };
__device__ __forceinline__ TileWalk xcd_walk(int n_tiles) {
    const int w = blockIdx.x, g = (int)gridDim.x;
#ifndef SED_NO_XCD_ORDER                  // (A/B builds: tools/build_full_variant.sh)
    if ((g & 7) == 0) {
        const int chunk = (n_tiles + 7) >> 3, lo = (w & 7) * chunk, hi = lo + chunk < n_tiles ? lo + chunk : n_tiles;
        return {lo + (w >> 3), hi, g >> 3};
    }
#endif
    return {w, n_tiles, g};
}
// debug knob (sed_debug_set): bit 0 = skip the fp64 atomics of the reduction epilogues (timing experiments only)
extern int g_sed_debug;

// mfma_f32_32x32x2f32 fragment maps (cdna_hip_programming.md section 3):
//   A[i][k]: lane l holds i = l & 31, k = l >> 5
//   B[k][j]: lane l holds k = l >> 5, j = l & 31
//   D[i][j]: lane l, reg r: j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// host launch helpers implemented per file; declared in kernels.h

// ---- tuning aid (make EXTRA=-DSED_TS): thread 0 of every workgroup stamps the 100 MHz wall clock (TS) or the
// shader clock (TSC) at phase boundaries into a per-translation-unit device array; tools/ts_kernel.py reads it
// through sed_debug_ts_<tag>.  Compiles to nothing in the product build.
#ifdef SED_TS
#define SED_TS_DEFINE(tag)                                                                                          \
    static __device__ unsigned long long g_ts[1024 * 16];                                                           \
    extern "C" int sed_debug_ts_##tag(unsigned long long* out, int n) {                                             \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts), sizeof(unsigned long long) * (n < 1024 * 16 ? n : 1024 * 16)); \
    }
#define TS(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_ts[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define TSC(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_ts[blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define SED_TS_DEFINE(tag)
#define TS(k) do { } while (0)
#define TSC(k) do { } while (0)
#endif

