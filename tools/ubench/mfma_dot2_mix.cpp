// Micro-benchmark behind csrc/grec.hip's pipe split: can a SIMD run v_mfma_f32_16x16x32_bf16 and v_dot2c_f32_bf16 side by
// side?  One 512-thread workgroup per CU (two waves per SIMD); per iteration each wave issues NM MFMAs and ND dot2, all on
// registers (no memory), in three orders: MFMA only, dot2 only, interleaved 1 : 4.  Prints shader cycles per iteration.
// Measured (MI355X, ns per iteration from HIP events): 48 MFMA per SIMD 378; 192 dot2 per SIMD 385; both interleaved 1 : 4
// 928 - no overlap, the sum plus a penalty; 192 v_fma_f32 223; MFMA + v_fma_f32 593 - again the sum.  So one useful MFMA
// column (1/16 of the tile) costs what dot2 costs, and the two do not run side by side: the H = 256 mat-vec has a floor of
// ~0.77 us per step on one CU either way, which is why grec.hip keeps all of it on dot2 (a split version - half of K on each
// pipe, 1 : 4 interleave pinned with sched_group_barrier - measured 137 / 166 us against 104 / 130).
// Build: hipcc --offload-arch=gfx950 -O3 -w mfma_dot2_mix.cpp -o mfma_dot2_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int MODE, int NCH>   // MODE 0 MFMA only, 1 dot2 only, 2 both interleaved, 3 v_fma_f32 only, 4 MFMA + v_fma_f32; NCH MFMA accumulation chains
__global__ __launch_bounds__(512) void k_mix(const unsigned int* __restrict__ w, float* __restrict__ out, long long* cyc, int iters) {
    const int t = threadIdx.x;
    u4 wq[24]; u4 wm[6];
#pragma unroll
    for (int i = 0; i < 24; ++i) wq[i] = *(const u4*)(w + (size_t)(i * 512 + t) * 4);
#pragma unroll
    for (int i = 0; i < 6; ++i) wm[i] = *(const u4*)(w + (size_t)((24 + i) * 512 + t) * 4);
    f4 acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    u4 h = wq[0];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (MODE != 1 && MODE != 3) acc[m % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wm[m % 6]), __builtin_bit_cast(bf8, h), acc[m % NCH], 0, 0, 0);
            if (MODE >= 3) {
                const u4 x = wq[m];
                a[0] = __builtin_fmaf(__builtin_bit_cast(float, x.x), __builtin_bit_cast(float, h.x), a[0]);
                a[1] = __builtin_fmaf(__builtin_bit_cast(float, x.y), __builtin_bit_cast(float, h.y), a[1]);
                a[2] = __builtin_fmaf(__builtin_bit_cast(float, x.z), __builtin_bit_cast(float, h.z), a[2]);
                a[3] = __builtin_fmaf(__builtin_bit_cast(float, x.w), __builtin_bit_cast(float, h.w), a[3]);
            } else if (MODE != 0) {
                const u4 x = wq[m];
                a[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.x), __builtin_bit_cast(bf2, h.x), a[0], false);
                a[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.y), __builtin_bit_cast(bf2, h.y), a[1], false);
                a[2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.z), __builtin_bit_cast(bf2, h.z), a[2], false);
                a[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.w), __builtin_bit_cast(bf2, h.w), a[3], false);
            }
            if (MODE == 2 || MODE == 4) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
        h.x ^= (unsigned int)it;          // (keeps the loop from collapsing)
    }
    const long long t1 = clock64();
    float sum = a[0] + a[1] + a[2] + a[3];
#pragma unroll
    for (int i = 0; i < NCH; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + t] = sum;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NCH>
static void run(const char* name, const unsigned int* w, float* out, long long* cyc, int iters) {
    k_mix<MODE, NCH><<<256, 512>>>(w, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_mix<MODE, NCH><<<256, 512>>>(w, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += c[i];
    m /= 256;
    printf("%-28s %8.1f counts/iter   %7.1f ns/iter  (per wave: 24 MFMA and/or 96 dot2 / fma; two waves per SIMD)\n", name, m / iters, ms * 1e6 / iters);
}

int main() {
    unsigned int* w; float* out; long long* cyc;
    hipMalloc(&w, 30 * 512 * 16); hipMemset(w, 0x3c, 30 * 512 * 16);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    run<0, 4>("mfma only, 4 chains", w, out, cyc, iters);
    run<0, 2>("mfma only, 2 chains", w, out, cyc, iters);
    run<0, 12>("mfma only, 12 chains", w, out, cyc, iters);
    run<1, 4>("dot2 only", w, out, cyc, iters);
    run<2, 4>("interleaved, 4 chains", w, out, cyc, iters);
    run<2, 12>("interleaved, 12 chains", w, out, cyc, iters);
    run<3, 4>("v_fma_f32 only", w, out, cyc, iters);
    run<4, 4>("mfma + v_fma_f32", w, out, cyc, iters);
    return 0;
}
