// Micro-benchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2_f32 on this box, from 1 or 2 waves
// per SIMD, with register operands only (no LDS, no memory).  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int blocks, int nacc, double flop_per_mfma, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 16 * nacc;
    const double total = mfma_per_wave * blocks * 4;
    printf("%-28s blocks %4d  %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per wave\n", name, blocks, ms,
           total * flop_per_mfma / ms * 1e-9, ms * 1e6 / mfma_per_wave);
}
int main() {
    float* d;
    hipMalloc(&d, 2048 * 256 * 4);
    run("16x16x4 8acc 1 wave/SIMD", k16<8>, 256, 8, 2.0 * 16 * 16 * 4, d);
    run("16x16x4 8acc 2 waves/SIMD", k16<8>, 512, 8, 2.0 * 16 * 16 * 4, d);
    run("16x16x4 2acc 1 wave/SIMD", k16<2>, 256, 2, 2.0 * 16 * 16 * 4, d);
    run("32x32x2 1acc (dependent chain) 1 wave/SIMD", k32<1>, 256, 1, 2.0 * 32 * 32 * 2, d);
    run("32x32x2 1acc (dependent chain) 2 waves/SIMD", k32<1>, 512, 1, 2.0 * 32 * 32 * 2, d);
    run("32x32x2 1acc (dependent chain) 3 waves/SIMD", k32<1>, 768, 1, 2.0 * 32 * 32 * 2, d);
    run("16x16x4 1acc (dependent chain) 1 wave/SIMD", k16<1>, 256, 1, 2.0 * 16 * 16 * 4, d);
    run("32x32x2 2acc 1 wave/SIMD", k32<2>, 256, 2, 2.0 * 32 * 32 * 2, d);
    run("32x32x2 2acc 2 waves/SIMD", k32<2>, 512, 2, 2.0 * 32 * 32 * 2, d);
    run("32x32x2 4acc 1 wave/SIMD", k32<4>, 256, 4, 2.0 * 32 * 32 * 2, d);
    return 0;
}
