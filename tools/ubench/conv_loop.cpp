// Micro-benchmark of the weight-stationary conv main loop in isolation (LDS halo constant, no global traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int PS = 66, HW = 18, HH = 10, RS = HW * PS, HALO = HH * RS;

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p16 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < HALO; i += 256) smem[i] = (float)(i % 7) * 0.125f;
    float bw[9][16];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s) bw[t][s] = 0.01f * (t + s + lane);
    __syncthreads();
    const float* Ab = smem + RS + (1 + p16) * PS + kq;
    f4 acc[8];
    for (int rb = 0; rb < 8; ++rb) acc[rb] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        auto load_a = [&](float (&a)[16], int gi) {
            const int t = gi / 8, sp = gi % 8;
            const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) a[q * 8 + rb] = Ab[(rb + dy) * RS + dx * PS + 4 * (2 * sp + q)];
        };
        auto mma = [&](const float (&a)[16], int gi) {
            const int t = gi / 8, sp = gi % 8;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rb = 0; rb < 8; ++rb)
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(VARIANT == 1 ? a[0] : a[q * 8 + rb], bw[t][2 * sp + q], acc[rb], 0, 0, 0);
        };
        float a0[16], a1[16];
        load_a(a0, 0);
#pragma unroll
        for (int gi = 0; gi < 72; gi += 2) {
            load_a(a1, gi + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, gi);
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < 72) load_a(a0, gi + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, gi + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int rb = 0; rb < 8; ++rb) s += acc[rb][0] + acc[rb][1] + acc[rb][2] + acc[rb][3];
    out[blockIdx.x * 256 + tid] = s;
}
template <typename K>
void run(const char* name, K kern, float* d) {
    const int iters = 40;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, HALO * 4 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 256, HALO * 4 * 2>>>(d, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 256, HALO * 4 * 2>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 1152;
    printf("%-40s %.3f ms  %.1f ns per MFMA per wave  (%.1f TFLOP/s)\n", name, ms, ms * 1e6 / mf, mf * 1024 * 2048 / ms * 1e-9);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * 4);
    run("conv loop, A from LDS", k<0>, d);
    run("conv loop, A = a[0] (reads still issued)", k<1>, d);
    return 0;
}
