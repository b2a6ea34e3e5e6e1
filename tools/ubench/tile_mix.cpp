// Micro-benchmark: one blk0-like tile = 10 v_mfma_f32_32x32x2_f32 (two chains of 5) + an epilogue of 16 x (exp2, add, rcp, fma)
// on the results, one wave per SIMD.  Variants: epilogue on the PREVIOUS tile's results (software pipelined) with the
// compiler free to interleave; MFMAs grouped then epilogue (sched_barrier between); epilogue only; MFMAs only.
// Build: hipcc --offload-arch=gfx950 -O3 tile_mix.cpp -o tile_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float epi(const f16v& l, const f16v& z) {
    float p = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) p = __builtin_fmaf(l[r], __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[r])), p);
    return p;
}
template <int MODE>   // 0 pipelined free, 1 pipelined grouped (sched barriers), 2 serial (epilogue of the same tile), 3 mfma only, 4 epilogue only
__global__ __launch_bounds__(256) void k_tile(float* out, int iters, float a0) {
    f16v al[2], az[2];
    for (int s = 0; s < 2; ++s)
        for (int r = 0; r < 16; ++r) { al[s][r] = 0.01f * r; az[s][r] = 0.02f * r; }
    float a[5], b[5], c[5];
    for (int i = 0; i < 5; ++i) { a[i] = a0 + i + threadIdx.x * 1e-3f; b[i] = a0 * 0.5f + i; c[i] = a0 * 0.25f - i; }
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cur = t & 1, prv = cur ^ 1;
            if (MODE != 4) {
                f16v l, z;
                for (int r = 0; r < 16; ++r) { l[r] = 0.f; z[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    l = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], l, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], c[s], z, 0, 0, 0);
                }
                al[cur] = l; az[cur] = z;
            }
            if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0 || MODE == 1) acc += epi(al[prv], az[prv]);
            if (MODE == 2) acc += epi(al[cur], az[cur]);
            if (MODE == 4) { acc += epi(al[prv], az[prv]); al[prv][0] = acc; }
            if (MODE == 3) acc += al[cur][0];
            if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
            a[0] += 1e-6f * acc;       // keeps the tiles dependent on the loop
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + al[0][1] + az[1][2];
}
template <typename K>
static void run(const char* name, K kern, int blocks, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 10, 1.f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<blocks, 256>>>(d, iters, 1.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-52s blocks %4d  %7.1f ns per tile per wave\n", name, blocks, best * 1e6 / iters / 2);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 1024 * 256 * 4);
    for (int blocks : {256, 768}) {
        run("pipelined, compiler interleaves", k_tile<0>, blocks, d);
        run("pipelined, MFMAs grouped then epilogue", k_tile<1>, blocks, d);
        run("serial: epilogue on the tile's own results", k_tile<2>, blocks, d);
        run("10 MFMAs only", k_tile<3>, blocks, d);
        run("epilogue only", k_tile<4>, blocks, d);
    }
    return 0;
}
