// Micro-benchmark: do v_mfma_f32_32x32x2_f32 and plain / transcendental VALU instructions overlap on one SIMD?
//   (a) inside one wave: [1 MFMA + N independent VALU] repeated; perfect overlap = max(64, cost of N VALU) cycles per group
//   (b) between two waves of a SIMD: wave 0 only MFMAs, wave 1 only VALU; perfect overlap = each runs at its solo speed
// Build: hipcc --offload-arch=gfx950 -O3 overlap.cpp -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV, int KIND>      // KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_rcp_f32
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a0) {
    f16v acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x * 1e-3f + i;
    const float a = a0, b = a0 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) v[i & 7] = __builtin_fmaf(v[i & 7], 0.999f, 0.5f);
                else if (KIND == 1) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
                else v[i & 7] = __builtin_amdgcn_rcpf(v[i & 7]);
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) v[i & 7] = __builtin_fmaf(v[i & 7], 0.999f, 0.5f);
                else if (KIND == 1) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
                else v[i & 7] = __builtin_amdgcn_rcpf(v[i & 7]);
            }
        }
    }
    float s = acc0[0] + acc1[3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// no MFMA at all: the VALU stream alone
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float a0) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) v[i & 7] = __builtin_fmaf(v[i & 7], 0.999f, 0.5f);
                else if (KIND == 1) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
                else v[i & 7] = __builtin_amdgcn_rcpf(v[i & 7]);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// two waves per SIMD (512 threads): waves 0-3 MFMA only, waves 4-7 VALU only (8 MFMAs resp. 8 x NV VALU per iteration)
template <int NV, int KIND>
__global__ __launch_bounds__(512) void k_pair(float* out, int iters, float a0) {
    if (threadIdx.x < 256) {
        f16v acc0, acc1;
        for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
        const float a = a0, b = a0 + 1;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[3];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x * 1e-3f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (KIND == 0) v[i & 7] = __builtin_fmaf(v[i & 7], 0.999f, 0.5f);
                    else if (KIND == 1) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
                    else v[i & 7] = __builtin_amdgcn_rcpf(v[i & 7]);
                }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <typename K>
static float run(K kern, int threads, int iters, float* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256, threads>>>(d, 10, 1.f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<256, threads>>>(d, iters, 1.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6f / iters / 8;      // ns per group (1 MFMA + NV VALU)
}

#define ROW(NV, KIND, NAME)                                                                                         \
    printf("%-10s NV %2d : mix %7.1f ns/group   valu alone %7.1f   two waves (mfma | valu) %7.1f\n", NAME, NV,   \
           run(k_mix<NV, KIND>, 256, 2000, d), run(k_valu<NV, KIND>, 256, 2000, d), run(k_pair<NV, KIND>, 512, 2000, d))
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    printf("one v_mfma_f32_32x32x2_f32 alone is 27 ns (64 cycles)\n");
    ROW(0, 0, "fma");
    ROW(8, 0, "fma");
    ROW(16, 0, "fma");
    ROW(24, 0, "fma");
    ROW(32, 0, "fma");
    ROW(2, 1, "exp2");
    ROW(4, 1, "exp2");
    ROW(8, 1, "exp2");
    ROW(4, 2, "rcp");
    ROW(8, 2, "rcp");
    return 0;
}
