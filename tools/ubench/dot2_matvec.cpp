// Micro-benchmark behind csrc/grec.hip: one workgroup (4 waves, one per SIMD, up to 512 VGPRs each) keeps a whole
// 768 x 256 bf16 matrix in registers - thread u holds rows u, 256 + u, 512 + u as 384 packed registers - and per "time step"
// reads the 256-vector h from LDS (broadcast ds_read_b128), forms its three dot products with v_dot2c_f32_bf16
// (__builtin_amdgcn_fdot2_f32_bf16), writes one new h value, barrier.  Reports ns per step for NACC accumulators per gate,
// and the bare dot2 issue rate.  Build: hipcc --offload-arch=gfx950 -O3 dot2_matvec.cpp -o dot2_matvec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

__device__ __forceinline__ bf2 as_bf2(unsigned int v) { return __builtin_bit_cast(bf2, v); }

template <int NACC>
__global__ __launch_bounds__(256) void k_step(const unsigned int* __restrict__ w, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) unsigned int hs[2][128];
    const int u = threadIdx.x;
    unsigned int wr[3][128];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const u4 v = *(const u4*)(w + ((size_t)(g * 32 + c) * 256 + u) * 4);
            wr[g][4 * c] = v.x; wr[g][4 * c + 1] = v.y; wr[g][4 * c + 2] = v.z; wr[g][4 * c + 3] = v.w;
        }
    if (u < 128) { hs[0][u] = 0x3c003c00u + u; hs[1][u] = 0; }
    __syncthreads();
    float hprev = 0.f;
    for (int s = 0; s < steps; ++s) {
        float acc[3][NACC];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[g][a] = 0.f;
        const unsigned int* hb = hs[s & 1];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const u4 h4 = *(const u4*)(hb + 4 * c);
            const unsigned int hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g][(4 * c + q) % NACC] = __builtin_amdgcn_fdot2_f32_bf16(as_bf2(wr[g][4 * c + q]), as_bf2(hv[q]), acc[g][(4 * c + q) % NACC], false);
        }
        float gs[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) { gs[g] = 0.f; for (int a = 0; a < NACC; ++a) gs[g] += acc[g][a]; }
        const float r = __builtin_amdgcn_rcpf(1.f + __expf(-gs[0])), z = __builtin_amdgcn_rcpf(1.f + __expf(-gs[1]));
        const float n = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * (r * gs[2]))) - 1.f;
        const float h = (1.f - z) * n + z * hprev;
        hprev = h;
        ((__bf16*)hs[(s + 1) & 1])[u] = (__bf16)h;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    out[blockIdx.x * 256 + u] = hprev;
}

// 512 threads (two waves per SIMD): thread (unit u = t >> 1, k half kh = t & 1) holds 3 x 128 weights = 192 packed
// registers - under the 256 ARCH VGPRs a VALU instruction can address (the 384-register variant above parks a third of the
// weights in AGPRs and pays a v_accvgpr_read per use); the two halves meet through one DPP quad_perm add per gate.
template <int NACC>
__global__ __launch_bounds__(512) void k_step2(const unsigned int* __restrict__ w, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) unsigned int hs[2][128];
    const int t = threadIdx.x, u = t >> 1, kh = t & 1;
    unsigned int wr[3][64];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const u4 v = *(const u4*)(w + ((size_t)(g * 32 + 16 * kh + c) * 256 + u) * 4);
            wr[g][4 * c] = v.x; wr[g][4 * c + 1] = v.y; wr[g][4 * c + 2] = v.z; wr[g][4 * c + 3] = v.w;
        }
    if (t < 128) { hs[0][t] = 0x3c003c00u + t; hs[1][t] = 0; }
    __syncthreads();
    float hprev = 0.f;
    for (int s = 0; s < steps; ++s) {
        float acc[3][NACC];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[g][a] = 0.f;
        const unsigned int* hb = hs[s & 1] + 64 * kh;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const u4 h4 = *(const u4*)(hb + 4 * c);
            const unsigned int hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g][(4 * c + q) % NACC] = __builtin_amdgcn_fdot2_f32_bf16(as_bf2(wr[g][4 * c + q]), as_bf2(hv[q]), acc[g][(4 * c + q) % NACC], false);
        }
        float gs[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            gs[g] = 0.f;
            for (int a = 0; a < NACC; ++a) gs[g] += acc[g][a];
            gs[g] += __shfl_xor(gs[g], 1);
        }
        const float r = __builtin_amdgcn_rcpf(1.f + __expf(-gs[0])), z = __builtin_amdgcn_rcpf(1.f + __expf(-gs[1]));
        const float n = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * (r * gs[2]))) - 1.f;
        const float h = (1.f - z) * n + z * hprev;
        hprev = h;
        if (kh == 0) ((__bf16*)hs[(s + 1) & 1])[u] = (__bf16)h;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    out[blockIdx.x * 512 + t] = hprev;
}

__global__ __launch_bounds__(256) void k_rate(float* out, int iters, unsigned int a0) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int a = a0 + threadIdx.x, b = 0x3c003c00u;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 64; ++u) acc[u & 7] = __builtin_amdgcn_fdot2_f32_bf16(as_bf2(a), as_bf2(b), acc[u & 7], false);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    unsigned int* w;
    float* o;
    hipMalloc(&w, 3 * 32 * 256 * 16);
    hipMemset(w, 0x3c, 3 * 32 * 256 * 16);
    hipMalloc(&o, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    auto time = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; };
    const int steps = 2000;
    for (int blocks : {48, 96, 256}) {
        printf("blocks %3d  1 acc/gate: %.0f ns/step", blocks, time([&] { k_step<1><<<blocks, 256>>>(w, o, steps); }) * 1e6 / steps);
        printf("   2 acc/gate: %.0f ns/step", time([&] { k_step<2><<<blocks, 256>>>(w, o, steps); }) * 1e6 / steps);
        printf("   4 acc/gate: %.0f ns/step\n", time([&] { k_step<4><<<blocks, 256>>>(w, o, steps); }) * 1e6 / steps);
    }
    for (int blocks : {48, 96, 256}) {
        printf("512 threads: blocks %3d  1 acc/gate: %.0f ns/step", blocks, time([&] { k_step2<1><<<blocks, 512>>>(w, o, steps); }) * 1e6 / steps);
        printf("   2 acc/gate: %.0f ns/step", time([&] { k_step2<2><<<blocks, 512>>>(w, o, steps); }) * 1e6 / steps);
        printf("   4 acc/gate: %.0f ns/step\n", time([&] { k_step2<4><<<blocks, 512>>>(w, o, steps); }) * 1e6 / steps);
    }
    const int iters = 4000;
    const float t = time([&] { k_rate<<<256, 256>>>(o, iters, 1u); });
    printf("bare v_dot2c_f32_bf16: %.2f ns per instruction per wave (1 wave/SIMD, 8 independent accumulators)\n", t * 1e6 / (iters * 64.0));
    return 0;
}
