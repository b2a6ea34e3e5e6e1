// Micro-benchmark: what the pieces of one GRU recurrence step (csrc/gru4.hip) cost on this box, 48 workgroups like the
// real launch: shader clock (from a stream of independent v_fma_f32), s_barrier among W waves, LDS write -> barrier -> read
// round trip, the 24-v_pk_fma_f32 block, the DPP quad reduction, the exp / rcp gate chain.
// Build: hipcc --offload-arch=gfx950 -O3 step_latency.cpp -o step_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float quad_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));
    return x;
}

// 64 independent v_fma_f32 per iteration, one wave per workgroup: 4 cycles each -> clock = 256 / (ns per iteration)
__global__ void k_clock(float* out, int iters, float a) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, 1.0f);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 64 independent v_pk_fma_f32 per iteration
__global__ void k_pkclock(float* out, int iters, float a) {
    v2f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v2f){(float)threadIdx.x + i, 1.f};
    const v2f aa = {a, a}, one = {1.f, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(acc[i], aa, one);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// dependent chain of 64 v_fma_f32 / v_pk_fma_f32 per iteration
__global__ void k_dep(float* out, int iters, float a) {
    float x = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = __builtin_fmaf(x, a, 1.0f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_pkdep(float* out, int iters, float a) {
    v2f x = {(float)threadIdx.x, 1.f};
    const v2f aa = {a, a}, one = {1.f, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = __builtin_elementwise_fma(x, aa, one);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.x + x.y;
}
// s_barrier only
__global__ void k_barrier(float* out, int iters) {
    for (int it = 0; it < iters; ++it) asm volatile("s_barrier" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = iters;
}
// LDS write -> wait -> barrier -> 4 x ds_read_b128 of what the OTHER waves wrote -> wait -> (sum feeds the next write)
__global__ void k_ldsrt(float* out, int iters, int nread) {
    __shared__ __attribute__((aligned(16))) float buf[2][1024];
    const int tid = threadIdx.x, l = tid & 63, kq = l & 3;
    float v = tid;
    for (int it = 0; it < iters; ++it) {
        float* b = buf[it & 1];
        b[tid & 255] = v;
        lds_barrier();
        v4f s = {0, 0, 0, 0};
        for (int q = 0; q < nread; ++q) s += *(const v4f*)(b + 16 * kq + 4 * q);
        v = s.x + s.y + s.z + s.w;
    }
    out[blockIdx.x * blockDim.x + tid] = v;
}
// the same without the barrier: LDS write -> read turn-around inside one wave
__global__ void k_ldsself(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[2][1024];
    const int tid = threadIdx.x, l = tid & 63, kq = l & 3;
    float v = tid;
    float* mine = buf[0] + (tid >> 6) * 64;
    for (int it = 0; it < iters; ++it) {
        mine[l] = v;
        v4f s = {0, 0, 0, 0};
        for (int q = 0; q < 4; ++q) s += *(const v4f*)(mine + 16 * kq + 4 * q);
        v = s.x + s.y + s.z + s.w;
    }
    out[blockIdx.x * blockDim.x + tid] = v;
}
// the mat-vec block of a step: 24 v_pk_fma_f32 in 6 chains + pair sums + quad sums, result feeds the next iteration
__global__ void k_matvec(float* out, int iters, float a) {
    v2f w[24];
    for (int i = 0; i < 24; ++i) w[i] = (v2f){a * (i + 1), a * (i + 2)};
    float h = threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        v2f acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = (v2f){0.f, 0.f};
        const v2f hh = {h, h * 0.5f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_elementwise_fma(w[6 * q + i], hh, acc[i]);
        const v2f s0 = acc[0] + acc[1], s1 = acc[2] + acc[3], s2 = acc[4] + acc[5];
        h = quad_sum(s0.x + s0.y) + quad_sum(s1.x + s1.y) + quad_sum(s2.x + s2.y);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
// the gate chain: sigmoid, sigmoid (parallel), tanh(gi + r * gh), blend
__global__ void k_gates(float* out, int iters, float a) {
    float h = threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        const float r = __builtin_amdgcn_rcpf(1.0f + __expf(-(h + a)));
        const float z = __builtin_amdgcn_rcpf(1.0f + __expf(-(h - a)));
        const float n = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * (a + r * h)));
        h = (1.0f - z) * n + z * h;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}

// 48 v_fmac_f32_dpp (row_ror) in 3 chains + the four-row sum through v_permlane32_swap / v_permlane16_swap
template <int S> __device__ __forceinline__ void fmac_ror(float& acc, float h, float w) {
    if constexpr (S == 0) acc = __builtin_fmaf(h, w, acc);
    else asm("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w), "n"(S));
}
template <int S> __device__ __forceinline__ void dot_step(float (&acc)[3], const float (&h)[3], const float (&w)[48]) {
    fmac_ror<S>(acc[0], h[0], w[S]);
    fmac_ror<S>(acc[1], h[1], w[16 + S]);
    fmac_ror<S>(acc[2], h[2], w[32 + S]);
    if constexpr (S < 15) dot_step<S + 1>(acc, h, w);
}
__device__ __forceinline__ float row4_sum(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__global__ void k_dppdot(float* out, int iters, float a) {
    float w[48];
    for (int i = 0; i < 48; ++i) w[i] = a * (i + 1);
    float h = threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        float acc[3] = {0.f, 0.f, 0.f};
        const float hv[3] = {h, h * 0.5f, h * 0.25f};
        dot_step<0>(acc, hv, w);
        h = row4_sum(acc[0] + acc[1] + acc[2]);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}

template <typename F>
static void run(const char* name, int threads, int iters, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s threads %4d  %8.1f ns per iteration\n", name, threads, best * 1e6 / iters);
}

int main() {
    float* d;
    hipMalloc(&d, 48 * 1024 * 4);
    const int G = 48, N = 20000;
    run("64 independent v_fma_f32", 64, N, [&](int n) { k_clock<<<G, 64>>>(d, n, 0.999f); });
    run("64 independent v_pk_fma_f32", 64, N, [&](int n) { k_pkclock<<<G, 64>>>(d, n, 0.999f); });
    run("64 dependent v_fma_f32", 64, N, [&](int n) { k_dep<<<G, 64>>>(d, n, 0.999f); });
    run("64 dependent v_pk_fma_f32", 64, N, [&](int n) { k_pkdep<<<G, 64>>>(d, n, 0.999f); });
    for (int t : {64, 256, 576, 640})
        run("s_barrier", t, N, [&](int n) { k_barrier<<<G, t>>>(d, n); });
    for (int t : {256, 576})
        for (int nr : {0, 1, 4})
            run(nr == 0 ? "ds_write, barrier" : nr == 1 ? "ds_write, barrier, 1 ds_read_b128" : "ds_write, barrier, 4 ds_read_b128", t, N,
                [&](int n) { k_ldsrt<<<G, t>>>(d, n, nr); });
    run("ds_write, 4 ds_read_b128 (own wave)", 64, N, [&](int n) { k_ldsself<<<G, 64>>>(d, n); });
    run("ds_write, 4 ds_read_b128 (own wave) x4 waves", 256, N, [&](int n) { k_ldsself<<<G, 256>>>(d, n); });
    run("mat-vec block (24 pk_fma + sums + dpp)", 64, N, [&](int n) { k_matvec<<<G, 64>>>(d, n, 1e-3f); });
    run("mat-vec block x4 waves", 256, N, [&](int n) { k_matvec<<<G, 256>>>(d, n, 1e-3f); });
    run("48 v_fmac_dpp + row4 sum", 64, N, [&](int n) { k_dppdot<<<G, 64>>>(d, n, 1e-3f); });
    run("48 v_fmac_dpp + row4 sum x4 waves", 256, N, [&](int n) { k_dppdot<<<G, 256>>>(d, n, 1e-3f); });
    run("gate chain", 64, N, [&](int n) { k_gates<<<G, 64>>>(d, n, 0.1f); });
    return 0;
}
