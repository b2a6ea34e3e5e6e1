// Micro-benchmark of the Winograd conv main loop (k_conv_wino in conv.hip): per k-step 6 ds_read2_b32, 16 VALU, 8
// v_mfma_f32_16x16x4_f32 on 8 accumulators with 128 weight registers, from 8 waves per workgroup (2 per SIMD).
// Prints shader cycles per k-step (ideal: 2 waves x 8 MFMA x 32 = 512).  Build: hipcc --offload-arch=gfx950 -O3 wino_loop.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LDSR, int VALU, int MOVS, int HALF = 0>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ U, const float* __restrict__ X, float* out,
                                            long long* cyc, int iters) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 12000; i += 512) smem[i] = X[i];
    __syncthreads();
    float uw[8][16];
    for (int p = 0; p < 8; ++p)
        for (int s = 0; s < 16; ++s) uw[p][s] = U[((wave * 8 + p) * 16 + s) * 64 + lane];
    const int i16 = lane & 15, kq = lane >> 4, ph = wave >> 2;
    const float sg = ph ? -1.f : 1.f;
    const float* Pa = smem + ((i16 >> 3) * 2 + 3 * ph) * 1188 + (i16 & 7) * 132 + kq;
    const float* Pm = smem + ((i16 >> 3) * 2 + 1 + ph) * 1188 + (i16 & 7) * 132 + kq + 2;
    const float* Pc = smem + ((i16 >> 3) * 2 + 2 - ph) * 1188 + (i16 & 7) * 132 + kq;
    f4 acc[8];
    for (int p = 0; p < 8; ++p) acc[p] = (f4){0, 0, 0, 0};
    float r[12];
    for (int q = 0; q < 4; ++q) { r[q] = Pa[q * 66]; r[4 + q] = Pm[q * 66]; r[8 + q] = Pc[q * 66]; }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            float va[4], vb[4];
            if (VALU && HALF) {       // one transform row per wave (4 positions x 32 output channels): 8 VALU, 8 LDS reads
                float Xr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) Xr[q] = fmaf(sg, r[8 + q], r[q]);
                va[0] = Xr[0] - Xr[2]; va[1] = Xr[1] + Xr[2]; va[2] = Xr[2] - Xr[1]; va[3] = Xr[1] - Xr[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) vb[q] = va[q];
            } else if (VALU) {
                float Xr[4], Yr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { Xr[q] = r[q] - r[8 + q]; Yr[q] = fmaf(sg, r[8 + q], r[4 + q]); }
                va[0] = Xr[0] - Xr[2]; va[1] = Xr[1] + Xr[2]; va[2] = Xr[2] - Xr[1]; va[3] = Xr[1] - Xr[3];
                vb[0] = Yr[0] - Yr[2]; vb[1] = Yr[1] + Yr[2]; vb[2] = Yr[2] - Yr[1]; vb[3] = Yr[1] - Yr[3];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { va[q] = r[q]; vb[q] = r[4 + q]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (LDSR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r[q] = Pa[q * 66 + 4 * ((s4 + 1) & 15)]; r[8 + q] = Pc[q * 66 + 4 * ((s4 + 1) & 15)];
                    if (!HALF) r[4 + q] = Pm[q * 66 + 4 * ((s4 + 1) & 15)];
                }
            } else if (MOVS) {
#pragma unroll
                for (int q = 0; q < 12; ++q) asm volatile("" : "+v"(r[q]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[j], uw[j][s4], acc[j], 0, 0, 0);
                acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[j], uw[4 + j][s4], acc[4 + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int p = 0; p < 8; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// SHARED transform (round 5 experiment): the four channel-group waves of a transform-row pair (same ph) compute IDENTICAL
// transformed operands.  Here wave (cg, ph) transforms only k-step 4 g + cg of every chunk g of four k-steps, writes its 8 values
// per lane to an LDS ring, and every wave reads the four k-steps' operands back (2 ds_read_b128 per k-step instead of 12 ds_read_b32
// + 16 VALU); one LDS-only barrier per chunk.  Per wave and chunk: 12 raw reads + 16 VALU + 2 ds_write_b128 + 8 ds_read_b128 + 32
// MFMAs (today: 48 raw reads + 64 VALU + 32 MFMAs).
__global__ __launch_bounds__(512, 1) void k_shared(const float* __restrict__ U, const float* __restrict__ X, float* out,
                                                   long long* cyc, int iters) {
    extern __shared__ float smem[];
    float* ring = smem + 12288;                               // [2][4 k-steps][2 ph][64 lanes][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 12000; i += 512) smem[i] = X[i];
    __syncthreads();
    float uw[8][16];
    for (int p = 0; p < 8; ++p)
        for (int s = 0; s < 16; ++s) uw[p][s] = U[((wave * 8 + p) * 16 + s) * 64 + lane];
    const int i16 = lane & 15, kq = lane >> 4, ph = wave >> 2, cg = wave & 3;
    const float sg = ph ? -1.f : 1.f;
    const float* Pa = smem + ((i16 >> 3) * 2 + 3 * ph) * 1188 + (i16 & 7) * 132 + kq;
    const float* Pm = smem + ((i16 >> 3) * 2 + 1 + ph) * 1188 + (i16 & 7) * 132 + kq + 2;
    const float* Pc = smem + ((i16 >> 3) * 2 + 2 - ph) * 1188 + (i16 & 7) * 132 + kq;
    f4 acc[8];
    for (int p = 0; p < 8; ++p) acc[p] = (f4){0, 0, 0, 0};
    auto produce = [&](int g, float (&r)[12]) {               // raw reads of k-step 4 g + cg
        const int s4 = (4 * g + cg) & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) { r[q] = Pa[q * 66 + 4 * s4]; r[4 + q] = Pm[q * 66 + 4 * s4]; r[8 + q] = Pc[q * 66 + 4 * s4]; }
    };
    auto publish = [&](int g, const float (&r)[12]) {
        float Xr[4], Yr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { Xr[q] = r[q] - r[8 + q]; Yr[q] = fmaf(sg, r[8 + q], r[4 + q]); }
        f4 a = {Xr[0] - Xr[2], Xr[1] + Xr[2], Xr[2] - Xr[1], Xr[1] - Xr[3]};
        f4 b = {Yr[0] - Yr[2], Yr[1] + Yr[2], Yr[2] - Yr[1], Yr[1] - Yr[3]};
        float* d = ring + ((((g & 1) * 4 + cg) * 2 + ph) * 64 + lane) * 8;
        *(f4*)d = a;
        *(f4*)(d + 4) = b;
    };
    float r[12];
    produce(0, r);
    publish(0, r);
    __syncthreads();
    const long long t0 = clock64();
    int g = 0;
    for (int it = 0; it < iters * 4; ++it, ++g) {             // one chunk of four k-steps per trip
        produce(g + 1, r);
        const float* src = ring + (((g & 1) * 4) * 2 + ph) * 64 * 8 + lane * 8;
        f4 an = *(const f4*)(src), bn = *(const f4*)(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f4 a = an, b = bn;
            if (j < 3) { an = *(const f4*)(src + (j + 1) * 2 * 64 * 8); bn = *(const f4*)(src + (j + 1) * 2 * 64 * 8 + 4); }   // one k-step ahead
            if (j == 1) publish(g + 1, r);
            __builtin_amdgcn_sched_barrier(0);
            const int s4 = 4 * (g & 3) + j;
            // (uw index must be compile-time: unrolled over the four chunk positions below)
            switch (g & 3) {
#define MM(G)                                                                                                          \
                case G:                                                                                                \
                    for (int q = 0; q < 4; ++q) {                                                                      \
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], uw[q][4 * G + j], acc[q], 0, 0, 0);         \
                        acc[4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[q], uw[4 + q][4 * G + j], acc[4 + q], 0, 0, 0); \
                    }                                                                                                  \
                    break;
                MM(0) MM(1) MM(2) MM(3)
#undef MM
            }
            (void)s4;
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    float s = 0;
    for (int p = 0; p < 8; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, float* U, float* X, float* out, long long* cyc) {
    const int iters = 200;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 130000);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 512, 130000>>>(U, X, out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 512, 130000>>>(U, X, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += h[i];
    c /= 256;
    printf("%-34s %.3f ms  %.0f cycles per k-step (2 waves/SIMD; 512 = MFMA-bound)  clock %.2f GHz  %.1f TFLOP/s\n", name, ms,
           c / (iters * 16.0), c / (ms * 1e6), 256.0 * 8 * iters * 16 * 8 * 2048.0 / ms * 1e-9);
}
int main() {
    float *U, *X, *out;
    long long* cyc;
    hipMalloc(&U, 64 * 16 * 64 * 4); hipMalloc(&X, 12000 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    float* h = (float*)malloc(64 * 16 * 64 * 4);
    for (int i = 0; i < 64 * 16 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    hipMemcpy(U, h, 64 * 16 * 64 * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 12000; ++i) h[i] = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(X, h, 12000 * 4, hipMemcpyHostToDevice);
    run("MFMA only", k<0, 0, 0>, U, X, out, cyc);
    run("MFMA + movs", k<0, 0, 1>, U, X, out, cyc);
    run("MFMA + VALU", k<0, 1, 1>, U, X, out, cyc);
    run("MFMA + LDS reads", k<1, 0, 0>, U, X, out, cyc);
    run("MFMA + LDS reads + VALU (real)", k<1, 1, 0>, U, X, out, cyc);
    run("MFMA + 8 LDS reads + 8 VALU", k<1, 1, 0, 1>, U, X, out, cyc);
    run("shared transform via LDS ring", k_shared, U, X, out, cyc);
    return 0;
}
