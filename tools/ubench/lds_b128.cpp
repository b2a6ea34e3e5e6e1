// Micro-benchmark: throughput of ds_read_b128 for the lane -> address maps the bf16 MFMA operand reads use
// (lane (n = lane & 31, kh = lane >> 5) reads 16 bytes at row n * stride + 16 * kh + 32 * kstep), as a function of the row
// stride, with 1 or 4 waves per CU issuing.  Build: hipcc --offload-arch=gfx950 -O3 lds_b128.cpp -o lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(float* out, int iters, int stride_bytes, int kh_bytes, int rows_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 40960 / 4; i += blockDim.x) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const int n = lane & rows_mask, kh = lane >> 5;
    const unsigned addr = (unsigned)(size_t)(lds - lds) + n * stride_bytes + kh * kh_bytes;       // LDS byte address (dynamic LDS starts at 0)
    v4f r0, r1, r2, r3, r4, r5, r6, r7, acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // eight reads in flight, then one wait: the compiler cannot hoist or merge volatile asm
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\t"
                     "ds_read_b128 %3, %8 offset:96\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:4128\n\t"
                     "ds_read_b128 %6, %8 offset:4160\n\tds_read_b128 %7, %8 offset:4192\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
        acc += r0 + r7;
    }
    out[blockIdx.x * blockDim.x + tid] = acc.x + acc.y + acc.z + acc.w + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x;
}

static void run(const char* name, int threads, int stride, int khb, int mask, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_read<<<256, threads, 40960>>>(d, 10, stride, khb, mask);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k_read<<<256, threads, 40960>>>(d, iters, stride, khb, mask);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double reads_per_cu = (double)iters * 8 * (threads / 64);
    const double ns = best * 1e6 / reads_per_cu;
    printf("%-44s waves %d  %6.2f ns per wave-read  = %6.1f B/ns per CU\n", name, threads / 64, ns, 1024.0 / ns);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 4);
    for (int threads : {64, 256}) {
        run("stride 144 B (KC 64 + 8 bf16), kh +16", threads, 144, 16, 31, d);
        run("stride 272 B (C 128 + 8 bf16), kh +16", threads, 272, 16, 31, d);
        run("stride 160 B, kh +16", threads, 160, 16, 31, d);
        run("stride 128 B (no pad), kh +16", threads, 128, 16, 31, d);
        run("stride 132 B, kh +16 (unaligned rows)", threads, 136, 16, 31, d);
        run("stride 16 B (fully contiguous), kh +512", threads, 16, 512, 31, d);
        run("stride 144 B, kh in another 4.6 KB block", threads, 144, 4608, 31, d);
        run("all lanes same address (broadcast)", threads, 0, 0, 31, d);
        run("16 rows only (n & 15), stride 144", threads, 144, 16, 15, d);
    }
    return 0;
}
