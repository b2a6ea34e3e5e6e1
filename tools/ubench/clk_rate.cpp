// What clock64() (s_memtime) counts: ticks per microsecond of wall clock (wall_clock64: 100 MHz) for (a) one idle-spinning wave
// and (b) the whole chip running dense fp64 FMAs at 12 waves per CU - the shader clock the power management actually grants.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clk_rate.cpp -o tools/ubench/clk_rate && tools/ubench/clk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_spin(unsigned long long* out, int iters, int heavy) {
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5, d = 0.25, e = 0.125;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (heavy) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { a = fma(a, b, c); d = fma(d, b, e); c = fma(c, b, a); e = fma(e, b, d); }
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (a + c + d + e == 12345.678) out[2] = 1;
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    for (int heavy = 0; heavy < 2; ++heavy) {
        const int grid = heavy ? 256 : 1, block = heavy ? 768 : 64, iters = heavy ? 4000 : 40000;
        k_spin<<<grid, block>>>(d, iters, heavy);
        hipDeviceSynchronize();
        k_spin<<<grid, block>>>(d, iters, heavy);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%s: clock64 ticks %llu over %.1f us of wall clock = %.3f GHz\n", heavy ? "dense fp64 FMA, 256 x 768 threads" : "one sleeping wave",
               h[0], h[1] / 100.0, h[0] / (h[1] * 10.0));
    }
    return 0;
}
