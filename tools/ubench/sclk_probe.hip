// sclk_probe.hip -- GPU-box micro-benchmark: the shader clock a kernel actually runs at as a function of
// how many waves it launches (a handful of voices keeps 17 of 1024 SIMDs busy: does the chip clock down?),
// and the issue interval of a lone wave there.  clock64() counts shader cycles (s_memtime),
// wall_clock64() a constant 100 MHz reference (s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 sclk_probe.hip -o sclk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_probe(float *out, long long *clk, unsigned iters, float a) {
    float x = a + (float)threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
    float *d; long long *c;
    const int maxb = 8192;
    CK(hipMalloc(&d, maxb * 64 * 4));
    CK(hipMalloc(&c, maxb * 16));
    long long *h = (long long *)malloc(maxb * 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned iters = 4096;                       // 262,144 dependent v_add per wave
    for (int rep = 0; rep < 2; rep++)
        for (int blocks : {1, 17, 64, 256, 1024, 2048, 8192}) {
            CK(hipEventRecord(e0));
            k_probe<<<blocks, 64>>>(d, c, iters, 1.0f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h, c, blocks * 16, hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (int b = 0; b < blocks; b++) { cyc += h[b * 2]; wall += h[b * 2 + 1]; }
            cyc /= blocks; wall /= blocks;
            printf("blocks=%5d  kernel %8.1f us  per wave: %10.0f shader cycles, %8.1f us (100 MHz ref) => sclk %.0f MHz, %.2f cycles/instr, %.2f ns/instr\n",
                   blocks, ms * 1e3, cyc, wall / 100.0, cyc / (wall / 100.0), cyc / (iters * 64.0), wall * 10.0 / (iters * 64.0));
        }
    return 0;
}
