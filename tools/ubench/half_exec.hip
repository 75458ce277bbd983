// half_exec.hip -- GPU-box micro-benchmark: does a lone wave issue faster when only part of its 64 lanes is active?  The exact Noise -> Filter
// voice (config 3) is bound by ONE wave's issue cadence (~5 cycles per VALU instruction whatever it is); if a wave with 32 or 16 active lanes
// ran its dependent chain faster, giving each filter wave fewer voices (the chip is three quarters idle at 4,096 voices) would shorten it.
// One wave per workgroup, one workgroup per CU (256 blocks), a dependent chain of (mul, add) pairs; lanes >= `active` leave at once.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off half_exec.hip -o half_exec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_chain(float *out, unsigned long long *cyc, unsigned iters, unsigned active, float a, float b) {
    if (threadIdx.x >= active) return;                    // the rest of the kernel runs with EXEC = the low `active` lanes
    float x = a + (float)threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) { x = x * a; x = x + b; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 64 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    const unsigned iters = 2000;
    for (unsigned active : {64u, 48u, 32u, 16u, 1u}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain, dim3(256), dim3(64), 0, 0, out, cyc, iters, active, 1.0000001f, 1e-9f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
            double avg = 0; for (int i = 0; i < 256; i++) avg += (double)h[i]; avg /= 256;
            if (rep) printf("active lanes %2u: %.3f s_memtime ticks per VALU instruction (%.1f us for %u instructions, %.2f ns each)\n", active, avg / (iters * 64.0), ms * 1e3, iters * 64, ms * 1e6 / (iters * 64.0));
        }
    }
    return 0;
}
