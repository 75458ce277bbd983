// GPU box: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/data_power tools/ubench/data_power.hip && tools/ubench/data_power
// Does the shader clock of a full-chip VALU kernel depend on the DATA it computes on?  (profiles/r04/NOTES.md 5a: the config-5 kernels'
// per-buffer time drifts inside one envelope stage, where the instruction stream does not change.)  One kernel, 2,048 waves
// (two per SIMD), a fixed stream of dependent v_mul_f32 / v_add_f32 on eight registers per lane for ~500 us, twenty launches back to back (the second round prints them one by one); the operands
// are (a) zeros, (b) small round numbers (few mantissa bits set), (c) random mantissas around 1.  clock64() counts shader
// cycles, wall_clock64() a constant 100 MHz reference: their ratio is the clock the kernel ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) k_stream(const float *__restrict__ seed, float *out, long long *clk, unsigned iters, float m, float a) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = seed[(blockIdx.x * 64 + threadIdx.x) * 8 + k];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // x = x * m + a with m just below 1 and a small: values stay in range, mantissas keep moving for random seeds
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                         "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t"
                         "v_add_f32 %0, %0, %9\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %9\n\tv_add_f32 %3, %3, %9\n\t"
                         "v_add_f32 %4, %4, %9\n\tv_add_f32 %5, %5, %9\n\tv_add_f32 %6, %6, %9\n\tv_add_f32 %7, %7, %9"
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(m), "v"(a));
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
    const int blocks = 2048, n = blocks * 64 * 8;
    std::vector<float> h(n);
    float *seed, *out; long long *clk;
    hipMalloc(&seed, n * 4); hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 16);
    std::vector<long long> hc(blocks * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char *name; int kind; float m, a; } cases[] = {
        {"zeros (x = 0, m = 0, a = 0)", 0, 0.0f, 0.0f},
        {"round numbers (x = 1, m = 1, a = 0)", 1, 1.0f, 0.0f},
        {"random mantissas (x in [1, 2), m = 0.99999994, a = 1.1920929e-07 * 0.7)", 2, 0.99999994f, 8.3446503e-08f},
        {"random mantissas, random signs (m = -0.99999994)", 2, -0.99999994f, 8.3446503e-08f},
    };
    for (int rep = 0; rep < 2; rep++)
        for (const Case &c : cases) {
            uint32_t s = 12345;
            for (int i = 0; i < n; i++) {
                s = s * 1664525u + 1013904223u;
                h[i] = c.kind == 0 ? 0.0f : c.kind == 1 ? 1.0f : 1.0f + (float)(s >> 9) / 8388608.0f;
            }
            hipMemcpy(seed, h.data(), n * 4, hipMemcpyHostToDevice);
            for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(64), 0, 0, seed, out, clk, 4000u, c.m, c.a);
            hipDeviceSynchronize();
            const int launches = 20;
            static hipEvent_t ev[launches + 1];
            if (!ev[0]) for (auto &e : ev) hipEventCreate(&e);
            hipEventRecord(ev[0], 0);
            for (int l = 0; l < launches; l++) {
                hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(64), 0, 0, seed, out, clk, 4000u, c.m, c.a);
                hipEventRecord(ev[l + 1], 0);
            }
            hipEventSynchronize(ev[launches]);
            float ms = 0; hipEventElapsedTime(&ms, ev[0], ev[launches]);
            if (rep == 1) {                                           // launch by launch: does the time step up as the run goes on?
                printf("    per launch (us):");
                for (int l = 0; l < launches; l++) { float t = 0; hipEventElapsedTime(&t, ev[l], ev[l + 1]); printf(" %.0f", t * 1e3); }
                printf("\n");
            }
            hipMemcpy(hc.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, ref = 0;
            for (int b = 0; b < blocks; b++) { cyc += (double)hc[b * 2]; ref += (double)hc[b * 2 + 1]; }
            printf("%-78s %8.1f us per launch, shader clock %.0f MHz\n", c.name, ms * 1e3 / launches, cyc / ref * 100.0);
        }
    return 0;
}
