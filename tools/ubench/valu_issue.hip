// valu_issue.hip -- GPU-box micro-benchmark: the f32 VALU issue rate of one gfx950 SIMD as a function
// of resident waves per SIMD (1..8) and independent dependency chains per lane (1..8).  The
// sequential per-voice kernels (k_nice, k_noise_filter, frame_loop) are one dependency chain per lane;
// this says how far from the issue ceiling a given occupancy can be and whether a second voice per
// lane (a second independent chain) would help.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ILP independent chains of (mul, add) pairs -- contraction is off, so 2 VALU ops per step per chain.
template <int ILP>
__global__ void __launch_bounds__(64) k_chain(float *out, unsigned iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = a + (float)(threadIdx.x + k);
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) { x[k] = x[k] * a; x[k] = x[k] + b; }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// same, but the chain alternates v_cndmask / v_add (integer-ish select work like pulse_sample)
template <int ILP>
__global__ void __launch_bounds__(64) k_chain_sel(float *out, unsigned iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = a + (float)(threadIdx.x + k);
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) { x[k] = x[k] < b ? x[k] + a : x[k] - b; }   // cmp, add, sub, cndmask
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, bool SEL>
static void run(float *d, int waves_per_simd, hipEvent_t e0, hipEvent_t e1) {
    const unsigned iters = 4096;
    const unsigned blocks = 256 * 4 * waves_per_simd;    // one 64-lane block per wave slot
    auto launch = [&] {
        if (SEL) k_chain_sel<ILP><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
        else k_chain<ILP><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 5;
    const double ops_per_wave = (double)iters * 16 * ILP * (SEL ? 4 : 2);
    const double per_simd = ops_per_wave * waves_per_simd;
    // cycles at 2.4 GHz
    const double cyc = us * 2400.0;
    printf("%s ilp=%d waves/simd=%d  %.1f us  %.3f wave64-VALU/cycle/SIMD (at 2.4 GHz)  %.2f cycles/op/wave\n", SEL ? "sel  " : "muladd",
           ILP, waves_per_simd, us, per_simd / cyc, cyc / ops_per_wave);
}

int main() {
    float *d;
    CK(hipMalloc(&d, 256 * 4 * 16 * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w : {1, 2, 4, 8}) {
        run<1, false>(d, w, e0, e1);
        run<2, false>(d, w, e0, e1);
        run<4, false>(d, w, e0, e1);
        run<8, false>(d, w, e0, e1);
    }
    for (int w : {1, 2, 4, 8}) {
        run<1, true>(d, w, e0, e1);
        run<2, true>(d, w, e0, e1);
        run<4, true>(d, w, e0, e1);
    }
    return 0;
}
