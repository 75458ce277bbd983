// nf_loop.hip -- what do the frame loops of the tolerant Noise -> Filter passes cost, piece by piece?  (round 5)
// 512 workgroups x 256 threads (2,048 waves = two per SIMD, the tolerant launches' shape), every lane 32 frames:
//   noise      xoshiro256++ + Random.float conversion, tile-wise test (noise_tile8)
//   svf        the 2x-oversampled state-variable step over a given input
//   both       noise feeding svf (pass A's loop)
//   both+out   ... + the output mix and a store per frame (pass B's loop)
// and the same at 1 and 4 waves per SIMD (grid / 2, grid x 2).  Prints microseconds per launch and cycles per frame per wave.
#include "../../zang_amd/csrc/common.hip.h"
#include "../../zang_amd/csrc/zmath.hip.h"
#include "../../zang_amd/csrc/dsp.hip.h"
#include "../../zang_amd/csrc/lanes.hip.h"
#include "../../zang_amd/csrc/noise_jump.hip.h"
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void tile8(ZXoshiro &r, float (&t)[8], bool &multi) {
    const ZXoshiro r0 = r;
    uint32_t hmin = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = 0.0f + (zrandom_float32_common(r, hmin) * 2.0f - 1.0f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hmin == 0u) != 0, 0)) {
        r = r0;
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = 0.0f + (zrandom_float32_multi(r, multi) * 2.0f - 1.0f);
    }
}

template <int KIND, int FRAMES>
__global__ void __launch_bounds__(256) k(float *out, uint32_t stride, const float *cutp, const float *resp) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    ZXoshiro r;
    zxoshiro_seed(r, v);
    const float cut = cutp[v & 4095], res = resp[v & 4095];
    float l = 0.0f, b = 0.0f, acc = 0.0f;
    bool multi = false;
    const uint32_t voff = (v & 4095) * 4u, orow = stride * 4u;
    for (uint32_t k0 = 0; k0 < FRAMES; k0 += 8) {
        float t[8];
        if (KIND == 1) {                 // svf only: inputs from registers
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = acc * 0.5f + (float)q;
        } else tile8(r, t, multi);
        if (KIND == 0) {
#pragma unroll
            for (int q = 0; q < 8; q++) acc += t[q];
        } else if (KIND == 1 || KIND == 2) {
#pragma unroll
            for (int q = 0; q < 8; q++) svf_step(l, b, t[q], cut, res);
        } else {
            const zh_rsrc_t ro = zrow_rsrc(out, stride, (blockIdx.x >> 4) * FRAMES + k0);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const SvfOut sv = svf_step(l, b, t[q], cut, res);
                zrow_store<1>(ro, voff, q * orow, 0.0f + (sv.l * 1.0f + sv.b * 0.0f + sv.h * 0.0f));
            }
        }
    }
    if (KIND != 3 || multi) out[(size_t)(v & 4095)] = l + b + acc + (multi ? 1.0f : 0.0f);
}

template <int KIND> static void run(const char *name, int wgs, float *out, const float *c, const float *rr, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> us;
    for (int rep = 0; rep < 12; rep++) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k<KIND, 32>), dim3(wgs), dim3(256), 0, st, out, 4096u, c, rr);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) us.push_back(ms * 1e3f / 50);
    }
    std::sort(us.begin(), us.end());
    const double med = us[us.size() / 2];
    const int waves_per_simd = wgs * 4 / 1024;
    printf("%-10s %4d workgroups (%d waves/SIMD): %7.2f us per launch = %6.1f cycles per frame per wave at 2.4 GHz (incl. ~2 us of launch)\n", name, wgs,
           waves_per_simd, med, med * 2400.0 / 32.0);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *out, *c, *r;
    CK(hipMalloc(&out, (size_t)4096 * 1024 * 4 * 2)); CK(hipMalloc(&c, 4096 * 4)); CK(hipMalloc(&r, 4096 * 4));
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = 0.02f + 0.5f * (i % 97) / 97.0f;
    CK(hipMemcpy(c, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 4096; i++) h[i] = 0.1f + 0.9f * (i % 89) / 89.0f;
    CK(hipMemcpy(r, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int wgs : {256, 512, 1024}) {
        run<0>("noise", wgs, out, c, r, st);
        run<1>("svf", wgs, out, c, r, st);
        run<2>("both", wgs, out, c, r, st);
        run<3>("both+out", wgs, out, c, r, st);
    }
    return 0;
}
