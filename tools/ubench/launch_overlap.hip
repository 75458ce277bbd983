// launch_overlap.hip -- can consecutive 16 MiB paint-shaped kernels overlap their ramps and tails, and through which launch form?
// (round 5)  The kernel has the headline kernel's shape: grid (16, 64) x 256 threads, every lane loads
// 8 x 16 bytes of per-voice constants, computes ~100 VALU instructions per frame for 4 frames of 4 voices and stores 4 x 16 bytes
// write-through; 20 launches into 20 distinct 16 MiB images of a 32-image ring.  Forms:
//   chain      hipLaunchKernelGGL x 20 on one stream (every AQL packet carries the barrier bit)
//   anyorder   hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) x 20 on one stream (no barrier bit), then an ordinary launch
//   streams N  round-robin over N streams, fork / join by events
//   graph      the chain captured into a hipGraph
//   graphN     the N-stream form captured into a hipGraph
// Prints microseconds per 20-launch region (median / min of 200 regions, wall clock around launch + hipStreamSynchronize, and HIP events).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_paint(float *img, const uint32_t *tab, uint32_t V, uint32_t stride, uint32_t fbase0) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v = blockIdx.x * 256 + lane * 4;
    uint4 w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = *reinterpret_cast<const uint4 *>(tab + (size_t)j * V + v);
    const uint32_t chunk = blockIdx.y * 4 + wave;
    const uint32_t c0 = chunk * 4;
    uint32_t cnt[4] = {w[7].x + (fbase0 + c0) * w[0].x, w[7].y + (fbase0 + c0) * w[0].y, w[7].z + (fbase0 + c0) * w[0].z, w[7].w + (fbase0 + c0) * w[0].w};
    const uint32_t ifr[4] = {w[0].x, w[0].y, w[0].z, w[0].w};
    const float g[4] = {__uint_as_float(w[2].x), __uint_as_float(w[2].y), __uint_as_float(w[2].z), __uint_as_float(w[2].w)};
    float *o = img + (size_t)c0 * stride + v;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(img + (size_t)c0 * stride, 0, 4u * stride * 4u, 0x00020000);
    uint32_t boff = lane * 16 + blockIdx.x * 1024;
#pragma unroll
    for (int i = 0; i < 4; i++, boff += stride * 4) {
        float val[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float p = __uint_as_float((cnt[j] >> 9) | 0x3f800000u) - 1.0f;
            float x = p;
#pragma unroll
            for (int q = 0; q < 10; q++) x = x * g[j] + (cnt[j] < ifr[j] ? p : 0.7f);      // ~25 instructions per sample
            val[j] = x;
            cnt[j] += ifr[j];
        }
        v4f acc = {val[0], val[1], val[2], val[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, acc), rs, boff, 0, 16);
    }
    (void)o;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const uint32_t V = 4096, F = 1024, K = 20, R = 32;
    const int regions = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<float *> ring(R);
    for (auto &p : ring) CK(hipMalloc(&p, (size_t)V * F * 4));
    uint32_t *tab;
    CK(hipMalloc(&tab, (size_t)8 * V * 4));
    std::vector<uint32_t> h(8 * V);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) | 0x3f000000u;
    CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join[4];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto &x : join) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    const dim3 grid(V / 256, F / 16), block(256);

    auto chain = [&](hipStream_t st) { for (uint32_t k = 0; k < K; k++) hipLaunchKernelGGL(k_paint, grid, block, 0, st, ring[k], tab, V, V, k * F); };
    auto anyorder = [&](hipStream_t st) {
        for (uint32_t k = 0; k < K; k++) hipExtLaunchKernelGGL(k_paint, grid, block, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ring[k], tab, V, V, k * F);
    };
    auto streams = [&](int n) {
        CK(hipEventRecord(fork, s[0]));
        for (int i = 1; i < n; i++) CK(hipStreamWaitEvent(s[i], fork, 0));
        for (uint32_t k = 0; k < K; k++) hipLaunchKernelGGL(k_paint, grid, block, 0, s[k % n], ring[k], tab, V, V, k * F);
        for (int i = 1; i < n; i++) { CK(hipEventRecord(join[i], s[i])); CK(hipStreamWaitEvent(s[0], join[i], 0)); }
    };
    auto capture = [&](auto fn) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        fn();
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        return ex;
    };
    hipGraphExec_t g_chain = capture([&] { chain(s[0]); });
    hipGraphExec_t g_any = capture([&] { anyorder(s[0]); });
    hipGraphExec_t g2 = capture([&] { streams(2); }), g3 = capture([&] { streams(3); }), g4 = capture([&] { streams(4); });

    auto measure = [&](const char *name, auto fn) {
        std::vector<double> wall, ev;
        for (int r = 0; r < regions + 20; r++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            CK(hipEventRecord(e0, s[0]));
            fn();
            CK(hipEventRecord(e1, s[0]));
            CK(hipStreamSynchronize(s[0]));
            const double t1 = now_us();
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 20) { wall.push_back(t1 - t0); ev.push_back(ms * 1e3); }
        }
        std::sort(wall.begin(), wall.end()); std::sort(ev.begin(), ev.end());
        printf("%-12s wall median %7.2f min %7.2f us | events median %7.2f min %7.2f us | per launch (events median) %.3f us = %.0f GB/s\n", name,
               wall[wall.size() / 2], wall[0], ev[ev.size() / 2], ev[0], ev[ev.size() / 2] / K, 16.777216e6 / (ev[ev.size() / 2] / K * 1e-6) / 1e9);
    };
    for (int pass = 0; pass < 2; pass++) {
        measure("chain", [&] { chain(s[0]); });
        measure("anyorder", [&] { anyorder(s[0]); });
        measure("streams2", [&] { streams(2); });
        measure("streams3", [&] { streams(3); });
        measure("streams4", [&] { streams(4); });
        measure("graph", [&] { CK(hipGraphLaunch(g_chain, s[0])); });
        measure("graph_any", [&] { CK(hipGraphLaunch(g_any, s[0])); });
        measure("graph2", [&] { CK(hipGraphLaunch(g2, s[0])); });
        measure("graph3", [&] { CK(hipGraphLaunch(g3, s[0])); });
        measure("graph4", [&] { CK(hipGraphLaunch(g4, s[0])); });
    }
    return 0;
}
