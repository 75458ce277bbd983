// store_floor.hip -- GPU-box micro-benchmark: how fast can a 16 MiB [1024][4096] f32 image be
// written by one launch, graph-replayed over a 32-image ring?  Variants: store flavour, frames per
// lane, threads per block.  Build: hipcc --offload-arch=gfx950 -O3 store_floor.hip -o store_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// lane = 4 voices, chunk of FC frames per wave; SM: 0 plain, 1 nt, 2 sc1 (buffer store), 3 sc0sc1
template <int SM>
__global__ void k_store(float *out, unsigned V, unsigned F, unsigned fc, float val) {
    const unsigned lanes = V / 4;
    const unsigned q = blockIdx.x * 64 + (threadIdx.x & 63);
    const unsigned chunk = blockIdx.y * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (q >= lanes) return;
    const unsigned c0 = chunk * fc;
    if (c0 >= F) return;
    const unsigned c1 = min(c0 + fc, F);
    const unsigned wchunk = __builtin_amdgcn_readfirstlane(chunk);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)wchunk * fc * V, 0, fc * V * 4, 0x00020000);
    unsigned boff = q * 16;
    float *o = out + (size_t)c0 * V + q * 4;
    v4f x = {val, val + 1, val + 2, val + 3};
    for (unsigned i = c0; i < c1; i++, o += V, boff += V * 4) {
        x.x += 1.0f;
        if (SM == 0) *(v4f *)o = x;
        else if (SM == 1) __builtin_nontemporal_store(x, (v4f *)o);
        else if (SM == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, x), rsrc, boff, 0, 16);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, x), rsrc, boff, 0, 17);
    }
}

// same geometry with ALU independent integer/float ops per 16-byte store (4 chains), sc1 stores
template <int ALU>
__global__ void k_store_alu(float *out, unsigned V, unsigned F, unsigned fc, float val) {
    const unsigned lanes = V / 4;
    const unsigned q = blockIdx.x * 64 + (threadIdx.x & 63);
    const unsigned chunk = blockIdx.y * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (q >= lanes) return;
    const unsigned c0 = chunk * fc;
    if (c0 >= F) return;
    const unsigned c1 = min(c0 + fc, F);
    const unsigned wchunk = __builtin_amdgcn_readfirstlane(chunk);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)wchunk * fc * V, 0, fc * V * 4, 0x00020000);
    unsigned boff = q * 16;
    v4f x = {val + q, val + 1, val + 2, val + 3};
    for (unsigned i = c0; i < c1; i++, boff += V * 4) {
#pragma unroll
        for (int k = 0; k < ALU / 4; k++) { x.x = x.x * 1.0001f + 0.5f; x.y = x.y * 0.9999f + 0.25f; x.z = x.z * 1.0002f - 0.5f; x.w = x.w * 0.9998f - 0.25f; }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, x), rsrc, boff, 0, 16);
    }
}

// the sc1 store kernel with the real kernel's prologue added step by step: LEVEL 1 = every thread loads
// its voice's freq / color / cnt (the stores depend on them), 2 = + a divide and some setup arithmetic,
// 3 = + the LDS exchange (write, barrier, 16-byte reads) that shares the setup among the block's 4 waves
template <int LEVEL>
__global__ void __launch_bounds__(256) k_store_pro(float *out, unsigned V, unsigned F, unsigned fc, const float *freq, const float *color,
                                                   const unsigned *cnt) {
    __shared__ __attribute__((aligned(16))) float sk[4][256];
    const unsigned lanes = V / 4;
    const unsigned lane = threadIdx.x & 63;
    const unsigned q = blockIdx.x * 64 + lane;
    const unsigned chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    float a = 1.0f, b = 2.0f, c = 3.0f, d = 4.0f;
    if (LEVEL >= 1) {
        const unsigned sv = blockIdx.x * 256 + threadIdx.x;
        float f = freq[sv], col = color[sv];
        unsigned cn = cnt[sv];
        if (LEVEL >= 2) { f = 0.7f / (f * 89478.0f * 2.3283064e-10f); col = f * (col - 1.0f) + 0.7f; }
        a = f; b = col; c = (float)cn; d = f * col;
        if (LEVEL >= 3) {
            sk[0][threadIdx.x] = a; sk[1][threadIdx.x] = b; sk[2][threadIdx.x] = c; sk[3][threadIdx.x] = d;
            __syncthreads();
            const v4f w0 = *(const v4f *)&sk[0][lane * 4], w1 = *(const v4f *)&sk[1][lane * 4];
            const v4f w2 = *(const v4f *)&sk[2][lane * 4], w3 = *(const v4f *)&sk[3][lane * 4];
            a = w0.x + w1.y; b = w0.y + w2.z; c = w0.z + w3.w; d = w0.w + w1.x;
        }
    }
    if (q >= lanes) return;
    const unsigned c0 = chunk * fc;
    if (c0 >= F) return;
    const unsigned c1 = min(c0 + fc, F);
    const unsigned wchunk = __builtin_amdgcn_readfirstlane(chunk);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)wchunk * fc * V, 0, fc * V * 4, 0x00020000);
    unsigned boff = q * 16;
    v4f x = {a, b, c, d};
    for (unsigned i = c0; i < c1; i++, boff += V * 4) {
        x.x += 1.0f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, x), rsrc, boff, 0, 16);
    }
}

// flat grid-stride fill, 16 B per thread (what a plain memset-like kernel does)
template <int SM>
__global__ void k_fill(float *out, size_t n4, float val) {
    v4f x = {val, val, val, val};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (SM == 0) ((v4f *)out)[i] = x;
        else __builtin_nontemporal_store(x, (v4f *)out + i);
    }
}

int main(int argc, char **argv) {
    // usage: store_floor [voices]   (default 4096 = the bench's 16 MiB image; the ring stays <= 8 GiB)
    const unsigned V = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 10) : 4096, F = 1024;
    const size_t img = (size_t)V * F * 4;
    const unsigned R = (unsigned)std::max<size_t>(2, std::min<size_t>(32, ((size_t)8 << 30) / img));
    printf("# %u voices x %u frames = %.0f MiB per launch, ring of %u images\n", V, F, img / 1048576.0, R);
    std::vector<float *> ring(R);
    for (auto &p : ring) CK(hipMalloc(&p, (size_t)V * F * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (unsigned r = 0; r < R; r++) launch(ring[r]);
        CK(hipStreamSynchronize(st));
        hipGraph_t g; hipGraphExec_t ge;
        // like bench.py: ONE graph holds every timed launch (a graph launch costs a ~4 us bubble on the stream)
        const unsigned rot = std::max(1u, 384u / R);                  // ring rotations per graph
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (unsigned q = 0; q < rot; q++)
            for (unsigned r = 0; r < R; r++) launch(ring[r]);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (rot * R);
        printf("%-44s %6.2f us/launch  %5.2f TB/s\n", name, us, (double)V * F * 4 / us / 1e6);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    char name[128];
    for (int sm = 0; sm < 4; sm++)
        for (unsigned tpb : {256u, 512u, 1024u})
            for (unsigned fc : {4u, 8u, 16u}) {
                const unsigned wpb = tpb / 64, chunks = (F + fc - 1) / fc;
                dim3 grid((V / 4 + 63) / 64, (chunks + wpb - 1) / wpb);
                snprintf(name, sizeof name, "store sm=%d tpb=%u fc=%u (%u blocks)", sm, tpb, fc, grid.x * grid.y);
                run(name, [&](float *o) {
                    if (sm == 0) hipLaunchKernelGGL(k_store<0>, grid, dim3(tpb), 0, st, o, V, F, fc, 1.0f);
                    else if (sm == 1) hipLaunchKernelGGL(k_store<1>, grid, dim3(tpb), 0, st, o, V, F, fc, 1.0f);
                    else if (sm == 2) hipLaunchKernelGGL(k_store<2>, grid, dim3(tpb), 0, st, o, V, F, fc, 1.0f);
                    else hipLaunchKernelGGL(k_store<3>, grid, dim3(tpb), 0, st, o, V, F, fc, 1.0f);
                });
            }
    for (unsigned fc : {4u, 8u}) {
        const unsigned tpb = 256, wpb = 4, chunks = (F + fc - 1) / fc;
        dim3 grid((V / 4 + 63) / 64, (chunks + wpb - 1) / wpb);
#define RUN_ALU(N) snprintf(name, sizeof name, "sc1 store + %d flops/store fc=%u", 2 * N, fc); run(name, [&](float *o) { hipLaunchKernelGGL(k_store_alu<N>, grid, dim3(tpb), 0, st, o, V, F, fc, 1.0f); });
        RUN_ALU(0) RUN_ALU(16) RUN_ALU(32) RUN_ALU(64) RUN_ALU(128)
    }
    {
        float *freq, *color; unsigned *cnt;
        CK(hipMalloc(&freq, V * 4)); CK(hipMalloc(&color, V * 4)); CK(hipMalloc(&cnt, V * 4));
        std::vector<float> h(V, 440.0f);
        CK(hipMemcpy(freq, h.data(), V * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(color, h.data(), V * 4, hipMemcpyHostToDevice));
        CK(hipMemset(cnt, 0, V * 4));
        const unsigned fc = 8, chunks = F / fc;
        dim3 grid((V / 4 + 63) / 64, (chunks + 3) / 4);
        run("sc1 store fc=8 + no prologue", [&](float *o) { hipLaunchKernelGGL(k_store_pro<0>, grid, dim3(256), 0, st, o, V, F, fc, freq, color, cnt); });
        run("sc1 store fc=8 + param loads", [&](float *o) { hipLaunchKernelGGL(k_store_pro<1>, grid, dim3(256), 0, st, o, V, F, fc, freq, color, cnt); });
        run("sc1 store fc=8 + loads + divide", [&](float *o) { hipLaunchKernelGGL(k_store_pro<2>, grid, dim3(256), 0, st, o, V, F, fc, freq, color, cnt); });
        run("sc1 store fc=8 + loads + divide + LDS share", [&](float *o) { hipLaunchKernelGGL(k_store_pro<3>, grid, dim3(256), 0, st, o, V, F, fc, freq, color, cnt); });
    }
    for (unsigned blocks : {1024u, 2048u, 4096u}) {
        snprintf(name, sizeof name, "flat fill plain, %u blocks x 256", blocks);
        run(name, [&](float *o) { hipLaunchKernelGGL(k_fill<0>, dim3(blocks), dim3(256), 0, st, o, (size_t)V * F / 4, 2.0f); });
        snprintf(name, sizeof name, "flat fill nt, %u blocks x 256", blocks);
        run(name, [&](float *o) { hipLaunchKernelGGL(k_fill<1>, dim3(blocks), dim3(256), 0, st, o, (size_t)V * F / 4, 2.0f); });
    }
    run("hipMemsetAsync", [&](float *o) { CK(hipMemsetAsync(o, 0, (size_t)V * F * 4, st)); });
    return 0;
}
