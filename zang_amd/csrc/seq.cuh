// seq.cuh -- the lane-per-voice sequential frame loop shared by the stateful paint kernels.
//
// One lane owns one voice and walks the span frame by frame (the reference's scalar loop
// with its loop-carried state, e.g. Filter.zig:124-147), state in VGPRs.  Images are
// [frame][voice], so the 64 lanes of a wave touch 256 contiguous bytes per frame.
// Frames are processed in chunks of CH: the loads of chunk k+1 (the `+=` read of the
// output and up to NIN input images) are issued before chunk k is computed, so HBM latency
// hides behind the dependent ALU chain instead of adding to it.
#pragma once
#include "common.cuh"
#include "lanes.cuh"

// f(frame, x[NIN], value&) -> painted.  Called for consecutive frames in order; it may
// carry state by reference capture.  painted == false leaves out[frame] untouched (ADD) or
// zero (ZERO_FIRST), like a reference loop that `continue`s or stops early.
// W = voices per lane (lanes.cuh): with W = 2, `out` / `in` point at the lane's first voice (even,
// 8-byte aligned), values are zf2 and `painted` is a per-voice mask.
template <int CH, bool ZF, int NIN, int W = 1, class F>
__device__ __forceinline__ void frame_loop(float *__restrict__ out, size_t ostride,
                                           const float *const *in, const size_t *istride,
                                           uint32_t start, uint32_t end, F &&f) {
    using T = typename LaneT<W>::F;
    constexpr int NI = NIN > 0 ? NIN : 1;
    const uint32_t n = end - start;
    const uint32_t nfull = n / CH;
    T oc[CH], xc[NI][CH];
    uint32_t i = start;
    if (nfull > 0) {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            if (!ZF) oc[k] = zload_f<W>(out + (size_t)(i + k) * ostride, 0);
#pragma unroll
            for (int j = 0; j < NIN; j++) xc[j][k] = zload_f<W>(in[j] + (size_t)(i + k) * istride[j], 0);
        }
    }
    for (uint32_t c = 0; c < nfull; c++, i += CH) {
        T on[CH], xn[NI][CH];
        const bool more = c + 1 < nfull;
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if (!ZF) on[k] = zload_f<W>(out + (size_t)(i + CH + k) * ostride, 0);
#pragma unroll
                for (int j = 0; j < NIN; j++) xn[j][k] = zload_f<W>(in[j] + (size_t)(i + CH + k) * istride[j], 0);
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            T x[NI];
#pragma unroll
            for (int j = 0; j < NIN; j++) x[j] = xc[j][k];
            T val = zsplat<T>(0.0f);
            const auto painted = f(i + k, x, val);
            T o = ZF ? zsplat<T>(0.0f) : oc[k];
            o = zsel(painted, o + val, o);
            if (ZF || zany(painted)) zstore_f<W>(out + (size_t)(i + k) * ostride, 0, o);
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if (!ZF) oc[k] = on[k];
#pragma unroll
                for (int j = 0; j < NIN; j++) xc[j][k] = xn[j][k];
            }
        }
    }
    for (; i < end; i++) {
        T x[NI];
#pragma unroll
        for (int j = 0; j < NIN; j++) x[j] = zload_f<W>(in[j] + (size_t)i * istride[j], 0);
        T val = zsplat<T>(0.0f);
        const auto painted = f(i, x, val);
        T o = ZF ? zsplat<T>(0.0f) : zload_f<W>(out + (size_t)i * ostride, 0);
        o = zsel(painted, o + val, o);
        if (ZF || zany(painted)) zstore_f<W>(out + (size_t)i * ostride, 0, o);
    }
}

// zero the span of one voice column (used when a ZERO_FIRST paint paints nothing)
__device__ __forceinline__ void zero_column(float *__restrict__ out, size_t ostride, uint32_t start, uint32_t end) {
    for (uint32_t i = start; i < end; i++) out[(size_t)i * ostride] = 0.0f;
}
