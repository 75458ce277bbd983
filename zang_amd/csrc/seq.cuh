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

// f(frame, x[NIN], value&) -> bool painted.  Called for consecutive frames in order; it may
// carry state by reference capture.  painted == false leaves out[frame] untouched (ADD) or
// zero (ZERO_FIRST), like a reference loop that `continue`s or stops early.
template <int CH, bool ZF, int NIN, class F>
__device__ __forceinline__ void frame_loop(float *__restrict__ out, size_t ostride,
                                           const float *const *in, const size_t *istride,
                                           uint32_t start, uint32_t end, F &&f) {
    constexpr int NI = NIN > 0 ? NIN : 1;
    const uint32_t n = end - start;
    const uint32_t nfull = n / CH;
    float oc[CH], xc[NI][CH];
    uint32_t i = start;
    if (nfull > 0) {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            if (!ZF) oc[k] = out[(size_t)(i + k) * ostride];
#pragma unroll
            for (int j = 0; j < NIN; j++) xc[j][k] = in[j][(size_t)(i + k) * istride[j]];
        }
    }
    for (uint32_t c = 0; c < nfull; c++, i += CH) {
        float on[CH], xn[NI][CH];
        const bool more = c + 1 < nfull;
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if (!ZF) on[k] = out[(size_t)(i + CH + k) * ostride];
#pragma unroll
                for (int j = 0; j < NIN; j++) xn[j][k] = in[j][(size_t)(i + CH + k) * istride[j]];
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            float x[NI];
#pragma unroll
            for (int j = 0; j < NIN; j++) x[j] = xc[j][k];
            float val = 0.0f;
            const bool painted = f(i + k, x, val);
            float o = ZF ? 0.0f : oc[k];
            if (painted) o = o + val;
            if (ZF || painted) out[(size_t)(i + k) * ostride] = o;
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
        float x[NI];
#pragma unroll
        for (int j = 0; j < NIN; j++) x[j] = in[j][(size_t)i * istride[j]];
        float val = 0.0f;
        const bool painted = f(i, x, val);
        float o = ZF ? 0.0f : out[(size_t)i * ostride];
        if (painted) o = o + val;
        if (ZF || painted) out[(size_t)i * ostride] = o;
    }
}

// zero the span of one voice column (used when a ZERO_FIRST paint paints nothing)
__device__ __forceinline__ void zero_column(float *__restrict__ out, size_t ostride, uint32_t start, uint32_t end) {
    for (uint32_t i = start; i < end; i++) out[(size_t)i * ostride] = 0.0f;
}
