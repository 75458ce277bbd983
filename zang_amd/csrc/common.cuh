// common.cuh -- shared host/device plumbing for libzang_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <new>
#include "../../include/zang_hip.h"

struct zh_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    // scratch for the two-pass voice mixdown: [blocks][frames] partial sums
    float *mix_partials;
    size_t mix_partials_floats;
};

struct zh_event {
    hipEvent_t ev;
};

#define ZH_TRY(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

static inline int zh_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ZH_OK : (int)e;
}

// ---- device-side views of the ABI structs -------------------------------------------
struct Img {           // mutable [frame][voice] image
    float *p;
    uint32_t stride;
    __device__ __forceinline__ float *at(uint32_t f, uint32_t v) const { return p + (size_t)f * stride + v; }
};
struct CImg {          // read-only image
    const float *p;
    uint32_t stride;
    __device__ __forceinline__ const float *at(uint32_t f, uint32_t v) const { return p + (size_t)f * stride + v; }
};
struct F32P {          // per-voice f32 parameter
    float value;
    const float *pv;
    __device__ __forceinline__ float get(uint32_t v) const { return pv ? pv[v] : value; }
};
struct BoolP {
    uint32_t value;
    const uint8_t *pv;
    __device__ __forceinline__ bool get(uint32_t v) const { return pv ? pv[v] != 0 : value != 0; }
};
struct CobP {          // ConstantOrBuffer
    uint32_t is_buffer;
    F32P c;
    CImg b;
};

static inline Img mk_img(const zh_buf &b) { return Img{b.ptr, b.stride}; }
static inline CImg mk_cimg(const zh_buf &b) { return CImg{b.ptr, b.stride}; }
static inline F32P mk_f32(const zh_f32 &f) { return F32P{f.value, f.per_voice}; }
static inline BoolP mk_bool(const zh_bool &b) { return BoolP{b.value, b.per_voice}; }
static inline CobP mk_cob(const zh_cob &c) {
    return CobP{c.tag == ZH_COB_BUFFER ? 1u : 0u, mk_f32(c.constant), mk_cimg(c.buffer)};
}

// Argument checks shared by every paint entry point.
static inline bool buf_covers(const zh_buf &b, uint32_t n_voices, uint32_t span_end) {
    return b.ptr != nullptr && b.voices >= n_voices && b.frames >= span_end && b.stride >= n_voices;
}
static inline bool cob_ok(const zh_cob &c, uint32_t n_voices, uint32_t span_end) {
    if (c.tag == ZH_COB_CONSTANT) return true;
    if (c.tag == ZH_COB_BUFFER) return buf_covers(c.buffer, n_voices, span_end);
    return false;
}

// One wave (64 voices) per workgroup for the sequential lane-per-voice kernels: at small
// voice counts this spreads the waves over as many CUs as possible.
constexpr int kSeqBlock = 64;
static inline dim3 seq_grid(uint32_t n) { return dim3((n + kSeqBlock - 1) / kSeqBlock); }

template <typename T> static inline int dev_alloc(T **p, size_t count) {
    *p = nullptr;
    if (count == 0) return ZH_OK;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    return e == hipSuccess ? ZH_OK : (int)e;
}
