// common.hip.h -- shared host/device plumbing for libzang_hip.so (gfx950 only).
#pragma once
// Under hiprtc (the zangscript kernels, script.hip) there are no libc / libstdc++ headers and no host
// runtime API: ZH_DEVICE_ONLY drops the host plumbing and keeps the device-side views.
#if defined(__HIPCC_RTC__)
#define ZH_DEVICE_ONLY 1
#endif
#include <hip/hip_runtime.h>
#if !defined(ZH_DEVICE_ONLY)
#include <stdint.h>
#include <stddef.h>
#include <new>
#endif
#include "rtc_types.hip.h"
#include "../../include/zang_hip.h"

#if !defined(ZH_DEVICE_ONLY)
#include <utility>
#include <vector>
#include <functional>
#include <memory>
#include <string>
// A module whose state is double-buffered and flips on the HOST at every paint (the chunked oscillators, osc.hip):
// a captured graph bakes in both buffer pointers, so the library records which buffer a capture started from and
// how many flips it holds, and zh_graph_launch reconciles the host-side index with that (ctx.hip).
struct zh_flipper {
    uint64_t id;                 // registry key (ctx.hip zh_flipper_*): a destroyed module is never dereferenced
    zh_ctx *ctx;
    uint32_t n;
    uint32_t *cnt[2];
    int cur;
    uint32_t words = 1;          // 32-bit words of state per voice in each of the two buffers (a buffer is [words][n])
};
struct zh_flip_use {
    uint64_t id;
    zh_flipper *f;
    int first_cur;               // f->cur when the capture first painted it
    uint32_t flips;              // paints of it inside the capture
};

// A capture recorded with ZH_CAPTURE_COALESCE (include/zang_hip.h; ctx.hip zh_epoch_*): paints that depend on nothing recorded
// before them -- the constant-frequency oscillators in table form, whose phase at any frame is the capture-entry counter plus
// frames * ifreq exactly -- are not launched when they are recorded: consecutive ones of the same module and span are held back
// and recorded as ONE launch of several buffers (grid.z).  Anything else the library records first launches what is held back
// (an "epoch" ends).
struct zh_flipper;
struct zh_co_batch {             // the paints held back: same module, span, flags and row stride; images that do not overlap
    bool active = false;
    const void *owner = nullptr;
    uint32_t start = 0, end = 0, stride = 0, key = 0, max = 0;
    bool flips = false;          // every launch of the batch flips its module's double-buffered state (the oscillators): keep the count even
    std::vector<float *> imgs;
    std::shared_ptr<void> items;  // what else the owner keeps per held paint (composite.hip: the mixdowns' params)
    std::function<void(hipStream_t, float *const *, uint32_t)> launch;
};

struct zh_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    uint32_t capture_flags;      // of the capture that is recording (0 outside one)
    bool epoch_open;             // paints are held back
    zh_co_batch co;
    uint32_t co_paints, co_launches;   // of the capture that is recording: paint calls held back, launches they became
    std::string last_form;       // kernels launched by the last entry point on this context that launched any (zh_last_form)
    bool form_fresh;             // the running entry point has not launched yet: its first launch starts the record afresh
    // scratch for the two-pass voice mixdown: [blocks][frames] partial sums.  Blocks are never freed while the
    // context lives: a captured graph keeps the pointer it was recorded with (basics.hip zh_mix_reserve).
    float *mix_partials;
    size_t mix_partials_floats;
    std::vector<float *> mix_retired;
    uint32_t graphs_live;        // graphs captured on this context and not yet destroyed: only they can still name a retired block
    std::vector<struct zh_graph *> graphs;   // ... themselves: zh_destroy clears their `ctx`, so that a graph destroyed AFTER its context touches nothing of it
    bool capturing;
    uint32_t capture_serial;     // counts the captures begun on this context (a module's pipeline chain belongs to one capture)
    std::vector<zh_flip_use> capture_log;
    // the kernels launched while the capture was recording, in order of first launch, with their counts (zh_graph_kernels): what
    // a replay of the graph runs -- a held-back batch goes out under the entry point that ends it, not under the paint call
    std::vector<std::pair<std::string, uint32_t>> capture_kernels;
    // the first error of a launch that a ZH_CAPTURE_COALESCE capture made on a held-back paint's behalf, after the paint call itself
    // had returned ZH_OK: zh_graph_end_capture returns it instead of a graph that lacks those paints
    int deferred_error = 0;
    void *noise_jump;            // xoshiro256++ jump tables (noise_jump.hip), built on first use, freed with the context
};

struct zh_graph {
    zh_ctx *ctx;
    hipGraph_t graph;
    hipGraphExec_t exec;
    std::vector<zh_flip_use> flips;
    uint32_t nodes = 0;          // nodes of the recorded graph
    uint32_t co_paints = 0, co_launches = 0;   // ZH_CAPTURE_COALESCE: paint calls held back while recording, launches they became
    std::vector<std::pair<std::string, uint32_t>> kernels;   // zh_ctx::capture_kernels of the capture
};

// Every entry point that allocates or launches runs with the context's device current and restores the caller's
// (two contexts on different GPUs in one process; a host that called hipSetDevice / torch.cuda.set_device elsewhere).
struct ZhDeviceGuard {
    int prev, want;
    explicit ZhDeviceGuard(int device) : prev(-1), want(device) {
        if (want < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
        if (prev != want) (void)hipSetDevice(want);
    }
    ~ZhDeviceGuard() { if (want >= 0 && prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    ZhDeviceGuard(const ZhDeviceGuard &) = delete;
    ZhDeviceGuard &operator=(const ZhDeviceGuard &) = delete;
};
// (ZH_GUARD also ends the open epoch of a coalescing capture: whatever the entry point records is ordered after every paint
// recorded so far.  The oscillator paints, which may join the epoch instead, use ZH_GUARD_EPOCH.)
void zh_epoch_barrier(zh_ctx *ctx);
extern thread_local zh_ctx *zh_tls_ctx;      // the context of the entry point that is running on this thread (dispatch bookkeeping)
struct ZhFormScope {                         // (entry points call entry points: only the outermost one starts a new record)
    zh_ctx *prev;
    explicit ZhFormScope(const zh_ctx *c) : prev(zh_tls_ctx) {
        if (!prev) { zh_tls_ctx = const_cast<zh_ctx *>(c); if (zh_tls_ctx) zh_tls_ctx->form_fresh = true; }
    }
    ~ZhFormScope() { zh_tls_ctx = prev; }
};
#define ZH_GUARD_EPOCH(ctxptr) const zh_ctx *_zh_gctx = (ctxptr); ZhDeviceGuard _zh_guard(_zh_gctx ? _zh_gctx->device : -1); ZhFormScope _zh_fscope(_zh_gctx)
#define ZH_LAUNCH(kernel, ...) do { zh_note_launch(zh_tls_ctx, #kernel); hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)
#define ZH_GUARD(ctxptr) ZH_GUARD_EPOCH(ctxptr); do { if (_zh_gctx && _zh_gctx->epoch_open) zh_epoch_barrier(const_cast<zh_ctx *>(_zh_gctx)); } while (0)
// ctx.hip: launch the batch that is held back (if any).  `last` = the epoch ends here: the batch may go out as two launches so
// that its module's flips in the capture stay even (osc.hip launch_osc_const)
void zh_epoch_flush_batch(zh_ctx *ctx, bool last);

// ctx.hip: registry of live flippers + the capture log
void zh_flipper_register(zh_flipper *f);
void zh_flipper_unregister(zh_flipper *f);
void zh_flipper_used(zh_flipper *f);         // call at the start of EVERY paint of a flipper module (in-place forms too)
void zh_flipper_painted(zh_flipper *f);      // call right BEFORE flipping f->cur in a paint

// dispatch.hip: the ONE table of form-selecting thresholds and range counts (rows in the order of this enum); zh_form = a row's
// value (its default, or the ZH_FORMS override); zh_range_frames = frames per range for a kernel that paints a span as several
// frame ranges at once, each replaying the cheap state walk of the frames before it (0 = paint sequentially): row `form` = number
// of ranges (-1 auto, 0 never); `target_waves` = waves the launch should reach; `max_voices` = above it the sequential form wins.
enum { ZF_SINE_RANGES, ZF_NOISE_RANGES, ZF_ENVELOPE_RANGES, ZF_SAMPLER_RANGES, ZF_DECIMATOR_RANGES, ZF_CURVE_RANGES, ZF_CYCLE_RANGES,
       ZF_PORTAMENTO_RANGES, ZF_PULSE_CTRL_RANGES, ZF_PULSE_CTRL_SUMS, ZF_TRISAW_CTRL_RANGES, ZF_TRISAW_CTRL_QUOT, ZF_PMOSC_RANGES,
       ZF_SCRIPT_RANGES, ZF_SCRIPT_RANGES_MAXV, ZF_OSC_FC, ZF_NICE_PC_MAX, ZF_NICE_PC4_MAX, ZF_NICE_WAVE_MAX, ZF_PMOSC_WAVE_MAX,
       ZF_NICE_MIX_ROLL, ZF_NICE_MIX_WG_MIN, ZF_NF_PC_MAX, ZF_NF_RING_MAX, ZF_FILTER_PC_MAX, ZF_FILTER_PC16_MAX, ZF_FILTER_PC_CTL_MAX,
       ZF_PINK_PIPE_MAX, ZF_PINK_TAPS, ZF_ECHOES_PC_MAX, ZF_DELAY_FRAMES_MAX, ZF_FILTER_TP_MAX, ZF_NF_TP_MAX, ZF_NICE_TP_MAX,
       ZF_PINK_TP_MAX, ZF_ECHOES_TP_MAX, ZF_NICE_MIX_FMA, ZF_BASICS_ROWS_MIN, ZF_SCRIPT_PC, ZF_SCRIPT_PC_MAXV, ZF_NF_TP_PIPE_FRAMES, ZF_DISTORTION_ROWS_MIN, ZF_DISTORTION_RC, ZF_COUNT };
long zh_form(int id);
bool zh_form_is_set(int id);                 // the row is overridden through ZH_FORMS
uint32_t zh_range_frames(uint32_t V, uint32_t n, int form, uint32_t target_waves, uint32_t max_voices);
// every kernel launch of the library: notes the kernel's name in the context of the entry point that is running (zh_last_form)
void zh_note_launch(zh_ctx *ctx, const char *kernel);
// a word on the NEXT launch for a capture's kernel list (zh_graph_kernels): "batch" = the instantiation that paints several buffers
// (the kernel name alone is the same: the template arguments at a launch site are names, not values)
extern thread_local const char *zh_tls_launch_detail;

// frames per group of the fused mixdown's partial layout [channel][frame / G][row][frame % G] (composite.hip writes, basics.hip
// k_mix_pass2_wide reads: one second-pass workgroup per group and channel)
constexpr int kMixGroupFrames = 8;

struct zh_event {
    hipEvent_t ev;
};
#endif

// every per-frame lambda must be inlined into frame_loop: an outlined closure forces the lane
// state (captured by reference) out of VGPRs into scratch
#define ZH_INLINE_LAMBDA __attribute__((always_inline))

#if !defined(ZH_DEVICE_ONLY)
#define ZH_TRY(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

static inline int zh_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ZH_OK : (int)e;
}
#endif

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

#if !defined(ZH_DEVICE_ONLY)
static inline Img mk_img(const zh_buf &b) { return Img{b.ptr, b.stride}; }
static inline CImg mk_cimg(const zh_buf &b) { return CImg{b.ptr, b.stride}; }
static inline F32P mk_f32(const zh_f32 &f) { return F32P{f.value, f.per_voice}; }
static inline BoolP mk_bool(const zh_bool &b) { return BoolP{b.value, b.per_voice}; }
static inline CobP mk_cob(const zh_cob &c) {
    return CobP{c.tag == ZH_COB_BUFFER ? 1u : 0u, mk_f32(c.constant), mk_cimg(c.buffer)};
}

// Argument checks shared by every paint entry point.
// Row strides are limited to 2^26 voices (256 MiB per row): the sequential kernels address a chunk of 8 rows with
// 32-bit byte offsets from the chunk's first row (lanes.hip.h zrow_*), so 8 * stride * 4 must stay below 2^32.
// (An image of 1024 such rows would be 256 GiB.)
constexpr uint32_t kMaxRowStride = 1u << 26;
static inline bool buf_covers(const zh_buf &b, uint32_t n_voices, uint32_t span_end) {
    return b.ptr != nullptr && b.voices >= n_voices && b.frames >= span_end && b.stride >= n_voices && b.stride <= kMaxRowStride;
}
// do two images share memory?  (a frame-range kernel re-reads input rows that another range may be writing when an
// input aliases the output; the sequential forms read a frame before they write it, as the reference's loops do)
static inline bool bufs_alias(const zh_buf &a, const zh_buf &b) {
    if (!a.ptr || !b.ptr) return false;
    const float *ae = a.ptr + (size_t)a.frames * a.stride, *be = b.ptr + (size_t)b.frames * b.stride;
    return a.ptr < be && b.ptr < ae;
}
static inline bool cob_aliases(const zh_cob &c, const zh_buf &b) { return c.tag == ZH_COB_BUFFER && bufs_alias(c.buffer, b); }
static inline bool cob_ok(const zh_cob &c, uint32_t n_voices, uint32_t span_end) {
    if (c.tag == ZH_COB_CONSTANT) return true;
    if (c.tag == ZH_COB_BUFFER) return buf_covers(c.buffer, n_voices, span_end);
    return false;
}

// One wave (64 voices) per workgroup for the sequential lane-per-voice kernels: at small
// voice counts this spreads the waves over as many CUs as possible.
#ifndef ZH_SEQ_BLOCK
#define ZH_SEQ_BLOCK 64
#endif
constexpr int kSeqBlock = ZH_SEQ_BLOCK;
static inline dim3 seq_grid(uint32_t n) { return dim3((n + kSeqBlock - 1) / kSeqBlock); }

template <typename T> static inline int dev_alloc(T **p, size_t count) {
    *p = nullptr;
    if (count == 0) return ZH_OK;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    return e == hipSuccess ? ZH_OK : (int)e;
}

#endif   // !ZH_DEVICE_ONLY

// ---- streaming image stores -------------------------------------------------------------
// A paint kernel leaves its whole output image dirty in the XCD L2s; with plain stores that
// data is written back at the kernel boundary, AFTER the compute phase (MI355X_MICROARCH.md
// price list, row "boundary": + bytes / 6 TB/s).  Write-through (sc1) stores drain to HBM
// while the kernel is still computing.  StoreMode selects the flavour (ZH_STORE_MODE env).
enum StoreMode { ST_PLAIN = 0, ST_NT = 1, ST_SC1 = 2, ST_SC0SC1 = 3 };

typedef float zv4f __attribute__((ext_vector_type(4)));
typedef unsigned int zv4u __attribute__((ext_vector_type(4)));

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t zh_rsrc_t;
// A buffer descriptor over [base, base + bytes): base must be wave-uniform.
__device__ __forceinline__ zh_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int SM>
__device__ __forceinline__ void store4(float *p, zh_rsrc_t rsrc, uint32_t byte_off, zv4f v) {
    if constexpr (SM == ST_PLAIN) *reinterpret_cast<zv4f *>(p) = v;
    else if constexpr (SM == ST_NT) __builtin_nontemporal_store(v, reinterpret_cast<zv4f *>(p));
    else if constexpr (SM == ST_SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zv4u, v), rsrc, byte_off, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zv4u, v), rsrc, byte_off, 0, 17);
}
template <int SM>
__device__ __forceinline__ void store1(float *p, zh_rsrc_t rsrc, uint32_t byte_off, float v) {
    if constexpr (SM == ST_PLAIN) *p = v;
    else if constexpr (SM == ST_NT) __builtin_nontemporal_store(v, p);
    else if constexpr (SM == ST_SC1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rsrc, byte_off, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rsrc, byte_off, 0, 17);
}
// an image sample written once and not read again by this kernel: non-temporal (lanes.hip.h zrow_store: the same through a descriptor)
__device__ __forceinline__ void store_row(float *p, float v) { __builtin_nontemporal_store(v, p); }
// the same write-through store through a flat global address (no descriptor: any image size, lane-varying rows)
// (inline assembly: the compiler's hazard recognizer does not look inside it.  gfx940+ needs two wait states between a store of more than
// 64 bits and a VALU write of its data registers -- without the s_nop the next instruction's result went out as the first component:
// found by tests/test_gpu_basics.py::test_basics_many_voices_default_form.)
__device__ __forceinline__ void store4_sc1(float *p, zv4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
#else   // host pass: kernels are only parsed, never run
typedef int zh_rsrc_t;
__device__ inline void store4_sc1(float *, zv4f) {}
__device__ inline void store_row(float *, float) {}
__device__ inline zh_rsrc_t make_rsrc(const void *, uint32_t) { return 0; }
template <int SM> __device__ inline void store4(float *, zh_rsrc_t, uint32_t, zv4f) {}
template <int SM> __device__ inline void store1(float *, zh_rsrc_t, uint32_t, float) {}
#endif

#if !defined(ZH_DEVICE_ONLY)
int zh_store_mode();   // ctx.hip: ZH_STORE_MODE env (default ST_SC1)
int zh_store_mode_env();   // ... -1 when the variable is not set
constexpr uint32_t kRowPadVoices = 1024;   // zh_buf_alloc: padding of rows that are a multiple of 64 KiB (zang_amd/runtime.py image_row_pad: the same)
#endif
