// osc.hip -- the farbrausch-v2 band-limited oscillators: PulseOsc (src/modules/PulseOsc.zig)
// and TriSawOsc (src/modules/TriSawOsc.zig).
//
// Constant-frequency paths: the phase is a wrapping u32 accumulator, cnt_i = cnt_0 + i*ifreq
// EXACTLY (PulseOsc.zig:111, TriSawOsc.zig:115), and the rolling 2-bit `state` is a function
// of cnt_i and cnt_i - ifreq only (:96,:100 == :141-142).  So a sample depends on nothing
// but its index: the kernel keeps one lane per voice but also splits the span into frame
// chunks across waves (grid.y), which is what fills the chip at small voice counts
// (4096 voices = 64 waves otherwise).  State is double-buffered so chunk threads can read
// cnt_0 while the chunk-0 thread publishes cnt_0 + n*ifreq.
//
// Controlled-frequency paths carry state sample to sample and use the sequential
// lane-per-voice loop (seq.cuh).
#include "common.cuh"
#include "zmath.cuh"
#include "dsp.cuh"
#include "seq.cuh"
#include "voices.cuh"
#include <stdlib.h>
#include <vector>

struct zh_pulseosc {
    zh_ctx *ctx;
    uint32_t n;
    uint32_t *cnt[2];
    int cur;
};

struct zh_trisawosc {
    zh_ctx *ctx;
    uint32_t n;
    uint32_t *cnt[2];
    float *t;
    int cur;
};

// ------------------------------------------------------------------ PulseOsc
struct PulseOscP {        // policy for the chunked kernels
    using K = PulseK;
    static __device__ __forceinline__ void setup(K &k, float srf, float freq, float color) {
        pulse_setup_color(k, color);
        pulse_setup_freq(k, srf, freq);
    }
    static __device__ __forceinline__ float sample(const K &k, uint32_t cnt) { return pulse_sample(k, cnt); }
    // walking consecutive frames: the previous frame's half-period bit is carried instead of recomputed
    static constexpr bool kShortChunks = true;              // osc_frames_per_lane
    using R = PulseRoll;
    static __device__ __forceinline__ R roll_init(const K &k, uint32_t cnt) { return pulse_roll_init(k, cnt); }
    static __device__ __forceinline__ float sample_roll(const K &k, uint32_t cnt, R &r) { return pulse_sample_roll(k, cnt, r); }
};

struct TriSawOscP;        // defined below

// grid: x = 64-voice groups, y = groups of 4 frame chunks; block = 256 = 4 waves, each wave a
// different chunk of the same 64 voices.
template <class OSC, bool ZF>
__global__ void __launch_bounds__(256) k_osc_const(const uint32_t *__restrict__ cnt_in, uint32_t *__restrict__ cnt_out,
                                                        uint32_t V, Img out, uint32_t start, uint32_t end, uint32_t fc,
                                                        float srf, float sr8, F32P freq_p, F32P color_p) {
    const uint32_t v = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const uint32_t c0 = start + chunk * fc;
    const uint32_t c1 = min(c0 + fc, end);
    const float freq = freq_p.get(v);
    const bool bad = freq < 0 || freq > sr8;                          // PulseOsc.zig:82-84
    const uint32_t cnt0 = cnt_in[v];
    typename OSC::K k;
    OSC::setup(k, srf, freq, color_p.get(v));
    if (chunk == 0) cnt_out[v] = bad ? cnt0 : cnt0 + (end - start) * k.ifreq;
    if (c0 >= end) return;
    float *o = out.at(c0, v);
    const size_t os = out.stride;
    if (bad) {
        if (ZF) for (uint32_t i = c0; i < c1; i++, o += os) *o = 0.0f;
        return;
    }
    uint32_t cnt = cnt0 + (c0 - start) * k.ifreq;
    typename OSC::R roll = OSC::roll_init(k, cnt);
#pragma unroll 4
    for (uint32_t i = c0; i < c1; i++, o += os) {
        const float val = OSC::sample_roll(k, cnt, roll);
        *o = (ZF ? 0.0f : *o) + val;
        cnt += k.ifreq;
    }
}

// 4 voices per lane: one 16-byte store per frame (1 KiB per wave-instruction) and four
// independent dependency chains per lane to cover the VALU->VCC wait states.
// grid: x = 256-voice groups, y = groups of 4 frame chunks; each of the 4 waves of a block
// renders a different chunk of the same 256 voices.
template <class OSC, bool ZF, int SM>
__global__ void __launch_bounds__(256) k_osc_const4(const uint32_t *__restrict__ cnt_in, uint32_t *__restrict__ cnt_out,
                                                    uint32_t V, Img out, uint32_t start, uint32_t end, uint32_t fc,
                                                    float srf, float sr8, F32P freq_p, F32P color_p) {
    // The 4 waves of a block render 4 different frame chunks of the SAME 256 voices, so the per-voice
    // setup (a divide and some conversions) is done once per block: thread t sets up voice base + t and
    // parks the constants in LDS, then every lane fetches its 4 voices' constants with 16-byte reads.
    using K = typename OSC::K;
    constexpr int KW = sizeof(K) / 4;                     // dwords of per-voice constants
    __shared__ __attribute__((aligned(16))) uint32_t sk[KW + 2][256];   // + cnt0, bad
    const uint32_t vbase = blockIdx.x * 256;
    {
        const uint32_t sv = vbase + threadIdx.x;
        if (sv < V) {
            const float freq = freq_p.get(sv);
            K k;
            OSC::setup(k, srf, freq, color_p.get(sv));
            const uint32_t *kw = reinterpret_cast<const uint32_t *>(&k);
#pragma unroll
            for (int j = 0; j < KW; j++) sk[j][threadIdx.x] = kw[j];
            sk[KW][threadIdx.x] = cnt_in[sv];
            sk[KW + 1][threadIdx.x] = (freq < 0 || freq > sr8) ? 1u : 0u;   // PulseOsc.zig:82-84, TriSawOsc.zig:84-86
        }
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t v = vbase + lane * 4;
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= V) return;                                   // V % 4 == 0 on this path
    const uint32_t c0 = start + chunk * fc;
    const uint32_t c1 = min(c0 + fc, end);
    K k[4];
    uint32_t cnt[4], cnt0[4];
    bool bad[4];
    {
        uint4 w[KW + 2];
#pragma unroll
        for (int j = 0; j < KW + 2; j++) w[j] = *reinterpret_cast<const uint4 *>(&sk[j][lane * 4]);
#pragma unroll
        for (int j = 0; j < KW; j++) {
            reinterpret_cast<uint32_t *>(&k[0])[j] = w[j].x; reinterpret_cast<uint32_t *>(&k[1])[j] = w[j].y;
            reinterpret_cast<uint32_t *>(&k[2])[j] = w[j].z; reinterpret_cast<uint32_t *>(&k[3])[j] = w[j].w;
        }
        cnt0[0] = w[KW].x; cnt0[1] = w[KW].y; cnt0[2] = w[KW].z; cnt0[3] = w[KW].w;
        bad[0] = w[KW + 1].x != 0; bad[1] = w[KW + 1].y != 0; bad[2] = w[KW + 1].z != 0; bad[3] = w[KW + 1].w != 0;
    }
    typename OSC::R roll[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { cnt[j] = cnt0[j] + (c0 - start) * k[j].ifreq; roll[j] = OSC::roll_init(k[j], cnt[j]); }
    if (chunk == 0) {
        uint4 o;
        o.x = bad[0] ? cnt0[0] : cnt0[0] + (end - start) * k[0].ifreq;
        o.y = bad[1] ? cnt0[1] : cnt0[1] + (end - start) * k[1].ifreq;
        o.z = bad[2] ? cnt0[2] : cnt0[2] + (end - start) * k[2].ifreq;
        o.w = bad[3] ? cnt0[3] : cnt0[3] + (end - start) * k[3].ifreq;
        // write-through like the image stores: a plain store would leave the line dirty in L2 and put its
        // write-back on the kernel boundary (measured: 0.14 us of a 4.6 us launch)
        const zh_rsrc_t crs = make_rsrc(cnt_out, V * 4u);
        store4<SM>(reinterpret_cast<float *>(cnt_out + v), crs, v * 4u, __builtin_bit_cast(zv4f, o));
    }
    if (c0 >= end) return;
    float *o = out.at(c0, v);
    const size_t os = out.stride;
    // descriptor over this wave's rows [c0, c1) of the image (wave-uniform base)
    const uint32_t wchunk = __builtin_amdgcn_readfirstlane(chunk);
    const uint32_t wc0 = start + wchunk * fc;
    const zh_rsrc_t rsrc = make_rsrc(out.p + (size_t)wc0 * out.stride, (uint32_t)((size_t)fc * out.stride * 4));
    uint32_t boff = lane * 16 + (uint32_t)((size_t)vbase * 4);
    // Common case, decided per wave: no silent voice among the wave's 256.  In ZERO_FIRST mode the
    // stored value is then `0.0f + val`, which equals `val` bit for bit: every arm of sample() ends in
    // `x + gain`, `x - gain` or `gain + x` with gain = 0.7, and an IEEE sum is -0.0 only if both addends
    // are -0.0, so val is never -0.0 (the one input 0.0f + x changes); NaNs pass through unchanged.
    if (!__any(bad[0] || bad[1] || bad[2] || bad[3])) {
#pragma unroll 2
        for (uint32_t i = c0; i < c1; i++, o += os, boff += (uint32_t)os * 4) {
            zv4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!ZF) acc = *reinterpret_cast<const zv4f *>(o);
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                val[j] = OSC::sample_roll(k[j], cnt[j], roll[j]);
                cnt[j] += k[j].ifreq;
            }
            if (ZF) { acc.x = val[0]; acc.y = val[1]; acc.z = val[2]; acc.w = val[3]; }
            else { acc.x += val[0]; acc.y += val[1]; acc.z += val[2]; acc.w += val[3]; }
            store4<SM>(o, rsrc, boff, acc);
        }
        return;
    }
#pragma unroll 2
    for (uint32_t i = c0; i < c1; i++, o += os, boff += (uint32_t)os * 4) {
        zv4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!ZF) acc = *reinterpret_cast<const zv4f *>(o);
        float val[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            val[j] = OSC::sample_roll(k[j], cnt[j], roll[j]);
            cnt[j] += k[j].ifreq;
        }
        // a silent voice (bad freq) paints nothing: out unchanged (ADD) / zero (ZERO_FIRST)
        acc.x = bad[0] ? acc.x : acc.x + val[0];
        acc.y = bad[1] ? acc.y : acc.y + val[1];
        acc.z = bad[2] ? acc.z : acc.z + val[2];
        acc.w = bad[3] ? acc.w : acc.w + val[3];
        store4<SM>(o, rsrc, boff, acc);
    }
}

template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_pulseosc_ctrl(uint32_t *__restrict__ cnt_io, uint32_t V, Img out,
                                                             uint32_t start, uint32_t end, float srf, float sr8,
                                                             CImg freq_b, F32P color_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    PulseOscLane o;
    o.cnt = cnt_io[v];
    o.srf = srf; o.sr8 = sr8;                                         // host-computed (same IEEE divides)
    pulse_setup_color(o.k, color_p.get(v));
    const float *ins[1] = {freq_b.p};
    const size_t istr[1] = {freq_b.stride};
    frame_loop<8, ZF, 1>(out.p, v, out.stride, ins, istr, start, end,
                         [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA { return o.frame_ctrl(x[0], val); });
    cnt_io[v] = o.cnt;
}

// ------------------------------------------------------------------ TriSawOsc
// the formulas live in voices.cuh (trisaw_setup / trisaw_sample / trisaw_naive)
struct TriSawOscP {
    using K = TriSawK;
    static __device__ __forceinline__ void setup(K &k, float srf, float freq, float color) { trisaw_setup(k, srf, freq, color); }
    static __device__ __forceinline__ float sample(const K &k, uint32_t cnt) { return trisaw_sample(k, cnt); }
    static constexpr bool kShortChunks = false;
    using R = int;                                          // nothing carried
    static __device__ __forceinline__ R roll_init(const K &, uint32_t) { return 0; }
    static __device__ __forceinline__ float sample_roll(const K &k, uint32_t cnt, R &) { return trisaw_sample(k, cnt); }
};

// TriSawOsc.zig:120-156: naive saw / triangle from an f32 phase; ignores cnt
template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_trisawosc_ctrl(float *__restrict__ t_io, uint32_t V, Img out,
                                                              uint32_t start, uint32_t end, float sample_rate,
                                                              CImg freq_b, F32P color_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    TriSawOscLane o;
    o.t = t_io[v];
    o.begin_ctrl(sample_rate, color_p.get(v));
    const float *ins[1] = {freq_b.p};
    const size_t istr[1] = {freq_b.stride};
    frame_loop<8, ZF, 1>(out.p, v, out.stride, ins, istr, start, end, [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA {
        val = o.frame_ctrl(x[0]);
        return true;
    });
    o.end_ctrl();
    t_io[v] = o.t;
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// Frames per lane for the chunked kernels.  PulseOsc, measured (tools/sweep_osc_fc.sh, 1024-frame images): 4 frames per
// lane is the best or within 5 % of the best at every voice count from 4,096 to 1 Mi and 11-15 % better than 64 between
// 65,536 and 524,288 voices (short waves interleave their ALU and store phases better, and the blocks sweep the image in
// row order); the per-voice setup is shared through LDS by the four chunks of a block, so short chunks cost little.
// TriSawOsc's setup (three divides) and sample (~45 instructions) are heavier: it keeps longer chunks -- enough of them
// for ~4096 waves, at least 8 frames (131,072 voices: 100 us against 131 us with 4 frames per lane).
// ZH_OSC_FC / ZH_OSC_SCALAR override for experiments.
static uint32_t osc_frames_per_lane(bool short_chunks, uint32_t lanes, uint32_t nframes) {
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("ZH_OSC_FC"); forced = e ? atoi(e) : 0; }
    if (forced > 0) return (uint32_t)forced;
    if (short_chunks) return 4u;
    const uint64_t groups = (lanes + 63) / 64;
    const uint64_t fc = (groups * nframes) / 4096u;
    uint32_t p = 8;
    while (p * 2 <= fc && p < 64) p *= 2;
    return p;
}
static bool osc_force_scalar() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("ZH_OSC_SCALAR"); v = e ? atoi(e) : 0; }
    return v != 0;
}

// Launch the chunked constant-frequency kernel of oscillator OSC (4 voices per lane with 16-byte
// write-through stores when the image and params allow it, one voice per lane otherwise).
template <class OSC>
static void launch_osc_const(zh_ctx *ctx, uint32_t n, const uint32_t *ci, uint32_t *co, const zh_buf &outb, uint32_t start,
                             uint32_t end, float sample_rate, const zh_f32 &freq, const zh_f32 &color, bool zf) {
    hipStream_t st = ctx->stream;
    Img out = mk_img(outb);
    const float srf = 4294967296.0f / sample_rate;        // SRfcobasefrq, PulseOsc.zig:87 / TriSawOsc.zig:88
    const float sr8 = sample_rate / 8.0f;                 // PulseOsc.zig:82 / TriSawOsc.zig:84
    const F32P fq = mk_f32(freq), col = mk_f32(color);
    const bool vec = n % 4 == 0 && outb.stride % 4 == 0 && aligned16(outb.ptr) && (!fq.pv || aligned16(fq.pv)) &&
                     (!col.pv || aligned16(col.pv)) && !osc_force_scalar();
    const uint32_t lanes = vec ? n / 4 : n;
    const uint32_t fc = osc_frames_per_lane(OSC::kShortChunks, lanes, end - start);
    const uint32_t chunks = (end - start + fc - 1) / fc;
    dim3 grid((lanes + 63) / 64, (chunks + 3) / 4);
    if (vec) {
#define ZH_LAUNCH_O4(ZF, SM) hipLaunchKernelGGL((k_osc_const4<OSC, ZF, SM>), grid, dim3(256), 0, st, ci, co, n, out, start, end, fc, srf, sr8, fq, col)
        const int sm = ((size_t)fc * outb.stride * 4 >> 32) ? ST_PLAIN : zh_store_mode();
        if (zf) { if (sm == ST_PLAIN) ZH_LAUNCH_O4(true, ST_PLAIN); else if (sm == ST_NT) ZH_LAUNCH_O4(true, ST_NT); else if (sm == ST_SC1) ZH_LAUNCH_O4(true, ST_SC1); else ZH_LAUNCH_O4(true, ST_SC0SC1); }
        else    { if (sm == ST_PLAIN) ZH_LAUNCH_O4(false, ST_PLAIN); else if (sm == ST_NT) ZH_LAUNCH_O4(false, ST_NT); else if (sm == ST_SC1) ZH_LAUNCH_O4(false, ST_SC1); else ZH_LAUNCH_O4(false, ST_SC0SC1); }
#undef ZH_LAUNCH_O4
    } else {
        if (zf) hipLaunchKernelGGL((k_osc_const<OSC, true>), grid, dim3(256), 0, st, ci, co, n, out, start, end, fc, srf, sr8, fq, col);
        else hipLaunchKernelGGL((k_osc_const<OSC, false>), grid, dim3(256), 0, st, ci, co, n, out, start, end, fc, srf, sr8, fq, col);
    }
}

template <class M> static int osc_common_check(M *m, uint32_t start, uint32_t end, const zh_buf *outputs,
                                               float sample_rate, const zh_cob &freq) {
    (void)sample_rate;
    if (!m || !outputs || end < start) return ZH_ERR_INVALID;
    if (!buf_covers(outputs[0], m->n, end)) return ZH_ERR_INVALID;
    if (!cob_ok(freq, m->n, end)) return ZH_ERR_INVALID;
    return ZH_OK;
}

extern "C" {

// -------- PulseOsc
int zh_pulseosc_create(zh_ctx *ctx, uint32_t n, zh_pulseosc **out) {
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_pulseosc *m = new (std::nothrow) zh_pulseosc();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr;
    int rc = dev_alloc(&m->cnt[0], n);
    if (!rc) rc = dev_alloc(&m->cnt[1], n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[0], 0, n * 4, ctx->stream);       // init(): cnt = 0 (:38-42)
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[1], 0, n * 4, ctx->stream);
    if (rc) { hipFree(m->cnt[0]); hipFree(m->cnt[1]); delete m; return rc; }
    *out = m;
    return ZH_OK;
}
int zh_pulseosc_destroy(zh_pulseosc *m) {
    if (!m) return ZH_ERR_INVALID;
    hipStreamSynchronize(m->ctx->stream);
    hipFree(m->cnt[0]); hipFree(m->cnt[1]);
    delete m;
    return ZH_OK;
}
int zh_pulseosc_get_state(zh_pulseosc *m, zh_pulseosc_state *host) {
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_download(m->ctx, host, m->cnt[m->cur], (size_t)m->n * 4);
}
int zh_pulseosc_set_state(zh_pulseosc *m, const zh_pulseosc_state *host) {
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_upload(m->ctx, m->cnt[m->cur], host, (size_t)m->n * 4);
}
int zh_pulseosc_paint(zh_pulseosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                      zh_bool note_id_changed, const zh_pulseosc_params *p, uint32_t flags) {
    (void)temps; (void)note_id_changed;                                             // PulseOsc.zig:52-53
    if (!p) return ZH_ERR_INVALID;
    int rc = osc_common_check(m, start, end, outputs, p->sample_rate, p->freq);
    if (rc) return rc;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    Img out = mk_img(outputs[0]);
    if (p->freq.tag == ZH_COB_CONSTANT) {
        launch_osc_const<PulseOscP>(m->ctx, m->n, m->cnt[m->cur], m->cnt[m->cur ^ 1], outputs[0], start, end, p->sample_rate,
                                    p->freq.constant, p->color, zf);
        m->cur ^= 1;
    } else {
        const float srf = 4294967296.0f / p->sample_rate;    // SRfcobasefrq, PulseOsc.zig:122
        const float sr8 = p->sample_rate / 8.0f;              // :134
        uint32_t *c = m->cnt[m->cur];
        const F32P col = mk_f32(p->color);
        if (zf) hipLaunchKernelGGL(k_pulseosc_ctrl<true>, seq_grid(m->n), dim3(kSeqBlock), 0, st, c, m->n, out, start, end, srf, sr8, mk_cimg(p->freq.buffer), col);
        else hipLaunchKernelGGL(k_pulseosc_ctrl<false>, seq_grid(m->n), dim3(kSeqBlock), 0, st, c, m->n, out, start, end, srf, sr8, mk_cimg(p->freq.buffer), col);
    }
    return zh_launch_status();
}

// -------- TriSawOsc
int zh_trisawosc_create(zh_ctx *ctx, uint32_t n, zh_trisawosc **out) {
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_trisawosc *m = new (std::nothrow) zh_trisawosc();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->t = nullptr;
    int rc = dev_alloc(&m->cnt[0], n);
    if (!rc) rc = dev_alloc(&m->cnt[1], n);
    if (!rc) rc = dev_alloc(&m->t, n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[0], 0, n * 4, ctx->stream);       // init() :39-44
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[1], 0, n * 4, ctx->stream);
    if (!rc && n) rc = (int)hipMemsetAsync(m->t, 0, n * 4, ctx->stream);
    if (rc) { hipFree(m->cnt[0]); hipFree(m->cnt[1]); hipFree(m->t); delete m; return rc; }
    *out = m;
    return ZH_OK;
}
int zh_trisawosc_destroy(zh_trisawosc *m) {
    if (!m) return ZH_ERR_INVALID;
    hipStreamSynchronize(m->ctx->stream);
    hipFree(m->cnt[0]); hipFree(m->cnt[1]); hipFree(m->t);
    delete m;
    return ZH_OK;
}
int zh_trisawosc_get_state(zh_trisawosc *m, zh_trisawosc_state *host) {
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> c(m->n);
    std::vector<float> t(m->n);
    int rc = zh_download(m->ctx, c.data(), m->cnt[m->cur], (size_t)m->n * 4);
    if (!rc) rc = zh_download(m->ctx, t.data(), m->t, (size_t)m->n * 4);
    if (rc) return rc;
    for (uint32_t i = 0; i < m->n; i++) { host[i].cnt = c[i]; host[i].t = t[i]; }
    return ZH_OK;
}
int zh_trisawosc_set_state(zh_trisawosc *m, const zh_trisawosc_state *host) {
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> c(m->n);
    std::vector<float> t(m->n);
    for (uint32_t i = 0; i < m->n; i++) { c[i] = host[i].cnt; t[i] = host[i].t; }
    int rc = zh_upload(m->ctx, m->cnt[m->cur], c.data(), (size_t)m->n * 4);
    if (!rc) rc = zh_upload(m->ctx, m->t, t.data(), (size_t)m->n * 4);
    return rc;
}
int zh_trisawosc_paint(zh_trisawosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                       zh_bool note_id_changed, const zh_trisawosc_params *p, uint32_t flags) {
    (void)temps; (void)note_id_changed;                                             // TriSawOsc.zig:54-55
    if (!p) return ZH_ERR_INVALID;
    int rc = osc_common_check(m, start, end, outputs, p->sample_rate, p->freq);
    if (rc) return rc;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    Img out = mk_img(outputs[0]);
    if (p->freq.tag == ZH_COB_CONSTANT) {
        launch_osc_const<TriSawOscP>(m->ctx, m->n, m->cnt[m->cur], m->cnt[m->cur ^ 1], outputs[0], start, end, p->sample_rate,
                                     p->freq.constant, p->color, zf);
        m->cur ^= 1;
    } else {
        if (zf) hipLaunchKernelGGL(k_trisawosc_ctrl<true>, seq_grid(m->n), dim3(kSeqBlock), 0, st, m->t, m->n, out, start, end, p->sample_rate, mk_cimg(p->freq.buffer), mk_f32(p->color));
        else hipLaunchKernelGGL(k_trisawosc_ctrl<false>, seq_grid(m->n), dim3(kSeqBlock), 0, st, m->t, m->n, out, start, end, p->sample_rate, mk_cimg(p->freq.buffer), mk_f32(p->color));
    }
    return zh_launch_status();
}

}  // extern "C"
