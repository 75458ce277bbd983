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
// lane-per-voice loop (seq.hip.h).
#include "common.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "seq.hip.h"
#include "voices.hip.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>

// Per-voice constants of the constant-frequency path, kept from one paint to the next ([KW + 1][n]: the policy's K
// words, then the bad-frequency flag).  Every constant-frequency paint that computes them also stores them; a paint
// flagged ZH_PAINT_PARAMS_UNCHANGED whose scalar params / array pointers match the paint that stored them loads them
// (eight 16-byte loads per lane) instead of repeating the per-voice divide, conversions and the LDS hand-off.
struct OscTable {
    uint32_t *words;             // [KW + 1][n]
    bool valid;
    bool pinned;                 // a captured graph holds a table-form paint: the table's contents are frozen from then on
    float sample_rate;
    zh_f32 freq, color;
};
static bool table_matches(const OscTable &t, float sample_rate, const zh_f32 &freq, const zh_f32 &color) {
    auto same = [](const zh_f32 &a, const zh_f32 &b) {
        return a.per_voice == b.per_voice && (a.per_voice || __builtin_bit_cast(uint32_t, a.value) == __builtin_bit_cast(uint32_t, b.value));
    };
    return t.valid && __builtin_bit_cast(uint32_t, t.sample_rate) == __builtin_bit_cast(uint32_t, sample_rate) &&
           same(t.freq, freq) && same(t.color, color);
}

struct zh_pulseosc : zh_flipper {
    OscTable tab;
    uint32_t *part;           // [64][n] per-range phase advances of a controlled-frequency span (k_pulseosc_ctrl_sums); n <= kPulsePartMaxVoices
};
constexpr uint32_t kPulsePartMaxVoices = 40960;

struct zh_trisawosc : zh_flipper {
    float *t;
    float *t_next;            // k_trisawosc_ctrl's frame ranges: the phase after the span, moved into `t` by k_commit_f32
    zh_buf quot;              // ... and their freq / sample_rate image (k_div_image), so that the replay is a load and an add
    OscTable tab;
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
    static constexpr bool kSawShortChunks = false;
    using R = PulseRoll;
    static __device__ __forceinline__ R roll_init(const K &k, uint32_t cnt) { return pulse_roll_init(k, cnt); }
    static __device__ __forceinline__ float sample_roll(const K &k, uint32_t cnt, R &r) { return pulse_sample_roll(k, cnt, r); }
    static constexpr bool kHasFast = false;                 // (no second form of the frame loops)
    template <bool FAST> static __device__ __forceinline__ float sample_f(const K &k, uint32_t cnt, R &r) { return pulse_sample_roll(k, cnt, r); }
    static __device__ __forceinline__ bool fast(const R (&)[4]) { return false; }
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
// grid: x = 256-voice groups, y = groups of 4 frame chunks, z = buffer of a batch; each of the 4 waves of a block
// renders a different chunk of the same 256 voices.
// A batch (zh_*_paint_batch) = nb consecutive paint calls with the same params, each over [start, end) of its own
// image: the phase counter of frame i of buffer b is exactly cnt0 + (b * (end - start) + (i - start)) * ifreq, so
// the buffers are independent too and share one launch's ramp and tail.
constexpr int kOscMaxBatch = 32;
struct OscArgs {
    const uint32_t *cnt_in;
    uint32_t *cnt_out;
    uint32_t *tab;               // [KW + 1][V] or nullptr
    uint32_t V, start, end, fc, stride, nb, prio;
    float srf, sr8;
    F32P freq, color;
    float *img[kOscMaxBatch];
};

// TAB = the per-voice constants come from the module's table instead of being computed (OscTable).
// FC4 = every wave renders exactly four frames (fc == 4 and the span is a multiple of 4): the frame loop unrolls completely.
// BATCH only names the instantiation a multi-buffer launch uses (same code): a profiler's per-kernel statistics then keep
// one-buffer launches and batches apart.
template <class OSC, bool ZF, int SM, bool TAB, bool FC4 = false, bool BATCH = false>
__global__ void __launch_bounds__(256) k_osc_const4(const OscArgs a) {
    using K = typename OSC::K;
    constexpr int KW = sizeof(K) / 4;                     // dwords of per-voice constants
    const uint32_t vbase = blockIdx.x * 256;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t v = vbase + lane * 4;
    const uint32_t V = a.V, start = a.start, end = a.end, fc = a.fc;
    // The block's four waves reach their first store one after the other instead of all at once (they share a SIMD's
    // issue slots when the grid fills the chip four waves deep): the store stream starts earlier.
    if (a.prio) {
        if (wave == 0) __builtin_amdgcn_s_setprio(3);
        else if (wave == 1) __builtin_amdgcn_s_setprio(2);
        else if (wave == 2) __builtin_amdgcn_s_setprio(1);
    }
    K k[4];
    uint32_t cnt[4], cnt0[4];
    bool bad[4];
    if constexpr (TAB) {
        if (v >= V) return;                               // V % 4 == 0 on this path
        uint4 w[KW + 2];
#pragma unroll
        for (int j = 0; j < KW + 1; j++) w[j] = *reinterpret_cast<const uint4 *>(a.tab + (size_t)j * V + v);
        w[KW + 1] = *reinterpret_cast<const uint4 *>(a.cnt_in + v);
#pragma unroll
        for (int j = 0; j < KW; j++) {
            reinterpret_cast<uint32_t *>(&k[0])[j] = w[j].x; reinterpret_cast<uint32_t *>(&k[1])[j] = w[j].y;
            reinterpret_cast<uint32_t *>(&k[2])[j] = w[j].z; reinterpret_cast<uint32_t *>(&k[3])[j] = w[j].w;
        }
        bad[0] = w[KW].x != 0; bad[1] = w[KW].y != 0; bad[2] = w[KW].z != 0; bad[3] = w[KW].w != 0;
        cnt0[0] = w[KW + 1].x; cnt0[1] = w[KW + 1].y; cnt0[2] = w[KW + 1].z; cnt0[3] = w[KW + 1].w;
    } else {
        // The 4 waves of a block render 4 different frame chunks of the SAME 256 voices, so the per-voice
        // setup (a divide and some conversions) is done once per block: thread t sets up voice base + t and
        // parks the constants in LDS, then every lane fetches its 4 voices' constants with 16-byte reads.
        __shared__ __attribute__((aligned(16))) uint32_t sk[KW + 2][256];   // + bad, cnt0
        {
            const uint32_t sv = vbase + threadIdx.x;
            if (sv < V) {
                const float freq = a.freq.get(sv);
                K ks;
                OSC::setup(ks, a.srf, freq, a.color.get(sv));
                const uint32_t *kw = reinterpret_cast<const uint32_t *>(&ks);
                const uint32_t isbad = (freq < 0 || freq > a.sr8) ? 1u : 0u;   // PulseOsc.zig:82-84, TriSawOsc.zig:84-86
#pragma unroll
                for (int j = 0; j < KW; j++) sk[j][threadIdx.x] = kw[j];
                sk[KW][threadIdx.x] = isbad;
                sk[KW + 1][threadIdx.x] = a.cnt_in[sv];
                if (a.tab && blockIdx.y == 0 && blockIdx.z == 0) {          // keep them for ZH_PAINT_PARAMS_UNCHANGED paints
#pragma unroll
                    for (int j = 0; j < KW; j++) a.tab[(size_t)j * V + sv] = kw[j];
                    a.tab[(size_t)KW * V + sv] = isbad;
                }
            }
        }
        __syncthreads();
        if (v >= V) return;                               // V % 4 == 0 on this path
        uint4 w[KW + 2];
#pragma unroll
        for (int j = 0; j < KW + 2; j++) w[j] = *reinterpret_cast<const uint4 *>(&sk[j][lane * 4]);
#pragma unroll
        for (int j = 0; j < KW; j++) {
            reinterpret_cast<uint32_t *>(&k[0])[j] = w[j].x; reinterpret_cast<uint32_t *>(&k[1])[j] = w[j].y;
            reinterpret_cast<uint32_t *>(&k[2])[j] = w[j].z; reinterpret_cast<uint32_t *>(&k[3])[j] = w[j].w;
        }
        bad[0] = w[KW].x != 0; bad[1] = w[KW].y != 0; bad[2] = w[KW].z != 0; bad[3] = w[KW].w != 0;
        cnt0[0] = w[KW + 1].x; cnt0[1] = w[KW + 1].y; cnt0[2] = w[KW + 1].z; cnt0[3] = w[KW + 1].w;
    }
    // (wave-uniform: said so explicitly, or the frame loops below become per-lane exec-mask loops that cannot be unrolled)
    const uint32_t chunk = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wave);
    const uint32_t nfr = end - start;
    const uint32_t c0 = start + chunk * fc;
    const uint32_t c1 = min(c0 + fc, end);
    const uint32_t fbase = blockIdx.z * nfr + (c0 - start);               // frames painted before this chunk, batch-wide
    typename OSC::R roll[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { cnt[j] = cnt0[j] + fbase * k[j].ifreq; roll[j] = OSC::roll_init(k[j], cnt[j]); }
    if (chunk == 0 && blockIdx.z == 0) {
        const uint32_t total = a.nb * nfr;
        uint4 o;
        o.x = bad[0] ? cnt0[0] : cnt0[0] + total * k[0].ifreq;
        o.y = bad[1] ? cnt0[1] : cnt0[1] + total * k[1].ifreq;
        o.z = bad[2] ? cnt0[2] : cnt0[2] + total * k[2].ifreq;
        o.w = bad[3] ? cnt0[3] : cnt0[3] + total * k[3].ifreq;
        // write-through like the image stores: a plain store would leave the line dirty in L2 and put its
        // write-back on the kernel boundary (measured: 0.14 us of a 4.6 us launch)
        const zh_rsrc_t crs = make_rsrc(a.cnt_out, V * 4u);
        store4<SM>(reinterpret_cast<float *>(a.cnt_out + v), crs, v * 4u, __builtin_bit_cast(zv4f, o));
    }
    if (c0 >= end) return;
    float *const img = a.img[blockIdx.z];
    const size_t os = a.stride;
    float *o = img + (size_t)c0 * os + v;
    // descriptor over this wave's rows [c0, c1) of the image (wave-uniform base)
    const uint32_t wchunk = __builtin_amdgcn_readfirstlane(chunk);
    const uint32_t wc0 = start + wchunk * fc;
    const zh_rsrc_t rsrc = make_rsrc(img + (size_t)wc0 * os, (uint32_t)((size_t)fc * os * 4));
    uint32_t boff = lane * 16 + (uint32_t)((size_t)vbase * 4);
    // (the frame loops once per form of the sample: OSC::fast -- every voice of the wave a sawtooth, TriSawOsc -- picks the light one)
    auto frames = [&](auto fast_tag) ZH_INLINE_LAMBDA {
        constexpr bool FAST = decltype(fast_tag)::value;
        // Common case, decided per wave: no silent voice among the wave's 256.  In ZERO_FIRST mode the
        // stored value is then `0.0f + val`, which equals `val` bit for bit: every arm of sample() ends in
        // `x + gain`, `x - gain` or `gain + x` with gain = 0.7, and an IEEE sum is -0.0 only if both addends
        // are -0.0, so val is never -0.0 (the one input 0.0f + x changes); NaNs pass through unchanged.
        if (!__any(bad[0] || bad[1] || bad[2] || bad[3])) {
            if constexpr (FC4) {
#pragma unroll
                for (uint32_t i = 0; i < 4; i++, o += os, boff += (uint32_t)os * 4) {
                    zv4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (!ZF) acc = *reinterpret_cast<const zv4f *>(o);
                    float val[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        val[j] = OSC::template sample_f<FAST>(k[j], cnt[j], roll[j]);
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
                    val[j] = OSC::template sample_f<FAST>(k[j], cnt[j], roll[j]);
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
                val[j] = OSC::template sample_f<FAST>(k[j], cnt[j], roll[j]);
                cnt[j] += k[j].ifreq;
            }
            // a silent voice (bad freq) paints nothing: out unchanged (ADD) / zero (ZERO_FIRST)
            acc.x = bad[0] ? acc.x : acc.x + val[0];
            acc.y = bad[1] ? acc.y : acc.y + val[1];
            acc.z = bad[2] ? acc.z : acc.z + val[2];
            acc.w = bad[3] ? acc.w : acc.w + val[3];
            store4<SM>(o, rsrc, boff, acc);
        }
    };
    if constexpr (OSC::kHasFast) {
        if (OSC::fast(roll)) { frames(std::true_type{}); return; }
    }
    frames(std::false_type{});
}

// Controlled frequency (PulseOsc.zig:116-157).  One kernel, two launch shapes: sequential (grid.y = 1, ch = the span,
// cnt_in == cnt_out), or -- few voices -- the span as grid.y frame ranges of `ch` frames at once, one wave per
// (64 voices, range): the phase counter that reaches frame f0 is the start counter plus the `ifreq` of every earlier frame
// whose frequency is in range (:134-136; u32 wrap-around adds, exact in any order), so a range first sums those (a multiply,
// a conversion and an add per earlier frame against the divide and ~25 instructions of a painted sample), then paints
// its frames; the range that ends the span publishes the counter into the other half of the double buffer.
// The per-range sums of that replay, each range over its OWN frames only (u32 adds are exact in any order, so the advance up
// to a range's first frame is the sum of the ranges before it): part[range][voice].  With them k_pulseosc_ctrl starts a range
// from <= 63 loads instead of replaying up to a whole span.
__global__ void __launch_bounds__(kSeqBlock) k_pulseosc_ctrl_sums(uint32_t *__restrict__ part, uint32_t V, uint32_t start, uint32_t end, uint32_t ch,
                                                                  float srf, float sr8, CImg freq_b) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    const float *fp = freq_b.p + (size_t)f0 * freq_b.stride + v;
    uint32_t sum = 0;
    auto add = [&](float f) ZH_INLINE_LAMBDA { sum += (f < 0 || f > sr8) ? 0u : zf32_to_u32(srf * f); };   // PulseOsc.zig:134-136
    uint32_t i = f0;
    for (; i + 32 <= f1; i += 32, fp += 32 * freq_b.stride) {
        float x[32];
#pragma unroll
        for (int k = 0; k < 32; k++) x[k] = fp[(size_t)k * freq_b.stride];
#pragma unroll
        for (int k = 0; k < 32; k++) add(x[k]);
    }
    for (; i < f1; i++, fp += freq_b.stride) add(*fp);
    part[(size_t)blockIdx.y * V + v] = sum;
}

template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_pulseosc_ctrl(const uint32_t *__restrict__ cnt_in, uint32_t *__restrict__ cnt_out, uint32_t V,
                                                             Img out, uint32_t start, uint32_t end, uint32_t ch, float srf, float sr8,
                                                             CImg freq_b, F32P color_p, const uint32_t *__restrict__ part) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    PulseOscLane o;
    o.cnt = cnt_in[v];
    o.srf = srf; o.sr8 = sr8;                                         // host-computed (same IEEE divides)
    pulse_setup_color(o.k, color_p.get(v));
    if (part) {
        for (uint32_t q = 0; q < blockIdx.y; q++) o.cnt += part[(size_t)q * V + v];
    } else {
        const float *fp = freq_b.p + (size_t)start * freq_b.stride + v;
        auto skip = [&](float f) ZH_INLINE_LAMBDA {
            const uint32_t ifreq = zf32_to_u32(srf * f);              // pulse_setup_freq's k.ifreq (dsp.hip.h), PulseOsc.zig:136
            o.cnt += (f < 0 || f > sr8) ? 0u : ifreq;                 // :134-135: a sample out of range neither paints nor advances
        };
        uint32_t i = start;
        for (; i + 32 <= f0; i += 32, fp += 32 * freq_b.stride) {       // 32 rows in flight: the replay is load-latency-bound otherwise
            float x[32];
#pragma unroll
            for (int k = 0; k < 32; k++) x[k] = fp[(size_t)k * freq_b.stride];
#pragma unroll
            for (int k = 0; k < 32; k++) skip(x[k]);
        }
        for (; i + 8 <= f0; i += 8, fp += 8 * freq_b.stride) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = fp[(size_t)k * freq_b.stride];
#pragma unroll
            for (int k = 0; k < 8; k++) skip(x[k]);
        }
        for (; i < f0; i++, fp += freq_b.stride) skip(*fp);
    }
    const float *ins[1] = {freq_b.p};
    const size_t istr[1] = {freq_b.stride};
    frame_loop<8, ZF, 1>(out.p, v, out.stride, ins, istr, f0, f1,
                         [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA { return o.frame_ctrl(x[0], val); });
    if (f1 == end) cnt_out[v] = o.cnt;
}

// ------------------------------------------------------------------ TriSawOsc
// the formulas live in voices.hip.h (trisaw_setup / trisaw_sample / trisaw_naive)
struct TriSawOscP {
    using K = TriSawK;
    static __device__ __forceinline__ void setup(K &k, float srf, float freq, float color) { trisaw_setup(k, srf, freq, color); }
    static __device__ __forceinline__ float sample(const K &k, uint32_t cnt) { return trisaw_sample(k, cnt); }
    static constexpr bool kShortChunks = false;
    static constexpr bool kSawShortChunks = true;           // ... unless the paint is a sawtooth for every voice (launch_osc_const)
    using R = bool;                                         // wave-uniform: every voice a sawtooth (voices.hip.h trisaw_sample_saw)
    static __device__ __forceinline__ R roll_init(const K &k, uint32_t) { return trisaw_all_saw(k); }
    static __device__ __forceinline__ float sample_roll(const K &k, uint32_t cnt, R &saw) { return saw ? trisaw_sample_saw(k, cnt) : trisaw_sample(k, cnt); }
    // k_osc_const4: the four voices of a lane decide together, ONE wave-uniform condition, and the frame loops exist once per form
    static constexpr bool kHasFast = true;
    template <bool FAST> static __device__ __forceinline__ float sample_f(const K &k, uint32_t cnt, R &) { return FAST ? trisaw_sample_saw(k, cnt) : trisaw_sample(k, cnt); }
    static __device__ __forceinline__ bool fast(const R (&r)[4]) { return r[0] && r[1] && r[2] && r[3]; }
};

// TriSawOsc.zig:120-156: naive saw / triangle from an f32 phase; ignores cnt
// One lane per voice walks the span (grid.y == 1, ch = the span, t_in == t_out), or -- few voices -- the span as grid.y
// frame ranges at once: the phase that reaches frame f0 is the start phase plus freq / sample_rate of every earlier frame,
// added in frame order (f32), so a range replays that (a load, the divide and an add per frame; the naive waveform is another
// ~20 instructions) and then paints its own frames.  The range that ends the span writes the phase to t_out (a buffer of
// its own; k_commit_f32 moves it into place once every range has read the start phase).
// QUOT: freq_b holds freq / sample_rate already (k_div_image painted it, frame-parallel): the divide -- eleven of the
// replay's thirteen instructions per frame -- is done once per sample instead of once per sample per later range.
__global__ void __launch_bounds__(256) k_div_image(Img q, CImg x, uint32_t V, uint32_t start, uint32_t end, float d) {
    const uint32_t v = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t piece = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const uint32_t f0 = start + piece * 32;
    if (f0 >= end) return;
    const uint32_t f1 = min(f0 + 32, end);
    float *qp = q.at(f0, v);
    const float *xp = x.at(f0, v);
    uint32_t f = f0;
    for (; f + 8 <= f1; f += 8, qp += 8 * (size_t)q.stride, xp += 8 * (size_t)x.stride) {   // 8 rows' loads ahead of their stores
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = xp[(size_t)k * x.stride];
#pragma unroll
        for (int k = 0; k < 8; k++) qp[(size_t)k * q.stride] = a[k] / d;
    }
    for (; f < f1; f++, qp += q.stride, xp += x.stride) *qp = *xp / d;
}
template <bool ZF, bool QUOT>
__global__ void __launch_bounds__(kSeqBlock) k_trisawosc_ctrl(const float *__restrict__ t_in, float *__restrict__ t_out, uint32_t V, Img out,
                                                              uint32_t start, uint32_t end, uint32_t ch, float sample_rate,
                                                              CImg freq_b, F32P color_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    TriSawOscLane o;
    o.t = t_in[v];
    o.begin_ctrl(sample_rate, color_p.get(v));
    replay_rows(freq_b.p + (size_t)start * freq_b.stride + v, freq_b.stride, f0 - start,
                [&](float x) ZH_INLINE_LAMBDA { o.t += QUOT ? x : x / sample_rate; });   // frame_ctrl's step (TriSawOsc.zig:151)
    const float *ins[1] = {freq_b.p};
    const size_t istr[1] = {freq_b.stride};
    frame_loop<8, ZF, 1>(out.p, v, out.stride, ins, istr, f0, f1, [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA {
        val = QUOT ? o.frame_ctrl_q(x[0]) : o.frame_ctrl(x[0]);
        return true;
    });
    if (f1 == end) {
        o.end_ctrl();
        t_out[v] = o.t;
    }
}
__global__ void __launch_bounds__(256) k_commit_f32(float *__restrict__ dst, const float *__restrict__ src, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// Frames per lane for the chunked kernels.  PulseOsc, measured (tools/sweep_osc_fc.sh, 1024-frame images): 4 frames per
// lane is the best or within 5 % of the best at every voice count from 4,096 to 1 Mi and 11-15 % better than 64 between
// 65,536 and 524,288 voices (short waves interleave their ALU and store phases better, and the blocks sweep the image in
// row order); the per-voice setup is shared through LDS by the four chunks of a block, so short chunks cost little.
// TriSawOsc's setup (three divides) and sample (~45 instructions) are heavier: it keeps longer chunks -- enough of them
// for ~4096 waves, at least 8 frames (131,072 voices: 100 us against 131 us with 4 frames per lane).
// (osc_fc in dispatch.hip overrides.  The A/B switches of rounds 1-4 -- scalar kernel forced, no wave priorities, no FC4 form, no
// constants table -- are gone with round 5: their forms were measured slower and are not selectable any more.)
static uint32_t osc_frames_per_lane(bool short_chunks, uint32_t lanes, uint32_t nframes) {
    const long forced = zh_form(ZF_OSC_FC);
    if (forced > 0) return (uint32_t)forced;
    // PulseOsc: 4 frames per lane; from 16,384 voices (4,096 lanes) 3 -- an odd number of rows between a lane's chunks spreads a
    // workgroup's rows over the HBM channels whatever the row stride, and with non-temporal stores (launch_osc_const) the paint
    // holds 0.83-0.86 of the HBM peak from 16,384 to 786,432 voices where 4 frames + write-through stores gave 0.63-0.83 by
    // voice count (profiles/r05/osc_large_voice_counts.txt).  Level at 4,096 voices: the headline size keeps 4.
    if (short_chunks) return lanes >= 4096u ? 3u : 4u;
    const uint64_t groups = (lanes + 63) / 64;
    const uint64_t fc = (groups * nframes) / 4096u;
    uint32_t p = 8;
    while (p * 2 <= fc && p < 64) p *= 2;
    return p;
}

template <class OSC> static bool osc_vec_ok(uint32_t n, const zh_buf *outs, uint32_t nb, const zh_f32 &freq, const zh_f32 &color) {
    bool vec = n % 4 == 0 && (!freq.per_voice || aligned16(freq.per_voice)) && (!color.per_voice || aligned16(color.per_voice));
    for (uint32_t b = 0; b < nb && vec; b++)
        vec = outs[b].stride % 4 == 0 && aligned16(outs[b].ptr) && outs[b].stride == outs[0].stride;
    return vec;
}

// Launch the chunked constant-frequency kernel of oscillator OSC for `nb` consecutive paints (one image each, same
// span and params): 4 voices per lane with 16-byte write-through stores when the images and params allow it (one
// launch for the whole batch), one voice per lane and one launch per image otherwise.  Flips m->cur once per paint.
template <class OSC, class M>
static void launch_osc_const(M *m, const zh_buf *outs, uint32_t nb, uint32_t start, uint32_t end, float sample_rate,
                             const zh_f32 &freq, const zh_f32 &color, uint32_t flags) {
    zh_ctx *ctx = m->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t n = m->n;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    const float srf = 4294967296.0f / sample_rate;        // SRfcobasefrq, PulseOsc.zig:87 / TriSawOsc.zig:88
    const float sr8 = sample_rate / 8.0f;                 // PulseOsc.zig:82 / TriSawOsc.zig:84
    const F32P fq = mk_f32(freq), col = mk_f32(color);
    const bool vec = osc_vec_ok<OSC>(n, outs, nb, freq, color);
    const uint32_t lanes = vec ? n / 4 : n;
    // a TriSawOsc paint whose color is one value <= 0 for all voices is a sawtooth everywhere (voices.hip.h trisaw_sample_saw: 9 instructions
    // a sample instead of 33): light enough for PulseOsc's short chunks and non-temporal stores
    const bool short_chunks = OSC::kShortChunks || (OSC::kSawShortChunks && !color.per_voice && color.value <= 0.0f);
    const uint32_t fc = osc_frames_per_lane(short_chunks, lanes, end - start);
    const uint32_t chunks = (end - start + fc - 1) / fc;
    if (vec) {
        // The table is read by every frame-chunk wave of a voice group, always on the same XCD (grid.x is a multiple of 8), so
        // it pays while an XCD's eighth of it stays in that XCD's 4 MiB L2: measured +3 ... +10 % up to 524,288 voices
        // (81 % of the HBM peak there), -20 % at 1,048,576 (3.7 MB per XCD: it thrashes) -- hence the size limit.
        constexpr size_t kw1 = sizeof(typename OSC::K) / 4 + 1;
        const bool table_fits = (size_t)n * kw1 * 4 <= ((size_t)16 << 20);
        const bool use_tab = (flags & ZH_PAINT_PARAMS_UNCHANGED) && m->tab.words && table_matches(m->tab, sample_rate, freq, color) && table_fits;
        // stores: ZH_STORE_MODE when set; else write-through (sc1), and non-temporal for the three-frame chunks of many voices up to
        // the table's size limit (above it -- 786,432 / 1,048,576 voices -- sc1 again: 0.84 against 0.72 of the HBM peak)
        const int sm = ((size_t)fc * outs[0].stride * 4 >> 32) ? ST_PLAIN
                       : zh_store_mode_env() >= 0 ? zh_store_mode_env() : (short_chunks && lanes >= 4096u && fc == 3 && table_fits) ? ST_NT : ST_SC1;
        const bool fc4 = fc == 4 && (end - start) % 4 == 0;
        // one launch of `a` (images and count filled in) over `cnt_b` buffers
        auto launch = [use_tab, fc4, zf, sm, lanes, chunks](OscArgs a, uint32_t cnt_b, hipStream_t st) {
            dim3 grid((lanes + 63) / 64, (chunks + 3) / 4, cnt_b);
#define ZH_LAUNCH_O4B(ZF, SM, B) do { if (use_tab && fc4) ZH_LAUNCH((k_osc_const4<OSC, ZF, SM, true, true, B>), grid, dim3(256), 0, st, a); \
                                      else if (use_tab) ZH_LAUNCH((k_osc_const4<OSC, ZF, SM, true, false, B>), grid, dim3(256), 0, st, a); \
                                      else if (fc4) ZH_LAUNCH((k_osc_const4<OSC, ZF, SM, false, true, B>), grid, dim3(256), 0, st, a); \
                                      else ZH_LAUNCH((k_osc_const4<OSC, ZF, SM, false, false, B>), grid, dim3(256), 0, st, a); } while (0)
#define ZH_LAUNCH_O4(ZF, SM) do { if (cnt_b > 1) { zh_tls_launch_detail = "batch"; ZH_LAUNCH_O4B(ZF, SM, true); } else ZH_LAUNCH_O4B(ZF, SM, false); } while (0)
            if (zf) { if (sm == ST_PLAIN) ZH_LAUNCH_O4(true, ST_PLAIN); else if (sm == ST_NT) ZH_LAUNCH_O4(true, ST_NT); else if (sm == ST_SC1) ZH_LAUNCH_O4(true, ST_SC1); else ZH_LAUNCH_O4(true, ST_SC0SC1); }
            else    { if (sm == ST_PLAIN) ZH_LAUNCH_O4(false, ST_PLAIN); else if (sm == ST_NT) ZH_LAUNCH_O4(false, ST_NT); else if (sm == ST_SC1) ZH_LAUNCH_O4(false, ST_SC1); else ZH_LAUNCH_O4(false, ST_SC0SC1); }
#undef ZH_LAUNCH_O4
#undef ZH_LAUNCH_O4B
        };
        // A recorded table-form paint reads the table at every replay, so once one has been captured nothing rewrites the
        // table any more (an eager setup-form paint with other params between replays used to: the replays then rendered with
        // THOSE constants): later unflagged paints compute their constants without storing them, and the host-side record is
        // invalidated so that later flagged paints take the setup form too.  A recorded setup-form paint never wrote it.
        if (use_tab && ctx->capturing) m->tab.pinned = true;
        const bool write_tab = !use_tab && !ctx->capturing && !m->tab.pinned;
        if (!use_tab && !ctx->capturing && m->tab.pinned) m->tab.valid = false;
        OscArgs a;
        a.tab = (use_tab || write_tab) ? m->tab.words : nullptr;
        a.V = n; a.start = start; a.end = end; a.fc = fc; a.stride = outs[0].stride; a.prio = 1u;
        a.srf = srf; a.sr8 = sr8; a.freq = fq; a.color = col;
        // A table-form paint of a ZH_CAPTURE_COALESCE capture is HELD BACK: consecutive such paints of one module over the same
        // span into images that do not overlap become ONE recorded launch of up to 32 buffers -- exactly what
        // zh_*_paint_batch launches: the phase of frame i of buffer b is cnt0 + (b * frames + i) * ifreq, the counters are read
        // from cnt[cur] and written to the other buffer once, and the module flips once per launch.  When the epoch ends
        // (zh_epoch_barrier: another call on the context, or the end of the capture) the last batch is recorded as TWO launches of
        // half the buffers each if one launch would leave the capture with an odd number of flips: a replay then ends on the
        // buffer it started from and zh_graph_launch has no counters to copy (an odd count costs a 16 KiB copy per replay; a
        // separate publish node cost 4.4 us of a 61 us replay in the first version of this path).
        if (use_tab && ctx->capturing && (ctx->capture_flags & ZH_CAPTURE_COALESCE)) {
            const uint32_t key = (zf ? 1u : 0u) | ((uint32_t)sm << 1) | ((uint32_t)fc << 8);
            for (uint32_t b = 0; b < nb; b++) {
                zh_co_batch &cb = ctx->co;
                const float *lo = outs[b].ptr + (size_t)start * outs[b].stride, *hi = outs[b].ptr + (size_t)end * outs[b].stride;
                bool join = cb.active && cb.owner == m && cb.start == start && cb.end == end && cb.stride == outs[b].stride && cb.key == key &&
                            cb.imgs.size() < (size_t)kOscMaxBatch;
                for (size_t q = 0; join && q < cb.imgs.size(); q++) {
                    const float *qlo = cb.imgs[q] + (size_t)start * cb.stride, *qhi = cb.imgs[q] + (size_t)end * cb.stride;
                    if (lo < qhi && qlo < hi) join = false;           // the same rows again: the recorded order decides what they hold
                }
                if (!join) {
                    zh_epoch_flush_batch(ctx, false);
                    cb.active = true; cb.owner = m; cb.start = start; cb.end = end; cb.stride = outs[b].stride; cb.key = key; cb.flips = true;
                    ctx->epoch_open = true;
                    const OscArgs ab = a;
                    cb.launch = [ab, launch, m](hipStream_t s2, float *const *imgs, uint32_t cnt) {
                        OscArgs x = ab;
                        x.cnt_in = m->cnt[m->cur]; x.cnt_out = m->cnt[m->cur ^ 1];
                        x.nb = cnt;
                        for (uint32_t i = 0; i < (uint32_t)kOscMaxBatch; i++) x.img[i] = i < cnt ? imgs[i] : nullptr;
                        launch(x, cnt, s2);
                        zh_flipper_painted(m);                        // (like every batch paint: the state moved to the other buffer)
                        m->cur ^= 1;
                    };
                }
                cb.imgs.push_back(outs[b].ptr);
                ctx->co_paints++;
            }
            return;
        }
        if (ctx->epoch_open) zh_epoch_barrier(ctx);                   // an ordered paint: after what was held back
        for (uint32_t b0 = 0; b0 < nb; b0 += kOscMaxBatch) {
            const uint32_t cnt_b = nb - b0 < (uint32_t)kOscMaxBatch ? nb - b0 : (uint32_t)kOscMaxBatch;
            a.cnt_in = m->cnt[m->cur]; a.cnt_out = m->cnt[m->cur ^ 1];
            a.nb = cnt_b;
            for (uint32_t b = 0; b < (uint32_t)kOscMaxBatch; b++) a.img[b] = b < cnt_b ? outs[b0 + b].ptr : nullptr;
            launch(a, cnt_b, st);
            // the setup form stored this call's constants (ordered before any later paint on the stream)
            if (!use_tab && a.tab) { m->tab.valid = true; m->tab.sample_rate = sample_rate; m->tab.freq = freq; m->tab.color = color; }
            // the batch advanced the state like cnt_b paints in a row but wrote it once, into the other buffer
            zh_flipper_painted(m);
            m->cur ^= 1;
        }
    } else {
        if (ctx->epoch_open) zh_epoch_barrier(ctx);
        dim3 grid((lanes + 63) / 64, (chunks + 3) / 4);
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t *ci = m->cnt[m->cur];
            uint32_t *co = m->cnt[m->cur ^ 1];
            Img out = mk_img(outs[b]);
            if (zf) ZH_LAUNCH((k_osc_const<OSC, true>), grid, dim3(256), 0, st, ci, co, n, out, start, end, fc, srf, sr8, fq, col);
            else ZH_LAUNCH((k_osc_const<OSC, false>), grid, dim3(256), 0, st, ci, co, n, out, start, end, fc, srf, sr8, fq, col);
            zh_flipper_painted(m);
            m->cur ^= 1;
        }
    }
}

template <class M> static int osc_common_check(M *m, uint32_t start, uint32_t end, const zh_buf *outputs, uint32_t nb,
                                               float sample_rate, const zh_cob &freq) {
    (void)sample_rate;
    if (!m || !outputs || end < start) return ZH_ERR_INVALID;
    for (uint32_t b = 0; b < nb; b++)
        if (!buf_covers(outputs[b], m->n, end)) return ZH_ERR_INVALID;
    if (!cob_ok(freq, m->n, end)) return ZH_ERR_INVALID;
    return ZH_OK;
}

template <class M> static int osc_create_common(zh_ctx *ctx, M *m, uint32_t n, int kw) {
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0;
    m->tab.words = nullptr; m->tab.valid = false; m->tab.pinned = false;
    int rc = dev_alloc(&m->cnt[0], n);
    if (!rc) rc = dev_alloc(&m->cnt[1], n);
    if (!rc) rc = dev_alloc(&m->tab.words, (size_t)(kw + 1) * n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[0], 0, (size_t)n * 4, ctx->stream);   // init(): cnt = 0 (PulseOsc.zig:38-42, TriSawOsc.zig:39-44)
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[1], 0, (size_t)n * 4, ctx->stream);
    return rc;
}
template <class M> static void osc_free_common(M *m) { hipFree(m->cnt[0]); hipFree(m->cnt[1]); hipFree(m->tab.words); }

extern "C" {

// -------- PulseOsc
int zh_pulseosc_create(zh_ctx *ctx, uint32_t n, zh_pulseosc **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_pulseosc *m = new (std::nothrow) zh_pulseosc();
    if (!m) return ZH_ERR_INVALID;
    m->part = nullptr;
    int rc = osc_create_common(ctx, m, n, (int)(sizeof(PulseK) / 4));
    if (!rc && n <= kPulsePartMaxVoices) rc = dev_alloc(&m->part, (size_t)64 * n);
    if (rc) { osc_free_common(m); hipFree(m->part); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
int zh_pulseosc_destroy(zh_pulseosc *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    osc_free_common(m); hipFree(m->part);
    delete m;
    return ZH_OK;
}
int zh_pulseosc_get_state(zh_pulseosc *m, zh_pulseosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_download(m->ctx, host, m->cnt[m->cur], (size_t)m->n * 4);
}
int zh_pulseosc_set_state(zh_pulseosc *m, const zh_pulseosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_upload(m->ctx, m->cnt[m->cur], host, (size_t)m->n * 4);
}
static int pulseosc_paint_n(zh_pulseosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, uint32_t nb,
                            const zh_pulseosc_params *p, uint32_t flags) {
    if (!p) return ZH_ERR_INVALID;
    int rc = osc_common_check(m, start, end, outputs, nb, p->sample_rate, p->freq);
    if (rc) return rc;
    if (m->n == 0 || end == start || nb == 0) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    if (p->freq.tag == ZH_COB_CONSTANT) {
        launch_osc_const<PulseOscP>(m, outputs, nb, start, end, p->sample_rate, p->freq.constant, p->color, flags);
    } else {
        if (m->ctx->epoch_open) zh_epoch_barrier(m->ctx);
        const float srf = 4294967296.0f / p->sample_rate;    // SRfcobasefrq, PulseOsc.zig:122
        const float sr8 = p->sample_rate / 8.0f;              // :134
        const F32P col = mk_f32(p->color);
        // 24,576 / 32,768 voices: 122 -> 61, 123 -> 77 us; from 65,536 voices the replay's re-read of the frequency image loses
        bool aliased = false;
        for (uint32_t b = 0; b < nb; b++) aliased = aliased || bufs_alias(p->freq.buffer, outputs[b]);
        const uint32_t chr = aliased ? 0 : zh_range_frames(m->n, end - start, ZF_PULSE_CTRL_RANGES, 2048, 40960);
        for (uint32_t b = 0; b < nb; b++) {
            Img out = mk_img(outputs[b]);
            const uint32_t *ci = m->cnt[m->cur];
            uint32_t *co = chr ? m->cnt[m->cur ^ 1] : m->cnt[m->cur];
            const uint32_t ch = chr ? chr : end - start;
            const dim3 grid((m->n + kSeqBlock - 1) / kSeqBlock, chr ? (end - start + chr - 1) / chr : 1);
            // the ranges' own sums first (1,024 / 4,096 / 16,384 voices: 28.6 / 36.7 / 49.3 -> 14.6 / 16.0 / 37.9 us; level from
            // 32,768 voices on); ZH_PULSE_CTRL_SUMS=0: every range replays the frames before it
            const long sums = zh_form(ZF_PULSE_CTRL_SUMS);
            const uint32_t *part = chr && sums && m->part && grid.y <= 64 ? m->part : nullptr;
            if (part) ZH_LAUNCH(k_pulseosc_ctrl_sums, grid, dim3(kSeqBlock), 0, st, m->part, m->n, start, end, ch, srf, sr8, mk_cimg(p->freq.buffer));
            if (zf) ZH_LAUNCH(k_pulseosc_ctrl<true>, grid, dim3(kSeqBlock), 0, st, ci, co, m->n, out, start, end, ch, srf, sr8, mk_cimg(p->freq.buffer), col, part);
            else ZH_LAUNCH(k_pulseosc_ctrl<false>, grid, dim3(kSeqBlock), 0, st, ci, co, m->n, out, start, end, ch, srf, sr8, mk_cimg(p->freq.buffer), col, part);
            if (chr) { zh_flipper_painted(m); m->cur ^= 1; }
        }
    }
    return zh_launch_status();
}
int zh_pulseosc_paint(zh_pulseosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                      zh_bool note_id_changed, const zh_pulseosc_params *p, uint32_t flags) { ZH_GUARD_EPOCH(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                             // PulseOsc.zig:52-53
    return pulseosc_paint_n(m, start, end, outputs, 1, p, flags);
}
int zh_pulseosc_paint_batch(zh_pulseosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, uint32_t n_buffers,
                            const zh_pulseosc_params *p, uint32_t flags) { ZH_GUARD_EPOCH(m ? m->ctx : nullptr);
    return pulseosc_paint_n(m, start, end, outputs, n_buffers, p, flags);
}

// -------- TriSawOsc
int zh_trisawosc_create(zh_ctx *ctx, uint32_t n, zh_trisawosc **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_trisawosc *m = new (std::nothrow) zh_trisawosc();
    if (!m) return ZH_ERR_INVALID;
    m->t = nullptr; m->t_next = nullptr; m->quot = zh_buf{};
    int rc = osc_create_common(ctx, m, n, (int)(sizeof(TriSawK) / 4));
    if (!rc) rc = dev_alloc(&m->t, n);
    if (!rc) rc = dev_alloc(&m->t_next, n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->t, 0, (size_t)n * 4, ctx->stream);
    if (rc) { osc_free_common(m); hipFree(m->t); hipFree(m->t_next); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
int zh_trisawosc_destroy(zh_trisawosc *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    osc_free_common(m); hipFree(m->t); hipFree(m->t_next); hipFree(m->quot.ptr);
    delete m;
    return ZH_OK;
}
int zh_trisawosc_get_state(zh_trisawosc *m, zh_trisawosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> c(m->n);
    std::vector<float> t(m->n);
    int rc = zh_download(m->ctx, c.data(), m->cnt[m->cur], (size_t)m->n * 4);
    if (!rc) rc = zh_download(m->ctx, t.data(), m->t, (size_t)m->n * 4);
    if (rc) return rc;
    for (uint32_t i = 0; i < m->n; i++) { host[i].cnt = c[i]; host[i].t = t[i]; }
    return ZH_OK;
}
int zh_trisawosc_set_state(zh_trisawosc *m, const zh_trisawosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> c(m->n);
    std::vector<float> t(m->n);
    for (uint32_t i = 0; i < m->n; i++) { c[i] = host[i].cnt; t[i] = host[i].t; }
    int rc = zh_upload(m->ctx, m->cnt[m->cur], c.data(), (size_t)m->n * 4);
    if (!rc) rc = zh_upload(m->ctx, m->t, t.data(), (size_t)m->n * 4);
    return rc;
}
static int trisawosc_paint_n(zh_trisawosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, uint32_t nb,
                             const zh_trisawosc_params *p, uint32_t flags) {
    if (!p) return ZH_ERR_INVALID;
    int rc = osc_common_check(m, start, end, outputs, nb, p->sample_rate, p->freq);
    if (rc) return rc;
    if (m->n == 0 || end == start || nb == 0) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    if (p->freq.tag == ZH_COB_CONSTANT) {
        launch_osc_const<TriSawOscP>(m, outputs, nb, start, end, p->sample_rate, p->freq.constant, p->color, flags);
    } else {
        if (m->ctx->epoch_open) zh_epoch_barrier(m->ctx);
        bool aliased = false;
        for (uint32_t b = 0; b < nb; b++) aliased = aliased || bufs_alias(p->freq.buffer, outputs[b]);
        // 4,096 voices: 106.6 us with per-lane branches in the waveform, 71.3 straight-line, 46.4 as 16 frame ranges (8 / 32 /
        // 64 ranges: 47.9 / 54.7 / 77.2 -- the replay's divide is half of a painted frame); 16,384 voices: 73.7 -> 66.2 with 8
        const uint32_t chr = aliased ? 0 : zh_range_frames(m->n, end - start, ZF_TRISAW_CTRL_RANGES, 1024, 16384);
        // ... and 39.3 us (1,024 voices: 43 -> 26) with the quotients painted first (module-owned image, allocated outside a
        // capture; ZH_TRISAW_CTRL_QUOT=0: never).  (Tried: two or four batches of 32 rows kept in flight by an explicit rotation in
        // these replays -- SineOsc 27 -> 40 / 60 us, TriSawOsc 39 -> 45 / 53: slower; the plain batch loops stay.)
        const long want_quot = zh_form(ZF_TRISAW_CTRL_QUOT);
        bool quot = false;
        if (chr && want_quot) {
            if ((m->quot.frames < end || !m->quot.ptr) && !m->ctx->capturing) {
                zh_buf nbuf;
                if (zh_buf_alloc(m->ctx, &nbuf, m->n, end) == ZH_OK) {
                    if (m->quot.ptr) m->ctx->mix_retired.push_back(m->quot.ptr);   // a captured graph may still name it: freed with the context
                    m->quot = nbuf;
                }
            }
            quot = m->quot.ptr && m->quot.frames >= end;
        }
        if (quot)
            ZH_LAUNCH(k_div_image, dim3((m->n + 63) / 64, ((end - start + 31) / 32 + 3) / 4), dim3(256), 0, st, mk_img(m->quot), mk_cimg(p->freq.buffer),
                               m->n, start, end, p->sample_rate);
        const CImg fimg = quot ? mk_cimg(m->quot) : mk_cimg(p->freq.buffer);
        for (uint32_t b = 0; b < nb; b++) {
            Img out = mk_img(outputs[b]);
            float *to = chr ? m->t_next : m->t;
            const uint32_t ch = chr ? chr : end - start;
            const dim3 grid((m->n + kSeqBlock - 1) / kSeqBlock, chr ? (end - start + chr - 1) / chr : 1);
#define ZH_TSC(ZF_, Q_) ZH_LAUNCH((k_trisawosc_ctrl<ZF_, Q_>), grid, dim3(kSeqBlock), 0, st, m->t, to, m->n, out, start, end, ch, p->sample_rate, fimg, mk_f32(p->color))
            if (zf) { if (quot) ZH_TSC(true, true); else ZH_TSC(true, false); }
            else { if (quot) ZH_TSC(false, true); else ZH_TSC(false, false); }
#undef ZH_TSC
            if (chr) ZH_LAUNCH(k_commit_f32, dim3((m->n + 255) / 256), dim3(256), 0, st, m->t, m->t_next, m->n);
        }
    }
    return zh_launch_status();
}
int zh_trisawosc_paint(zh_trisawosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                       zh_bool note_id_changed, const zh_trisawosc_params *p, uint32_t flags) { ZH_GUARD_EPOCH(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                             // TriSawOsc.zig:54-55
    return trisawosc_paint_n(m, start, end, outputs, 1, p, flags);
}
int zh_trisawosc_paint_batch(zh_trisawosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, uint32_t n_buffers,
                             const zh_trisawosc_params *p, uint32_t flags) { ZH_GUARD_EPOCH(m ? m->ctx : nullptr);
    return trisawosc_paint_n(m, start, end, outputs, n_buffers, p, flags);
}

}  // extern "C"
