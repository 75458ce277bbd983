// modules.hip -- paint kernels for SineOsc, Noise, Envelope, Gate, Filter, Sampler, Decimator
// and Distortion (src/modules/*.zig), one wavefront lane per voice.
//
// Stateful modules (SineOsc, Noise, Envelope, Filter, Sampler, Decimator) carry an f32/u64
// recurrence from sample to sample that must be replayed add by add to reproduce the
// reference's bits, so they use the sequential frame loop of seq.hip.h: state is
// loaded into VGPRs once per span, walked over the span, stored once.  Stateless modules
// (Gate, Distortion) are additionally split into frame chunks across waves.
#include "common.hip.h"
#include "ring.hip.h"
#include <string.h>
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "seq.hip.h"
#include "envelope.hip.h"
#include "voices.hip.h"
#define ZH_FILTER_TP_PINK 1
#include "filter_tp.hip.h"
#include <vector>

template <typename T> static int upload_field(zh_ctx *ctx, T *dev, const std::vector<T> &h) {
    return zh_upload(ctx, dev, h.data(), h.size() * sizeof(T));
}
template <typename T> static int download_field(zh_ctx *ctx, std::vector<T> &h, const T *dev, size_t n) {
    h.resize(n);
    return zh_download(ctx, h.data(), dev, n * sizeof(T));
}

// =================================================================== SineOsc
// The phase `t` is double-buffered like the chunked oscillators' counters (zh_flipper: a captured graph bakes both pointers
// in, zh_graph_launch reconciles): the frame-range kernel's ranges all read the span's start phase while the last range
// publishes the end phase.  cnt[] holds the f32 bits.
struct zh_sineosc : zh_flipper {
    float *t() const { return reinterpret_cast<float *>(cnt[cur]); }
};

template <bool ZF, bool FB, bool PB, bool TOL = false>
__global__ void __launch_bounds__(kSeqBlock) k_sineosc(float *__restrict__ t_io, uint32_t V, Img out, uint32_t start,
                                                       uint32_t end, float sample_rate, CobP freq, CobP phase) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    SineOscLane o;
    o.t = t_io[v];
    constexpr int NIN = (FB ? 1 : 0) + (PB ? 1 : 0);
    const float *ins[2] = {nullptr, nullptr};
    size_t istr[2] = {0, 0};
    if (FB) { ins[0] = freq.b.p; istr[0] = freq.b.stride; }
    if (PB) { ins[FB ? 1 : 0] = phase.b.p; istr[FB ? 1 : 0] = phase.b.stride; }
    o.begin(sample_rate, FB ? 0.0f : freq.c.get(v));
    const float phase_c = PB ? 0.0f : phase.c.get(v);
    if constexpr (TOL) {
        frame_loop<8, ZF, NIN>(out.p, v, out.stride, ins, istr, start, end,
                               [&](uint32_t, const float (&x)[NIN > 0 ? NIN : 1], float &val) ZH_INLINE_LAMBDA {
            val = o.template frame<FB, 2>(FB ? x[0] : 0.0f, PB ? x[(FB && NIN > 1) ? 1 : 0] : phase_c);
            return true;
        });
    } else if constexpr (!FB && !PB) {
        // chunks whose arguments stay far below zsinf's Payne-Hanek range (every chunk, in practice) run the sine without its
        // rare-path branch: the eight sines of a chunk are then one basic block and interleave
        frame_loop_gen<8, ZF>(out.p, v, out.stride, start, end, [&](uint32_t) ZH_INLINE_LAMBDA { return o.small_args(phase_c, 8.0f); },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.template frame<false, false>(0.0f, phase_c); return true; },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.template frame<false, true>(0.0f, phase_c); return true; });
    } else {
        frame_loop<8, ZF, NIN>(out.p, v, out.stride, ins, istr, start, end,
                               [&](uint32_t, const float (&x)[NIN > 0 ? NIN : 1], float &val) ZH_INLINE_LAMBDA {
            val = o.template frame<FB>(x[0], PB ? x[FB ? 1 : 0] : phase_c);
            return true;
        });
    }
    o.end();
    t_io[v] = o.t;
}

// Few voices: the span as many frame ranges at once, one wave per (64 voices, range).  The phase that reaches frame f0 is
// the span's start phase after f0 - start additions of `t_step` (or of `freq[i] * inv_sr`), each rounded to f32 -- so a
// range first REPLAYS those additions (one dependent add per earlier frame: cheap beside the ~50 instructions of a musl
// sine) and then paints its own frames exactly like k_sineosc; the range that ends the span publishes the wrapped phase.
// Every frame is written once, so `+=` paints need no scratch.  147 -> see profiles/r04/NOTES.md 5a at 4,096 voices.
template <bool ZF, bool FB, bool PB, bool TOL = false>
__global__ void __launch_bounds__(64) k_sineosc_ranges(const float *__restrict__ t_in, float *__restrict__ t_out, uint32_t V, Img out,
                                                       uint32_t start, uint32_t end, uint32_t ch, float sample_rate, CobP freq, CobP phase) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    SineOscLane o;
    o.t = t_in[v];
    o.begin(sample_rate, FB ? 0.0f : freq.c.get(v));
    if (FB) {
        replay_rows(freq.b.p + (size_t)start * freq.b.stride + v, freq.b.stride, f0 - start,
                    [&](float x) ZH_INLINE_LAMBDA { o.t += x * o.inv_sr; });                // SineOsc.zig:73 / :83
    } else {
        // (ranges start at multiples of 8 frames from the span start: eight dependent adds per loop round)
        uint32_t i = start;
        for (; i + 8 <= f0; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) o.t += o.t_step;                          // :50 / :59
        }
        for (; i < f0; i++) o.t += o.t_step;
    }
    constexpr int NIN = (FB ? 1 : 0) + (PB ? 1 : 0);
    const float *ins[2] = {nullptr, nullptr};
    size_t istr[2] = {0, 0};
    if (FB) { ins[0] = freq.b.p; istr[0] = freq.b.stride; }
    if (PB) { ins[FB ? 1 : 0] = phase.b.p; istr[FB ? 1 : 0] = phase.b.stride; }
    const float phase_c = PB ? 0.0f : phase.c.get(v);
    if constexpr (TOL) {
        frame_loop<8, ZF, NIN>(out.p, v, out.stride, ins, istr, f0, f1,
                               [&](uint32_t, const float (&x)[NIN > 0 ? NIN : 1], float &val) ZH_INLINE_LAMBDA {
            val = o.template frame<FB, 2>(FB ? x[0] : 0.0f, PB ? x[(FB && NIN > 1) ? 1 : 0] : phase_c);
            return true;
        });
    } else if constexpr (!FB && !PB) {
        frame_loop_gen<8, ZF>(out.p, v, out.stride, f0, f1, [&](uint32_t) ZH_INLINE_LAMBDA { return o.small_args(phase_c, 8.0f); },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.template frame<false, false>(0.0f, phase_c); return true; },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.template frame<false, true>(0.0f, phase_c); return true; });
    } else {
        frame_loop<8, ZF, NIN>(out.p, v, out.stride, ins, istr, f0, f1,
                               [&](uint32_t, const float (&x)[NIN > 0 ? NIN : 1], float &val) ZH_INLINE_LAMBDA {
            val = o.template frame<FB>(x[0], PB ? x[FB ? 1 : 0] : phase_c);
            return true;
        });
    }
    if (f1 == end) { o.end(); t_out[v] = o.t; }
}

// =================================================================== Noise
struct zh_noise {
    zh_ctx *ctx; uint32_t n; uint64_t *s[4]; float *b; /* [7][n] */
    // the frame-range form of white noise (noise_jump.hip): states after the span, multi-draw flags, and the image an
    // ADD paint is rendered into before it is added to the output
    uint64_t *nx[4]; uint32_t *flag; zh_buf scratch;
    uint32_t *err;               // k_pink_pipe: a ring wait ran into its bound (never in a correct run; reported by get_state)
    // ZH_PAINT_TOLERANT pink noise (filter_tp.hip.h k_pink_tp_a / _b): scratch, allocated by the first tolerant paint outside a capture
    uint64_t *tp_cs; float *tp_e; uint32_t *tp_flag; uint32_t tp_serial;
    float *tp_taps;              // [2][7][n]: the taps a tolerant pink paint's piece ends on, read by the span's next piece (Noise.zig:55: `var b` runs over the whole span)
};
// noise_jump.hip
uint32_t zh_noise_range_frames(uint32_t V, uint32_t n);
int zh_noise_paint_ranges(zh_ctx *ctx, uint64_t *const s[4], uint64_t *const next[4], uint32_t *flag, uint32_t V, const zh_buf &outb,
                          uint32_t start, uint32_t end, uint32_t ch);
const uint4 *zh_noise_jump_tables(zh_ctx *ctx);

__global__ void k_noise_seed(uint64_t *s0, uint64_t *s1, uint64_t *s2, uint64_t *s3, float *b, uint32_t n, uint64_t first_seed) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    ZXoshiro r;
    zxoshiro_seed(r, first_seed + v);                                 // Noise.zig:26-29
    s0[v] = r.s0; s1[v] = r.s1; s2[v] = r.s2; s3[v] = r.s3;
    for (int j = 0; j < 7; j++) b[(size_t)j * n + v] = 0.0f;          // :30
}

template <bool ZF, bool PINK>
__global__ void __launch_bounds__(kSeqBlock) k_noise(uint64_t *__restrict__ s0, uint64_t *__restrict__ s1,
                                                     uint64_t *__restrict__ s2, uint64_t *__restrict__ s3,
                                                     const float *__restrict__ bst, uint32_t V, Img out,
                                                     uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    NoiseLane o;
    o.r = ZXoshiro{s0[v], s1[v], s2[v], s3[v]};
    o.begin();
    if (PINK) {
#pragma unroll
        for (int j = 0; j < 7; j++) o.b[j] = bst[(size_t)j * V + v];  // `var b = self.b` (Noise.zig:55); always 0, see :68
    }
    const float *const *no_in = nullptr;
    frame_loop<8, ZF, 0>(out.p, v, out.stride, no_in, nullptr, start, end, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
        val = o.template frame<PINK>();
        return true;
    });
    // Noise.zig:68 is `b = self.b;` -- the taps are never written back (reference quirk, kept)
    s0[v] = o.r.s0; s1[v] = o.r.s1; s2[v] = o.r.s2; s3[v] = o.r.s3;   // :71
}

// Pink noise at a small voice count, from a rendered white image (the frame-range kernel of noise_jump.hip paints it first).
// Paul Kellett's filter (Noise.zig:58-66) is six independent one-pole taps over the white samples plus a sum in a fixed order
// -- ~30 dependent-issue instructions per sample for one wave.  Here seven waves per 64 voices form a pipeline through LDS:
// wave k = 0..5 runs tap k and adds it to the running sum of the taps before it (s_k = s_{k-1} + b_k, the reference's
// left-to-right order), the last wave adds b[6] and white * 0.5362, steps b[6] and does the `+=` into the output image.
// 32-frame tiles, two slots per stage boundary, monotonic tile counters polled in LDS (ring.hip.h).  Per tile a stage reads
// the incoming sums in one batch (read by read the compiler waits out the LDS latency in every frame), frees the slot, runs
// its 3-4 VALU instructions and one ds_write per sample, and publishes.  Every wave fetches the white rows it needs from the
// image itself (L2), one tile ahead: the ring waits are compiler barriers, nothing else would move a load across them.
constexpr uint32_t kPinkWaves = 7, kPinkCH = 32, kPinkNS = 2;         // (16-frame tiles with 4 slots: 67 -> 77 us at 4,096 voices)
struct PinkShared {
    float sums[6][kPinkNS][kPinkCH][64];
    uint32_t ready[6], freed[6];                                      // boundary k: tiles stage k has written / stage k + 1 has read
};
struct PinkTiles {
    uint32_t n, nt, lane, voice;                                      // voice = the lane's (clamped) voice: its offset inside a row
    __device__ __forceinline__ uint32_t frames(uint32_t c) const { return c < nt ? min(kPinkCH, n - c * kPinkCH) : 0; }
    // rows [c * CH, c * CH + frames(c)) of one voice column; a full tile runs unguarded, fully unrolled code
    // (`base` = the span's first row, wave-uniform)
    __device__ __forceinline__ void load(float (&dst)[kPinkCH], const float *base, size_t stride, uint32_t c) const {
        const uint32_t nf = frames(c);
        const float *p0 = base + voice + (size_t)(c * kPinkCH) * stride;
        if (nf == kPinkCH) {                                          // (plain global loads: through a buffer descriptor with scalar row offsets the kernel measured 5 % slower)
#pragma unroll
            for (uint32_t k = 0; k < kPinkCH; k++) dst[k] = p0[(size_t)k * stride];
        } else {
#pragma unroll
            for (uint32_t k = 0; k < kPinkCH; k++) dst[k] = k < nf ? p0[(size_t)k * stride] : 0.0f;
        }
    }
};
// KIND 0: tap 0 (nothing to add to), 1: taps 1..4, 2: tap 5 (`-0.7616 * b - white * 0.0168980`, :64)
template <int KIND>
__device__ __forceinline__ bool pink_tap_stage(PinkShared &sh, const PinkTiles &t, uint32_t stage, float cc, float dd, float b, const float *wp, size_t wstride) {
    constexpr uint32_t CH = kPinkCH, NS = kPinkNS;
    bool ok = true;
    auto tile = [&](const float (&w)[CH], float (&wn)[CH], uint32_t c) ZH_INLINE_LAMBDA {
        const uint32_t nf = t.frames(c), slot = c & (NS - 1);
        if (KIND > 0) ok = ring_wait_ge(&sh.ready[stage - 1], c + 1);
        if (c >= NS) ok = ok && ring_wait_ge(&sh.freed[stage], c + 1 - NS);
        float sp[CH];
        if (KIND > 0) {
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) sp[k] = sh.sums[stage - 1][slot][k][t.lane];
            ring_publish(&sh.freed[stage - 1], c + 1, t.lane);
        }
        float (*sout)[64] = sh.sums[stage][slot];
        auto tap = [&](uint32_t k) ZH_INLINE_LAMBDA {
            const float m1 = cc * b, m2 = w[k] * dd;                  // :59-64
            b = KIND == 2 ? m1 - m2 : m1 + m2;
            sout[k][t.lane] = KIND > 0 ? sp[k] + b : b;               // :65, left to right
        };
        // the next tile's white rows are requested right after this tile's first use of its own: the compiler waits for
        // EVERY outstanding load at that first use (vmcnt(0)), so a request made earlier would be waited out on the spot
        if (nf == CH) {
            tap(0);
            __builtin_amdgcn_sched_barrier(0);
            t.load(wn, wp, wstride, c + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t k = 1; k < CH; k++) tap(k);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) if (k < nf) tap(k);
        }
        ring_publish(&sh.ready[stage], c + 1, t.lane);
    };
    float wa[CH], wb[CH];
    t.load(wa, wp, wstride, 0);
    for (uint32_t c = 0; c < t.nt && ok; c += 2) {                    // two tiles per turn: the white rows ping-pong between wa and wb
        tile(wa, wb, c);
        if (c + 1 < t.nt && ok) tile(wb, wa, c + 1);
    }
    return ok;
}
template <bool ZF>
__global__ void __launch_bounds__(64 * kPinkWaves) k_pink_pipe(const float *__restrict__ bst, uint32_t V, Img out, CImg white, uint32_t start,
                                                               uint32_t end, uint32_t *__restrict__ err) {
    constexpr uint32_t CH = kPinkCH, NS = kPinkNS;
    __shared__ PinkShared sh;
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t v = blockIdx.x * 64 + lane, vc = min(v, V - 1);
    if (threadIdx.x < 6) { sh.ready[threadIdx.x] = 0; sh.freed[threadIdx.x] = 0; }
    __syncthreads();
    const uint32_t n = end - start;
    const PinkTiles t{n, (n + CH - 1) / CH, lane, vc};
    const float *wp = white.p + (size_t)start * white.stride;         // wave-uniform; the lane adds its voice
    bool ok = true;
    if (wave < 6) {
        // Noise.zig:59-64
        const float cc = wave == 0 ? 0.99886f : wave == 1 ? 0.99332f : wave == 2 ? 0.96900f : wave == 3 ? 0.86650f : wave == 4 ? 0.55000f : -0.7616f;
        const float dd = wave == 0 ? 0.0555179f : wave == 1 ? 0.0750759f : wave == 2 ? 0.1538520f : wave == 3 ? 0.3104856f : wave == 4 ? 0.5329522f : 0.0168980f;
        const float b = bst[(size_t)wave * V + vc];                   // `var b = self.b` (:55)
        if (wave == 0) ok = pink_tap_stage<0>(sh, t, wave, cc, dd, b, wp, white.stride);
        else if (wave == 5) ok = pink_tap_stage<2>(sh, t, wave, cc, dd, b, wp, white.stride);
        else ok = pink_tap_stage<1>(sh, t, wave, cc, dd, b, wp, white.stride);
    } else {
        float b6 = bst[(size_t)6 * V + vc];
        float *ob = out.p + (size_t)start * out.stride, *op = ob + vc;
        auto tile = [&](const float (&w)[CH], float (&wn)[CH], const float (&base)[CH], float (&basen)[CH], uint32_t c) ZH_INLINE_LAMBDA {
            const uint32_t nf = t.frames(c), slot = c & (NS - 1);
            ok = ring_wait_ge(&sh.ready[5], c + 1);
            float res[CH], sp[CH];
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) sp[k] = sh.sums[5][slot][k][lane];
            ring_publish(&sh.freed[5], c + 1, lane);
            auto fin = [&](uint32_t k) ZH_INLINE_LAMBDA {
                const float val = (sp[k] + b6) + w[k] * 0.5362f;      // :65
                b6 = w[k] * 0.115926f;                                // :66
                res[k] = (ZF ? 0.0f : base[k]) + val;
            };
            float *o0 = op + (size_t)(c * CH) * out.stride;
            if (nf == CH) {
                fin(0);
                __builtin_amdgcn_sched_barrier(0);
                t.load(wn, wp, white.stride, c + 1);                  // (see pink_tap_stage)
                if (!ZF) t.load(basen, ob, out.stride, c + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (uint32_t k = 1; k < CH; k++) fin(k);
                if (v < V) {
#pragma unroll
                    for (uint32_t k = 0; k < CH; k++) o0[(size_t)k * out.stride] = res[k];
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) if (k < nf) fin(k);
                if (v < V) {
#pragma unroll
                    for (uint32_t k = 0; k < CH; k++) if (k < nf) o0[(size_t)k * out.stride] = res[k];
                }
            }
        };
        float wa[CH], wb[CH], ba[CH], bb[CH];
        t.load(wa, wp, white.stride, 0);
        if (!ZF) t.load(ba, ob, out.stride, 0);
        for (uint32_t c = 0; c < t.nt && ok; c += 2) {
            tile(wa, wb, ba, bb, c);
            if (c + 1 < t.nt && ok) tile(wb, wa, bb, ba, c + 1);
        }
    }
    if (!ok && lane == 0) *err = 1u;
}

// The pipeline above puts seven waves on a CU's four SIMDs (waves that share a SIMD share its one VALU issue per four cycles)
// and a stage's tile costs ~1.5 us -- 48 us for 1,024 frames.  Here FOUR waves per 64 voices, one per SIMD, one barrier per
// 32-frame step, every role in its own copy of the step loop, tiles of float4 (four frames of a lane side by side: a 16-byte
// LDS access per four frames):
//   wave 0   taps 0, 1                s1 = b0 + b1
//   wave 1   taps 2, 3                s3 = (s1 + b2) + b3          one step behind
//   wave 2   taps 4, 5                s5 = (s3 + b4) + b5          two steps behind
//   wave 3   (s5 + b6) + white * 0.5362, b6, `+=`, the store (Noise.zig:65-66), three steps behind; it also fetches the white
//            rows for everybody: requested three tiles ahead -- UNCONDITIONALLY: a request inside a branch makes the row arrays
//            meet at the join, where the compiler waits for every outstanding load -- and published as a float4 tile a step
//            before wave 0 needs it (every wave fetching its own rows was a quarter of the instructions of each: kernel 40 -> 34 us)
// The sums are the reference's left-to-right chain cut at three places => same operations on the same values, same bits.
// Needs one whole tile in the span (n >= 32): requests past the last whole tile re-read it.
constexpr uint32_t kPinkWhiteSlots = 8;
template <uint32_t LAG, uint32_t CH>
__device__ __forceinline__ void pink_pair_role(float4 (*wq)[CH / 4][64], float4 (*sin)[CH / 4][64], float4 (*sout)[CH / 4][64], uint32_t lane, uint32_t n, uint32_t nt,
                                               uint32_t steps, float ca, float da, float ba, float cb, float db, float bb) {
    constexpr uint32_t Q = CH / 4;
    constexpr bool MINUS = LAG == 2;                                  // tap 5: `-0.7616 * b - white * 0.0168980` (:64)
    auto frames = [&](uint32_t c) ZH_INLINE_LAMBDA { return c < nt ? min(CH, n - c * CH) : 0u; };
    auto pair = [&](float w, float s) ZH_INLINE_LAMBDA {              // Noise.zig:59-65: two taps, added to the sum so far
        ba = ca * ba + w * da;
        const float m1 = cb * bb, m2 = w * db;
        bb = MINUS ? m1 - m2 : m1 + m2;
        return LAG == 0 ? ba + bb : (s + ba) + bb;
    };
    for (uint32_t c = 0; c < steps; c++) {                            // step c: tile c - LAG (an index wrapped below zero has no frames)
        const uint32_t t = c - LAG, nf = frames(t);
        float4 (*tw)[64] = wq[t & (kPinkWhiteSlots - 1)], (*ti)[64] = sin[t & 1], (*to)[64] = sout[t & 1];
        if (nf == CH) {
            float4 w[Q], xs[Q];
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) { w[q] = tw[q][lane]; if (LAG > 0) xs[q] = ti[q][lane]; }
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) {
                const float s0 = pair(w[q].x, LAG > 0 ? xs[q].x : 0.0f), s1 = pair(w[q].y, LAG > 0 ? xs[q].y : 0.0f);
                const float s2 = pair(w[q].z, LAG > 0 ? xs[q].z : 0.0f), s3 = pair(w[q].w, LAG > 0 ? xs[q].w : 0.0f);
                to[q][lane] = make_float4(s0, s1, s2, s3);
            }
        } else {
            for (uint32_t k = 0; k < nf; k++) {
                const float s = LAG > 0 ? reinterpret_cast<const float *>(&ti[k >> 2][lane])[k & 3] : 0.0f;
                reinterpret_cast<float *>(&to[k >> 2][lane])[k & 3] = pair(reinterpret_cast<const float *>(&tw[k >> 2][lane])[k & 3], s);
            }
        }
        __syncthreads();
    }
}
// CH = frames per tile: 32 (112 KB of LDS: one workgroup per CU = 16,384 voices at once) or 16 (56 KB: two per CU)
template <bool ZF, uint32_t CH>
__global__ void __launch_bounds__(256) k_pink_taps(const float *__restrict__ bst, uint32_t V, Img out, CImg white, uint32_t start, uint32_t end) {
    constexpr uint32_t Q = CH / 4, NW = kPinkWhiteSlots;
    __shared__ float4 w_q[NW][Q][64];                                 // white tiles: written a step before wave 0 reads them, last read by wave 3 four steps later
    __shared__ float4 s_q[3][2][Q][64];                               // s1, s3, s5: two tiles each
    const uint32_t lane = threadIdx.x & 63, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t v = blockIdx.x * 64 + lane, vc = min(v, V - 1);
    const uint32_t n = end - start, nt = (n + CH - 1) / CH, whole = n / CH;
    const uint32_t steps = (nt + 3 + 1) / 2 * 2;                      // tile t leaves the last wave at step t + 3; an even count, the same for every role
    const uint32_t voff = vc * 4u;
    auto frames = [&](uint32_t c) ZH_INLINE_LAMBDA { return c < nt ? min(CH, n - c * CH) : 0u; };
    auto at = [&](float4 (*t)[64], uint32_t k) ZH_INLINE_LAMBDA -> float & { return reinterpret_cast<float *>(&t[k >> 2][lane])[k & 3]; };
    auto tap_b = [&](uint32_t k) ZH_INLINE_LAMBDA { return bst[(size_t)k * V + vc]; };   // `var b = self.b` (:55)
    if (role < 3) {
        __syncthreads();                                              // (tile 0 of the white rows is in place)
        // Noise.zig:59-64
        if (role == 0) pink_pair_role<0, CH>(w_q, s_q[0], s_q[0], lane, n, nt, steps, 0.99886f, 0.0555179f, tap_b(0), 0.99332f, 0.0750759f, tap_b(1));
        else if (role == 1) pink_pair_role<1, CH>(w_q, s_q[0], s_q[1], lane, n, nt, steps, 0.96900f, 0.1538520f, tap_b(2), 0.86650f, 0.3104856f, tap_b(3));
        else pink_pair_role<2, CH>(w_q, s_q[1], s_q[2], lane, n, nt, steps, 0.55000f, 0.5329522f, tap_b(4), -0.7616f, 0.0168980f, tap_b(5));
    } else {
        float b6 = tap_b(6);
        const uint32_t wrow = (uint32_t)white.stride * 4u, orow = (uint32_t)out.stride * 4u;
        auto request_w = [&](uint32_t c, float (&w)[CH]) ZH_INLINE_LAMBDA {
            const zh_rsrc_t rw = zrow_rsrc(white.p, white.stride, start + min(c, whole - 1) * CH);
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) w[k] = zrow_load<1>(rw, voff, k * wrow);
        };
        auto request_b = [&](uint32_t c, float (&base)[CH]) ZH_INLINE_LAMBDA {
            if (ZF) return;
            const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + min(c, whole - 1) * CH);
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) base[k] = zrow_load<1>(ro, voff, k * orow);
        };
        auto publish = [&](uint32_t c, const float (&w)[CH]) ZH_INLINE_LAMBDA {
            const uint32_t nf = frames(c);
            float4 (*t)[64] = w_q[c & (NW - 1)];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) t[q][lane] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
            } else if (nf > 0) {
                const zh_rsrc_t rw = zrow_rsrc(white.p, white.stride, start + c * CH);
                for (uint32_t k = 0; k < nf; k++) at(t, k) = zrow_load<1>(rw, voff, k * wrow);
            }
        };
        auto tile = [&](uint32_t c, const float (&base)[CH]) ZH_INLINE_LAMBDA {
            const uint32_t nf = frames(c);
            float4 (*si)[64] = s_q[2][c & 1], (*tw)[64] = w_q[c & (NW - 1)];
            const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + min(c, nt - 1) * CH);
            auto fin = [&](float s5, float wk, float bk) ZH_INLINE_LAMBDA {
                const float val = (s5 + b6) + wk * 0.5362f;           // :65
                b6 = wk * 0.115926f;                                  // :66
                return (ZF ? 0.0f : bk) + val;
            };
            if (nf == CH) {
                float4 xs[Q], w[Q];
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) { xs[q] = si[q][lane]; w[q] = tw[q][lane]; }
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    const float r0 = fin(xs[q].x, w[q].x, base[4 * q]), r1 = fin(xs[q].y, w[q].y, base[4 * q + 1]);
                    const float r2 = fin(xs[q].z, w[q].z, base[4 * q + 2]), r3 = fin(xs[q].w, w[q].w, base[4 * q + 3]);
                    if (v < V) {
                        zrow_store<1>(ro, voff, (4 * q) * orow, r0); zrow_store<1>(ro, voff, (4 * q + 1) * orow, r1);
                        zrow_store<1>(ro, voff, (4 * q + 2) * orow, r2); zrow_store<1>(ro, voff, (4 * q + 3) * orow, r3);
                    }
                }
            } else {
                for (uint32_t k = 0; k < nf; k++) {
                    const float r = fin(at(si, k), at(tw, k), ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
                    if (v < V) zrow_store<1>(ro, voff, k * orow, r);
                }
            }
        };
        float wa[CH], wb[CH], ba[CH], bb[CH];                         // white rows on their way to LDS and output rows, of the even / odd tiles
        request_w(0, wa); request_w(1, wb);
        request_b(0, ba); request_b(1, bb);
        publish(0, wa); request_w(2, wa);
        __syncthreads();
        for (uint32_t c = 0; c < steps; c += 2) {
            // step c (even): white tile c + 1 goes to LDS, tile c + 3 is requested; this wave's own tile is c - 3 (odd)
            publish(c + 1, wb); request_w(c + 3, wb);
            tile(c - 3, bb); request_b(c - 3 + 2, bb);
            __syncthreads();
            publish(c + 2, wa); request_w(c + 4, wa);
            tile(c - 2, ba); request_b(c - 2 + 2, ba);
            __syncthreads();
        }
    }
}

// =================================================================== Envelope
// The state is double-buffered (zh_flipper, 4 words per voice: state, t, last_value, start as [4][n]): the frame-range kernel
// reads the start state from one buffer while its last range writes the end state into the other, and the host flips (what a
// captured graph baked in is reconciled at launch, ctx.hip).  The one-wave walk updates the current buffer in place.
struct zh_envelope : zh_flipper {
    uint32_t *st(int b) const { return cnt[b]; }
    float *f(int b, int k) const { return reinterpret_cast<float *>(cnt[b] + (size_t)k * n); }
};

// FT >= 0: the three curves share that tag (the usual case; the host checks), so the per-frame curve needs no selects.
// Chunks of 8 frames in which no voice of the wave can end a stage run EnvLane::frame_quiet (frame_loop_gen).
// 4,096 voices: 84.6 us with the generic frame() in every frame (54 instructions), 35 us with quiet chunks (a third form for
// chunks without any voice in a timed stage -- a constant per voice -- made it 48: the uniform test inside the unrolled frames cost more than it saved).
template <bool ZF, int FT>
__global__ void __launch_bounds__(kSeqBlock) k_envelope(uint32_t *__restrict__ st, float *__restrict__ t,
                                                        float *__restrict__ lastv, float *__restrict__ startv, uint32_t V,
                                                        Img out, uint32_t start, uint32_t end, EnvParamsP p, BoolP nic) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    EnvLaneT<1, FT> e;
    e.state = st[v]; e.t = t[v]; e.last_value = lastv[v]; e.start = startv[v];
    env_load(e, p, v);
    e.begin(nic.get(v));
    frame_loop_gen<8, ZF>(out.p, v, out.stride, start, end, [&](uint32_t) ZH_INLINE_LAMBDA { return e.quiet(8); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { return e.frame_quiet(val); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { return e.frame(val); });
    st[v] = e.state; t[v] = e.t; lastv[v] = e.last_value; startv[v] = e.start;
}

// A span as frame ranges (grid.y) at small voice counts.  The envelope's walk from frame to frame is its clock: a range
// replays the frames before it 8 at a time where no voice of the wave can end a stage (EnvLaneT::quiet / skip_quiet<8>: eight
// additions and one curve evaluation per chunk), frame by frame around a stage end, then paints its own frames like
// k_envelope.  The range that ends the span writes the state into the other half of the double buffer (`next`, [4][V]).
template <bool ZF, int FT>
__global__ void __launch_bounds__(64) k_envelope_ranges(const uint32_t *__restrict__ st, const float *__restrict__ t, const float *__restrict__ lastv,
                                                        const float *__restrict__ startv, uint32_t *__restrict__ next, uint32_t V, Img out,
                                                        uint32_t start, uint32_t end, uint32_t ch, EnvParamsP p, BoolP nic) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    EnvLaneT<1, FT> e;
    e.state = st[v]; e.t = t[v]; e.last_value = lastv[v]; e.start = startv[v];
    env_load(e, p, v);
    e.begin(nic.get(v));
    uint32_t i = start;
    // (32 frames per test where that holds: a test is six instructions and a scalar branch, as much as eight clock steps)
    for (; i + 32 <= f0 && e.quiet(32); i += 32) e.template skip_clock<32>();
    for (; i + 8 <= f0; i += 8) {
        if (e.quiet(8)) e.template skip_clock<8>();
        else {
#pragma unroll
            for (int k = 0; k < 8; k++) { float val; (void)e.frame(val); }
        }
    }
    for (; i < f0; i++) { float val; (void)e.frame(val); }
    frame_loop_gen<8, ZF>(out.p, v, out.stride, f0, f1, [&](uint32_t) ZH_INLINE_LAMBDA { return e.quiet(8); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { return e.frame_quiet(val); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { return e.frame(val); });
    if (f1 == end) {
        next[v] = e.state; next[(size_t)V + v] = __builtin_bit_cast(uint32_t, e.t);
        next[(size_t)2 * V + v] = __builtin_bit_cast(uint32_t, e.last_value); next[(size_t)3 * V + v] = __builtin_bit_cast(uint32_t, e.start);
    }
}
// =================================================================== Gate (stateless)
struct zh_gate { zh_ctx *ctx; uint32_t n; };

// Gate.zig:28-30: if note_on, out[span] += 1.0.  grid: x = 64-voice groups, y = 32-frame chunks.
template <bool ZF>
__global__ void __launch_bounds__(256) k_gate(uint32_t V, Img out, uint32_t start, uint32_t end, BoolP note_on) {
    const uint32_t v = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const uint32_t c0 = start + chunk * 32, c1 = min(c0 + 32, end);
    const bool on = note_on.get(v);
    if (!on && !ZF) return;
    float *o = out.at(c0, v);
    uint32_t i = c0;
    for (; i + 8 <= c1; i += 8, o += 8 * (size_t)out.stride) {        // `+=`: 8 rows' loads ahead of their stores
        float base[8];
#pragma unroll
        for (int k = 0; k < 8; k++) base[k] = ZF ? 0.0f : o[(size_t)k * out.stride];
#pragma unroll
        for (int k = 0; k < 8; k++) o[(size_t)k * out.stride] = on ? base[k] + 1.0f : base[k];   // (plain stores: non-temporal ones measured 9.5 % slower at 131,072 voices, 3.6 % faster at 4,096)
    }
    for (; i < c1; i++, o += out.stride) {
        const float base = ZF ? 0.0f : *o;
        *o = on ? base + 1.0f : base;
    }
}

// Four voices per lane, three consecutive rows of a 256-voice column per wave, a workgroup = four consecutive chunks: the chunked
// oscillator's launch shape (16-byte write-through stores).  Taken when the voice count and the image allow 16-byte accesses.
// 4,096 voices 5.6 -> 4.5 us, 131,072 voices 98.1 -> 87.6 (profiles/r05/ab_gate4.txt).
template <bool ZF>
__global__ void __launch_bounds__(256) k_gate4(uint32_t nvq, Img out, uint32_t start, uint32_t end, BoolP note_on) {
    const uint32_t q = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (q >= nvq) return;
    const uint32_t v = q * 4, r0 = start + chunk * 3;
    bool on[4];
#pragma unroll
    for (int j = 0; j < 4; j++) on[j] = note_on.get(v + j);
    if (!ZF && !(on[0] || on[1] || on[2] || on[3])) return;
    zv4f base[3];
#pragma unroll
    for (int k = 0; k < 3; k++) base[k] = (!ZF && r0 + k < end) ? *reinterpret_cast<const zv4f *>(out.at(r0 + k, v)) : zv4f{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (r0 + k < end) {
            zv4f o = base[k];
            o.x = on[0] ? o.x + 1.0f : o.x; o.y = on[1] ? o.y + 1.0f : o.y; o.z = on[2] ? o.z + 1.0f : o.z; o.w = on[3] ? o.w + 1.0f : o.w;   // Gate.zig:28-30
            store4_sc1(out.at(r0 + k, v), o);
        }
    }
}

// =================================================================== Filter
struct zh_filter { zh_ctx *ctx; uint32_t n; float *l, *b;
                   float *tp_e; uint32_t tp_serial = 0; };    // ZH_PAINT_TOLERANT scratch (filter_tp.hip.h kFilterTpFloats per voice), allocated by the first tolerant paint outside a capture

template <bool ZF, bool CB, bool RB>
__global__ void __launch_bounds__(kSeqBlock) k_filter(float *__restrict__ l_io, float *__restrict__ b_io, uint32_t V, Img out,
                                                      CImg input, uint32_t start, uint32_t end, float l_mul, float b_mul,
                                                      float h_mul, CobP cutoff, CobP res_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    constexpr int NIN = 1 + (CB ? 1 : 0) + (RB ? 1 : 0);
    const float *ins[3] = {input.p, nullptr, nullptr};
    size_t istr[3] = {input.stride, 0, 0};
    if (CB) { ins[1] = cutoff.b.p; istr[1] = cutoff.b.stride; }
    if (RB) { ins[CB ? 2 : 1] = res_p.b.p; istr[CB ? 2 : 1] = res_p.b.stride; }
    FilterLane o;
    o.l = l_io[v]; o.b = b_io[v];
    o.begin(ZH_FILTER_LOW_PASS, CB ? 0.0f : cutoff.c.get(v), RB ? 0.0f : res_p.c.get(v));
    o.l_mul = l_mul; o.b_mul = b_mul; o.h_mul = h_mul;               // the host resolved the type (:98-109)
    frame_loop<8, ZF, NIN>(out.p, v, out.stride, ins, istr, start, end, [&](uint32_t, const float (&x)[NIN], float &val) ZH_INLINE_LAMBDA {
        val = o.template frame<CB, RB>(x[0], CB ? x[1] : 0.0f, RB ? x[CB ? 2 : 1] : 0.0f);
        return true;
    });
    l_io[v] = o.l; b_io[v] = o.b;
}

// Few voices, constant cutoff and resonance: the frame's work spread over THREE waves per 64 voices (the form of
// composite.hip's k_nice_pc4).  Wave 0 fetches the input rows two tiles ahead and adds the filter's input offset
// (in = x + fcdcoffset, Filter.zig:135: a function of the sample alone); wave 1 runs the state-variable recurrence alone
// (svf_core_mid, 15 VALU instructions per sample) and hands on l and the b of :139; wave 2 redoes :143-144 from them for h
// and the final b (svf_finish), then the output mix (:146), the `+=` and the store, three tiles behind the loader.  One
// barrier per 32-frame tile.  Same operations on the same values as k_filter => same bits.
// What the recurrence wave costs beyond its 15 instructions is its LDS traffic: alone with its barriers it takes 29 us for
// 1,024 frames, 34 us with its tile reads and 44 us with three values written per frame as two-dword instructions -- an LDS
// instruction costs a lone wave about three VALU issues.  Hence tiles of float4 (four frames of a lane side by side: one
// 16-byte instruction per four frames) and two values per frame instead of three.
// CH = frames per tile: 32 (64 KB of LDS: two workgroups per CU = 32,768 voices) or 16 (32 KB: five per CU).
template <bool ZF, uint32_t CH>
__global__ void __launch_bounds__(192) k_filter_pc(float *__restrict__ l_io, float *__restrict__ b_io, uint32_t V, Img out, CImg input,
                                                   uint32_t start, uint32_t end, float l_mul, float b_mul, float h_mul, F32P cutoff, F32P res_p) {
    constexpr uint32_t Q = CH / 4;
    __shared__ float4 in_q[4][Q][64], l_q[2][Q][64], b_q[2][Q][64];
    const uint32_t lane = threadIdx.x & 63, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 loader, 1 filter, 2 writer
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < V;
    const uint32_t vc = live ? v : V - 1;                             // lanes past the last voice repeat voice V-1 (and store nothing)
    const uint32_t n = end - start, nchunks = (n + CH - 1) / CH;
    const uint32_t voff = vc * 4u, orow = (uint32_t)out.stride * 4u, irow = (uint32_t)input.stride * 4u;
    FilterLane o;
    o.l = l_io[vc]; o.b = b_io[vc];
    o.begin(ZH_FILTER_LOW_PASS, cutoff.get(vc), res_p.get(vc));        // cut / res (:114, :118); the mix coefficients come from the host
    auto frames = [&](uint32_t c) ZH_INLINE_LAMBDA { return c < nchunks ? min(CH, n - c * CH) : 0u; };
    // frame k of this lane inside a tile of float4 (the scalar path of a partial last tile)
    auto at = [&](float4 (*t)[64], uint32_t k) ZH_INLINE_LAMBDA -> float & { return reinterpret_cast<float *>(&t[k >> 2][lane])[k & 3]; };
    // Step c: the loader publishes tile c (its rows were requested two steps earlier); the filter wave computes tile c - 2
    // out of registers while it fetches tile c - 1 from LDS for the next step; the writer finishes and stores tile c - 3.
    // Every role runs its own copy of the step loop (the same number of barriers in each): as branches of one loop body the
    // roles' register arrays were merged at the join behind an `s_waitcnt vmcnt(0)`.  Two steps per iteration, so that the
    // arrays that alternate between steps keep their registers.
    const uint32_t last = nchunks + 2;
    if (role == 0) {
        float xa[CH], xb[CH];                                         // rows of the tiles c (even / odd), then c + 2
        auto request = [&](uint32_t c, float (&x)[CH]) ZH_INLINE_LAMBDA {
            if (frames(c) == CH) {
                const zh_rsrc_t ri = zrow_rsrc(input.p, input.stride, start + c * CH);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) x[k] = zrow_load<1>(ri, voff, k * irow);
            }
        };
        auto publish = [&](uint32_t c, float (&x)[CH]) ZH_INLINE_LAMBDA {
            const uint32_t nf = frames(c);
            float4 (*t)[64] = in_q[c & 3];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++)
                    t[q][lane] = make_float4(x[4 * q] + kSvfDcOffset, x[4 * q + 1] + kSvfDcOffset, x[4 * q + 2] + kSvfDcOffset, x[4 * q + 3] + kSvfDcOffset);
            } else {
                const zh_rsrc_t ri = zrow_rsrc(input.p, input.stride, start + c * CH);
                for (uint32_t k = 0; k < nf; k++) at(t, k) = zrow_load<1>(ri, voff, k * irow) + kSvfDcOffset;
            }
            request(c + 2, x);
        };
        request(0, xa); request(1, xb);
        for (uint32_t c = 0; c <= last; c += 2) {
            if (c < nchunks) publish(c, xa);
            __syncthreads();
            if (c + 1 <= last) {
                if (c + 1 < nchunks) publish(c + 1, xb);
                __syncthreads();
            }
        }
    } else if (role == 1) {
        float4 fa[Q], fb[Q];                                          // the tile in hand / the next one
#pragma unroll
        for (uint32_t q = 0; q < Q; q++) fa[q] = fb[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto step = [&](uint32_t c, float4 (&cur)[Q], float4 (&nxt)[Q]) ZH_INLINE_LAMBDA {
            if (c == 0 || c > nchunks + 1) return;
            const float4 (*tn)[64] = in_q[(c - 1) & 3];              // (complete only if that tile is a whole one: otherwise unused)
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) nxt[q] = tn[q][lane];
            if (c == 1) return;
            const uint32_t d = c - 2, nf = frames(d);
            float4 (*tl)[64] = l_q[d & 1], (*tb)[64] = b_q[d & 1];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    const SvfMid m0 = svf_core_mid(o.l, o.b, cur[q].x, o.cut, o.res);
                    const SvfMid m1 = svf_core_mid(o.l, o.b, cur[q].y, o.cut, o.res);
                    const SvfMid m2 = svf_core_mid(o.l, o.b, cur[q].z, o.cut, o.res);
                    const SvfMid m3 = svf_core_mid(o.l, o.b, cur[q].w, o.cut, o.res);
                    tl[q][lane] = make_float4(m0.l, m1.l, m2.l, m3.l);
                    tb[q][lane] = make_float4(m0.b1, m1.b1, m2.b1, m3.b1);
                }
            } else {                                                  // (the last tile: the loader has stopped, its buffer stays)
                float4 (*ti)[64] = in_q[d & 3];
                for (uint32_t k = 0; k < nf; k++) {
                    const SvfMid m = svf_core_mid(o.l, o.b, at(ti, k), o.cut, o.res);
                    at(tl, k) = m.l; at(tb, k) = m.b1;
                }
            }
        };
        for (uint32_t c = 0; c <= last; c += 2) {
            step(c, fa, fb);
            __syncthreads();
            if (c + 1 <= last) {
                step(c + 1, fb, fa);
                __syncthreads();
            }
        }
    } else {
        float bn[CH];                                                 // the output rows of the tile after the one in hand
        for (uint32_t c = 0; c <= last; c++) {
            if (c > 2) {
                const uint32_t d = c - 3, nf = frames(d);
                const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + d * CH);
                float4 (*ti)[64] = in_q[d & 3], (*tl)[64] = l_q[d & 1], (*tb)[64] = b_q[d & 1];
                auto one = [&](uint32_t k, float in, float l, float b1, float base) ZH_INLINE_LAMBDA {
                    const SvfOut sv = svf_finish(l, b1, in, o.cut, o.res);
                    const float val = sv.l * l_mul + sv.b * b_mul + sv.h * h_mul;    // :146
                    if (live) zrow_store<1>(ro, voff, k * orow, base + val);
                };
                if (nf == CH) {
                    float4 xi[Q], xl[Q], xb[Q];
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) { xi[q] = ti[q][lane]; xl[q] = tl[q][lane]; xb[q] = tb[q][lane]; }
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) {
                        one(4 * q, xi[q].x, xl[q].x, xb[q].x, ZF ? 0.0f : bn[4 * q]);
                        one(4 * q + 1, xi[q].y, xl[q].y, xb[q].y, ZF ? 0.0f : bn[4 * q + 1]);
                        one(4 * q + 2, xi[q].z, xl[q].z, xb[q].z, ZF ? 0.0f : bn[4 * q + 2]);
                        one(4 * q + 3, xi[q].w, xl[q].w, xb[q].w, ZF ? 0.0f : bn[4 * q + 3]);
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++) one(k, at(ti, k), at(tl, k), at(tb, k), ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
                }
            }
            if (!ZF && c >= 2 && frames(c - 2) == CH) {               // the output rows of the tile written at the next step
                const zh_rsrc_t rn = zrow_rsrc(out.p, out.stride, start + (c - 2) * CH);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) bn[k] = zrow_load<1>(rn, voff, k * orow);
            }
            __syncthreads();
        }
    }
    if (live && role == 1) { l_io[v] = o.l; b_io[v] = o.b; }
}

// k_filter_pc with the cutoff and / or the resonance as control images (a filter sweep: Filter.zig:120-129 clamps them per frame): the
// loader fetches their rows beside the input's, clamps, and publishes them as tiles of their own; the recurrence wave takes a frame's
// cut / res from registers like its input (one more 16-byte LDS read per four frames and image), the writer reads the same tiles
// three steps later (four-deep rings, like in_q).  The one-wave walk this replaces took 95.5 us per 1,024 frames at 4,096 voices
// (the exact form of the table's "Filter low-pass, cutoff image" row).  CH = 32 with one image (96 KB of LDS), 16 with both (64 KB).
template <bool ZF, uint32_t CH, bool CB, bool RB>
__global__ void __launch_bounds__(192) k_filter_pc_ctl(float *__restrict__ l_io, float *__restrict__ b_io, uint32_t V, Img out, CImg input,
                                                       uint32_t start, uint32_t end, float l_mul, float b_mul, float h_mul, CobP cutoff, CobP res_p) {
    static_assert(CB || RB, "constant cutoff and resonance: k_filter_pc");
    constexpr uint32_t Q = CH / 4;
    __shared__ float4 in_q[4][Q][64], cut_q[CB ? 4 : 1][Q][64], res_q[RB ? 4 : 1][Q][64], l_q[2][Q][64], b_q[2][Q][64];
    const uint32_t lane = threadIdx.x & 63, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 loader, 1 filter, 2 writer
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < V;
    const uint32_t vc = live ? v : V - 1;
    const uint32_t n = end - start, nchunks = (n + CH - 1) / CH;
    const uint32_t voff = vc * 4u, orow = (uint32_t)out.stride * 4u, irow = (uint32_t)input.stride * 4u;
    const uint32_t crow = CB ? (uint32_t)cutoff.b.stride * 4u : 0u, rrow = RB ? (uint32_t)res_p.b.stride * 4u : 0u;
    FilterLane o;
    o.l = l_io[vc]; o.b = b_io[vc];
    o.begin(ZH_FILTER_LOW_PASS, CB ? 0.0f : cutoff.c.get(vc), RB ? 0.0f : res_p.c.get(vc));   // the constant one's cut / res (:114, :118)
    auto clampcut = [](float x) ZH_INLINE_LAMBDA { return zclampf(x, 0.0f, 1.0f); };            // :126
    auto clampres = [](float x) ZH_INLINE_LAMBDA { return 1.0f - zclampf(x, 0.0f, 1.0f); };     // :128
    auto frames = [&](uint32_t c) ZH_INLINE_LAMBDA { return c < nchunks ? min(CH, n - c * CH) : 0u; };
    auto at = [&](float4 (*t)[64], uint32_t k) ZH_INLINE_LAMBDA -> float & { return reinterpret_cast<float *>(&t[k >> 2][lane])[k & 3]; };
    const uint32_t last = nchunks + 2;
    if (role == 0) {
        float xa[CH], xb[CH], ca[CB ? CH : 1], cb_[CB ? CH : 1], ra[RB ? CH : 1], rb_[RB ? CH : 1];
        auto request = [&](uint32_t c, float (&x)[CH], float (&cu)[CB ? CH : 1], float (&re)[RB ? CH : 1]) ZH_INLINE_LAMBDA {
            if (frames(c) == CH) {
                const zh_rsrc_t ri = zrow_rsrc(input.p, input.stride, start + c * CH);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) x[k] = zrow_load<1>(ri, voff, k * irow);
                if constexpr (CB) {
                    const zh_rsrc_t rc = zrow_rsrc(cutoff.b.p, cutoff.b.stride, start + c * CH);
#pragma unroll
                    for (uint32_t k = 0; k < CH; k++) cu[k] = zrow_load<1>(rc, voff, k * crow);
                }
                if constexpr (RB) {
                    const zh_rsrc_t rr = zrow_rsrc(res_p.b.p, res_p.b.stride, start + c * CH);
#pragma unroll
                    for (uint32_t k = 0; k < CH; k++) re[k] = zrow_load<1>(rr, voff, k * rrow);
                }
            }
        };
        auto publish = [&](uint32_t c, float (&x)[CH], float (&cu)[CB ? CH : 1], float (&re)[RB ? CH : 1]) ZH_INLINE_LAMBDA {
            const uint32_t nf = frames(c);
            float4 (*t)[64] = in_q[c & 3];
            float4 (*tc)[64] = cut_q[CB ? (c & 3) : 0], (*tr)[64] = res_q[RB ? (c & 3) : 0];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    t[q][lane] = make_float4(x[4 * q] + kSvfDcOffset, x[4 * q + 1] + kSvfDcOffset, x[4 * q + 2] + kSvfDcOffset, x[4 * q + 3] + kSvfDcOffset);
                    if constexpr (CB) tc[q][lane] = make_float4(clampcut(cu[4 * q]), clampcut(cu[4 * q + 1]), clampcut(cu[4 * q + 2]), clampcut(cu[4 * q + 3]));
                    if constexpr (RB) tr[q][lane] = make_float4(clampres(re[4 * q]), clampres(re[4 * q + 1]), clampres(re[4 * q + 2]), clampres(re[4 * q + 3]));
                }
            } else {
                const zh_rsrc_t ri = zrow_rsrc(input.p, input.stride, start + c * CH);
                for (uint32_t k = 0; k < nf; k++) {
                    at(t, k) = zrow_load<1>(ri, voff, k * irow) + kSvfDcOffset;
                    if constexpr (CB) at(tc, k) = clampcut(zrow_load<1>(zrow_rsrc(cutoff.b.p, cutoff.b.stride, start + c * CH), voff, k * crow));
                    if constexpr (RB) at(tr, k) = clampres(zrow_load<1>(zrow_rsrc(res_p.b.p, res_p.b.stride, start + c * CH), voff, k * rrow));
                }
            }
            request(c + 2, x, cu, re);
        };
        request(0, xa, ca, ra); request(1, xb, cb_, rb_);
        for (uint32_t c = 0; c <= last; c += 2) {
            if (c < nchunks) publish(c, xa, ca, ra);
            __syncthreads();
            if (c + 1 <= last) {
                if (c + 1 < nchunks) publish(c + 1, xb, cb_, rb_);
                __syncthreads();
            }
        }
    } else if (role == 1) {
        float4 fa[Q], fb[Q], ca[CB ? Q : 1], cb_[CB ? Q : 1], ra[RB ? Q : 1], rb_[RB ? Q : 1];     // the tile in hand / the next one
#pragma unroll
        for (uint32_t q = 0; q < Q; q++) fa[q] = fb[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t q = 0; q < (CB ? Q : 1); q++) ca[q] = cb_[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t q = 0; q < (RB ? Q : 1); q++) ra[q] = rb_[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto step = [&](uint32_t c, float4 (&cur)[Q], float4 (&nxt)[Q], float4 (&ccur)[CB ? Q : 1], float4 (&cnxt)[CB ? Q : 1],
                        float4 (&rcur)[RB ? Q : 1], float4 (&rnxt)[RB ? Q : 1]) ZH_INLINE_LAMBDA {
            if (c == 0 || c > nchunks + 1) return;
            const float4 (*tn)[64] = in_q[(c - 1) & 3];              // (complete only if that tile is a whole one: otherwise unused)
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) nxt[q] = tn[q][lane];
            if constexpr (CB) {
                const float4 (*tcn)[64] = cut_q[(c - 1) & 3];
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) cnxt[q] = tcn[q][lane];
            }
            if constexpr (RB) {
                const float4 (*trn)[64] = res_q[(c - 1) & 3];
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) rnxt[q] = trn[q][lane];
            }
            if (c == 1) return;
            const uint32_t d = c - 2, nf = frames(d);
            float4 (*tl)[64] = l_q[d & 1], (*tb)[64] = b_q[d & 1];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    const SvfMid m0 = svf_core_mid(o.l, o.b, cur[q].x, CB ? ccur[q].x : o.cut, RB ? rcur[q].x : o.res);
                    const SvfMid m1 = svf_core_mid(o.l, o.b, cur[q].y, CB ? ccur[q].y : o.cut, RB ? rcur[q].y : o.res);
                    const SvfMid m2 = svf_core_mid(o.l, o.b, cur[q].z, CB ? ccur[q].z : o.cut, RB ? rcur[q].z : o.res);
                    const SvfMid m3 = svf_core_mid(o.l, o.b, cur[q].w, CB ? ccur[q].w : o.cut, RB ? rcur[q].w : o.res);
                    tl[q][lane] = make_float4(m0.l, m1.l, m2.l, m3.l);
                    tb[q][lane] = make_float4(m0.b1, m1.b1, m2.b1, m3.b1);
                }
            } else {                                                  // (the last tile: the loader has stopped, its buffers stay)
                float4 (*ti)[64] = in_q[d & 3];
                float4 (*tc)[64] = cut_q[CB ? (d & 3) : 0], (*tr)[64] = res_q[RB ? (d & 3) : 0];
                for (uint32_t k = 0; k < nf; k++) {
                    const SvfMid m = svf_core_mid(o.l, o.b, at(ti, k), CB ? at(tc, k) : o.cut, RB ? at(tr, k) : o.res);
                    at(tl, k) = m.l; at(tb, k) = m.b1;
                }
            }
        };
        for (uint32_t c = 0; c <= last; c += 2) {
            step(c, fa, fb, ca, cb_, ra, rb_);
            __syncthreads();
            if (c + 1 <= last) {
                step(c + 1, fb, fa, cb_, ca, rb_, ra);
                __syncthreads();
            }
        }
    } else {
        float bn[CH];                                                 // the output rows of the tile after the one in hand
        for (uint32_t c = 0; c <= last; c++) {
            if (c > 2) {
                const uint32_t d = c - 3, nf = frames(d);
                const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + d * CH);
                float4 (*ti)[64] = in_q[d & 3], (*tl)[64] = l_q[d & 1], (*tb)[64] = b_q[d & 1];
                float4 (*tc)[64] = cut_q[CB ? (d & 3) : 0], (*tr)[64] = res_q[RB ? (d & 3) : 0];
                auto one = [&](uint32_t k, float in, float l, float b1, float cu, float re, float base) ZH_INLINE_LAMBDA {
                    const SvfOut sv = svf_finish(l, b1, in, cu, re);
                    const float val = sv.l * l_mul + sv.b * b_mul + sv.h * h_mul;    // :146
                    if (live) zrow_store<1>(ro, voff, k * orow, base + val);
                };
                if (nf == CH) {
                    float4 xi[Q], xl[Q], xb[Q], xc[CB ? Q : 1], xr[RB ? Q : 1];
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) {
                        xi[q] = ti[q][lane]; xl[q] = tl[q][lane]; xb[q] = tb[q][lane];
                        if constexpr (CB) xc[q] = tc[q][lane];
                        if constexpr (RB) xr[q] = tr[q][lane];
                    }
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) {
                        one(4 * q, xi[q].x, xl[q].x, xb[q].x, CB ? xc[CB ? q : 0].x : o.cut, RB ? xr[RB ? q : 0].x : o.res, ZF ? 0.0f : bn[4 * q]);
                        one(4 * q + 1, xi[q].y, xl[q].y, xb[q].y, CB ? xc[CB ? q : 0].y : o.cut, RB ? xr[RB ? q : 0].y : o.res, ZF ? 0.0f : bn[4 * q + 1]);
                        one(4 * q + 2, xi[q].z, xl[q].z, xb[q].z, CB ? xc[CB ? q : 0].z : o.cut, RB ? xr[RB ? q : 0].z : o.res, ZF ? 0.0f : bn[4 * q + 2]);
                        one(4 * q + 3, xi[q].w, xl[q].w, xb[q].w, CB ? xc[CB ? q : 0].w : o.cut, RB ? xr[RB ? q : 0].w : o.res, ZF ? 0.0f : bn[4 * q + 3]);
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++)
                        one(k, at(ti, k), at(tl, k), at(tb, k), CB ? at(tc, k) : o.cut, RB ? at(tr, k) : o.res, ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
                }
            }
            if (!ZF && c >= 2 && frames(c - 2) == CH) {               // the output rows of the tile written at the next step
                const zh_rsrc_t rn = zrow_rsrc(out.p, out.stride, start + (c - 2) * CH);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) bn[k] = zrow_load<1>(rn, voff, k * orow);
            }
            __syncthreads();
        }
    }
    if (live && role == 1) { l_io[v] = o.l; b_io[v] = o.b; }
}

__global__ void k_cutoff_from_frequency(uint32_t n, float *__restrict__ out, const float *__restrict__ freq, float sample_rate) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = zcutoff_from_frequency(freq[i], sample_rate);
}

__global__ void k_pow(uint32_t n, float *__restrict__ out, const float *__restrict__ x, const float *__restrict__ y) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = zpowf_pos(x[i], y[i]);
}

template <int FN>   // 0 sin, 1 cos, 2 atan
__global__ void k_sincos(uint32_t n, float *__restrict__ out, const float *__restrict__ x) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = FN == 2 ? zatanf(x[i]) : (FN == 1 ? zcosf(x[i]) : zsinf(x[i]));
}

// =================================================================== Sampler
// `t` double-buffered (cnt[] holds the f32 bits), like zh_sineosc: the frame-range kernel's ranges read the start position
// while the last range publishes the end position
struct zh_sampler : zh_flipper {
    float *t() const { return reinterpret_cast<float *>(cnt[cur]); }
};

struct SampleP {
    const uint8_t *data;
    uint64_t data_len;
    uint32_t num_channels, sample_rate_in, format, channel, loop, whole;
    int32_t num_samples;        // data.len / bytes_per_sample / num_channels (Sampler.zig:42)
    double inv_num_samples;     // 1.0 / num_samples (0 when there are none): sampler_mod
};

// Sampler.zig:23-33.  FMT is a compile-time format (the kernel is instantiated per format and loop flag): with the
// format switch, the loop test and the bounds test as branches inside the frame loop every frame was its own basic
// block and paid the full latency of its PCM gathers; as straight-line code the loads of a chunk's 8 frames overlap.
// `whole`: the PCM base is aligned to the sample size, so a sample is one load instead of byte_count byte gathers.
template <int N> struct zint { static constexpr int value = N; };
constexpr int kSampleEmpty = -1;                                     // no samples: every read is 0 (and nothing is loaded)
template <int FMT>
__device__ __forceinline__ float sampler_decode(const uint8_t *data, size_t i, bool whole) {
    if constexpr (FMT == kSampleEmpty) return 0.0f;
    else if constexpr (FMT == ZH_SAMPLE_U8) return ((float)data[i] - 127.5f) / 127.5f;
    else {
        constexpr int byte_count = FMT + 1;
        const uint8_t *p = data + i * byte_count;
        int32_t sval;
        if constexpr (byte_count == 2) sval = whole ? (int16_t)*reinterpret_cast<const uint16_t *>(p) : (int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8));
        else if constexpr (byte_count == 3) {
            const uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
            sval = (int32_t)(u << 8) >> 8;
        } else sval = whole ? (int32_t)*reinterpret_cast<const uint32_t *>(p)
                            : (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
        // `sval / max` with max = 2^(bits-1): dividing by a power of two and multiplying by its reciprocal are the
        // same IEEE operation (both exact up to the one rounding of the result), so no divide sequence is needed
        const float inv_max = 1.0f / (float)(1u << (byte_count * 8 - 1));
        return (float)sval * inv_max;
    }
}

// @mod(index, n) for n > 0 (floored, Sampler.zig:43): the quotient from one f64 multiply by 1/n (any i32 over any
// positive i32 is far inside f64's 53 bits; the estimate is off by at most one), then two exact corrections -- a
// dozen instructions instead of the ~40 of a 32-bit integer remainder.
__device__ __forceinline__ int32_t sampler_mod(int32_t index, int32_t n, double inv_n) {
    const int32_t q = (int32_t)floor((double)index * inv_n);
    int32_t r = (int32_t)((uint32_t)index - (uint32_t)q * (uint32_t)n);
    r = r < 0 ? r + n : r;
    r = r >= n ? r - n : r;
    return r;
}

// Sampler.zig:35-58 (num_samples == 0 with loop: division by zero in the reference; DEFINED as silence = kSampleEmpty)
// the sample at an index that is already resolved (looped: inside [0, num_samples); else tested here)
template <int FMT>
__device__ __forceinline__ float sampler_at(const SampleP &s, int32_t index) {
    if constexpr (FMT == kSampleEmpty) return 0.0f;
    const bool in = index >= 0 && index < s.num_samples;
    const size_t i = (size_t)(in ? index : 0) * s.num_channels + s.channel;
    const float val = sampler_decode<FMT>(s.data, i, s.whole != 0);
    return in ? val : 0.0f;
}
template <int FMT, bool LOOP>
__device__ __forceinline__ float sampler_get_sample(const SampleP &s, int32_t index1) {
    if constexpr (FMT == kSampleEmpty) return 0.0f;
    return sampler_at<FMT>(s, LOOP ? sampler_mod(index1, s.num_samples, s.inv_num_samples) : index1);   // @mod: floored
}

// One kernel, two launch shapes.  Sequential: grid.y = 1, ch = the whole span, t_in == t_out.  Few voices: grid.y frame
// ranges of `ch` frames at once, one wave per (64 voices, range) -- the play position that reaches frame f0 is the start
// position after f0 - start additions of `ratio`, each rounded to f32, so a range first REPLAYS those additions (the
// no-resampling path needs none: its index is t0 + frame), then paints its frames; the range that ends the span
// publishes the position (t_out is the other half of the double buffer).
template <bool ZF, int FMT, bool LOOP>
__global__ void __launch_bounds__(kSeqBlock) k_sampler(const float *__restrict__ t_in, float *__restrict__ t_out, uint32_t V, Img out,
                                                       uint32_t start, uint32_t end, uint32_t ch, SampleP s, F32P out_rate, BoolP nic) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    const bool last = f1 == end;
    float t = t_in[v];
    if (nic.get(v)) t = 0.0f;                                         // Sampler.zig:91-93
    const uint32_t len = end - start;
    const float ratio = (float)s.sample_rate_in / out_rate.get(v);    // :97
    const float *const *no_in = nullptr;
    if (ratio < 0.0f && !LOOP) {                                      // :99-102 (t keeps the reset)
        if (ZF) zero_column(out.p + v, out.stride, f0, f1);
        if (last) t_out[v] = t;
        return;
    }
    if (ratio > 0.9999f && ratio < 1.0001f) {                         // :105-114 no resampling
        const int32_t t0 = zf32_to_i32(roundf(t));
        frame_loop<8, ZF, 0>(out.p, v, out.stride, no_in, nullptr, f0, f1, [&](uint32_t i, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
            val = sampler_get_sample<FMT, LOOP>(s, (int32_t)((uint32_t)t0 + (i - start)));
            return true;
        });
        t += (float)len;
    } else {                                                          // :116-130 linear resampling
        uint32_t i = start;
        for (; i + 8 <= f0; i += 8) {                                 // the earlier frames' `self.t += ratio` (:129), add by add
#pragma unroll
            for (int k = 0; k < 8; k++) t += ratio;
        }
        for (; i < f0; i++) t += ratio;
        // looped: @mod(t0 + 1, n) is @mod(t0, n) + 1, wrapped once -- except where t0 + 1 itself wraps (t0 = INT32_MAX: the
        // conversion's saturation value), whose remainder is the same for every frame
        const int32_t r_wrap = (LOOP && FMT != kSampleEmpty) ? sampler_mod(INT32_MIN, s.num_samples, s.inv_num_samples) : 0;
        auto general = [&](float &val) ZH_INLINE_LAMBDA {
            const int32_t t0 = zf32_to_i32(floorf(t));
            const int32_t t1 = (int32_t)((uint32_t)t0 + 1u);
            const float tfrac = (float)t1 - t;                        // :121
            float s0, s1;
            if constexpr (LOOP && FMT != kSampleEmpty) {
                const int32_t r0 = sampler_mod(t0, s.num_samples, s.inv_num_samples);
                int32_t r1 = r0 + 1 == s.num_samples ? 0 : r0 + 1;
                r1 = t0 == INT32_MAX ? r_wrap : r1;
                s0 = sampler_at<FMT>(s, r0);
                s1 = sampler_at<FMT>(s, r1);
            } else {
                s0 = sampler_get_sample<FMT, LOOP>(s, t0);
                s1 = sampler_get_sample<FMT, LOOP>(s, t1);
            }
            val = s0 * (1.0f - tfrac) + s1 * tfrac;
            t += ratio;
        };
        // Frames of 1, 2 or 4 bytes (u8 x 1 / 2 / 4 channels, aligned s16 x 1 / 2, aligned s32 x 1): the two samples of an
        // interpolation sit in neighbouring frames, so ONE load of two frames fetches both (the gathers, not the arithmetic,
        // bound this kernel: a 64-lane gather is 16 address cycles of the CU's texture path, two of them per sample).  The
        // pair is read at frame rb = clamp(r0, 0, n - 2); d = r0 - rb says which halves the frame wants: 0 -> (lo, hi);
        // 1 -> r0 is the last frame: (hi, looped ? frame 0 : nothing); -1 -> r0 = -1: (nothing, lo); else both indices are
        // outside the data.  Chunks in which a play position could reach the float -> i32 conversion's saturation (|t| near
        // 2^31: the looped path's r_wrap case) or in which the looped index cannot be stepped from the previous frame's
        // (|ratio| + 1 >= n) take the general body.
        constexpr int kBps = FMT == kSampleEmpty ? 0 : FMT + 1;
        const uint32_t fb = s.num_channels * (uint32_t)kBps;          // bytes per frame
        const bool pair_ok = kBps != 0 && kBps != 3 && (fb == 1 || fb == 2 || fb == 4) && s.num_samples >= 2 && (kBps == 1 || s.whole);
        auto paired = [&](auto fb_tag) ZH_INLINE_LAMBDA {
            constexpr int FB = decltype(fb_tag)::value;
            const int32_t n = s.num_samples;
            const uint32_t sh = s.channel * 8u * (uint32_t)kBps;      // the channel's bit offset inside a frame
            const float s_first = sampler_at<FMT>(s, 0);
            const bool step_ok = __builtin_fabsf(ratio) + 1.0f < (float)n;
            bool have_prev = false;
            int32_t t0_prev = 0, r0_prev = 0;
            frame_loop_gen<8, ZF>(out.p, v, out.stride, f0, f1,
                [&](uint32_t) ZH_INLINE_LAMBDA {
                    have_prev = false;
                    const bool ok = step_ok && __builtin_fabsf(t) + 9.0f * __builtin_fabsf(ratio) < 2.0e9f;   // (a NaN fails)
                    return __builtin_amdgcn_ballot_w64(!ok) == 0;
                },
                [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
                    const int32_t t0 = zf32_to_i32(floorf(t));
                    const int32_t t1 = (int32_t)((uint32_t)t0 + 1u);
                    const float tfrac = (float)t1 - t;                // :121
                    int32_t r0 = t0;
                    if constexpr (LOOP) {
                        if (have_prev) {                              // @mod(t0, n) from the previous frame's: |t0 - t0_prev| < n
                            r0 = r0_prev + (t0 - t0_prev);
                            r0 = r0 >= n ? r0 - n : r0;
                            r0 = r0 < 0 ? r0 + n : r0;
                        } else {
                            r0 = sampler_mod(t0, n, s.inv_num_samples);
                        }
                        t0_prev = t0; r0_prev = r0; have_prev = true;
                    }
                    const int32_t rb = min(max(r0, 0), n - 2);
                    const int32_t d = r0 - rb;
                    uint32_t w0, w1;                                  // frames rb and rb + 1
                    if constexpr (FB == 1) {
                        uint16_t w; __builtin_memcpy(&w, s.data + (size_t)rb, 2);
                        w0 = w & 0xffu; w1 = w >> 8;
                    } else if constexpr (FB == 2) {
                        uint32_t w; __builtin_memcpy(&w, s.data + (size_t)rb * 2, 4);
                        w0 = w & 0xffffu; w1 = w >> 16;
                    } else {
                        uint2 w; __builtin_memcpy(&w, s.data + (size_t)rb * 4, 8);
                        w0 = w.x; w1 = w.y;
                    }
                    w0 >>= sh; w1 >>= sh;
                    float lo, hi;
                    if constexpr (FMT == ZH_SAMPLE_U8) {
                        lo = ((float)(w0 & 0xffu) - 127.5f) / 127.5f;
                        hi = ((float)(w1 & 0xffu) - 127.5f) / 127.5f;
                    } else if constexpr (FMT == ZH_SAMPLE_S16_LSB) {
                        lo = (float)(int16_t)(w0 & 0xffffu) * (1.0f / 32768.0f);
                        hi = (float)(int16_t)(w1 & 0xffffu) * (1.0f / 32768.0f);
                    } else {
                        lo = (float)(int32_t)w0 * (1.0f / 2147483648.0f);
                        hi = (float)(int32_t)w1 * (1.0f / 2147483648.0f);
                    }
                    const float s0 = d == 0 ? lo : (d == 1 ? hi : 0.0f);
                    const float s1 = d == 0 ? hi : (d == 1 ? (LOOP ? s_first : 0.0f) : (d == -1 ? lo : 0.0f));
                    val = s0 * (1.0f - tfrac) + s1 * tfrac;
                    t += ratio;
                    return true;
                },
                [&](uint32_t, float &val) ZH_INLINE_LAMBDA { general(val); return true; });
        };
        if (pair_ok && fb == 1) paired(zint<1>{});
        else if (pair_ok && fb == 2) paired(zint<2>{});
        else if (pair_ok) paired(zint<4>{});
        else {
            frame_loop<8, ZF, 0>(out.p, v, out.stride, no_in, nullptr, f0, f1, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
                general(val);
                return true;
            });
        }
    }
    if (!last) return;
    // :133-135: compared against data.len in BYTES (reference quirk, kept)
    if (t >= (float)s.data_len && LOOP) t -= (float)s.data_len;
    t_out[v] = t;
}

// =================================================================== Decimator
// state double-buffered like zh_envelope's: two words per voice, [dval n][dcount n]
struct zh_decimator : zh_flipper {
    float *dval(int b) const { return reinterpret_cast<float *>(cnt[b]); }
    float *dcount(int b) const { return reinterpret_cast<float *>(cnt[b] + n); }
};

template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_decimator(float *__restrict__ dval_io, float *__restrict__ dcount_io, uint32_t V,
                                                         Img out, CImg input, uint32_t start, uint32_t end,
                                                         float sample_rate, F32P fake_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    DecimatorLane o;
    o.dval = dval_io[v]; o.dcount = dcount_io[v];
    o.begin(sample_rate, fake_p.get(v));
    const float *ins[1] = {input.p};
    const size_t istr[1] = {input.stride};
    frame_loop<8, ZF, 1>(out.p, v, out.stride, ins, istr, start, end,
                         [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA { return o.frame(x[0], val); });
    o.end();
    dval_io[v] = o.dval; dcount_io[v] = o.dcount;
}

// A span as frame ranges (grid.y): the walk from frame to frame is the fractional counter alone (four instructions,
// no loads) -- a range replays it for the frames before it, remembering the latest frame that sampled, fetches that one
// input sample, and paints its own frames like k_decimator.  The range that ends the span writes the end state to `next`,
// the other half of the module's double buffer (the host flips: zh_flipper).
template <bool ZF>
__global__ void __launch_bounds__(64) k_decimator_ranges(const float *__restrict__ dval_in, const float *__restrict__ dcount_in,
                                                         float *__restrict__ next, uint32_t V, Img out, CImg input, uint32_t start,
                                                         uint32_t end, uint32_t ch, float sample_rate, F32P fake_p) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    DecimatorLane o;
    o.dval = dval_in[v]; o.dcount = dcount_in[v];
    o.begin(sample_rate, fake_p.get(v));
    if (f0 > start) {
        const float dcount0 = o.dcount;
        uint32_t last = 0xFFFFFFFFu;
        uint32_t i = start;
        if (__builtin_amdgcn_ballot_w64(!o.walk_is_plain()) == 0) {   // (wave-uniform) every state the module produces itself
            uint32_t idx = start;
            for (; i + 8 <= f0; i += 8) {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) o.step_plain(idx, last);
            }
            for (; i < f0; i++) o.step_plain(idx, last);
        } else {
            for (; i + 8 <= f0; i += 8) {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) o.step(i + q, last);
            }
            for (; i < f0; i++) o.step(i, last);
        }
        const float held = *input.at(last == 0xFFFFFFFFu ? start : last, v);
        if (o.mode == 1) { if (last != 0xFFFFFFFFu) o.dval = held; } else o.dcount = dcount0;
    }
    const float *ins[1] = {input.p};
    const size_t istr[1] = {input.stride};
    frame_loop<8, ZF, 1>(out.p, v, out.stride, ins, istr, f0, f1,
                         [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA { return o.frame(x[0], val); });
    if (f1 == end) {
        o.end();
        next[v] = o.dval; next[(size_t)V + v] = o.dcount;
    }
}
// =================================================================== Distortion (stateless)
struct zh_distortion { zh_ctx *ctx; uint32_t n; };

// grid: x = 64-voice groups, y = groups of 4 chunks of DIST_FC frames; per chunk each lane
// recomputes its voice's gain1 = pow(2, ingain*8 - 2) (Distortion.zig:41).
#ifndef ZH_DIST_FC
#define ZH_DIST_FC 32
#endif
constexpr uint32_t DIST_FC = ZH_DIST_FC;
template <bool ZF, bool OVERDRIVE>
__global__ void __launch_bounds__(256) k_distortion(uint32_t V, Img out, CImg input, uint32_t start, uint32_t end,
                                                    F32P ingain, F32P outgain, F32P offset) {
    const uint32_t v = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const uint32_t c0 = start + chunk * DIST_FC, c1 = min(c0 + DIST_FC, end);
    if (c0 >= end) return;
    DistortionLane d;
    d.begin(OVERDRIVE ? ZH_DISTORTION_OVERDRIVE : ZH_DISTORTION_CLIP, ingain.get(v), outgain.get(v), offset.get(v));
    float *o = out.at(c0, v);
    const float *in = input.at(c0, v);
    uint32_t i = c0;
    for (; i + 8 <= c1; i += 8, o += 8 * (size_t)out.stride, in += 8 * (size_t)input.stride) {   // 8 frames' loads ahead of their stores
        float x[8], old[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { x[k] = in[(size_t)k * input.stride]; old[k] = ZF ? 0.0f : o[(size_t)k * out.stride]; }
#pragma unroll
        for (int k = 0; k < 8; k++) store_row(o + (size_t)k * out.stride, old[k] + d.frame(x[k]));
    }
    for (; i < c1; i++, o += out.stride, in += input.stride) store_row(o, (ZF ? 0.0f : *o) + d.frame(*in));
}

// Many voices (distortion_rows_min, dispatch.hip): the chunked oscillator's launch shape, like the basics.zig operations take from
// 32,768 voices (basics.hip k_elementwise_chunks) -- four voices per lane, 16-byte loads and write-through stores, a wave = RC consecutive
// rows of a 256-voice column, a workgroup four consecutive chunks.  The per-voice constants (gain1 = pow(2, ingain * 8 - 2), the
// overdrive's outgain / atan(gain1): Distortion.zig:41-45) cost a powf and an atanf: computed ONCE per workgroup, thread t for voice
// base + t, and handed to the lanes through LDS -- per lane and chunk they made this shape slower than one voice per lane
// (profiles/r05/ab_gate4.txt).  Same DistortionLane::begin / frame per voice: same bits.
template <bool ZF, bool OVERDRIVE, int RC>
__global__ void __launch_bounds__(256) k_distortion_chunks(uint32_t V, Img out, CImg input, uint32_t start, uint32_t nframes,
                                                           F32P ingain, F32P outgain, F32P offset) {
    __shared__ __attribute__((aligned(16))) float sk[3][256];
    const uint32_t vbase = blockIdx.x * 256, lane = threadIdx.x & 63;
    {
        const uint32_t sv = vbase + threadIdx.x;
        if (sv < V) {
            DistortionLane d;
            d.begin(OVERDRIVE ? ZH_DISTORTION_OVERDRIVE : ZH_DISTORTION_CLIP, ingain.get(sv), outgain.get(sv), offset.get(sv));
            sk[0][threadIdx.x] = d.gain1; sk[1][threadIdx.x] = d.offs; sk[2][threadIdx.x] = d.gain2;
        }
    }
    __syncthreads();
    const uint32_t v = vbase + lane * 4;
    if (v >= V) return;                                               // V % 4 == 0 on this path
    const zv4f g1 = *reinterpret_cast<const zv4f *>(&sk[0][lane * 4]), of = *reinterpret_cast<const zv4f *>(&sk[1][lane * 4]),
               g2 = *reinterpret_cast<const zv4f *>(&sk[2][lane * 4]);
    DistortionLane d[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { d[j].overdrive = OVERDRIVE; d[j].gain1 = g1[j]; d[j].offs = of[j]; d[j].gain2 = g2[j]; }
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6), r0 = chunk * RC;
    zv4f x[RC], old[RC];
#pragma unroll
    for (int k = 0; k < RC; k++) {
        x[k] = old[k] = zv4f{0, 0, 0, 0};
        if (r0 + k < nframes) {
            x[k] = *reinterpret_cast<const zv4f *>(input.at(start + r0 + k, v));
            if (!ZF) old[k] = *reinterpret_cast<const zv4f *>(out.at(start + r0 + k, v));
        }
    }
#pragma unroll
    for (int k = 0; k < RC; k++) {
        if (r0 + k < nframes) {
            zv4f o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = old[k][j] + d[j].frame(x[k][j]);
            store4_sc1(out.at(start + r0 + k, v), o);
        }
    }
}

// =================================================================== Curve
// state double-buffered like zh_envelope's: four words per voice, [t n][current_song_note n][offset n][next_song_note n]
struct zh_curve_module : zh_flipper {};

// One lane per voice walks the span (grid.y == 1, ch = the span, st_in == st_out), or -- few voices -- the span as grid.y frame
// ranges.  What a paint leaves behind (t, cur, off, next) is decided in begin(), before the first frame; the frame walk only
// carries the running curve span and its accumulator, which begin(r0) sets up for a range's first frame (spans that end
// before it are dropped as they stream by, the running span's accumulator is stepped from its own first frame).  The range
// that ends the span writes the state (into the other half of the double buffer when there are several).
template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_curve(const uint32_t *__restrict__ st_in, uint32_t *__restrict__ st_out, uint32_t V,
                                                     Img out, uint32_t start, uint32_t end, uint32_t ch, float sample_rate, uint32_t function,
                                                     const zh_curve_node *__restrict__ curve, uint32_t n_curve, BoolP nic) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    const size_t N = V;
    CurveLane o;
    CurveTable tb;
    o.t = __builtin_bit_cast(float, st_in[v]); o.cur = st_in[N + v]; o.off = (int32_t)st_in[2 * N + v]; o.next = st_in[3 * N + v];
    o.begin(tb, sample_rate, function, curve, n_curve, end - start, nic.get(v), f0 - start);
    frame_loop_gen<8, ZF>(out.p, v, out.stride, f0, f1, [&](uint32_t i) ZH_INLINE_LAMBDA { return o.quiet(i - start, 8); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { return o.frame_in_span(val); },
                          [&](uint32_t i, float &val) ZH_INLINE_LAMBDA { return o.frame(tb, i - start, val); });
    if (f1 == end) {
        st_out[v] = __builtin_bit_cast(uint32_t, o.t); st_out[N + v] = o.cur; st_out[2 * N + v] = (uint32_t)o.off; st_out[3 * N + v] = o.next;
    }
}

// =================================================================== Cycle
// `t` double-buffered (cnt[] holds the f32 bits), like zh_sineosc
struct zh_cycle : zh_flipper {
    float *t() const { return reinterpret_cast<float *>(cnt[cur]); }
};

// One lane per voice walks the span (grid.y == 1, ch = the span, t_in == t_out), or -- few voices, constant speed -- the span as
// grid.y frame ranges: the walk IS the value (t, t += step, t -= trunc(t)), so a range replays three instructions per earlier
// frame against the frame's eight issue slots with its `+=` and store; the range that ends the span publishes t.
template <bool ZF, bool SB>
__global__ void __launch_bounds__(kSeqBlock) k_cycle(const float *__restrict__ t_in, float *__restrict__ t_out, uint32_t V, Img out, uint32_t start,
                                                     uint32_t end, uint32_t ch, float sample_rate, CobP speed) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    CycleLane o;
    o.t = t_in[v];
    o.begin(sample_rate, SB ? 0.0f : speed.c.get(v));
    if (!SB) {
        uint32_t i = start;
        for (; i + 8 <= f0; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) (void)o.template frame<false>(0.0f);
        }
        for (; i < f0; i++) (void)o.template frame<false>(0.0f);
    }
    const float *ins[1] = {SB ? speed.b.p : nullptr};
    const size_t istr[1] = {speed.b.stride};
    frame_loop<8, ZF, SB ? 1 : 0>(out.p, v, out.stride, ins, istr, f0, f1, [&](uint32_t, const float (&x)[1], float &val) ZH_INLINE_LAMBDA {
        val = o.template frame<SB>(x[0]);
        return true;
    });
    if (f1 == end) t_out[v] = o.t;
}

// =================================================================== Portamento
// state double-buffered like zh_envelope's: three words per voice, [t n][last_value n][start n]
struct zh_portamento : zh_flipper {
    float *f(int b, int k) const { return reinterpret_cast<float *>(cnt[b] + (size_t)k * n); }
};

template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_portamento(float *__restrict__ t_io, float *__restrict__ last_io,
                                                          float *__restrict__ start_io, uint32_t V, Img out, uint32_t start,
                                                          uint32_t end, float sample_rate, uint32_t curve_tag, F32P duration,
                                                          F32P goal_p, BoolP note_on, BoolP prev_note_on, BoolP nic) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    PortamentoLane o;
    o.t = t_io[v]; o.last = last_io[v]; o.st = start_io[v];
    o.begin(sample_rate, curve_tag, duration.get(v), goal_p.get(v), note_on.get(v), prev_note_on.get(v), nic.get(v));
    frame_loop_gen<8, ZF>(out.p, v, out.stride, start, end, [&](uint32_t) ZH_INLINE_LAMBDA { return o.all_flat(); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.goal; return true; },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.frame(); return true; });
    t_io[v] = o.t; last_io[v] = o.last; start_io[v] = o.st;
}

// A span as frame ranges (few voices): a range replays the glide's clock for the frames before it -- nothing for a wave whose
// voices have all arrived, 8 additions per chunk while none can arrive, frame by frame around an arrival -- then paints its
// own frames like k_portamento; the range that ends the span writes the state into the other half of the double buffer.
template <bool ZF>
__global__ void __launch_bounds__(64) k_portamento_ranges(const float *__restrict__ t_in, const float *__restrict__ last_in, const float *__restrict__ start_in,
                                                          float *__restrict__ next, uint32_t V, Img out, uint32_t start, uint32_t end, uint32_t ch,
                                                          float sample_rate, uint32_t curve_tag, F32P duration, F32P goal_p, BoolP note_on,
                                                          BoolP prev_note_on, BoolP nic) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    PortamentoLane o;
    o.t = t_in[v]; o.last = last_in[v]; o.st = start_in[v];
    o.begin(sample_rate, curve_tag, duration.get(v), goal_p.get(v), note_on.get(v), prev_note_on.get(v), nic.get(v));
    if (!o.all_flat()) {
        uint32_t i = start;
        for (; i + 8 <= f0; i += 8) {
            if (o.quiet(8)) o.template skip_quiet<8>();
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) (void)o.frame();
            }
        }
        for (; i < f0; i++) (void)o.frame();
    }
    frame_loop_gen<8, ZF>(out.p, v, out.stride, f0, f1, [&](uint32_t) ZH_INLINE_LAMBDA { return o.all_flat(); },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.goal; return true; },
                          [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = o.frame(); return true; });
    if (f1 == end) { next[v] = o.t; next[(size_t)V + v] = o.last; next[(size_t)2 * V + v] = o.st; }
}

// =================================================================== host side
template <class M> static int paint_check(M *m, uint32_t start, uint32_t end, const zh_buf *outputs) {
    if (!m || !outputs || end < start) return ZH_ERR_INVALID;
    if (!buf_covers(outputs[0], m->n, end)) return ZH_ERR_INVALID;
    return ZH_OK;
}

#define ZH_ZF_LAUNCH(KERNEL, GRID, BLOCK, ...)                                                         \
    do {                                                                                               \
        if (zf) ZH_LAUNCH((KERNEL<true>), GRID, BLOCK, 0, st, __VA_ARGS__);                   \
        else ZH_LAUNCH((KERNEL<false>), GRID, BLOCK, 0, st, __VA_ARGS__);                     \
    } while (0)

static bool curve_ok(const zh_curve &c) { return c.tag <= ZH_CURVE_CUBED; }
static EnvParamsP mk_env_params(const zh_envelope_params *p) {
    return EnvParamsP{p->sample_rate, p->attack.tag, p->decay.tag, p->release.tag, mk_f32(p->attack.duration),
                      mk_f32(p->decay.duration), mk_f32(p->release.duration), mk_f32(p->sustain_volume), mk_bool(p->note_on)};
}

// shared by the modules whose state is one double-buffered 32-bit word per voice (zh_sineosc, zh_sampler)
template <class M> static int flip1_create(zh_ctx *ctx, uint32_t n, M **out) {
    if (!ctx || !out) return ZH_ERR_INVALID;
    M *m = new (std::nothrow) M();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0;
    int rc = dev_alloc(&m->cnt[0], n);
    if (!rc) rc = dev_alloc(&m->cnt[1], n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[0], 0, (size_t)n * 4, ctx->stream);   // init(): 0.0
    if (!rc && n) rc = (int)hipMemsetAsync(m->cnt[1], 0, (size_t)n * 4, ctx->stream);
    if (rc) { (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
template <class M> static int flip1_destroy(M *m) {
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]);
    delete m;
    return ZH_OK;
}

extern "C" {

// ------------------------------------------------------------------ SineOsc
int zh_sineosc_create(zh_ctx *ctx, uint32_t n, zh_sineosc **out) { ZH_GUARD(ctx); return flip1_create(ctx, n, out); }   // init(): t = 0 (SineOsc.zig:18-22)
int zh_sineosc_destroy(zh_sineosc *m) { ZH_GUARD(m ? m->ctx : nullptr); return flip1_destroy(m); }
int zh_sineosc_get_state(zh_sineosc *m, zh_sineosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_download(m->ctx, host, m->t(), (size_t)m->n * 4);
}
int zh_sineosc_set_state(zh_sineosc *m, const zh_sineosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_upload(m->ctx, m->t(), host, (size_t)m->n * 4);
}
int zh_sineosc_paint(zh_sineosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                     zh_bool note_id_changed, const zh_sineosc_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // SineOsc.zig:32-33
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || !cob_ok(p->freq, m->n, end) || !cob_ok(p->phase, m->n, end)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    const bool tol = (flags & ZH_PAINT_TOLERANT) != 0;                               // the sine in f32 (zmath.hip.h zsinf_tol); the phase walk is exact
    hipStream_t st = m->ctx->stream;
    const bool fb = p->freq.tag == ZH_COB_BUFFER, pb = p->phase.tag == ZH_COB_BUFFER;
    Img out = mk_img(outputs[0]);
    CobP f = mk_cob(p->freq), ph = mk_cob(p->phase);
    // few voices: frame ranges at once (k_sineosc_ranges); ZH_SINE_RANGES = number of ranges, 0 = never
    // constant frequency and phase: the replay is two adds a frame, ranges pay up to 4 waves per SIMD at any voice count
    // (24,576 / 32,768 / 65,536 / 131,072 voices: 162 -> 58, 165 -> 68, 174 -> 141, 266 -> 246 us); with a control image the
    // replay re-reads the image (32,768 voices: 166 -> 90 us with 4 ranges, no gain from 65,536 on)
    const bool aliased = cob_aliases(p->freq, outputs[0]) || cob_aliases(p->phase, outputs[0]);
    const uint32_t ch = end > start && !aliased ? zh_range_frames(m->n, end - start, ZF_SINE_RANGES, fb || pb ? 2048 : 4096, fb || pb ? 65536 : 1u << 20) : 0;
    if (ch) {
        const float *t_in = m->t();
        float *t_out = reinterpret_cast<float *>(m->cnt[m->cur ^ 1]);
        const dim3 grid((m->n + 63) / 64, (end - start + ch - 1) / ch);
#define ZH_SINE_R2(FB, PB, TOL)                                                                                     \
    do {                                                                                                            \
        if (zf) ZH_LAUNCH((k_sineosc_ranges<true, FB, PB, TOL>), grid, dim3(64), 0, st, t_in, t_out, m->n, out, start, end, ch, p->sample_rate, f, ph); \
        else ZH_LAUNCH((k_sineosc_ranges<false, FB, PB, TOL>), grid, dim3(64), 0, st, t_in, t_out, m->n, out, start, end, ch, p->sample_rate, f, ph);  \
    } while (0)
#define ZH_SINE_R(FB, PB) do { if (tol) ZH_SINE_R2(FB, PB, true); else ZH_SINE_R2(FB, PB, false); } while (0)
        if (fb && pb) ZH_SINE_R(true, true);
        else if (fb) ZH_SINE_R(true, false);
        else if (pb) ZH_SINE_R(false, true);
        else ZH_SINE_R(false, false);
#undef ZH_SINE_R
#undef ZH_SINE_R2
        zh_flipper_painted(m);
        m->cur ^= 1;
        return zh_launch_status();
    }
#define ZH_SINE2(FB, PB, TOL)                                                                                       \
    do {                                                                                                            \
        if (zf) ZH_LAUNCH((k_sineosc<true, FB, PB, TOL>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->t(), m->n, out, start, end, p->sample_rate, f, ph); \
        else ZH_LAUNCH((k_sineosc<false, FB, PB, TOL>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->t(), m->n, out, start, end, p->sample_rate, f, ph);  \
    } while (0)
#define ZH_SINE(FB, PB) do { if (tol) ZH_SINE2(FB, PB, true); else ZH_SINE2(FB, PB, false); } while (0)
    if (fb && pb) ZH_SINE(true, true);
    else if (fb) ZH_SINE(true, false);
    else if (pb) ZH_SINE(false, true);
    else ZH_SINE(false, false);
#undef ZH_SINE
#undef ZH_SINE2
    return zh_launch_status();
}

// ------------------------------------------------------------------ Noise
static void noise_free(zh_noise *m) {
    for (auto &x : m->s) (void)hipFree(x);
    for (auto &x : m->nx) (void)hipFree(x);
    (void)hipFree(m->b); (void)hipFree(m->flag); (void)hipFree(m->scratch.ptr); (void)hipFree(m->err);
    (void)hipFree(m->tp_cs); (void)hipFree(m->tp_e); (void)hipFree(m->tp_flag); (void)hipFree(m->tp_taps);
}
int zh_noise_create(zh_ctx *ctx, uint32_t n, uint64_t first_seed, zh_noise **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_noise *m = new (std::nothrow) zh_noise();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->b = nullptr; m->flag = nullptr; m->err = nullptr;
    m->tp_cs = nullptr; m->tp_e = nullptr; m->tp_flag = nullptr; m->tp_serial = 0; m->tp_taps = nullptr;
    for (int i = 0; i < 4; i++) m->s[i] = m->nx[i] = nullptr;
    memset(&m->scratch, 0, sizeof m->scratch);
    int rc = 0;
    for (int i = 0; i < 4 && !rc; i++) rc = dev_alloc(&m->s[i], n);
    for (int i = 0; i < 4 && !rc; i++) rc = dev_alloc(&m->nx[i], n);
    if (!rc) rc = dev_alloc(&m->b, (size_t)7 * n);
    if (!rc) rc = dev_alloc(&m->flag, n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->flag, 0, (size_t)n * 4, ctx->stream);
    if (!rc) rc = dev_alloc(&m->err, 1);
    if (!rc) rc = (int)hipMemsetAsync(m->err, 0, 4, ctx->stream);
    if (rc) { noise_free(m); delete m; return rc; }
    if (n) ZH_LAUNCH(k_noise_seed, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, m->s[0], m->s[1], m->s[2], m->s[3], m->b, n, first_seed);
    // the jump tables are per context and built on first use; doing that here keeps it out of paint (and out of any capture)
    if (n && zh_noise_range_frames(n, 1024)) (void)zh_noise_jump_tables(ctx);
    *out = m;
    return zh_launch_status();
}
int zh_noise_destroy(zh_noise *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    noise_free(m);
    delete m;
    return ZH_OK;
}
int zh_noise_get_state(zh_noise *m, zh_noise_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    uint32_t ring_error = 0;
    if (zh_download(m->ctx, &ring_error, m->err, 4) != ZH_OK || ring_error) return ZH_ERR_INVALID;
    std::vector<uint64_t> s;
    std::vector<float> b;
    for (int i = 0; i < 4; i++) {
        int rc = download_field(m->ctx, s, m->s[i], m->n);
        if (rc) return rc;
        for (uint32_t v = 0; v < m->n; v++) host[v].r[i] = s[v];
    }
    int rc = download_field(m->ctx, b, m->b, (size_t)7 * m->n);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->n; v++) { for (int j = 0; j < 7; j++) host[v].b[j] = b[(size_t)j * m->n + v]; host[v].reserved = 0; }
    return ZH_OK;
}
int zh_noise_set_state(zh_noise *m, const zh_noise_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint64_t> s(m->n);
    std::vector<float> b((size_t)7 * m->n);
    for (int i = 0; i < 4; i++) {
        for (uint32_t v = 0; v < m->n; v++) s[v] = host[v].r[i];
        int rc = upload_field(m->ctx, m->s[i], s);
        if (rc) return rc;
    }
    for (uint32_t v = 0; v < m->n; v++) for (int j = 0; j < 7; j++) b[(size_t)j * m->n + v] = host[v].b[j];
    return upload_field(m->ctx, m->b, b);
}
int zh_noise_paint(zh_noise *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                   zh_bool note_id_changed, const zh_noise_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // Noise.zig:42-43
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || p->color > ZH_NOISE_PINK) return ZH_ERR_INVALID;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    // Few voices: the white samples as many frame ranges of the span at once (noise_jump.hip).  White noise with zero +
    // paint writes the output directly; the reference's `+=` onto existing content, and pink noise, render the white into a
    // module-owned image first (the repair of a multi-draw voice needs the noise on its own) and add / filter it from there.
    const bool pink = p->color == ZH_NOISE_PINK;
    // ZH_PAINT_TOLERANT, pink, few voices: the taps as chunks at once over exactly generated white noise (k_pink_tp_a / _b)
    if ((flags & ZH_PAINT_TOLERANT) && pink && end - start >= 128 && outputs[0].stride <= (1u << 24)) {
        const uint32_t Cw = zh_tp_chunks(m->n, ZF_PINK_TP_MAX, 1024);
        const uint4 *tables = Cw >= 2 ? zh_noise_jump_tables(m->ctx) : nullptr;
        if (tables && !m->tp_cs && !m->ctx->capturing) {
            int arc = dev_alloc(&m->tp_cs, (size_t)kTpMaxChunks * 4 * m->n);
            if (!arc) arc = dev_alloc(&m->tp_e, (size_t)kTpMaxChunks * 7 * m->n);
            if (!arc) arc = dev_alloc(&m->tp_flag, m->n);
            if (!arc) arc = dev_alloc(&m->tp_taps, (size_t)2 * 7 * m->n);
            if (!arc) arc = (int)hipMemsetAsync(m->tp_flag, 0, (size_t)m->n * 4, st);
            if (arc) { (void)hipFree(m->tp_cs); (void)hipFree(m->tp_e); (void)hipFree(m->tp_flag); (void)hipFree(m->tp_taps); m->tp_cs = nullptr; m->tp_e = nullptr; m->tp_flag = nullptr; m->tp_taps = nullptr; (void)hipGetLastError(); }
        }
        if (tables && m->tp_cs) {
            const uint32_t L = 32u * max(1u, 32u / Cw);
            PinkTpArgs a;
            for (int i = 0; i < 4; i++) a.s[i] = m->s[i];
            a.b0 = m->b; a.cs = m->tp_cs; a.e = m->tp_e; a.flag = m->tp_flag; a.tables = tables;
            a.V = m->n; a.L = L; a.out = mk_img(outputs[0]);
            const uint32_t piece = min(kTpMaxChunks * L, ((uint32_t)kNoiseJumpTables * 32u / L) * L + L);   // (C - 1) * L / 32 <= kNoiseJumpTables
            // The taps run over the WHOLE span (Noise.zig:55-68) and are never stored in the module: a piece leaves the taps it ends
            // on in tp_taps[piece & 1] and the next piece starts from them (every piece used to restart from m->b).
            uint32_t pc = 0;
            for (uint32_t s0 = start; s0 < end; s0 += piece, pc++) {
                a.start = s0; a.end = min(s0 + piece, end);
                a.b0 = pc == 0 ? m->b : m->tp_taps + (size_t)((pc - 1) & 1u) * 7 * m->n;
                a.b_end = m->tp_taps + (size_t)(pc & 1u) * 7 * m->n;
                a.C = (a.end - a.start + L - 1) / L;
                if (++m->tp_serial == 0) m->tp_serial = 1;
                a.serial = m->tp_serial;
                a.per = (m->n + 255u) / 256u;
                const dim3 grid(((a.C + 7u) / 8u) * 8u * a.per);                         // filter_tp.hip.h nf_tp_block
                const dim3 grid_b(8u * a.C * ((a.per + 7u) / 8u));                         // nf_tp_block_b
                ZH_LAUNCH(k_pink_tp_a, grid, dim3(256), 0, st, a);
                if (zf) ZH_LAUNCH(k_pink_tp_b<true>, grid_b, dim3(256), 0, st, a);
                else ZH_LAUNCH(k_pink_tp_b<false>, grid_b, dim3(256), 0, st, a);
            }
            return zh_launch_status();
        }
    }
    const uint32_t ch = zh_noise_range_frames(m->n, end - start);
    // pink, 1,024 / 4,096 / 16,384 / 32,768 voices: 124 / 116 / 118 / 130 us in one loop, 57 / 67 / 84 / 161 us as white ranges +
    // the seven-stage chain (the white kernel is 10-29 us of that), 43 / 54 / 68 / 123 us as white ranges + k_pink_taps
    const uint32_t pink_max = (uint32_t)zh_form(ZF_PINK_PIPE_MAX);
    if (ch && (!pink || m->n <= pink_max)) {
        if (zf && !pink) {
            rc = zh_noise_paint_ranges(m->ctx, m->s, m->nx, m->flag, m->n, outputs[0], start, end, ch);
            if (rc != ZH_ERR_UNSUPPORTED) return rc;
        } else {
            if ((m->scratch.frames < end || !m->scratch.ptr) && !m->ctx->capturing) {
                zh_buf nb;
                if (zh_buf_alloc(m->ctx, &nb, m->n, end) == ZH_OK) {
                    if (m->scratch.ptr) m->ctx->mix_retired.push_back(m->scratch.ptr);   // a captured graph may still name it: freed with the context
                    m->scratch = nb;
                }
            }
            if (m->scratch.ptr && m->scratch.frames >= end) {
                rc = zh_noise_paint_ranges(m->ctx, m->s, m->nx, m->flag, m->n, m->scratch, start, end, ch);
                if (rc == ZH_OK && !pink) return zh_add_into(m->ctx, start, end, outputs[0], m->scratch);
                if (rc == ZH_OK) {
                    const dim3 grid((m->n + 63) / 64);
                    const long tapsf = zh_form(ZF_PINK_TAPS);             // 0 = the chain of seven stages (k_pink_pipe)
                    const bool taps = tapsf != 0 && end - start >= 32 && outputs[0].stride <= (1u << 24) && m->scratch.stride <= (1u << 24);   // (32-row tiles: 32-bit offsets)
                    if (taps && m->n <= 16384 && tapsf != 16) {   // (ZH_PINK_TAPS=16 forces the 16-frame tiles)
                        if (zf) ZH_LAUNCH((k_pink_taps<true, 32>), grid, dim3(256), 0, st, m->b, m->n, mk_img(outputs[0]), mk_cimg(m->scratch), start, end);
                        else ZH_LAUNCH((k_pink_taps<false, 32>), grid, dim3(256), 0, st, m->b, m->n, mk_img(outputs[0]), mk_cimg(m->scratch), start, end);
                    } else if (taps) {                                    // 16-frame tiles: two workgroups per CU
                        if (zf) ZH_LAUNCH((k_pink_taps<true, 16>), grid, dim3(256), 0, st, m->b, m->n, mk_img(outputs[0]), mk_cimg(m->scratch), start, end);
                        else ZH_LAUNCH((k_pink_taps<false, 16>), grid, dim3(256), 0, st, m->b, m->n, mk_img(outputs[0]), mk_cimg(m->scratch), start, end);
                    } else {
                        if (zf) ZH_LAUNCH(k_pink_pipe<true>, grid, dim3(64 * kPinkWaves), 0, st, m->b, m->n, mk_img(outputs[0]), mk_cimg(m->scratch), start, end, m->err);
                        else ZH_LAUNCH(k_pink_pipe<false>, grid, dim3(64 * kPinkWaves), 0, st, m->b, m->n, mk_img(outputs[0]), mk_cimg(m->scratch), start, end, m->err);
                    }
                    return zh_launch_status();
                }
                if (rc != ZH_ERR_UNSUPPORTED) return rc;
            }
        }
    }
    Img out = mk_img(outputs[0]);
#define ZH_NOISE(ZF_, PINK_) ZH_LAUNCH((k_noise<ZF_, PINK_>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->s[0], m->s[1], m->s[2], m->s[3], m->b, m->n, out, start, end)
    if (p->color == ZH_NOISE_PINK) { if (zf) ZH_NOISE(true, true); else ZH_NOISE(false, true); }
    else { if (zf) ZH_NOISE(true, false); else ZH_NOISE(false, false); }
#undef ZH_NOISE
    return zh_launch_status();
}

// ------------------------------------------------------------------ Envelope
int zh_envelope_create(zh_ctx *ctx, uint32_t n, zh_envelope **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_envelope *m = new (std::nothrow) zh_envelope();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0; m->words = 4;
    int rc = dev_alloc(&m->cnt[0], (size_t)4 * n);
    if (!rc) rc = dev_alloc(&m->cnt[1], (size_t)4 * n);
    for (int b = 0; b < 2 && !rc && n; b++) rc = (int)hipMemsetAsync(m->cnt[b], 0, (size_t)4 * n * 4, ctx->stream);   // init() :26-31: idle, painter zeros
    if (rc) { (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
int zh_envelope_destroy(zh_envelope *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]);
    delete m;
    return ZH_OK;
}
int zh_envelope_get_state(zh_envelope *m, zh_envelope_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> w;
    int rc = download_field(m->ctx, w, m->cnt[m->cur], (size_t)4 * m->n);
    if (rc) return rc;
    const size_t n = m->n;
    for (uint32_t v = 0; v < m->n; v++) {
        zh_envelope_state e;
        e.state = w[v];
        memcpy(&e.t, &w[n + v], 4); memcpy(&e.last_value, &w[2 * n + v], 4); memcpy(&e.start, &w[3 * n + v], 4);
        host[v] = e;
    }
    return ZH_OK;
}
int zh_envelope_set_state(zh_envelope *m, const zh_envelope_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    const size_t n = m->n;
    std::vector<uint32_t> w(4 * n);
    for (uint32_t v = 0; v < m->n; v++) {
        w[v] = host[v].state;
        memcpy(&w[n + v], &host[v].t, 4); memcpy(&w[2 * n + v], &host[v].last_value, 4); memcpy(&w[3 * n + v], &host[v].start, 4);
    }
    return upload_field(m->ctx, m->cnt[m->cur], w);
}
int zh_envelope_paint(zh_envelope *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                      zh_bool note_id_changed, const zh_envelope_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || !curve_ok(p->attack) || !curve_ok(p->decay) || !curve_ok(p->release)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;            // an empty span still runs the state prologue (Envelope.zig:41-50)
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    const int ft = p->attack.tag == p->decay.tag && p->decay.tag == p->release.tag && p->attack.tag != ZH_CURVE_INSTANTANEOUS ? (int)p->attack.tag : -1;
    const int c = m->cur;
    // few voices: frame ranges with a replay of the clock (4,096 / 8,192 / 16,384 / 32,768 voices: 35 / 35 / 36 / 42 us as one
    // walk, 22.4 / 23.9 / 25.8 / 33.3 us; wave targets 1,024 / 2,048 / 4,096 measured, 2,048 best or level everywhere)
    const uint32_t ch = end > start ? zh_range_frames(m->n, end - start, ZF_ENVELOPE_RANGES, 2048, 40960) : 0;
    if (ch) {
        const dim3 grid((m->n + 63) / 64, (end - start + ch - 1) / ch);
#define ZH_ENVR(FT_)                                                                                                         \
    do {                                                                                                                     \
        if (zf) ZH_LAUNCH((k_envelope_ranges<true, FT_>), grid, dim3(64), 0, st, m->st(c), m->f(c, 1), m->f(c, 2), m->f(c, 3), m->cnt[c ^ 1], m->n, \
                                   mk_img(outputs[0]), start, end, ch, mk_env_params(p), mk_bool(note_id_changed));         \
        else ZH_LAUNCH((k_envelope_ranges<false, FT_>), grid, dim3(64), 0, st, m->st(c), m->f(c, 1), m->f(c, 2), m->f(c, 3), m->cnt[c ^ 1], m->n, \
                                mk_img(outputs[0]), start, end, ch, mk_env_params(p), mk_bool(note_id_changed));            \
    } while (0)
        switch (ft) {
        case ZH_CURVE_LINEAR: ZH_ENVR(ZH_CURVE_LINEAR); break;
        case ZH_CURVE_SQUARED: ZH_ENVR(ZH_CURVE_SQUARED); break;
        case ZH_CURVE_CUBED: ZH_ENVR(ZH_CURVE_CUBED); break;
        default: ZH_ENVR(-1); break;
        }
#undef ZH_ENVR
        zh_flipper_painted(m);
        m->cur ^= 1;
        return zh_launch_status();
    }
#define ZH_ENV(FT_)                                                                                                          \
    do {                                                                                                                     \
        if (zf) ZH_LAUNCH((k_envelope<true, FT_>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->st(c), m->f(c, 1), m->f(c, 2), m->f(c, 3), m->n, \
                                   mk_img(outputs[0]), start, end, mk_env_params(p), mk_bool(note_id_changed));             \
        else ZH_LAUNCH((k_envelope<false, FT_>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->st(c), m->f(c, 1), m->f(c, 2), m->f(c, 3), m->n, \
                                mk_img(outputs[0]), start, end, mk_env_params(p), mk_bool(note_id_changed));                \
    } while (0)
    switch (ft) {
    case ZH_CURVE_LINEAR: ZH_ENV(ZH_CURVE_LINEAR); break;
    case ZH_CURVE_SQUARED: ZH_ENV(ZH_CURVE_SQUARED); break;
    case ZH_CURVE_CUBED: ZH_ENV(ZH_CURVE_CUBED); break;
    default: ZH_ENV(-1); break;
    }
#undef ZH_ENV
    return zh_launch_status();
}

// ------------------------------------------------------------------ Gate
int zh_gate_create(zh_ctx *ctx, uint32_t n, zh_gate **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    *out = new (std::nothrow) zh_gate{ctx, n};
    return *out ? ZH_OK : ZH_ERR_INVALID;
}
int zh_gate_destroy(zh_gate *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    delete m;
    return ZH_OK;
}
int zh_gate_paint(zh_gate *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                  zh_bool note_id_changed, const zh_gate_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // Gate.zig:24-26
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p) return ZH_ERR_INVALID;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    if (m->n % 4 == 0 && outputs[0].stride % 4 == 0 && ((uintptr_t)outputs[0].ptr & 15u) == 0) {
        const uint32_t nvq = m->n / 4, chunks3 = (end - start + 2) / 3;
        ZH_ZF_LAUNCH(k_gate4, dim3((nvq + 63) / 64, (chunks3 + 3) / 4), dim3(256), nvq, mk_img(outputs[0]), start, end, mk_bool(p->note_on));
        return zh_launch_status();
    }
    const uint32_t chunks = (end - start + 31) / 32;
    ZH_ZF_LAUNCH(k_gate, dim3((m->n + 63) / 64, (chunks + 3) / 4), dim3(256), m->n, mk_img(outputs[0]), start, end, mk_bool(p->note_on));
    return zh_launch_status();
}

// ------------------------------------------------------------------ Filter
int zh_filter_create(zh_ctx *ctx, uint32_t n, zh_filter **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_filter *m = new (std::nothrow) zh_filter{ctx, n, nullptr, nullptr};
    if (!m) return ZH_ERR_INVALID;
    int rc = dev_alloc(&m->l, n);
    if (!rc) rc = dev_alloc(&m->b, n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->l, 0, n * 4, ctx->stream);           // init() :37-42
    if (!rc && n) rc = (int)hipMemsetAsync(m->b, 0, n * 4, ctx->stream);
    if (rc) { (void)hipFree(m->l); (void)hipFree(m->b); delete m; return rc; }
    *out = m;
    return ZH_OK;
}
int zh_filter_destroy(zh_filter *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    (void)hipFree(m->l); (void)hipFree(m->b); (void)hipFree(m->tp_e);
    delete m;
    return ZH_OK;
}
int zh_filter_get_state(zh_filter *m, zh_filter_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<float> l, b;
    int rc = download_field(m->ctx, l, m->l, m->n);
    if (!rc) rc = download_field(m->ctx, b, m->b, m->n);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->n; v++) host[v] = zh_filter_state{l[v], b[v]};
    return ZH_OK;
}
int zh_filter_set_state(zh_filter *m, const zh_filter_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<float> l(m->n), b(m->n);
    for (uint32_t v = 0; v < m->n; v++) { l[v] = host[v].l; b[v] = host[v].b; }
    int rc = upload_field(m->ctx, m->l, l);
    if (!rc) rc = upload_field(m->ctx, m->b, b);
    return rc;
}
int zh_filter_paint(zh_filter *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                    zh_bool note_id_changed, const zh_filter_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // Filter.zig:52-53
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || p->type > ZH_FILTER_ALL_PASS || !buf_covers(p->input, m->n, end) || !cob_ok(p->cutoff, m->n, end) ||
        !cob_ok(p->res, m->n, end))
        return ZH_ERR_INVALID;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    zh_buf o = outputs[0], in = p->input;
    o.voices = m->n; in.voices = m->n;
    if (p->type == ZH_FILTER_BYPASS) {                                             // :91-97: out += in, state untouched
        if (zf) { rc = zh_zero(m->ctx, start, end, o); if (rc) return rc; }
        return zh_add_into(m->ctx, start, end, o, in);
    }
    float l_mul = 0.0f, b_mul = 0.0f, h_mul = 0.0f;                                // :98-109
    switch (p->type) {
    case ZH_FILTER_LOW_PASS: l_mul = 1.0f; break;
    case ZH_FILTER_BAND_PASS: b_mul = 1.0f; break;
    case ZH_FILTER_HIGH_PASS: h_mul = 1.0f; break;
    case ZH_FILTER_NOTCH: l_mul = 1.0f; h_mul = 1.0f; break;
    default: l_mul = 1.0f; b_mul = 1.0f; h_mul = 1.0f; break;
    }
    const bool cb = p->cutoff.tag == ZH_COB_BUFFER, rb = p->res.tag == ZH_COB_BUFFER;
    Img out = mk_img(outputs[0]);
    CImg inp = mk_cimg(p->input);
    CobP cut = mk_cob(p->cutoff), res = mk_cob(p->res);
#define ZH_FILTER(CB, RB)                                                                                            \
    do {                                                                                                             \
        if (zf) ZH_LAUNCH((k_filter<true, CB, RB>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut, res); \
        else ZH_LAUNCH((k_filter<false, CB, RB>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut, res);  \
    } while (0)
    // few voices, constant cutoff / resonance: the three-wave pipeline (1,024 / 4,096 / 16,384 / 32,768 voices: 54.5 / 56.1 /
    // 57.3 / 75.8 us in one wave, 41 / 41.8 / 45 / 51 us; its 64 KB of LDS per workgroup allow 32,768 voices at once, with
    // 16-frame tiles it goes on to 65,536 voices)
    const uint32_t pc_max = (uint32_t)zh_form(ZF_FILTER_PC_MAX);
    // 16-frame tiles above filter_pc_max voices (filter_pc_max=0 alone switches both off)
    const uint32_t pc16_max = (pc_max == 0 && !zh_form_is_set(ZF_FILTER_PC16_MAX)) ? 0u : (uint32_t)zh_form(ZF_FILTER_PC16_MAX);     // 36,864 / 49,152 / 65,536 voices: 97 / 103 / 116 us in one wave, 72 / 79 / 104; 81,920: 126 against 168
    // ZH_PAINT_TOLERANT, few voices: the span as chunks at once (filter_tp.hip.h); every other case paints with an exact form
    if ((flags & ZH_PAINT_TOLERANT) && !bufs_alias(p->input, outputs[0]) && !cob_aliases(p->cutoff, outputs[0]) && !cob_aliases(p->res, outputs[0]) &&
        outputs[0].stride <= (1u << 24) && p->input.stride <= (1u << 24) && (!cb || p->cutoff.buffer.stride <= (1u << 24)) && (!rb || p->res.buffer.stride <= (1u << 24))) {
        if (!m->tp_e && !m->ctx->capturing && zh_tp_chunks(m->n, ZF_FILTER_TP_MAX, end - start) >= 2 &&
            (dev_alloc(&m->tp_e, kFilterTpFloats * m->n) != ZH_OK ||
             hipMemsetAsync(m->tp_e + kFilterTpFlagAt * m->n, 0, (size_t)m->n * 4, st) != hipSuccess)) { (void)hipFree(m->tp_e); m->tp_e = nullptr; (void)hipGetLastError(); }
        if (m->tp_e && zh_filter_tp_launch(st, m->l, m->b, m->tp_e, m->tp_serial, m->n, out, inp, start, end, zf, l_mul, b_mul, h_mul, cut, res))
            return zh_launch_status();
    }
    // (a tile's 32 rows are addressed with 32-bit offsets from one descriptor: row strides up to 2^24 voices)
    const bool tile_strides_ok = outputs[0].stride <= (1u << 24) && p->input.stride <= (1u << 24) && (!cb || p->cutoff.buffer.stride <= (1u << 24)) &&
                                 (!rb || p->res.buffer.stride <= (1u << 24));
    if (!cb && !rb && m->n <= max(pc_max, pc16_max) && end - start >= 64 && !bufs_alias(p->input, outputs[0]) && tile_strides_ok) {
        const dim3 grid((m->n + 63) / 64);
        if (m->n <= pc_max) {
            if (zf) ZH_LAUNCH((k_filter_pc<true, 32>), grid, dim3(192), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut.c, res.c);
            else ZH_LAUNCH((k_filter_pc<false, 32>), grid, dim3(192), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut.c, res.c);
        } else {
            if (zf) ZH_LAUNCH((k_filter_pc<true, 16>), grid, dim3(192), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut.c, res.c);
            else ZH_LAUNCH((k_filter_pc<false, 16>), grid, dim3(192), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut.c, res.c);
        }
        return zh_launch_status();
    }
    // control images at few voices: the same pipeline with the images' rows as tiles of their own (k_filter_pc_ctl; a cutoff sweep at
    // 4,096 / 16,384 / 32,768 voices: 95.5 / 101.7 / 139.9 us as the one-wave walk, 46.8 / 48.9 / 98.4; 65,536: 160 against 195 --
    // the walk).  ZH_FILTER_PC_CTL_MAX = largest voice count (0 = never).
    {
        const uint32_t ctl_max = (uint32_t)zh_form(ZF_FILTER_PC_CTL_MAX);
        if ((cb || rb) && m->n <= ctl_max && end - start >= 64 && !bufs_alias(p->input, outputs[0]) && !cob_aliases(p->cutoff, outputs[0]) &&
            !cob_aliases(p->res, outputs[0]) && tile_strides_ok) {
            const dim3 grid((m->n + 63) / 64);
#define ZH_FPCC(CH_, CB_, RB_)                                                                                       \
            do {                                                                                                     \
                if (zf) ZH_LAUNCH((k_filter_pc_ctl<true, CH_, CB_, RB_>), grid, dim3(192), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut, res); \
                else ZH_LAUNCH((k_filter_pc_ctl<false, CH_, CB_, RB_>), grid, dim3(192), 0, st, m->l, m->b, m->n, out, inp, start, end, l_mul, b_mul, h_mul, cut, res);  \
            } while (0)
            if (cb && rb) ZH_FPCC(16, true, true);
            else if (cb) ZH_FPCC(32, true, false);
            else ZH_FPCC(32, false, true);
#undef ZH_FPCC
            return zh_launch_status();
        }
    }
    if (cb && rb) ZH_FILTER(true, true);
    else if (cb) ZH_FILTER(true, false);
    else if (rb) ZH_FILTER(false, true);
    else ZH_FILTER(false, false);
#undef ZH_FILTER
    return zh_launch_status();
}
int zh_filter_cutoff_from_frequency(zh_ctx *ctx, uint32_t n, float *cutoff_out, const float *frequency, float sample_rate) { ZH_GUARD(ctx);
    if (!ctx || (n && (!cutoff_out || !frequency))) return ZH_ERR_INVALID;
    if (!n) return ZH_OK;
    ZH_LAUNCH(k_cutoff_from_frequency, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, cutoff_out, frequency, sample_rate);
    return zh_launch_status();
}

int zh_pow(zh_ctx *ctx, uint32_t n, float *out, const float *x, const float *y) { ZH_GUARD(ctx);
    if (!ctx || (n && (!out || !x || !y))) return ZH_ERR_INVALID;
    if (!n) return ZH_OK;
    ZH_LAUNCH(k_pow, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, out, x, y);
    return zh_launch_status();
}

int zh_sin(zh_ctx *ctx, uint32_t n, float *out, const float *x) { ZH_GUARD(ctx);
    if (!ctx || (n && (!out || !x))) return ZH_ERR_INVALID;
    if (!n) return ZH_OK;
    ZH_LAUNCH(k_sincos<0>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, out, x);
    return zh_launch_status();
}
int zh_cos(zh_ctx *ctx, uint32_t n, float *out, const float *x) { ZH_GUARD(ctx);
    if (!ctx || (n && (!out || !x))) return ZH_ERR_INVALID;
    if (!n) return ZH_OK;
    ZH_LAUNCH(k_sincos<1>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, out, x);
    return zh_launch_status();
}

int zh_atan(zh_ctx *ctx, uint32_t n, float *out, const float *x) { ZH_GUARD(ctx);
    if (!ctx || (n && (!out || !x))) return ZH_ERR_INVALID;
    if (!n) return ZH_OK;
    ZH_LAUNCH(k_sincos<2>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, out, x);
    return zh_launch_status();
}

// ------------------------------------------------------------------ Sampler
int zh_sampler_create(zh_ctx *ctx, uint32_t n, zh_sampler **out) { ZH_GUARD(ctx); return flip1_create(ctx, n, out); }   // init() :71-75
int zh_sampler_destroy(zh_sampler *m) { ZH_GUARD(m ? m->ctx : nullptr); return flip1_destroy(m); }
int zh_sampler_get_state(zh_sampler *m, zh_sampler_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_download(m->ctx, host, m->t(), (size_t)m->n * 4);
}
int zh_sampler_set_state(zh_sampler *m, const zh_sampler_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_upload(m->ctx, m->t(), host, (size_t)m->n * 4);
}
int zh_sampler_paint(zh_sampler *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                     zh_bool note_id_changed, const zh_sampler_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || p->sample.format > ZH_SAMPLE_S32_LSB || (p->sample.data_len && !p->sample.data)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    zh_buf o = outputs[0];
    o.voices = m->n;
    if (p->channel >= p->sample.num_channels) {                                    // :87-89: nothing, not even the t reset
        return zf ? zh_zero(m->ctx, start, end, o) : ZH_OK;
    }
    const uint64_t bps = (uint64_t)p->sample.format + 1;
    SampleP s;
    s.data = p->sample.data;
    s.data_len = p->sample.data_len;
    s.num_channels = (uint32_t)p->sample.num_channels;
    s.sample_rate_in = (uint32_t)p->sample.sample_rate;
    s.format = p->sample.format;
    s.channel = (uint32_t)p->channel;
    s.loop = p->loop ? 1u : 0u;
    s.whole = ((uintptr_t)p->sample.data % bps) == 0 ? 1u : 0u;
    const uint64_t count = p->sample.data_len / bps / p->sample.num_channels;
    if (count > 0x7fffffffull) return ZH_ERR_INVALID;                               // the reference's @intCast(i32, ...) traps (:42)
    s.num_samples = (int32_t)count;
    s.inv_num_samples = s.num_samples > 0 ? 1.0 / (double)s.num_samples : 0.0;
    const Img img = mk_img(outputs[0]);
    const F32P rate = mk_f32(p->sample_rate);
    const BoolP nicp = mk_bool(note_id_changed);
    const int fmt = s.num_samples == 0 ? kSampleEmpty : (int)s.format;
    // few voices: the span as frame ranges at once (see k_sampler); ZH_SAMPLER_RANGES = number of ranges, 0 = never
    // 24,576 / 32,768 / 65,536 / 131,072 / 262,144 voices, sequential -> ranges: 145 -> 43, 145 -> 55, 143 -> 96, 212 -> 187, 377 -> 349 us
    uint32_t ch = end > start ? zh_range_frames(m->n, end - start, ZF_SAMPLER_RANGES, 16384, 1u << 20) : 0;
    const bool ranges = ch != 0;
    if (!ranges) ch = end > start ? end - start : 1;
    const float *t_in = m->t();
    float *t_out = ranges ? reinterpret_cast<float *>(m->cnt[m->cur ^ 1]) : m->t();
    const dim3 grid((m->n + kSeqBlock - 1) / kSeqBlock, ranges ? (end - start + ch - 1) / ch : 1);
#define ZH_SMP(ZF_, F_, L_) ZH_LAUNCH((k_sampler<ZF_, F_, L_>), grid, dim3(kSeqBlock), 0, st, t_in, t_out, m->n, img, start, end, ch, s, rate, nicp)
#define ZH_SMP_L(ZF_, F_) do { if (s.loop) ZH_SMP(ZF_, F_, true); else ZH_SMP(ZF_, F_, false); } while (0)
#define ZH_SMP_F(ZF_) do { switch (fmt) { case kSampleEmpty: ZH_SMP_L(ZF_, kSampleEmpty); break; case ZH_SAMPLE_U8: ZH_SMP_L(ZF_, ZH_SAMPLE_U8); break; \
        case ZH_SAMPLE_S16_LSB: ZH_SMP_L(ZF_, ZH_SAMPLE_S16_LSB); break; case ZH_SAMPLE_S24_LSB: ZH_SMP_L(ZF_, ZH_SAMPLE_S24_LSB); break; \
        default: ZH_SMP_L(ZF_, ZH_SAMPLE_S32_LSB); break; } } while (0)
    if (zf) ZH_SMP_F(true); else ZH_SMP_F(false);
#undef ZH_SMP_F
#undef ZH_SMP_L
#undef ZH_SMP
    if (ranges) { zh_flipper_painted(m); m->cur ^= 1; }
    return zh_launch_status();
}

// ------------------------------------------------------------------ Decimator
int zh_decimator_create(zh_ctx *ctx, uint32_t n, zh_decimator **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_decimator *m = new (std::nothrow) zh_decimator();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0; m->words = 2;
    int rc = dev_alloc(&m->cnt[0], (size_t)2 * n);
    if (!rc) rc = dev_alloc(&m->cnt[1], (size_t)2 * n);
    if (!rc && n) {                                                                // init() :14-19: dval 0, dcount 1
        std::vector<float> init((size_t)2 * n, 0.0f);
        for (uint32_t v = 0; v < n; v++) init[(size_t)n + v] = 1.0f;
        rc = upload_field(ctx, m->dval(0), init);
        if (!rc) rc = upload_field(ctx, m->dval(1), init);
    }
    if (rc) { (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
int zh_decimator_destroy(zh_decimator *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]);
    delete m;
    return ZH_OK;
}
int zh_decimator_get_state(zh_decimator *m, zh_decimator_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<float> w;
    int rc = download_field(m->ctx, w, m->dval(m->cur), (size_t)2 * m->n);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->n; v++) host[v] = zh_decimator_state{w[v], w[(size_t)m->n + v]};
    return ZH_OK;
}
int zh_decimator_set_state(zh_decimator *m, const zh_decimator_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<float> w((size_t)2 * m->n);
    for (uint32_t v = 0; v < m->n; v++) { w[v] = host[v].dval; w[(size_t)m->n + v] = host[v].dcount; }
    return upload_field(m->ctx, m->dval(m->cur), w);
}
int zh_decimator_paint(zh_decimator *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                       zh_bool note_id_changed, const zh_decimator_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // Decimator.zig:29-30
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || !buf_covers(p->input, m->n, end)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;            // an empty span still resets state when fake >= sample_rate (:37-38)
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    // 4,096 voices: 94 us sequential with per-lane branches, 39 straight-line, 31 as 16 frame ranges (the replay is 8 issue
    // slots a frame against 13.5 for a painted frame: 4 / 8 / 16 / 32 / 64 ranges = 32.4 / 31.5 / 30.6 / 31.5 / 35.4 us)
    const uint32_t ch = end > start && !bufs_alias(p->input, outputs[0]) ? zh_range_frames(m->n, end - start, ZF_DECIMATOR_RANGES, m->n <= 32768 ? 1024 : 2048, 65536) : 0;   // (40,960 / 49,152 / 65,536 voices: 97 / 100 / 110 -> 69 / 77 / 106 us)
    if (ch) {
        const dim3 grid((m->n + 63) / 64, (end - start + ch - 1) / ch);
        const int c = m->cur;
        ZH_ZF_LAUNCH(k_decimator_ranges, grid, dim3(64), m->dval(c), m->dcount(c), m->dval(c ^ 1), m->n, mk_img(outputs[0]), mk_cimg(p->input),
                     start, end, ch, p->sample_rate, mk_f32(p->fake_sample_rate));
        zh_flipper_painted(m);
        m->cur ^= 1;
        return zh_launch_status();
    }
    ZH_ZF_LAUNCH(k_decimator, seq_grid(m->n), dim3(kSeqBlock), m->dval(m->cur), m->dcount(m->cur), m->n, mk_img(outputs[0]),
                 mk_cimg(p->input), start, end, p->sample_rate, mk_f32(p->fake_sample_rate));
    return zh_launch_status();
}

// ------------------------------------------------------------------ Curve
int zh_curve_module_create(zh_ctx *ctx, uint32_t n, zh_curve_module **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_curve_module *m = new (std::nothrow) zh_curve_module();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0; m->words = 4;
    int rc = dev_alloc(&m->cnt[0], (size_t)4 * n);
    if (!rc) rc = dev_alloc(&m->cnt[1], (size_t)4 * n);
    for (int b = 0; b < 2 && !rc && n; b++) rc = (int)hipMemsetAsync(m->cnt[b], 0, (size_t)4 * n * 4, ctx->stream);   // init() :46-54
    if (rc) { (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
int zh_curve_module_destroy(zh_curve_module *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]);
    delete m;
    return ZH_OK;
}
int zh_curve_module_get_state(zh_curve_module *m, zh_curve_module_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> w;
    int rc = download_field(m->ctx, w, m->cnt[m->cur], (size_t)4 * m->n);
    if (rc) return rc;
    const size_t n = m->n;
    for (uint32_t v = 0; v < m->n; v++) {
        zh_curve_module_state e;
        memcpy(&e.t, &w[v], 4);
        e.current_song_note = w[n + v]; e.current_song_note_offset = (int32_t)w[2 * n + v]; e.next_song_note = w[3 * n + v];
        host[v] = e;
    }
    return ZH_OK;
}
int zh_curve_module_set_state(zh_curve_module *m, const zh_curve_module_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    const size_t n = m->n;
    std::vector<uint32_t> w(4 * n);
    for (uint32_t v = 0; v < m->n; v++) {
        memcpy(&w[v], &host[v].t, 4);
        w[n + v] = host[v].current_song_note; w[2 * n + v] = (uint32_t)host[v].current_song_note_offset; w[3 * n + v] = host[v].next_song_note;
    }
    return upload_field(m->ctx, m->cnt[m->cur], w);
}
int zh_curve_module_paint(zh_curve_module *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                          zh_bool note_id_changed, const zh_curve_module_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || p->function > ZH_CURVE_FN_SMOOTHSTEP || (p->curve_len && !p->curve) || p->curve_len > 0xFFFFFFFFull) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;            // an empty span still resets on note_id_changed (:66-71)
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    // 4,096 voices: 89 us with a span search behind a per-lane test in every unrolled frame; 36.5 us with the spans streamed
    // into a table in begin() and chunks without a span change running without the test (seq.hip.h frame_loop_gen).  Painted
    // as frame ranges it first took 28 us whatever the range count -- not begin() (3 us: a paint of 8 frames is 3.8 us in all) but
    // the accumulator replay as a rolled scalar loop and a second begin() in a state-advance kernel;
    // done properly (running span set up by begin(r0), unrolled replay, state in a flipped double buffer) frame ranges take
    // 9.2 / 11.7 / 17.0 / 29.4 us at 1,024 / 4,096 / 16,384 / 32,768 voices against 34.9 / 36.5 / 37.9 / 43.2 (an empty span still
    // runs begin(): the one-range form)
    const uint32_t chr = end > start ? zh_range_frames(m->n, end - start, ZF_CURVE_RANGES, 2048, 40960) : 0;   // (40,960 voices: 45.7 -> 36.9 us; no gain from 49,152)
    const uint32_t ch = chr ? chr : (end > start ? end - start : 1);
    const dim3 grid((m->n + kSeqBlock - 1) / kSeqBlock, chr ? (end - start + chr - 1) / chr : 1);
    ZH_ZF_LAUNCH(k_curve, grid, dim3(kSeqBlock), m->cnt[m->cur], m->cnt[chr ? m->cur ^ 1 : m->cur], m->n, mk_img(outputs[0]), start, end, ch,
                 p->sample_rate, p->function, p->curve, (uint32_t)p->curve_len, mk_bool(note_id_changed));
    if (chr) { zh_flipper_painted(m); m->cur ^= 1; }
    return zh_launch_status();
}

// ------------------------------------------------------------------ Cycle
int zh_cycle_create(zh_ctx *ctx, uint32_t n, zh_cycle **out) { ZH_GUARD(ctx); return flip1_create(ctx, n, out); }   // init() :16-20: t = 0
int zh_cycle_destroy(zh_cycle *m) { ZH_GUARD(m ? m->ctx : nullptr); return flip1_destroy(m); }
int zh_cycle_get_state(zh_cycle *m, zh_cycle_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_download(m->ctx, host, m->t(), (size_t)m->n * 4);
}
int zh_cycle_set_state(zh_cycle *m, const zh_cycle_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    return zh_upload(m->ctx, m->t(), host, (size_t)m->n * 4);
}
int zh_cycle_paint(zh_cycle *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                   zh_bool note_id_changed, const zh_cycle_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // Cycle.zig:30-31
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || !cob_ok(p->speed, m->n, end)) return ZH_ERR_INVALID;
    if (m->n == 0 || end == start) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    Img out = mk_img(outputs[0]);
    CobP sp = mk_cob(p->speed);
    // constant speed, few voices: frame ranges (1,024 / 4,096 / 16,384 voices: 17.6 / 19.5 / 20.4 us as one walk per voice, 10.3 / 12.5 / 15.1 us)
    const bool sb = p->speed.tag == ZH_COB_BUFFER;
    const uint32_t chr = sb ? 0 : zh_range_frames(m->n, end - start, ZF_CYCLE_RANGES, 1024, 16384);
    const uint32_t ch = chr ? chr : end - start;
    const dim3 grid((m->n + kSeqBlock - 1) / kSeqBlock, chr ? (end - start + chr - 1) / chr : 1);
    const float *t_in = m->t();
    float *t_out = chr ? reinterpret_cast<float *>(m->cnt[m->cur ^ 1]) : m->t();
#define ZH_CYCLE(ZF_, SB_) ZH_LAUNCH((k_cycle<ZF_, SB_>), grid, dim3(kSeqBlock), 0, st, t_in, t_out, m->n, out, start, end, ch, p->sample_rate, sp)
    if (sb) { if (zf) ZH_CYCLE(true, true); else ZH_CYCLE(false, true); }
    else { if (zf) ZH_CYCLE(true, false); else ZH_CYCLE(false, false); }
#undef ZH_CYCLE
    if (chr) { zh_flipper_painted(m); m->cur ^= 1; }
    return zh_launch_status();
}

// ------------------------------------------------------------------ Portamento
int zh_portamento_create(zh_ctx *ctx, uint32_t n, zh_portamento **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_portamento *m = new (std::nothrow) zh_portamento();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0; m->words = 3;
    int rc = dev_alloc(&m->cnt[0], (size_t)3 * n);
    if (!rc) rc = dev_alloc(&m->cnt[1], (size_t)3 * n);
    for (int b = 0; b < 2 && !rc && n; b++) rc = (int)hipMemsetAsync(m->cnt[b], 0, (size_t)3 * n * 4, ctx->stream);   // Painter.init(), painter.zig:38-44
    if (rc) { (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]); delete m; return rc; }
    zh_flipper_register(m);
    *out = m;
    return ZH_OK;
}
int zh_portamento_destroy(zh_portamento *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]);
    delete m;
    return ZH_OK;
}
int zh_portamento_get_state(zh_portamento *m, zh_portamento_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<float> w;
    int rc = download_field(m->ctx, w, m->f(m->cur, 0), (size_t)3 * m->n);
    if (rc) return rc;
    const size_t n = m->n;
    for (uint32_t v = 0; v < m->n; v++) host[v] = zh_portamento_state{w[v], w[n + v], w[2 * n + v]};
    return ZH_OK;
}
int zh_portamento_set_state(zh_portamento *m, const zh_portamento_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    const size_t n = m->n;
    std::vector<float> w(3 * n);
    for (uint32_t v = 0; v < m->n; v++) { w[v] = host[v].t; w[n + v] = host[v].last_value; w[2 * n + v] = host[v].start; }
    return upload_field(m->ctx, m->f(m->cur, 0), w);
}
int zh_portamento_paint(zh_portamento *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                        zh_bool note_id_changed, const zh_portamento_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || !curve_ok(p->curve)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;            // an empty span still applies newCurve / the instantaneous jump
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    const int c = m->cur;
    // few voices: frame ranges (1,024 / 4,096 / 16,384 / 32,768 voices: 14.8 / 16.5 / 17.7 / 27.1 us as one walk per voice, 3.4 / 6.0 / 13.0 / 26.4 us)
    const uint32_t ch = end > start ? zh_range_frames(m->n, end - start, ZF_PORTAMENTO_RANGES, 2048, 32768) : 0;
    if (ch) {
        const dim3 grid((m->n + 63) / 64, (end - start + ch - 1) / ch);
        ZH_ZF_LAUNCH(k_portamento_ranges, grid, dim3(64), m->f(c, 0), m->f(c, 1), m->f(c, 2), m->f(c ^ 1, 0), m->n, mk_img(outputs[0]), start, end, ch,
                     p->sample_rate, p->curve.tag, mk_f32(p->curve.duration), mk_f32(p->goal), mk_bool(p->note_on), mk_bool(p->prev_note_on),
                     mk_bool(note_id_changed));
        zh_flipper_painted(m);
        m->cur ^= 1;
        return zh_launch_status();
    }
    ZH_ZF_LAUNCH(k_portamento, seq_grid(m->n), dim3(kSeqBlock), m->f(c, 0), m->f(c, 1), m->f(c, 2), m->n, mk_img(outputs[0]),
                 start, end, p->sample_rate, p->curve.tag, mk_f32(p->curve.duration), mk_f32(p->goal), mk_bool(p->note_on),
                 mk_bool(p->prev_note_on), mk_bool(note_id_changed));
    return zh_launch_status();
}

// ------------------------------------------------------------------ Distortion
int zh_distortion_create(zh_ctx *ctx, uint32_t n, zh_distortion **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    *out = new (std::nothrow) zh_distortion{ctx, n};
    return *out ? ZH_OK : ZH_ERR_INVALID;
}
int zh_distortion_destroy(zh_distortion *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    delete m;
    return ZH_OK;
}
int zh_distortion_paint(zh_distortion *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                        zh_bool note_id_changed, const zh_distortion_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // Distortion.zig:35-37
    int rc = paint_check(m, start, end, outputs);
    if (rc) return rc;
    if (!p || p->type > ZH_DISTORTION_CLIP || !buf_covers(p->input, m->n, end)) return ZH_ERR_INVALID;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST;
    hipStream_t st = m->ctx->stream;
    const uint32_t chunks = (end - start + DIST_FC - 1) / DIST_FC;
    dim3 grid((m->n + 63) / 64, (chunks + 3) / 4);
    Img out = mk_img(outputs[0]);
    CImg in = mk_cimg(p->input);
    F32P ig = mk_f32(p->ingain), og = mk_f32(p->outgain), of = mk_f32(p->offset);
    // many voices, rows that take 16-byte accesses: the chunked shape (k_distortion_chunks).  Clip only by default: at 131,072 voices
    // 224 -> 178 us (6.0 TB/s); the overdrive (an atanf per sample: issue time beside the stream) is no faster in it at any chunk
    // length, 238.6 against 244-250 us (profiles/r06/ab_distortion.txt) -- it takes the shape only where the row is set by hand (tests).
    if ((long)m->n >= zh_form(ZF_DISTORTION_ROWS_MIN) && (p->type == ZH_DISTORTION_CLIP || zh_form_is_set(ZF_DISTORTION_ROWS_MIN)) && m->n % 4 == 0 && outputs[0].stride % 4 == 0 && p->input.stride % 4 == 0 &&
        ((uintptr_t)outputs[0].ptr & 15u) == 0 && ((uintptr_t)p->input.ptr & 15u) == 0) {
        const uint32_t nframes = end - start;
        // rows per wave (distortion_rc; 0 = 3): the setup is amortised over more rows, the loads in flight per wave grow with them
        const long rcf = zh_form(ZF_DISTORTION_RC);
        const int rc = rcf == 6 ? 6 : rcf == 8 ? 8 : 3;
        const dim3 g((m->n + 255) / 256, ((nframes + rc - 1) / rc + 3) / 4);
#define ZH_DISTC(ZF_, OD_, RC_) ZH_LAUNCH((k_distortion_chunks<ZF_, OD_, RC_>), g, dim3(256), 0, st, m->n, out, in, start, nframes, ig, og, of)
#define ZH_DISTR(ZF_, OD_) do { if (rc == 3) ZH_DISTC(ZF_, OD_, 3); else if (rc == 6) ZH_DISTC(ZF_, OD_, 6); else ZH_DISTC(ZF_, OD_, 8); } while (0)
        if (p->type == ZH_DISTORTION_OVERDRIVE) { if (zf) ZH_DISTR(true, true); else ZH_DISTR(false, true); }
        else { if (zf) ZH_DISTR(true, false); else ZH_DISTR(false, false); }
#undef ZH_DISTR
#undef ZH_DISTC
        return zh_launch_status();
    }
#define ZH_DIST(ZF_, OD_) ZH_LAUNCH((k_distortion<ZF_, OD_>), grid, dim3(256), 0, st, m->n, out, in, start, end, ig, og, of)
    if (p->type == ZH_DISTORTION_OVERDRIVE) { if (zf) ZH_DIST(true, true); else ZH_DIST(false, true); }
    else { if (zf) ZH_DIST(true, false); else ZH_DIST(false, false); }
#undef ZH_DIST
    return zh_launch_status();
}

}  // extern "C"
