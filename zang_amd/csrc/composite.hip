// composite.hip -- the reference's composite instruments as single fused kernels:
//   NiceInstrument  (examples/modules.zig:189-248)  PulseOsc -> x0.5 -> Filter(LP) -> x Envelope
//   PMOscInstrument (examples/modules.zig:80-128) over PhaseModOscillator (:6-77)
// The reference runs each stage as its own loop through `temps` buffers; every hand-off is an
// exact f32 store/load, so keeping the value in a register instead gives the same bits as
// the unfused composition -- provided every `zero` + `+=` pair is kept as `0.0f + x` and no
// multiply-add is fused (-ffp-contract=off).  The two temps of a NiceInstrument voice never
// touch HBM: per voice-sample the kernel writes 4 B (or nothing, in the mix variant).
#include "common.hip.h"
#include <memory>
#include "ring.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "seq.hip.h"
#include "envelope.hip.h"
#include "voices.hip.h"
#include "noise_jump.hip.h"
#define ZH_FILTER_TP_NOISE 1
#include "filter_tp.hip.h"
#include <vector>
#include <string.h>
#include <stdlib.h>

// basics.hip
int zh_mix_reserve(zh_ctx *ctx, size_t floats);
void zh_mix_pass2_launch(zh_ctx *ctx, uint32_t tiles, uint32_t nframes, float *dst, int zero_first);
void zh_mix_pass2_launch_at(zh_ctx *ctx, const float *partials, uint32_t tiles, uint32_t nframes, float *dst, int zero_first);
void zh_mix_pass2_wide_launch(zh_ctx *ctx, const float *partials, size_t channel_stride, uint32_t rows, uint32_t nframes, float *dst0,
                              float *dst1, int channels, int zero_first);
void zh_mix_pass2_wide_batch_launch(zh_ctx *ctx, const float *partials, size_t channel_stride, uint32_t rows, uint32_t nframes,
                                    float *const *dst0, float *const *dst1, uint32_t n_buffers, int channels, int zero_first);

struct zh_nice {
    zh_ctx *ctx;
    uint32_t n;
    float *color;                 // init(color), per voice
    uint32_t *cnt;                // osc
    float *fl, *fb;               // flt
    uint32_t *estate;             // env
    float *et, *elast, *estart;
    uint32_t *tp;                 // ZH_PAINT_TOLERANT scratch (k_nice_tp_a / _b), allocated by the first tolerant paint outside a capture
};

// The six state words per voice (carrier.t, modulator.t, envelope {state, t, last_value, start}) are one [6][n] block, double-
// buffered through the flipper registry (common.hip.h): the frame-range kernel reads the start state from the current block
// while its last range writes the end state into the other one, and the host flips.  view() points the named fields at the
// current block; every entry point calls it first (a graph launch may have flipped).
struct zh_pmosc : zh_flipper {
    float *release_duration;
    float *tc, *tm;               // carrier.t, modulator.t
    uint32_t *estate;
    float *et, *elast, *estart;
    void view() {
        uint32_t *b = cnt[cur];
        const size_t N = n;
        tc = reinterpret_cast<float *>(b); tm = reinterpret_cast<float *>(b + N); estate = b + 2 * N;
        et = reinterpret_cast<float *>(b + 3 * N); elast = reinterpret_cast<float *>(b + 4 * N); estart = reinterpret_cast<float *>(b + 5 * N);
    }
};

#include "nice.hip.h"
static uint32_t nice_pc_max() { return (uint32_t)zh_form(ZF_NICE_PC_MAX); }

template <bool ZF, int W>
__global__ void __launch_bounds__(kSeqBlock) k_nice(NiceArgs a, Img out, uint32_t start, uint32_t end) {
    using F = typename LaneT<W>::F;
    const uint32_t v = (blockIdx.x * kSeqBlock + threadIdx.x) * W;
    if (v >= a.V) return;
    NiceLaneT<W> n;
    nice_load<W>(n, a, v);
    const float *const *no_in = nullptr;
    PulseRoll roll = 0;
    if constexpr (W == 1) {
        // 8-frame chunks (frame_loop_gen): one in which no voice of the wave is inside a timed envelope stage multiplies by one
        // constant per voice (sustain, idle: the envelope's frame changes nothing), one in which no stage can end runs the
        // envelope without its stage-end test; the others take the full frame
        n.roll_begin(roll);
        bool flat = false;
        float e0c = 0.0f;
        frame_loop_gen2<8, ZF>(out.p, v, out.stride, start, end,
            [&](uint32_t) ZH_INLINE_LAMBDA {
                flat = !zany_wave(n.env.mode == ENV_MODE_TOWARD);
                if (flat) { e0c = n.env_quiet(); return 1; }
                if (!n.env.quiet(8)) return 0;
                return __all(n.env.mode == ENV_MODE_TOWARD) ? 2 : 1;      // 2: every voice inside a stage, nothing to select
            },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
                const float t1 = n.template tail_filter<ZF>(n.osc_next(roll));   // (ZERO_FIRST: the caller adds val to 0, see tail_filter)
                val = (flat ? e0c : n.env.frame_masked_quiet()) * t1;   // NiceLane::tail
                return true;
            },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
                const float t1 = n.template tail_filter<ZF>(n.osc_next(roll));
                val = n.env.frame_masked_all_toward_quiet() * t1;
                return true;
            },
            [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = n.template tail<ZF>(n.osc_next(roll)); return true; });
    } else {
        frame_loop<8, ZF, 0, W>(out.p, v, out.stride, no_in, nullptr, start, end, [&](uint32_t, const F (&)[1], F &val) ZH_INLINE_LAMBDA {
            val = n.frame();
            return zmask<typename LaneT<W>::M>(true);
        });
    }
    nice_store<W>(n, a, v);
}

// Few voices (fewer waves than the chip has SIMDs): the frame's three chains -- oscillator (phase counter),
// envelope (clock + state machine) and filter ((l, b) through every sample) -- only meet in values, never in
// state.  One workgroup of THREE waves owns 64 voices: wave 0 produces the oscillator samples and wave 1 the
// envelope values one tile of 32 frames ahead into LDS ([frame][voice]); wave 2 filters the previous tile,
// multiplies and writes the image; one barrier per 32 frames.  The longest chain (the filter's ~27
// instructions) sets the pace instead of the sum (~64).  Lanes past the last voice run voice V-1 again in all
// three waves, so their stores repeat V-1's values at V-1's address and nothing inside the chains is masked.
// Same per-voice operations in the same order => same bits as k_nice.
template <bool ZF>
__global__ void __launch_bounds__(192) k_nice_pc(NiceArgs a, Img out, uint32_t start, uint32_t end) {
    constexpr uint32_t CH = 32;
    __shared__ float osc_t[2][CH][64], env_t[2][CH][64];
    const uint32_t lane = threadIdx.x & 63, role = threadIdx.x >> 6;   // 0 oscillator, 1 envelope, 2 filter
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < a.V;
    const uint32_t vc = live ? v : a.V - 1;
    const uint32_t n_frames = end - start, nchunks = (n_frames + CH - 1) / CH;
    NiceLane n;
    nice_load<1>(n, a, vc);
    PulseRoll roll;
    n.roll_begin(roll);
    const uint32_t voff = vc * 4u, orow = (uint32_t)out.stride * 4u;
    // software pipeline, one barrier per step in uniform control flow: in step c the producers fill tile c while the
    // filter wave drains tile c-1 (the other LDS buffer); nchunks + 1 steps
    for (uint32_t c = 0; c <= nchunks; c++) {
        if (role == 0 && c < nchunks) {
            const uint32_t nf = min(CH, n_frames - c * CH);
            float (*t)[64] = osc_t[c & 1];
            auto one = [&](uint32_t k) ZH_INLINE_LAMBDA { t[k][lane] = n.osc_next(roll); };
            if (nf == CH) {
#pragma unroll 8
                for (uint32_t k = 0; k < CH; k++) one(k);
            } else {
                for (uint32_t k = 0; k < nf; k++) one(k);
            }
        } else if (role == 1 && c < nchunks) {
            const uint32_t nf = min(CH, n_frames - c * CH);
            float (*t)[64] = env_t[c & 1];
            if (nf == CH) {
                if (!zany_wave(n.env.mode == ENV_MODE_TOWARD)) {       // (see k_nice_pc4)
                    const float e0 = n.env_quiet();
#pragma unroll 8
                    for (uint32_t k = 0; k < CH; k++) t[k][lane] = e0;
                } else if (n.env.quiet(CH)) {
#pragma unroll 8
                    for (uint32_t k = 0; k < CH; k++) t[k][lane] = n.env.frame_masked_quiet();
                } else {
#pragma unroll 8
                    for (uint32_t k = 0; k < CH; k++) t[k][lane] = n.tail_env();
                }
            } else {
                for (uint32_t k = 0; k < nf; k++) t[k][lane] = n.tail_env();
            }
        } else if (role == 2 && c > 0) {
            const uint32_t d = c - 1, nf = min(CH, n_frames - d * CH);
            const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + d * CH);
            const float (*to)[64] = osc_t[d & 1];
            const float (*te)[64] = env_t[d & 1];
            auto one = [&](uint32_t k, float t0, float e0, float o) ZH_INLINE_LAMBDA {
                const float t1 = n.tail_filter(t0);
                zrow_store<1>(ro, voff, k * orow, o + e0 * t1);        // multiply :246: out += temps[0]*temps[1]
            };
            if (nf == CH) {
                float xo[CH], xe[CH], oc[CH];                          // both tile columns first: one LDS wait per tile
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) { xo[k] = to[k][lane]; xe[k] = te[k][lane]; oc[k] = ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow); }
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) one(k, xo[k], xe[k], oc[k]);
            } else {
                for (uint32_t k = 0; k < nf; k++) one(k, to[k][lane], te[k][lane], ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
            }
        }
        __syncthreads();
    }
    if (live && role == 0) a.cnt[v] = n.cnt;
    if (live && role == 1) { a.estate[v] = n.env.state; a.et[v] = n.env.t; a.elast[v] = n.env.last_value; a.estart[v] = n.env.start; }
    if (live && role == 2) { a.fl[v] = n.l; a.fb[v] = n.b; }
}

// k_nice_pc with the filter wave reduced to the recurrence itself: the oscillator wave also adds the filter's input offset
// (in = temps[0] + fcdcoffset, Filter.zig:135 -- a function of the sample alone), the filter wave runs svf_core (15 VALU
// instructions per sample) and hands (l, b, h) on through LDS, and a FOURTH wave does the low-pass mix, the multiply with the
// envelope, the `+=` and the store, two tiles behind the producers.  Same operations on the same values => same bits.
// Measured at 4,096 voices with one wave's arithmetic compiled out at a time: 71 us as four waves with the full envelope
// frame, 57.5 without the envelope's (it was the slowest wave: hence the quiet tiles below), then 57 -> 53.4 without the
// oscillator's, 54.7 without the filter's, 42.9 without both (LDS traffic, barriers and the writer).
// Round 2, later: what the filter wave costs beyond its 15 instructions is its LDS traffic (k_filter_pc, modules.hip: an LDS
// instruction costs a lone wave about three VALU issues), so the tiles are float4 -- four frames of a lane side by side, one
// 16-byte access per four frames -- the filter wave fetches its next tile while it computes (the producers run a step further
// ahead), and every role runs its own copy of the step loop (as branches of one loop body the roles' register arrays were
// merged at the join behind an `s_waitcnt vmcnt(0)`: the writer waited out its stores every step).
template <bool ZF>
__global__ void __launch_bounds__(256) k_nice_pc4(NiceArgs a, Img out, uint32_t start, uint32_t end) {
    constexpr uint32_t CH = 32, Q = CH / 4;
    __shared__ float4 in_q[2][Q][64], env_q[2][Q][64], l_q[2][Q][64], b_q[2][Q][64];
    const uint32_t lane = threadIdx.x & 63, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 oscillator, 1 envelope, 2 filter, 3 writer
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < a.V;
    const uint32_t vc = live ? v : a.V - 1;
    const uint32_t n_frames = end - start, nchunks = (n_frames + CH - 1) / CH;
    NiceLane n;
    nice_load<1>(n, a, vc);
    const uint32_t voff = vc * 4u, orow = (uint32_t)out.stride * 4u;
    auto frames = [&](uint32_t c) ZH_INLINE_LAMBDA { return c < nchunks ? min(CH, n_frames - c * CH) : 0u; };
    // frame k of this lane inside a tile of float4 (the scalar path of a partial last tile)
    auto at = [&](float4 (*t)[64], uint32_t k) ZH_INLINE_LAMBDA -> float & { return reinterpret_cast<float *>(&t[k >> 2][lane])[k & 3]; };
    // Step c: the oscillator wave publishes tile c; the filter wave computes tile c - 2 out of registers while it fetches
    // tile c - 1; the envelope wave (which reads nobody's output) paints tile c - 2; the writer mixes and stores tile c - 3.
    const uint32_t last = nchunks + 2;
    if (role == 0) {
        PulseRoll roll;
        n.roll_begin(roll);
        for (uint32_t c = 0; c <= last; c++) {
            if (c < nchunks) {
                const uint32_t nf = frames(c);
                float4 (*t)[64] = in_q[c & 1];
                if (nf == CH) {
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) {
                        const float x0 = n.osc_next(roll) + kSvfDcOffset, x1 = n.osc_next(roll) + kSvfDcOffset;
                        const float x2 = n.osc_next(roll) + kSvfDcOffset, x3 = n.osc_next(roll) + kSvfDcOffset;
                        t[q][lane] = make_float4(x0, x1, x2, x3);
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++) at(t, k) = n.osc_next(roll) + kSvfDcOffset;
                }
            }
            __syncthreads();
        }
        if (live) a.cnt[v] = n.cnt;
    } else if (role == 1) {
        for (uint32_t c = 0; c <= last; c++) {
            if (c >= 2 && c - 2 < nchunks) {
                const uint32_t d = c - 2, nf = frames(d);
                float4 (*t)[64] = env_q[d & 1];
                if (nf == CH) {
                    // the envelope wave was the slowest of the four (compiled out: 71 -> 57 us): a tile in which no voice is inside
                    // a timed stage is one constant per voice, one in which no stage can end runs without the stage-end test
                    if (!zany_wave(n.env.mode == ENV_MODE_TOWARD)) {
                        const float e0 = n.env_quiet();
#pragma unroll
                        for (uint32_t q = 0; q < Q; q++) t[q][lane] = make_float4(e0, e0, e0, e0);
                    } else if (n.env.quiet(CH)) {
#pragma unroll
                        for (uint32_t q = 0; q < Q; q++) {
                            const float e0 = n.env.frame_masked_quiet(), e1 = n.env.frame_masked_quiet();
                            const float e2 = n.env.frame_masked_quiet(), e3 = n.env.frame_masked_quiet();
                            t[q][lane] = make_float4(e0, e1, e2, e3);
                        }
                    } else {
#pragma unroll
                        for (uint32_t q = 0; q < Q; q++) {
                            const float e0 = n.tail_env(), e1 = n.tail_env(), e2 = n.tail_env(), e3 = n.tail_env();
                            t[q][lane] = make_float4(e0, e1, e2, e3);
                        }
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++) at(t, k) = n.tail_env();
                }
            }
            __syncthreads();
        }
        if (live) { a.estate[v] = n.env.state; a.et[v] = n.env.t; a.elast[v] = n.env.last_value; a.estart[v] = n.env.start; }
    } else if (role == 2) {
        float4 fa[Q], fb[Q];                                          // the tile in hand / the next one
#pragma unroll
        for (uint32_t q = 0; q < Q; q++) fa[q] = fb[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto step = [&](uint32_t c, float4 (&cur)[Q], float4 (&nxt)[Q]) ZH_INLINE_LAMBDA {
            if (c == 0 || c > nchunks + 1) return;
            const float4 (*tn)[64] = in_q[(c - 1) & 1];              // (complete only if that tile is a whole one: otherwise unused)
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) nxt[q] = tn[q][lane];
            if (c == 1) return;
            const uint32_t d = c - 2, nf = frames(d);
            float4 (*tl)[64] = l_q[d & 1], (*tb)[64] = b_q[d & 1];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    const SvfOut s0 = svf_core(n.l, n.b, cur[q].x, n.cut, n.res);
                    const SvfOut s1 = svf_core(n.l, n.b, cur[q].y, n.cut, n.res);
                    const SvfOut s2 = svf_core(n.l, n.b, cur[q].z, n.cut, n.res);
                    const SvfOut s3 = svf_core(n.l, n.b, cur[q].w, n.cut, n.res);
                    tl[q][lane] = make_float4(s0.l, s1.l, s2.l, s3.l);     // (h is not needed: dsp.hip.h svf_lowpass_into_zero)
                    tb[q][lane] = make_float4(s0.b, s1.b, s2.b, s3.b);
                }
            } else {                                                  // (the last tile: the oscillator wave has stopped, its buffer stays)
                float4 (*ti)[64] = in_q[d & 1];
                for (uint32_t k = 0; k < nf; k++) {
                    const SvfOut s = svf_core(n.l, n.b, at(ti, k), n.cut, n.res);
                    at(tl, k) = s.l; at(tb, k) = s.b;
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
        if (live) { a.fl[v] = n.l; a.fb[v] = n.b; }
    } else {
        float bn[CH];                                                 // the output rows of the tile after the one in hand
        for (uint32_t c = 0; c <= last; c++) {
            if (c > 2) {
                const uint32_t d = c - 3, nf = frames(d);
                const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + d * CH);
                float4 (*tl)[64] = l_q[d & 1], (*tb)[64] = b_q[d & 1], (*te)[64] = env_q[d & 1];
                auto one = [&](uint32_t k, float l, float b, float e0, float o) ZH_INLINE_LAMBDA {
                    const float t1 = svf_lowpass_into_zero(l, b);              // NiceLane::tail_filter's mix
                    zrow_store<1>(ro, voff, k * orow, o + e0 * t1);            // multiply :246: out += temps[0]*temps[1]
                };
                if (nf == CH) {
                    float4 xl[Q], xb[Q], xe[Q];
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) { xl[q] = tl[q][lane]; xb[q] = tb[q][lane]; xe[q] = te[q][lane]; }
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) {
                        one(4 * q, xl[q].x, xb[q].x, xe[q].x, ZF ? 0.0f : bn[4 * q]);
                        one(4 * q + 1, xl[q].y, xb[q].y, xe[q].y, ZF ? 0.0f : bn[4 * q + 1]);
                        one(4 * q + 2, xl[q].z, xb[q].z, xe[q].z, ZF ? 0.0f : bn[4 * q + 2]);
                        one(4 * q + 3, xl[q].w, xb[q].w, xe[q].w, ZF ? 0.0f : bn[4 * q + 3]);
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++) one(k, at(tl, k), at(tb, k), at(te, k), ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
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
}

#include "nice_mix.hip.h"
// ------------------------------------------------------------------ NiceInstrument, ZH_PAINT_TOLERANT (few voices)
// The time-parallel Filter of filter_tp.hip.h inside the fused voice: the oscillator's counter at any frame is exact integer
// arithmetic (cnt + j * ifreq), the filter's parameters are constant over a paint (cutoffFromFrequency(freq * 8), res 0.7), so a
// chunk can compute its filter's zero-state response on its own (pass A) and, after the scan, run the reference's own frame from
// the chunk's start state (pass B).  The envelope is a clock and a state machine that depend on nothing but time: the chunk-0
// lane of every voice walks it over the WHOLE span once in pass A (eight clock steps and one curve evaluation per quiet chunk,
// frame by frame around a stage end) and leaves its four state words at every chunk boundary -- one walk of the span per voice
// instead of one replay per chunk.  Pass B re-derives the running stage from those words (EnvLaneT::resolve: a function of the
// state, the clock and the params).  Oscillator, envelope and both their states are exact; the filter and the product carry the
// tolerant Filter's error (<= 1e-5 of the voice's peak, measured ~1e-6).
// Scratch words per voice: 7 (the state at span start: pass B's lanes load it while the last chunk's lane stores the live arrays)
// + 4 per chunk (envelope) + 2 per chunk + 2 (filter e_j, slot 0 = start state).
constexpr uint32_t kNiceTpMaxChunks = kTpMaxChunks;
constexpr uint32_t kNiceTpWords = 7 + 4 * kNiceTpMaxChunks + 2 * (kNiceTpMaxChunks + 1);
struct NiceTpArgs {
    NiceArgs a;                  // the live state arrays + params
    uint32_t *tp;                // [kNiceTpWords][V]
    uint32_t start, end, L;
    Img out;
};
__device__ __forceinline__ uint32_t *nice_tp_state0(const NiceTpArgs &t, uint32_t w) { return t.tp + (size_t)w * t.a.V; }
__device__ __forceinline__ uint32_t *nice_tp_env(const NiceTpArgs &t, uint32_t j, uint32_t w) { return t.tp + (size_t)(7 + 4 * j + w) * t.a.V; }
__device__ __forceinline__ float2 *nice_tp_e(const NiceTpArgs &t, uint32_t slot) {
    return reinterpret_cast<float2 *>(t.tp + (size_t)(7 + 4 * kNiceTpMaxChunks) * t.a.V) + (size_t)slot * t.a.V;
}
__device__ __forceinline__ void nice_tp_put_env(const NiceTpArgs &t, uint32_t j, uint32_t v, const NiceLane &n) {
    nice_tp_env(t, j, 0)[v] = n.env.state; nice_tp_env(t, j, 1)[v] = __builtin_bit_cast(uint32_t, n.env.t);
    nice_tp_env(t, j, 2)[v] = __builtin_bit_cast(uint32_t, n.env.last_value); nice_tp_env(t, j, 3)[v] = __builtin_bit_cast(uint32_t, n.env.start);
}

// A voice whose filter cutoff is near zero (filter_tp.hip.h kTpExactCutBelow; here cutoffFromFrequency(8 x freq): a voice below ~4 Hz)
// takes no chunk start state from the scan: its lane replays the filter over the frames before its chunk from the span's start
// state -- the reference's own recurrence over the same oscillator samples (the counter of any frame is exact), bit for bit.
__device__ __forceinline__ void nice_tp_exact_start(NiceLane &n, float2 s0, uint32_t cnt_start, uint32_t frames_before) {
    if (!(n.cut < kTpExactCutBelow)) return;
    float l = s0.x, b = s0.y;
    uint32_t c = cnt_start;
    for (uint32_t f = 0; f < frames_before; f++) {
        svf_step(l, b, n.osc(c), n.cut, n.res);                        // NiceLane::tail_filter's recurrence (k_nice_tp_a)
        c += n.k.ifreq;
    }
    n.l = l; n.b = b;
}
// grid: x = 256-voice groups, y = chunk; block = 256
__global__ void __launch_bounds__(256) k_nice_tp_a(const NiceTpArgs t) {
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    const NiceArgs &a = t.a;
    if (v >= a.V) return;
    NiceLane n;
    nice_load(n, a, v);                                                // the span's start state, begin() with this paint's params
    const uint32_t cnt0 = n.cnt;
    const uint32_t f0 = min(t.start + j * t.L, t.end), f1 = min(f0 + t.L, t.end);
    if (j == 0) {
        // the start state as the paint found it (before begin()): what pass B's lanes begin() from
        nice_tp_state0(t, 0)[v] = a.cnt[v]; nice_tp_state0(t, 1)[v] = __builtin_bit_cast(uint32_t, a.fl[v]); nice_tp_state0(t, 2)[v] = __builtin_bit_cast(uint32_t, a.fb[v]);
        nice_tp_state0(t, 3)[v] = a.estate[v]; nice_tp_state0(t, 4)[v] = __builtin_bit_cast(uint32_t, a.et[v]);
        nice_tp_state0(t, 5)[v] = __builtin_bit_cast(uint32_t, a.elast[v]); nice_tp_state0(t, 6)[v] = __builtin_bit_cast(uint32_t, a.estart[v]);
        nice_tp_e(t, 0)[v] = make_float2(n.l, n.b);
    }
    // the filter's zero-state response over this chunk's oscillator samples
    float l = 0.0f, b = 0.0f;
    uint32_t c = cnt0 + (f0 - t.start) * n.k.ifreq;                    // exact: the counter of frame f0 (a silent voice's ifreq is 0)
#pragma unroll 8
    for (uint32_t f = f0; f < f1; f++) {
        svf_step(l, b, n.osc(c), n.cut, n.res);                        // NiceLane::tail_filter's recurrence
        c += n.k.ifreq;
    }
    nice_tp_e(t, j + 1)[v] = make_float2(l, b);
    if (j != 0) return;
    // the envelope over the whole span, its state left at every chunk boundary (the first chunk's is begin()'s own)
    nice_tp_put_env(t, 0, v, n);
    const uint32_t nchunks = (t.end - t.start + t.L - 1) / t.L;
    for (uint32_t q = 1; q < nchunks; q++) {
        // no voice of the wave inside a timed stage (a held note's sustain, an idle voice): a chunk of frames changes nothing;
        // no stage can end within the chunk: the clocks step, one curve evaluation per eight frames, no test in between;
        // else eight frames at a time with the test, frame by frame around a stage end
        if (zany_wave(n.env.mode == ENV_MODE_TOWARD)) {
            if (n.env.quiet((int)t.L)) {
                // (the curve once, for the state the chunk leaves: last_value is rewritten from the clock by whatever reads it next)
                for (uint32_t i = 8; i < t.L; i += 8) n.env.template skip_clock<8>();
                n.env.template skip_quiet<8>();
            } else {
                for (uint32_t i = 0; i < t.L; i += 8) {
                    if (n.env.quiet(8)) n.env.template skip_quiet<8>();
                    else {
#pragma unroll
                        for (int k = 0; k < 8; k++) (void)n.env.frame_masked();
                    }
                }
            }
        }
        nice_tp_put_env(t, q, v, n);
    }
}

template <bool ZF>
__global__ void __launch_bounds__(256) k_nice_tp_b(const NiceTpArgs t) {
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    const NiceArgs &a = t.a;
    if (v >= a.V) return;
    const uint32_t f0 = min(t.start + j * t.L, t.end), f1 = min(f0 + t.L, t.end);
    if (f1 == f0) return;
    NiceArgs a0 = a;                                                   // load from the copy pass A made of the start state
    a0.cnt = nice_tp_state0(t, 0); a0.fl = reinterpret_cast<float *>(nice_tp_state0(t, 1)); a0.fb = reinterpret_cast<float *>(nice_tp_state0(t, 2));
    a0.estate = nice_tp_state0(t, 3); a0.et = reinterpret_cast<float *>(nice_tp_state0(t, 4));
    a0.elast = reinterpret_cast<float *>(nice_tp_state0(t, 5)); a0.estart = reinterpret_cast<float *>(nice_tp_state0(t, 6));
    NiceLane n;
    nice_load(n, a0, v);
    const uint32_t cnt_f0 = n.cnt;
    n.cnt = n.cnt + (f0 - t.start) * n.k.ifreq;
    {   // the filter's state at the chunk's first frame
        const float2 s0 = nice_tp_e(t, 0)[v];
        n.l = s0.x; n.b = s0.y;
        const float2 *e = nice_tp_e(t, 1) + v;
        const size_t V = a.V;
        svf_scan<kNiceTpMaxChunks - 1>(n.l, n.b, n.cut, n.res, t.L, j, [&](uint32_t i) ZH_INLINE_LAMBDA { return e[(size_t)i * V]; });
        nice_tp_exact_start(n, s0, cnt_f0, f0 - t.start);
    }
    // the envelope at the chunk's first frame: the four state words of pass A's walk, the running stage re-derived from them
    n.env.state = nice_tp_env(t, j, 0)[v]; n.env.t = __builtin_bit_cast(float, nice_tp_env(t, j, 1)[v]);
    n.env.last_value = __builtin_bit_cast(float, nice_tp_env(t, j, 2)[v]); n.env.start = __builtin_bit_cast(float, nice_tp_env(t, j, 3)[v]);
    n.env.resolve(true);
    n.env.refresh_derived();
    // the frames, exactly as k_nice paints them
    PulseRoll roll;
    n.roll_begin(roll);
    bool flat = false;
    float e0c = 0.0f;
    frame_loop_gen2<8, ZF>(t.out.p, v, t.out.stride, f0, f1,
        [&](uint32_t) ZH_INLINE_LAMBDA {
            flat = !zany_wave(n.env.mode == ENV_MODE_TOWARD);
            if (flat) { e0c = n.env_quiet(); return 1; }
            if (!n.env.quiet(8)) return 0;
            return __all(n.env.mode == ENV_MODE_TOWARD) ? 2 : 1;
        },
        [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
            const float t1 = n.template tail_filter<ZF>(n.osc_next(roll));
            val = (flat ? e0c : n.env.frame_masked_quiet()) * t1;
            return true;
        },
        [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
            const float t1 = n.template tail_filter<ZF>(n.osc_next(roll));
            val = n.env.frame_masked_all_toward_quiet() * t1;
            return true;
        },
        [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = n.template tail<ZF>(n.osc_next(roll)); return true; });
    if (f1 == t.end) nice_store(n, a, v);                              // whoever painted the span's last frame leaves the states
}

// The mixdown workload with the flag (zh_nice_paint_mix[_stereo], few voices): pass A is k_nice_tp_a; pass B sets a chunk's voices up
// like k_nice_tp_b and runs the fused mixdown's own frame code (nice_mix_frames) over the chunk's frames -- 32-frame tiles, so the
// chunks are multiples of 32 frames -- one partial row per wave, then the ordinary second pass.
template <int C, bool ROLL>
__global__ void __launch_bounds__(256) k_nice_mix_tp_b(const NiceTpArgs t, float *__restrict__ partials, F32P gain_l, F32P gain_r) {
    __shared__ float tile_all[4][MIXF][MIXS];
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    const NiceArgs &a = t.a;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*tile)[MIXS] = tile_all[wave];
    const uint32_t nframes = t.end - t.start;
    const uint32_t wave_global = blockIdx.x * 4 + wave, rows = gridDim.x * 4;
    const size_t channel_stride = (size_t)((nframes + kMixGroupFrames - 1) / kMixGroupFrames) * rows * kMixGroupFrames;
    const bool live = v < a.V;
    const uint32_t vc = live ? v : a.V - 1;
    const uint32_t f0 = min(t.start + j * t.L, t.end), f1 = min(f0 + t.L, t.end);
    NiceArgs a0 = a;                                                   // the start state as pass A copied it
    a0.cnt = nice_tp_state0(t, 0); a0.fl = reinterpret_cast<float *>(nice_tp_state0(t, 1)); a0.fb = reinterpret_cast<float *>(nice_tp_state0(t, 2));
    a0.estate = nice_tp_state0(t, 3); a0.et = reinterpret_cast<float *>(nice_tp_state0(t, 4));
    a0.elast = reinterpret_cast<float *>(nice_tp_state0(t, 5)); a0.estart = reinterpret_cast<float *>(nice_tp_state0(t, 6));
    NiceLane n;
    nice_load(n, a0, vc);
    const uint32_t cnt_f0 = n.cnt;
    n.cnt = n.cnt + (f0 - t.start) * n.k.ifreq;
    {
        const float2 s0 = nice_tp_e(t, 0)[vc];
        n.l = s0.x; n.b = s0.y;
        const float2 *e = nice_tp_e(t, 1) + vc;
        const size_t V = a.V;
        svf_scan<kNiceTpMaxChunks - 1>(n.l, n.b, n.cut, n.res, t.L, j, [&](uint32_t i) ZH_INLINE_LAMBDA { return e[(size_t)i * V]; });
        nice_tp_exact_start(n, s0, cnt_f0, f0 - t.start);
    }
    n.env.state = nice_tp_env(t, j, 0)[vc]; n.env.t = __builtin_bit_cast(float, nice_tp_env(t, j, 1)[vc]);
    n.env.last_value = __builtin_bit_cast(float, nice_tp_env(t, j, 2)[vc]); n.env.start = __builtin_bit_cast(float, nice_tp_env(t, j, 3)[vc]);
    n.env.resolve(true);
    n.env.refresh_derived();
    if (!live) nice_silence(n);
    const uint32_t rf = lane & (MIXF - 1), rh = lane >> 5;
    PulseRoll roll;
    n.roll_begin(roll);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 g2[C == 2 ? 32 : 1];
    nice_mix_gains<C>(g2, gain_l, gain_r, live, v, wave, rh);
    nice_mix_frames<C, ROLL, 0>(n, roll, g2, tile, partials, channel_stride, rows, wave_global, t.start, f1, lane, rf, rh, nullptr, 0, f0);
    if (live && f1 == t.end && f1 > f0) nice_store(n, a, v);
}

// ------------------------------------------------------------------ PMOscInstrument voice
struct PMOscArgs {
    float *release_duration;
    float *tc, *tm;
    uint32_t *estate;
    float *et, *elast, *estart;
    uint32_t V;
    float sample_rate;
    F32P freq;
    BoolP note_on, nic;
};


struct PMLane {
    float tc, tm;                 // carrier.t, modulator.t
    float mod_freq, inv_sr, t_step;
    EnvLaneCubed env;                                                 // all three curves are cubed (:118-125)

    __device__ __forceinline__ void begin(float sample_rate, float freq, float release_duration, bool note_on, bool new_note) {
        env.sample_rate = sample_rate;                                 // examples/modules.zig:118-125
        env.sustain_volume = 0.5f;
        env.attack = CurveP{ZH_CURVE_CUBED, 0.025f};
        env.decay = CurveP{ZH_CURVE_CUBED, 0.1f};
        env.release = CurveP{ZH_CURVE_CUBED, release_duration};
        env.note_on = note_on;
        env.begin(new_note);
        mod_freq = freq * 1.0f;                                        // set(temps[0], freq * ratio), ratio = 1 (:45)
        inv_sr = 1.0f / sample_rate;                                   // modulator: controlled-frequency path (SineOsc.zig:66)
        t_step = freq / sample_rate;                                   // carrier: constant-frequency path (:44)
    }

    // One frame in two parts.  step(): everything that carries state from frame to frame -- both phase
    // accumulators and the envelope -- returning what this frame's sample is made from.  value(): the
    // two sines and the products, a pure function of those three numbers (so it can be evaluated for
    // many frames at once, k_pmosc_spans_wave).
    // step = step_phase + step_env: two chains that never read each other's state.
    __device__ __forceinline__ void step_phase(float &tm_i, float &tc_i) {
        tm_i = tm;
        tm += mod_freq * inv_sr;                                       // modulator: t += freq_buf[i] * inv_sr (:59-63)
        tc_i = tc;
        tc += t_step;                                                  // carrier: t += t_step (:69-74)
    }
    __device__ __forceinline__ float step_env() {
        return env.frame_masked();                                     // envelope -> temps[1] (zeroed)   (:117-125)
    }
    __device__ __forceinline__ void step(float &tm_i, float &tc_i, float &e0) {
        step_phase(tm_i, tc_i);
        e0 = step_env();
    }
    // SINMODE (voices.hip.h sine_osc_sin): 1 musl's sinf, 0 without its rare-path branch; + 2 = ZH_PAINT_TOLERANT: the CARRIER's
    // sine in f32.  The modulator's stays musl's: its value is added to the carrier's phase BEFORE that sum is rounded to f32, and
    // with the phase at 64 (a 3 kHz note at the end of a buffer) one ulp of the sum is 4.8e-5 of a cycle's argument -- a modulator
    // off by 3e-7 flips that rounding for 4 % of the samples (tests/test_gpu_tolerant.py measured 5e-5 before this was kept exact).
    template <int SINMODE = 1>
    static __device__ __forceinline__ float value(float tm_i, float tc_i, float e0) {
        const float m = 0.0f + sine_osc_sin<(SINMODE & 1)>(tm_i + 0.0f);   // modulator.paint -> temps[1] (zeroed): sin(t + 0.0)
        const float ph = 0.0f + m * 1.0f;                              // temps[0] = 0 + temps[1] * multiplier (1.0)   (:64-66)
        const float c = 0.0f + sine_osc_sin<(SINMODE & 2) ? 2 : SINMODE>(tc_i + ph);   // carrier.paint -> temps[1] (zeroed): sin(t + phase[i])
        const float osc = 0.0f + c;                                    // PhaseModOscillator output (zeroed) += temps[1]   (:75)
        return osc * e0;                                               // multiply(out, temps[0], temps[1]) :126
    }
    template <int SINMODE = 1>
    __device__ __forceinline__ float frame() {
        float tm_i, tc_i, e0;
        step(tm_i, tc_i, e0);
        return value<SINMODE>(tm_i, tc_i, e0);
    }
    // true (wave-wide) when neither sine's argument can reach zsinf's rare path in the next `frames` frames: both phases move
    // by a constant per frame, the carrier's phase offset is a sine (|ph| <= 1); NaN compares false
    __device__ __forceinline__ bool small_args(float frames) const {
        const bool ok = __builtin_fabsf(tm) + frames * __builtin_fabsf(mod_freq * inv_sr) < kSineOscSmallT &&
                        __builtin_fabsf(tc) + frames * __builtin_fabsf(t_step) + 1.0f < kSineOscSmallT;
        return __builtin_amdgcn_ballot_w64(!ok) == 0;
    }

    // end of one paint call: envelope cascade, and both SineOsc `t - trunc(t)` wraps (SineOsc.zig:40)
    __device__ __forceinline__ void end() {
        tc = tc - truncf(tc);
        tm = tm - truncf(tm);
    }
};

__device__ __forceinline__ void pm_load(PMLane &n, const PMOscArgs &a, uint32_t v) {
    n.tc = a.tc[v]; n.tm = a.tm[v];
    n.env.state = a.estate[v]; n.env.t = a.et[v]; n.env.last_value = a.elast[v]; n.env.start = a.estart[v];
}
__device__ __forceinline__ void pm_store(const PMLane &n, const PMOscArgs &a, uint32_t v) {
    a.tc[v] = n.tc; a.tm[v] = n.tm;
    a.estate[v] = n.env.state; a.et[v] = n.env.t; a.elast[v] = n.env.last_value; a.estart[v] = n.env.start;
}

// The frames [f0, f1) of one voice column.  8-frame chunks: where no sine argument can reach zsinf's rare path and no
// envelope stage can end, the frame is one straight-line block -- branch-free sines, and the envelope as a per-voice
// constant (no voice of the wave inside a timed stage), without its selects (every voice inside one) or with them; every
// other chunk takes the general frame.
// TOL (ZH_PAINT_TOLERANT): the carrier's sine by zsinf_tol (zmath.hip.h); see PMLane::value.
template <bool ZF, bool TOL = false>
__device__ __forceinline__ void pm_paint_frames(PMLane &n, const Img &out, uint32_t v, uint32_t f0, uint32_t f1) {
    bool flat = false;
    float e0c = 0.0f;
    constexpr int SM_FAST = TOL ? 2 : 0, SM_ANY = TOL ? 3 : 1;
    frame_loop_gen2<8, ZF>(out.p, v, out.stride, f0, f1,
        [&](uint32_t) ZH_INLINE_LAMBDA {
            if (!n.small_args(8.0f) || !n.env.quiet(8)) return 0;
            flat = !zany_wave(n.env.mode == ENV_MODE_TOWARD);
            if (flat) { e0c = n.env.frame_masked_quiet(); return 1; }     // (changes nothing where no voice is in a stage)
            return __all(n.env.mode == ENV_MODE_TOWARD) ? 2 : 1;
        },
        [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
            float tm_i, tc_i;
            n.step_phase(tm_i, tc_i);
            val = PMLane::value<SM_FAST>(tm_i, tc_i, flat ? e0c : n.env.frame_masked_quiet());
            return true;
        },
        [&](uint32_t, float &val) ZH_INLINE_LAMBDA {
            float tm_i, tc_i;
            n.step_phase(tm_i, tc_i);
            val = PMLane::value<SM_FAST>(tm_i, tc_i, n.env.frame_masked_all_toward_quiet());
            return true;
        },
        [&](uint32_t, float &val) ZH_INLINE_LAMBDA { val = n.template frame<SM_ANY>(); return true; });
}

template <bool ZF, bool TOL = false>
__global__ void __launch_bounds__(kSeqBlock) k_pmosc(PMOscArgs a, Img out, uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= a.V) return;
    PMLane n;
    pm_load(n, a, v);
    n.begin(a.sample_rate, a.freq.get(v), a.release_duration[v], a.note_on.get(v), a.nic.get(v));
    pm_paint_frames<ZF, TOL>(n, out, v, start, end);
    n.end();
    pm_store(n, a, v);
}

// Few voices: the span as frame ranges at once, one wave per (64 voices, range).  What a PMOscInstrument voice carries from
// frame to frame is two f32 phase accumulators and the envelope's state machine -- ~16 instructions per frame -- while the
// frame's value is two musl sines on top of them (~150): a range REPLAYS the walk of the frames before it (step(), values
// discarded) and then paints its own frames exactly like k_pmosc.  The range that ends the span writes the end state to
// `next` ([6][V]), the other half of the module's double buffer (the host flips: zh_flipper).
template <bool ZF, bool TOL = false>
__global__ void __launch_bounds__(64) k_pmosc_ranges(PMOscArgs a, uint32_t *__restrict__ next, Img out, uint32_t start, uint32_t end, uint32_t ch) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (v >= a.V) return;
    const uint32_t f0 = start + blockIdx.y * ch, f1 = min(f0 + ch, end);
    PMLane n;
    pm_load(n, a, v);
    n.begin(a.sample_rate, a.freq.get(v), a.release_duration[v], a.note_on.get(v), a.nic.get(v));
    // the replay: 8 frames at a time where no voice of the wave can end an envelope stage (EnvLaneT::quiet) -- then only the
    // three clocks step (the envelope's value is evaluated once per chunk, for the state it leaves) -- frame by frame otherwise
    // (32 at a time while that holds: the test costs as much as four frames' clocks; last_value is not kept up during the replay --
    // every frame painted afterwards rewrites it from the clock before anything reads it, and every range paints at least one)
    uint32_t i = start;
    for (; i + 32 <= f0 && n.env.quiet(32); i += 32) {
        n.env.template skip_clock<32>();
#pragma unroll
        for (int k = 0; k < 32; k++) { float tm_i, tc_i; n.step_phase(tm_i, tc_i); }
    }
    for (; i + 8 <= f0; i += 8) {
        if (n.env.quiet(8)) {
            n.env.template skip_clock<8>();
#pragma unroll
            for (int k = 0; k < 8; k++) { float tm_i, tc_i; n.step_phase(tm_i, tc_i); }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) { float tm_i, tc_i, e0; n.step(tm_i, tc_i, e0); }
        }
    }
    for (; i < f0; i++) {
        float tm_i, tc_i, e0;
        n.step(tm_i, tc_i, e0);
    }
    pm_paint_frames<ZF, TOL>(n, out, v, f0, f1);
    if (f1 != end) return;
    n.end();
    const size_t V = a.V;
    next[v] = __builtin_bit_cast(uint32_t, n.tc); next[V + v] = __builtin_bit_cast(uint32_t, n.tm);
    next[2 * V + v] = n.env.state; next[3 * V + v] = __builtin_bit_cast(uint32_t, n.env.t);
    next[4 * V + v] = __builtin_bit_cast(uint32_t, n.env.last_value); next[5 * V + v] = __builtin_bit_cast(uint32_t, n.env.start);
}
// ------------------------------------------------------------------ span-table paints
// One launch = for every voice, the reference's Trigger loop over its sub-spans
// (examples/example_song.zig:336-347): begin() at a sub-span's first frame, end() after its last,
// nothing painted between sub-spans.
struct SpanTableP {
    uint32_t K;
    const uint32_t *count, *start, *end;
    const float *freq;
    const uint8_t *note_on, *nic;
};

// The wave walks the buffer in segments that end at the next sub-span boundary of ANY of its lanes
// (a wave-wide minimum): inside a segment no lane starts or ends a sub-span, so the frame loop is
// the plain one of k_nice / k_pmosc with an `active` select -- checking every lane's boundaries on
// every frame made this kernel 3x slower per frame than k_nice.  Boundaries mostly coincide (every
// voice has one at each 1024-frame buffer edge), so segments are long.  `live` = the lane owns a voice;
// all 64 lanes take part in the minimum.
template <bool ZF, class Lane, class Begin, class End>
__device__ __forceinline__ void span_walk(Lane &n, const SpanTableP &tb, uint32_t V, uint32_t v, bool live, Img out,
                                          uint32_t buf_start, uint32_t buf_end, Begin &&begin, End &&end_fn) {
    const uint32_t cnt = live ? min(tb.count[v], tb.K) : 0;
    uint32_t k = 0, cur_end = 0;
    uint32_t next_start = cnt > 0 ? tb.start[v] : 0xffffffffu;
    bool active = false;
    const float *const *no_in = nullptr;
    auto advance = [&](uint32_t i) ZH_INLINE_LAMBDA {
        for (;;) {
            if (active) {
                if (i == cur_end) {
                    end_fn(); active = false; k++;
                    next_start = k < cnt ? tb.start[(size_t)k * V + v] : 0xffffffffu;
                    continue;
                }
                break;
            }
            if (i == next_start) {
                const size_t idx = (size_t)k * V + v;
                cur_end = tb.end[idx];
                begin(tb.freq[idx], tb.note_on[idx] != 0, tb.nic[idx] != 0);
                active = true;
                continue;
            }
            break;
        }
    };
    uint32_t i = buf_start;
    while (i < buf_end) {
        advance(i);                                             // sub-spans that end / begin at frame i
        uint32_t ev = active ? cur_end : next_start;            // this lane's next boundary (> i)
        ev = (ev > i && ev < buf_end) ? ev : buf_end;           // unsorted / out-of-range entries never fire, as before
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ev = min(ev, (uint32_t)__shfl_xor((int)ev, off));
        const uint32_t seg_end = __builtin_amdgcn_readfirstlane(ev);
        if (live) {
            frame_loop<8, ZF, 0>(out.p, v, out.stride, no_in, nullptr, i, seg_end, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
                if (!active) return false;
                val = n.frame();
                return true;
            });
        }
        i = seg_end;
    }
    advance(buf_end);          // a sub-span that ends with the buffer; empty sub-spans at buf_end
}

template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_nice_spans(NiceArgs a, SpanTableP tb, Img out, uint32_t start, uint32_t end) {
    const uint32_t v0 = blockIdx.x * kSeqBlock + threadIdx.x;
    const bool live = v0 < a.V;
    const uint32_t v = live ? v0 : 0;                           // idle lanes shadow voice 0 read-only and store nothing
    NiceLane n;
    n.cnt = a.cnt[v]; n.l = a.fl[v]; n.b = a.fb[v];
    n.env.state = a.estate[v]; n.env.t = a.et[v]; n.env.last_value = a.elast[v]; n.env.start = a.estart[v];
    n.bad = true; n.k = PulseK{0, 0, 0.0f, 0.0f, 0.0f, 0.0f}; n.g = n.ng = 0.0f; n.cut = n.res = 0.0f;
    const float color = a.color[v];
    span_walk<ZF>(n, tb, a.V, v, live, out, start, end,
                  [&](float freq, bool on, bool nic) ZH_INLINE_LAMBDA { n.begin(a.sample_rate, a.srf, a.sr8, freq, color, on, nic); },
                  [&]() ZH_INLINE_LAMBDA {});
    if (!live) return;
    a.cnt[v] = n.cnt; a.fl[v] = n.l; a.fb[v] = n.b;
    a.estate[v] = n.env.state; a.et[v] = n.env.t; a.elast[v] = n.env.last_value; a.estart[v] = n.env.start;
}

template <bool ZF>
__global__ void __launch_bounds__(kSeqBlock) k_pmosc_spans(PMOscArgs a, SpanTableP tb, Img out, uint32_t start, uint32_t end) {
    const uint32_t v0 = blockIdx.x * kSeqBlock + threadIdx.x;
    const bool live = v0 < a.V;
    const uint32_t v = live ? v0 : 0;
    PMLane n;
    pm_load(n, a, v);
    n.mod_freq = n.inv_sr = n.t_step = 0.0f;
    const float rel = a.release_duration[v];
    span_walk<ZF>(n, tb, a.V, v, live, out, start, end,
                  [&](float freq, bool on, bool nic) ZH_INLINE_LAMBDA { n.begin(a.sample_rate, freq, rel, on, nic); },
                  [&]() ZH_INLINE_LAMBDA { n.end(); });
    if (!live) return;
    pm_store(n, a, v);
}

// NiceInstrument for a handful of voices (config 4: 10 + 4): one WAVE owns one voice and its lanes are 64
// consecutive frames.  What limits a lone voice is the number of instructions on the frame-to-frame
// chain (a lone wave issues one every ~5 cycles), so a frame is taken apart by what really carries state:
//   * the oscillator is a pure function of the phase counter: evaluated for the 64 frames at once
//     (NiceLane::osc at cnt + lane*ifreq);
//   * the envelope's only sequential part is its clock: EnvLane::block64 walks it (2 instructions per
//     frame, nothing at all while idle or sustaining) and evaluates the curve in all lanes at once;
//   * the filter carries (l, b) through every sample: its core (15 operations) runs through the 64 frames
//     in every lane alike, reading frame j's input with a readlane and leaving (l, b, h) in LDS slot j;
//     the input offset before it and the output mix after it are done for all 64 frames at once.
// About 20 instructions per frame instead of 70.  Same per-voice operations in the same order => same
// bits.  Sub-span semantics as in span_walk.
template <bool ZF>
__global__ void __launch_bounds__(64) k_nice_spans_wave(NiceArgs a, SpanTableP tb, Img out, uint32_t start, uint32_t end) {
    __shared__ float walk_s[64], svf_l[64], svf_b[64];
    const uint32_t v = blockIdx.x, lane = threadIdx.x;
    NiceLane n;
    n.cnt = a.cnt[v]; n.l = a.fl[v]; n.b = a.fb[v];
    n.env.state = a.estate[v]; n.env.t = a.et[v]; n.env.last_value = a.elast[v]; n.env.start = a.estart[v];
    n.bad = true; n.k = PulseK{0, 0, 0.0f, 0.0f, 0.0f, 0.0f}; n.g = n.ng = 0.0f; n.cut = n.res = 0.0f;
    const float color = a.color[v];
    const uint32_t cnt = min(tb.count[v], tb.K);
    float *col = out.p + v;
    const size_t os = out.stride;
    auto zero = [&](uint32_t f0, uint32_t f1) ZH_INLINE_LAMBDA {
        if (ZF) for (uint32_t f = f0 + lane; f < f1; f += 64) col[(size_t)f * os] = 0.0f;
    };
    uint32_t i = start;
    for (uint32_t k = 0; k < cnt; k++) {
        const size_t idx = (size_t)k * a.V + v;
        const uint32_t s0 = tb.start[idx], s1 = tb.end[idx];
        if (s0 < i || s0 > end) break;                          // never reached in order: nothing further fires
        zero(i, s0);
        n.begin(a.sample_rate, a.srf, a.sr8, tb.freq[idx], color, tb.note_on[idx] != 0, tb.nic[idx] != 0);
        const bool ends = s1 >= s0 && s1 <= end;                // otherwise the sub-span runs to the buffer end
        const uint32_t seg_end = ends ? s1 : end;
        for (uint32_t f0 = s0; f0 < seg_end; f0 += 64) {
            const uint32_t nf = min(64u, seg_end - f0);
            const float t0_mine = n.osc(n.cnt + lane * n.k.ifreq);
            if (!n.bad) n.cnt += nf * n.k.ifreq;
            const float e0 = n.env.block64(nf, lane, walk_s);   // temps[0] = 0 (+ envelope)
            // NiceLane::tail_filter with only svf_core on the chain: the input offset is added for all 64
            // frames at once before it, the output mix after it from the captured (l, b, h) of each frame
            const float in_mine = t0_mine + kSvfDcOffset;      // Filter.zig:135
            auto step = [&](uint32_t j) ZH_INLINE_LAMBDA {
                const SvfOut s = svf_core(n.l, n.b, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, in_mine), (int)j)), n.cut, n.res);
                svf_l[j] = s.l; svf_b[j] = s.b;
            };
            uint32_t j = 0;
            for (; j + 8 <= nf; j += 8) {
#pragma unroll
                for (int q = 0; q < 8; q++) step(j + q);
            }
            for (; j < nf; j++) step(j);
            if (lane < nf) {
                const float t1 = svf_lowpass_into_zero(svf_l[lane], svf_b[lane]);   // temps[1] = 0 + low-pass
                float *o = col + (size_t)(f0 + lane) * os;
                *o = (ZF ? 0.0f : *o) + e0 * t1;               // multiply :246: out += temps[0]*temps[1]
            }
        }
        i = seg_end;
        if (!ends) break;
    }
    zero(i, end);
    if (lane == 0) {
        a.cnt[v] = n.cnt; a.fl[v] = n.l; a.fb[v] = n.b;
        a.estate[v] = n.env.state; a.et[v] = n.env.t; a.elast[v] = n.env.last_value; a.estart[v] = n.env.start;
    }
}

// A handful of PMOscInstrument voices (config 4: three) cannot use lane-per-voice parallelism, and two
// musl sines per sample in f64 make the serial walk slow.  Here one WAVE owns one voice and its 64 lanes
// are 64 consecutive frames.  The state-carrying part of a frame is three running sums -- the two phase
// accumulators and the envelope's clock -- walked once per wave (zwalk64 / EnvLane::block64: two
// instructions per frame each) with lane j receiving frame j's values; then every lane evaluates
// PMLane::value for its frame: the sines and the envelope curve, once per 64 frames instead of once per
// frame.  Same per-voice operations in the same order => same bits.
// Sub-span semantics are span_walk's: begin() at a sub-span's first frame, end() after its last,
// nothing painted in between, a malformed table entry never fires.
template <bool ZF>
__global__ void __launch_bounds__(64) k_pmosc_spans_wave(PMOscArgs a, SpanTableP tb, Img out, uint32_t start, uint32_t end) {
    __shared__ float walk_s[64];
    const uint32_t v = blockIdx.x, lane = threadIdx.x;
    PMLane n;
    pm_load(n, a, v);
    n.mod_freq = n.inv_sr = n.t_step = 0.0f;
    const float rel = a.release_duration[v];
    const uint32_t cnt = min(tb.count[v], tb.K);
    float *col = out.p + v;
    const size_t os = out.stride;
    auto zero = [&](uint32_t f0, uint32_t f1) ZH_INLINE_LAMBDA {
        if (ZF) for (uint32_t f = f0 + lane; f < f1; f += 64) col[(size_t)f * os] = 0.0f;
    };
    uint32_t i = start;
    for (uint32_t k = 0; k < cnt; k++) {
        const size_t idx = (size_t)k * a.V + v;
        const uint32_t s0 = tb.start[idx], s1 = tb.end[idx];
        if (s0 < i || s0 > end) break;                          // never reached in order: nothing further fires
        zero(i, s0);
        n.begin(a.sample_rate, tb.freq[idx], rel, tb.note_on[idx] != 0, tb.nic[idx] != 0);
        const bool ends = s1 >= s0 && s1 <= end;                // otherwise the sub-span runs to the buffer end, unfinished
        const uint32_t seg_end = ends ? s1 : end;
        for (uint32_t f0 = s0; f0 < seg_end; f0 += 64) {
            const uint32_t nf = min(64u, seg_end - f0);
            const float my_tm = zwalk64<true>(n.tm, n.mod_freq * n.inv_sr, 0, nf, lane, walk_s);   // PMLane::step_phase, 64 frames
            const float my_tc = zwalk64<true>(n.tc, n.t_step, 0, nf, lane, walk_s);
            const float my_e = n.env.block64(nf, lane, walk_s);                                      // PMLane::step_env
            if (lane < nf) {
                float *o = col + (size_t)(f0 + lane) * os;
                *o = (ZF ? 0.0f : *o) + PMLane::value(my_tm, my_tc, my_e);
            }
        }
        i = seg_end;
        if (!ends) break;
        n.end();
    }
    zero(i, end);
    if (lane == 0) pm_store(n, a, v);
}

// ------------------------------------------------------------------ Noise -> Filter voice
struct zh_noise_filter {
    zh_ctx *ctx; uint32_t n; uint64_t *s[4]; float *nb; /* [7][n] */ float *l, *b;
    uint32_t *err;               // k_noise_filter_ring: a ring wait ran into its bound (reported by get_state)
    // ZH_PAINT_TOLERANT (filter_tp.hip.h): scratch of the two-pass form, allocated by the first tolerant paint outside a capture
    uint64_t *tp_cs; float2 *tp_e; uint32_t *tp_flag; uint32_t tp_serial;
    // ... recorded PIPELINED in a ZH_CAPTURE_COALESCE capture (zh_noise_filter_paint): a second scratch set, the generator states pass A
    // predicts for the next paint, and where the chain of the recording capture stands
    uint64_t *tp_cs2; float2 *tp_e2; uint32_t *tp_flag2;
    uint64_t *tp_pred[2][4];
    uint32_t pipe_capture, pipe_n;   // capture_serial of the chain, paints in it so far (0: the next pipelined paint starts a chain)
};

__global__ void k_nf_seed(uint64_t *s0, uint64_t *s1, uint64_t *s2, uint64_t *s3, uint32_t n, uint64_t first_seed) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    ZXoshiro r;
    zxoshiro_seed(r, first_seed + v);                                  // Noise.zig:26-29
    s0[v] = r.s0; s1[v] = r.s1; s2[v] = r.s2; s3[v] = r.s3;
}

template <bool ZF, bool PINK>
__global__ void __launch_bounds__(kSeqBlock) k_noise_filter(uint64_t *__restrict__ s0, uint64_t *__restrict__ s1,
                                                            uint64_t *__restrict__ s2, uint64_t *__restrict__ s3,
                                                            const float *__restrict__ bst, float *__restrict__ l_io,
                                                            float *__restrict__ b_io, uint32_t V, Img out, uint32_t start,
                                                            uint32_t end, float l_mul, float b_mul, float h_mul, F32P cutoff,
                                                            F32P res_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    ZXoshiro r{s0[v], s1[v], s2[v], s3[v]};
    float pb[7] = {0, 0, 0, 0, 0, 0, 0};
    if (PINK) {
#pragma unroll
        for (int j = 0; j < 7; j++) pb[j] = bst[(size_t)j * V + v];
    }
    const float cut = zclampf(cutoff.get(v), 0.0f, 1.0f);              // Filter.zig:114
    const float res = 1.0f - zclampf(res_p.get(v), 0.0f, 1.0f);        // :118
    float l = l_io[v], b = b_io[v];
    const float *const *no_in = nullptr;
    frame_loop<8, ZF, 0>(out.p, v, out.stride, no_in, nullptr, start, end, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
        const float white = zrandom_float32(r) * 2.0f - 1.0f;          // Noise.zig:51 / :58
        float nz = white;
        if (PINK) nz = pink_step(pb, white);                           // :59-66
        const float temp = 0.0f + nz;                                  // zero(temp); temp += noise
        const SvfOut s = svf_step(l, b, temp, cut, res);               // Filter.zig:135-144
        val = s.l * l_mul + s.b * b_mul + s.h * h_mul;                 // :146
        return true;
    });
    s0[v] = r.s0; s1[v] = r.s1; s2[v] = r.s2; s3[v] = r.s3;
    l_io[v] = l; b_io[v] = b;
}

// Few voices (fewer waves than the chip has SIMDs): a lone wave issues one instruction per ~5 cycles, so what
// a voice costs is the number of instructions on its frame-to-frame chain.  The fused frame is two chains that
// meet in a single value: the noise sample (xoshiro256++, float conversion, pink taps: ~26 instructions) and
// the filter (~22).  Here one workgroup of TWO waves owns 64 voices: wave 0 produces noise one tile of 32
// frames ahead into LDS ([frame][voice], conflict-free), wave 1 filters the previous tile and writes the image;
// one barrier per 32 frames (a single call site, in uniform control flow).  Same per-voice operations in the same
// order => same bits as k_noise_filter.
template <bool ZF, bool PINK>
__global__ void __launch_bounds__(128) k_noise_filter_pc(uint64_t *__restrict__ s0, uint64_t *__restrict__ s1,
                                                         uint64_t *__restrict__ s2, uint64_t *__restrict__ s3,
                                                         const float *__restrict__ bst, float *__restrict__ l_io,
                                                         float *__restrict__ b_io, uint32_t V, Img out, uint32_t start,
                                                         uint32_t end, float l_mul, float b_mul, float h_mul, F32P cutoff,
                                                         F32P res_p) {
    constexpr uint32_t CH = 32;
    __shared__ float tile[2][CH][64];
    const uint32_t lane = threadIdx.x & 63;
    const bool producer = threadIdx.x < 64;
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < V;                                           // no early return: both waves meet at the barriers
    const uint32_t vc = live ? v : V - 1;
    const uint32_t n = end - start, nchunks = (n + CH - 1) / CH;
    // producer state (wave 0) / consumer state (wave 1); the other wave's copy is loaded but never used
    ZXoshiro r{s0[vc], s1[vc], s2[vc], s3[vc]};
    float pb[7] = {0, 0, 0, 0, 0, 0, 0};
    if (PINK) {
#pragma unroll
        for (int j = 0; j < 7; j++) pb[j] = bst[(size_t)j * V + vc];
    }
    const float cut = zclampf(cutoff.get(vc), 0.0f, 1.0f);             // Filter.zig:114
    const float res = 1.0f - zclampf(res_p.get(vc), 0.0f, 1.0f);       // :118
    float l = l_io[vc], b = b_io[vc];
    const uint32_t voff = vc * 4u, orow = (uint32_t)out.stride * 4u;
    // software pipeline, one barrier per step in uniform control flow: in step c the producer fills tile c while the
    // consumer drains tile c-1 (the other LDS buffer); nchunks + 1 steps
    for (uint32_t c = 0; c <= nchunks; c++) {
        if (producer && c < nchunks) {
            const uint32_t nf = min(CH, n - c * CH);
            float (*t)[64] = tile[c & 1];
            auto one = [&](uint32_t k) ZH_INLINE_LAMBDA {
                const float white = zrandom_float32(r) * 2.0f - 1.0f;  // Noise.zig:51 / :58
                t[k][lane] = PINK ? pink_step(pb, white) : white;      // :59-66
            };
            if (nf == CH) {
#pragma unroll 8
                for (uint32_t k = 0; k < CH; k++) one(k);
            } else {
                for (uint32_t k = 0; k < nf; k++) one(k);
            }
        } else if (!producer && c > 0) {
            const uint32_t d = c - 1, nf = min(CH, n - d * CH);
            const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + d * CH);
            const float (*t)[64] = tile[d & 1];
            // lanes past the last voice run voice V-1 again (same state, same noise from the producer's twin lane):
            // their stores repeat V-1's values at V-1's address, so nothing needs masking inside the chain
            auto one = [&](uint32_t k, float nz, float o) ZH_INLINE_LAMBDA {
                const float temp = 0.0f + nz;                          // zero(temp); temp += noise
                const SvfOut sv = svf_step(l, b, temp, cut, res);      // Filter.zig:135-144
                const float val = sv.l * l_mul + sv.b * b_mul + sv.h * h_mul;   // :146
                zrow_store<1>(ro, voff, k * orow, o + val);
            };
            if (nf == CH) {
                float x[CH], oc[CH];                                   // the whole tile column first: one LDS wait per tile, not one per frame
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) { x[k] = t[k][lane]; oc[k] = ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow); }
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) one(k, x[k], oc[k]);
            } else {
                for (uint32_t k = 0; k < nf; k++) one(k, t[k][lane], ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
            }
        }
        __syncthreads();
    }
    if (live && producer) { s0[v] = r.s0; s1[v] = r.s1; s2[v] = r.s2; s3[v] = r.s3; }
    if (live && !producer) { l_io[v] = l; b_io[v] = b; }
}

// White noise, few voices, FIVE waves per 64 voices, decoupled by LDS rings instead of barriers.
// At 64 waves on 1,024 SIMDs a wave issues one VALU instruction per ~5 cycles whatever it is (an LDS instruction costs it
// ~15), so a voice costs the instructions on the busiest wave's per-sample path.  The frame's work is cut where its values meet:
//   producers 0-2     xoshiro256++ -> Random.float -> `in = (0 + white) + fcdcoffset` (~30 instructions per sample, the longest
//                     piece, hence three of them): they take 128-frame stretches in turn and hop over the other two's
//                     stretches with one application of the T^256 jump table (noise_jump.hip.h; ~950 instructions);
//   filter            the state-variable recurrence alone (svf_core: 15 instructions) -> (l, b, h) per sample;
//   writer            the output mix, the `+=` and the image store (everything after the recurrence).
// Rings: noise tiles [8][32 frames][64 voices] producer -> filter, (l, b) tiles [3][2][32][64] filter -> writer;
// monotonic tile counters in LDS, polled (bounded) with s_sleep.  Measured per wave (s_memtime, 4,096 voices): the filter
// wave is the busiest, 3,500 cycles per 32-frame tile of which 2,400 are its 481 VALU instructions.
// A tile that holds one of Random.float's multi-draw samples (2^-41 per sample) leaves the other producers' stretches
// misaligned from there on: the filter wave finishes that tile, keeps where the voice stands (frame, generator state
// after the tile, filter state), the writer drops that lane's later stores, and at the end of the kernel the lane walks
// the rest of its span sequentially.  Same per-voice operations in the same order => the bits of k_noise_filter.
#if defined(ZH_NF_PROF)
__device__ unsigned long long zh_dbg[16];
extern "C" __attribute__((visibility("default"))) int zh_debug_counters(unsigned long long *out16, int reset) {
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(zh_dbg), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(zh_dbg), z, sizeof z); }
    return 0;
}
#define ZH_DBG_ADD(i, v) do { if (lane == 0) atomicAdd(&zh_dbg[i], (unsigned long long)(v)); } while (0)
#define ZH_DBG_NOW() __builtin_readcyclecounter()
#else
#define ZH_DBG_ADD(i, v) do { (void)sizeof(v); } while (0)
#define ZH_DBG_NOW() 0ull
#endif
struct NfArgs {
    uint64_t *s[4];
    float *l, *b;
    uint32_t *err;               // set when a ring wait ran into its bound (never in a correct run)
    const uint4 *table256;       // T^256: one application hops the other two producers' 128-draw stretches
    uint32_t V, start, end;
    Img out;
    float l_mul, b_mul, h_mul;
    F32P cutoff, res;
};
constexpr uint32_t kNfProducers = 3;

template <bool ZF>
__global__ void __launch_bounds__(64 * (kNfProducers + 2)) k_noise_filter_ring(const NfArgs a) {
    constexpr uint32_t CH = 32, NS = 8, ST = 4, NONE = 0xFFFFFFFFu, NP = kNfProducers, NT = 64 * (NP + 2);   // ST = tiles per stretch
    __shared__ uint4 tbl[kNoiseJumpEntries];
    __shared__ float4 tile[NS][CH / 4][64];                            // float4 = four frames of a lane side by side (one 16-byte access)
    __shared__ float4 l_q[3][CH / 4][64], b_q[3][CH / 4][64];          // three slots: the filter runs up to two tiles ahead of the writer
    __shared__ uint64_t after[NP][4][64];                              // per producer: generator state after its first multi-draw tile
    __shared__ uint32_t first_multi[NP][64];                           // per producer: that tile's index
    __shared__ uint32_t dead_from[64];                                 // first tile the writer must not store (NONE: all)
    __shared__ uint32_t ready[NS];                                     // ready[slot] = tile + 1 once the tile is in the slot
    __shared__ uint32_t lbh_ready, writ_done;                          // tiles published to / written out by the writer (which also frees the noise slot)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // 0 .. NP-1: producers; NP: filter; NP+1: writer
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < a.V;
    const uint32_t vc = live ? v : a.V - 1;                            // lanes past the last voice run voice V-1 again
    const uint32_t n = a.end - a.start, nt = (n + CH - 1) / CH;
    noise_jump_load(tbl, a.table256, threadIdx.x, NT);
    if (threadIdx.x < 64) {
        dead_from[threadIdx.x] = NONE;
        for (uint32_t q = 0; q < NP; q++) first_multi[q][threadIdx.x] = NONE;
    }
    if (threadIdx.x < NS) ready[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { lbh_ready = 0u; writ_done = 0u; }
    __syncthreads();
    const uint32_t orow = (uint32_t)a.out.stride * 4u;
    constexpr uint32_t Q = CH / 4;
    // frame k of this lane inside a tile of float4 (partial tiles, the careful form)
    auto at = [&](float4 (*t)[64], uint32_t k) ZH_INLINE_LAMBDA -> float & { return reinterpret_cast<float *>(&t[k >> 2][lane])[k & 3]; };
    const float cut = zclampf(a.cutoff.get(vc), 0.0f, 1.0f);           // Filter.zig:114
    const float res = 1.0f - zclampf(a.res.get(vc), 0.0f, 1.0f);       // :118
    bool ok = true;
    ZXoshiro r{0, 0, 0, 0};
    float l = 0.0f, b = 0.0f;
    uint32_t dead_tile = NONE;                                         // filter wave: the multi-draw tile of this lane
    if (wave < NP) {
        // ---------------------------------------------------------------- producer `wave`: stretches wave, wave + NP, ...
        r = ZXoshiro{a.s[0][vc], a.s[1][vc], a.s[2][vc], a.s[3][vc]};
        // producer p starts 128 p draws in: producer 1 steps the generator 128 times (the table in LDS is T^256; this wave has
        // the time: it shares its SIMD with nobody), producer 2 applies the table once
        static_assert(kNfProducers == 3 && CH * ST == 128, "the T^256 table hops two 128-draw stretches");
        if (wave == 1) { for (uint32_t q = 0; q < CH * ST; q++) (void)zxoshiro_next(r); }
        else if (wave == 2) noise_jump_apply(r, tbl);
        bool had_multi = false;
        for (uint32_t j = wave; ST * j < nt && ok; j += NP) {
            for (uint32_t q = 0; q < ST && ok; q++) {
                const uint32_t c = ST * j + q;
                if (c >= nt) break;
                if (c >= NS) ok = ring_wait_ge(&writ_done, c + 1 - NS);          // the slot's previous tile has been read (by the filter, then the writer)
                const uint32_t nf = min(CH, n - c * CH), slot = c & (NS - 1);
                float4 (*t)[64] = tile[slot];
                bool multi = false;
                // the tile without Random.float's rare-branch test per sample; a draw with a zero high word anywhere in it
                // (2^-32 per sample) sends the whole wave through the tile again in the careful form
                const ZXoshiro r_tile = r;
                uint32_t hmin = 0xFFFFFFFFu;
                auto fast = [&]() ZH_INLINE_LAMBDA {
                    const float white = zrandom_float32_common(r, hmin) * 2.0f - 1.0f;   // Noise.zig:51
                    return (0.0f + white) + kSvfDcOffset;              // zero(temp); temp += noise; in = temp + fcdcoffset (Filter.zig:135)
                };
                if (nf == CH) {
#pragma unroll 2
                    for (uint32_t q = 0; q < Q; q++) {
                        const float x0 = fast(), x1 = fast(), x2 = fast(), x3 = fast();
                        t[q][lane] = make_float4(x0, x1, x2, x3);
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++) at(t, k) = fast();
                }
                if (__builtin_amdgcn_ballot_w64(hmin == 0u) != 0) {
                    r = r_tile;
                    for (uint32_t k = 0; k < nf; k++) {
                        const float white = zrandom_float32_multi(r, multi) * 2.0f - 1.0f;
                        at(t, k) = (0.0f + white) + kSvfDcOffset;
                    }
                }
                const bool first = multi && !had_multi;
                if (first) {
                    had_multi = true;
                    first_multi[wave][lane] = c;
                    after[wave][0][lane] = r.s0; after[wave][1][lane] = r.s1; after[wave][2][lane] = r.s2; after[wave][3][lane] = r.s3;
                }
                // (the flag tells the filter wave that first_multi is worth reading for this tile: 2^-41 per sample)
                ring_publish(&ready[slot], (c + 1) | (__builtin_amdgcn_ballot_w64(first) != 0 ? kRingFlag : 0u), lane);
            }
            if (ST * (j + NP) < nt) noise_jump_apply(r, tbl);          // over the other two producers' stretches: 256 draws
        }
    } else if (wave == NP) {
        // ---------------------------------------------------------------- filter: the recurrence alone
        // (l after Filter.zig:142 and b after :139 go on to the writer, which redoes :143-144 from them and the noise tile:
        // two values per frame through LDS instead of three -- an LDS instruction costs a lone wave about three VALU issues)
        l = a.l[vc]; b = a.b[vc];
        // Round 3: the ring counters are read A TILE AHEAD of their use.  A wave's LDS instructions return in order, so a
        // counter load issued behind the sixteen 16-byte writes of a tile comes back after all of them: polled on the spot
        // (round 2) the two waits and the write drain before the publish cost this wave ~1,000 of a tile's 3,500 cycles
        // (measured with each compiled out: ready 7.0, writ_done 3.5, first_multi 1.3 us of 55).  Now a step loads, at its
        // top, the counters the NEXT decisions need (is tile c + 2 in its slot? has the writer released the slot tile c + 1
        // goes to?), computes its tile, and only then looks at them -- they are old by then, which is harmless for monotonic
        // counters: a stale value can only send the wave to the polling loop it used to run every tile.  Three register
        // buffers: the tile in hand, the next one (loaded during the previous step) and the one after.
        float4 f0[Q], f1[Q], f2[Q];
        uint32_t w0 = 0, w1 = 0, w2 = 0;                               // the ready word each buffer's tile was published with
        auto load_tile = [&](uint32_t c, float4 (&x)[Q]) ZH_INLINE_LAMBDA {
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) x[q] = tile[c & (NS - 1)][q][lane];
        };
        bool writ_ok = true;                                           // the (l, b) slot of the step about to run is free
        auto step = [&](uint32_t c, float4 (&cur)[Q], uint32_t wcur, float4 (&nxt2)[Q], uint32_t &wnxt2) ZH_INLINE_LAMBDA {
            const uint32_t nf = min(CH, n - c * CH), p = (c / ST) % NP;
            // issued now, looked at after the tile's arithmetic
            const uint32_t seen_ready = c + 2 < nt ? __atomic_load_n(&ready[(c + 2) & (NS - 1)], __ATOMIC_RELAXED) : 0u;
            const uint32_t seen_writ = __atomic_load_n(&writ_done, __ATOMIC_RELAXED);
            const bool multi = (wcur & kRingFlag) != 0 && first_multi[p][lane] == c;   // (wave-uniform guard: no LDS read per tile)
            asm volatile("" ::: "memory");
            { const unsigned long long t0 = ZH_DBG_NOW();
            if (c >= 3 && !writ_ok && ok) { ok = ring_wait_ge(&writ_done, c - 2); ZH_DBG_ADD(2, 1); }    // the (l, b) slot's previous tile (c - 3) has been written out
            ZH_DBG_ADD(3, ZH_DBG_NOW() - t0); }
            float4 (*tl)[64] = l_q[c % 3], (*tb)[64] = b_q[c % 3];
            if (dead_tile != NONE) {
                // stopped: (l, b) stay as they were after the multi-draw tile; the lane's later stores are dropped anyway
            } else if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    const SvfMid m0 = svf_core_mid(l, b, cur[q].x, cut, res);   // Filter.zig:138-144
                    const SvfMid m1 = svf_core_mid(l, b, cur[q].y, cut, res);
                    const SvfMid m2 = svf_core_mid(l, b, cur[q].z, cut, res);
                    const SvfMid m3 = svf_core_mid(l, b, cur[q].w, cut, res);
                    tl[q][lane] = make_float4(m0.l, m1.l, m2.l, m3.l);
                    tb[q][lane] = make_float4(m0.b1, m1.b1, m2.b1, m3.b1);
                }
            } else {
                float4 (*ti)[64] = tile[c & (NS - 1)];                 // (not freed before the writer has read it)
                for (uint32_t k = 0; k < nf; k++) {
                    const SvfMid m = svf_core_mid(l, b, at(ti, k), cut, res);
                    at(tl, k) = m.l; at(tb, k) = m.b1;
                }
            }
            ring_publish_writes(&lbh_ready, c + 1, lane);              // (behind this wave's tile writes in its LDS queue: no drain)
            if (multi && dead_tile == NONE) {                          // this tile drew more than once per sample somewhere
                dead_tile = c;
                dead_from[lane] = c + 1;                               // (reaches the writer before tile c + 1 does)
            }
            if (c + 2 < nt && ok) {                                    // tile c + 2 into the buffer that held tile c - 1
                { const unsigned long long t0 = ZH_DBG_NOW();
                wnxt2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen_ready);
                if (!ring_reached(wnxt2, c + 3)) { ok = ring_wait_ge(&ready[(c + 2) & (NS - 1)], c + 3, &wnxt2); ZH_DBG_ADD(0, 1); }
                ZH_DBG_ADD(1, ZH_DBG_NOW() - t0); }
                asm volatile("" ::: "memory");
                load_tile(c + 2, nxt2);
            }
            writ_ok = (int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)seen_writ) - (c - 1)) >= 0;   // step c + 1 needs writ_done >= c - 1
        };
        // the first two tiles: waited for on the spot
        const unsigned long long tw0 = ZH_DBG_NOW();
        ok = ring_wait_ge(&ready[0], 1, &w0);
        ZH_DBG_ADD(4, ZH_DBG_NOW() - tw0);
        load_tile(0, f0);
        if (nt > 1 && ok) { ok = ring_wait_ge(&ready[1], 2, &w1); asm volatile("" ::: "memory"); load_tile(1, f1); }
        const unsigned long long tl0 = ZH_DBG_NOW();
        for (uint32_t c = 0; c < nt && ok; c += 3) {
            step(c, f0, w0, f2, w2);
            if (c + 1 < nt && ok) step(c + 1, f1, w1, f0, w0);
            if (c + 2 < nt && ok) step(c + 2, f2, w2, f1, w1);
        }
        ZH_DBG_ADD(5, ZH_DBG_NOW() - tl0);
        ZH_DBG_ADD(6, 1);
    } else {
        // ---------------------------------------------------------------- writer: Filter.zig:143-144 again, mix, +=, store
        for (uint32_t c = 0; c < nt && ok; c++) {
            const uint32_t nf = min(CH, n - c * CH);
            ok = ring_wait_ge(&lbh_ready, c + 1);
            // descriptor over exactly this tile's rows: a lane that has stopped (offset 2^31) is out of its range, its
            // loads return 0 and its stores are dropped
            const uint32_t voff = c >= dead_from[lane] ? 0x80000000u : vc * 4u;
            const zh_rsrc_t ro = make_rsrc(a.out.p + (size_t)(a.start + c * CH) * a.out.stride, CH * orow);
            float4 (*ti)[64] = tile[c & (NS - 1)], (*tl)[64] = l_q[c % 3], (*tb)[64] = b_q[c % 3];
            auto one = [&](uint32_t k, float in, float lv, float b1, float o) ZH_INLINE_LAMBDA {
                const SvfOut sv = svf_finish(lv, b1, in, cut, res);
                const float val = sv.l * a.l_mul + sv.b * a.b_mul + sv.h * a.h_mul;   // :146
                zrow_store<1>(ro, voff, k * orow, o + val);
            };
            if (nf == CH) {
                float4 xi[Q], xl[Q], xb[Q];
                float oc[CH];
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) { xi[q] = ti[q][lane]; xl[q] = tl[q][lane]; xb[q] = tb[q][lane]; }
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) oc[k] = ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow);
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    one(4 * q, xi[q].x, xl[q].x, xb[q].x, oc[4 * q]);
                    one(4 * q + 1, xi[q].y, xl[q].y, xb[q].y, oc[4 * q + 1]);
                    one(4 * q + 2, xi[q].z, xl[q].z, xb[q].z, oc[4 * q + 2]);
                    one(4 * q + 3, xi[q].w, xl[q].w, xb[q].w, oc[4 * q + 3]);
                }
            } else {
                for (uint32_t k = 0; k < nf; k++) one(k, at(ti, k), at(tl, k), at(tb, k), ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow));
            }
            ring_publish(&writ_done, c + 1, lane);
        }
    }
    if (!ok && lane == 0) *a.err = 1u;
    __syncthreads();
    // the span's last tile belongs to stretch (nt - 1) / ST: its producer holds the generator state after the span
    if (live && wave == ((nt - 1) / ST) % NP && dead_from[lane] == NONE) { a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3; }
    if (wave == NP) {
        if (dead_tile != NONE) {
            // the rest of a stopped voice's span, walked like k_noise_filter from where the multi-draw tile left it
            // (the writer's stores of the earlier tiles are complete: they precede the barrier above in its program order,
            // and the frames written here are disjoint from them)
            const uint32_t p = (dead_tile / ST) % NP;
            ZXoshiro rr{after[p][0][lane], after[p][1][lane], after[p][2][lane], after[p][3][lane]};
            const uint32_t f0 = min(a.start + (dead_tile + 1) * CH, a.end);
            if (live) {
                const float *const *no_in = nullptr;
                frame_loop<8, ZF, 0>(a.out.p, v, a.out.stride, no_in, nullptr, f0, a.end, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
                    const float white = zrandom_float32(rr) * 2.0f - 1.0f;
                    const float temp = 0.0f + white;
                    const SvfOut s = svf_step(l, b, temp, cut, res);
                    val = s.l * a.l_mul + s.b * a.b_mul + s.h * a.h_mul;
                    return true;
                });
                a.s[0][v] = rr.s0; a.s[1][v] = rr.s1; a.s[2][v] = rr.s2; a.s[3][v] = rr.s3;
            }
        }
        if (live) { a.l[v] = l; a.b[v] = b; }
    }
}

// bypass: out += noise, filter state untouched (Filter.zig:91-97)
template <bool ZF, bool PINK>
__global__ void __launch_bounds__(kSeqBlock) k_noise_filter_bypass(uint64_t *__restrict__ s0, uint64_t *__restrict__ s1,
                                                                   uint64_t *__restrict__ s2, uint64_t *__restrict__ s3,
                                                                   const float *__restrict__ bst, uint32_t V, Img out,
                                                                   uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= V) return;
    ZXoshiro r{s0[v], s1[v], s2[v], s3[v]};
    float pb[7] = {0, 0, 0, 0, 0, 0, 0};
    if (PINK) {
#pragma unroll
        for (int j = 0; j < 7; j++) pb[j] = bst[(size_t)j * V + v];
    }
    const float *const *no_in = nullptr;
    frame_loop<8, ZF, 0>(out.p, v, out.stride, no_in, nullptr, start, end, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
        const float white = zrandom_float32(r) * 2.0f - 1.0f;
        float nz = white;
        if (PINK) nz = pink_step(pb, white);
        val = 0.0f + nz;
        return true;
    });
    s0[v] = r.s0; s1[v] = r.s1; s2[v] = r.s2; s3[v] = r.s3;
}

static void nf_free(zh_noise_filter *m) {
    for (auto &x : m->s) (void)hipFree(x);
    (void)hipFree(m->nb); (void)hipFree(m->l); (void)hipFree(m->b);
    (void)hipFree(m->err);
    (void)hipFree(m->tp_cs); (void)hipFree(m->tp_e); (void)hipFree(m->tp_flag);
    (void)hipFree(m->tp_cs2); (void)hipFree(m->tp_e2); (void)hipFree(m->tp_flag2);
    for (auto &set : m->tp_pred) for (auto &x : set) (void)hipFree(x);
}

__global__ void k_fill_f32(float *p, uint32_t n, F32P src) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = src.get(i);
}
__global__ void k_fill_u32(uint32_t *p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

template <typename T> static int up(zh_ctx *ctx, T *dev, const std::vector<T> &h) { return zh_upload(ctx, dev, h.data(), h.size() * sizeof(T)); }
template <typename T> static int down(zh_ctx *ctx, std::vector<T> &h, const T *dev, size_t n) { h.resize(n); return zh_download(ctx, h.data(), dev, n * sizeof(T)); }

static NiceArgs nice_args(zh_nice *m, const zh_nice_params *p, zh_bool nic) {
    NiceArgs a;
    a.color = m->color; a.cnt = m->cnt; a.fl = m->fl; a.fb = m->fb;
    a.estate = m->estate; a.et = m->et; a.elast = m->elast; a.estart = m->estart;
    a.V = m->n;
    a.sample_rate = p->sample_rate;
    a.srf = 4294967296.0f / p->sample_rate;                            // PulseOsc.zig:87
    a.sr8 = p->sample_rate / 8.0f;                                     // :82
    a.freq = mk_f32(p->freq);
    a.note_on = mk_bool(p->note_on);
    a.nic = mk_bool(nic);
    return a;
}

static void nice_free(zh_nice *m) {
    (void)hipFree(m->color); (void)hipFree(m->cnt); (void)hipFree(m->fl); (void)hipFree(m->fb);
    (void)hipFree(m->estate); (void)hipFree(m->et); (void)hipFree(m->elast); (void)hipFree(m->estart);
    (void)hipFree(m->tp);
}
static void pmosc_free(zh_pmosc *m) { (void)hipFree(m->release_duration); (void)hipFree(m->cnt[0]); (void)hipFree(m->cnt[1]); }

extern "C" {

// ------------------------------------------------------------------ NiceInstrument
int zh_nice_create(zh_ctx *ctx, uint32_t n, zh_f32 color, zh_nice **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_nice *m = new (std::nothrow) zh_nice();
    if (!m) return ZH_ERR_INVALID;
    *m = zh_nice{ctx, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = dev_alloc(&m->color, n);
    if (!rc) rc = dev_alloc(&m->cnt, n);
    if (!rc) rc = dev_alloc(&m->fl, n);
    if (!rc) rc = dev_alloc(&m->fb, n);
    if (!rc) rc = dev_alloc(&m->estate, n);
    if (!rc) rc = dev_alloc(&m->et, n);
    if (!rc) rc = dev_alloc(&m->elast, n);
    if (!rc) rc = dev_alloc(&m->estart, n);
    if (rc) { nice_free(m); delete m; return rc; }
    if (n) {
        hipStream_t st = ctx->stream;
        ZH_LAUNCH(k_fill_f32, dim3((n + 255) / 256), dim3(256), 0, st, m->color, n, mk_f32(color));
        void *zeros[] = {m->cnt, m->fl, m->fb, m->estate, m->et, m->elast, m->estart};   // sub-module init()s
        for (void *z : zeros) { hipError_t e = hipMemsetAsync(z, 0, (size_t)n * 4, st); if (e != hipSuccess) { nice_free(m); delete m; return (int)e; } }
    }
    *out = m;
    return zh_launch_status();
}
int zh_nice_destroy(zh_nice *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    nice_free(m);
    delete m;
    return ZH_OK;
}
int zh_nice_get_state(zh_nice *m, zh_nice_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    std::vector<uint32_t> cnt, es;
    std::vector<float> l, b, t, lv, sv;
    int rc = down(m->ctx, cnt, m->cnt, m->n);
    if (!rc) rc = down(m->ctx, l, m->fl, m->n);
    if (!rc) rc = down(m->ctx, b, m->fb, m->n);
    if (!rc) rc = down(m->ctx, es, m->estate, m->n);
    if (!rc) rc = down(m->ctx, t, m->et, m->n);
    if (!rc) rc = down(m->ctx, lv, m->elast, m->n);
    if (!rc) rc = down(m->ctx, sv, m->estart, m->n);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->n; v++) {
        host[v].osc.cnt = cnt[v];
        host[v].flt = zh_filter_state{l[v], b[v]};
        host[v].env = zh_envelope_state{es[v], t[v], lv[v], sv[v]};
    }
    return ZH_OK;
}
int zh_nice_set_state(zh_nice *m, const zh_nice_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    const uint32_t n = m->n;
    std::vector<uint32_t> cnt(n), es(n);
    std::vector<float> l(n), b(n), t(n), lv(n), sv(n);
    for (uint32_t v = 0; v < n; v++) {
        cnt[v] = host[v].osc.cnt; l[v] = host[v].flt.l; b[v] = host[v].flt.b;
        es[v] = host[v].env.state; t[v] = host[v].env.t; lv[v] = host[v].env.last_value; sv[v] = host[v].env.start;
    }
    int rc = up(m->ctx, m->cnt, cnt);
    if (!rc) rc = up(m->ctx, m->fl, l);
    if (!rc) rc = up(m->ctx, m->fb, b);
    if (!rc) rc = up(m->ctx, m->estate, es);
    if (!rc) rc = up(m->ctx, m->et, t);
    if (!rc) rc = up(m->ctx, m->elast, lv);
    if (!rc) rc = up(m->ctx, m->estart, sv);
    return rc;
}
int zh_nice_paint(zh_nice *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                  zh_bool note_id_changed, const zh_nice_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;   // the fused kernel keeps both temps in registers
    if (!m || !outputs || !p || end < start || !buf_covers(outputs[0], m->n, end)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    hipStream_t st = m->ctx->stream;
    NiceArgs a = nice_args(m, p, note_id_changed);
    const Img out = mk_img(outputs[0]);
    const bool zf = (flags & ZH_PAINT_ZERO_FIRST) != 0;
    // ZH_PAINT_TOLERANT, few voices: the span as chunks at once, two passes (k_nice_tp_a / _b); spans of one launch pair
    if ((flags & ZH_PAINT_TOLERANT) && end - start >= 128 && end - start <= kNiceTpMaxChunks * 128u) {
        const uint32_t C = zh_tp_chunks(m->n, ZF_NICE_TP_MAX, end - start);
        if (C >= 2 && !m->tp && !m->ctx->capturing && dev_alloc(&m->tp, (size_t)kNiceTpWords * m->n) != ZH_OK) { m->tp = nullptr; (void)hipGetLastError(); }
        if (C >= 2 && m->tp) {
            NiceTpArgs t;
            t.a = a; t.tp = m->tp; t.start = start; t.end = end; t.L = ((end - start + C - 1) / C + 7) / 8 * 8; t.out = out;   // whole 8-frame groups per chunk
            const dim3 grid((m->n + 255) / 256, (end - start + t.L - 1) / t.L);
            ZH_LAUNCH(k_nice_tp_a, grid, dim3(256), 0, st, t);
            if (zf) ZH_LAUNCH(k_nice_tp_b<true>, grid, dim3(256), 0, st, t);
            else ZH_LAUNCH(k_nice_tp_b<false>, grid, dim3(256), 0, st, t);
            return zh_launch_status();
        }
    }
    if (m->n <= nice_pc_max() && end > start && outputs[0].stride <= (1u << 24)) {   // (32-row tiles: 32-bit offsets)
        // up to ZH_NICE_PC_MAX voices the three chains of a frame run in three waves side by side (k_nice_pc)
        // four waves (k_nice_pc4) up to ZH_NICE_PC4_MAX voices: 4,096 / 16,384 / 32,768 voices 44 / 47 / 53.5 us against 60.5 / 62.5 /
        // 63 for the three-wave form; its 64 KB of LDS per workgroup allow two workgroups per CU = 32,768 voices, beyond that
        // the three-wave form stays (65,536 voices: 82 us against 111)
        const long pc4_max = zh_form(ZF_NICE_PC4_MAX);
        if ((long)m->n <= pc4_max) {
            if (zf) ZH_LAUNCH(k_nice_pc4<true>, seq_grid(m->n), dim3(256), 0, st, a, out, start, end);
            else ZH_LAUNCH(k_nice_pc4<false>, seq_grid(m->n), dim3(256), 0, st, a, out, start, end);
        } else {
            if (zf) ZH_LAUNCH(k_nice_pc<true>, seq_grid(m->n), dim3(192), 0, st, a, out, start, end);
            else ZH_LAUNCH(k_nice_pc<false>, seq_grid(m->n), dim3(192), 0, st, a, out, start, end);
        }
    } else {
        if (zf) ZH_LAUNCH((k_nice<true, 1>), seq_grid(m->n), dim3(kSeqBlock), 0, st, a, out, start, end);
        else ZH_LAUNCH((k_nice<false, 1>), seq_grid(m->n), dim3(kSeqBlock), 0, st, a, out, start, end);
    }
    return zh_launch_status();
}
static bool nice_mix_roll() { return zh_form(ZF_NICE_MIX_ROLL) != 0; }   // 149.6 vs 148.2 us at 131,072 voices, 908 vs 890 us at 1,048,576
// One partial row per workgroup (a barrier per chunk in the sum phase) or one per wave.  A/B on one box (tools/ab_env.sh,
// profiles/r04/ab_nice_mix_wg.txt): 131,072 voices 108.45 -> 108.35 us per buffer all-in (0.6 us in an earlier comparison) with a quarter of
// the partial traffic; 4,096 voices 99.3 -> 104.6 (sixteen workgroups on 256 CUs: the barrier couples waves that otherwise run at their own pace).
// nice_mix_wg_min (dispatch.hip) = the smallest voice count that combines per workgroup.  (Eight-wave workgroups -- one row per
// 512 voices, 2.1 MB of partial rows instead of 4.2 at 131,072 voices -- were a switch through round 4 and slower: the barrier
// over eight waves costs more than the rows save, 107.5 -> 109.6 us per buffer, profiles/r04/ab_nice_mix_wg8.txt; traffic is not
// this kernel's bound.  Gone with round 5.)
// Returns the waves per combining workgroup: 0 (a row per wave) or 4.
static int nice_mix_wg(uint32_t n_voices) { return (long)n_voices >= zh_form(ZF_NICE_MIX_WG_MIN) ? 4 : 0; }
constexpr uint32_t kNiceCoalesceReserve = 8;
static int nice_paint_mix_n(zh_nice *m, uint32_t start, uint32_t end, float *mix_l, float *mix_r, const zh_f32 *gain_l,
                            const zh_f32 *gain_r, zh_bool note_id_changed, const zh_nice_params *p, uint32_t flags) {
    const bool stereo = mix_r != nullptr;
    if (!m || !mix_l || !p || end < start) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    const uint32_t nframes = end - start;
    const int wg = nice_mix_wg(m->n);
    const uint32_t bs = 256u;
    const uint32_t blocks = (m->n + bs - 1) / bs;
    const uint32_t rows = wg ? blocks : blocks * 4;                     // one partial row per workgroup (256 / 512 voices) / per wave of 64 voices
    const size_t per_channel = (size_t)rows * kMixGroupFrames * ((nframes + kMixGroupFrames - 1) / kMixGroupFrames ? (nframes + kMixGroupFrames - 1) / kMixGroupFrames : 1);   // [frame / G][row][frame % G]
    // (stereo: room for the 8 buffers a coalescing capture merges, reserved while the scratch may still grow)
    int rc = zh_mix_reserve(m->ctx, per_channel * (stereo ? 2 : 1) * (stereo && !m->ctx->capturing ? kNiceCoalesceReserve : 1u));
    if (rc) return rc;
    hipStream_t st = m->ctx->stream;
    NiceArgs a = nice_args(m, p, note_id_changed);
    float *part = m->ctx->mix_partials;
    const int zf = (int)(flags & ZH_PAINT_ZERO_FIRST);
    // ZH_PAINT_TOLERANT, few voices: the span as chunks at once (k_nice_tp_a, k_nice_mix_tp_b), chunks of whole 32-frame tiles
    if ((flags & ZH_PAINT_TOLERANT) && !wg && nframes >= 128 && nframes <= kNiceTpMaxChunks * 128u) {
        const uint32_t Cn = zh_tp_chunks(m->n, ZF_NICE_TP_MAX, nframes);
        if (Cn >= 2 && !m->tp && !m->ctx->capturing && dev_alloc(&m->tp, (size_t)kNiceTpWords * m->n) != ZH_OK) { m->tp = nullptr; (void)hipGetLastError(); }
        if (Cn >= 2 && m->tp) {
            NiceTpArgs t;
            t.a = a; t.tp = m->tp; t.start = start; t.end = end; t.L = ((nframes + Cn - 1) / Cn + MIXF - 1) / MIXF * MIXF; t.out = Img{nullptr, 0};
            const dim3 grid(blocks, (nframes + t.L - 1) / t.L);
            ZH_LAUNCH(k_nice_tp_a, grid, dim3(256), 0, st, t);
            const bool roll = nice_mix_roll();
            if (stereo) {
                const F32P gl = mk_f32(*gain_l), gr = mk_f32(*gain_r);
                if (roll) ZH_LAUNCH((k_nice_mix_tp_b<2, true>), grid, dim3(256), 0, st, t, part, gl, gr);
                else ZH_LAUNCH((k_nice_mix_tp_b<2, false>), grid, dim3(256), 0, st, t, part, gl, gr);
                zh_mix_pass2_wide_launch(m->ctx, part, per_channel, rows, nframes, mix_l + start, mix_r + start, 2, zf);
            } else {
                const F32P none = mk_f32(zh_f32{0.0f, 0, nullptr});
                if (roll) ZH_LAUNCH((k_nice_mix_tp_b<1, true>), grid, dim3(256), 0, st, t, part, none, none);
                else ZH_LAUNCH((k_nice_mix_tp_b<1, false>), grid, dim3(256), 0, st, t, part, none, none);
                zh_mix_pass2_wide_launch(m->ctx, part, per_channel, rows, nframes, mix_l + start, nullptr, 1, zf);
            }
            return zh_launch_status();
        }
    }
    // ZH_PAINT_TOLERANT, more voices than the chunked form takes: the same kernel with multiply-adds fused (nice_mix_fma.hip)
    if ((flags & ZH_PAINT_TOLERANT) && zh_form(ZF_NICE_MIX_FMA) != 0) {
        const F32P none = mk_f32(zh_f32{0.0f, 0, nullptr});
        zh_nice_mix_fma_launch(stereo ? 2 : 1, nice_mix_roll(), wg, blocks, st, a, start, end, part, stereo ? mk_f32(*gain_l) : none, stereo ? mk_f32(*gain_r) : none);
        if (nframes) zh_mix_pass2_wide_launch(m->ctx, part, per_channel, rows, nframes, mix_l + start, stereo ? mix_r + start : nullptr, stereo ? 2 : 1, zf);
        return zh_launch_status();
    }
#define ZH_NMIX(C_, ROLL_, NW_, GL_, GR_) ZH_LAUNCH((k_nice_mix<C_, ROLL_, NW_>), dim3(blocks), dim3(bs), 0, st, a, start, end, part, GL_, GR_)
#define ZH_NMIX_W(C_, ROLL_, GL_, GR_) do { if (wg) ZH_NMIX(C_, ROLL_, 4, GL_, GR_); else ZH_NMIX(C_, ROLL_, 0, GL_, GR_); } while (0)
    const bool roll = nice_mix_roll();
    if (stereo) {
        const F32P gl = mk_f32(*gain_l), gr = mk_f32(*gain_r);
        if (roll) ZH_NMIX_W(2, true, gl, gr); else ZH_NMIX_W(2, false, gl, gr);
        if (nframes) zh_mix_pass2_wide_launch(m->ctx, part, per_channel, rows, nframes, mix_l + start, mix_r + start, 2, zf);
    } else {
        const F32P none = mk_f32(zh_f32{0.0f, 0, nullptr});
        if (roll) ZH_NMIX_W(1, true, none, none); else ZH_NMIX_W(1, false, none, none);
        if (nframes) zh_mix_pass2_wide_launch(m->ctx, part, per_channel, rows, nframes, mix_l + start, nullptr, 1, zf);
    }
#undef ZH_NMIX_W
#undef ZH_NMIX
    return zh_launch_status();
}
int zh_nice_paint_mix(zh_nice *m, uint32_t start, uint32_t end, float *mix, zh_bool note_id_changed,
                      const zh_nice_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    return nice_paint_mix_n(m, start, end, mix, nullptr, nullptr, nullptr, note_id_changed, p, flags);
}
// A stereo mixdown paint of a ZH_CAPTURE_COALESCE capture is held back (common.hip.h zh_co_batch): consecutive ones of one module over the
// same span with the same gains and flags into different mix rows become ONE launch of up to 8 buffers -- the launch
// zh_nice_paint_mix_stereo_batch makes (the state words stay in registers from buffer to buffer, one second pass): same bits,
// 108.4 -> 105.1 us per buffer at 131,072 voices.  The partial-sum scratch is sized for 8 buffers by every eager stereo paint (it
// cannot grow while a capture records); a batch the scratch cannot hold goes out buffer by buffer.
constexpr uint32_t kNiceCoalesceMax = 8;
static int nice_paint_mix_batch_impl(zh_nice *m, uint32_t start, uint32_t end, uint32_t n_buffers, float *const *mix_left,
                                     float *const *mix_right, zh_f32 gain_left, zh_f32 gain_right, const zh_bool *note_id_changed,
                                     const zh_nice_params *params, uint32_t flags);
struct NiceHeld { float *l, *r; zh_bool nic; zh_nice_params p; };
struct NiceHeldBatch { std::vector<NiceHeld> v; zh_f32 gl, gr; uint32_t flags, sample_rate_bits; };   // (what a joining paint must share, each compared on its own)
static bool same_f32(const zh_f32 &a, const zh_f32 &b) {
    return a.per_voice == b.per_voice && (a.per_voice || __builtin_bit_cast(uint32_t, a.value) == __builtin_bit_cast(uint32_t, b.value));
}
int zh_nice_paint_mix_stereo(zh_nice *m, uint32_t start, uint32_t end, float *mix_left, float *mix_right, zh_f32 gain_left,
                             zh_f32 gain_right, zh_bool note_id_changed, const zh_nice_params *p, uint32_t flags) { ZH_GUARD_EPOCH(m ? m->ctx : nullptr);
    if (!mix_right) return ZH_ERR_INVALID;
    zh_ctx *ctx = m ? m->ctx : nullptr;
    const bool hold = ctx && ctx->capturing && (ctx->capture_flags & ZH_CAPTURE_COALESCE) && mix_left && p && end > start && m->n &&
                      (!(flags & ZH_PAINT_TOLERANT) || nice_mix_wg(m->n));      // (a tolerant paint of few voices takes the chunked form, alone)
    // Held back, the paint is launched later, by whatever ends the batch, and this call has long returned: so everything the launch
    // could refuse is looked at NOW -- the partial-sum scratch cannot grow while a capture records (zh_mix_reserve), and a paint whose
    // one buffer it cannot hold is not held: it goes out in order and this call returns what the launch says, as in a capture without
    // the flag (the recorded graph silently lacked such paints).
    bool fits = false;
    if (hold) {
        const uint32_t blocks = (m->n + 255u) / 256u, rows = nice_mix_wg(m->n) ? blocks : blocks * 4;
        const uint32_t groups = (end - start + kMixGroupFrames - 1) / kMixGroupFrames;
        fits = (size_t)rows * kMixGroupFrames * groups * 2 <= ctx->mix_partials_floats;
    }
    if (!hold || !fits) {
        if (ctx && ctx->epoch_open) zh_epoch_barrier(ctx);
        return nice_paint_mix_n(m, start, end, mix_left, mix_right, &gain_left, &gain_right, note_id_changed, p, flags);
    }
    zh_co_batch &cb = ctx->co;
    std::shared_ptr<NiceHeldBatch> held = cb.active && cb.owner == m ? std::static_pointer_cast<NiceHeldBatch>(cb.items) : nullptr;
    const uint32_t sr_bits = __builtin_bit_cast(uint32_t, p->sample_rate);
    bool join = held && cb.start == start && cb.end == end && held->flags == flags && held->sample_rate_bits == sr_bits && held->v.size() < kNiceCoalesceMax &&
                same_f32(held->gl, gain_left) && same_f32(held->gr, gain_right);
    for (size_t q = 0; join && q < held->v.size(); q++) {
        const float *rows[2] = {held->v[q].l, held->v[q].r};
        for (const float *x : rows)                                     // the same mix row again: the recorded order decides what it holds
            if ((mix_left < x + end && x < mix_left + end) || (mix_right < x + end && x < mix_right + end)) join = false;
    }
    if (!join) {
        zh_epoch_flush_batch(ctx, false);
        held = std::make_shared<NiceHeldBatch>();
        held->gl = gain_left; held->gr = gain_right; held->flags = flags; held->sample_rate_bits = sr_bits;
        cb.items = held;
        cb.active = true; cb.owner = m; cb.start = start; cb.end = end; cb.stride = 0; cb.key = 0; cb.flips = false;
        ctx->epoch_open = true;
        std::shared_ptr<NiceHeldBatch> items = held;
        cb.launch = [m, start, end, gain_left, gain_right, flags, items](hipStream_t, float *const *, uint32_t cnt) {
            const std::vector<NiceHeld> &v = items->v;
            float *l[kNiceCoalesceMax], *r[kNiceCoalesceMax];
            zh_bool nic[kNiceCoalesceMax];
            zh_nice_params ps[kNiceCoalesceMax];
            for (uint32_t i = 0; i < cnt; i++) { l[i] = v[i].l; r[i] = v[i].r; nic[i] = v[i].nic; ps[i] = v[i].p; }
            int rc = cnt > 1 ? nice_paint_mix_batch_impl(m, start, end, cnt, l, r, gain_left, gain_right, nic, ps, flags) : ZH_ERR_UNSUPPORTED;
            if (rc != ZH_OK) {                                          // one buffer, or a scratch too small for the batch: buffer by buffer
                for (uint32_t i = 0; i < cnt; i++) {
                    rc = nice_paint_mix_n(m, start, end, l[i], r[i], &gain_left, &gain_right, nic[i], &ps[i], flags);
                    if (rc != ZH_OK && !m->ctx->deferred_error) m->ctx->deferred_error = rc;    // (zh_graph_end_capture returns it)
                }
            }
        };
    }
    held->v.push_back(NiceHeld{mix_left, mix_right, note_id_changed, *p});
    cb.imgs.push_back(mix_left);
    ctx->co_paints++;
    return ZH_OK;
}

static int nice_paint_mix_batch_impl(zh_nice *m, uint32_t start, uint32_t end, uint32_t n_buffers, float *const *mix_left,
                                     float *const *mix_right, zh_f32 gain_left, zh_f32 gain_right, const zh_bool *note_id_changed,
                                     const zh_nice_params *params, uint32_t flags) {
    if (!m || !mix_left || !mix_right || !note_id_changed || !params || end < start || n_buffers > (uint32_t)kNiceMixMaxBatch) return ZH_ERR_INVALID;
    for (uint32_t k = 0; k < n_buffers; k++) {
        if (!mix_left[k] || !mix_right[k]) return ZH_ERR_INVALID;
        // (one sample rate per batch: srf / sr8 are launch constants)
        if (__builtin_bit_cast(uint32_t, params[k].sample_rate) != __builtin_bit_cast(uint32_t, params[0].sample_rate)) return ZH_ERR_UNSUPPORTED;
    }
    if (m->n == 0 || n_buffers == 0) return ZH_OK;
    const uint32_t nframes = end - start;
    const int wg = nice_mix_wg(m->n);
    const uint32_t bs = 256u;
    const uint32_t blocks = (m->n + bs - 1) / bs, rows = wg ? blocks : blocks * 4;
    const size_t per_channel = (size_t)rows * kMixGroupFrames * ((nframes + kMixGroupFrames - 1) / kMixGroupFrames ? (nframes + kMixGroupFrames - 1) / kMixGroupFrames : 1);
    int rc = zh_mix_reserve(m->ctx, per_channel * 2 * n_buffers);
    if (rc) return rc;
    NiceBatchArgs b;
    b.a = nice_args(m, &params[0], note_id_changed[0]);
    b.nb = n_buffers;
    for (int k = 0; k < kNiceMixMaxBatch; k++) {
        const uint32_t j = (uint32_t)k < n_buffers ? (uint32_t)k : 0u;
        b.freq[k] = mk_f32(params[j].freq); b.note_on[k] = mk_bool(params[j].note_on); b.nic[k] = mk_bool(note_id_changed[j]);
    }
    float *part = m->ctx->mix_partials;
    hipStream_t st = m->ctx->stream;
#define ZH_NMIXB(ROLL_, NW_) ZH_LAUNCH((k_nice_mix_batch<2, ROLL_, NW_>), dim3(blocks), dim3(bs), 0, st, b, start, end, part, mk_f32(gain_left), mk_f32(gain_right))
    if ((flags & ZH_PAINT_TOLERANT) && zh_form(ZF_NICE_MIX_FMA) != 0)   // (a batch never takes the chunked form: its buffers are consecutive in time)
        zh_nice_mix_batch_fma_launch(nice_mix_roll(), wg, blocks, st, b, start, end, part, mk_f32(gain_left), mk_f32(gain_right));
    else if (wg) { if (nice_mix_roll()) ZH_NMIXB(true, 4); else ZH_NMIXB(false, 4); }
    else { if (nice_mix_roll()) ZH_NMIXB(true, 0); else ZH_NMIXB(false, 0); }
#undef ZH_NMIXB
    if (nframes) {
        float *l[kNiceMixMaxBatch], *r[kNiceMixMaxBatch];
        for (uint32_t k = 0; k < n_buffers; k++) { l[k] = mix_left[k] + start; r[k] = mix_right[k] + start; }
        zh_mix_pass2_wide_batch_launch(m->ctx, part, per_channel, rows, nframes, l, r, n_buffers, 2, (int)(flags & ZH_PAINT_ZERO_FIRST));
    }
    return zh_launch_status();
}

int zh_nice_paint_mix_stereo_batch(zh_nice *m, uint32_t start, uint32_t end, uint32_t n_buffers, float *const *mix_left,
                                   float *const *mix_right, zh_f32 gain_left, zh_f32 gain_right, const zh_bool *note_id_changed,
                                   const zh_nice_params *params, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    return nice_paint_mix_batch_impl(m, start, end, n_buffers, mix_left, mix_right, gain_left, gain_right, note_id_changed, params, flags);
}

static bool span_table_ok(const zh_span_table *t) {
    return t && t->max_spans > 0 && t->count && t->start && t->end && t->freq && t->note_on && t->note_id_changed;
}
static SpanTableP mk_span_table(const zh_span_table *t) {
    return SpanTableP{t->max_spans, t->count, t->start, t->end, t->freq, t->note_on, t->note_id_changed};
}

int zh_nice_paint_spans(zh_nice *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                        float sample_rate, const zh_span_table *table, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    if (!m || !outputs || end < start || !buf_covers(outputs[0], m->n, end) || !span_table_ok(table)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    hipStream_t st = m->ctx->stream;
    zh_nice_params p;
    memset(&p, 0, sizeof p);
    p.sample_rate = sample_rate;
    zh_bool no = {0, 0, nullptr};
    NiceArgs a = nice_args(m, &p, no);
    const bool zf = (flags & ZH_PAINT_ZERO_FIRST) != 0;
    const long wave_max = zh_form(ZF_NICE_WAVE_MAX);
    if ((long)m->n <= wave_max) {                            // few voices: one wave per voice, lanes = frames
        if (zf) ZH_LAUNCH(k_nice_spans_wave<true>, dim3(m->n), dim3(64), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
        else ZH_LAUNCH(k_nice_spans_wave<false>, dim3(m->n), dim3(64), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
    } else if (zf) ZH_LAUNCH(k_nice_spans<true>, seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
    else ZH_LAUNCH(k_nice_spans<false>, seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
    return zh_launch_status();
}

// ------------------------------------------------------------------ Noise -> Filter voice
int zh_noise_filter_create(zh_ctx *ctx, uint32_t n, uint64_t first_seed, zh_noise_filter **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_noise_filter *m = new (std::nothrow) zh_noise_filter();
    if (!m) return ZH_ERR_INVALID;
    memset(m, 0, sizeof *m);
    m->ctx = ctx; m->n = n;
    int rc = 0;
    for (int i = 0; i < 4 && !rc; i++) rc = dev_alloc(&m->s[i], n);
    if (!rc) rc = dev_alloc(&m->nb, (size_t)7 * n);
    if (!rc) rc = dev_alloc(&m->l, n);
    if (!rc) rc = dev_alloc(&m->b, n);
    if (!rc) rc = dev_alloc(&m->err, 1);
    if (!rc) rc = (int)hipMemsetAsync(m->err, 0, 4, ctx->stream);
    if (!rc && n) {
        rc = (int)hipMemsetAsync(m->nb, 0, (size_t)7 * n * 4, ctx->stream);
        if (!rc) rc = (int)hipMemsetAsync(m->l, 0, (size_t)n * 4, ctx->stream);
        if (!rc) rc = (int)hipMemsetAsync(m->b, 0, (size_t)n * 4, ctx->stream);
    }
    if (rc) { nf_free(m); delete m; return rc; }
    if (n) ZH_LAUNCH(k_nf_seed, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, m->s[0], m->s[1], m->s[2], m->s[3], n, first_seed);
    if (n) (void)zh_noise_jump_tables(ctx);                       // built on first use per context: here, not inside a paint (or a capture)
    *out = m;
    return zh_launch_status();
}
int zh_noise_filter_destroy(zh_noise_filter *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    nf_free(m);
    delete m;
    return ZH_OK;
}
int zh_noise_filter_get_state(zh_noise_filter *m, zh_noise_filter_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    uint32_t ring_error = 0;
    if (zh_download(m->ctx, &ring_error, m->err, 4) != ZH_OK || ring_error) return ZH_ERR_INVALID;
    std::vector<uint64_t> s;
    std::vector<float> nb, l, b;
    for (int i = 0; i < 4; i++) {
        int rc = down(m->ctx, s, m->s[i], m->n);
        if (rc) return rc;
        for (uint32_t v = 0; v < m->n; v++) host[v].noise.r[i] = s[v];
    }
    int rc = down(m->ctx, nb, m->nb, (size_t)7 * m->n);
    if (!rc) rc = down(m->ctx, l, m->l, m->n);
    if (!rc) rc = down(m->ctx, b, m->b, m->n);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->n; v++) {
        for (int j = 0; j < 7; j++) host[v].noise.b[j] = nb[(size_t)j * m->n + v];
        host[v].noise.reserved = 0;
        host[v].flt = zh_filter_state{l[v], b[v]};
    }
    return ZH_OK;
}
int zh_noise_filter_set_state(zh_noise_filter *m, const zh_noise_filter_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    m->pipe_n = 0;                                                    // (a pipelined recording's next paint starts from this state)
    const uint32_t n = m->n;
    std::vector<uint64_t> s(n);
    std::vector<float> nb((size_t)7 * n), l(n), b(n);
    for (int i = 0; i < 4; i++) {
        for (uint32_t v = 0; v < n; v++) s[v] = host[v].noise.r[i];
        int rc = up(m->ctx, m->s[i], s);
        if (rc) return rc;
    }
    for (uint32_t v = 0; v < n; v++) {
        for (int j = 0; j < 7; j++) nb[(size_t)j * n + v] = host[v].noise.b[j];
        l[v] = host[v].flt.l; b[v] = host[v].flt.b;
    }
    int rc = up(m->ctx, m->nb, nb);
    if (!rc) rc = up(m->ctx, m->l, l);
    if (!rc) rc = up(m->ctx, m->b, b);
    return rc;
}
int zh_noise_filter_paint(zh_noise_filter *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                          zh_bool note_id_changed, const zh_noise_filter_params *p, uint32_t flags) { ZH_GUARD_EPOCH(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;
    if (!m || !outputs || !p || end < start || !buf_covers(outputs[0], m->n, end)) return ZH_ERR_INVALID;
    if (p->color > ZH_NOISE_PINK || p->type > ZH_FILTER_ALL_PASS) return ZH_ERR_INVALID;
    if (m->n == 0 || end == start) return ZH_OK;
    const bool zf = flags & ZH_PAINT_ZERO_FIRST, pink = p->color == ZH_NOISE_PINK;
    hipStream_t st = m->ctx->stream;
    Img out = mk_img(outputs[0]);
    // the pipelined recording's chain (below): any paint that is not its next link ends it
    const uint32_t chain_n = (m->ctx->capturing && m->pipe_capture == m->ctx->capture_serial) ? m->pipe_n : 0u;
    m->pipe_n = 0;
    if (p->type == ZH_FILTER_BYPASS) {
        if (m->ctx->epoch_open) zh_epoch_barrier(m->ctx);
#define ZH_NFB(ZF_, PK_) ZH_LAUNCH((k_noise_filter_bypass<ZF_, PK_>), seq_grid(m->n), dim3(kSeqBlock), 0, st, m->s[0], m->s[1], m->s[2], m->s[3], m->nb, m->n, out, start, end)
        if (zf) { if (pink) ZH_NFB(true, true); else ZH_NFB(true, false); } else { if (pink) ZH_NFB(false, true); else ZH_NFB(false, false); }
#undef ZH_NFB
        return zh_launch_status();
    }
    float l_mul = 0.0f, b_mul = 0.0f, h_mul = 0.0f;                                 // Filter.zig:98-109
    switch (p->type) {
    case ZH_FILTER_LOW_PASS: l_mul = 1.0f; break;
    case ZH_FILTER_BAND_PASS: b_mul = 1.0f; break;
    case ZH_FILTER_HIGH_PASS: h_mul = 1.0f; break;
    case ZH_FILTER_NOTCH: l_mul = 1.0f; h_mul = 1.0f; break;
    default: l_mul = 1.0f; b_mul = 1.0f; h_mul = 1.0f; break;
    }
    // ZH_PAINT_TOLERANT, white noise, few voices: the span as 32..128-frame chunks at once, two passes (filter_tp.hip.h).  Pieces
    // of <= 32 chunks; L (a multiple of 32, the jump tables' step) by voice count: ~2,048 waves in flight.
    if ((flags & ZH_PAINT_TOLERANT) && !pink && end - start >= 128 && outputs[0].stride <= (1u << 24)) {
        const uint32_t Cw = zh_tp_chunks(m->n, ZF_NF_TP_MAX, 1024);                  // chunks wanted for a 1,024-frame buffer
        const uint4 *tables = Cw >= 2 ? zh_noise_jump_tables(m->ctx) : nullptr;
        if (tables && !m->tp_cs && !m->ctx->capturing) {
            int arc = dev_alloc(&m->tp_cs, (size_t)kNfTpMaxChunks * 4 * m->n);
            if (!arc) arc = dev_alloc(&m->tp_e, (size_t)(kNfTpMaxChunks + 1) * m->n);
            if (!arc) arc = dev_alloc(&m->tp_flag, m->n);
            if (!arc) arc = (int)hipMemsetAsync(m->tp_flag, 0, (size_t)m->n * 4, st);
            if (arc) { (void)hipFree(m->tp_cs); (void)hipFree(m->tp_e); (void)hipFree(m->tp_flag); m->tp_cs = nullptr; m->tp_e = nullptr; m->tp_flag = nullptr; (void)hipGetLastError(); }
            // what a pipelined recording needs beside it (optional: without it the paints are recorded one after the other)
            if (!arc) {
                int prc = dev_alloc(&m->tp_cs2, (size_t)kNfTpMaxChunks * 4 * m->n);
                if (!prc) prc = dev_alloc(&m->tp_e2, (size_t)(kNfTpMaxChunks + 1) * m->n);
                if (!prc) prc = dev_alloc(&m->tp_flag2, m->n);
                if (!prc) prc = (int)hipMemsetAsync(m->tp_flag2, 0, (size_t)m->n * 4, st);
                for (int q = 0; q < 2 && !prc; q++)
                    for (int i = 0; i < 4 && !prc; i++) prc = dev_alloc(&m->tp_pred[q][i], m->n);
                if (prc) { (void)hipFree(m->tp_cs2); m->tp_cs2 = nullptr; (void)hipGetLastError(); }   // (tp_cs2 == null: not pipelined; the rest is freed with the module)
            }
        }
        if (tables && m->tp_cs) {
            const uint32_t L = 32u * max(1u, 32u / Cw);                                 // a multiple of 32 draws: the jump tables' step
            NfTpArgs a;
            for (int i = 0; i < 4; i++) a.s[i] = m->s[i];
            a.l = m->l; a.b = m->b; a.cs = m->tp_cs; a.e = m->tp_e; a.flag = m->tp_flag; a.tables = tables;
            a.V = m->n; a.L = L; a.out = out;
            a.l_mul = l_mul; a.b_mul = b_mul; a.h_mul = h_mul; a.cutoff = mk_f32(p->cutoff); a.res = mk_f32(p->res);
            const uint32_t piece = min(kNfTpMaxChunks * L, ((uint32_t)kNoiseJumpTables * 32u / L) * L + L);   // chunk starts within the tables' reach: (C - 1) * L / 32 <= kNoiseJumpTables
            for (int i = 0; i < 4; i++) { a.s_in[i] = m->s[i]; a.pred[i] = nullptr; }
            a.e_next0 = nullptr; a.flag_next = nullptr; a.serial_next = 0; a.snapshot = 1;
            // Recorded in a ZH_CAPTURE_COALESCE capture, consecutive one-piece paints are PIPELINED: pass B of paint n is held back (common.hip.h
            // zh_co_batch) until the next paint of this module arrives, and then goes out in ONE launch with that paint's pass A
            // (k_nf_tp_ba) -- pass A of paint n + 1 needs nothing pass B of paint n makes (NfTpArgs), so one's launch ramp, table load and
            // jump hide behind the other's frame loops (20.5 -> 16 us per buffer at 4,096 voices).  When anything else is recorded, or the capture ends, the held pass B goes out on
            // its own.  Same kernels' code, same values as the paints one after the other; a voice that met one of Random.float's
            // multi-draw samples (2^-41 per sample) is walked sequentially -- the reference's own walk -- in every later paint of the chain.
            zh_ctx *ctx = m->ctx;
            if (ctx->capturing && (ctx->capture_flags & ZH_CAPTURE_COALESCE) && m->tp_cs2 && end - start <= piece) {
                {   // (pass A and pass B of two buffers share the launch: the chip is full with longer chunks too, and every chunk less
                    // saves 40 bytes of scratch traffic per voice -- nf_tp_pipe_frames)
                    const uint32_t want = (uint32_t)zh_form(ZF_NF_TP_PIPE_FRAMES) / 32u * 32u;
                    if (want >= 32u && want <= 512u && (end - start + want - 1) / want >= 2u && ((end - start + want - 1) / want - 1) * (want / 32u) <= (uint32_t)kNoiseJumpTables) a.L = want;
                }
                const uint32_t L = a.L;
                struct Pending { NfTpArgs b; uint32_t grid_b; bool zf; };
                zh_co_batch &cb = ctx->co;
                std::shared_ptr<Pending> pend = cb.active && cb.owner == m ? std::static_pointer_cast<Pending>(cb.items) : nullptr;
                const bool chained = pend && chain_n > 0;
                if (!chained && ctx->epoch_open) zh_epoch_barrier(ctx);                               // (another module's held batch: in order)
                const uint32_t n = chained ? chain_n : 0u, q = n & 1u;
                a.start = start; a.end = end; a.C = (end - start + L - 1) / L;
                if (++m->tp_serial == 0) m->tp_serial = 1;
                a.serial = m->tp_serial;
                a.serial_next = m->tp_serial + 1u == 0u ? 1u : m->tp_serial + 1u;
                a.per = (m->n + 255u) / 256u;
                a.cs = q ? m->tp_cs2 : m->tp_cs; a.e = q ? m->tp_e2 : m->tp_e; a.flag = q ? m->tp_flag2 : m->tp_flag;
                a.e_next0 = q ? m->tp_e : m->tp_e2; a.flag_next = q ? m->tp_flag : m->tp_flag2;
                a.snapshot = n == 0;
                for (int i = 0; i < 4; i++) { a.s_in[i] = n == 0 ? m->s[i] : m->tp_pred[q ^ 1u][i]; a.pred[i] = m->tp_pred[q][i]; }
                const uint32_t Ca = (a.C + kNfTpFusedSub - 1) / kNfTpFusedSub;                       // pass A's slots in the shared launch
                const uint32_t grid_a1 = ((a.C + 7u) / 8u) * 8u * a.per, grid_a = ((Ca + 7u) / 8u) * 8u * a.per, grid_b = 8u * a.C * ((a.per + 7u) / 8u);
                if (chained) {                                                                          // pass B of paint n - 1 + this paint's pass A
                    const Pending pb = *pend;
                    cb.active = false; cb.imgs.clear(); cb.items.reset(); cb.launch = nullptr;        // (taken over: not launched by the flush)
                    ctx->co_launches++;
                    if (pb.zf) ZH_LAUNCH(k_nf_tp_ba<true>, dim3(pb.grid_b + grid_a), dim3(256), 0, st, pb.b, a, grid_a);
                    else ZH_LAUNCH(k_nf_tp_ba<false>, dim3(pb.grid_b + grid_a), dim3(256), 0, st, pb.b, a, grid_a);
                } else {
                    ctx->co_launches++;
                    ZH_LAUNCH(k_nf_tp_a, dim3(grid_a1), dim3(256), 0, st, a);
                }
                auto np = std::make_shared<Pending>(Pending{a, grid_b, zf});
                cb.active = true; cb.owner = m; cb.start = start; cb.end = end; cb.stride = 0; cb.key = 0; cb.flips = false;
                cb.items = np; cb.imgs.assign(1, out.p);
                cb.launch = [np](hipStream_t s2, float *const *, uint32_t) {
                    if (np->zf) ZH_LAUNCH(k_nf_tp_b<true>, dim3(np->grid_b), dim3(256), 0, s2, np->b);
                    else ZH_LAUNCH(k_nf_tp_b<false>, dim3(np->grid_b), dim3(256), 0, s2, np->b);
                };
                ctx->epoch_open = true;
                ctx->co_paints++;
                m->pipe_capture = ctx->capture_serial; m->pipe_n = n + 1;
                return zh_launch_status();
            }
            if (ctx->epoch_open) zh_epoch_barrier(ctx);
            for (uint32_t s0 = start; s0 < end; s0 += piece) {
                a.start = s0; a.end = min(s0 + piece, end);
                a.C = (a.end - a.start + L - 1) / L;
                if (++m->tp_serial == 0) m->tp_serial = 1;
                a.serial = m->tp_serial;
                a.per = (m->n + 255u) / 256u;
                const dim3 grid(((a.C + 7u) / 8u) * 8u * a.per);                         // chunk j of every group on XCD j % 8 (nf_tp_block)
                const dim3 grid_b(8u * a.C * ((a.per + 7u) / 8u));                         // every chunk of a voice group on XCD g % 8 (nf_tp_block_b)
                ZH_LAUNCH(k_nf_tp_a, grid, dim3(256), 0, st, a);
                if (zf) ZH_LAUNCH(k_nf_tp_b<true>, grid_b, dim3(256), 0, st, a);
                else ZH_LAUNCH(k_nf_tp_b<false>, grid_b, dim3(256), 0, st, a);
            }
            return zh_launch_status();
        }
    }
    if (m->ctx->epoch_open) zh_epoch_barrier(m->ctx);                 // an ordered paint: after what was held back
    // up to ZH_NF_PC_MAX voices (default 65,536: measured 75 vs 110 us at 4,096 voices, 111 vs 133 us at 65,536, equal at
    // 131,072) the noise and the filter run in two waves side by side (k_noise_filter_pc); above, one wave does both
    const uint32_t pc_max = (uint32_t)zh_form(ZF_NF_PC_MAX);
    // White noise at small voice counts: three producer waves, a filter wave and a writer wave per 64 voices
    // (k_noise_filter_ring).  ZH_NF_RING_MAX: largest voice count that takes it.
    const uint32_t ring_max = (uint32_t)zh_form(ZF_NF_RING_MAX);
    if (!pink && m->n <= ring_max && end - start >= 128 && outputs[0].stride <= (1u << 24)) {
        const uint4 *tables = zh_noise_jump_tables(m->ctx);
        if (tables) {
            NfArgs a;
            for (int i = 0; i < 4; i++) a.s[i] = m->s[i];
            a.l = m->l; a.b = m->b; a.err = m->err;
            a.table256 = tables + (size_t)7 * kNoiseJumpEntries;         // table j - 1 holds T^(32 j)
            a.V = m->n; a.start = start; a.end = end; a.out = out;
            a.l_mul = l_mul; a.b_mul = b_mul; a.h_mul = h_mul; a.cutoff = mk_f32(p->cutoff); a.res = mk_f32(p->res);
            const dim3 grid((m->n + 63) / 64), block(64 * (kNfProducers + 2));
            if (zf) ZH_LAUNCH(k_noise_filter_ring<true>, grid, block, 0, st, a);
            else ZH_LAUNCH(k_noise_filter_ring<false>, grid, block, 0, st, a);
            return zh_launch_status();
        }
    }
#define ZH_NF(K_, BLK_, ZF_, PK_) ZH_LAUNCH((K_<ZF_, PK_>), seq_grid(m->n), dim3(BLK_), 0, st, m->s[0], m->s[1], m->s[2], m->s[3], m->nb, m->l, m->b, m->n, out, start, end, l_mul, b_mul, h_mul, mk_f32(p->cutoff), mk_f32(p->res))
    if (m->n <= pc_max && outputs[0].stride <= (1u << 24)) {                            // (32-row tiles: 32-bit offsets)
        if (zf) { if (pink) ZH_NF(k_noise_filter_pc, 128, true, true); else ZH_NF(k_noise_filter_pc, 128, true, false); }
        else { if (pink) ZH_NF(k_noise_filter_pc, 128, false, true); else ZH_NF(k_noise_filter_pc, 128, false, false); }
    } else {
        if (zf) { if (pink) ZH_NF(k_noise_filter, kSeqBlock, true, true); else ZH_NF(k_noise_filter, kSeqBlock, true, false); }
        else { if (pink) ZH_NF(k_noise_filter, kSeqBlock, false, true); else ZH_NF(k_noise_filter, kSeqBlock, false, false); }
    }
#undef ZH_NF
    return zh_launch_status();
}

// ------------------------------------------------------------------ PMOscInstrument
int zh_pmosc_create(zh_ctx *ctx, uint32_t n, zh_f32 release_duration, zh_pmosc **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_pmosc *m = new (std::nothrow) zh_pmosc();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->n = n; m->cur = 0; m->cnt[0] = m->cnt[1] = nullptr; m->id = 0; m->words = 6;
    m->release_duration = nullptr;
    int rc = dev_alloc(&m->release_duration, n);
    if (!rc) rc = dev_alloc(&m->cnt[0], (size_t)6 * n);
    if (!rc) rc = dev_alloc(&m->cnt[1], (size_t)6 * n);
    if (rc) { pmosc_free(m); delete m; return rc; }
    if (n) {
        hipStream_t st = ctx->stream;
        ZH_LAUNCH(k_fill_f32, dim3((n + 255) / 256), dim3(256), 0, st, m->release_duration, n, mk_f32(release_duration));
        for (int b = 0; b < 2; b++) { hipError_t e = hipMemsetAsync(m->cnt[b], 0, (size_t)6 * n * 4, st); if (e != hipSuccess) { pmosc_free(m); delete m; return (int)e; } }
    }
    m->view();
    zh_flipper_register(m);
    *out = m;
    return zh_launch_status();
}
int zh_pmosc_destroy(zh_pmosc *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    zh_flipper_unregister(m);
    pmosc_free(m);
    delete m;
    return ZH_OK;
}
int zh_pmosc_get_state(zh_pmosc *m, zh_pmosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    m->view();
    std::vector<uint32_t> es;
    std::vector<float> tc, tm, t, lv, sv;
    int rc = down(m->ctx, tc, m->tc, m->n);
    if (!rc) rc = down(m->ctx, tm, m->tm, m->n);
    if (!rc) rc = down(m->ctx, es, m->estate, m->n);
    if (!rc) rc = down(m->ctx, t, m->et, m->n);
    if (!rc) rc = down(m->ctx, lv, m->elast, m->n);
    if (!rc) rc = down(m->ctx, sv, m->estart, m->n);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->n; v++) {
        host[v].carrier.t = tc[v]; host[v].modulator.t = tm[v];
        host[v].env = zh_envelope_state{es[v], t[v], lv[v], sv[v]};
    }
    return ZH_OK;
}
int zh_pmosc_set_state(zh_pmosc *m, const zh_pmosc_state *host) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !host) return ZH_ERR_INVALID;
    m->view();
    const uint32_t n = m->n;
    std::vector<uint32_t> es(n);
    std::vector<float> tc(n), tm(n), t(n), lv(n), sv(n);
    for (uint32_t v = 0; v < n; v++) {
        tc[v] = host[v].carrier.t; tm[v] = host[v].modulator.t;
        es[v] = host[v].env.state; t[v] = host[v].env.t; lv[v] = host[v].env.last_value; sv[v] = host[v].env.start;
    }
    int rc = up(m->ctx, m->tc, tc);
    if (!rc) rc = up(m->ctx, m->tm, tm);
    if (!rc) rc = up(m->ctx, m->estate, es);
    if (!rc) rc = up(m->ctx, m->et, t);
    if (!rc) rc = up(m->ctx, m->elast, lv);
    if (!rc) rc = up(m->ctx, m->estart, sv);
    return rc;
}
int zh_pmosc_paint(zh_pmosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                   zh_bool note_id_changed, const zh_pmosc_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    if (!m || !outputs || !p || end < start || !buf_covers(outputs[0], m->n, end)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    hipStream_t st = m->ctx->stream;
    m->view();
    PMOscArgs a{m->release_duration, m->tc, m->tm, m->estate, m->et, m->elast, m->estart, m->n, p->sample_rate,
                mk_f32(p->freq), mk_bool(p->note_on), mk_bool(note_id_changed)};
    // few voices: frame ranges at once (k_pmosc_ranges); ZH_PMOSC_RANGES = number of ranges, 0 = never
    // 16 / 32 / 64 ranges: 75.5 / 70.9 / 80.7 us at 4,096 voices (the replay costs the same whatever the count); 24,576 /
    // 32,768 / 65,536 / 131,072 voices, sequential -> ranges: 370 -> 150, 370 -> 176, 353 -> 277, 499 -> 476 us
    const uint32_t ch = zh_range_frames(m->n, end - start, ZF_PMOSC_RANGES, m->n <= 16384 ? 2048 : 4096, 131072);
    if (ch) {
        const dim3 grid((m->n + 63) / 64, (end - start + ch - 1) / ch);
        uint32_t *next = m->cnt[m->cur ^ 1];
        if (flags & ZH_PAINT_TOLERANT) {
            if (flags & ZH_PAINT_ZERO_FIRST) ZH_LAUNCH((k_pmosc_ranges<true, true>), grid, dim3(64), 0, st, a, next, mk_img(outputs[0]), start, end, ch);
            else ZH_LAUNCH((k_pmosc_ranges<false, true>), grid, dim3(64), 0, st, a, next, mk_img(outputs[0]), start, end, ch);
        } else if (flags & ZH_PAINT_ZERO_FIRST) ZH_LAUNCH(k_pmosc_ranges<true>, grid, dim3(64), 0, st, a, next, mk_img(outputs[0]), start, end, ch);
        else ZH_LAUNCH(k_pmosc_ranges<false>, grid, dim3(64), 0, st, a, next, mk_img(outputs[0]), start, end, ch);
        zh_flipper_painted(m);
        m->cur ^= 1;
        m->view();
        return zh_launch_status();
    }
    if (flags & ZH_PAINT_TOLERANT) {
        if (flags & ZH_PAINT_ZERO_FIRST) ZH_LAUNCH((k_pmosc<true, true>), seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_img(outputs[0]), start, end);
        else ZH_LAUNCH((k_pmosc<false, true>), seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_img(outputs[0]), start, end);
    } else if (flags & ZH_PAINT_ZERO_FIRST) ZH_LAUNCH(k_pmosc<true>, seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_img(outputs[0]), start, end);
    else ZH_LAUNCH(k_pmosc<false>, seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_img(outputs[0]), start, end);
    return zh_launch_status();
}

int zh_pmosc_paint_spans(zh_pmosc *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                         float sample_rate, const zh_span_table *table, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps;
    if (!m || !outputs || end < start || !buf_covers(outputs[0], m->n, end) || !span_table_ok(table)) return ZH_ERR_INVALID;
    if (m->n == 0) return ZH_OK;
    zh_flipper_used(m);                     // a capture must know the state buffer this paint starts from, flip or not (ctx.hip)
    hipStream_t st = m->ctx->stream;
    m->view();
    PMOscArgs a{m->release_duration, m->tc, m->tm, m->estate, m->et, m->elast, m->estart, m->n, sample_rate,
                F32P{0.0f, nullptr}, BoolP{0, nullptr}, BoolP{0, nullptr}};
    const bool zf = (flags & ZH_PAINT_ZERO_FIRST) != 0;
    const long wave_max = zh_form(ZF_PMOSC_WAVE_MAX);
    if ((long)m->n <= wave_max) {                            // few voices: one wave per voice, lanes = frames
        if (zf) ZH_LAUNCH(k_pmosc_spans_wave<true>, dim3(m->n), dim3(64), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
        else ZH_LAUNCH(k_pmosc_spans_wave<false>, dim3(m->n), dim3(64), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
    } else if (zf) ZH_LAUNCH(k_pmosc_spans<true>, seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
    else ZH_LAUNCH(k_pmosc_spans<false>, seq_grid(m->n), dim3(kSeqBlock), 0, st, a, mk_span_table(table), mk_img(outputs[0]), start, end);
    return zh_launch_status();
}

}  // extern "C"
