// nice.hip.h -- the NiceInstrument voice (examples/modules.zig:189-248) as per-lane device code: shared by composite.hip (the exact
// forms, -ffp-contract=off) and nice_mix_fma.hip (the ZH_PAINT_TOLERANT mixdown of many voices, compiled with contraction on).
#pragma once
#include "common.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "seq.hip.h"
#include "envelope.hip.h"
#include "voices.hip.h"

// ------------------------------------------------------------------ NiceInstrument voice
// W voices per lane (lanes.hip.h): W = 2 turns the f32 arithmetic of a voice pair into packed ops.
template <int W>
struct NiceLaneT {
    using F = typename LaneT<W>::F;
    using U = typename LaneT<W>::U;
    using M = typename LaneT<W>::M;
    // PulseOsc
    U cnt;
    PulseKT<W> k;
    M bad;
    F g, ng;                                                          // +-0.7, or +0.0 / +0.0 for a silent voice
    // Filter
    F l, b, cut, res;
    EnvLaneT<W, ZH_CURVE_CUBED> env;                                  // all three curves are cubed (:238-245)

    __device__ __forceinline__ void begin(float sample_rate, float srf, float sr8, F freq, F color, M note_on, M new_note) {
        bad = zor(freq < zsplat<F>(0.0f), freq > zsplat<F>(sr8));      // PulseOsc.zig:82-84
        pulse_setup(k, srf, freq, color);
        // A silent voice paints nothing and keeps its phase counter.  Expressed in the constants instead of two
        // selects per sample: with everything zeroed, b0 = b1 = b2 = false, the sample is ngain = +0.0 and
        // cnt + ifreq == cnt.
        const F zf = zsplat<F>(0.0f);
        const typename LaneT<W>::U zu = zsplatu<typename LaneT<W>::U>(0u);
        k.ifreq = zsel(bad, zu, k.ifreq); k.brpt = zsel(bad, zu, k.brpt);
        k.gdf2 = zsel(bad, zf, k.gdf2); k.cc121 = zsel(bad, zf, k.cc121); k.cc212 = zsel(bad, zf, k.cc212);
        g = zsel(bad, zf, zsplat<F>(0.7f)); ng = zsel(bad, zf, zsplat<F>(-0.7f));
        // Filter params: cutoff = cutoffFromFrequency(freq * 8, sr), res = 0.7 (examples/modules.zig:231-235)
        const F f8 = freq * 8.0f;
#pragma unroll
        for (int c = 0; c < W; c++) zput(cut, c, zclampf(zcutoff_from_frequency(zget(f8, c), sample_rate), 0.0f, 1.0f));
        res = zsplat<F>(1.0f - zclampf(0.7f, 0.0f, 1.0f));
        // Envelope params (:238-245)
        env.sample_rate = sample_rate;
        env.sustain_volume = zsplat<F>(0.8f);
        env.attack = CurvePT<W>{ZH_CURVE_CUBED, zsplat<F>(0.01f)};
        env.decay = CurvePT<W>{ZH_CURVE_CUBED, zsplat<F>(0.1f)};
        env.release = CurvePT<W>{ZH_CURVE_CUBED, zsplat<F>(0.5f)};
        env.note_on = note_on;
        env.begin(new_note);
    }

    // One frame of examples/modules.zig:220-246 in two parts.  osc(c): the oscillator half -- temps[0] for
    // the frame whose phase counter is c, a pure function of c (the u32 counter of frame j is exactly
    // cnt + j*ifreq), so k_nice_spans_wave evaluates it for 64 frames at once.  tail(t0): the filter and
    // the envelope, which carry state from frame to frame; returns env*flt (the value added to out).
    // (temps[0] = 0 + pulse, then * 0.5 :226: the `0 +` is not computed.  It changes the bits only of a pulse of -0.0 -- into
    // +0.0 -- and this value's one and only use is the filter's `input + fcdcoffset` (Filter.zig:135, svf_step): a zero of
    // either sign plus 2^-18 is 2^-18.  A silent voice (bad freq) yields ngain = +0.0.)
    __device__ __forceinline__ F osc(U c) const {
        const F pv = pulse_sample<W>(k, c, g, ng);
        return pv * 0.5f;                                              // multiplyWithScalar :226
    }
    // tail = two chains that never read each other's state: the filter over the oscillator samples and
    // the envelope; their product is the value added to out.
    // INTO_ZERO: the caller adds the frame's value to a ZERO (a ZERO_FIRST paint, the mixdown's `0.0f + x`).  The low-pass
    // mix's own `0 +` (temps[1] was zeroed) is then not computed: it only turns a mix of -0.0 into +0.0, the product with
    // the envelope is a zero of some sign either way (or the same NaN), and the caller's `0 + product` is +0.0 for both.
    template <bool INTO_ZERO = false>
    __device__ __forceinline__ F tail_filter(F t0) {
        // temps[1] = 0 + low-pass(temps[0])   (Filter.zig:135-146 with l_mul = 1, b_mul = h_mul = 0)
        const SvfOutT<F> s = svf_step(l, b, t0, cut, res);
        if constexpr (INTO_ZERO) return s.l + s.b * 0.0f;
        else return svf_lowpass_into_zero(s.l, s.b);
    }
    __device__ __forceinline__ F tail_env() { return env.frame_masked(); }   // temps[0] = 0 (+ envelope)
    template <bool INTO_ZERO = false>
    __device__ __forceinline__ F tail(F t0) {
        const F t1 = tail_filter<INTO_ZERO>(t0);
        const F e0 = tail_env();
        return e0 * t1;                                                // multiply :246: out += temps[0]*temps[1]
    }
    template <bool INTO_ZERO = false>
    __device__ __forceinline__ F frame() {
        const F t0 = osc(cnt);
        cnt = cnt + k.ifreq;
        return tail<INTO_ZERO>(t0);
    }
    // While no voice of a wave is inside a timed envelope stage (sustain: paintFlat's constant, Envelope.zig:68-70; idle or
    // the assert case: nothing painted) the envelope's frame changes no state and yields the same value every frame:
    // env_quiet() is that value, frame_quiet() the frame without the envelope's ~12 instructions.  Exactly what frame()
    // computes and commits in those modes (frame_masked: every update is a select on `toward`).
    __device__ __forceinline__ F env_quiet() const { return zbits_f(zbits_u(zsplat<F>(0.0f) + env.sustain_volume) & env.m_painted); }
    template <bool INTO_ZERO = false>
    __device__ __forceinline__ F frame_quiet(F e0) {
        const F t0 = osc(cnt);
        cnt = cnt + k.ifreq;
        return e0 * tail_filter<INTO_ZERO>(t0);
    }
    // the same two with the carried-mask oscillator (roll_begin() first; every lane takes every frame)
    template <bool INTO_ZERO = false>
    __device__ __forceinline__ F frame_roll(PulseRoll &roll) { return tail<INTO_ZERO>(osc_next(roll)); }
    template <bool INTO_ZERO = false>
    __device__ __forceinline__ F frame_quiet_roll(F e0, PulseRoll &roll) { return e0 * tail_filter<INTO_ZERO>(osc_next(roll)); }
    // The oscillator half for a walker in which every active lane takes every frame in order from a begin()
    // executed by all of them together (k_nice, k_nice_pc -- not the span kernels, whose lanes begin sub-spans
    // at different frames): the previous frame's half-period bit is carried as the wave's lane
    // mask (dsp.hip.h pulse_sample_roll) instead of being recomputed.  roll_begin() after begin().
    __device__ __forceinline__ void roll_begin(PulseRoll &roll) const {
        static_assert(W == 1, "lane masks: one voice per lane");
        roll = pulse_roll_init(k, cnt);
    }
    __device__ __forceinline__ F osc_next(PulseRoll &roll) {
        const F pv = pulse_sample_roll(k, cnt, roll, g, ng);             // (no `0 +`: see osc())
        cnt = cnt + k.ifreq;
        return pv * 0.5f;
    }
};
using NiceLane = NiceLaneT<1>;

struct NiceArgs {
    float *color;
    uint32_t *cnt;
    float *fl, *fb;
    uint32_t *estate;
    float *et, *elast, *estart;
    uint32_t V;
    float sample_rate, srf, sr8;
    F32P freq;
    BoolP note_on, nic;
};

// `v` = the lane's first voice (W = 2: even)
// the seven state words a paint loads and stores (what carries a voice from one paint call to the next)
template <int W>
__device__ __forceinline__ void nice_load_state(NiceLaneT<W> &n, const NiceArgs &a, uint32_t v) {
    n.cnt = zload_u<W>(a.cnt, v); n.l = zload_f<W>(a.fl, v); n.b = zload_f<W>(a.fb, v);
    n.env.state = zload_u<W>(a.estate, v); n.env.t = zload_f<W>(a.et, v);
    n.env.last_value = zload_f<W>(a.elast, v); n.env.start = zload_f<W>(a.estart, v);
}
template <int W>
__device__ __forceinline__ void nice_load(NiceLaneT<W> &n, const NiceArgs &a, uint32_t v) {
    nice_load_state(n, a, v);
    n.begin(a.sample_rate, a.srf, a.sr8, zget_f32p<W>(a.freq, v), zload_f<W>(a.color, v), zget_boolp<W>(a.note_on, v), zget_boolp<W>(a.nic, v));
}
template <int W>
__device__ __forceinline__ void nice_store(NiceLaneT<W> &n, const NiceArgs &a, uint32_t v) {
    zstore_u<W>(a.cnt, v, n.cnt); zstore_f<W>(a.fl, v, n.l); zstore_f<W>(a.fb, v, n.b);
    zstore_u<W>(a.estate, v, n.env.state); zstore_f<W>(a.et, v, n.env.t);
    zstore_f<W>(a.elast, v, n.env.last_value); zstore_f<W>(a.estart, v, n.env.start);
}

// (Two voices per lane -- W = 2, lanes.hip.h: packed f32 arithmetic -- was built and parity-tested through round 4 and slower at
// every voice count: 131,072 voices 211 us vs 182, 1 Mi voices 1296 vs 1222; tools/ubench/valu_ops.hip shows why: a v_pk_*_f32
// costs 4.2 issue cycles per SIMD against 2.5 for a plain op, and every compare and select is paid twice.  Its dispatch and
// its switch are gone (round 5); the W-generic per-sample code stays.)
