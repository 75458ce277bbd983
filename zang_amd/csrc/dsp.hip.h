// dsp.hip.h -- the per-sample building blocks shared by the module kernels and the fused voices, so
// that each of the reference's formulas exists exactly once on the device.
#pragma once
#include "zmath.hip.h"
#include "lanes.hip.h"

// ---- PulseOsc (src/modules/PulseOsc.zig) -------------------------------------------------------
template <int W>
struct PulseKT {          // per-voice constants of PulseOsc.zig:88-95
    typename LaneT<W>::U ifreq, brpt;
    typename LaneT<W>::F gdf2 /* gdf * 2.0 */, col, cc121, cc212;
};
using PulseK = PulseKT<1>;

// `srf` = fc32bit / sample_rate (PulseOsc.zig:87) is wave-uniform: the host computes it once
// (IEEE f32 divide, same bits as the device's correctly rounded divide).
__device__ __forceinline__ void pulse_setup_freq(PulseK &k, float srf, float freq) {
    const float gain = 0.7f;
    k.ifreq = zf32_to_u32(srf * freq);
    k.gdf2 = (gain / zutof23(k.ifreq)) * 2.0f;
    k.cc121 = k.gdf2 * (k.col - 1.0f) + gain;
    k.cc212 = k.gdf2 * k.col - gain;
}
__device__ __forceinline__ void pulse_setup_color(PulseK &k, float color) {
    k.brpt = zftou32(zclamp01(color));
    k.col = zutof23(k.brpt);
}
// the once-per-paint setup of a voice pair is done per component by the scalar code above
__device__ __forceinline__ void pulse_setup(PulseKT<1> &k, float srf, float freq, float color) {
    pulse_setup_color(k, color);
    pulse_setup_freq(k, srf, freq);
}
__device__ __forceinline__ void pulse_setup(PulseKT<2> &k, float srf, zf2 freq, zf2 color) {
    PulseK a, b;
    pulse_setup(a, srf, freq.x, color.x);
    pulse_setup(b, srf, freq.y, color.y);
    k.ifreq = zu2{a.ifreq, b.ifreq}; k.brpt = zu2{a.brpt, b.brpt};
    k.gdf2 = zf2{a.gdf2, b.gdf2}; k.col = zf2{a.col, b.col};
    k.cc121 = zf2{a.cc121, b.cc121}; k.cc212 = zf2{a.cc212, b.cc212};
}

__device__ __forceinline__ zf2 zutof23(zu2 x) { return zbits_f((x >> 9) | 0x3f800000u) - 1.0f; }

// The 6-way switch of PulseOsc.zig:102-110.  transition = b0 | b1<<1 | b2<<2 with
// b0 = cnt < brpt, b1 = (cnt - ifreq) < brpt, b2 = cnt < ifreq:
//   b0 == b1: flat -> 3: gain, 0: -gain, 7: cc121, 4: cc212
//   b0 != b1: ramp -> 2: gdf*2*(col-p) + gain, 5: gdf*2*p - gain (x - gain == x + (-gain) exactly).
// Transitions 1 and 6 (`else => unreachable`, :109) cannot occur for ANY u32 cnt, ifreq, brpt:
//   1 = (b0, !b1, !b2): !b2 means cnt >= ifreq, so cnt - ifreq does not wrap and is <= cnt < brpt => b1.
//   6 = (!b0, b1, b2):  b2 means cnt < ifreq, so cnt - ifreq = cnt + 2^32 - ifreq >= cnt >= brpt => !b1.
// Hence b0 != b1 implies the ramp case and no third arm is needed.
// Values are selected, never blended (gdf is inf when ifreq < 512).
// The result is never -0.0: every arm ends in `x + gain` / `x - gain` with gain = 0.7, and an IEEE
// sum is -0.0 only when both addends are -0.0.
// `gain` / `ngain` are +-0.7 (PulseOsc.zig:88); a caller may pass per-voice values to silence a voice without a
// per-sample select: with k zeroed and gain = ngain = +0.0 every sample is +0.0 and cnt + ifreq == cnt.
template <int W>
__device__ __forceinline__ typename LaneT<W>::F pulse_sample(const PulseKT<W> &k, typename LaneT<W>::U cnt,
                                                             typename LaneT<W>::F gain, typename LaneT<W>::F ngain) {
    using F = typename LaneT<W>::F;
    using M = typename LaneT<W>::M;
    const F p = zutof23(cnt);
    const M b0 = cnt < k.brpt;
    const M b1 = (cnt - k.ifreq) < k.brpt;
    const M b2 = cnt < k.ifreq;
    // sg = the flat level of this half period.  In the two ramp cases b2 == b0 (table above), and the ramp's
    // offset is -sg: `x + (-0.7)` and `x - 0.7` are the same IEEE operation, so one select serves both.
    const F sg = zsel(b0, gain, ngain);
    const F ramp = k.gdf2 * zsel(b0, p, k.col - p) - sg;
    const F flat = zsel(b2, zsel(b0, k.cc121, k.cc212), sg);
    return zsel(b0 == b1, flat, ramp);
}

template <int W>
__device__ __forceinline__ typename LaneT<W>::F pulse_sample(const PulseKT<W> &k, typename LaneT<W>::U cnt) {
    using F = typename LaneT<W>::F;
    return pulse_sample<W>(k, cnt, zsplat<F>(0.7f), zsplat<F>(-0.7f));
}

// The same sample for a walker that visits consecutive frames: b1 of this frame is b0 of the previous one
// (cnt_prev = cnt - ifreq exactly, in u32), the reference's own rolling 2-bit state (PulseOsc.zig:97-101).
// The carried bit travels as the wave's 64-bit lane mask in an SGPR pair (ballot / inverse ballot): comparing it
// with this frame's mask is one scalar xnor, and no per-lane 0/1 value is ever materialised.  `prev` must start
// as pulse_roll_init(k, cnt) at the first frame of a paint call (the per-voice constants may have changed since
// the last call).  Lanes that are inactive at a ballot contribute 0 bits that only they would read.
// (Not for hiprtc: its compiler lacks the inverse-ballot builtin; generated script kernels use pulse_sample.)
#if !defined(__HIPCC_RTC__)
typedef unsigned long long PulseRoll;
__device__ __forceinline__ PulseRoll pulse_roll_init(const PulseK &k, uint32_t cnt) {
    return __builtin_amdgcn_ballot_w64((cnt - k.ifreq) < k.brpt);
}
__device__ __forceinline__ float pulse_sample_roll(const PulseK &k, uint32_t cnt, PulseRoll &prev, float gain = 0.7f, float ngain = -0.7f) {
#if defined(ZH_NO_PULSE_ROLL)                                        // A/B builds: the stateless form
    (void)prev;
    return pulse_sample<1>(k, cnt, gain, ngain);
#endif
    const float p = zutof23(cnt);
    const bool b0 = cnt < k.brpt;
    const bool b2 = cnt < k.ifreq;
    const float sg = b0 ? gain : ngain;
    const float ramp = k.gdf2 * (b0 ? p : k.col - p) - sg;
    const float flat = b2 ? (b0 ? k.cc121 : k.cc212) : sg;
    const PulseRoll m0 = __builtin_amdgcn_ballot_w64(b0);
    const bool same = __builtin_amdgcn_inverse_ballot_w64(~(m0 ^ prev));   // b0 == b1
    prev = m0;
    return same ? flat : ramp;
}
#endif   // !__HIPCC_RTC__

// ---- Filter (src/modules/Filter.zig:130-146): one 2x-oversampled state-variable step ------------
template <class F> struct SvfOutT { F l, b, h; };
using SvfOut = SvfOutT<float>;
constexpr float kSvfDcOffset = 3.814697265625e-6f;                   // fcdcoffset, Filter.zig:8
// the state-carrying part, given in = input + fcdcoffset (:135, a function of the input sample alone)
template <class F>
__device__ __forceinline__ SvfOutT<F> svf_core(F &l, F &b, F in, F cut, F res) {
    l += cut * b - kSvfDcOffset;                                      // :138
    b += cut * (in - b * res - l);                                    // :139
    l += cut * b;                                                     // :142
    const F h = in - b * res - l;                                     // :143
    b += cut * h;                                                     // :144
    return SvfOutT<F>{l, b, h};
}
// svf_core in two parts for the wave pipelines (k_filter_pc): the recurrence wave runs svf_core_mid, which carries (l, b)
// through the whole step but hands on only l after :142 and b after :139; the writer wave redoes :143-144 from them
// (svf_finish: the same operations on the same values) for the h and the final b of the output mix.
template <class F> struct SvfMidT { F l, b1; };
using SvfMid = SvfMidT<float>;
template <class F>
__device__ __forceinline__ SvfMidT<F> svf_core_mid(F &l, F &b, F in, F cut, F res) {
    l += cut * b - kSvfDcOffset;                                      // :138
    b += cut * (in - b * res - l);                                    // :139
    l += cut * b;                                                     // :142
    const F b1 = b;
    const F h = in - b * res - l;                                     // :143
    b += cut * h;                                                     // :144
    return SvfMidT<F>{l, b1};
}
template <class F>
__device__ __forceinline__ SvfOutT<F> svf_finish(F l, F b1, F in, F cut, F res) {
    const F h = in - b1 * res - l;                                    // :143
    return SvfOutT<F>{l, b1 + cut * h, h};                            // :144
}
template <class F>
__device__ __forceinline__ SvfOutT<F> svf_step(F &l, F &b, F input, F cut, F res) {
    return svf_core(l, b, input + kSvfDcOffset, cut, res);
}
// `0 + (l * 1 + b * 0 + h * 0)`: the low-pass mix (Filter.zig:98-109, 146: l_mul = 1, b_mul = h_mul = 0) added into a ZEROED temp,
// as the composites do (examples/modules.zig:229-236, 439) -- in three operations instead of six, same bits for every input:
// l * 1 is l.  After a step h is non-finite only if b is (:144 comes last: finite + cut * non-finite is non-finite for every
// cut in [0, 1], 0 * inf = NaN included), so b * 0 alone decides between a NaN and zeros; with b finite the two products are
// zeros, l plus zeros of any sign is l for l != 0 (infinities included) and some zero otherwise, and the outer `0 +` turns
// that zero into +0 whatever its sign.  (Not for a mix added into a live output: there the sign of a zero sum shows.)
template <class F>
__device__ __forceinline__ F svf_lowpass_into_zero(F l, F b) { return zsplat<F>(0.0f) + (l + b * 0.0f); }

// ---- Noise (src/modules/Noise.zig:58-66): one sample of Paul Kellett's pink filter -------------
__device__ __forceinline__ float pink_step(float (&b)[7], float white) {
    b[0] = 0.99886f * b[0] + white * 0.0555179f;
    b[1] = 0.99332f * b[1] + white * 0.0750759f;
    b[2] = 0.96900f * b[2] + white * 0.1538520f;
    b[3] = 0.86650f * b[3] + white * 0.3104856f;
    b[4] = 0.55000f * b[4] + white * 0.5329522f;
    b[5] = -0.7616f * b[5] - white * 0.0168980f;
    const float out = b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + white * 0.5362f;   // :65
    b[6] = white * 0.115926f;                                         // :66
    return out;
}
