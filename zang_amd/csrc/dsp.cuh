// dsp.cuh -- the per-sample building blocks shared by the module kernels and the fused voices, so
// that each of the reference's formulas exists exactly once on the device.
#pragma once
#include "zmath.cuh"

// ---- PulseOsc (src/modules/PulseOsc.zig) -------------------------------------------------------
struct PulseK {           // per-voice constants of PulseOsc.zig:88-95
    uint32_t ifreq, brpt;
    float gdf2 /* gdf * 2.0 */, col, cc121, cc212;
};

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
__device__ __forceinline__ float pulse_sample(const PulseK &k, uint32_t cnt) {
    const float gain = 0.7f;
    const float p = zutof23(cnt);
    const bool b0 = cnt < k.brpt;
    const bool b1 = (uint32_t)(cnt - k.ifreq) < k.brpt;
    const bool b2 = cnt < k.ifreq;
    const float ramp = k.gdf2 * (b2 ? p : k.col - p) + (b2 ? -gain : gain);
    const float flat = b2 ? (b0 ? k.cc121 : k.cc212) : (b0 ? gain : -gain);
    return (b0 == b1) ? flat : ramp;
}

// ---- Filter (src/modules/Filter.zig:130-146): one 2x-oversampled state-variable step ------------
struct SvfOut { float l, b, h; };
__device__ __forceinline__ SvfOut svf_step(float &l, float &b, float input, float cut, float res) {
    const float fcdcoffset = 3.814697265625e-6f;                      // Filter.zig:8
    const float in = input + fcdcoffset;                              // :135
    l += cut * b - fcdcoffset;                                        // :138
    b += cut * (in - b * res - l);                                    // :139
    l += cut * b;                                                     // :142
    const float h = in - b * res - l;                                 // :143
    b += cut * h;                                                     // :144
    return SvfOut{l, b, h};
}

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
