// voices.hip.h -- every builtin module as a per-lane object: state + begin() (the per-paint prologue)
// + frame() (one sample) + end() (the per-paint epilogue).  The standalone module kernels
// (modules.hip, osc.hip) and the fused kernels the zangscript backend generates (zangscript/emit_hip.py,
// compiled by script.hip through hiprtc) are both built from these, so each of the reference's
// formulas exists once on the device.  frame() returns the painted value; where a module can paint
// nothing (silent oscillator, Gate off, Envelope idle ...) it returns `bool painted` and writes `val`.
#pragma once
#include "common.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "envelope.hip.h"

// ---- SineOsc (src/modules/SineOsc.zig) -----------------------------------------------------------
// SINMODE: 1 = musl's sinf for every argument, 0 = the same without its rare-path branch (the caller has checked the argument
// range), 2 = ZH_PAINT_TOLERANT's zsinf_tol (zmath.hip.h) of the same rounded argument
template <int SINMODE = 1>
__device__ __forceinline__ float sine_osc_sin(float t) {
    const float x = t * 3.14159265358979323846f * 2.0f;              // :4-6
    if constexpr (SINMODE == 2) return zsinf_tol(x);
    else return zsinf<SINMODE != 0>(x);
}
// |t| below this keeps (t * pi) * 2 below kZSinNoLargeBelow: sine_osc_sin<false> is exact
constexpr float kSineOscSmallT = 6.0e7f;

struct SineOscLane {
    float t;                                                          // state (:16-22)
    float t_step, inv_sr;
    __device__ __forceinline__ void begin(float sample_rate, float freq_const) {
        t_step = freq_const / sample_rate;                            // :44 (unused when freq is a buffer)
        inv_sr = 1.0f / sample_rate;                                  // :66
    }
    template <bool FB, int SINMODE = 1> __device__ __forceinline__ float frame(float freq_i, float phase_i) {
        const float val = sine_osc_sin<SINMODE>(t + phase_i);
        if (FB) t += freq_i * inv_sr; else t += t_step;
        return val;
    }
    __device__ __forceinline__ void end() { t = t - truncf(t); }      // :40
    // constant frequency and phase: true (wave-wide) when no voice's sine argument can reach zsinf's rare path in the next
    // `frames` frames (t moves by t_step a frame; NaN compares false) -- the chunk then runs frame<false, false>
    __device__ __forceinline__ bool small_args(float phase_c, float frames) const { return small_args_step(t_step, phase_c, frames); }
    // the same for a frequency buffer known to hold one value all span: `step` = that value * inv_sr
    __device__ __forceinline__ bool small_args_step(float step, float phase_c, float frames) const {
        const bool ok = __builtin_fabsf(t) + frames * __builtin_fabsf(step) + __builtin_fabsf(phase_c) < kSineOscSmallT;
        return __builtin_amdgcn_ballot_w64(!ok) == 0;
    }
};

// ---- PulseOsc (src/modules/PulseOsc.zig) ---------------------------------------------------------
struct PulseOscLane {
    uint32_t cnt;                                                     // state (:36-42)
    PulseK k;
    float srf, sr8;
    bool bad;
    // constant frequency (:77-121)
    __device__ __forceinline__ void begin_const(float sample_rate, float freq, float color) {
        srf = 4294967296.0f / sample_rate;                            // :87
        sr8 = sample_rate / 8.0f;                                     // :82
        bad = freq < 0 || freq > sr8;                                 // :82-84: paints nothing
        pulse_setup_color(k, color);
        pulse_setup_freq(k, srf, freq);
    }
    __device__ __forceinline__ bool frame_const(float &val) {
        if (bad) return false;
        val = pulse_sample(k, cnt);
        cnt += k.ifreq;
        return true;
    }
    // controlled frequency (:123-171): the per-voice constants are recomputed every sample
    __device__ __forceinline__ void begin_ctrl(float sample_rate, float color) {
        srf = 4294967296.0f / sample_rate;
        sr8 = sample_rate / 8.0f;
        pulse_setup_color(k, color);
    }
    __device__ __forceinline__ bool frame_ctrl(float s_freq, float &val) {
        if (s_freq < 0 || s_freq > sr8) return false;                 // :134-135
        pulse_setup_freq(k, srf, s_freq);
        val = pulse_sample(k, cnt);
        cnt += k.ifreq;
        return true;
    }
};

// ---- TriSawOsc (src/modules/TriSawOsc.zig) -------------------------------------------------------
struct TriSawK {          // :90-99
    uint32_t ifreq, brpt;
    float f, omf, rcpf, col, c1, c2;
};

// TriSawOsc.zig:103-114 as value selects with the mask logic of pulse_sample:
//   b0 == b1, no wrap  (3 / 0): c1|c2 * (p + p - f)                       c1 when b0 else c2
//   b0 == b1, wrap     (7 / 4): -rcpf * (gain + (c1|c2 * omf) * (p + p + omf))
//   b0 != b1           (2)    : rcpf * (c2*p^2 - c1*(p - f)^2)
//                      (5)    : -rcpf * (gain + c2*(p + omf)^2 - c1*p^2)
//   1 and 6 (`else => unreachable`) cannot occur for any u32 inputs (proof at pulse_sample).
// Every arm is the reference's own expression, so the selected value has the reference's bits;
// unselected arms may be inf/NaN (c1 = +inf when color == 0) and are discarded, never blended.
__device__ __forceinline__ void trisaw_setup(TriSawK &k, float srf, float freq, float color) {   // :90-99
    const float gain = 0.7f;
    k.ifreq = zf32_to_u32(srf * freq);
    k.brpt = zftou32(zclamp01(color));
    k.f = zutof23(k.ifreq);
    k.omf = 1.0f - k.f;
    k.rcpf = 1.0f / k.f;
    k.col = zutof23(k.brpt);
    k.c1 = gain / k.col;
    k.c2 = -gain / (1.0f - k.col);
}
__device__ __forceinline__ float trisaw_sample(const TriSawK &k, uint32_t cnt) {
    const float gain = 0.7f;
    const float p = zutof23(cnt) - k.col;
    const bool b0 = cnt < k.brpt;
    const bool b1 = (uint32_t)(cnt - k.ifreq) < k.brpt;
    const bool b2 = cnt < k.ifreq;
    const float cx = b0 ? k.c1 : k.c2;
    const float flat_nowrap = cx * (p + p - k.f);
    const float flat_wrap = -k.rcpf * (gain + cx * k.omf * (p + p + k.omf));
    const float ramp2 = k.rcpf * (k.c2 * (p * p) - k.c1 * ((p - k.f) * (p - k.f)));
    const float ramp5 = -k.rcpf * (gain + k.c2 * ((p + k.omf) * (p + k.omf)) - k.c1 * (p * p));
    const float flat = b2 ? flat_wrap : flat_nowrap;
    const float ramp = b2 ? ramp5 : ramp2;
    const float v = (b0 == b1) ? flat : ramp;
    return gain + v;
}
// color <= 0, the sawtooth (the reference's own recipes use it: examples/modules.zig:163, example_detuned.zig:71): brpt == 0, so
// `cnt < brpt` is false for every cnt, the state (:101-104) is 0b000 or 0b100 and only the two flat arms with c2 can be selected --
// trisaw_sample's own expressions for them, 15 instructions instead of 36.  Taken when EVERY voice of the wave is a sawtooth
// (trisaw_all_saw: a wave-uniform branch, no per-lane select).
__device__ __forceinline__ bool trisaw_all_saw(const TriSawK &k) { return __builtin_amdgcn_ballot_w64(k.brpt != 0u) == 0; }
__device__ __forceinline__ float trisaw_sample_saw(const TriSawK &k, uint32_t cnt) {
    const float gain = 0.7f;
    const float p = zutof23(cnt) - k.col;
    const bool b2 = cnt < k.ifreq;
    const float flat_nowrap = k.c2 * (p + p - k.f);
    const float flat_wrap = -k.rcpf * (gain + k.c2 * k.omf * (p + p + k.omf));
    return gain + (b2 ? flat_wrap : flat_nowrap);
}
// the naive saw / triangle of the controlled-frequency path (:120-156): uses the f32 phase, ignores cnt
// (every arm computed, then selected: `saw` and the triangle's three pieces differ from lane to lane, and per-lane
// branches cost more exec-mask instructions than the two multiplies an arm is)
__device__ __forceinline__ float trisaw_naive(float t, bool saw) {
    const float fr = t - floorf(t);
    const float sawv = fr * 2.0f - 1.0f;                              // :128
    const float up = fr * 4.0f;                                       // :137-143
    const float down = 1.0f - (fr - 0.25f) * 4.0f;
    const float up2 = (fr - 0.75f) * 4.0f - 1.0f;
    const float tri = zsel_hard(fr < 0.25f, up, zsel_hard(fr < 0.75f, down, up2));
    return 0.7f * zsel_hard(saw, sawv, tri);
}

struct TriSawOscLane {
    uint32_t cnt;                                                     // state (:36-44)
    float t;
    TriSawK k;
    float sample_rate;
    bool bad, saw;
    bool all_saw;                                                     // wave-uniform: every voice of the wave has brpt == 0
    __device__ __forceinline__ void begin_const(float sr, float freq, float color) {
        bad = freq < 0 || freq > sr / 8.0f;                           // :84-86
        trisaw_setup(k, 4294967296.0f / sr, freq, color);
        all_saw = trisaw_all_saw(k);
    }
    __device__ __forceinline__ bool frame_const(float &val) {
        if (bad) return false;
        val = all_saw ? trisaw_sample_saw(k, cnt) : trisaw_sample(k, cnt);
        cnt += k.ifreq;
        return true;
    }
    __device__ __forceinline__ void begin_ctrl(float sr, float color) {
        sample_rate = sr;
        saw = color < 0.25f || color > 0.75f;                         // :137-150
    }
    __device__ __forceinline__ float frame_ctrl(float s_freq) {
        const float val = trisaw_naive(t, saw);
        t += s_freq / sample_rate;
        return val;
    }
    __device__ __forceinline__ float frame_ctrl_q(float q) {         // frame_ctrl with q = s_freq / sample_rate computed beforehand
        const float val = trisaw_naive(t, saw);
        t += q;
        return val;
    }
    __device__ __forceinline__ void end_ctrl() { t = t - truncf(t); } // :155
};

// ---- Noise (src/modules/Noise.zig) ---------------------------------------------------------------
struct NoiseLane {
    ZXoshiro r;                                                       // state (:22-32)
    float b[7];                                                       // `var b = self.b` (:55): never written back (:68) => always 0
    __device__ __forceinline__ void begin() {
#pragma unroll
        for (int j = 0; j < 7; j++) b[j] = 0.0f;
    }
    template <bool PINK> __device__ __forceinline__ float frame() {
        const float white = zrandom_float32(r) * 2.0f - 1.0f;         // :51 / :58
        return PINK ? pink_step(b, white) : white;                    // :59-66
    }
};

// ---- Filter (src/modules/Filter.zig) -------------------------------------------------------------
struct FilterLane {
    float l, b;                                                       // state (:34-42)
    float cut, res, l_mul, b_mul, h_mul;
    bool bypass;
    __device__ __forceinline__ void begin(uint32_t type, float cutoff_const, float res_const) {
        bypass = type == ZH_FILTER_BYPASS;                            // :91-97: out += in, state untouched
        l_mul = (type == ZH_FILTER_LOW_PASS || type == ZH_FILTER_NOTCH || type == ZH_FILTER_ALL_PASS) ? 1.0f : 0.0f;   // :98-109
        b_mul = (type == ZH_FILTER_BAND_PASS || type == ZH_FILTER_ALL_PASS) ? 1.0f : 0.0f;
        h_mul = (type == ZH_FILTER_HIGH_PASS || type == ZH_FILTER_NOTCH || type == ZH_FILTER_ALL_PASS) ? 1.0f : 0.0f;
        cut = zclampf(cutoff_const, 0.0f, 1.0f);                      // :114
        res = 1.0f - zclampf(res_const, 0.0f, 1.0f);                  // :118
    }
    template <bool CB, bool RB> __device__ __forceinline__ float frame(float x, float cutoff_i, float res_i) {
        if (bypass) return x;
        if (CB) cut = zclampf(cutoff_i, 0.0f, 1.0f);                  // :126
        if (RB) res = 1.0f - zclampf(res_i, 0.0f, 1.0f);              // :128
        const SvfOut s = svf_step(l, b, x, cut, res);                 // :135-144
        return s.l * l_mul + s.b * b_mul + s.h * h_mul;               // :146
    }
    // frame() in three parts, for the role-wave form of a generated kernel (zscript_emit.hip), which deals them to different
    // waves: pre() is a function of the input sample alone, core() carries (l, b), post() is the mix.  post(x, core(pre(x))) is
    // frame(x): the same operations on the same values.
    __device__ __forceinline__ float pre(float x) const { return x + kSvfDcOffset; }                        // :135
    template <bool CB, bool RB> __device__ __forceinline__ SvfOut core(float in, float cutoff_i, float res_i) {
        if (bypass) return SvfOut{0.0f, 0.0f, 0.0f};
        if (CB) cut = zclampf(cutoff_i, 0.0f, 1.0f);
        if (RB) res = 1.0f - zclampf(res_i, 0.0f, 1.0f);
        return svf_core(l, b, in, cut, res);                          // :138-144
    }
    __device__ __forceinline__ float mix(float sl, float sb, float sh) const { return sl * l_mul + sb * b_mul + sh * h_mul; }   // :146
    __device__ __forceinline__ float post(float x, float sl, float sb, float sh) const { return bypass ? x : mix(sl, sb, sh); }  // :91-97

};

// ---- Decimator (src/modules/Decimator.zig) -------------------------------------------------------
struct DecimatorLane {
    float dval, dcount;                                               // state (:11-19): dval = 0, dcount = 1
    int mode;
    float ratio;
    __device__ __forceinline__ void begin(float sample_rate, float fake) {
        mode = fake >= sample_rate ? 0 : (fake > 0.0f ? 1 : 2);       // :34, :39
        ratio = fake / sample_rate;                                   // :40
    }
    // straight-line (per-lane branches on mode and on the trigger cost ~30 exec-mask instructions a frame): mode 0 is
    // addInto (:35), mode 2 (fake <= 0 or NaN) paints nothing, mode 1 is the sample-and-hold of :46-51
    __device__ __forceinline__ bool frame(float x, float &val) {
        const float dc = dcount + ratio;                              // :46
        const bool trig = dc >= 1.0f;                                 // :47
        const float dcn = trig ? dc - 1.0f : dc;                      // :49
        const bool run = mode == 1;
        dval = (run && trig) ? x : dval;                              // :48
        dcount = run ? dcn : dcount;
        val = mode == 0 ? x : dval;                                   // :35 / :51
        return mode != 2;
    }
    // the walk of frame() without the input: `last` = index of the latest frame that sampled (unchanged if none)
    __device__ __forceinline__ void step(uint32_t i, uint32_t &last) {
        const float dc = dcount + ratio;
        const bool trig = dc >= 1.0f;
        dcount = trig ? dc - 1.0f : dc;
        last = trig ? i : last;
    }
    // step() for 0 <= dcount <= 1 and 0 < ratio < 1 (every state the module itself produces): then dcount + ratio is below 2 and
    // `dc >= 1 ? dc - 1 : dc` is dc - floor(dc), one v_fract_f32 (exact: the subtraction of 1 from a value in [1, 2) is); the
    // frame index rides in a VGPR so that the whole step is five VALU instructions and no scalar ones
    __device__ __forceinline__ bool walk_is_plain() const { return mode != 1 || (dcount >= 0.0f && dcount <= 1.0f && ratio > 0.0f && ratio < 1.0f); }
    __device__ __forceinline__ void step_plain(uint32_t &idx, uint32_t &last) {
        const float dc = dcount + ratio;
        last = dc >= 1.0f ? idx : last;
        dcount = __builtin_amdgcn_fractf(dc);
        idx += 1;
    }
    __device__ __forceinline__ void end() {
        if (mode == 0) { dval = 0.0f; dcount = 1.0f; }                // :37-38
    }
};

// ---- Distortion (src/modules/Distortion.zig), stateless ------------------------------------------
struct DistortionLane {
    float gain1, offs, gain2;
    bool overdrive;
    __device__ __forceinline__ void begin(uint32_t type, float ingain, float outgain, float offset) {
        overdrive = type == ZH_DISTORTION_OVERDRIVE;
        gain1 = zpowf_pos(2.0f, ingain * 8.0f - 2.0f);                // :41
        offs = gain1 * offset;
        gain2 = overdrive ? outgain / zatanf(gain1) : outgain;        // :45 / :55
    }
    __device__ __forceinline__ float frame(float x) {
        const float a0 = x * gain1 + offs;
        if (overdrive) return gain2 * zatanf(a0);                     // :50-51
        return gain2 * (a0 < -1.0f ? -1.0f : (a0 > 1.0f ? 1.0f : a0));   // :60-62
    }
};

// ---- Cycle (src/modules/Cycle.zig) ---------------------------------------------------------------
struct CycleLane {
    float t;                                                          // state
    float step, isr;
    __device__ __forceinline__ void begin(float sample_rate, float speed_const) {
        step = speed_const / sample_rate;                             // :37
        isr = 1.0f / sample_rate;                                     // :48
    }
    template <bool SB> __device__ __forceinline__ float frame(float speed_i) {
        const float val = t;                                          // :41
        if (SB) t += speed_i * isr; else t += step;                   // :42 / :53
        t -= truncf(t);                                               // :43
        return val;
    }
};

// ---- Portamento (src/modules/Portamento.zig over painter.zig) ------------------------------------
struct PortamentoLane {
    float t, last, st;                                                // state: painter {t, last_value, start}
    float goal, t_step;
    uint32_t tag;
    bool flat;
    __device__ __forceinline__ void begin(float sample_rate, uint32_t curve_tag, float duration, float goal_, bool note_on,
                                          bool prev_note_on, bool note_id_changed) {
        goal = goal_;
        tag = (note_on && prev_note_on) ? curve_tag : (uint32_t)ZH_CURVE_INSTANTANEOUS;   // :33-36
        if (note_on && note_id_changed) { st = last; t = 0.0f; }                          // :38-40 newCurve
        // paintToward's entry (painter.zig:69-80), then either the glide or paintFlat(goal) (:43-47)
        flat = false;
        if (t >= 1.0f) flat = true;
        else if (tag == ZH_CURVE_INSTANTANEOUS) { t = 1.0f; last = goal; flat = true; }
        t_step = 1.0f / (duration * sample_rate);                                          // painter.zig:97
    }
    // one paintToward step (painter.zig:103-116) or paintFlat's goal, as selects: per-lane branches on `flat` and on the
    // curve tag cost more exec-mask instructions than the arithmetic they skip
    __device__ __forceinline__ float curve_at(float tn) const {
        const float it = 1.0f - tn;
        const float sq = it * it;
        return zsel_hard(tag == ZH_CURVE_SQUARED, 1.0f - sq, zsel_hard(tag == ZH_CURVE_CUBED, 1.0f - sq * it, tn));
    }
    __device__ __forceinline__ float frame() {
        float tn = t + t_step;
        const bool fin = tn >= 1.0f;
        tn = fin ? 1.0f : tn;
        const float lv = st + curve_at(tn) * (goal - st);
        const float val = flat ? goal : lv;
        t = flat ? t : tn;
        last = flat ? last : lv;
        flat = flat || fin;
        return val;
    }
    // frame_loop_gen (seq.hip.h): once every voice of the wave has arrived (paintFlat, Portamento.zig:43-47) a frame is the
    // goal itself and nothing changes
    __device__ __forceinline__ bool all_flat() const { return __builtin_amdgcn_ballot_w64(!flat) == 0; }
    // N frames with the values discarded (a frame-range kernel's replay), for a wave in which no gliding voice can arrive
    // within them (the margin argument of EnvLaneT::quiet): the clock steps N times, the value it leaves is evaluated once
    __device__ __forceinline__ bool quiet(int n) const {
        return __builtin_amdgcn_ballot_w64(!flat && !(t + (float)(n + 1) * t_step < 0.999f)) == 0;
    }
    template <int N> __device__ __forceinline__ void skip_quiet() {
        float tt = t;
#pragma unroll
        for (int k = 0; k < N; k++) tt = tt + t_step;
        const float lv = st + curve_at(tt) * (goal - st);
        t = flat ? t : tt;
        last = flat ? last : lv;
    }
};

// ---- Curve (src/modules/Curve.zig) ---------------------------------------------------------------
struct CurveSpanNode { int32_t frame; float value; };                  // :11-14

// The reference builds the paint's (<= 32) span nodes (getCurveSpanNodes, :130-184) and then asks for one curve span after
// the other (getNextCurveSpan, :188-255: at frame 0 and wherever a span ends -- its `while (start < out.len)` loop).  Both
// walk the node list front to back and a span depends on two neighbouring nodes only, so begin() streams: a node is final
// once its successor lies on a different frame (:165-167 replaces a node that shares its frame), and a final pair (node,
// successor) yields that node's spans -- a gap up to its frame, if any, then its segment.  The first span becomes the
// running span, in registers; the ones after it go to a per-lane table (CurveTable, scratch memory) that the frame walk
// reads at span changes.  All lanes stay on the same frame.
constexpr uint32_t kCurveSpans = 36;                                   // <= 32 nodes: a gap + 31 segments + the last node's span + the tail
struct CurveTable {                                                   // its own object: one dynamic index keeps a whole object in scratch
    uint32_t s_end[kCurveSpans];                                      // bit 31: the span has values
    float s_acc[kCurveSpans], s_step[kCurveSpans], s_sv[kCurveSpans], s_vd[kCurveSpans];
};
struct CurveLane {
    float t;                                                          // state (:36-49)
    uint32_t cur, next;
    int32_t off;
    uint32_t out_len, function;
    uint32_t n_spans, k;                                              // spans of this paint, the running one's index
    // the running span
    uint32_t span_end;
    bool has_values;
    float acc, step, start_value, value_delta;
    // begin()'s stream
    uint32_t dest_start, r0, run_start;                               // r0: the first relative frame the walk will ask for; run_start: where the running span began
    bool nodes_done;

    __device__ __forceinline__ void add_span(CurveTable &tb, uint32_t end_, bool values, float acc_, float step_, float sv_, float vd_) {
        const uint32_t from = dest_start;
        dest_start = end_;
        if (end_ <= r0) return;                                       // over before the first frame wanted (a frame range's begin)
        if (n_spans == 0) { run_start = from; span_end = end_; has_values = values; acc = acc_; step = step_; start_value = sv_; value_delta = vd_; }
        if (n_spans < kCurveSpans) {
            tb.s_end[n_spans] = end_ | (values ? 0x80000000u : 0u);
            tb.s_acc[n_spans] = acc_; tb.s_step[n_spans] = step_; tb.s_sv[n_spans] = sv_; tb.s_vd[n_spans] = vd_;
            n_spans++;
        }
    }
    // getNextCurveSpan's loop body for node `a` (successor `b` when has_b), from the current dest_start on (:196-250)
    __device__ __forceinline__ void node_spans(CurveTable &tb, CurveSpanNode a, bool has_b, CurveSpanNode b) {
        const int32_t dest_end = (int32_t)out_len;
        if (nodes_done || (int32_t)dest_start >= dest_end) return;
        if (a.frame >= dest_end) { nodes_done = true; return; }                                        // `break`: the rest is the tail
        const int32_t end_pos = has_b ? min(dest_end, b.frame) : dest_end;
        if (end_pos <= (int32_t)dest_start) return;                                                    // `continue`
        if (a.frame > (int32_t)dest_start) add_span(tb, (uint32_t)a.frame, false, 0.0f, 0.0f, 0.0f, 0.0f);   // the gap before the node
        float acc_ = 0.0f, step_ = 0.0f, sv_ = 0.0f, vd_ = 0.0f;
        if (has_b) {                                                                                    // :84-107
            const float start_x = (float)((int32_t)dest_start - a.frame) / (float)(b.frame - a.frame);   // :95
            sv_ = a.value;
            vd_ = b.value - a.value;
            const float x_step = 1.0f / (float)(b.frame - a.frame);                                     // :100
            if (function == ZH_CURVE_FN_LINEAR) { acc_ = sv_ + start_x * vd_; step_ = x_step * vd_; }
            else { acc_ = start_x; step_ = x_step; }
        }
        add_span(tb, (uint32_t)end_pos, has_b, acc_, step_, sv_, vd_);
    }

    // r0_ = relative frame the walk starts at (a frame range; 0 = the whole paint): frame() is then called for r0_, r0_ + 1, ...
    __device__ __forceinline__ void begin(CurveTable &tb, float sample_rate, uint32_t function_, const zh_curve_node *__restrict__ curve,
                                          uint32_t n_curve, uint32_t out_len_, bool note_id_changed, uint32_t r0_ = 0) {
        function = function_;
        out_len = out_len_;
        r0 = r0_; run_start = 0;
        if (note_id_changed) { cur = 0; off = 0; next = 0; t = 0.0f; }   // :66-71
        n_spans = 0; dest_start = 0; nodes_done = false;
        span_end = out_len; has_values = false;
        acc = step = start_value = value_delta = 0.0f;
        const float buf_time = (float)out_len / sample_rate;          // getCurveSpanNodes
        const float end_t = t + buf_time;
        // a = the latest final node, b = the latest node (final once a node on another frame follows)
        CurveSpanNode a{0, 0.0f}, b{0, 0.0f};
        bool has_a = false, has_b = false;
        uint32_t count = 0;
        if (cur < next) { b.frame = off; b.value = curve[cur].value; has_b = true; count = 1; }   // :142-148
        bool one_past = false;
        for (uint32_t i = next; i < n_curve; i++) {
            const zh_curve_node nd = curve[i];
            if (nd.t >= end_t) { if (!one_past) one_past = true; else break; }                           // :153-160
            const float f = (nd.t - t) / buf_time;
            const int32_t rel = zf32_to_i32(f * (float)out_len);
            const CurveSpanNode nn{rel, nd.value};
            if (has_b && b.frame == rel) b = nn;                                                        // :165-167: replaces the node on its frame
            else if (count < 32) {
                if (has_b) { if (has_a) node_spans(tb, a, true, b); a = b; has_a = true; }
                b = nn; has_b = true; count++;
            }
            if (!one_past) { cur = next; off = 0; next += 1; }                                          // :173-177
        }
        if (has_b) {
            if (has_a) node_spans(tb, a, true, b);
            node_spans(tb, b, false, b);
        }
        if (dest_start < out_len) add_span(tb, out_len, false, 0.0f, 0.0f, 0.0f, 0.0f);                 // the tail (no node left: :252-254)
        t += buf_time;                                                 // :180
        off -= (int32_t)out_len;                                       // :181
        k = 0;
        // the running span's frames before r0: its accumulator stepped as frame() would have (a gap's is never read)
        uint32_t i = run_start;
        for (; i + 8 <= r0; i += 8) {
#pragma unroll
            for (int q = 0; q < 8; q++) acc += step;
        }
        for (; i < r0; i++) acc += step;
    }
    __device__ __forceinline__ void advance_span(const CurveTable &tb) {
        k = k + 1 < kCurveSpans ? k + 1 : k;
        span_end = tb.s_end[k] & 0x7FFFFFFFu; has_values = (tb.s_end[k] >> 31) != 0;
        acc = tb.s_acc[k]; step = tb.s_step[k]; start_value = tb.s_sv[k]; value_delta = tb.s_vd[k];
    }
    // r = frame index relative to the paint span's start.  Straight-line apart from the (rare) span change: `function` is
    // the same for every voice, and a span without values (a gap) never reads acc, so the accumulator runs unconditionally
    __device__ __forceinline__ bool frame(const CurveTable &tb, uint32_t r, float &val) {
        if (r == span_end) advance_span(tb);
        return frame_in_span(val);
    }
    // frame() where the running span is known not to end (seq.hip.h frame_loop_gen): no lane's span ends in [r, r + n)
    __device__ __forceinline__ bool quiet(uint32_t r, uint32_t n) const { return __builtin_amdgcn_ballot_w64(span_end - r < n) == 0; }
    __device__ __forceinline__ bool frame_in_span(float &val) {
        if (function == ZH_CURVE_FN_LINEAR) val = acc;                                                  // :109-112
        else val = start_value + acc * acc * (3.0f - 2.0f * acc) * value_delta;                         // :117-121
        acc += step;
        return has_values;
    }
};

// ---- a canned track: NoteTracker (src/zang/notes.zig:138-207) feeding Trigger (src/zang/trigger.zig) ----
// What zangscript's `from <track>, <speed> begin ... end` runs per paint call (codegen_zig.zig:359-389):
// the notes that fall into this buffer become impulses, Trigger cuts the span into sub-spans that belong
// to one note each.  The lane builds that list (<= 33 sub-spans) in begin(); frames outside every
// sub-span (before the first note) are not painted.  A note is identified by its index in the track
// (generated note ids are index + 1, codegen_zig.zig:497), which is all Trigger needs to remember.
struct TrackLane {
    uint32_t next;                                                    // NoteTracker.next_song_event
    float t;                                                          // NoteTracker.t
    uint32_t cur;                                                     // Trigger.note: 0 = null, else note index + 1
    uint32_t n;                                                       // sub-spans of this paint call
    uint32_t s_start[33], s_end[33], s_note[33];
    bool s_new[33];

    __device__ __forceinline__ void begin(const float *__restrict__ times, uint32_t n_notes, float sample_rate,
                                          uint32_t out_len, bool reset) {
        if (reset) { next = 0; t = 0.0f; cur = 0; }                   // tracker.reset(); trigger.reset()
        // NoteTracker.consume (notes.zig:161-205)
        uint32_t imp_frame[32], imp_note[32];
        uint32_t count = 0;
        const float buf_time = (float)out_len / sample_rate;
        const float end_t = t + buf_time;
        while (next < n_notes) {
            const float note_t = times[next];
            if (!(note_t < end_t)) break;
            const float f = (note_t - t) / buf_time;                  // 0 to 1
            const uint32_t rel = min(zf32_to_u32(f * (float)out_len), out_len - 1);
            if (count < 32) { imp_frame[count] = rel; imp_note[count] = next; count++; }   // a 33rd impulse is dropped
            next += 1;
        }
        t = end_t;
        // Trigger.next until exhausted (trigger.zig:80-195)
        n = 0;
        uint32_t start = 0, ii = 0;
        while (start < out_len) {
            uint32_t span_end = out_len, note = 0;
            bool have = false;
            if (cur != 0) {                                           // carryOver (:108-142)
                if (ii < count) {
                    if (imp_frame[ii] > start) { have = true; span_end = min(out_len, imp_frame[ii]); note = cur; }
                } else {
                    have = true; note = cur;
                }
            }
            if (!have) {                                              // getNextNoteSpan (:144-195)
                const uint32_t ii0 = ii;
                for (uint32_t i = ii0; i < count; i++) {
                    const uint32_t fr = imp_frame[i];
                    if (fr >= out_len) break;
                    if (fr > start) { span_end = fr; break; }         // gap before the note begins
                    ii += 1;
                    const uint32_t end_c = (i + 1 < count) ? min(out_len, imp_frame[i + 1]) : out_len;
                    if (end_c <= start) continue;                     // the next impulse starts at the same time
                    span_end = end_c;
                    note = imp_note[i] + 1;
                    break;
                }
            }
            if (note != 0) {
                if (n < 33) {
                    s_start[n] = start; s_end[n] = span_end; s_note[n] = note - 1;
                    s_new[n] = cur == 0 || note != cur;               // note_id_changed (:93-96)
                    n++;
                }
                cur = note;
            }
            start = span_end;
        }
    }
};
