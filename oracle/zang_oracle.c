/* zang_oracle.c -- ORACLE: scalar, single-voice CPU restatement of zang's module
 * paint() hot path.  TEST INFRASTRUCTURE ONLY: it is imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product
 * (zang_amd/ fails loudly when libzang_hip.so is missing; it has no CPU fallback).
 *
 * PARITY UNPINNED for the paint arithmetic: the reference (Zig, /root/reference) cannot
 * be compiled in this image (no Zig toolchain) and its own tests hold no vector for any
 * paint()/basics/Painter result.  What pins this file:
 *   - tests/golden/known_answers.json : hand/numpy-derived known answers K1-K4 of
 *     SURVEY.md 8c (xoshiro256++ seed-0 sequence, PulseOsc, Filter impulse response,
 *     Decimator, Painter steps),
 *   - tests/test_oracle_numpy.py      : an independent numpy-float32 re-derivation of
 *     each module from the cited reference lines,
 *   - tests/test_oracle_math.py       : libm restatements vs correctly-rounded values.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).  Same struct fields, same operation order, one scalar loop with the
 * same loop-carried dependencies.  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fno-fast-math; Zig emits no fused multiply-adds here).
 *
 * Where the reference is undefined (assert / unreachable / checked-cast panic in safe
 * builds) this file DEFINES the behaviour and says so; the HIP path implements the same
 * definition.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include "zmath_ref.h"
#include "zang_oracle.h"

/* ------------------------------------------------------------------ basics.zig */
/* src/zang/basics.zig:12-22 */
void zo_zero(size_t start, size_t end, float *dest) { for (size_t i = start; i < end; i++) dest[i] = 0.0f; }
void zo_set(size_t start, size_t end, float *dest, float a) { for (size_t i = start; i < end; i++) dest[i] = a; }
void zo_copy(size_t start, size_t end, float *dest, const float *src) { for (size_t i = start; i < end; i++) dest[i] = src[i]; }
/* basics.zig:24-29  dest += a + b  (the sum a+b is formed first) */
void zo_add(size_t start, size_t end, float *dest, const float *a, const float *b) {
    for (size_t i = start; i < end; i++) dest[i] += a[i] + b[i];
}
/* basics.zig:31-36 */
void zo_add_into(size_t start, size_t end, float *dest, const float *src) {
    for (size_t i = start; i < end; i++) dest[i] += src[i];
}
/* basics.zig:38-43 */
void zo_add_scalar(size_t start, size_t end, float *dest, const float *a, float b) {
    for (size_t i = start; i < end; i++) dest[i] += a[i] + b;
}
/* basics.zig:45-50 */
void zo_add_scalar_into(size_t start, size_t end, float *dest, float a) {
    for (size_t i = start; i < end; i++) dest[i] += a;
}
/* basics.zig:52-57  dest += a*b, not fused */
void zo_multiply(size_t start, size_t end, float *dest, const float *a, const float *b) {
    for (size_t i = start; i < end; i++) dest[i] += a[i] * b[i];
}
/* basics.zig:59-64 */
void zo_multiply_with(size_t start, size_t end, float *dest, const float *a) {
    for (size_t i = start; i < end; i++) dest[i] *= a[i];
}
/* basics.zig:66-71 */
void zo_multiply_scalar(size_t start, size_t end, float *dest, const float *a, float b) {
    for (size_t i = start; i < end; i++) dest[i] += a[i] * b;
}
/* basics.zig:73-78 */
void zo_multiply_with_scalar(size_t start, size_t end, float *dest, float a) {
    for (size_t i = start; i < end; i++) dest[i] *= a;
}

/* ------------------------------------------------------------------ painter.zig */
typedef struct { float *buf; size_t len; size_t i; float sample_rate; } zo_paint_state; /* painter.zig:11-23 */

/* painter.zig:47-50 */
static void painter_new_curve(zo_painter *p) { p->start = p->last_value; p->t = 0.0f; }

/* painter.zig:53-58 */
static void painter_paint_flat(zo_painter *p, zo_paint_state *st, float value) {
    (void)p;
    zo_add_scalar_into(st->i, st->len, st->buf, value);
    st->i = st->len;
}

/* painter.zig:63-120 */
static int painter_paint_toward(zo_painter *p, zo_paint_state *st, zo_curve curve, float goal) {
    if (p->t >= 1.0f) return 1;                               /* :69-71 */
    if (curve.tag == ZO_CURVE_INSTANTANEOUS) {                /* :76-80 paints nothing */
        p->t = 1.0f;
        p->last_value = goal;
        return 1;
    }
    size_t i = st->i;
    const float t_step = 1.0f / (curve.duration * st->sample_rate); /* :97 */
    int finished = 0;
    while (!finished && i < st->len) {                        /* :102 */
        p->t += t_step;
        if (p->t >= 1.0f) { p->t = 1.0f; finished = 1; }
        const float it = 1.0f - p->t;
        float tp;
        switch (curve.tag) {                                  /* :109-113 */
        case ZO_CURVE_LINEAR: tp = p->t; break;
        case ZO_CURVE_SQUARED: tp = 1.0f - it * it; break;
        default: tp = 1.0f - it * it * it; break;             /* (it*it)*it */
        }
        p->last_value = p->start + tp * (goal - p->start);   /* :114 */
        st->buf[i] += p->last_value;
        i++;
    }
    st->i = i;
    return finished;
}

/* ------------------------------------------------------------------ SineOsc.zig */
void zo_sineosc_init(zo_sineosc *s) { s->t = 0.0f; }                 /* SineOsc.zig:18-22 */

static inline float sine_osc_sin(float t) {                          /* SineOsc.zig:4-6: (t*pi)*2 */
    return zr_sinf(t * 3.14159265358979323846f * 2.0f);
}

/* SineOsc.zig:24-87 */
void zo_sineosc_paint(zo_sineosc *self, size_t start, size_t end, float *out0,
                      float sample_rate, zo_cob freq, zo_cob phase) {
    float *output = out0 + start;
    const size_t len = end - start;
    float t = self->t;
    if (freq.tag == ZO_COB_CONSTANT) {
        const float t_step = freq.constant / sample_rate;            /* :44 */
        if (phase.tag == ZO_COB_CONSTANT) {
            for (size_t i = 0; i < len; i++) { output[i] += sine_osc_sin(t + phase.constant); t += t_step; }
        } else {
            const float *ps = phase.buffer + start;
            for (size_t i = 0; i < len; i++) { output[i] += sine_osc_sin(t + ps[i]); t += t_step; }
        }
    } else {
        const float *fs = freq.buffer + start;
        const float inv_sr = 1.0f / sample_rate;                     /* :66 */
        if (phase.tag == ZO_COB_CONSTANT) {
            for (size_t i = 0; i < len; i++) { output[i] += sine_osc_sin(t + phase.constant); t += fs[i] * inv_sr; }
        } else {
            const float *ps = phase.buffer + start;
            for (size_t i = 0; i < len; i++) { output[i] += sine_osc_sin(t + ps[i]); t += fs[i] * inv_sr; }
        }
    }
    self->t = t - truncf(t);                                         /* :40 (defer) */
}

/* ------------------------------------------------------------------ PulseOsc.zig / TriSawOsc.zig helpers */
static const float FC32BIT = 4294967296.0f;                          /* PulseOsc.zig:12 */
static inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }   /* :14-16 */
static inline float utof23(uint32_t x) { return zr_u2f((x >> 9) | 0x3f800000u) - 1.0f; } /* :19-21 */
static inline uint32_t ftou32(float v) { return zr_f32_to_u32(v * FC32BIT * 0.99995f); }   /* :24-26 */

void zo_pulseosc_init(zo_pulseosc *s) { s->cnt = 0; }                /* PulseOsc.zig:38-42 */

/* one output value of the 6-way switch, PulseOsc.zig:102-110 / 144-152.
 * `else => unreachable` is DEFINED here as contributing +0.0. */
static inline float pulse_value(uint32_t transition, float gain, float gdf, float col, float p,
                                float cc121, float cc212) {
    switch (transition) {
    case 3: return gain;
    case 0: return -gain;
    case 2: return gdf * 2.0f * (col - p) + gain;
    case 5: return gdf * 2.0f * p - gain;
    case 7: return cc121;
    case 4: return cc212;
    default: return 0.0f;
    }
}

/* PulseOsc.zig:75-114 */
static void pulseosc_paint_const(zo_pulseosc *self, float *output, size_t len, float sample_rate,
                                 float freq, float color) {
    if (freq < 0 || freq > sample_rate / 8.0f) return;               /* :82-84 */
    uint32_t cnt = self->cnt;
    const float SRfcobasefrq = FC32BIT / sample_rate;
    const uint32_t ifreq = zr_f32_to_u32(SRfcobasefrq * freq);
    const uint32_t brpt = ftou32(clamp01(color));
    const float gain = 0.7f;
    const float gdf = gain / utof23(ifreq);
    const float col = utof23(brpt);
    const float cc121 = gdf * 2.0f * (col - 1.0f) + gain;
    const float cc212 = gdf * 2.0f * col - gain;
    uint32_t state = ((uint32_t)(cnt - ifreq) < brpt) ? 3u : 0u;     /* :96 */
    for (size_t i = 0; i < len; i++) {
        const float p = utof23(cnt);
        state = ((state << 1) | (cnt < brpt ? 1u : 0u)) & 3u;
        const uint32_t transition = state | ((cnt < ifreq ? 1u : 0u) << 2);
        output[i] += pulse_value(transition, gain, gdf, col, p, cc121, cc212);
        cnt += ifreq;
    }
    self->cnt = cnt;
}

/* PulseOsc.zig:116-157 */
static void pulseosc_paint_ctrl(zo_pulseosc *self, float *output, size_t len, float sample_rate,
                                const float *freq, float color) {
    uint32_t cnt = self->cnt;
    const float SRfcobasefrq = FC32BIT / sample_rate;
    const uint32_t brpt = ftou32(clamp01(color));
    const float gain = 0.7f;
    const float col = utof23(brpt);
    for (size_t i = 0; i < len; i++) {
        const float s_freq = freq[i];
        if (s_freq < 0 || s_freq > sample_rate / 8.0f) continue;    /* :134-135: cnt not advanced */
        const uint32_t ifreq = zr_f32_to_u32(SRfcobasefrq * s_freq);
        const float gdf = gain / utof23(ifreq);
        const float cc121 = gdf * 2.0f * (col - 1.0f) + gain;
        const float cc212 = gdf * 2.0f * col - gain;
        const float p = utof23(cnt);
        const uint32_t c = ((uint32_t)(cnt - ifreq) < brpt) ? 1u : 0u;
        const uint32_t state = (cnt < brpt ? 1u : 0u) | (c << 1);
        const uint32_t transition = state | ((cnt < ifreq ? 1u : 0u) << 2);
        output[i] += pulse_value(transition, gain, gdf, col, p, cc121, cc212);
        cnt += ifreq;
    }
    self->cnt = cnt;
}

/* PulseOsc.zig:44-73 */
void zo_pulseosc_paint(zo_pulseosc *self, size_t start, size_t end, float *out0,
                       float sample_rate, zo_cob freq, float color) {
    if (freq.tag == ZO_COB_CONSTANT)
        pulseosc_paint_const(self, out0 + start, end - start, sample_rate, freq.constant, color);
    else
        pulseosc_paint_ctrl(self, out0 + start, end - start, sample_rate, freq.buffer + start, color);
}

/* ------------------------------------------------------------------ TriSawOsc.zig */
void zo_trisawosc_init(zo_trisawosc *s) { s->cnt = 0; s->t = 0.0f; } /* TriSawOsc.zig:39-44 */

static inline float sqr(float v) { return v * v; }                   /* :10-12 */

/* TriSawOsc.zig:77-118 */
static void trisaw_paint_const(zo_trisawosc *self, float *output, size_t len, float sample_rate,
                               float freq, float color) {
    if (freq < 0 || freq > sample_rate / 8.0f) return;               /* :84-86 */
    uint32_t cnt = self->cnt;
    const float SRfcobasefrq = FC32BIT / sample_rate;
    const uint32_t ifreq = zr_f32_to_u32(SRfcobasefrq * freq);
    const uint32_t brpt = ftou32(clamp01(color));
    const float gain = 0.7f;
    const float f = utof23(ifreq);
    const float omf = 1.0f - f;
    const float rcpf = 1.0f / f;
    const float col = utof23(brpt);
    const float c1 = gain / col;
    const float c2 = -gain / (1.0f - col);
    uint32_t state = ((uint32_t)(cnt - ifreq) < brpt) ? 3u : 0u;
    for (size_t i = 0; i < len; i++) {
        const float p = utof23(cnt) - col;
        state = ((state << 1) | (cnt < brpt ? 1u : 0u)) & 3u;
        const uint32_t s = state | ((cnt < ifreq ? 1u : 0u) << 2);
        float v;
        switch (s) {                                                 /* :106-114 */
        case 3: v = c1 * (p + p - f); break;
        case 0: v = c2 * (p + p - f); break;
        case 2: v = rcpf * (c2 * sqr(p) - c1 * sqr(p - f)); break;
        case 5: v = -rcpf * (gain + c2 * sqr(p + omf) - c1 * sqr(p)); break;
        case 7: v = -rcpf * (gain + c1 * omf * (p + p + omf)); break;
        case 4: v = -rcpf * (gain + c2 * omf * (p + p + omf)); break;
        default: v = 0.0f; break;                                    /* unreachable: defined as 0 */
        }
        output[i] += gain + v;
        cnt += ifreq;
    }
    self->cnt = cnt;
}

/* TriSawOsc.zig:120-156 (naive f32-phase saw / triangle; ignores cnt) */
static void trisaw_paint_ctrl(zo_trisawosc *self, float *output, size_t len, float sample_rate,
                              const float *freq, float color) {
    float t = self->t;
    const float gain = 0.7f;
    for (size_t i = 0; i < len; i++) {
        float frac;
        if (color < 0.25f || color > 0.75f) {
            frac = (t - floorf(t)) * 2.0f - 1.0f;
        } else {
            frac = t - floorf(t);
            if (frac < 0.25f) frac = frac * 4.0f;
            else if (frac < 0.75f) frac = 1.0f - (frac - 0.25f) * 4.0f;
            else frac = (frac - 0.75f) * 4.0f - 1.0f;
        }
        output[i] += gain * frac;
        t += freq[i] / sample_rate;
    }
    self->t = t - truncf(t);                                         /* :155 */
}

/* TriSawOsc.zig:46-75 */
void zo_trisawosc_paint(zo_trisawosc *self, size_t start, size_t end, float *out0,
                        float sample_rate, zo_cob freq, float color) {
    if (freq.tag == ZO_COB_CONSTANT)
        trisaw_paint_const(self, out0 + start, end - start, sample_rate, freq.constant, color);
    else
        trisaw_paint_ctrl(self, out0 + start, end - start, sample_rate, freq.buffer + start, color);
}

/* ------------------------------------------------------------------ Noise.zig */
/* Noise.zig:25-32.  The reference seeds from a process-global counter in init() order
 * (:9,:26); here the caller passes that ordinal explicitly (global voice index). */
void zo_noise_init(zo_noise *n, uint64_t seed) {
    zr_xoshiro r;
    zr_xoshiro_init(&r, seed);
    memcpy(n->r, r.s, sizeof n->r);
    for (int i = 0; i < 7; i++) n->b[i] = 0.0f;
}

/* Noise.zig:34-72 */
void zo_noise_paint(zo_noise *self, size_t start, size_t end, float *out, uint32_t color) {
    zr_xoshiro r;
    memcpy(r.s, self->r, sizeof r.s);
    if (color == ZO_NOISE_WHITE) {
        for (size_t i = start; i < end; i++) out[i] += zr_random_float32(&r) * 2.0f - 1.0f;
    } else {
        float b[7];
        memcpy(b, self->b, sizeof b);
        for (size_t i = start; i < end; i++) {
            const float white = zr_random_float32(&r) * 2.0f - 1.0f;
            b[0] = 0.99886f * b[0] + white * 0.0555179f;
            b[1] = 0.99332f * b[1] + white * 0.0750759f;
            b[2] = 0.96900f * b[2] + white * 0.1538520f;
            b[3] = 0.86650f * b[3] + white * 0.3104856f;
            b[4] = 0.55000f * b[4] + white * 0.5329522f;
            b[5] = -0.7616f * b[5] - white * 0.0168980f;
            out[i] += b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + white * 0.5362f;
            b[6] = white * 0.115926f;
        }
        /* Noise.zig:68 reads `b = self.b;` -- the filter taps are NOT written back, so
         * pink's state restarts from self.b (zeros after init) on every paint.  Reproduced. */
    }
    memcpy(self->r, r.s, sizeof r.s);                                /* :71 */
}

/* ------------------------------------------------------------------ Envelope.zig */
void zo_envelope_init(zo_envelope *e) {                              /* Envelope.zig:26-31 */
    e->state = ZO_ENV_IDLE;
    e->painter.t = 0.0f; e->painter.last_value = 0.0f; e->painter.start = 0.0f;
}

static void env_change_state(zo_envelope *e, uint32_t s) { e->state = s; painter_new_curve(&e->painter); } /* :33-36 */

/* Envelope.zig:38-73.  The asserts at :45 (state != release when note_on without a new
 * note) and :72 are checks only; with them compiled out a voice in `release` that gets
 * note_on without note_id_changed matches none of the stage tests and paints nothing.
 * That literal fall-through is the behaviour DEFINED here. */
static void env_paint_on(zo_envelope *e, float *buf, size_t len, const zo_envelope_params *p, int new_note) {
    zo_paint_state ps = { buf, len, 0, p->sample_rate };
    if (new_note) env_change_state(e, ZO_ENV_ATTACK);
    if (e->state == ZO_ENV_IDLE) env_change_state(e, ZO_ENV_ATTACK);
    if (e->state == ZO_ENV_ATTACK) {
        if (painter_paint_toward(&e->painter, &ps, p->attack, 1.0f)) {
            if (p->sustain_volume < 1.0f) env_change_state(e, ZO_ENV_DECAY);
            else env_change_state(e, ZO_ENV_SUSTAIN);
        }
    }
    if (e->state == ZO_ENV_DECAY) {
        if (painter_paint_toward(&e->painter, &ps, p->decay, p->sustain_volume))
            env_change_state(e, ZO_ENV_SUSTAIN);
    }
    if (e->state == ZO_ENV_SUSTAIN) painter_paint_flat(&e->painter, &ps, p->sustain_volume);
}

/* Envelope.zig:77-90 */
static void env_paint_off(zo_envelope *e, float *buf, size_t len, const zo_envelope_params *p) {
    if (e->state == ZO_ENV_IDLE) return;
    if (e->state != ZO_ENV_RELEASE) env_change_state(e, ZO_ENV_RELEASE);
    zo_paint_state ps = { buf, len, 0, p->sample_rate };
    if (painter_paint_toward(&e->painter, &ps, p->release, 0.0f)) env_change_state(e, ZO_ENV_IDLE);
}

/* Envelope.zig:92-109 */
void zo_envelope_paint(zo_envelope *self, size_t start, size_t end, float *out0,
                       int note_id_changed, const zo_envelope_params *params) {
    float *output = out0 + start;
    if (params->note_on) env_paint_on(self, output, end - start, params, note_id_changed);
    else env_paint_off(self, output, end - start, params);
}

/* ------------------------------------------------------------------ Gate.zig:15-30 */
void zo_gate_paint(size_t start, size_t end, float *out0, int note_on) {
    if (note_on) zo_add_scalar_into(start, end, out0, 1.0f);
}

/* ------------------------------------------------------------------ Filter.zig */
void zo_filter_init(zo_filter *f) { f->l = 0.0f; f->b = 0.0f; }      /* Filter.zig:37-42 */

/* std.math.clamp(v, lo, hi) == @max(lo, @min(v, hi)); @min/@max return the non-NaN
 * operand.  Written as explicit compare/selects (first operand wins ties, so
 * clamp(-0.0, 0, 1) == +0.0) so the device code can state the identical selection. */
static inline float zminf(float a, float b) { return (a <= b || b != b) ? a : b; }
static inline float zmaxf(float a, float b) { return (a >= b || b != b) ? a : b; }
static inline float zclampf(float v, float lo, float hi) { return zmaxf(lo, zminf(v, hi)); }

/* Filter.zig:20-23 */
float zo_filter_cutoff_from_frequency(float frequency, float sample_rate) {
    const float v = 2.0f * (1.0f - zr_cosf(3.14159265358979323846f * frequency / sample_rate));
    return sqrtf(zclampf(v, 0.0f, 1.0f));
}

/* Filter.zig:44-151 */
void zo_filter_paint(zo_filter *self, size_t start, size_t end, float *out0, const float *input0,
                     uint32_t type, zo_cob cutoff, zo_cob res_p) {
    static const float fcdcoffset = 3.814697265625e-6f;              /* :8 */
    float *output = out0 + start;
    const float *input = input0 + start;
    const size_t len = end - start;
    float l_mul = 0.0f, b_mul = 0.0f, h_mul = 0.0f;
    switch (type) {                                                  /* :90-110 */
    case ZO_FILTER_BYPASS:
        for (size_t i = 0; i < len; i++) output[i] += input[i];
        return;
    case ZO_FILTER_LOW_PASS: l_mul = 1.0f; break;
    case ZO_FILTER_BAND_PASS: b_mul = 1.0f; break;
    case ZO_FILTER_HIGH_PASS: h_mul = 1.0f; break;
    case ZO_FILTER_NOTCH: l_mul = 1.0f; h_mul = 1.0f; break;
    default: l_mul = 1.0f; b_mul = 1.0f; h_mul = 1.0f; break;       /* all_pass */
    }
    const int cut_const = cutoff.tag == ZO_COB_CONSTANT, res_const = res_p.tag == ZO_COB_CONSTANT;
    const float *cut_buf = cut_const ? NULL : cutoff.buffer + start;
    const float *res_buf = res_const ? NULL : res_p.buffer + start;
    float cut = 0.0f, res = 0.0f;
    if (cut_const) cut = zclampf(cutoff.constant, 0.0f, 1.0f);       /* :114 */
    if (res_const) res = 1.0f - zclampf(res_p.constant, 0.0f, 1.0f); /* :118 */
    float l = self->l, b = self->b;
    for (size_t i = 0; i < len; i++) {
        if (!cut_const) cut = zclampf(cut_buf[i], 0.0f, 1.0f);
        if (!res_const) res = 1.0f - zclampf(res_buf[i], 0.0f, 1.0f);
        const float in = input[i] + fcdcoffset;                      /* :135 */
        l += cut * b - fcdcoffset;                                   /* :138 */
        b += cut * (in - b * res - l);                               /* :139 */
        l += cut * b;                                                /* :142 */
        const float h = in - b * res - l;                            /* :143 */
        b += cut * h;                                                /* :144 */
        output[i] += l * l_mul + b * b_mul + h * h_mul;              /* :146 */
    }
    self->l = l; self->b = b;
}

/* ------------------------------------------------------------------ Sampler.zig */
void zo_sampler_init(zo_sampler *s) { s->t = 0.0f; }                 /* Sampler.zig:71-75 */

/* Sampler.zig:23-33 */
static float sampler_decode_signed(int byte_count, const uint8_t *data, size_t index) {
    const uint8_t *p = data + index * (size_t)byte_count;
    int32_t sval;
    if (byte_count == 2) sval = (int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8));
    else if (byte_count == 3) {
        uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
        sval = (int32_t)(u << 8) >> 8;
    } else sval = (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    const float max = (float)(1u << (byte_count * 8 - 1));
    return (float)sval / max;
}

/* Sampler.zig:35-58.  num_samples == 0 with loop is a division-by-zero panic in the
 * reference; DEFINED here as silence. */
static float sampler_get_sample(const zo_sampler_params *p, int32_t index1) {
    const size_t bps = (size_t)p->format + 1u;                       /* u8:1 s16:2 s24:3 s32:4 */
    const int32_t num_samples = (int32_t)(p->data_len / bps / p->num_channels);
    int32_t index = index1;
    if (p->loop) {
        if (num_samples == 0) return 0.0f;
        index = index1 % num_samples;                                /* @mod: floored */
        if (index < 0) index += num_samples;
    }
    if (index >= 0 && index < num_samples) {
        const size_t i = (size_t)index * p->num_channels + p->channel;
        switch (p->format) {
        case ZO_SAMPLE_U8: return ((float)p->data[i] - 127.5f) / 127.5f;
        case ZO_SAMPLE_S16: return sampler_decode_signed(2, p->data, i);
        case ZO_SAMPLE_S24: return sampler_decode_signed(3, p->data, i);
        default: return sampler_decode_signed(4, p->data, i);
        }
    }
    return 0.0f;
}

/* Sampler.zig:77-136 */
void zo_sampler_paint(zo_sampler *self, size_t start, size_t end, float *out0,
                      int note_id_changed, const zo_sampler_params *params) {
    if (params->channel >= params->num_channels) return;             /* :87-89 */
    if (note_id_changed) self->t = 0.0f;                             /* :91-93 */
    float *out = out0 + start;
    const size_t len = end - start;
    const float ratio = (float)params->sample_rate_in / params->sample_rate; /* :97 */
    if (ratio < 0.0f && !params->loop) return;                       /* :99-102 */
    if (ratio > 0.9999f && ratio < 1.0001f) {                        /* :105 */
        const int32_t t = zr_f32_to_i32(zr_roundf(self->t));
        for (size_t i = 0; i < len; i++) out[i] += sampler_get_sample(params, (int32_t)((uint32_t)t + (uint32_t)i));
        self->t += (float)len;                                       /* :114 */
    } else {
        for (size_t i = 0; i < len; i++) {
            const int32_t t0 = zr_f32_to_i32(floorf(self->t));
            const int32_t t1 = (int32_t)((uint32_t)t0 + 1u);
            const float tfrac = (float)t1 - self->t;                 /* :121 */
            const float s0 = sampler_get_sample(params, t0);
            const float s1 = sampler_get_sample(params, t1);
            const float s = s0 * (1.0f - tfrac) + s1 * tfrac;
            out[i] += s;
            self->t += ratio;
        }
    }
    /* :133-135 compares against data.len in BYTES (reference quirk, reproduced) */
    if (self->t >= (float)params->data_len && params->loop) self->t -= (float)params->data_len;
}

/* ------------------------------------------------------------------ Decimator.zig */
void zo_decimator_init(zo_decimator *d) { d->dval = 0.0f; d->dcount = 1.0f; } /* Decimator.zig:14-19 */

/* Decimator.zig:21-57 */
void zo_decimator_paint(zo_decimator *self, size_t start, size_t end, float *output,
                        float sample_rate, const float *input, float fake_sample_rate) {
    if (fake_sample_rate >= sample_rate) {
        zo_add_into(start, end, output, input);
        self->dval = 0.0f; self->dcount = 1.0f;
    } else if (fake_sample_rate > 0.0f) {
        const float ratio = fake_sample_rate / sample_rate;
        float dcount = self->dcount, dval = self->dval;
        for (size_t i = start; i < end; i++) {
            dcount += ratio;
            if (dcount >= 1.0f) { dval = input[i]; dcount -= 1.0f; }
            output[i] += dval;
        }
        self->dcount = dcount; self->dval = dval;
    }
}

/* ------------------------------------------------------------------ Distortion.zig:27-66 */
void zo_distortion_paint(size_t start, size_t end, float *output, const float *input,
                         uint32_t type, float ingain, float outgain, float offset) {
    const float gain1 = zr_powf(2.0f, ingain * 8.0f - 2.0f);         /* :41 */
    if (type == ZO_DISTORTION_OVERDRIVE) {
        const float gain2 = outgain / zr_atanf(gain1);
        const float offs = gain1 * offset;
        for (size_t i = start; i < end; i++) {
            const float a = zr_atanf(input[i] * gain1 + offs);
            output[i] += gain2 * a;
        }
    } else {
        const float gain2 = outgain;
        const float offs = gain1 * offset;
        for (size_t i = start; i < end; i++) {
            const float a = input[i] * gain1 + offs;
            const float b = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
            output[i] += gain2 * b;
        }
    }
}

/* ------------------------------------------------------------------ composites (examples/modules.zig) */
void zo_nice_init(zo_nice_instrument *n, float color) {              /* examples/modules.zig:203-210 */
    n->color = color;
    zo_pulseosc_init(&n->osc); zo_filter_init(&n->flt); zo_envelope_init(&n->env);
}

/* NiceInstrument.paint, examples/modules.zig:212-247 */
void zo_nice_paint(zo_nice_instrument *self, size_t start, size_t end, float *out0,
                   float *temp0, float *temp1, int note_id_changed,
                   float sample_rate, float freq, int note_on) {
    zo_zero(start, end, temp0);
    zo_cob f = { ZO_COB_CONSTANT, freq, NULL };
    zo_pulseosc_paint(&self->osc, start, end, temp0, sample_rate, f, self->color);
    zo_multiply_with_scalar(start, end, temp0, 0.5f);
    zo_zero(start, end, temp1);
    zo_cob cut = { ZO_COB_CONSTANT, zo_filter_cutoff_from_frequency(freq * 8.0f, sample_rate), NULL };
    zo_cob res = { ZO_COB_CONSTANT, 0.7f, NULL };
    zo_filter_paint(&self->flt, start, end, temp1, temp0, ZO_FILTER_LOW_PASS, cut, res);
    zo_zero(start, end, temp0);
    zo_envelope_params ep = { sample_rate, { ZO_CURVE_CUBED, 0.01f }, { ZO_CURVE_CUBED, 0.1f },
                              { ZO_CURVE_CUBED, 0.5f }, 0.8f, note_on };
    zo_envelope_paint(&self->env, start, end, temp0, note_id_changed, &ep);
    zo_multiply(start, end, out0, temp0, temp1);
}

/* note_frequencies.c5 (src/zang-12tet.zig:3-7,96): pow(f32, pow(f32, 2, 1/12), 3), evaluated by the Zig compiler with std.math.pow */
float zo_note_c5(void) { return zr_powf(zr_powf(2.0f, 1.0f / 12.0f), 3.0f); }

/* FilteredSawtoothInstrument (examples/modules.zig:130-187): TriSawOsc(color 0) * 1.5, Envelope(cubed .025 / .1 / 1.0,
 * sustain 0.5), multiply, Filter(low_pass, cutoffFromFrequency(440 * c5), res 0.7) into the output.  `freq` is the
 * reference's ConstantOrBuffer. */
void zo_filtered_sawtooth_init(zo_filtered_sawtooth *s) {            /* :145-151 */
    zo_trisawosc_init(&s->osc); zo_envelope_init(&s->env); zo_filter_init(&s->flt);
}
void zo_filtered_sawtooth_paint(zo_filtered_sawtooth *self, size_t start, size_t end, float *out0,
                                float *temp0, float *temp1, float *temp2, int note_id_changed,
                                float sample_rate, zo_cob freq, int note_on) {
    zo_zero(start, end, temp0);                                                          /* :161 */
    zo_trisawosc_paint(&self->osc, start, end, temp0, sample_rate, freq, 0.0f);          /* :162-166 */
    zo_multiply_with_scalar(start, end, temp0, 1.5f);                                    /* :167 boost sawtooth volume */
    zo_zero(start, end, temp1);                                                          /* :168 */
    zo_envelope_params ep = { sample_rate, { ZO_CURVE_CUBED, 0.025f }, { ZO_CURVE_CUBED, 0.1f },
                              { ZO_CURVE_CUBED, 1.0f }, 0.5f, note_on };
    zo_envelope_paint(&self->env, start, end, temp1, note_id_changed, &ep);              /* :169-176 */
    zo_zero(start, end, temp2);                                                          /* :177 */
    zo_multiply(start, end, temp2, temp0, temp1);                                        /* :178 */
    /* note_frequencies.c5 = semitone +3 of the a4-relative table, examples/common/note_frequencies.zig */
    zo_cob cut = { ZO_COB_CONSTANT, zo_filter_cutoff_from_frequency(440.0f * zo_note_c5(), sample_rate), NULL };
    zo_cob res = { ZO_COB_CONSTANT, 0.7f, NULL };
    zo_filter_paint(&self->flt, start, end, out0, temp2, ZO_FILTER_LOW_PASS, cut, res);  /* :179-187 */
}

/* HardSquareInstrument (examples/modules.zig:250-289): PulseOsc(color 0.5) times Gate, multiplied INTO the output. */
void zo_hard_square_init(zo_hard_square *s) { zo_pulseosc_init(&s->osc); }                /* :263-268 (Gate has no state) */
void zo_hard_square_paint(zo_hard_square *self, size_t start, size_t end, float *out0, float *temp0, float *temp1,
                          int note_id_changed, float sample_rate, float freq, int note_on) {
    (void)note_id_changed;
    zo_zero(start, end, temp0);                                                          /* :278 */
    zo_cob f = { ZO_COB_CONSTANT, freq, NULL };
    zo_pulseosc_paint(&self->osc, start, end, temp0, sample_rate, f, 0.5f);              /* :279-283 */
    zo_zero(start, end, temp1);                                                          /* :284 */
    zo_gate_paint(start, end, temp1, note_on);                                           /* :285-287 */
    zo_multiply(start, end, out0, temp0, temp1);                                         /* :288 */
}

void zo_pmosc_init(zo_pmosc_instrument *p, float release_duration) { /* examples/modules.zig:93-99 */
    p->release_duration = release_duration;
    zo_sineosc_init(&p->carrier); zo_sineosc_init(&p->modulator); zo_envelope_init(&p->env);
}

/* PMOscInstrument.paint (examples/modules.zig:101-127) over PhaseModOscillator.paint
 * (:34-76) with relative = true, ratio = constant(1.0), multiplier = constant(1.0). */
void zo_pmosc_paint(zo_pmosc_instrument *self, size_t start, size_t end, float *out0,
                    float *temp0, float *temp1, float *temp2, int note_id_changed,
                    float sample_rate, float freq, int note_on) {
    zo_zero(start, end, temp0);
    /* PhaseModOscillator.paint(outputs = {temp0}, temps = {temp1, temp2}) */
    zo_set(start, end, temp1, freq * 1.0f);                          /* :45 */
    zo_zero(start, end, temp2);
    zo_cob mf = { ZO_COB_BUFFER, 0.0f, temp1 }, zero_phase = { ZO_COB_CONSTANT, 0.0f, NULL };
    zo_sineosc_paint(&self->modulator, start, end, temp2, sample_rate, mf, zero_phase);
    zo_zero(start, end, temp1);
    zo_multiply_scalar(start, end, temp1, temp2, 1.0f);              /* :66 */
    zo_zero(start, end, temp2);
    zo_cob cf = { ZO_COB_CONSTANT, freq, NULL }, cp = { ZO_COB_BUFFER, 0.0f, temp1 };
    zo_sineosc_paint(&self->carrier, start, end, temp2, sample_rate, cf, cp);
    zo_add_into(start, end, temp0, temp2);                           /* :75 */
    /* back in PMOscInstrument.paint */
    zo_zero(start, end, temp1);
    zo_envelope_params ep = { sample_rate, { ZO_CURVE_CUBED, 0.025f }, { ZO_CURVE_CUBED, 0.1f },
                              { ZO_CURVE_CUBED, self->release_duration }, 0.5f, note_on };
    zo_envelope_paint(&self->env, start, end, temp1, note_id_changed, &ep);
    zo_multiply(start, end, out0, temp0, temp1);
}

/* ------------------------------------------------------------------ mixdown.zig */
/* src/zang/mixdown.zig:28-57 */
void zo_mixdown_s16lsb(uint8_t *dst, const float *mix, size_t n, size_t num_channels,
                       size_t channel_index, float vol) {
    const float mul = vol * 32767.0f;
    for (size_t i = 0; i < n; i++) {
        const float value = mix[i] * mul;
        int16_t c;
        if (value <= -32767.0f) c = -32767;
        else if (value >= 32766.0f) c = 32766;
        else if (value != value) c = 0;
        else c = (int16_t)value;
        const size_t index = (i * num_channels + channel_index) * 2;
        dst[index + 0] = (uint8_t)(c & 0xFF);
        dst[index + 1] = (uint8_t)((c >> 8) & 0xFF);
    }
}
/* src/zang/mixdown.zig:59-86 */
void zo_mixdown_s8(uint8_t *dst, const float *mix, size_t n, size_t num_channels,
                   size_t channel_index, float vol) {
    const float mul = vol * 127.0f;
    for (size_t i = 0; i < n; i++) {
        const float value = mix[i] * mul;
        int8_t c;
        if (value <= -127.0f) c = -127;
        else if (value >= 126.0f) c = 126;
        else if (value != value) c = 0;
        else c = (int8_t)value;
        dst[i * num_channels + channel_index] = (uint8_t)c;
    }
}

/* exported math probes for tests/test_oracle_math.py */
float zo_math_sinf(float x) { return zr_sinf(x); }
float zo_math_cosf(float x) { return zr_cosf(x); }
float zo_math_atanf(float x) { return zr_atanf(x); }
float zo_math_powf(float x, float y) { return zr_powf(x, y); }
float zo_math_expf(float x) { return zr_expf(x); }
float zo_math_logf(float x) { return zr_logf(x); }
void zo_math_sinf_n(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = zr_sinf(x[i]); }
void zo_math_cosf_n(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = zr_cosf(x[i]); }
void zo_math_atanf_n(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = zr_atanf(x[i]); }
void zo_math_pow2f_n(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = zr_powf(2.0f, x[i]); }
void zo_xoshiro_seq(uint64_t seed, uint64_t *state_out4, uint64_t *out, size_t n) {
    zr_xoshiro r; zr_xoshiro_init(&r, seed);
    memcpy(state_out4, r.s, 32);
    for (size_t i = 0; i < n; i++) out[i] = zr_xoshiro_next(&r);
}

/* ------------------------------------------------------------------ cpu_baseline drivers
 * bench.py times these on the host cores ("port" baseline: the reference binary cannot be
 * built here).  They run the reference's per-buffer sequence for `voices` independent module
 * instances, one voice after another on one thread -- the reference's execution model
 * (one audio thread, examples/example.zig:35).  `scratch` is one voice's []f32 of `frames`
 * floats, reused like the reference reuses its temp buffers; the returned value folds
 * every sample so the work cannot be optimised away. */
double zo_bench_pulseosc(uint32_t voices, uint32_t frames, uint32_t buffers, float sample_rate,
                         const float *freq, const float *color, zo_pulseosc *states, float *scratch) {
    double fold = 0.0;
    for (uint32_t b = 0; b < buffers; b++) {
        for (uint32_t v = 0; v < voices; v++) {
            zo_cob f = { ZO_COB_CONSTANT, freq[v], NULL };
            zo_zero(0, frames, scratch);                                    /* examples/modules.zig:220 */
            zo_pulseosc_paint(&states[v], 0, frames, scratch, sample_rate, f, color[v]);
            fold += scratch[(b + v) % frames];
        }
    }
    return fold;
}

/* config 3: Noise(white) -> temp -> Filter(low_pass), per voice per buffer (examples/example_stereo.zig:71-82) */
double zo_bench_noise_filter(uint32_t voices, uint32_t frames, uint32_t buffers, const float *cutoff, const float *res,
                             zo_noise *noise, zo_filter *flt, float *scratch /* 2*frames */) {
    double fold = 0.0;
    float *temp = scratch, *out = scratch + frames;
    for (uint32_t b = 0; b < buffers; b++) {
        for (uint32_t v = 0; v < voices; v++) {
            zo_cob c = { ZO_COB_CONSTANT, cutoff[v], NULL }, r = { ZO_COB_CONSTANT, res[v], NULL };
            zo_zero(0, frames, temp);
            zo_noise_paint(&noise[v], 0, frames, temp, ZO_NOISE_WHITE);
            zo_zero(0, frames, out);
            zo_filter_paint(&flt[v], 0, frames, out, temp, ZO_FILTER_LOW_PASS, c, r);
            fold += out[(b + v) % frames];
        }
    }
    return fold;
}

/* config 5 voice: NiceInstrument, note on for buffers 0-23 of every 48, then off */
double zo_bench_nice(uint32_t voices, uint32_t frames, uint32_t buffers, float sample_rate, const float *freq,
                     zo_nice_instrument *inst, float *scratch /* 3*frames */) {
    double fold = 0.0;
    float *t0 = scratch, *t1 = scratch + frames, *out = scratch + 2 * frames;
    for (uint32_t b = 0; b < buffers; b++) {
        const uint32_t k = b % 48;
        for (uint32_t v = 0; v < voices; v++) {
            zo_zero(0, frames, out);
            zo_nice_paint(&inst[v], 0, frames, out, t0, t1, k == 0, sample_rate, freq[v], k < 24);
            fold += out[(b + v) % frames];
        }
    }
    return fold;
}

/* ------------------------------------------------------------------ Cycle.zig */
void zo_cycle_init(zo_cycle *c) { c->t = 0.0f; }                    /* Cycle.zig:16-20 */

/* Cycle.zig:22-59 */
void zo_cycle_paint(zo_cycle *self, size_t start, size_t end, float *out, float sample_rate, zo_cob speed) {
    float t = self->t;
    if (speed.tag == ZO_COB_CONSTANT) {
        const float step = speed.constant / sample_rate;            /* :37 */
        for (size_t i = start; i < end; i++) { out[i] += t; t += step; t -= truncf(t); }
    } else {
        const float isr = 1.0f / sample_rate;                       /* :48 */
        for (size_t i = start; i < end; i++) { out[i] += t; t += speed.buffer[i] * isr; t -= truncf(t); }
    }
    self->t = t;
}

/* ------------------------------------------------------------------ Portamento.zig */
void zo_portamento_init(zo_portamento *p) { p->painter.t = 0.0f; p->painter.last_value = 0.0f; p->painter.start = 0.0f; }

/* Portamento.zig:21-48 */
void zo_portamento_paint(zo_portamento *self, size_t start, size_t end, float *out0, int note_id_changed,
                         float sample_rate, zo_curve curve_p, float goal, int note_on, int prev_note_on) {
    float *output = out0 + start;
    zo_curve curve = (note_on && prev_note_on) ? curve_p : (zo_curve){ ZO_CURVE_INSTANTANEOUS, 0.0f };   /* :33-36 */
    if (note_on && note_id_changed) painter_new_curve(&self->painter);                                  /* :38-40 */
    zo_paint_state ps = { output, end - start, 0, sample_rate };
    if (painter_paint_toward(&self->painter, &ps, curve, goal))                                         /* :43 */
        painter_paint_flat(&self->painter, &ps, goal);                                                  /* :46 */
}

/* ------------------------------------------------------------------ Curve.zig */
void zo_curve_init(zo_curve_module *c) {                             /* Curve.zig:46-54 */
    c->t = 0.0f; c->current_song_note = 0; c->current_song_note_offset = 0; c->next_song_note = 0;
}

typedef struct { int32_t frame; float value; } curve_span_node;      /* Curve.zig:11-14 */

/* Curve.zig:130-184.  A 33rd node in one paint overflows the reference's [32] array (UB); DEFINED: dropped. */
static size_t curve_get_span_nodes(zo_curve_module *self, float sample_rate, size_t out_len, const zo_curve_node *curve,
                                   size_t n_curve, curve_span_node *nodes) {
    size_t count = 0;
    const float buf_time = (float)out_len / sample_rate;
    const float end_t = self->t + buf_time;
    if (self->current_song_note < self->next_song_note) {            /* :142-148 carry-over */
        nodes[count].frame = self->current_song_note_offset;
        nodes[count].value = curve[self->current_song_note].value;
        count++;
    }
    int one_past = 0;
    for (size_t k = self->next_song_note; k < n_curve; k++) {
        const float note_t = curve[k].t;
        if (note_t >= end_t) {                                       /* :153-160 keep one note past the end */
            if (!one_past) one_past = 1; else break;
        }
        const float f = (note_t - self->t) / buf_time;
        const int32_t rel_frame_index = zr_f32_to_i32(f * (float)out_len);
        if (count > 0 && nodes[count - 1].frame == rel_frame_index) count--;   /* :165-167 */
        if (count < 32) {
            nodes[count].frame = rel_frame_index;
            nodes[count].value = curve[k].value;
            count++;
        }
        if (!one_past) {                                             /* :173-177 */
            self->current_song_note = self->next_song_note;
            self->current_song_note_offset = 0;
            self->next_song_note += 1;
        }
    }
    self->t += buf_time;                                             /* :180 */
    self->current_song_note_offset -= (int32_t)out_len;              /* :181 */
    return count;
}

typedef struct { size_t start, end; int has_values; curve_span_node a, b; } curve_span;

/* Curve.zig:188-255 */
static curve_span curve_next_span(const curve_span_node *nodes, size_t n, size_t dest_start_, size_t dest_end_) {
    const int32_t dest_start = (int32_t)dest_start_, dest_end = (int32_t)dest_end_;
    for (size_t i = 0; i < n; i++) {
        const int32_t start_pos = nodes[i].frame;
        if (start_pos >= dest_end) break;
        const int32_t end_pos = (i + 1 < n) ? (nodes[i + 1].frame < dest_end ? nodes[i + 1].frame : dest_end) : dest_end;
        if (end_pos <= dest_start) continue;
        const int32_t note_start_clipped = start_pos > dest_start ? start_pos : dest_start;
        if (note_start_clipped > dest_start)
            return (curve_span){ (size_t)dest_start, (size_t)note_start_clipped, 0, {0, 0}, {0, 0} };
        const int32_t note_end_clipped = end_pos > dest_end ? dest_end : end_pos;
        curve_span s = { (size_t)note_start_clipped, (size_t)note_end_clipped, i + 1 < n, nodes[i], {0, 0} };
        if (i + 1 < n) s.b = nodes[i + 1];
        return s;
    }
    return (curve_span){ (size_t)dest_start, (size_t)dest_end, 0, {0, 0}, {0, 0} };
}

/* Curve.zig:56-128 */
void zo_curve_paint(zo_curve_module *self, size_t span_start, size_t span_end, float *out0, int note_id_changed,
                    float sample_rate, uint32_t function, const zo_curve_node *curve, size_t n_curve) {
    if (note_id_changed) zo_curve_init(self);                        /* :66-71 */
    float *out = out0 + span_start;
    const size_t out_len = span_end - span_start;
    curve_span_node nodes[32];
    const size_t n = curve_get_span_nodes(self, sample_rate, out_len, curve, n_curve, nodes);
    size_t start = 0;
    while (start < out_len) {
        const curve_span cs = curve_next_span(nodes, n, start, out_len);
        if (cs.has_values) {
            const int32_t paint_start = (int32_t)cs.start;
            const int32_t fstart = cs.a.frame, fend = cs.b.frame;
            const float start_x = (float)(paint_start - fstart) / (float)(fend - fstart);   /* :95 */
            const float start_value = cs.a.value;
            const float value_delta = cs.b.value - cs.a.value;
            const float x_step = 1.0f / (float)(fend - fstart);      /* :100 */
            if (function == ZO_CURVE_FN_LINEAR) {
                float y = start_value + start_x * value_delta;
                const float y_step = x_step * value_delta;
                for (size_t i = cs.start; i < cs.end; i++) { out[i] += y; y += y_step; }
            } else {
                float x = start_x;
                for (size_t i = cs.start; i < cs.end; i++) {
                    const float v = x * x * (3.0f - 2.0f * x) * value_delta;   /* :118 */
                    out[i] += start_value + v;
                    x += x_step;
                }
            }
        }
        start = cs.end;
    }
}

/* ------------------------------------------------------------------ delay.zig + the delay composites
 * zo_delay holds a caller-allocated ring of `delay_samples` floats (Delay(n), src/zang/delay.zig:7-18). */
void zo_delay_init(zo_delay *d, float *ring, size_t delay_samples) {
    d->ring = ring; d->delay_samples = delay_samples; d->index = 0;
    for (size_t i = 0; i < delay_samples; i++) ring[i] = 0.0f;
}

/* delay.zig:28-57 */
static size_t delay_read(zo_delay *d, float *out, size_t out_len) {
    const size_t actual = out_len > d->delay_samples ? d->delay_samples : out_len;
    const size_t index = d->index;
    const size_t len = (d->delay_samples - index) < actual ? (d->delay_samples - index) : actual;
    for (size_t i = 0; i < len; i++) out[i] += d->ring[index + i];
    if (len < actual) {
        const size_t b_len = actual - len;
        for (size_t i = 0; i < b_len; i++) out[len + i] += d->ring[i];
    }
    return actual;
}

/* delay.zig:62-89 */
static void delay_write(zo_delay *d, const float *input, size_t n) {
    const size_t index = d->index;
    const size_t len = (d->delay_samples - index) < n ? (d->delay_samples - index) : n;
    memcpy(d->ring + index, input, len * sizeof(float));
    if (len < n) {
        const size_t b_len = n - len;
        memcpy(d->ring, input + len, b_len * sizeof(float));
        d->index = b_len;
    } else {
        d->index += len;
        if (d->index == d->delay_samples) d->index = 0;
    }
}

/* SimpleDelay.paint, examples/modules.zig:363-385 */
void zo_simple_delay_paint(zo_delay *self, size_t start, size_t end, float *out0, const float *input) {
    while (start < end) {
        const size_t samples_read = delay_read(self, out0 + start, end - start);
        delay_write(self, input + start, samples_read);
        start += samples_read;
    }
}

/* FilteredEchoes.paint, examples/modules.zig:411-460 */
void zo_filtered_echoes_paint(zo_delay *delay, zo_filter *filter, size_t start, size_t end, float *output,
                              float *temp0, float *temp1, const float *input, float feedback_volume, float cutoff) {
    while (start < end) {
        zo_zero(start, end, temp0);
        const size_t samples_read = delay_read(delay, temp0 + start, end - start);
        const size_t s1 = start, e1 = start + samples_read;
        zo_multiply_with_scalar(s1, e1, temp0, feedback_volume);
        zo_add_into(s1, e1, temp0, input);
        zo_zero(s1, e1, temp1);
        zo_cob c = { ZO_COB_CONSTANT, cutoff, NULL }, r = { ZO_COB_CONSTANT, 0.0f, NULL };
        zo_filter_paint(filter, s1, e1, temp1, temp0, ZO_FILTER_LOW_PASS, c, r);
        zo_add_into(s1, e1, output, temp1);
        delay_write(delay, temp1 + s1, samples_read);
        start += samples_read;
    }
}
