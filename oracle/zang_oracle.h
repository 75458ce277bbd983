/* zang_oracle.h -- ORACLE (test infrastructure only).  Declarations for
 * zang_oracle.c: a scalar single-voice CPU restatement of zang's paint() path.
 * Buffers are per-voice contiguous float[frames], exactly like the reference's []f32;
 * (start, end) is zang.Span (src/zang/basics.zig:3-10). */
#ifndef ZANG_ORACLE_H
#define ZANG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

/* zang.ConstantOrBuffer, src/zang/constant_or_buffer.zig:4-7 */
enum { ZO_COB_CONSTANT = 0, ZO_COB_BUFFER = 1 };
typedef struct { uint32_t tag; float constant; const float *buffer; } zo_cob;

/* zang.PaintCurve, src/zang/painter.zig:25-30 */
enum { ZO_CURVE_INSTANTANEOUS = 0, ZO_CURVE_LINEAR = 1, ZO_CURVE_SQUARED = 2, ZO_CURVE_CUBED = 3 };
typedef struct { uint32_t tag; float duration; } zo_curve;

/* zang.Painter, src/zang/painter.zig:33-44 */
typedef struct { float t, last_value, start; } zo_painter;

typedef struct { float t; } zo_sineosc;                    /* SineOsc.zig:16 */
typedef struct { uint32_t cnt; } zo_pulseosc;              /* PulseOsc.zig:36 */
typedef struct { uint32_t cnt; float t; } zo_trisawosc;    /* TriSawOsc.zig:36-37 */
typedef struct { uint64_t r[4]; float b[7]; } zo_noise;    /* Noise.zig:22-23 */
enum { ZO_NOISE_WHITE = 0, ZO_NOISE_PINK = 1 };            /* Noise.zig:11-14 */
enum { ZO_ENV_IDLE = 0, ZO_ENV_ATTACK, ZO_ENV_DECAY, ZO_ENV_SUSTAIN, ZO_ENV_RELEASE }; /* Envelope.zig:15-21 */
typedef struct { uint32_t state; zo_painter painter; } zo_envelope;   /* Envelope.zig:23-24 */
typedef struct {                                            /* Envelope.zig:6-13 */
    float sample_rate; zo_curve attack, decay, release; float sustain_volume; int32_t note_on;
} zo_envelope_params;
typedef struct { float l, b; } zo_filter;                  /* Filter.zig:34-35 */
enum { ZO_FILTER_BYPASS = 0, ZO_FILTER_LOW_PASS, ZO_FILTER_BAND_PASS, ZO_FILTER_HIGH_PASS,
       ZO_FILTER_NOTCH, ZO_FILTER_ALL_PASS };              /* Filter.zig:10-17 */
typedef struct { float t; } zo_sampler;                    /* Sampler.zig:69 */
enum { ZO_SAMPLE_U8 = 0, ZO_SAMPLE_S16, ZO_SAMPLE_S24, ZO_SAMPLE_S32 }; /* Sampler.zig:9-14 */
typedef struct {                                            /* Sampler.zig:16-21, 62-67 */
    float sample_rate;          /* Params.sample_rate (output rate) */
    size_t num_channels;        /* Sample.num_channels */
    size_t sample_rate_in;      /* Sample.sample_rate */
    uint32_t format;            /* Sample.format */
    const uint8_t *data; size_t data_len; /* Sample.data */
    size_t channel; int32_t loop;
} zo_sampler_params;
typedef struct { float dval, dcount; } zo_decimator;       /* Decimator.zig:11-12 */
enum { ZO_DISTORTION_OVERDRIVE = 0, ZO_DISTORTION_CLIP = 1 }; /* Distortion.zig:8-11 */
enum { ZO_CURVE_FN_LINEAR = 0, ZO_CURVE_FN_SMOOTHSTEP = 1 };  /* Curve.zig:4-7 */
typedef struct { float value, t; } zo_curve_node;          /* zang.CurveNode, src/zang/curve.zig:3-6 */
typedef struct { float t; size_t current_song_note; int32_t current_song_note_offset; size_t next_song_note; } zo_curve_module; /* Curve.zig:36-41 */
typedef struct { float t; } zo_cycle;                      /* Cycle.zig:14 */
typedef struct { zo_painter painter; } zo_portamento;      /* Portamento.zig:13 */
typedef struct { float *ring; size_t delay_samples; size_t index; } zo_delay;   /* Delay(n), src/zang/delay.zig:9-10 */
typedef struct { float color; zo_pulseosc osc; zo_filter flt; zo_envelope env; } zo_nice_instrument;
typedef struct { float release_duration; zo_sineosc carrier, modulator; zo_envelope env; } zo_pmosc_instrument;
typedef struct { zo_trisawosc osc; zo_envelope env; zo_filter flt; } zo_filtered_sawtooth;   /* examples/modules.zig:141-143 */
typedef struct { zo_pulseosc osc; } zo_hard_square;                                        /* examples/modules.zig:260-261 */

void zo_zero(size_t start, size_t end, float *dest);
void zo_set(size_t start, size_t end, float *dest, float a);
void zo_copy(size_t start, size_t end, float *dest, const float *src);
void zo_add(size_t start, size_t end, float *dest, const float *a, const float *b);
void zo_add_into(size_t start, size_t end, float *dest, const float *src);
void zo_add_scalar(size_t start, size_t end, float *dest, const float *a, float b);
void zo_add_scalar_into(size_t start, size_t end, float *dest, float a);
void zo_multiply(size_t start, size_t end, float *dest, const float *a, const float *b);
void zo_multiply_with(size_t start, size_t end, float *dest, const float *a);
void zo_multiply_scalar(size_t start, size_t end, float *dest, const float *a, float b);
void zo_multiply_with_scalar(size_t start, size_t end, float *dest, float a);

void zo_sineosc_init(zo_sineosc *s);
void zo_sineosc_paint(zo_sineosc *self, size_t start, size_t end, float *out0, float sample_rate, zo_cob freq, zo_cob phase);
void zo_pulseosc_init(zo_pulseosc *s);
void zo_pulseosc_paint(zo_pulseosc *self, size_t start, size_t end, float *out0, float sample_rate, zo_cob freq, float color);
void zo_trisawosc_init(zo_trisawosc *s);
void zo_trisawosc_paint(zo_trisawosc *self, size_t start, size_t end, float *out0, float sample_rate, zo_cob freq, float color);
void zo_noise_init(zo_noise *n, uint64_t seed);
void zo_noise_paint(zo_noise *self, size_t start, size_t end, float *out, uint32_t color);
void zo_envelope_init(zo_envelope *e);
void zo_envelope_paint(zo_envelope *self, size_t start, size_t end, float *out0, int note_id_changed, const zo_envelope_params *params);
void zo_gate_paint(size_t start, size_t end, float *out0, int note_on);
void zo_filter_init(zo_filter *f);
float zo_filter_cutoff_from_frequency(float frequency, float sample_rate);
void zo_filter_paint(zo_filter *self, size_t start, size_t end, float *out0, const float *input0, uint32_t type, zo_cob cutoff, zo_cob res);
void zo_sampler_init(zo_sampler *s);
void zo_sampler_paint(zo_sampler *self, size_t start, size_t end, float *out0, int note_id_changed, const zo_sampler_params *params);
void zo_decimator_init(zo_decimator *d);
void zo_decimator_paint(zo_decimator *self, size_t start, size_t end, float *output, float sample_rate, const float *input, float fake_sample_rate);
void zo_distortion_paint(size_t start, size_t end, float *output, const float *input, uint32_t type, float ingain, float outgain, float offset);
void zo_nice_init(zo_nice_instrument *n, float color);
void zo_nice_paint(zo_nice_instrument *self, size_t start, size_t end, float *out0, float *temp0, float *temp1,
                   int note_id_changed, float sample_rate, float freq, int note_on);
void zo_filtered_sawtooth_init(zo_filtered_sawtooth *s);
void zo_filtered_sawtooth_paint(zo_filtered_sawtooth *self, size_t start, size_t end, float *out0, float *temp0, float *temp1, float *temp2,
                                int note_id_changed, float sample_rate, zo_cob freq, int note_on);
void zo_hard_square_init(zo_hard_square *s);
void zo_hard_square_paint(zo_hard_square *self, size_t start, size_t end, float *out0, float *temp0, float *temp1,
                          int note_id_changed, float sample_rate, float freq, int note_on);
float zo_note_c5(void);
void zo_pmosc_init(zo_pmosc_instrument *p, float release_duration);
void zo_pmosc_paint(zo_pmosc_instrument *self, size_t start, size_t end, float *out0, float *temp0, float *temp1,
                    float *temp2, int note_id_changed, float sample_rate, float freq, int note_on);
void zo_mixdown_s16lsb(uint8_t *dst, const float *mix, size_t n, size_t num_channels, size_t channel_index, float vol);
void zo_mixdown_s8(uint8_t *dst, const float *mix, size_t n, size_t num_channels, size_t channel_index, float vol);
void zo_curve_init(zo_curve_module *c);
void zo_curve_paint(zo_curve_module *self, size_t span_start, size_t span_end, float *out0, int note_id_changed,
                    float sample_rate, uint32_t function, const zo_curve_node *curve, size_t n_curve);
void zo_cycle_init(zo_cycle *c);
void zo_cycle_paint(zo_cycle *self, size_t start, size_t end, float *out, float sample_rate, zo_cob speed);
void zo_portamento_init(zo_portamento *p);
void zo_portamento_paint(zo_portamento *self, size_t start, size_t end, float *out0, int note_id_changed,
                         float sample_rate, zo_curve curve, float goal, int note_on, int prev_note_on);
void zo_delay_init(zo_delay *d, float *ring, size_t delay_samples);
void zo_simple_delay_paint(zo_delay *self, size_t start, size_t end, float *out0, const float *input);
void zo_filtered_echoes_paint(zo_delay *delay, zo_filter *filter, size_t start, size_t end, float *output,
                              float *temp0, float *temp1, const float *input, float feedback_volume, float cutoff);
double zo_bench_pulseosc(uint32_t voices, uint32_t frames, uint32_t buffers, float sample_rate,
                         const float *freq, const float *color, zo_pulseosc *states, float *scratch);
double zo_bench_noise_filter(uint32_t voices, uint32_t frames, uint32_t buffers, const float *cutoff, const float *res,
                             zo_noise *noise, zo_filter *flt, float *scratch);
double zo_bench_nice(uint32_t voices, uint32_t frames, uint32_t buffers, float sample_rate, const float *freq,
                     zo_nice_instrument *inst, float *scratch);
#endif
