// host_paint.hip -- single-voice, host-pointer wrappers over the batched paint entry points
// (include/zang_hip.h, "single-voice host-pointer wrappers").  Host code only: stage the voice's
// []f32 slices into one-voice device images, run the batched paint with n = 1, copy back.
#include "common.hip.h"
#include <vector>

namespace {

// One-voice device images for this call; freed on scope exit.
struct Stage {
    zh_ctx *ctx;
    std::vector<zh_buf> bufs;
    int rc = ZH_OK;
    explicit Stage(zh_ctx *c) : ctx(c) {}
    ~Stage() { for (auto &b : bufs) zh_buf_free(ctx, &b); }
    zh_buf image(const float *host, uint32_t frames) {          // upload host[0..frames)
        zh_buf b{};
        if (rc) return b;
        rc = zh_buf_alloc(ctx, &b, 1, frames ? frames : 1);
        if (rc) return b;
        bufs.push_back(b);
        if (frames) rc = zh_buf_upload_voice(ctx, b, 0, host, frames);
        return b;
    }
    zh_cob cob(const zh_hcob &h, uint32_t frames) {
        zh_cob c{};
        c.tag = h.tag;
        c.constant.value = h.constant;
        if (h.tag == ZH_COB_BUFFER) {
            if (!h.buffer) { rc = rc ? rc : ZH_ERR_INVALID; return c; }
            c.buffer = image(h.buffer, frames);
        }
        return c;
    }
};

inline zh_f32 f32v(float x) { zh_f32 f{}; f.value = x; return f; }
inline zh_bool boolv(uint32_t x) { zh_bool b{}; b.value = x ? 1u : 0u; return b; }
inline zh_curve curvev(const zh_hcurve &h) { zh_curve c{}; c.tag = h.tag; c.duration.value = h.duration; return c; }

// Generic driver: create(1) -> set_state -> paint -> download outputs[0][start..end) -> get_state -> destroy
template <class M, class State, class Create, class Destroy, class SetState, class GetState, class Paint>
int run_voice(zh_ctx *ctx, State *state, uint32_t start, uint32_t end, float *const *outputs, Stage &st, Create create,
              Destroy destroy, SetState set_state, GetState get_state, Paint paint) {
    if (!ctx || !outputs || !outputs[0] || end < start) return ZH_ERR_INVALID;
    zh_buf out = st.image(outputs[0], end);
    if (st.rc) return st.rc;
    M *m = nullptr;
    int rc = create(&m);
    if (rc) return rc;
    if (state) rc = set_state(m, state);
    if (!rc) rc = paint(m, &out);
    if (!rc && end > start) {
        std::vector<float> tmp(end);
        rc = zh_buf_download_voice(ctx, tmp.data(), out, 0, end);
        if (!rc) for (uint32_t i = start; i < end; i++) outputs[0][i] = tmp[i];
    }
    if (!rc && state) rc = get_state(m, state);
    destroy(m);
    return rc;
}

}  // namespace

extern "C" {

int zh_noise_state_init(zh_noise_state *state, uint64_t seed) {
    if (!state) return ZH_ERR_INVALID;
    uint64_t sm = seed;                                   // std.rand.Xoshiro256.init: four SplitMix64 outputs
    for (int i = 0; i < 4; i++) {
        sm += 0x9e3779b97f4a7c15ull;
        uint64_t z = sm;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        state->r[i] = z ^ (z >> 31);
    }
    for (int j = 0; j < 7; j++) state->b[j] = 0.0f;
    state->reserved = 0;
    return ZH_OK;
}

int zh_decimator_state_init(zh_decimator_state *state) {
    if (!state) return ZH_ERR_INVALID;
    state->dval = 0.0f;
    state->dcount = 1.0f;
    return ZH_OK;
}

int zh_sineosc_paint_host(zh_ctx *ctx, zh_sineosc_state *state, uint32_t start, uint32_t end, float *const *outputs,
                          float *const *temps, uint32_t nic, const zh_sineosc_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_sineosc_params dp{};
    dp.sample_rate = p->sample_rate; dp.freq = st.cob(p->freq, end); dp.phase = st.cob(p->phase, end);
    if (st.rc) return st.rc;
    return run_voice<zh_sineosc>(ctx, state, start, end, outputs, st, [&](zh_sineosc **m) { return zh_sineosc_create(ctx, 1, m); },
        zh_sineosc_destroy, zh_sineosc_set_state, zh_sineosc_get_state,
        [&](zh_sineosc *m, const zh_buf *o) { return zh_sineosc_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_pulseosc_paint_host(zh_ctx *ctx, zh_pulseosc_state *state, uint32_t start, uint32_t end, float *const *outputs,
                           float *const *temps, uint32_t nic, const zh_pulseosc_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_pulseosc_params dp{};
    dp.sample_rate = p->sample_rate; dp.freq = st.cob(p->freq, end); dp.color = f32v(p->color);
    if (st.rc) return st.rc;
    return run_voice<zh_pulseosc>(ctx, state, start, end, outputs, st, [&](zh_pulseosc **m) { return zh_pulseosc_create(ctx, 1, m); },
        zh_pulseosc_destroy, zh_pulseosc_set_state, zh_pulseosc_get_state,
        [&](zh_pulseosc *m, const zh_buf *o) { return zh_pulseosc_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_trisawosc_paint_host(zh_ctx *ctx, zh_trisawosc_state *state, uint32_t start, uint32_t end, float *const *outputs,
                            float *const *temps, uint32_t nic, const zh_trisawosc_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_trisawosc_params dp{};
    dp.sample_rate = p->sample_rate; dp.freq = st.cob(p->freq, end); dp.color = f32v(p->color);
    if (st.rc) return st.rc;
    return run_voice<zh_trisawosc>(ctx, state, start, end, outputs, st, [&](zh_trisawosc **m) { return zh_trisawosc_create(ctx, 1, m); },
        zh_trisawosc_destroy, zh_trisawosc_set_state, zh_trisawosc_get_state,
        [&](zh_trisawosc *m, const zh_buf *o) { return zh_trisawosc_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_noise_paint_host(zh_ctx *ctx, zh_noise_state *state, uint32_t start, uint32_t end, float *const *outputs,
                        float *const *temps, uint32_t nic, const zh_noise_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_noise_params dp{p->color};
    return run_voice<zh_noise>(ctx, state, start, end, outputs, st, [&](zh_noise **m) { return zh_noise_create(ctx, 1, 0, m); },
        zh_noise_destroy, zh_noise_set_state, zh_noise_get_state,
        [&](zh_noise *m, const zh_buf *o) { return zh_noise_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_envelope_paint_host(zh_ctx *ctx, zh_envelope_state *state, uint32_t start, uint32_t end, float *const *outputs,
                           float *const *temps, uint32_t nic, const zh_envelope_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_envelope_params dp{};
    dp.sample_rate = p->sample_rate; dp.attack = curvev(p->attack); dp.decay = curvev(p->decay); dp.release = curvev(p->release);
    dp.sustain_volume = f32v(p->sustain_volume); dp.note_on = boolv(p->note_on);
    return run_voice<zh_envelope>(ctx, state, start, end, outputs, st, [&](zh_envelope **m) { return zh_envelope_create(ctx, 1, m); },
        zh_envelope_destroy, zh_envelope_set_state, zh_envelope_get_state,
        [&](zh_envelope *m, const zh_buf *o) { return zh_envelope_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_gate_paint_host(zh_ctx *ctx, void *state_unused, uint32_t start, uint32_t end, float *const *outputs,
                       float *const *temps, uint32_t nic, const zh_gate_host_params *p) { ZH_GUARD(ctx);
    (void)temps; (void)state_unused;
    if (!ctx || !p) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_gate_params dp{boolv(p->note_on)};
    return run_voice<zh_gate>(ctx, (int *)nullptr, start, end, outputs, st, [&](zh_gate **m) { return zh_gate_create(ctx, 1, m); },
        zh_gate_destroy, [](zh_gate *, int *) { return ZH_OK; }, [](zh_gate *, int *) { return ZH_OK; },
        [&](zh_gate *m, const zh_buf *o) { return zh_gate_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_filter_paint_host(zh_ctx *ctx, zh_filter_state *state, uint32_t start, uint32_t end, float *const *outputs,
                         float *const *temps, uint32_t nic, const zh_filter_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p || !p->input) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_filter_params dp{};
    dp.input = st.image(p->input, end); dp.type = p->type; dp.cutoff = st.cob(p->cutoff, end); dp.res = st.cob(p->res, end);
    if (st.rc) return st.rc;
    return run_voice<zh_filter>(ctx, state, start, end, outputs, st, [&](zh_filter **m) { return zh_filter_create(ctx, 1, m); },
        zh_filter_destroy, zh_filter_set_state, zh_filter_get_state,
        [&](zh_filter *m, const zh_buf *o) { return zh_filter_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_sampler_paint_host(zh_ctx *ctx, zh_sampler_state *state, uint32_t start, uint32_t end, float *const *outputs,
                          float *const *temps, uint32_t nic, const zh_sampler_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p || (p->data_len && !p->data)) return ZH_ERR_INVALID;
    Stage st(ctx);
    void *pcm = nullptr;
    int rc = zh_malloc(ctx, &pcm, p->data_len ? p->data_len : 1);
    if (!rc && p->data_len) rc = zh_upload(ctx, pcm, p->data, p->data_len);
    if (rc) { zh_free(ctx, pcm); return rc; }
    zh_sampler_params dp{};
    dp.sample_rate = f32v(p->sample_rate);
    dp.sample.num_channels = p->num_channels; dp.sample.sample_rate = p->sample_rate_in; dp.sample.format = p->format;
    dp.sample.data = (const uint8_t *)pcm; dp.sample.data_len = p->data_len;
    dp.channel = p->channel; dp.loop = p->loop;
    rc = run_voice<zh_sampler>(ctx, state, start, end, outputs, st, [&](zh_sampler **m) { return zh_sampler_create(ctx, 1, m); },
        zh_sampler_destroy, zh_sampler_set_state, zh_sampler_get_state,
        [&](zh_sampler *m, const zh_buf *o) { return zh_sampler_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
    zh_free(ctx, pcm);
    return rc;
}

int zh_decimator_paint_host(zh_ctx *ctx, zh_decimator_state *state, uint32_t start, uint32_t end, float *const *outputs,
                            float *const *temps, uint32_t nic, const zh_decimator_host_params *p) { ZH_GUARD(ctx);
    (void)temps;
    if (!ctx || !state || !p || !p->input) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_decimator_params dp{};
    dp.sample_rate = p->sample_rate; dp.input = st.image(p->input, end); dp.fake_sample_rate = f32v(p->fake_sample_rate);
    if (st.rc) return st.rc;
    return run_voice<zh_decimator>(ctx, state, start, end, outputs, st, [&](zh_decimator **m) { return zh_decimator_create(ctx, 1, m); },
        zh_decimator_destroy, zh_decimator_set_state, zh_decimator_get_state,
        [&](zh_decimator *m, const zh_buf *o) { return zh_decimator_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

int zh_distortion_paint_host(zh_ctx *ctx, void *state_unused, uint32_t start, uint32_t end, float *const *outputs,
                             float *const *temps, uint32_t nic, const zh_distortion_host_params *p) { ZH_GUARD(ctx);
    (void)temps; (void)state_unused;
    if (!ctx || !p || !p->input) return ZH_ERR_INVALID;
    Stage st(ctx);
    zh_distortion_params dp{};
    dp.input = st.image(p->input, end); dp.type = p->type; dp.ingain = f32v(p->ingain); dp.outgain = f32v(p->outgain); dp.offset = f32v(p->offset);
    if (st.rc) return st.rc;
    return run_voice<zh_distortion>(ctx, (int *)nullptr, start, end, outputs, st, [&](zh_distortion **m) { return zh_distortion_create(ctx, 1, m); },
        zh_distortion_destroy, [](zh_distortion *, int *) { return ZH_OK; }, [](zh_distortion *, int *) { return ZH_OK; },
        [&](zh_distortion *m, const zh_buf *o) { return zh_distortion_paint(m, start, end, o, nullptr, boolv(nic), &dp, ZH_PAINT_ADD); });
}

}  // extern "C"
