// harness.cpp -- random sequences through the C ABI against libzang_hip built host-only with ASan + UBSan over hip_stub.cpp
// (tools/host_asan.sh): begin_capture (with and without ZH_CAPTURE_COALESCE), paints of flipper modules (PulseOsc / TriSawOsc in
// table form, SineOsc as frame ranges), held-back batches (oscillators, stereo mixdowns), the pipelined Noise -> Filter recording,
// eager paints between replays, end / launch / destroy in every order -- modules before their graphs, the context before its graphs.
// No kernel runs: what is checked is the host bookkeeping (ctx.hip, osc.hip, composite.hip, basics.hip) for memory errors, and at
// the end that no device allocation, stream or graph is left.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "zang_hip.h"
extern "C" void hip_stub_report(unsigned long *, unsigned long *, unsigned long *, unsigned long *, unsigned long *);

static unsigned long long rs = 88172645463325252ull;
static unsigned rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 11); }
static unsigned pick(unsigned n) { return rnd() % n; }
#define OK(x) do { int rc_ = (x); if (rc_ != ZH_OK) { fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, rc_); exit(1); } } while (0)

enum Kind { PULSE, TRISAW, SINE, NF, NICE, NKIND };
struct Mod { Kind kind; void *h; unsigned V; float *freq, *color; bool painted; };
struct Ctx {
    zh_ctx *c;
    std::vector<Mod> mods;
    std::vector<zh_graph *> graphs;
    std::vector<zh_buf> imgs;            // [1024][V] images by voice count (made on demand)
    float *mix;                          // 64 rows of 1,024 floats
    bool capturing; unsigned flags;
};
static const unsigned kVoices[] = {64, 300, 4096, 20000};
static zh_f32 f32c(float v) { zh_f32 x; memset(&x, 0, sizeof x); x.value = v; return x; }
static zh_f32 f32p(const float *p) { zh_f32 x; memset(&x, 0, sizeof x); x.per_voice = p; return x; }
static zh_bool boolc(unsigned v) { zh_bool x; memset(&x, 0, sizeof x); x.value = v; return x; }
static zh_cob cobc(zh_f32 c) { zh_cob x; memset(&x, 0, sizeof x); x.tag = ZH_COB_CONSTANT; x.constant = c; return x; }

static zh_buf image(Ctx &cx, unsigned V, unsigned which) {
    for (unsigned tries = 0;; tries++) {
        unsigned n = 0;
        for (zh_buf &b : cx.imgs) if (b.voices == V && n++ == which) return b;
        zh_buf b;
        OK(zh_buf_alloc(cx.c, &b, V, 1024));
        cx.imgs.push_back(b);
    }
}
static void add_module(Ctx &cx) {
    Mod m; m.kind = (Kind)pick(NKIND); m.V = kVoices[pick(4)]; m.h = nullptr; m.painted = false;
    OK(zh_malloc(cx.c, (void **)&m.freq, (size_t)m.V * 4)); OK(zh_malloc(cx.c, (void **)&m.color, (size_t)m.V * 4));
    switch (m.kind) {
    case PULSE: OK(zh_pulseosc_create(cx.c, m.V, (zh_pulseosc **)&m.h)); break;
    case TRISAW: OK(zh_trisawosc_create(cx.c, m.V, (zh_trisawosc **)&m.h)); break;
    case SINE: OK(zh_sineosc_create(cx.c, m.V, (zh_sineosc **)&m.h)); break;
    case NF: OK(zh_noise_filter_create(cx.c, m.V, 7, (zh_noise_filter **)&m.h)); break;
    default: OK(zh_nice_create(cx.c, m.V, f32p(m.color), (zh_nice **)&m.h)); break;
    }
    cx.mods.push_back(m);
}
static void drop_module(Ctx &cx, size_t i) {
    Mod &m = cx.mods[i];
    switch (m.kind) {
    case PULSE: OK(zh_pulseosc_destroy((zh_pulseosc *)m.h)); break;
    case TRISAW: OK(zh_trisawosc_destroy((zh_trisawosc *)m.h)); break;
    case SINE: OK(zh_sineosc_destroy((zh_sineosc *)m.h)); break;
    case NF: OK(zh_noise_filter_destroy((zh_noise_filter *)m.h)); break;
    default: OK(zh_nice_destroy((zh_nice *)m.h)); break;
    }
    OK(zh_free(cx.c, m.freq)); OK(zh_free(cx.c, m.color));
    cx.mods.erase(cx.mods.begin() + (long)i);
}
// one paint; inside a capture the return code may be ZH_ERR_UNSUPPORTED (a scratch that cannot grow while recording): allowed
struct Fixed { bool on; unsigned s, e, zf, tol, nic; };
static void paint(Ctx &cx, Mod &m, const Fixed *fx = nullptr) {
    const bool whole = pick(3) != 0;
    unsigned s = whole ? 0 : pick(900), e = whole ? 1024 : s + pick(1024 - s) + 1;
    unsigned flags = (pick(2) ? ZH_PAINT_ZERO_FIRST : 0);
    if (fx) { s = fx->s; e = fx->e; flags = fx->zf ? ZH_PAINT_ZERO_FIRST : 0; }
    static unsigned next_img = 0;
    zh_buf out = image(cx, m.V, cx.capturing ? (pick(5) ? next_img++ % 12 : pick(12)) : pick(3));   // mostly a ring of images (batches form), now and then one again
    int rc = ZH_OK;
    switch (m.kind) {
    case PULSE: case TRISAW: {
        if (m.painted && (fx || pick(4))) flags |= ZH_PAINT_PARAMS_UNCHANGED;
        if (m.kind == PULSE) { zh_pulseosc_params p; memset(&p, 0, sizeof p); p.sample_rate = 48000.0f; p.freq = cobc(f32p(m.freq)); p.color = f32p(m.color);
            rc = zh_pulseosc_paint((zh_pulseosc *)m.h, s, e, &out, nullptr, boolc(0), &p, flags); }
        else { zh_trisawosc_params p; memset(&p, 0, sizeof p); p.sample_rate = 48000.0f; p.freq = cobc(f32p(m.freq)); p.color = f32p(m.color);
            rc = zh_trisawosc_paint((zh_trisawosc *)m.h, s, e, &out, nullptr, boolc(0), &p, flags); }
        break; }
    case SINE: { zh_sineosc_params p; memset(&p, 0, sizeof p); p.sample_rate = 48000.0f; p.freq = cobc(f32p(m.freq)); p.phase = cobc(f32c(0.25f));
        if (pick(2)) flags |= ZH_PAINT_TOLERANT;
        rc = zh_sineosc_paint((zh_sineosc *)m.h, s, e, &out, nullptr, boolc(0), &p, flags); break; }
    case NF: { zh_noise_filter_params p; memset(&p, 0, sizeof p); p.color = ZH_NOISE_WHITE; p.type = 1 + pick(5); p.cutoff = f32p(m.freq); p.res = f32c(0.5f);
        if (fx ? fx->tol : pick(3)) flags |= ZH_PAINT_TOLERANT;
        rc = zh_noise_filter_paint((zh_noise_filter *)m.h, s, e, &out, nullptr, boolc(0), &p, flags); break; }
    default: { zh_nice_params p; memset(&p, 0, sizeof p); p.sample_rate = 48000.0f; p.freq = f32p(m.freq); p.note_on = boolc(pick(2));
        static unsigned next_row = 0;
        const unsigned row = fx ? next_row++ % 31 : pick(31);
        rc = zh_nice_paint_mix_stereo((zh_nice *)m.h, s, e, cx.mix + (size_t)row * 2048, cx.mix + (size_t)row * 2048 + 1024, f32p(m.color), (fx ? fx->nic : pick(2)) ? f32p(m.freq) : f32p(m.color),
                                      boolc(pick(2)), &p, flags | ((fx ? fx->tol : !pick(4)) ? ZH_PAINT_TOLERANT : 0u)); break; }
    }
    if (rc != ZH_OK && !(cx.capturing && rc == ZH_ERR_UNSUPPORTED)) { fprintf(stderr, "paint kind %d V %u span (%u, %u) flags %u -> %d\n", (int)m.kind, m.V, s, e, flags, rc); exit(1); }
    m.painted = rc == ZH_OK;
}

int main(int argc, char **argv) {
    const unsigned rounds = argc > 1 ? (unsigned)atoi(argv[1]) : 300, seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    rs ^= (unsigned long long)seed * 0x9e3779b97f4a7c15ull;
    unsigned long paints = 0, captures = 0, launches = 0, refused = 0, held = 0, held_launches = 0;
    for (unsigned round = 0; round < rounds; round++) {
        Ctx cx; cx.capturing = false; cx.flags = 0;
        OK(zh_create(&cx.c, 0));
        OK(zh_malloc(cx.c, (void **)&cx.mix, 64 * 2048 * 4));
        for (unsigned k = 0, n = 1 + pick(4); k < n; k++) add_module(cx);
        for (unsigned step = 0, nsteps = 20 + pick(60); step < nsteps; step++) {
            const unsigned op = pick(100);
            if (cx.capturing) {
                if (op < 70 && !cx.mods.empty()) {                       // a run of paints, mostly of one module (batches form)
                    Mod &m = cx.mods[pick((unsigned)cx.mods.size())];
                    // half of the runs: the same span and flags paint after paint, as a host loop over buffers makes them (batches / the pipeline form)
                    Fixed fx{true, 0, 1024, pick(2), pick(3) != 0, pick(2)};
                    if (!pick(3)) { fx.s = pick(400); fx.e = fx.s + 128 + pick(400); }
                    const bool burst = pick(2) != 0;
                    for (unsigned k = 0, n = 1 + pick(40); k < n; k++) { paint(cx, pick(12) ? m : cx.mods[pick((unsigned)cx.mods.size())], burst ? &fx : nullptr); paints++; }
                } else if (op < 80 && !cx.imgs.empty()) {
                    OK(zh_zero(cx.c, 0, 1024, cx.imgs[pick((unsigned)cx.imgs.size())]));     // another library call between paints: ends the epoch
                } else {
                    zh_graph *g = nullptr;
                    const int rc = zh_graph_end_capture(cx.c, &g);
                    cx.capturing = false;
                    if (rc == ZH_OK) { cx.graphs.push_back(g); unsigned a, b, c2; OK(zh_graph_info(g, &a, &b, &c2)); held += b; held_launches += c2; char kn[512]; OK(zh_graph_kernels(g, kn, sizeof kn)); }
                    else if (rc == ZH_ERR_UNSUPPORTED) refused++;        // a held-back paint that could not be launched after all (deferred error)
                    else { fprintf(stderr, "zh_graph_end_capture -> %d\n", rc); return 1; }
                }
                continue;
            }
            if (op < 25 && !cx.mods.empty()) { paint(cx, cx.mods[pick((unsigned)cx.mods.size())]); paints++; }
            else if (op < 40) { cx.flags = pick(3) ? ZH_CAPTURE_COALESCE : 0; OK(zh_graph_begin_capture_flags(cx.c, cx.flags)); cx.capturing = true; captures++; }
            else if (op < 65 && !cx.graphs.empty()) {
                const int rc = zh_graph_launch(cx.c, cx.graphs[pick((unsigned)cx.graphs.size())]);
                if (rc != ZH_OK && rc != ZH_ERR_INVALID) { fprintf(stderr, "zh_graph_launch -> %d\n", rc); return 1; }   // INVALID: a module of the capture was destroyed since
                launches++;
            }
            else if (op < 72 && !cx.graphs.empty()) { const size_t i = pick((unsigned)cx.graphs.size()); OK(zh_graph_destroy(cx.graphs[i])); cx.graphs.erase(cx.graphs.begin() + (long)i); }
            else if (op < 80 && cx.mods.size() < 6) add_module(cx);
            else if (op < 86 && !cx.mods.empty()) drop_module(cx, pick((unsigned)cx.mods.size()));
            else if (op < 92 && !cx.mods.empty()) {                      // state round trip of a flipper between paints and replays
                Mod &m = cx.mods[pick((unsigned)cx.mods.size())];
                if (m.kind == PULSE) { std::vector<zh_pulseosc_state> st(m.V); OK(zh_pulseosc_get_state((zh_pulseosc *)m.h, st.data())); OK(zh_pulseosc_set_state((zh_pulseosc *)m.h, st.data())); }
            }
            else OK(zh_sync(cx.c));
        }
        if (cx.capturing) { zh_graph *g = nullptr; if (zh_graph_end_capture(cx.c, &g) == ZH_OK) cx.graphs.push_back(g); cx.capturing = false; }
        // teardown: modules, then -- half of the time -- the context BEFORE its graphs
        while (!cx.mods.empty()) drop_module(cx, pick((unsigned)cx.mods.size()));
        for (zh_buf &b : cx.imgs) OK(zh_buf_free(cx.c, &b));
        OK(zh_free(cx.c, cx.mix));
        const bool ctx_first = pick(2) != 0;
        if (!ctx_first) for (zh_graph *g : cx.graphs) OK(zh_graph_destroy(g));
        OK(zh_destroy(cx.c));
        if (ctx_first) for (zh_graph *g : cx.graphs) OK(zh_graph_destroy(g));
    }
    unsigned long kl, kc, blocks, graphs, streams;
    hip_stub_report(&kl, &kc, &blocks, &graphs, &streams);
    printf("%u contexts, %lu captures, %lu paints (%lu held back -> %lu launches), %lu replays, %lu captures refused for a deferred error; %lu kernel launches eager + %lu recorded; "
           "left over: %lu device blocks, %lu graphs, %lu streams\n", rounds, captures, paints, held, held_launches, launches, refused, kl, kc, blocks, graphs, streams);
    return (blocks || graphs || streams) ? 2 : 0;
}
