// host_parity.cpp -- the C++ host API (include/zang_hip.hpp) used the way examples/modules.zig uses zang:
// NiceInstrument.paint (examples/modules.zig:212-247) written out module by module through device temps,
// for a batch of voices, checked bit for bit against (a) the oracle's scalar restatement of the same
// recipe, one voice at a time, and (b) the fused zh_nice kernel.  Test infrastructure (links the oracle).
// Build: g++ -std=c++17 -Iinclude -Ioracle tests/cpp/host_parity.cpp -Lzang_amd -lzang_hip oracle/libzang_oracle.so
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "zang_hip.hpp"
extern "C" {
#include "zang_oracle.h"
}

// the reference's NiceInstrument (examples/modules.zig:189-248), batch form
struct NiceInstrument {
    static constexpr size_t num_outputs = 1;
    static constexpr size_t num_temps = 2;
    struct Params {
        float sample_rate;
        const zang::DeviceArray<float> &freq;      // per voice
        bool note_on;
    };

    zang::Context &ctx;
    const zang::DeviceArray<float> &color;          // init(color), per voice
    mod::PulseOsc osc;
    mod::Filter flt;
    mod::Envelope env;
    zang::DeviceArray<float> cutoff_freq, cutoff;   // scratch for cutoffFromFrequency(freq * 8)

    NiceInstrument(zang::Context &c, uint32_t n, const zang::DeviceArray<float> &color_, const std::vector<float> &freq8)
        : ctx(c), color(color_), osc(c, n), flt(c, n), env(c, n), cutoff_freq(c, freq8), cutoff(c, std::vector<float>(n, 0.0f)) {}

    void paint(zang::Span span, std::array<zh_buf, 1> outputs, std::array<zh_buf, 2> temps, bool note_id_changed, const Params &params) {
        const uint32_t n = (uint32_t)color.size();
        zang::zero(ctx, span, temps[0]);                                                   // :220
        osc.paint(span, {temps[0]}, {}, zang::boolean(note_id_changed),
                  {params.sample_rate, 0, zang::constant(params.freq), zang::f32(color)});   // :221-225
        zang::multiplyWithScalar(ctx, span, temps[0], 0.5f);                               // :226
        zang::zero(ctx, span, temps[1]);                                                   // :227
        mod::cutoffFromFrequency(ctx, n, const_cast<float *>(cutoff.get()), cutoff_freq.get(), params.sample_rate);   // :231-234
        flt.paint(span, {temps[1]}, {}, zang::boolean(note_id_changed),
                  {temps[0], ZH_FILTER_LOW_PASS, 0, zang::constant(cutoff), zang::constant(0.7f)});                   // :228-236
        zang::zero(ctx, span, temps[0]);                                                   // :237
        env.paint(span, {temps[0]}, {}, zang::boolean(note_id_changed),
                  {params.sample_rate, 0, zang::PaintCurve::cubed(0.01f), zang::PaintCurve::cubed(0.1f), zang::PaintCurve::cubed(0.5f),
                   zang::f32(0.8f), zang::boolean(params.note_on)});                       // :238-245
        zang::multiply(ctx, span, outputs[0], temps[0], temps[1]);                         // :246
    }
};

static bool same_bits(const std::vector<float> &a, const std::vector<float> &b, const char *what) {
    if (a.size() != b.size() || memcmp(a.data(), b.data(), a.size() * 4) != 0) {
        size_t bad = 0, first = a.size();
        for (size_t i = 0; i < a.size() && i < b.size(); i++)
            if (memcmp(&a[i], &b[i], 4)) { if (first == a.size()) first = i; bad++; }
        printf("FAIL %s: %zu differing samples, first at %zu (%g vs %g)\n", what, bad, first, a[first], b[first]);
        return false;
    }
    printf("ok   %s (%zu samples, bit-exact)\n", what, a.size());
    return true;
}

int main() {
    const uint32_t V = 300, F = 1024;
    const float SR = 48000.0f;
    std::vector<float> freq(V), color(V), freq8(V);
    for (uint32_t v = 0; v < V; v++) {
        freq[v] = 55.0f * powf(2.0f, 6.0f * (float)v / V);
        color[v] = 0.1f + 0.8f * (float)((v * 37) % V) / V;
        freq8[v] = freq[v] * 8.0f;
    }
    freq[0] = 7000.0f; freq8[0] = freq[0] * 8.0f;        // silent voice (freq > sr/8)
    bool ok = true;
    try {
        zang::Context ctx(0);
        zang::DeviceArray<float> dfreq(ctx, freq), dcolor(ctx, color);
        zang::Image out(ctx, V, F), t0(ctx, V, F), t1(ctx, V, F), fused_out(ctx, V, F);
        NiceInstrument inst(ctx, V, dcolor, freq8);
        mod::NiceInstrument fused(ctx, V, zang::f32(dcolor));

        // the paint sequence: note on (new note), continue in two sub-spans, note off
        struct Call { zang::Span span; bool note_on, nic; };
        const Call calls[] = {{{0, F}, true, true}, {{0, 400}, true, false}, {{400, F}, false, false}, {{0, F}, false, false}};
        std::vector<zo_nice_instrument> ref_state(V);
        for (uint32_t v = 0; v < V; v++) zo_nice_init(&ref_state[v], color[v]);
        std::vector<float> rt0(F), rt1(F);
        int k = 0;
        for (const Call &c : calls) {
            std::vector<float> base((size_t)V * F);
            for (size_t i = 0; i < base.size(); i++) base[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.0f;   // `+=` onto existing content
            out.upload(base);
            fused_out.upload(base);
            inst.paint(c.span, {out}, {t0, t1}, c.nic, {SR, dfreq, c.note_on});
            fused.paint(c.span, {fused_out}, {t0, t1}, zang::boolean(c.nic), {SR, 0, zang::f32(dfreq), zang::boolean(c.note_on)});
            ctx.sync();
            std::vector<float> ref = base;
            for (uint32_t v = 0; v < V; v++)
                zo_nice_paint(&ref_state[v], c.span.start, c.span.end, &ref[(size_t)v * F], rt0.data(), rt1.data(), c.nic ? 1 : 0, SR, freq[v], c.note_on ? 1 : 0);
            char what[96];
            snprintf(what, sizeof what, "paint %d [%u,%u): modules through temps (C++ host API) vs oracle", k, c.span.start, c.span.end);
            ok &= same_bits(out.download(), ref, what);
            snprintf(what, sizeof what, "paint %d [%u,%u): fused zh_nice vs oracle", k, c.span.start, c.span.end);
            ok &= same_bits(fused_out.download(), ref, what);
            k++;
        }
        // error behaviour: the Zig paint cannot fail; the C ABI rejects a span beyond the image and the wrapper throws
        try {
            fused.paint({0, F + 1}, {fused_out}, {t0, t1}, zang::boolean(false), {SR, 0, zang::f32(dfreq), zang::boolean(true)});
            printf("FAIL: span beyond the image was accepted\n");
            ok = false;
        } catch (const zang::Error &e) {
            printf("ok   out-of-range span rejected: %s\n", e.what());
        }
        // the opt-in tolerant form from C++: flags = ZH_PAINT_ZERO_FIRST | ZH_PAINT_TOLERANT (include/zang_hip.h): north_star's 1e-5,
        // here measured against the oracle relative to each voice's peak over the span
        {
            mod::NiceInstrument tol(ctx, V, zang::f32(dcolor));
            std::vector<zo_nice_instrument> st(V);
            for (uint32_t v = 0; v < V; v++) zo_nice_init(&st[v], color[v]);
            double worst = 0.0;
            const Call two[] = {{{0, F}, true, true}, {{0, F}, false, false}};
            for (const Call &c : two) {
                tol.paint(c.span, {fused_out}, {t0, t1}, zang::boolean(c.nic), {SR, 0, zang::f32(dfreq), zang::boolean(c.note_on)}, ZH_PAINT_ZERO_FIRST | ZH_PAINT_TOLERANT);
                ctx.sync();
                const std::vector<float> got = fused_out.download();
                std::vector<float> ref((size_t)V * F, 0.0f);
                for (uint32_t v = 0; v < V; v++) {
                    zo_nice_paint(&st[v], 0, F, &ref[(size_t)v * F], rt0.data(), rt1.data(), c.nic ? 1 : 0, SR, freq[v], c.note_on ? 1 : 0);
                    double peak = fmax(fabs((double)st[v].flt.l), fabs((double)st[v].flt.b)), err = 0.0;
                    for (uint32_t f = 0; f < F; f++) {
                        peak = fmax(peak, fabs((double)ref[(size_t)v * F + f]));
                        err = fmax(err, fabs((double)got[(size_t)v * F + f] - (double)ref[(size_t)v * F + f]));
                    }
                    if (peak > 0.0) worst = fmax(worst, err / peak);
                }
            }
            if (worst <= 1e-5) printf("ok   zh_nice_paint with ZH_PAINT_TOLERANT: worst sample %.2e of its voice's peak (allowed 1e-5)\n", worst);
            else { printf("FAIL: tolerant paint off by %.2e of the peak\n", worst); ok = false; }
        }
        // the mixdown from C++: four buffers as separate calls and as one batch launch, then the exchange step through a one-rank
        // RCCL communicator (the sum over one rank is the identity): same bits everywhere
        {
            const uint32_t B = 4;
            mod::NiceInstrument ma(ctx, V, zang::f32(dcolor)), mb(ctx, V, zang::f32(dcolor));
            zang::DeviceArray<float> la(ctx, std::vector<float>((size_t)B * F, 0.0f)), ra(ctx, std::vector<float>((size_t)B * F, 0.0f));
            zang::DeviceArray<float> lb(ctx, std::vector<float>((size_t)B * F, 0.0f)), rb(ctx, std::vector<float>((size_t)B * F, 0.0f));
            std::vector<mod::NiceInstrument::Params> ps;
            std::vector<zh_bool> nics;
            std::vector<float *> pl, pr;
            for (uint32_t k = 0; k < B; k++) {
                ps.push_back({SR, 0, zang::f32(dfreq), zang::boolean(k < 2)});
                nics.push_back(zang::boolean(k == 0));
                pl.push_back(lb.get() + (size_t)k * F); pr.push_back(rb.get() + (size_t)k * F);
                mod::paintMixStereo(ma, {0, F}, la.get() + (size_t)k * F, ra.get() + (size_t)k * F, zang::f32(0.75f), zang::f32(0.25f), nics[k], ps[k], ZH_PAINT_ZERO_FIRST);
            }
            mod::paintMixStereoBatch(mb, {0, F}, pl, pr, zang::f32(0.75f), zang::f32(0.25f), nics, ps, ZH_PAINT_ZERO_FIRST);
            ctx.sync();
            ok &= same_bits(lb.download(), la.download(), "stereo mixdown, 4 buffers in one launch vs 4 launches (left)");
            ok &= same_bits(rb.download(), ra.download(), "stereo mixdown, 4 buffers in one launch vs 4 launches (right)");
            if (zang::Comm::available()) {
                zang::Comm comm(ctx, 1, 0, zang::Comm::uniqueId());
                comm.allreduceMix(lb.get(), (size_t)B * F);
                comm.reduceMix(rb.get(), (size_t)B * F, 0);
                ctx.sync();
                ok &= same_bits(lb.download(), la.download(), "zh_allreduce_mix over a one-rank RCCL communicator (left block)");
                ok &= same_bits(rb.download(), ra.download(), "zh_reduce_mix over a one-rank RCCL communicator (right block)");
            } else {
                printf("FAIL: librccl not available: %s\n", zh_comm_last_error());
                ok = false;
            }
        }
    } catch (const std::exception &e) {
        printf("FAIL: %s\n", e.what());
        return 2;
    }
    printf(ok ? "PASS\n" : "FAILED\n");
    return ok ? 0 : 1;
}
