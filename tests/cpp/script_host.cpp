// script_host.cpp -- the reference's zangscript flow from a compiled host, through the C ABI alone: the script
// TEXT is compiled by the library's own compiler (zh_zscript_compile -> zh_zscript_generate_hip, the
// counterpart of `zangc -o scriptgen.zig`, examples/example_script.zig:6-8), the generated HIP source is built
// and loaded (zh_script_load: hiprtc inside libzang_hip.so), and module `Pluck` of
// tests/golden/script_modules.txt is painted.  The same module's generated Zig (zh_zscript_generate_zig) is
// the sequence of calls written out below with the oracle, one voice at a time, so the check is:
// fused kernel == the generated Zig's operations, bit for bit.  No Python anywhere.
// usage: script_host <script.txt>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include "zang_hip.hpp"
extern "C" {
#include "zang_oracle.h"
}

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s script.txt\n", argv[0]); return 2; }
    std::ifstream f(argv[1]);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    // front-end + HIP backend
    zh_zscript *zsc = nullptr;
    std::vector<char> err(1 << 14);
    if (zh_zscript_compile(text.c_str(), argv[1], 3, &zsc, err.data(), err.size()) != 0) { printf("FAIL compile:\n%s\n", err.data()); return 1; }
    char *hip = nullptr, *zig = nullptr;
    if (zh_zscript_generate_hip(zsc, "Pluck", 0, &hip) != 0 || zh_zscript_generate_zig(zsc, &zig) != 0) { printf("FAIL generate\n"); return 1; }
    uint32_t words = 0, noise = 0, nparams = 0;
    char name[64], merr[256];
    zh_zscript_module_info(zsc, 0, name, sizeof name, &words, &noise, &nparams, merr, sizeof merr);
    printf("compiled `%s`: %u state words/voice, %u params; generated Zig is %zu bytes, HIP %zu bytes\n", name, words, nparams, strlen(zig), strlen(hip));
    if (strcmp(name, "Pluck") != 0 || nparams != 3 || !strstr(zig, "zang.multiplyScalar(span, temps[0], temps[1], 0.25);")) { printf("FAIL metadata\n"); return 1; }
    const std::string src = hip;
    zh_zscript_free_text(hip);
    zh_zscript_free_text(zig);
    zh_zscript_destroy(zsc);
    const uint32_t V = 130, F = 512;
    const float SR = 48000.0f;
    try {
        zang::Context ctx(0);
        zh_script *script = nullptr;
        std::vector<char> log(1 << 16);
        int rc = zh_script_load(ctx.get(), src.c_str(), &script, log.data(), log.size());
        if (rc) { printf("FAIL zh_script_load (%d):\n%s\n", rc, log.data()); return 1; }
        zh_script_module *pluck = nullptr;
        zang::check(zh_script_module_create(script, "Pluck", V, words, 0, &pluck), "zh_script_module_create");

        std::vector<float> freq(V);
        std::vector<uint8_t> on(V), off(V, 0);
        for (uint32_t v = 0; v < V; v++) { freq[v] = 80.0f + 17.0f * v; on[v] = (v % 5) != 0; }
        zang::DeviceArray<float> dfreq(ctx, freq);
        zang::DeviceArray<uint8_t> don(ctx, on), doff(ctx, off);
        zang::Image out(ctx, V, F);

        // Pluck.Params = { sample_rate: f32, freq: zang.ConstantOrBuffer, note_on: bool }
        auto params = [&](const zang::DeviceArray<uint8_t> &note_on) {
            std::vector<zh_script_param> p(3);
            memset(p.data(), 0, p.size() * sizeof(zh_script_param));
            p[0].kind = ZH_SP_CONSTANT; p[0].f = SR;
            p[1].kind = ZH_SP_COB; p[1].pf = dfreq.get();           // per-voice constant
            p[2].kind = ZH_SP_BOOLEAN; p[2].pb = note_on.get();
            return p;
        };
        struct Call { uint32_t s, e; bool nic; bool note_on; };
        const Call calls[] = {{0, 300, true, true}, {300, F, false, true}, {0, F, false, false}};
        std::vector<zo_sineosc> osc(V);
        std::vector<zo_envelope> env(V);
        for (uint32_t v = 0; v < V; v++) { zo_sineosc_init(&osc[v]); zo_envelope_init(&env[v]); }
        bool ok = true;
        int k = 0;
        for (const Call &c : calls) {
            std::vector<float> base((size_t)V * F, 0.25f);
            out.upload(base);
            const zh_buf outs[1] = {out};
            auto p = params(c.note_on ? don : doff);
            zang::check(zh_script_module_paint(pluck, c.s, c.e, outs, zang::boolean(c.nic), p.data(), (uint32_t)p.size(), ZH_PAINT_ADD), "zh_script_module_paint");
            ctx.sync();
            // the generated Zig of Pluck (num_temps = 3), with the oracle
            std::vector<float> ref = base, t0(F), t1(F), t2(F);
            for (uint32_t v = 0; v < V; v++) {
                float *o = &ref[(size_t)v * F];
                zo_set(c.s, c.e, t0.data(), freq[v]);                                        // switch (params.freq) .constant => zang.set
                zo_zero(c.s, c.e, t1.data());
                zo_sineosc_paint(&osc[v], c.s, c.e, t1.data(), SR, zo_cob{1, 0.0f, t0.data()}, zo_cob{0, 0.0f, nullptr});
                zo_zero(c.s, c.e, t0.data());
                zo_multiply_scalar(c.s, c.e, t0.data(), t1.data(), 0.25f);
                for (uint32_t i = c.s; i < c.e; i++) t1[i] = 0.0f > t0[i] ? 0.0f : t0[i];   // std.math.max(0.0, temps[0][i])
                zo_zero(c.s, c.e, t0.data());
                zo_envelope_params ep = {SR, {3, 0.02f}, {2, 0.15f}, {1, 0.8f}, 0.6f, (c.note_on && on[v]) ? 1 : 0};
                zo_envelope_paint(&env[v], c.s, c.e, t0.data(), c.nic ? 1 : 0, &ep);
                zo_zero(c.s, c.e, t2.data());
                zo_multiply(c.s, c.e, t2.data(), t1.data(), t0.data());
                zo_multiply_scalar(c.s, c.e, o, t2.data(), 3.14159265358979323846f);       // outputs[0] += temps[2] * std.math.pi
            }
            const std::vector<float> got = out.download();
            const bool same = memcmp(got.data(), ref.data(), got.size() * 4) == 0;
            printf("%s paint %d [%u,%u): generated kernel vs the generated Zig's operations on the oracle\n", same ? "ok  " : "FAIL", k, c.s, c.e);
            ok &= same;
            k++;
        }
        zh_script_module_destroy(pluck);
        zh_script_destroy(script);
        printf(ok ? "PASS\n" : "FAILED\n");
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        printf("FAIL: %s\n", e.what());
        return 2;
    }
}
