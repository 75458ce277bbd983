// dispatch.hip -- ONE table of everything that selects a kernel form by voice count.
//
// Every paint entry point picks its kernel form from the voice count, the span and its arguments; the voice-count
// thresholds and the frame-range counts it uses are the rows of kForms below -- name, default, what the value selects and
// where that was measured.  A paint asks zh_form(ZF_x); nothing else in the library reads a form switch.
// One environment variable overrides rows, for A/B runs and for the parity tests that must reach every form on one box:
//     ZH_FORMS="nice_pc_max=0,sine_ranges=8"
// read ONCE, at the first paint -- unless ZH_ENV_LIVE=1 was set when the library was loaded (tests/conftest.py does: the tests
// flip rows between paints).  The rows are exported through the C ABI (zh_form_count / zh_form_info): INTEGRATION.md's list is
// generated from them (tools/gen_form_docs.py) and tests/test_gpu_dispatch.py walks them.
// Which kernels a paint really launched: zh_last_form (every launch of the library goes through ZH_LAUNCH, common.hip.h).
#include "common.hip.h"
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <string>

namespace {
struct FormRow { const char *name; long def; const char *doc; };
// `*_ranges`: -1 = the library's choice (about `waves` waves in flight up to `max` voices, zh_range_frames), 0 = never, k = k ranges.
// `*_max` / `*_min`: a voice count.
const FormRow kForms[ZF_COUNT] = {
    /* ZF_SINE_RANGES        */ {"sine_ranges", -1, "SineOsc: frame ranges of the span at once, each replaying the phase walk before it (k_sineosc_ranges); auto: ~4,096 waves up to 1 Mi voices, ~2,048 up to 65,536 with a control image"},
    /* ZF_NOISE_RANGES       */ {"noise_ranges", -1, "Noise: frame ranges from jumped generator states (noise_jump.hip); auto: by voice count up to 65,536"},
    /* ZF_ENVELOPE_RANGES    */ {"envelope_ranges", -1, "Envelope: frame ranges, each replaying the clock (k_envelope_ranges); auto: ~2,048 waves up to 40,960 voices"},
    /* ZF_SAMPLER_RANGES     */ {"sampler_ranges", -1, "Sampler: frame ranges (positions are a product, no replay); auto: ~16,384 waves up to 1 Mi voices"},
    /* ZF_DECIMATOR_RANGES   */ {"decimator_ranges", -1, "Decimator: frame ranges replaying the hold counter; auto: ~1,024-2,048 waves up to 65,536 voices"},
    /* ZF_CURVE_RANGES       */ {"curve_ranges", -1, "Curve: frame ranges; auto: ~2,048 waves up to 40,960 voices"},
    /* ZF_CYCLE_RANGES       */ {"cycle_ranges", -1, "Cycle: frame ranges (constant speed); auto: ~1,024 waves up to 16,384 voices"},
    /* ZF_PORTAMENTO_RANGES  */ {"portamento_ranges", -1, "Portamento: frame ranges; auto: ~2,048 waves up to 32,768 voices"},
    /* ZF_PULSE_CTRL_RANGES  */ {"pulse_ctrl_ranges", -1, "PulseOsc with a frequency image: frame ranges over the summed phase advances; auto: ~2,048 waves up to 40,960 voices"},
    /* ZF_PULSE_CTRL_SUMS    */ {"pulse_ctrl_sums", 1, "... 1: a first kernel sums each range's own advances (k_pulseosc_ctrl_sums), 0: every range replays the frames before it"},
    /* ZF_TRISAW_CTRL_RANGES */ {"trisaw_ctrl_ranges", -1, "TriSawOsc with a frequency image: frame ranges; auto: ~1,024 waves up to 16,384 voices"},
    /* ZF_TRISAW_CTRL_QUOT   */ {"trisaw_ctrl_quot", 1, "... 1: freq / sample_rate painted once into a module-owned image first (k_div_image), 0: divided in every replay"},
    /* ZF_PMOSC_RANGES       */ {"pmosc_ranges", -1, "PMOscInstrument: frame ranges (k_pmosc_ranges); auto: ~2,048-4,096 waves up to 131,072 voices"},
    /* ZF_SCRIPT_RANGES      */ {"script_ranges", -1, "generated script kernels: frame ranges for modules without a delay ring; auto: by voice count"},
    /* ZF_SCRIPT_RANGES_MAXV */ {"script_ranges_maxv", 131072, "... the largest voice count that takes them"},
    /* ZF_OSC_FC             */ {"osc_fc", 0, "constant-frequency PulseOsc / TriSawOsc: frames per lane of the chunked kernel; 0 = PulseOsc 4, from 16,384 voices 3 with non-temporal stores (profiles/r05/osc_large_voice_counts.txt); TriSawOsc 8-64 by voice count (tools/sweep_osc_fc.sh), PulseOsc's when the paint is a sawtooth for every voice (one color <= 0)"},
    /* ZF_NICE_PC_MAX        */ {"nice_pc_max", 65536, "NiceInstrument: up to here the oscillator, envelope and filter chains run in three waves side by side (k_nice_pc: 72 vs 146 us at 4,096 voices, 107 vs 168 at 65,536; slower at 131,072)"},
    /* ZF_NICE_PC4_MAX       */ {"nice_pc4_max", 32768, "... and up to here in four (k_nice_pc4: 44 / 47 / 53.5 us at 4,096 / 16,384 / 32,768 voices against 60.5 / 62.5 / 63)"},
    /* ZF_NICE_WAVE_MAX      */ {"nice_wave_max", 64, "zh_nice_paint_spans: up to here one WAVE per voice, lanes = frames (k_nice_spans_wave)"},
    /* ZF_PMOSC_WAVE_MAX     */ {"pmosc_wave_max", 64, "zh_pmosc_paint_spans: the same (k_pmosc_spans_wave)"},
    /* ZF_NICE_MIX_ROLL      */ {"nice_mix_roll", 1, "fused mixdown: the oscillator's half-period bit carried as a lane mask (1) or recomputed (0): 148.2 vs 149.6 us at 131,072 voices"},
    /* ZF_NICE_MIX_WG_MIN    */ {"nice_mix_wg_min", 65536, "fused mixdown: from here one partial row per 256-voice workgroup instead of per wave (a quarter of the partial traffic; 4,096 voices: 99 -> 105 us with it, 131,072: level)"},
    /* ZF_NF_PC_MAX          */ {"nf_pc_max", 65536, "Noise -> Filter voice: up to here noise and filter in two waves side by side (k_noise_filter_pc: 75 vs 110 us at 4,096 voices, 111 vs 133 at 65,536)"},
    /* ZF_NF_RING_MAX        */ {"nf_ring_max", 16384, "... white noise: up to here three producer waves, a filter wave and a writer wave per 64 voices (k_noise_filter_ring)"},
    /* ZF_FILTER_PC_MAX      */ {"filter_pc_max", 32768, "Filter, constant cutoff / resonance: up to here the three-wave pipeline over 32-frame tiles (k_filter_pc: 41-51 us against 54-76 in one wave); 0 = never (the 16-frame form too unless filter_pc16_max is given), 1 = the 16-frame form only"},
    /* ZF_FILTER_PC16_MAX    */ {"filter_pc16_max", 65536, "... 16-frame tiles above filter_pc_max up to here (36,864 / 65,536 voices: 72 / 104 us against 97 / 116)"},
    /* ZF_FILTER_PC_CTL_MAX  */ {"filter_pc_ctl_max", 32768, "... with control images (k_filter_pc_ctl: 47-98 us against 96-140 up to 32,768 voices)"},
    /* ZF_PINK_PIPE_MAX      */ {"pink_pipe_max", 32768, "pink Noise: up to here white frame ranges + the taps as a second kernel (k_pink_taps / k_pink_pipe) instead of one loop"},
    /* ZF_PINK_TAPS          */ {"pink_taps", 1, "... 1: four-wave tap kernel (32-frame tiles up to 16,384 voices, 16 above), 16: the 16-frame tiles always, 0: the seven-stage chain k_pink_pipe"},
    /* ZF_ECHOES_PC_MAX      */ {"echoes_pc_max", 65536, "FilteredEchoes: up to here ring reader / filter / writer as waves side by side (k_filtered_echoes_pc)"},
    /* ZF_DELAY_FRAMES_MAX   */ {"delay_frames_max", 4294967295, "Delay: up to here the frames of a delay length at once instead of the walk (57 -> 13 us at 4,096 voices, 423 -> 271 at 131,072: every voice count)"},
    /* ZF_FILTER_TP_MAX      */ {"filter_tp_max", 16384, "ZH_PAINT_TOLERANT Filter: the largest voice count painted as chunks at once (filter_tp.hip.h)"},
    /* ZF_NF_TP_MAX          */ {"nf_tp_max", 16384, "... Noise -> Filter voice"},
    /* ZF_NICE_TP_MAX        */ {"nice_tp_max", 16384, "... NiceInstrument and its fused mixdown"},
    /* ZF_PINK_TP_MAX        */ {"pink_tp_max", 16384, "... pink Noise"},
    /* ZF_ECHOES_TP_MAX      */ {"echoes_tp_max", 6144, "... FilteredEchoes (six image streams: level with the exact form at 8,192 voices)"},
    /* ZF_NICE_MIX_FMA       */ {"nice_mix_fma", 1, "ZH_PAINT_TOLERANT fused mixdown above nice_tp_max voices: 1 = the kernel compiled with multiply-adds fused (nice_mix_fma.hip), 0 = the exact kernel"},
    /* ZF_BASICS_ROWS_MIN    */ {"basics_rows_min", 32768, "basics.zig operations (zero, set, copy, add ..., multiply ...): from here three consecutive rows of a 256-voice column per wave (k_elementwise_chunks) instead of the row-striding loop: 131,072 voices 12-23 % faster"},
    /* ZF_SCRIPT_PC          */ {"script_pc", -1, "generated script kernels: the role-wave form (zs_paint_pc_<name>: the body's builtin calls as producer / recurrence / writer waves over LDS tiles); auto: where the emitter found the lane form unable to take frame ranges and the longest role well below the whole body; 0 = never, 1 = every module that has the form"},
    /* ZF_SCRIPT_PC_MAXV     */ {"script_pc_maxv", 65536, "... the largest voice count that takes it -- half of it for a module whose role form does more than 1.25 x the body's work or needs more than 9 waves (FilteredSawtooth: 53 / 57 / 66 / 110 us at 4,096 / 16,384 / 32,768 / 65,536 voices against 172-181 in one wave per 64 voices; slower at 131,072)"},
    /* ZF_NF_TP_PIPE_FRAMES  */ {"nf_tp_pipe_frames", 0, "ZH_PAINT_TOLERANT Noise -> Filter voice recorded pipelined (ZH_CAPTURE_COALESCE, k_nf_tp_ba): frames per chunk, a multiple of 32; 0 = as outside a pipeline (32 at 4,096 voices)"},
    /* ZF_DISTORTION_ROWS_MIN */ {"distortion_rows_min", 32768, "Distortion clip: from here four voices per lane, three consecutive rows of a 256-voice column per wave, the per-voice constants once per workgroup through LDS (k_distortion_chunks: 224 -> 178 us at 131,072 voices); the overdrive too when this row is set by hand (no faster: 239 against 244-250 us)"},
    /* ZF_DISTORTION_RC      */ {"distortion_rc", 0, "... rows per wave of that form: 3 (0), 6 or 8"},
};

struct Overrides { bool set[ZF_COUNT]; long val[ZF_COUNT]; };
Overrides parse_forms() {
    Overrides o;
    memset(&o, 0, sizeof o);
    const char *e = getenv("ZH_FORMS");
    if (!e) return o;
    std::string s(e);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string item = s.substr(i, j - i);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) {
            const std::string name = item.substr(0, eq);
            for (int k = 0; k < ZF_COUNT; k++)
                if (name == kForms[k].name) { o.set[k] = true; o.val[k] = strtol(item.c_str() + eq + 1, nullptr, 10); }
        }
        i = j + 1;
    }
    return o;
}
}   // namespace

bool zh_form_is_set(int id) {
    if (id < 0 || id >= ZF_COUNT) return false;
    return parse_forms().set[id];
}

long zh_form(int id) {
    if (id < 0 || id >= ZF_COUNT) return 0;
    static const bool live = [] { const char *e = getenv("ZH_ENV_LIVE"); return e && e[0] == '1'; }();
    if (live) {                                               // (tests: a few hundred bytes parsed per look-up)
        const Overrides o = parse_forms();
        return o.set[id] ? o.val[id] : kForms[id].def;
    }
    static const Overrides once = parse_forms();
    return once.set[id] ? once.val[id] : kForms[id].def;
}

uint32_t zh_range_frames(uint32_t V, uint32_t n, int form, uint32_t target_waves, uint32_t max_voices) {
    const long forced = zh_form(form);                                            // -1 = auto, 0 = off, k = k ranges
    if (forced == 0 || V == 0 || n < 128 || V > max_voices) return 0;
    const uint32_t waves = (V + 63) / 64;
    uint32_t want = forced > 0 ? (uint32_t)forced : target_waves / waves;
    if (want < 2) return 0;
    if (want > 64) want = 64;
    const uint32_t ch = ((n + want - 1) / want + 7) / 8 * 8;
    return (n + ch - 1) / ch >= 2 ? ch : 0;
}

thread_local const char *zh_tls_launch_detail = nullptr;

void zh_note_launch(zh_ctx *ctx, const char *kernel) {
    const char *detail = zh_tls_launch_detail;
    zh_tls_launch_detail = nullptr;
    if (!ctx) return;
    if (ctx->capturing) {                                     // the capture's own list: every launch counted, the detail kept
        const char *b0 = kernel;
        while (*b0 == '(' || *b0 == ' ') b0++;
        size_t n0 = 0;
        while (b0[n0] && b0[n0] != '<' && b0[n0] != ')' && b0[n0] != ' ') n0++;
        std::string key(b0, n0);
        if (detail) { key += '['; key += detail; key += ']'; }
        bool found = false;
        for (auto &kv : ctx->capture_kernels) if (kv.first == key) { kv.second++; found = true; break; }
        if (!found && ctx->capture_kernels.size() < 64) ctx->capture_kernels.emplace_back(key, 1u);
    }
    std::string &f = ctx->last_form;
    if (ctx->form_fresh) { f.clear(); ctx->form_fresh = false; }
    if (f.size() > 480) return;
    // "(k_nice_mix<2, true, NW>)" -> "k_nice_mix": the template arguments at the call site are names, not values
    const char *b = kernel;
    while (*b == '(' || *b == ' ') b++;
    size_t n = 0;
    while (b[n] && b[n] != '<' && b[n] != ')' && b[n] != ' ') n++;
    if (!f.empty()) {
        const size_t last = f.rfind(',');
        const std::string prev = last == std::string::npos ? f : f.substr(last + 1);
        if (prev.compare(0, std::string::npos, b, n) == 0) return;               // the same kernel again (pieces of a long span)
        f += ',';
    }
    f.append(b, n);
}

extern "C" {

int zh_form_count(void) { return ZF_COUNT; }

int zh_form_info(int index, const char **name, long *default_value, long *current_value, const char **doc) {
    if (index < 0 || index >= ZF_COUNT) return ZH_ERR_INVALID;
    if (name) *name = kForms[index].name;
    if (default_value) *default_value = kForms[index].def;
    if (current_value) *current_value = zh_form(index);
    if (doc) *doc = kForms[index].doc;
    return ZH_OK;
}

int zh_last_form(zh_ctx *ctx, char *out, size_t n) {
    if (!ctx || !out || n == 0) return ZH_ERR_INVALID;
    const std::string &f = ctx->last_form;
    const size_t k = f.size() < n - 1 ? f.size() : n - 1;
    memcpy(out, f.data(), k);
    out[k] = 0;
    return ZH_OK;
}

}  // extern "C"
