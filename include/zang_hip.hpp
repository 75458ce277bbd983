// zang_hip.hpp -- C++17 host-side mirror of zang's `zang` and `modules` namespaces over the C ABI of
// include/zang_hip.h (header-only; link with -lzang_hip).
//
// The reference is Zig and its toolchain is absent from the build image, so the compiled-language host
// side above the C ABI is this header: the same names, argument order and meaning as the Zig code
//     module.paint(span, outputs, temps, note_id_changed, params)          (src/modules/SineOsc.zig:24-31)
//     zang.zero(span, buf) / zang.multiply(span, dest, a, b) / ...          (src/zang/basics.zig:12-78)
// with one difference of kind: a module object is a BATCH of n voices on the GPU and a buffer is a device
// image [frame][voice] (zang::Image) instead of a host []f32.  Errors: the Zig paint functions cannot fail;
// here a non-zero return of the C ABI throws zang::Error.  bindings/zang_hip.zig is the same thing for a
// Zig host; zang_amd/zang.py + modules.py for Python.  tests/cpp/host_parity.cpp uses this header the way
// examples/modules.zig uses zang.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "zang_hip.h"

namespace zang {

struct Error : std::runtime_error {
    int code;
    Error(int rc, const std::string &what) : std::runtime_error(what + ": " + zh_error_string(rc) + " (" + std::to_string(rc) + ")"), code(rc) {}
};
inline void check(int rc, const char *what) {
    if (rc != 0) throw Error(rc, what);
}

struct Span {                                   // src/zang/basics.zig:3-10
    uint32_t start, end;
    static Span init(uint32_t start, uint32_t end) { return Span{start, end}; }
};

class Context {
    zh_ctx *h_ = nullptr;

public:
    explicit Context(int device = 0) { check(zh_create(&h_, device), "zh_create"); }
    ~Context() { if (h_) zh_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    zh_ctx *get() const { return h_; }
    void sync() { check(zh_sync(h_), "zh_sync"); }
};

// The multi-GPU exchange as a collective (zh_comm_*: RCCL over xGMI, one rank per process, enqueued on the context's
// stream).  `id` = the 128 bytes rank 0 obtained from Comm::uniqueId() and handed to the other processes over any host
// channel; the constructor blocks until all `world` ranks have called it.
class Comm {
    zh_comm *h_ = nullptr;

public:
    using Id = std::array<uint8_t, ZH_COMM_ID_BYTES>;
    static bool available() { return zh_comm_available() == 1; }
    static Id uniqueId() { Id id{}; check(zh_comm_unique_id(id.data()), "zh_comm_unique_id"); return id; }
    Comm(Context &c, uint32_t world, uint32_t rank, const Id &id) { check(zh_comm_create(c.get(), world, rank, id.data(), &h_), "zh_comm_create"); }
    ~Comm() { if (h_) zh_comm_destroy(h_); }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    zh_comm *get() const { return h_; }
    void allreduceMix(float *mix_dev, size_t n) { check(zh_allreduce_mix(h_, mix_dev, n), "zh_allreduce_mix"); }
    void reduceMix(float *mix_dev, size_t n, uint32_t root) { check(zh_reduce_mix(h_, mix_dev, n, root), "zh_reduce_mix"); }
};

// A device sample image [frame][voice]: column v is what the reference calls one `[]f32` of voice v.
class Image {
    Context &ctx_;
    zh_buf b_{};

public:
    Image(Context &ctx, uint32_t voices, uint32_t frames) : ctx_(ctx) { check(zh_buf_alloc(ctx.get(), &b_, voices, frames), "zh_buf_alloc"); }
    ~Image() { zh_buf_free(ctx_.get(), &b_); }
    Image(const Image &) = delete;
    Image &operator=(const Image &) = delete;
    operator zh_buf() const { return b_; }
    uint32_t voices() const { return b_.voices; }
    uint32_t frames() const { return b_.frames; }
    // host layout: one contiguous []f32 per voice ([voice][frame])
    void upload(const std::vector<float> &voice_major) { check(zh_buf_upload_voices(ctx_.get(), b_, voice_major.data(), b_.frames), "zh_buf_upload_voices"); }
    std::vector<float> download() const {
        std::vector<float> out((size_t)b_.voices * b_.frames);
        check(zh_buf_download_voices(ctx_.get(), out.data(), b_, b_.frames), "zh_buf_download_voices");
        return out;
    }
};

// A device array of per-voice parameters (f32 or bool-as-u8).
template <class T> class DeviceArray {
    Context &ctx_;
    T *p_ = nullptr;
    size_t n_;

public:
    DeviceArray(Context &ctx, const std::vector<T> &host) : ctx_(ctx), n_(host.size()) {
        check(zh_malloc(ctx.get(), reinterpret_cast<void **>(&p_), n_ * sizeof(T)), "zh_malloc");
        check(zh_upload(ctx.get(), p_, host.data(), n_ * sizeof(T)), "zh_upload");
    }
    ~DeviceArray() { zh_free(ctx_.get(), p_); }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    const T *get() const { return p_; }
    T *get() { return p_; }
    size_t size() const { return n_; }
    std::vector<T> download() const {
        std::vector<T> host(n_);
        check(zh_download(ctx_.get(), host.data(), p_, n_ * sizeof(T)), "zh_download");
        return host;
    }
};

// ---- values
inline zh_f32 f32(float v) { return zh_f32{v, 0, nullptr}; }
inline zh_f32 f32(const DeviceArray<float> &per_voice) { return zh_f32{0.0f, 0, per_voice.get()}; }
inline zh_bool boolean(bool v) { return zh_bool{v ? 1u : 0u, 0, nullptr}; }
inline zh_bool boolean(const DeviceArray<uint8_t> &per_voice) { return zh_bool{0, 0, per_voice.get()}; }

// zang.constant / zang.buffer (src/zang/constant_or_buffer.zig:9-15)
inline zh_cob constant(float v) { return zh_cob{ZH_COB_CONSTANT, 0, f32(v), zh_buf{}}; }
inline zh_cob constant(const DeviceArray<float> &per_voice) { return zh_cob{ZH_COB_CONSTANT, 0, f32(per_voice), zh_buf{}}; }
inline zh_cob buffer(const zh_buf &b) { return zh_cob{ZH_COB_BUFFER, 0, zh_f32{}, b}; }

// zang.PaintCurve (src/zang/painter.zig:25-30)
struct PaintCurve {
    static zh_curve instantaneous() { return zh_curve{ZH_CURVE_INSTANTANEOUS, 0, f32(0.0f)}; }
    static zh_curve linear(float d) { return zh_curve{ZH_CURVE_LINEAR, 0, f32(d)}; }
    static zh_curve squared(float d) { return zh_curve{ZH_CURVE_SQUARED, 0, f32(d)}; }
    static zh_curve cubed(float d) { return zh_curve{ZH_CURVE_CUBED, 0, f32(d)}; }
};

// ---- basics.zig:12-78 (same names, `span` first)
inline void zero(Context &c, Span s, zh_buf dest) { check(zh_zero(c.get(), s.start, s.end, dest), "zero"); }
inline void set(Context &c, Span s, zh_buf dest, float a) { check(zh_set(c.get(), s.start, s.end, dest, f32(a)), "set"); }
inline void copy(Context &c, Span s, zh_buf dest, zh_buf src) { check(zh_copy(c.get(), s.start, s.end, dest, src), "copy"); }
inline void add(Context &c, Span s, zh_buf dest, zh_buf a, zh_buf b) { check(zh_add(c.get(), s.start, s.end, dest, a, b), "add"); }
inline void addInto(Context &c, Span s, zh_buf dest, zh_buf src) { check(zh_add_into(c.get(), s.start, s.end, dest, src), "addInto"); }
inline void addScalar(Context &c, Span s, zh_buf dest, zh_buf a, float b) { check(zh_add_scalar(c.get(), s.start, s.end, dest, a, f32(b)), "addScalar"); }
inline void addScalarInto(Context &c, Span s, zh_buf dest, float a) { check(zh_add_scalar_into(c.get(), s.start, s.end, dest, f32(a)), "addScalarInto"); }
inline void multiply(Context &c, Span s, zh_buf dest, zh_buf a, zh_buf b) { check(zh_multiply(c.get(), s.start, s.end, dest, a, b), "multiply"); }
inline void multiplyWith(Context &c, Span s, zh_buf dest, zh_buf a) { check(zh_multiply_with(c.get(), s.start, s.end, dest, a), "multiplyWith"); }
inline void multiplyScalar(Context &c, Span s, zh_buf dest, zh_buf a, float b) { check(zh_multiply_scalar(c.get(), s.start, s.end, dest, a, f32(b)), "multiplyScalar"); }
inline void multiplyWithScalar(Context &c, Span s, zh_buf dest, float a) { check(zh_multiply_with_scalar(c.get(), s.start, s.end, dest, f32(a)), "multiplyWithScalar"); }

// the sum of all voices of an image: dst_dev[f] (+)= sum_v src[f][v] (V x zang.addInto, basics.zig:31-36; fixed-order tree sum)
inline void mixdownVoices(Context &c, Span s, float *dst_dev, zh_buf src, uint32_t flags = ZH_PAINT_ADD) {
    check(zh_mixdown_voices(c.get(), s.start, s.end, dst_dev, src, flags), "mixdownVoices");
}
// zang.mixDown (src/zang/mixdown.zig:8-86): f32 mix -> interleaved s8 / s16 LE PCM with clamping, on the device
inline void mixDown(Context &c, uint8_t *dst_dev, const float *mix_dev, uint32_t n, uint32_t audio_format, uint32_t num_channels,
                    uint32_t channel_index, float vol) {
    check(zh_mix_down(c.get(), dst_dev, mix_dev, n, audio_format, num_channels, channel_index, vol), "mixDown");
}

}  // namespace zang

namespace mod {

// One class per module: `num_outputs`, `num_temps`, `Params` (the C ABI's params struct, fields in the Zig
// declaration order) and paint(span, outputs, temps, note_id_changed, params) like the Zig declarations.
#define ZANG_HIP_MODULE(Name, prefix, NT, CREATE_ARGS_DECL, CREATE_ARGS_USE)                                            \
    class Name {                                                                                                          \
        zh_##prefix *h_ = nullptr;                                                                                        \
                                                                                                                          \
    public:                                                                                                               \
        static constexpr size_t num_outputs = 1;                                                                          \
        static constexpr size_t num_temps = NT;                                                                           \
        using Params = zh_##prefix##_params;                                                                              \
        Name(zang::Context &ctx, uint32_t n_voices CREATE_ARGS_DECL) {                                                    \
            zang::check(zh_##prefix##_create(ctx.get(), n_voices CREATE_ARGS_USE, &h_), "zh_" #prefix "_create");         \
        }                                                                                                                 \
        ~Name() { if (h_) zh_##prefix##_destroy(h_); }                                                                    \
        Name(const Name &) = delete;                                                                                      \
        Name &operator=(const Name &) = delete;                                                                           \
        zh_##prefix *get() const { return h_; }                                                                           \
        void paint(zang::Span span, const std::array<zh_buf, 1> &outputs, const std::array<zh_buf, NT> &temps,            \
                   zh_bool note_id_changed, const Params &params, uint32_t flags = ZH_PAINT_ADD) {                        \
            zang::check(zh_##prefix##_paint(h_, span.start, span.end, outputs.data(), NT ? temps.data() : nullptr,         \
                                            note_id_changed, &params, flags), "zh_" #prefix "_paint");                     \
        }                                                                                                                 \
    };

#define ZANG_HIP_NO_ARGS
#define ZANG_HIP_COMMA_SEED , uint64_t first_seed = 0
#define ZANG_HIP_USE_SEED , first_seed
#define ZANG_HIP_COMMA_F32 , zh_f32 init_value
#define ZANG_HIP_USE_F32 , init_value

ZANG_HIP_MODULE(SineOsc, sineosc, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)            // src/modules/SineOsc.zig
ZANG_HIP_MODULE(PulseOsc, pulseosc, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)          // src/modules/PulseOsc.zig
ZANG_HIP_MODULE(TriSawOsc, trisawosc, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)        // src/modules/TriSawOsc.zig
ZANG_HIP_MODULE(Noise, noise, 0, ZANG_HIP_COMMA_SEED, ZANG_HIP_USE_SEED)            // src/modules/Noise.zig (seed = first_seed + voice)
ZANG_HIP_MODULE(Envelope, envelope, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)          // src/modules/Envelope.zig
ZANG_HIP_MODULE(Gate, gate, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)                  // src/modules/Gate.zig
ZANG_HIP_MODULE(Filter, filter, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)              // src/modules/Filter.zig
ZANG_HIP_MODULE(Decimator, decimator, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)        // src/modules/Decimator.zig
ZANG_HIP_MODULE(Distortion, distortion, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)      // src/modules/Distortion.zig
ZANG_HIP_MODULE(Cycle, cycle, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)                // src/modules/Cycle.zig
ZANG_HIP_MODULE(Portamento, portamento, 0, ZANG_HIP_NO_ARGS, ZANG_HIP_NO_ARGS)      // src/modules/Portamento.zig
// the fused composites of examples/modules.zig (temps are accepted and unused: they live in registers)
ZANG_HIP_MODULE(NiceInstrument, nice, 2, ZANG_HIP_COMMA_F32, ZANG_HIP_USE_F32)      // :189-248, init(color)
ZANG_HIP_MODULE(PMOscInstrument, pmosc, 3, ZANG_HIP_COMMA_F32, ZANG_HIP_USE_F32)    // :80-128, init(release_duration)

#undef ZANG_HIP_MODULE
#undef ZANG_HIP_NO_ARGS
#undef ZANG_HIP_COMMA_SEED
#undef ZANG_HIP_USE_SEED
#undef ZANG_HIP_COMMA_F32
#undef ZANG_HIP_USE_F32

// NiceInstrument painted and mixed down over its voices without materialising per-voice output: mono, or two channels with a
// per-voice gain each (a two-output module fed `voice * pan_c`, examples/example_stereo.zig:92-98)
inline void paintMix(NiceInstrument &m, zang::Span span, float *mix_dev, zh_bool note_id_changed, const NiceInstrument::Params &params,
                     uint32_t flags = ZH_PAINT_ADD) {
    zang::check(zh_nice_paint_mix(m.get(), span.start, span.end, mix_dev, note_id_changed, &params, flags), "zh_nice_paint_mix");
}
inline void paintMixStereo(NiceInstrument &m, zang::Span span, float *mix_left_dev, float *mix_right_dev, zh_f32 gain_left, zh_f32 gain_right,
                           zh_bool note_id_changed, const NiceInstrument::Params &params, uint32_t flags = ZH_PAINT_ADD) {
    zang::check(zh_nice_paint_mix_stereo(m.get(), span.start, span.end, mix_left_dev, mix_right_dev, gain_left, gain_right, note_id_changed,
                                         &params, flags), "zh_nice_paint_mix_stereo");
}
// n consecutive paintMixStereo calls (per-buffer params and note_id_changed) as one launch: state in registers between buffers
inline void paintMixStereoBatch(NiceInstrument &m, zang::Span span, const std::vector<float *> &mix_left_dev, const std::vector<float *> &mix_right_dev,
                                zh_f32 gain_left, zh_f32 gain_right, const std::vector<zh_bool> &note_id_changed,
                                const std::vector<NiceInstrument::Params> &params, uint32_t flags = ZH_PAINT_ADD) {
    if (mix_left_dev.size() != params.size() || mix_right_dev.size() != params.size() || note_id_changed.size() != params.size())
        throw zang::Error(ZH_ERR_INVALID, "paintMixStereoBatch: one mix pair, one note_id_changed and one Params per buffer");
    zang::check(zh_nice_paint_mix_stereo_batch(m.get(), span.start, span.end, (uint32_t)params.size(), mix_left_dev.data(), mix_right_dev.data(),
                                               gain_left, gain_right, note_id_changed.data(), params.data(), flags), "zh_nice_paint_mix_stereo_batch");
}
// n consecutive paints of a constant-frequency oscillator (same span and params, buffer b into outputs[b]) as one launch
inline void paintBatch(PulseOsc &m, zang::Span span, const std::vector<zh_buf> &outputs, const PulseOsc::Params &params, uint32_t flags = ZH_PAINT_ADD) {
    zang::check(zh_pulseosc_paint_batch(m.get(), span.start, span.end, outputs.data(), (uint32_t)outputs.size(), &params, flags), "zh_pulseosc_paint_batch");
}
inline void paintBatch(TriSawOsc &m, zang::Span span, const std::vector<zh_buf> &outputs, const TriSawOsc::Params &params, uint32_t flags = ZH_PAINT_ADD) {
    zang::check(zh_trisawosc_paint_batch(m.get(), span.start, span.end, outputs.data(), (uint32_t)outputs.size(), &params, flags), "zh_trisawosc_paint_batch");
}

// mod.Filter.cutoffFromFrequency (Filter.zig:20-23), elementwise on the device
inline void cutoffFromFrequency(zang::Context &c, uint32_t n, float *cutoff_out_dev, const float *frequency_dev, float sample_rate) {
    zang::check(zh_filter_cutoff_from_frequency(c.get(), n, cutoff_out_dev, frequency_dev, sample_rate), "cutoffFromFrequency");
}

}  // namespace mod

// std.math.sin / cos / pow (f32) elementwise on the device, bit-identical to what the modules compute with
namespace zang { namespace math {
inline void sin(zang::Context &c, uint32_t n, float *out_dev, const float *x_dev) { zang::check(zh_sin(c.get(), n, out_dev, x_dev), "sin"); }
inline void cos(zang::Context &c, uint32_t n, float *out_dev, const float *x_dev) { zang::check(zh_cos(c.get(), n, out_dev, x_dev), "cos"); }
inline void atan(zang::Context &c, uint32_t n, float *out_dev, const float *x_dev) { zang::check(zh_atan(c.get(), n, out_dev, x_dev), "atan"); }
inline void pow(zang::Context &c, uint32_t n, float *out_dev, const float *x_dev, const float *y_dev) { zang::check(zh_pow(c.get(), n, out_dev, x_dev, y_dev), "pow"); }
} }  // namespace zang::math
