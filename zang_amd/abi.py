"""ctypes mirror of include/zang_hip.h (the C ABI of libzang_hip.so).

Loading fails loudly when the library is missing: the product has no CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzang_hip.so")

u32, u64, f32, vp = C.c_uint32, C.c_uint64, C.c_float, C.c_void_p

ZH_OK = 0
ZH_ERR_INVALID, ZH_ERR_UNSUPPORTED, ZH_ERR_NO_DEVICE = -1, -2, -3
ZH_ERR_COMM, ZH_ERR_RCCL_BASE = -4, -100
COMM_ID_BYTES = 128
PAINT_ADD, PAINT_ZERO_FIRST = 0, 1
PAINT_PARAMS_UNCHANGED = 4
PAINT_TOLERANT = 8
ZH_CAPTURE_COALESCE = 1
MIX_SEQUENTIAL = 2
AUDIO_SIGNED8, AUDIO_SIGNED16_LSB = 0, 1
COB_CONSTANT, COB_BUFFER = 0, 1
CURVE_INSTANTANEOUS, CURVE_LINEAR, CURVE_SQUARED, CURVE_CUBED = 0, 1, 2, 3
NOISE_WHITE, NOISE_PINK = 0, 1
ENV_IDLE, ENV_ATTACK, ENV_DECAY, ENV_SUSTAIN, ENV_RELEASE = range(5)
FILTER_BYPASS, FILTER_LOW_PASS, FILTER_BAND_PASS, FILTER_HIGH_PASS, FILTER_NOTCH, FILTER_ALL_PASS = range(6)
SAMPLE_U8, SAMPLE_S16_LSB, SAMPLE_S24_LSB, SAMPLE_S32_LSB = range(4)
DISTORTION_OVERDRIVE, DISTORTION_CLIP = 0, 1


class Buf(C.Structure):
    _fields_ = [("ptr", vp), ("voices", u32), ("frames", u32), ("stride", u32), ("reserved", u32)]


class F32(C.Structure):
    _fields_ = [("value", f32), ("reserved", u32), ("per_voice", vp)]


class Bool(C.Structure):
    _fields_ = [("value", u32), ("reserved", u32), ("per_voice", vp)]


class Cob(C.Structure):
    _fields_ = [("tag", u32), ("reserved", u32), ("constant", F32), ("buffer", Buf)]


class Curve(C.Structure):
    _fields_ = [("tag", u32), ("reserved", u32), ("duration", F32)]


class SineOscParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("freq", Cob), ("phase", Cob)]


class SineOscState(C.Structure):
    _fields_ = [("t", f32)]


class PulseOscParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("freq", Cob), ("color", F32)]


class PulseOscState(C.Structure):
    _fields_ = [("cnt", u32)]


TriSawOscParams = PulseOscParams


class TriSawOscState(C.Structure):
    _fields_ = [("cnt", u32), ("t", f32)]


class NoiseParams(C.Structure):
    _fields_ = [("color", u32)]


class NoiseState(C.Structure):
    _fields_ = [("r", u64 * 4), ("b", f32 * 7), ("reserved", u32)]


class EnvelopeParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("attack", Curve), ("decay", Curve), ("release", Curve),
                ("sustain_volume", F32), ("note_on", Bool)]


class EnvelopeState(C.Structure):
    _fields_ = [("state", u32), ("t", f32), ("last_value", f32), ("start", f32)]


class GateParams(C.Structure):
    _fields_ = [("note_on", Bool)]


class FilterParams(C.Structure):
    _fields_ = [("input", Buf), ("type", u32), ("reserved", u32), ("cutoff", Cob), ("res", Cob)]


class FilterState(C.Structure):
    _fields_ = [("l", f32), ("b", f32)]


class Sample(C.Structure):
    _fields_ = [("num_channels", u64), ("sample_rate", u64), ("format", u32), ("reserved", u32),
                ("data", vp), ("data_len", u64)]


class SamplerParams(C.Structure):
    _fields_ = [("sample_rate", F32), ("sample", Sample), ("channel", u64), ("loop", u32), ("reserved", u32)]


class SamplerState(C.Structure):
    _fields_ = [("t", f32)]


class DecimatorParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("input", Buf), ("fake_sample_rate", F32)]


class DecimatorState(C.Structure):
    _fields_ = [("dval", f32), ("dcount", f32)]


class DistortionParams(C.Structure):
    _fields_ = [("input", Buf), ("type", u32), ("reserved", u32), ("ingain", F32), ("outgain", F32), ("offset", F32)]


class DelayParams(C.Structure):
    _fields_ = [("input", Buf)]


class FilteredEchoesParams(C.Structure):
    _fields_ = [("input", Buf), ("feedback_volume", F32), ("cutoff", F32)]


class NoiseFilterParams(C.Structure):
    _fields_ = [("color", u32), ("type", u32), ("cutoff", F32), ("res", F32)]


class NoiseFilterState(C.Structure):
    _fields_ = [("noise", NoiseState), ("flt", FilterState)]


class ScriptParam(C.Structure):            # zh_script_param
    _fields_ = [("kind", u32), ("u", u32), ("f", f32), ("is_buffer", u32), ("pf", vp), ("pb", vp), ("stride", u32), ("reserved", u32)]


SP_CONSTANT, SP_BOOLEAN, SP_COB, SP_BUFFER, SP_ENUM, SP_CURVE = range(6)
SCRIPT_MAX_PARAMS = 16


class CurveNode(C.Structure):
    _fields_ = [("value", f32), ("t", f32)]


class CurveModuleParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("function", u32), ("curve", vp), ("curve_len", u64)]


class CurveModuleState(C.Structure):
    _fields_ = [("t", f32), ("current_song_note", u32), ("current_song_note_offset", C.c_int32), ("next_song_note", u32)]


class CycleParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("speed", Cob)]


class CycleState(C.Structure):
    _fields_ = [("t", f32)]


class PortamentoParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("curve", Curve), ("goal", F32), ("note_on", Bool), ("prev_note_on", Bool)]


class PortamentoState(C.Structure):
    _fields_ = [("t", f32), ("last_value", f32), ("start", f32)]


class NiceParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("reserved", u32), ("freq", F32), ("note_on", Bool)]


class NiceState(C.Structure):
    _fields_ = [("osc", PulseOscState), ("flt", FilterState), ("env", EnvelopeState)]


PMOscParams = NiceParams


class PMOscState(C.Structure):
    _fields_ = [("carrier", SineOscState), ("modulator", SineOscState), ("env", EnvelopeState)]


class SpanTable(C.Structure):
    _fields_ = [("max_spans", u32), ("reserved", u32), ("count", vp), ("start", vp), ("end", vp), ("freq", vp),
                ("note_on", vp), ("note_id_changed", vp)]


class Impulse(C.Structure):
    _fields_ = [("frame", u64), ("note_id", u64), ("event_id", u64)]


class Iap(C.Structure):
    _fields_ = [("impulses", C.POINTER(Impulse)), ("paramses", vp), ("len", u64)]


class PaintSpan(C.Structure):
    _fields_ = [("start", u64), ("end", u64), ("note_id_changed", u32), ("reserved", u32), ("params", C.c_uint8 * 64)]


class HCob(C.Structure):
    _fields_ = [("tag", u32), ("constant", f32), ("buffer", C.POINTER(f32))]


class HCurve(C.Structure):
    _fields_ = [("tag", u32), ("duration", f32)]


class SineOscHostParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("freq", HCob), ("phase", HCob)]


class PulseOscHostParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("freq", HCob), ("color", f32)]


class NoiseHostParams(C.Structure):
    _fields_ = [("color", u32)]


class EnvelopeHostParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("attack", HCurve), ("decay", HCurve), ("release", HCurve), ("sustain_volume", f32), ("note_on", u32)]


class GateHostParams(C.Structure):
    _fields_ = [("note_on", u32)]


class FilterHostParams(C.Structure):
    _fields_ = [("input", C.POINTER(f32)), ("type", u32), ("cutoff", HCob), ("res", HCob)]


class SamplerHostParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("num_channels", u64), ("sample_rate_in", u64), ("format", u32), ("data", C.POINTER(C.c_uint8)),
                ("data_len", u64), ("channel", u64), ("loop", u32)]


class DecimatorHostParams(C.Structure):
    _fields_ = [("sample_rate", f32), ("input", C.POINTER(f32)), ("fake_sample_rate", f32)]


class DistortionHostParams(C.Structure):
    _fields_ = [("input", C.POINTER(f32)), ("type", u32), ("ingain", f32), ("outgain", f32), ("offset", f32)]


P = C.POINTER
_hpaint = lambda params: [vp, vp, u32, u32, C.POINTER(C.POINTER(f32)), C.POINTER(C.POINTER(f32)), u32, C.POINTER(params)]
_paint = lambda params: [vp, u32, u32, P(Buf), P(Buf), Bool, P(params), u32]

# name -> (restype, argtypes); must list every ZH_API symbol of include/zang_hip.h
SIGNATURES = {
    "zh_create": (C.c_int, [P(vp), C.c_int]),
    "zh_destroy": (C.c_int, [vp]),
    "zh_set_stream": (C.c_int, [vp, vp]),
    "zh_get_stream": (vp, [vp]),
    "zh_sync": (C.c_int, [vp]),
    "zh_error_string": (C.c_char_p, [C.c_int]),
    "zh_version": (C.c_char_p, []),
    "zh_malloc": (C.c_int, [vp, P(vp), C.c_size_t]),
    "zh_free": (C.c_int, [vp, vp]),
    "zh_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "zh_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "zh_buf_alloc": (C.c_int, [vp, P(Buf), u32, u32]),
    "zh_buf_free": (C.c_int, [vp, P(Buf)]),
    "zh_buf_upload_voices": (C.c_int, [vp, Buf, vp, u32]),
    "zh_buf_download_voices": (C.c_int, [vp, vp, Buf, u32]),
    "zh_buf_upload_voice": (C.c_int, [vp, Buf, u32, vp, u32]),
    "zh_buf_download_voice": (C.c_int, [vp, vp, Buf, u32, u32]),
    "zh_form_count": (C.c_int, []),
    "zh_form_info": (C.c_int, [C.c_int, P(C.c_char_p), P(C.c_long), P(C.c_long), P(C.c_char_p)]),
    "zh_last_form": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "zh_graph_begin_capture": (C.c_int, [vp]),
    "zh_graph_begin_capture_flags": (C.c_int, [vp, C.c_uint32]),
    "zh_graph_info": (C.c_int, [vp, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]),
    "zh_graph_kernels": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "zh_graph_end_capture": (C.c_int, [vp, P(vp)]),
    "zh_graph_launch": (C.c_int, [vp, vp]),
    "zh_graph_destroy": (C.c_int, [vp]),
    "zh_event_create": (C.c_int, [vp, P(vp)]),
    "zh_event_destroy": (C.c_int, [vp]),
    "zh_event_record": (C.c_int, [vp, vp]),
    "zh_event_elapsed_ms": (C.c_int, [vp, vp, P(f32)]),
    "zh_zero": (C.c_int, [vp, u32, u32, Buf]),
    "zh_set": (C.c_int, [vp, u32, u32, Buf, F32]),
    "zh_copy": (C.c_int, [vp, u32, u32, Buf, Buf]),
    "zh_add": (C.c_int, [vp, u32, u32, Buf, Buf, Buf]),
    "zh_add_into": (C.c_int, [vp, u32, u32, Buf, Buf]),
    "zh_add_scalar": (C.c_int, [vp, u32, u32, Buf, Buf, F32]),
    "zh_add_scalar_into": (C.c_int, [vp, u32, u32, Buf, F32]),
    "zh_multiply": (C.c_int, [vp, u32, u32, Buf, Buf, Buf]),
    "zh_multiply_with": (C.c_int, [vp, u32, u32, Buf, Buf]),
    "zh_multiply_scalar": (C.c_int, [vp, u32, u32, Buf, Buf, F32]),
    "zh_multiply_with_scalar": (C.c_int, [vp, u32, u32, Buf, F32]),
    "zh_mixdown_voices": (C.c_int, [vp, u32, u32, vp, Buf, u32]),
    "zh_ipc_alloc": (C.c_int, [vp, C.c_size_t, P(vp), vp]),
    "zh_ipc_open": (C.c_int, [vp, vp, P(vp)]),
    "zh_ipc_close": (C.c_int, [vp, vp]),
    "zh_sum_slots": (C.c_int, [vp, vp, vp, u32, C.c_size_t, C.c_size_t, u32]),
    "zh_comm_available": (C.c_int, []),
    "zh_comm_library": (C.c_char_p, []),
    "zh_comm_version": (C.c_int, []),
    "zh_comm_last_error": (C.c_char_p, []),
    "zh_comm_unique_id": (C.c_int, [vp]),
    "zh_comm_create": (C.c_int, [vp, u32, u32, vp, P(vp)]),
    "zh_comm_set_timeout": (C.c_int, [C.c_double]),
    "zh_comm_check": (C.c_int, [vp]),
    "zh_comm_abort": (C.c_int, [vp]),
    "zh_comm_destroy": (C.c_int, [vp]),
    "zh_comm_world": (C.c_int, [vp]),
    "zh_comm_rank": (C.c_int, [vp]),
    "zh_allreduce_mix": (C.c_int, [vp, vp, C.c_size_t]),
    "zh_reduce_mix": (C.c_int, [vp, vp, C.c_size_t, u32]),
    "zh_nice_paint_mix_stereo_batch": (C.c_int, [vp, u32, u32, u32, P(vp), P(vp), F32, F32, P(Bool), P(NiceParams), u32]),
    "zh_sineosc_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_sineosc_destroy": (C.c_int, [vp]),
    "zh_sineosc_get_state": (C.c_int, [vp, vp]),
    "zh_sineosc_set_state": (C.c_int, [vp, vp]),
    "zh_sineosc_paint": (C.c_int, _paint(SineOscParams)),
    "zh_pulseosc_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_pulseosc_destroy": (C.c_int, [vp]),
    "zh_pulseosc_get_state": (C.c_int, [vp, vp]),
    "zh_pulseosc_set_state": (C.c_int, [vp, vp]),
    "zh_pulseosc_paint": (C.c_int, _paint(PulseOscParams)),
    "zh_pulseosc_paint_batch": (C.c_int, [vp, u32, u32, P(Buf), u32, P(PulseOscParams), u32]),
    "zh_trisawosc_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_trisawosc_destroy": (C.c_int, [vp]),
    "zh_trisawosc_get_state": (C.c_int, [vp, vp]),
    "zh_trisawosc_set_state": (C.c_int, [vp, vp]),
    "zh_trisawosc_paint": (C.c_int, _paint(TriSawOscParams)),
    "zh_trisawosc_paint_batch": (C.c_int, [vp, u32, u32, P(Buf), u32, P(TriSawOscParams), u32]),
    "zh_noise_create": (C.c_int, [vp, u32, u64, P(vp)]),
    "zh_noise_destroy": (C.c_int, [vp]),
    "zh_noise_get_state": (C.c_int, [vp, vp]),
    "zh_noise_set_state": (C.c_int, [vp, vp]),
    "zh_selftest_noise_jump": (C.c_int, [u64, u32]),
    "zh_noise_paint": (C.c_int, _paint(NoiseParams)),
    "zh_envelope_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_envelope_destroy": (C.c_int, [vp]),
    "zh_envelope_get_state": (C.c_int, [vp, vp]),
    "zh_envelope_set_state": (C.c_int, [vp, vp]),
    "zh_envelope_paint": (C.c_int, _paint(EnvelopeParams)),
    "zh_gate_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_gate_destroy": (C.c_int, [vp]),
    "zh_gate_paint": (C.c_int, _paint(GateParams)),
    "zh_filter_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_filter_destroy": (C.c_int, [vp]),
    "zh_filter_get_state": (C.c_int, [vp, vp]),
    "zh_filter_set_state": (C.c_int, [vp, vp]),
    "zh_filter_paint": (C.c_int, _paint(FilterParams)),
    "zh_filter_cutoff_from_frequency": (C.c_int, [vp, u32, vp, vp, f32]),
    "zh_sampler_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_sampler_destroy": (C.c_int, [vp]),
    "zh_sampler_get_state": (C.c_int, [vp, vp]),
    "zh_sampler_set_state": (C.c_int, [vp, vp]),
    "zh_sampler_paint": (C.c_int, _paint(SamplerParams)),
    "zh_decimator_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_decimator_destroy": (C.c_int, [vp]),
    "zh_decimator_get_state": (C.c_int, [vp, vp]),
    "zh_decimator_set_state": (C.c_int, [vp, vp]),
    "zh_decimator_paint": (C.c_int, _paint(DecimatorParams)),
    "zh_distortion_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_distortion_destroy": (C.c_int, [vp]),
    "zh_distortion_paint": (C.c_int, _paint(DistortionParams)),
    "zh_curve_module_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_curve_module_destroy": (C.c_int, [vp]),
    "zh_curve_module_get_state": (C.c_int, [vp, vp]),
    "zh_curve_module_set_state": (C.c_int, [vp, vp]),
    "zh_curve_module_paint": (C.c_int, _paint(CurveModuleParams)),
    "zh_cycle_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_cycle_destroy": (C.c_int, [vp]),
    "zh_cycle_get_state": (C.c_int, [vp, vp]),
    "zh_cycle_set_state": (C.c_int, [vp, vp]),
    "zh_cycle_paint": (C.c_int, _paint(CycleParams)),
    "zh_portamento_create": (C.c_int, [vp, u32, P(vp)]),
    "zh_portamento_destroy": (C.c_int, [vp]),
    "zh_portamento_get_state": (C.c_int, [vp, vp]),
    "zh_portamento_set_state": (C.c_int, [vp, vp]),
    "zh_portamento_paint": (C.c_int, _paint(PortamentoParams)),
    "zh_poly_voice_create": (C.c_int, [u32, u32, u32, u64, vp, vp, vp, P(vp)]),
    "zh_poly_voice_destroy": (C.c_int, [vp]),
    "zh_poly_voice_reset": (C.c_int, [vp]),
    "zh_poly_voice_schedule": (C.c_int, [vp, f32, vp, u32, u32, vp, vp, vp, vp, vp]),
    "zh_zscript_compile": (C.c_int, [C.c_char_p, C.c_char_p, u32, P(vp), C.c_char_p, C.c_size_t]),
    "zh_zscript_destroy": (C.c_int, [vp]),
    "zh_zscript_free_text": (None, [vp]),
    "zh_zscript_generate_zig": (C.c_int, [vp, P(vp)]),
    "zh_zscript_generate_hip": (C.c_int, [vp, C.c_char_p, C.c_int, P(vp)]),
    "zh_zscript_generate_hip_forms": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_uint32, P(vp)]),
    "zh_zscript_module_count": (u32, [vp]),
    "zh_zscript_module_info": (C.c_int, [vp, u32, C.c_char_p, C.c_size_t, P(u32), P(u32), P(u32), C.c_char_p, C.c_size_t]),
    "zh_zscript_module_param": (C.c_int, [vp, u32, u32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "zh_zscript_module_num_temps": (C.c_int, [vp, u32, P(u32)]),
    "zh_script_compile": (C.c_int, [C.c_char_p, P(vp), P(C.c_size_t), C.c_char_p, C.c_size_t]),
    "zh_script_free_code": (None, [vp]),
    "zh_script_load": (C.c_int, [vp, C.c_char_p, P(vp), C.c_char_p, C.c_size_t]),
    "zh_script_load_code": (C.c_int, [vp, vp, C.c_size_t, P(vp)]),
    "zh_script_destroy": (C.c_int, [vp]),
    "zh_script_module_create": (C.c_int, [vp, C.c_char_p, u32, u32, u64, P(vp)]),
    "zh_script_module_destroy": (C.c_int, [vp]),
    "zh_script_module_ranges_ok": (C.c_int, [vp]),
    "zh_script_module_get_state": (C.c_int, [vp, vp]),
    "zh_script_module_set_state": (C.c_int, [vp, vp]),
    "zh_script_module_paint": (C.c_int, [vp, u32, u32, P(Buf), Bool, P(ScriptParam), u32, u32]),
    "zh_nice_create": (C.c_int, [vp, u32, F32, P(vp)]),
    "zh_nice_destroy": (C.c_int, [vp]),
    "zh_nice_get_state": (C.c_int, [vp, vp]),
    "zh_nice_set_state": (C.c_int, [vp, vp]),
    "zh_nice_paint": (C.c_int, _paint(NiceParams)),
    "zh_nice_paint_mix": (C.c_int, [vp, u32, u32, vp, Bool, P(NiceParams), u32]),
    "zh_nice_paint_mix_stereo": (C.c_int, [vp, u32, u32, vp, vp, F32, F32, Bool, P(NiceParams), u32]),
    "zh_delay_create": (C.c_int, [vp, u32, u32, P(vp)]),
    "zh_delay_destroy": (C.c_int, [vp]),
    "zh_delay_reset": (C.c_int, [vp]),
    "zh_delay_get_state": (C.c_int, [vp, vp, vp]),
    "zh_delay_set_state": (C.c_int, [vp, vp, vp]),
    "zh_delay_paint": (C.c_int, _paint(DelayParams)),
    "zh_filtered_echoes_create": (C.c_int, [vp, u32, u32, P(vp)]),
    "zh_filtered_echoes_destroy": (C.c_int, [vp]),
    "zh_filtered_echoes_reset": (C.c_int, [vp]),
    "zh_filtered_echoes_get_state": (C.c_int, [vp, vp, vp, vp]),
    "zh_filtered_echoes_set_state": (C.c_int, [vp, vp, vp, vp]),
    "zh_filtered_echoes_paint": (C.c_int, _paint(FilteredEchoesParams)),
    "zh_noise_filter_create": (C.c_int, [vp, u32, u64, P(vp)]),
    "zh_noise_filter_destroy": (C.c_int, [vp]),
    "zh_noise_filter_get_state": (C.c_int, [vp, vp]),
    "zh_noise_filter_set_state": (C.c_int, [vp, vp]),
    "zh_noise_filter_paint": (C.c_int, _paint(NoiseFilterParams)),
    "zh_pmosc_create": (C.c_int, [vp, u32, F32, P(vp)]),
    "zh_pmosc_destroy": (C.c_int, [vp]),
    "zh_pmosc_get_state": (C.c_int, [vp, vp]),
    "zh_pmosc_set_state": (C.c_int, [vp, vp]),
    "zh_pmosc_paint": (C.c_int, _paint(PMOscParams)),
    "zh_pow": (C.c_int, [vp, u32, vp, vp, vp]),
    "zh_sin": (C.c_int, [vp, u32, vp, vp]),
    "zh_cos": (C.c_int, [vp, u32, vp, vp]),
    "zh_atan": (C.c_int, [vp, u32, vp, vp]),
    "zh_mix_down": (C.c_int, [vp, vp, vp, u32, u32, u32, u32, f32]),
    "zh_nice_paint_spans": (C.c_int, [vp, u32, u32, P(Buf), P(Buf), f32, P(SpanTable), u32]),
    "zh_pmosc_paint_spans": (C.c_int, [vp, u32, u32, P(Buf), P(Buf), f32, P(SpanTable), u32]),
    "zh_noise_state_init": (C.c_int, [vp, u64]),
    "zh_decimator_state_init": (C.c_int, [vp]),
    "zh_sineosc_paint_host": (C.c_int, _hpaint(SineOscHostParams)),
    "zh_pulseosc_paint_host": (C.c_int, _hpaint(PulseOscHostParams)),
    "zh_trisawosc_paint_host": (C.c_int, _hpaint(PulseOscHostParams)),
    "zh_noise_paint_host": (C.c_int, _hpaint(NoiseHostParams)),
    "zh_envelope_paint_host": (C.c_int, _hpaint(EnvelopeHostParams)),
    "zh_gate_paint_host": (C.c_int, _hpaint(GateHostParams)),
    "zh_filter_paint_host": (C.c_int, _hpaint(FilterHostParams)),
    "zh_sampler_paint_host": (C.c_int, _hpaint(SamplerHostParams)),
    "zh_decimator_paint_host": (C.c_int, _hpaint(DecimatorHostParams)),
    "zh_distortion_paint_host": (C.c_int, _hpaint(DistortionHostParams)),
    "zh_impulse_queue_create": (C.c_int, [u32, P(vp)]),
    "zh_impulse_queue_destroy": (C.c_int, [vp]),
    "zh_impulse_queue_push": (C.c_int, [vp, u64, u64, vp]),
    "zh_impulse_queue_consume": (C.c_int, [vp, P(Iap)]),
    "zh_note_tracker_create": (C.c_int, [u32, u64, vp, P(f32), P(u64), P(vp)]),
    "zh_note_tracker_destroy": (C.c_int, [vp]),
    "zh_note_tracker_reset": (C.c_int, [vp]),
    "zh_note_tracker_consume": (C.c_int, [vp, f32, u64, u64, P(Iap)]),
    "zh_polyphony_dispatcher_create": (C.c_int, [u32, u32, u32, P(vp)]),
    "zh_polyphony_dispatcher_destroy": (C.c_int, [vp]),
    "zh_polyphony_dispatcher_reset": (C.c_int, [vp]),
    "zh_polyphony_dispatcher_dispatch": (C.c_int, [vp, Iap, P(Iap)]),
    "zh_trigger_create": (C.c_int, [u32, P(vp)]),
    "zh_trigger_destroy": (C.c_int, [vp]),
    "zh_trigger_reset": (C.c_int, [vp]),
    "zh_trigger_counter": (C.c_int, [vp, u64, u64, Iap]),
    "zh_trigger_next": (C.c_int, [vp, P(PaintSpan)]),
}

_lib = None


class ZangHipError(RuntimeError):
    pass


def load(strict=True):
    """Load libzang_hip.so and bind every declared entry point.

    strict=True raises if any symbol of SIGNATURES is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64; load it first so that libzang_hip.so binds to the same HIP
    # runtime (two runtimes in one process cannot both see the device).
    import torch  # noqa: F401
    global LIB_PATH
    LIB_PATH = os.environ.get("ZANG_HIP_LIB", LIB_PATH)       # A/B timing of alternative builds of the same library
    if not os.path.exists(LIB_PATH):
        raise ZangHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). zang_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise ZangHipError(f"libzang_hip.so lacks symbols: {missing}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        lib = load(strict=False)
        msg = lib.zh_error_string(rc).decode()
        if rc == ZH_ERR_COMM or rc <= ZH_ERR_RCCL_BASE:
            msg += ": " + lib.zh_comm_last_error().decode()
        raise ZangHipError(f"{what} failed: {rc} ({msg})")
