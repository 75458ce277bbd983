"""ctypes binding of the ORACLE (oracle/libzang_oracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by zang_amd (the product has no CPU path).

Buffers are numpy float32 arrays, one contiguous row per voice ([voice][frame]), exactly
as the reference hands each module a `[]f32` (src/modules/SineOsc.zig:24-31).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzang_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("zang_oracle.c", "zang_oracle.h", "zmath_ref.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Cob(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("constant", C.c_float), ("buffer", C.POINTER(C.c_float))]


class Curve(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("duration", C.c_float)]


class Painter(C.Structure):
    _fields_ = [("t", C.c_float), ("last_value", C.c_float), ("start", C.c_float)]


class SineOsc(C.Structure):
    _fields_ = [("t", C.c_float)]


class PulseOsc(C.Structure):
    _fields_ = [("cnt", C.c_uint32)]


class TriSawOsc(C.Structure):
    _fields_ = [("cnt", C.c_uint32), ("t", C.c_float)]


class Noise(C.Structure):
    _fields_ = [("r", C.c_uint64 * 4), ("b", C.c_float * 7)]


class Envelope(C.Structure):
    _fields_ = [("state", C.c_uint32), ("painter", Painter)]


class EnvelopeParams(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("attack", Curve), ("decay", Curve), ("release", Curve),
                ("sustain_volume", C.c_float), ("note_on", C.c_int32)]


class Filter(C.Structure):
    _fields_ = [("l", C.c_float), ("b", C.c_float)]


class Sampler(C.Structure):
    _fields_ = [("t", C.c_float)]


class SamplerParams(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("num_channels", C.c_size_t), ("sample_rate_in", C.c_size_t),
                ("format", C.c_uint32), ("data", C.POINTER(C.c_uint8)), ("data_len", C.c_size_t),
                ("channel", C.c_size_t), ("loop", C.c_int32)]


class Decimator(C.Structure):
    _fields_ = [("dval", C.c_float), ("dcount", C.c_float)]


class CurveNode(C.Structure):
    _fields_ = [("value", C.c_float), ("t", C.c_float)]


class CurveModule(C.Structure):
    _fields_ = [("t", C.c_float), ("current_song_note", C.c_size_t), ("current_song_note_offset", C.c_int32),
                ("next_song_note", C.c_size_t)]


class Delay(C.Structure):
    _fields_ = [("ring", C.POINTER(C.c_float)), ("delay_samples", C.c_size_t), ("index", C.c_size_t)]


class Cycle(C.Structure):
    _fields_ = [("t", C.c_float)]


class Portamento(C.Structure):
    _fields_ = [("painter", Painter)]


class NiceInstrument(C.Structure):
    _fields_ = [("color", C.c_float), ("osc", PulseOsc), ("flt", Filter), ("env", Envelope)]


class PMOscInstrument(C.Structure):
    _fields_ = [("release_duration", C.c_float), ("carrier", SineOsc), ("modulator", SineOsc), ("env", Envelope)]


class FilteredSawtooth(C.Structure):      # examples/modules.zig:141-143
    _fields_ = [("osc", TriSawOsc), ("env", Envelope), ("flt", Filter)]


class HardSquare(C.Structure):            # examples/modules.zig:260-261 (Gate has no state)
    _fields_ = [("osc", PulseOsc)]


COB_CONSTANT, COB_BUFFER = 0, 1
CURVE_INSTANTANEOUS, CURVE_LINEAR, CURVE_SQUARED, CURVE_CUBED = 0, 1, 2, 3
NOISE_WHITE, NOISE_PINK = 0, 1
ENV_IDLE, ENV_ATTACK, ENV_DECAY, ENV_SUSTAIN, ENV_RELEASE = range(5)
FILTER_BYPASS, FILTER_LOW_PASS, FILTER_BAND_PASS, FILTER_HIGH_PASS, FILTER_NOTCH, FILTER_ALL_PASS = range(6)
SAMPLE_U8, SAMPLE_S16, SAMPLE_S24, SAMPLE_S32 = range(4)
DISTORTION_OVERDRIVE, DISTORTION_CLIP = 0, 1

_F = C.POINTER(C.c_float)
_lib = None


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def constant(x):
    return Cob(COB_CONSTANT, float(x), None)


def buffer(a):
    c = Cob(COB_BUFFER, 0.0, fptr(a))
    c._keep = a
    return c


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    z, f, u32, i32 = C.c_size_t, C.c_float, C.c_uint32, C.c_int
    sig = {
        "zo_zero": (None, [z, z, _F]),
        "zo_set": (None, [z, z, _F, f]),
        "zo_copy": (None, [z, z, _F, _F]),
        "zo_add": (None, [z, z, _F, _F, _F]),
        "zo_add_into": (None, [z, z, _F, _F]),
        "zo_add_scalar": (None, [z, z, _F, _F, f]),
        "zo_add_scalar_into": (None, [z, z, _F, f]),
        "zo_multiply": (None, [z, z, _F, _F, _F]),
        "zo_multiply_with": (None, [z, z, _F, _F]),
        "zo_multiply_scalar": (None, [z, z, _F, _F, f]),
        "zo_multiply_with_scalar": (None, [z, z, _F, f]),
        "zo_sineosc_init": (None, [C.POINTER(SineOsc)]),
        "zo_sineosc_paint": (None, [C.POINTER(SineOsc), z, z, _F, f, Cob, Cob]),
        "zo_pulseosc_init": (None, [C.POINTER(PulseOsc)]),
        "zo_pulseosc_paint": (None, [C.POINTER(PulseOsc), z, z, _F, f, Cob, f]),
        "zo_trisawosc_init": (None, [C.POINTER(TriSawOsc)]),
        "zo_trisawosc_paint": (None, [C.POINTER(TriSawOsc), z, z, _F, f, Cob, f]),
        "zo_noise_init": (None, [C.POINTER(Noise), C.c_uint64]),
        "zo_noise_paint": (None, [C.POINTER(Noise), z, z, _F, u32]),
        "zo_envelope_init": (None, [C.POINTER(Envelope)]),
        "zo_envelope_paint": (None, [C.POINTER(Envelope), z, z, _F, i32, C.POINTER(EnvelopeParams)]),
        "zo_gate_paint": (None, [z, z, _F, i32]),
        "zo_filter_init": (None, [C.POINTER(Filter)]),
        "zo_filter_cutoff_from_frequency": (f, [f, f]),
        "zo_filter_paint": (None, [C.POINTER(Filter), z, z, _F, _F, u32, Cob, Cob]),
        "zo_sampler_init": (None, [C.POINTER(Sampler)]),
        "zo_sampler_paint": (None, [C.POINTER(Sampler), z, z, _F, i32, C.POINTER(SamplerParams)]),
        "zo_decimator_init": (None, [C.POINTER(Decimator)]),
        "zo_decimator_paint": (None, [C.POINTER(Decimator), z, z, _F, f, _F, f]),
        "zo_distortion_paint": (None, [z, z, _F, _F, u32, f, f, f]),
        "zo_nice_init": (None, [C.POINTER(NiceInstrument), f]),
        "zo_nice_paint": (None, [C.POINTER(NiceInstrument), z, z, _F, _F, _F, i32, f, f, i32]),
        "zo_filtered_sawtooth_init": (None, [C.POINTER(FilteredSawtooth)]),
        "zo_filtered_sawtooth_paint": (None, [C.POINTER(FilteredSawtooth), z, z, _F, _F, _F, _F, i32, f, Cob, i32]),
        "zo_hard_square_init": (None, [C.POINTER(HardSquare)]),
        "zo_hard_square_paint": (None, [C.POINTER(HardSquare), z, z, _F, _F, _F, i32, f, f, i32]),
        "zo_note_c5": (f, []),
        "zo_pmosc_init": (None, [C.POINTER(PMOscInstrument), f]),
        "zo_pmosc_paint": (None, [C.POINTER(PMOscInstrument), z, z, _F, _F, _F, _F, i32, f, f, i32]),
        "zo_mixdown_s16lsb": (None, [C.POINTER(C.c_uint8), _F, z, z, z, f]),
        "zo_mixdown_s8": (None, [C.POINTER(C.c_uint8), _F, z, z, z, f]),
        "zo_math_sinf": (f, [f]), "zo_math_cosf": (f, [f]), "zo_math_atanf": (f, [f]),
        "zo_math_powf": (f, [f, f]), "zo_math_expf": (f, [f]), "zo_math_logf": (f, [f]),
        "zo_math_sinf_n": (None, [_F, _F, z]), "zo_math_cosf_n": (None, [_F, _F, z]),
        "zo_math_atanf_n": (None, [_F, _F, z]), "zo_math_pow2f_n": (None, [_F, _F, z]),
        "zo_curve_init": (None, [C.POINTER(CurveModule)]),
        "zo_curve_paint": (None, [C.POINTER(CurveModule), z, z, _F, i32, f, u32, C.POINTER(CurveNode), z]),
        "zo_delay_init": (None, [C.POINTER(Delay), _F, z]),
        "zo_simple_delay_paint": (None, [C.POINTER(Delay), z, z, _F, _F]),
        "zo_filtered_echoes_paint": (None, [C.POINTER(Delay), C.POINTER(Filter), z, z, _F, _F, _F, _F, f, f]),
        "zo_cycle_init": (None, [C.POINTER(Cycle)]),
        "zo_cycle_paint": (None, [C.POINTER(Cycle), z, z, _F, f, Cob]),
        "zo_portamento_init": (None, [C.POINTER(Portamento)]),
        "zo_portamento_paint": (None, [C.POINTER(Portamento), z, z, _F, i32, f, Curve, f, i32, i32]),
        "zo_bench_pulseosc": (C.c_double, [u32, u32, u32, f, _F, _F, C.POINTER(PulseOsc), _F]),
        "zo_bench_noise_filter": (C.c_double, [u32, u32, u32, _F, _F, C.POINTER(Noise), C.POINTER(Filter), _F]),
        "zo_bench_nice": (C.c_double, [u32, u32, u32, f, _F, C.POINTER(NiceInstrument), _F]),
        "zo_xoshiro_seq": (None, [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), z]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def curve(tag, duration=0.0):
    return Curve(tag, float(duration))
