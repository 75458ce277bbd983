"""Host-side mirror of the reference's `modules` namespace (src/modules.zig:1-13) for the
paint path: each class keeps the Zig module's interface --

    num_outputs, num_temps, Params, init(), paint(span, outputs, temps, note_id_changed, params)

(src/modules/SineOsc.zig:8-31) -- but one instance holds n_voices independent module states
on the device and one paint() call renders all of them through libzang_hip.so.
`outputs`/`temps` are lists of [frame][voice] float32 CUDA tensors; scalar params may be
Python floats (broadcast) or float32 CUDA tensors [n_voices]; `note_id_changed`/`note_on`
may be bools or uint8/bool CUDA tensors [n_voices].
"""
import ctypes as C
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import abi
from .runtime import as_bool, as_buf, as_f32, default_context


def _bufarray(tensors):
    if not tensors:
        return None
    arr = (abi.Buf * len(tensors))(*[as_buf(t) for t in tensors])
    return arr


def _np_state(ctype, n):
    return np.zeros(n, dtype=np.dtype(ctype))


class _Module:
    num_outputs = 1
    num_temps = 0
    _prefix = ""
    _state_ctype = None

    def __init__(self, n_voices, ctx=None, *create_args):
        self.ctx = ctx or default_context()
        self.n_voices = int(n_voices)
        self.lib = self.ctx.lib
        h = C.c_void_p()
        create = getattr(self.lib, f"zh_{self._prefix}_create")
        abi.check(create(self.ctx.handle, self.n_voices, *create_args, C.byref(h)), f"zh_{self._prefix}_create")
        self.handle = h
        self.ctx._children.add(self)

    @classmethod
    def init(cls, n_voices, ctx=None):
        return cls(n_voices, ctx)

    def close(self):
        if getattr(self, "handle", None):
            getattr(self.lib, f"zh_{self._prefix}_destroy")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # state as a numpy structured array mirroring the Zig struct fields
    def state(self):
        st = _np_state(self._state_ctype, self.n_voices)
        abi.check(getattr(self.lib, f"zh_{self._prefix}_get_state")(self.handle, st.ctypes.data), "get_state")
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.dtype(self._state_ctype))
        assert st.shape == (self.n_voices,)
        abi.check(getattr(self.lib, f"zh_{self._prefix}_set_state")(self.handle, st.ctypes.data), "set_state")

    def _paint(self, span, outputs, temps, note_id_changed, cparams, zero_first, extra_flags=0):
        outs = _bufarray(outputs)
        tmps = _bufarray(temps)
        fn = getattr(self.lib, f"zh_{self._prefix}_paint")
        rc = fn(self.handle, span.start, span.end, outs, tmps, as_bool(note_id_changed), C.byref(cparams),
                (abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD) | extra_flags)
        abi.check(rc, f"zh_{self._prefix}_paint")

    def _paint_batch(self, span, images, cparams, zero_first, extra_flags=0):
        outs = _bufarray(images)
        fn = getattr(self.lib, f"zh_{self._prefix}_paint_batch")
        rc = fn(self.handle, span.start, span.end, outs, len(images), C.byref(cparams),
                (abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD) | extra_flags)
        abi.check(rc, f"zh_{self._prefix}_paint_batch")


class SineOsc(_Module):
    """src/modules/SineOsc.zig"""
    _prefix = "sineosc"
    _state_ctype = abi.SineOscState

    @dataclass
    class Params:
        sample_rate: float
        freq: Any   # zang.constant(...) | zang.buffer(...)
        phase: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT -- the sine in f32 (within 1e-5 of the peak, measured 3e-7); the phase state stays exact."""
        cp = abi.SineOscParams(params.sample_rate, 0, params.freq, params.phase)
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_TOLERANT if tolerant else 0)


class PulseOsc(_Module):
    """src/modules/PulseOsc.zig"""
    _prefix = "pulseosc"
    _state_ctype = abi.PulseOscState

    @dataclass
    class Params:
        sample_rate: float
        freq: Any
        color: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, params_unchanged=False):
        """`params_unchanged`: the caller states that `params` (per-voice array contents included) are what the
        previous paint of this module got (ZH_PAINT_PARAMS_UNCHANGED, include/zang_hip.h); same bits either way."""
        cp = abi.PulseOscParams(params.sample_rate, 0, params.freq, as_f32(params.color))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_PARAMS_UNCHANGED if params_unchanged else 0)

    def paint_batch(self, span, images, params, zero_first=False, params_unchanged=False):
        """len(images) consecutive paint calls with the same span and params, call b into images[b], as one launch
        when the frequency is constant (zh_pulseosc_paint_batch)."""
        cp = abi.PulseOscParams(params.sample_rate, 0, params.freq, as_f32(params.color))
        self._paint_batch(span, images, cp, zero_first, abi.PAINT_PARAMS_UNCHANGED if params_unchanged else 0)


class TriSawOsc(_Module):
    """src/modules/TriSawOsc.zig"""
    _prefix = "trisawosc"
    _state_ctype = abi.TriSawOscState

    @dataclass
    class Params:
        sample_rate: float
        freq: Any
        color: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, params_unchanged=False):
        """`params_unchanged`: the caller states that `params` (per-voice array contents included) are what the
        previous paint of this module got (ZH_PAINT_PARAMS_UNCHANGED, include/zang_hip.h); same bits either way."""
        cp = abi.TriSawOscParams(params.sample_rate, 0, params.freq, as_f32(params.color))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_PARAMS_UNCHANGED if params_unchanged else 0)

    def paint_batch(self, span, images, params, zero_first=False, params_unchanged=False):
        """len(images) consecutive paint calls with the same span and params, call b into images[b], as one launch
        when the frequency is constant (zh_trisawosc_paint_batch)."""
        cp = abi.TriSawOscParams(params.sample_rate, 0, params.freq, as_f32(params.color))
        self._paint_batch(span, images, cp, zero_first, abi.PAINT_PARAMS_UNCHANGED if params_unchanged else 0)


class Noise(_Module):
    """src/modules/Noise.zig.  `first_seed` is the global index of voice 0: voice v is seeded
    like the (first_seed+v)-th Noise.init() of a process (Noise.zig:9,26)."""
    _prefix = "noise"
    _state_ctype = abi.NoiseState
    white, pink = abi.NOISE_WHITE, abi.NOISE_PINK

    @dataclass
    class Params:
        color: int

    def __init__(self, n_voices, ctx=None, first_seed=0):
        super().__init__(n_voices, ctx, C.c_uint64(first_seed))

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT -- pink noise at few voices: the taps as chunks at once over exactly generated white
        noise (1e-5 of the voice's peak); white noise and the generator's state are always exact."""
        self._paint(span, outputs, temps, note_id_changed, abi.NoiseParams(params.color), zero_first, abi.PAINT_TOLERANT if tolerant else 0)


class Envelope(_Module):
    """src/modules/Envelope.zig over src/zang/painter.zig"""
    _prefix = "envelope"
    _state_ctype = abi.EnvelopeState

    @dataclass
    class Params:
        sample_rate: float
        attack: Any     # zang.PaintCurve.*
        decay: Any
        release: Any
        sustain_volume: Any
        note_on: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        cp = abi.EnvelopeParams(params.sample_rate, 0, params.attack, params.decay, params.release,
                                as_f32(params.sustain_volume), as_bool(params.note_on))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first)


class Gate(_Module):
    """src/modules/Gate.zig (stateless)"""
    _prefix = "gate"

    @dataclass
    class Params:
        note_on: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        self._paint(span, outputs, temps, note_id_changed, abi.GateParams(as_bool(params.note_on)), zero_first)

    def state(self):
        raise AttributeError("Gate has no state")


class Filter(_Module):
    """src/modules/Filter.zig"""
    _prefix = "filter"
    _state_ctype = abi.FilterState
    bypass, low_pass, band_pass, high_pass, notch, all_pass = range(6)

    @dataclass
    class Params:
        input: Any
        type: int
        cutoff: Any
        res: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT (opt-in, 1e-5 of the signal's peak instead of bits; include/zang_hip.h)."""
        cp = abi.FilterParams(as_buf(params.input), params.type, 0, params.cutoff, params.res)
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_TOLERANT if tolerant else 0)

    @staticmethod
    def cutoffFromFrequency(frequency, sample_rate, ctx=None):
        """Filter.cutoffFromFrequency (Filter.zig:20-23) for a float32 CUDA tensor of frequencies."""
        import torch
        c = ctx or default_context()
        out = torch.empty_like(frequency)
        abi.check(c.lib.zh_filter_cutoff_from_frequency(c.handle, frequency.numel(), out.data_ptr(),
                                                        frequency.data_ptr(), float(sample_rate)), "cutoff")
        return out


class Sampler(_Module):
    """src/modules/Sampler.zig"""
    _prefix = "sampler"
    _state_ctype = abi.SamplerState
    unsigned8, signed16_lsb, signed24_lsb, signed32_lsb = range(4)

    @dataclass
    class Sample:
        num_channels: int
        sample_rate: int
        format: int
        data: Any       # uint8 CUDA tensor

    @dataclass
    class Params:
        sample_rate: Any
        sample: Any
        channel: int
        loop: bool

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        s = params.sample
        cs = abi.Sample(s.num_channels, s.sample_rate, s.format, 0, s.data.data_ptr(), s.data.numel())
        cp = abi.SamplerParams(as_f32(params.sample_rate), cs, params.channel, 1 if params.loop else 0, 0)
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first)


class Decimator(_Module):
    """src/modules/Decimator.zig"""
    _prefix = "decimator"
    _state_ctype = abi.DecimatorState

    @dataclass
    class Params:
        sample_rate: float
        input: Any
        fake_sample_rate: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        cp = abi.DecimatorParams(params.sample_rate, 0, as_buf(params.input), as_f32(params.fake_sample_rate))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first)


class Distortion(_Module):
    """src/modules/Distortion.zig (stateless)"""
    _prefix = "distortion"
    overdrive, clip = 0, 1

    @dataclass
    class Params:
        input: Any
        type: int
        ingain: Any
        outgain: Any
        offset: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        cp = abi.DistortionParams(as_buf(params.input), params.type, 0, as_f32(params.ingain),
                                  as_f32(params.outgain), as_f32(params.offset))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first)

    def state(self):
        raise AttributeError("Distortion has no state")


class Curve(_Module):
    """src/modules/Curve.zig.  `curve` is a float32 CUDA tensor [n_nodes, 2] of (value, t) rows
    (zang.CurveNode), shared by all voices."""
    _prefix = "curve_module"
    _state_ctype = abi.CurveModuleState
    linear, smoothstep = 0, 1

    @dataclass
    class Params:
        sample_rate: float
        function: int
        curve: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        c = params.curve
        assert c.is_cuda and c.dim() == 2 and c.shape[1] == 2 and c.is_contiguous()
        cp = abi.CurveModuleParams(params.sample_rate, params.function, c.data_ptr(), c.shape[0])
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first)


class Cycle(_Module):
    """src/modules/Cycle.zig"""
    _prefix = "cycle"
    _state_ctype = abi.CycleState

    @dataclass
    class Params:
        sample_rate: float
        speed: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        self._paint(span, outputs, temps, note_id_changed, abi.CycleParams(params.sample_rate, 0, params.speed), zero_first)


class Portamento(_Module):
    """src/modules/Portamento.zig"""
    _prefix = "portamento"
    _state_ctype = abi.PortamentoState

    @dataclass
    class Params:
        sample_rate: float
        curve: Any
        goal: Any
        note_on: Any
        prev_note_on: Any

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        cp = abi.PortamentoParams(params.sample_rate, 0, params.curve, as_f32(params.goal), as_bool(params.note_on),
                                  as_bool(params.prev_note_on))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first)


class NiceInstrument(_Module):
    """examples/modules.zig:189-248 as one fused kernel (temps are accepted and ignored)."""
    _prefix = "nice"
    _state_ctype = abi.NiceState
    num_temps = 2

    @dataclass
    class Params:
        sample_rate: float
        freq: Any
        note_on: Any

    def __init__(self, n_voices, color, ctx=None):
        self._color = as_f32(color)
        super().__init__(n_voices, ctx, self._color)

    @classmethod
    def init(cls, n_voices, color, ctx=None):
        return cls(n_voices, color, ctx)

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT -- few voices: the filter as chunks at once (1e-5 of the voice's peak); oscillator and
        envelope, and their states, stay exact."""
        cp = abi.NiceParams(params.sample_rate, 0, as_f32(params.freq), as_bool(params.note_on))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_TOLERANT if tolerant else 0)

    def paint_spans(self, span, outputs, temps, sample_rate, table, zero_first=False):
        """Render every voice's Trigger sub-spans of this buffer in one launch (zang_amd.spans.SpanTable)."""
        rc = self.lib.zh_nice_paint_spans(self.handle, span.start, span.end, _bufarray(outputs), _bufarray(temps),
                                          float(sample_rate), C.byref(table.c), abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD)
        abi.check(rc, "zh_nice_paint_spans")

    def paint_mix(self, span, mix, note_id_changed, params, zero_first=False, tolerant=False):
        """The fused chain followed by the voice mixdown into mix[frames] (device float32)."""
        cp = abi.NiceParams(params.sample_rate, 0, as_f32(params.freq), as_bool(params.note_on))
        rc = self.lib.zh_nice_paint_mix(self.handle, span.start, span.end, mix.data_ptr(), as_bool(note_id_changed),
                                        C.byref(cp), (abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD) | (abi.PAINT_TOLERANT if tolerant else 0))
        abi.check(rc, "zh_nice_paint_mix")

    def paint_mix_stereo(self, span, mix_left, mix_right, gain_left, gain_right, note_id_changed, params, zero_first=False, tolerant=False):
        """Two channels: mix_c[f] (+)= sum over voices of voice[f] * gain_c[voice] (examples/example_stereo.zig:92-98)."""
        cp = abi.NiceParams(params.sample_rate, 0, as_f32(params.freq), as_bool(params.note_on))
        rc = self.lib.zh_nice_paint_mix_stereo(self.handle, span.start, span.end, mix_left.data_ptr(), mix_right.data_ptr(),
                                               as_f32(gain_left), as_f32(gain_right), as_bool(note_id_changed), C.byref(cp),
                                               (abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD) | (abi.PAINT_TOLERANT if tolerant else 0))
        abi.check(rc, "zh_nice_paint_mix_stereo")

    def paint_mix_stereo_batch(self, span, mix_lefts, mix_rights, gain_left, gain_right, note_id_changeds, paramses, zero_first=False, tolerant=False):
        """len(paramses) consecutive paint_mix_stereo calls in ONE launch (zh_nice_paint_mix_stereo_batch): call b with
        paramses[b] / note_id_changeds[b] into mix_lefts[b] / mix_rights[b]; same bits as the separate calls.
        `tolerant`: the kernel with multiply-adds fused (within 1e-5 of the voices' peaks, not the reference's bits)."""
        n = len(paramses)
        assert n == len(mix_lefts) == len(mix_rights) == len(note_id_changeds) and n <= 16
        ls = (C.c_void_p * n)(*[t.data_ptr() for t in mix_lefts])
        rs = (C.c_void_p * n)(*[t.data_ptr() for t in mix_rights])
        nics = (abi.Bool * n)(*[as_bool(x) for x in note_id_changeds])
        cps = (abi.NiceParams * n)(*[abi.NiceParams(p.sample_rate, 0, as_f32(p.freq), as_bool(p.note_on)) for p in paramses])
        rc = self.lib.zh_nice_paint_mix_stereo_batch(self.handle, span.start, span.end, n, ls, rs, as_f32(gain_left), as_f32(gain_right),
                                                     nics, cps, (abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD) | (abi.PAINT_TOLERANT if tolerant else 0))
        abi.check(rc, "zh_nice_paint_mix_stereo_batch")


class SimpleDelay(_Module):
    """examples/modules.zig:341-386 over zang.Delay(delay_samples) (src/zang/delay.zig)."""
    _prefix = "delay"

    @dataclass
    class Params:
        input: Any

    def __init__(self, n_voices, delay_samples, ctx=None):
        self.delay_samples = int(delay_samples)
        super().__init__(n_voices, ctx, C.c_uint32(self.delay_samples))

    @classmethod
    def init(cls, n_voices, delay_samples, ctx=None):
        return cls(n_voices, delay_samples, ctx)

    def reset(self):
        abi.check(self.lib.zh_delay_reset(self.handle), "zh_delay_reset")

    def state(self):
        rings = np.zeros((self.n_voices, self.delay_samples), np.float32)
        index = np.zeros(self.n_voices, np.uint32)
        abi.check(self.lib.zh_delay_get_state(self.handle, rings.ctypes.data, index.ctypes.data), "get_state")
        return rings, index

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False):
        self._paint(span, outputs, temps, note_id_changed, abi.DelayParams(as_buf(params.input)), zero_first)


class FilteredEchoes(_Module):
    """examples/modules.zig:390-461."""
    _prefix = "filtered_echoes"
    num_temps = 2

    @dataclass
    class Params:
        input: Any
        feedback_volume: Any
        cutoff: Any

    def __init__(self, n_voices, delay_samples, ctx=None):
        self.delay_samples = int(delay_samples)
        super().__init__(n_voices, ctx, C.c_uint32(self.delay_samples))

    @classmethod
    def init(cls, n_voices, delay_samples, ctx=None):
        return cls(n_voices, delay_samples, ctx)

    def reset(self):
        abi.check(self.lib.zh_filtered_echoes_reset(self.handle), "reset")

    def state(self):
        rings = np.zeros((self.n_voices, self.delay_samples), np.float32)
        index = np.zeros(self.n_voices, np.uint32)
        flt = np.zeros(self.n_voices, dtype=np.dtype(abi.FilterState))
        abi.check(self.lib.zh_filtered_echoes_get_state(self.handle, rings.ctypes.data, index.ctypes.data, flt.ctypes.data), "get_state")
        return rings, index, flt

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT -- few voices: pieces of <= delay_samples frames, the filter of each as chunks at once
        (1e-5 of the voice's peak; include/zang_hip.h)."""
        cp = abi.FilteredEchoesParams(as_buf(params.input), as_f32(params.feedback_volume), as_f32(params.cutoff))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_TOLERANT if tolerant else 0)


class NoiseFilter(_Module):
    """Noise -> Filter as one fused kernel (examples/example_stereo.zig:71-82; BASELINE config 3)."""
    _prefix = "noise_filter"
    _state_ctype = abi.NoiseFilterState
    num_temps = 1

    @dataclass
    class Params:
        color: int
        type: int
        cutoff: Any
        res: Any

    def __init__(self, n_voices, ctx=None, first_seed=0):
        super().__init__(n_voices, ctx, C.c_uint64(first_seed))

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT (opt-in, 1e-5 of the signal's peak instead of bits; include/zang_hip.h)."""
        cp = abi.NoiseFilterParams(params.color, params.type, as_f32(params.cutoff), as_f32(params.res))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_TOLERANT if tolerant else 0)


class PMOscInstrument(_Module):
    """examples/modules.zig:80-128 (PhaseModOscillator :6-77 inside) as one fused kernel."""
    _prefix = "pmosc"
    _state_ctype = abi.PMOscState
    num_temps = 3

    @dataclass
    class Params:
        sample_rate: float
        freq: Any
        note_on: Any

    def __init__(self, n_voices, release_duration, ctx=None):
        self._rel = as_f32(release_duration)
        super().__init__(n_voices, ctx, self._rel)

    @classmethod
    def init(cls, n_voices, release_duration, ctx=None):
        return cls(n_voices, release_duration, ctx)

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """tolerant=True: ZH_PAINT_TOLERANT -- the carrier's sine in f32; phases and envelope state stay exact."""
        cp = abi.PMOscParams(params.sample_rate, 0, as_f32(params.freq), as_bool(params.note_on))
        self._paint(span, outputs, temps, note_id_changed, cp, zero_first, abi.PAINT_TOLERANT if tolerant else 0)

    def paint_spans(self, span, outputs, temps, sample_rate, table, zero_first=False):
        """Render every voice's Trigger sub-spans of this buffer in one launch (zang_amd.spans.SpanTable)."""
        rc = self.lib.zh_pmosc_paint_spans(self.handle, span.start, span.end, _bufarray(outputs), _bufarray(temps),
                                           float(sample_rate), C.byref(table.c), abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD)
        abi.check(rc, "zh_pmosc_paint_spans")
