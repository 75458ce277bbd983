"""Host-side mirror of src/zang/notes.zig and src/zang/trigger.zig over the C ABI's scheduling
entry points (include/zang_hip.h, "event scheduling").  Like the Zig generics, everything is
parameterised by the note-params type: a ctypes type (c_float, or a Structure with a `note_on`
field for the PolyphonyDispatcher).  No GPU is needed for any of this.
"""
import ctypes as C
from dataclasses import dataclass

from . import abi

Impulse = abi.Impulse


@dataclass
class ImpulsesAndParamses:
    """Notes(P).ImpulsesAndParamses (notes.zig:66-70) as two Python lists (copies)."""
    impulses: list
    paramses: list

    def __len__(self):
        return len(self.impulses)


def _to_c(params_type, iap):
    n = len(iap.impulses)
    imps = (abi.Impulse * max(n, 1))(*[abi.Impulse(i.frame, i.note_id, i.event_id) for i in iap.impulses])
    pars = (params_type * max(n, 1))(*iap.paramses)
    c = abi.Iap(imps, C.cast(pars, C.c_void_p), n)
    c._keep = (imps, pars)
    return c


def _from_c(params_type, c):
    n = int(c.len)
    imps = [abi.Impulse(c.impulses[i].frame, c.impulses[i].note_id, c.impulses[i].event_id) for i in range(n)]
    psize = C.sizeof(params_type)
    pars = []
    for i in range(n):
        v = params_type.from_buffer_copy(C.string_at(c.paramses + i * psize, psize))
        pars.append(v.value if hasattr(v, "value") else v)
    return ImpulsesAndParamses(imps, pars)


def _value(params_type, p):
    return p if isinstance(p, params_type) else params_type(p)


class _Handle:
    _destroy = None

    def __init__(self):
        self.lib = abi.load()
        self.handle = C.c_void_p()

    def close(self):
        if self.handle:
            getattr(self.lib, self._destroy)(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def Notes(params_type):
    """zang.Notes(NoteParamsType) (notes.zig:64)."""
    psize = C.sizeof(params_type)

    class ImpulseQueue(_Handle):          # notes.zig:72-128
        _destroy = "zh_impulse_queue_destroy"

        def __init__(self):
            super().__init__()
            abi.check(self.lib.zh_impulse_queue_create(psize, C.byref(self.handle)), "zh_impulse_queue_create")

        init = classmethod(lambda cls: cls())

        def push(self, impulse_frame, note_id, params):
            p = _value(params_type, params)
            abi.check(self.lib.zh_impulse_queue_push(self.handle, impulse_frame, note_id, C.byref(p)), "push")

        def consume(self):
            c = abi.Iap()
            abi.check(self.lib.zh_impulse_queue_consume(self.handle, C.byref(c)), "consume")
            return _from_c(params_type, c)

    @dataclass
    class SongEvent:                      # notes.zig:130-134
        params: object
        t: float
        note_id: int

    class NoteTracker(_Handle):           # notes.zig:138-207
        _destroy = "zh_note_tracker_destroy"

        def __init__(self, song):
            super().__init__()
            n = len(song)
            pars = (params_type * max(n, 1))(*[_value(params_type, e.params) for e in song])
            ts = (C.c_float * max(n, 1))(*[e.t for e in song])
            ids = (C.c_uint64 * max(n, 1))(*[e.note_id for e in song])
            abi.check(self.lib.zh_note_tracker_create(psize, n, C.cast(pars, C.c_void_p), ts, ids, C.byref(self.handle)), "zh_note_tracker_create")

        init = classmethod(lambda cls, song: cls(song))

        def reset(self):
            abi.check(self.lib.zh_note_tracker_reset(self.handle), "reset")

        def consume(self, sample_rate, span):
            c = abi.Iap()
            abi.check(self.lib.zh_note_tracker_consume(self.handle, sample_rate, span.start, span.end, C.byref(c)), "consume")
            return _from_c(params_type, c)

    def PolyphonyDispatcher(polyphony):   # notes.zig:209-349
        note_on_offset = getattr(params_type, "note_on").offset

        class _PD(_Handle):
            _destroy = "zh_polyphony_dispatcher_destroy"

            def __init__(self):
                super().__init__()
                abi.check(self.lib.zh_polyphony_dispatcher_create(polyphony, psize, note_on_offset, C.byref(self.handle)), "create")

            init = classmethod(lambda cls: cls())

            def reset(self):
                abi.check(self.lib.zh_polyphony_dispatcher_reset(self.handle), "reset")

            def dispatch(self, iap):
                out = (abi.Iap * polyphony)()
                abi.check(self.lib.zh_polyphony_dispatcher_dispatch(self.handle, _to_c(params_type, iap), out), "dispatch")
                return [_from_c(params_type, out[i]) for i in range(polyphony)]

        return _PD

    ns = type("Notes", (), {})
    ns.ImpulsesAndParamses = ImpulsesAndParamses
    ns.ImpulseQueue = ImpulseQueue
    ns.SongEvent = SongEvent
    ns.NoteTracker = NoteTracker
    ns.PolyphonyDispatcher = staticmethod(PolyphonyDispatcher)
    return ns


def Trigger(params_type):
    """zang.Trigger(ParamsType) (trigger.zig:26)."""
    psize = C.sizeof(params_type)

    @dataclass
    class NewPaintReturnValue:            # trigger.zig:50-54
        span: object
        params: object
        note_id_changed: bool

    class _Trigger(_Handle):
        _destroy = "zh_trigger_destroy"

        def __init__(self):
            super().__init__()
            abi.check(self.lib.zh_trigger_create(psize, C.byref(self.handle)), "zh_trigger_create")

        init = classmethod(lambda cls: cls())

        def reset(self):
            abi.check(self.lib.zh_trigger_reset(self.handle), "reset")

        def counter(self, span, iap):
            """Returns an opaque counter; the Trigger keeps the impulse arrays alive until the next counter()."""
            self._iap = _to_c(params_type, iap)
            abi.check(self.lib.zh_trigger_counter(self.handle, span.start, span.end, self._iap), "counter")
            return self

        def next(self, ctr=None):
            from .zang import Span
            out = abi.PaintSpan()
            rc = self.lib.zh_trigger_next(self.handle, C.byref(out))
            if rc < 0:
                abi.check(rc, "zh_trigger_next")
            if rc == 0:
                return None
            p = params_type.from_buffer_copy(bytes(out.params)[:psize])
            return NewPaintReturnValue(Span(int(out.start), int(out.end)), p.value if hasattr(p, "value") else p,
                                       bool(out.note_id_changed))

    _Trigger.NewPaintReturnValue = NewPaintReturnValue
    return _Trigger
