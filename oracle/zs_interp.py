"""CPU execution of a compiled zangscript module, one voice at a time, exactly as the Zig code printed
by the reference's backend would run it (src/zangscript/codegen_zig.zig:110-457): whole-span buffer
operations over numbered temps, builtin modules through the oracle's paint functions.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): the checker for the fused HIP kernels the
zangscript backend generates.  Parity unpinned at this level (no Zig toolchain to run the generated
Zig); the front-end that produces the instruction list IS pinned by the reference's golden text
(tests/test_zangscript.py)."""
import ctypes as C

import numpy as np

from . import pyoracle as po

F32 = np.float32
_ENUM_INDEX = {}


def _enum_index(enum, label):
    return [v.label for v in enum.values].index(label)


def note_tracker_consume(st, times, sample_rate, start, end):
    """NoteTracker.consume (src/zang/notes.zig:161-205) in f32: st = {"next", "t"}; returns the impulses
    [(frame, note index)] of the notes that fall into this span (a 33rd impulse is dropped: the reference's
    arrays hold 32) as (frame, note id, note index)."""
    out_len = end - start
    buf_time = F32(F32(out_len) / F32(sample_rate))
    t0 = F32(st["t"])
    end_t = F32(t0 + buf_time)
    impulses = []
    while st["next"] < len(times):
        note_t = F32(times[st["next"]])
        if not (note_t < end_t):
            break
        with np.errstate(all="ignore"):
            f = F32(F32(note_t - t0) / buf_time)
            x = F32(f * F32(out_len))
        rel = 0 if not (x == x) or x <= 0 else (0xFFFFFFFF if x >= 4294967296.0 else int(x))   # saturating, like the device
        rel = min(rel, out_len - 1)
        if len(impulses) < 32:
            impulses.append((start + rel, st["next"] + 1, st["next"]))   # note id = index + 1 (codegen_zig.zig:497)
        st["next"] += 1
    st["t"] = end_t
    return impulses


def trigger_spans(cur, impulses, start, end):
    """Trigger.counter + next() until null (src/zang/trigger.zig:66-195).  cur = the trigger's note: None
    or (note id, params); impulses = [(frame, note id, params)].  Returns
    ([(start, end, params, note_id_changed)], cur)."""
    spans, ii, pos = [], 0, start
    while pos < end:
        span_end, note, have = end, None, False
        if cur is not None:                                      # carryOver (:108-142)
            if ii < len(impulses):
                if impulses[ii][0] > pos:
                    have, span_end, note = True, min(end, impulses[ii][0]), cur
            else:
                have, note = True, cur
        if not have:                                             # getNextNoteSpan (:144-195)
            for i in range(ii, len(impulses)):
                fr = impulses[i][0]
                if fr >= end:
                    break
                if fr > pos:
                    span_end = fr                                # gap before the note begins
                    break
                ii += 1
                end_c = min(end, impulses[i + 1][0]) if i + 1 < len(impulses) else end
                if end_c <= pos:
                    continue                                     # the next impulse starts at the same time
                span_end, note = end_c, (impulses[i][1], impulses[i][2])
                break
        if note is not None:
            spans.append((pos, span_end, note[1], cur is None or note[0] != cur[0]))
            cur = note
        pos = span_end
    return spans, cur


class Instance:
    """One voice of script module `module_index`: the fields of the generated struct (init():
    codegen_zig.zig:542-556) and paint()."""

    def __init__(self, script, module_index, seeds):
        self.s, self.mi = script, module_index
        self.L = po.lib()
        self.mr = script.module_results[module_index]
        self.module = script.modules[module_index]
        # zang.Delay(n).init(): zeroed ring, index 0 (src/zang/delay.zig:12-17)
        self.delays = [[np.zeros(n, F32), 0] for n in self.mr.delays]
        # NoteTracker.init(&track.notes) / Trigger.init() (codegen_zig.zig:549-554)
        self.trackers = [{"next": 0, "t": F32(0), "cur": None} for _ in self.mr.note_trackers]
        self.track_params = None
        self.fields = []
        for callee_index in self.mr.fields:                      # init order = field order, depth first
            callee = script.modules[callee_index]
            if callee.scope is not None:
                self.fields.append(Instance(script, callee_index, seeds))
                continue
            n = callee.builtin_name
            L = self.L
            if n == "SineOsc":
                st = po.SineOsc(); L.zo_sineosc_init(C.byref(st))
            elif n == "PulseOsc":
                st = po.PulseOsc(); L.zo_pulseosc_init(C.byref(st))
            elif n == "TriSawOsc":
                st = po.TriSawOsc(); L.zo_trisawosc_init(C.byref(st))
            elif n == "Noise":
                st = po.Noise(); L.zo_noise_init(C.byref(st), next(seeds))
            elif n == "Envelope":
                st = po.Envelope(); L.zo_envelope_init(C.byref(st))
            elif n == "Filter":
                st = po.Filter(); L.zo_filter_init(C.byref(st))
            elif n == "Decimator":
                st = po.Decimator(); L.zo_decimator_init(C.byref(st))
            elif n == "Cycle":
                st = po.Cycle(); L.zo_cycle_init(C.byref(st))
            elif n == "Portamento":
                st = po.Portamento(); L.zo_portamento_init(C.byref(st))
            elif n == "Curve":
                st = po.CurveModule(); L.zo_curve_init(C.byref(st))
            else:
                st = None                                        # Gate, Distortion: stateless
            self.fields.append(st)

    # ---- values
    def _val(self, r, temps, floats, params):
        k = r.kind
        if k == "temp_buffer":
            return temps[r.index]
        if k == "temp_float":
            return floats[r.index]
        if k == "literal_number":
            return F32(r.value.value)
        if k == "literal_boolean":
            return bool(r.value)
        if k == "literal_enum_value":
            return (r.value, self._val(r.payload, temps, floats, params) if r.payload is not None else None)
        if k == "literal_curve":
            return [(t.value, v.value) for t, v in self.s.curves[r.index].points]
        if k == "self_param":
            return params[r.index]
        if k == "track_param":
            return self.track_params[r.index]
        raise NotImplementedError(k)

    @staticmethod
    def _cob(v):
        return po.buffer(v) if isinstance(v, np.ndarray) else po.constant(float(v))

    @staticmethod
    def _curve(enum, v):
        label, payload = v
        return po.curve(_enum_index(enum, label), float(payload) if payload is not None else 0.0)

    def _un(self, op, x):
        L = self.L
        if isinstance(x, np.ndarray):
            if op in ("sin", "cos"):
                src = np.ascontiguousarray(x)
                out = np.empty_like(src)
                getattr(L, "zo_math_%sf_n" % op)(po.fptr(src), po.fptr(out), len(src))      # (x, y, n)
                return out
            return {"abs": np.abs, "neg": np.negative, "sqrt": np.sqrt}[op](x)
        x = F32(x)
        if op == "sin":
            return F32(L.zo_math_sinf(x))
        if op == "cos":
            return F32(L.zo_math_cosf(x))
        return F32({"abs": abs, "neg": lambda t: -t, "sqrt": np.sqrt}[op](x))

    def _bin(self, op, a, b):
        """sub/div/pow/max/min on f32 scalars or arrays (elementwise, one IEEE rounding each)."""
        if op == "sub":
            return a - b
        if op == "div":
            with np.errstate(all="ignore"):
                return a / b
        if op in ("max", "min"):                                  # std.math.max / min: comparison selects
            c = (a > b) if op == "max" else (a < b)
            if isinstance(c, np.ndarray):
                return np.where(c, a, b).astype(F32)
            return a if c else b
        if op == "pow":
            if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
                n = len(a) if isinstance(a, np.ndarray) else len(b)
                aa = a if isinstance(a, np.ndarray) else np.full(n, a, F32)
                bb = b if isinstance(b, np.ndarray) else np.full(n, b, F32)
                return np.array([self.L.zo_math_powf(float(x), float(y)) for x, y in zip(aa, bb)], F32)
            return F32(self.L.zo_math_powf(float(a), float(b)))
        if op == "add":
            return a + b
        if op == "mul":
            return a * b
        raise AssertionError(op)

    # ---- paint
    def paint(self, start, end, out, nic, params):
        """params: list in declaration order (sample_rate first): f32 scalar, bool, np.float32 array
        (buffer / cob buffer), (label, payload) enum tuple, [(t, value)] curve."""
        temps = [np.zeros(len(out), F32) for _ in range(self.mr.num_temps)]
        self._run(self.mr.instructions, start, end, out, nic, params, temps, {})

    def _run(self, instructions, start, end, out, nic, params, temps, floats):
        L = self.L
        sl = slice(start, end)

        def dest(d):
            return out if d.kind == "output" else temps[d.index]

        def store(d, values):                                    # the explicit `while` loops of the generated Zig
            t = dest(d)
            if d.kind == "output":
                t[sl] = t[sl] + values
            else:
                t[sl] = values

        for ins in instructions:
            k = ins.kind
            if k == "delay":                                     # codegen_zig.zig:391-455
                d = dest(ins.out)
                if ins.out.kind != "output":
                    L.zo_zero(start, end, po.fptr(d))
                ring, n = self.delays[ins.delay_index][0], len(self.delays[ins.delay_index][0])
                fb, fbout = temps[ins.feedback_temp], temps[ins.feedback_out_temp]
                pos = start
                while pos < end:
                    fbout[pos:end] = 0
                    fb[pos:end] = 0
                    cnt = min(end - pos, n)                      # readDelayBuffer (delay.zig:28-57): `+=`, wraps
                    idx = self.delays[ins.delay_index][1]
                    for j in range(cnt):
                        fb[pos + j] = fb[pos + j] + ring[(idx + j) % n]
                    self._run(ins.instructions, pos, pos + cnt, out, nic, params, temps, floats)   # body dests are explicit
                    for j in range(cnt):                         # writeDelayBuffer (:62-89)
                        ring[(idx + j) % n] = fbout[pos + j]
                    self.delays[ins.delay_index][1] = (idx + cnt) % n
                    pos += cnt
                continue
            if k == "track_call":                                # codegen_zig.zig:359-389
                trk = self.trackers[ins.note_tracker_index]
                track = self.s.tracks[ins.track_index]
                note_on = [i for i, p in enumerate(self.module.params) if p.name == "note_on"]
                reset = bool(params[note_on[0]] and nic) if note_on else bool(nic)
                if reset:
                    trk["next"], trk["t"], trk["cur"] = 0, F32(0), None
                speed = F32(self._val(ins.speed, temps, floats, params))
                with np.errstate(all="ignore"):
                    sr = F32(F32(params[0]) / speed)
                impulses = note_tracker_consume(trk, [n.t.value for n in track.notes], sr, start, end)
                spans, trk["cur"] = trigger_spans(trk["cur"], impulses, start, end)
                for (s0, s1, note, changed) in spans:
                    self.track_params = [self._val(r, temps, floats, params) for r in self.s.track_results[ins.track_index][note]]
                    self._run(ins.instructions, s0, s1, out, reset or changed, params, temps, floats)
                self.track_params = None
                continue
            if k in ("copy_buffer", "float_to_buffer", "cob_to_buffer"):
                src = params[ins.in_self_param] if k == "cob_to_buffer" else self._val(ins.src, temps, floats, params)
                d = dest(ins.out)
                if isinstance(src, np.ndarray):
                    (L.zo_add_into if ins.out.kind == "output" else L.zo_copy)(start, end, po.fptr(d), po.fptr(src))
                else:
                    (L.zo_add_scalar_into if ins.out.kind == "output" else L.zo_set)(start, end, po.fptr(d), float(src))
            elif k == "arith_float":
                floats[ins.out] = self._un(ins.op, self._val(ins.a, temps, floats, params))
            elif k == "arith_float_float":
                floats[ins.out] = F32(self._bin(ins.op, F32(self._val(ins.a, temps, floats, params)), F32(self._val(ins.b, temps, floats, params))))
            elif k == "arith_buffer":
                store(ins.out, self._un(ins.op, self._val(ins.a, temps, floats, params)[sl].copy()))
            elif k in ("arith_float_buffer", "arith_buffer_float", "arith_buffer_buffer"):
                a, b = self._val(ins.a, temps, floats, params), self._val(ins.b, temps, floats, params)
                if ins.op in ("add", "mul"):
                    d = dest(ins.out)
                    if ins.out.kind != "output":
                        L.zo_zero(start, end, po.fptr(d))
                    if k == "arith_buffer_buffer":
                        (L.zo_add if ins.op == "add" else L.zo_multiply)(start, end, po.fptr(d), po.fptr(a), po.fptr(b))
                    else:
                        buf, flt = (b, a) if k == "arith_float_buffer" else (a, b)
                        (L.zo_add_scalar if ins.op == "add" else L.zo_multiply_scalar)(start, end, po.fptr(d), po.fptr(buf), float(flt))
                else:
                    aa = a[sl] if isinstance(a, np.ndarray) else F32(a)
                    bb = b[sl] if isinstance(b, np.ndarray) else F32(b)
                    store(ins.out, np.asarray(self._bin(ins.op, aa, bb), F32))
            elif k == "call":
                callee_index = self.mr.fields[ins.field_index]
                callee = self.s.modules[callee_index]
                args = [self._val(r, temps, floats, params) for r in ins.args]
                d = dest(ins.out)
                if ins.out.kind != "output":
                    L.zo_zero(start, end, po.fptr(d))
                field = self.fields[ins.field_index]
                if callee.scope is not None:
                    field.paint(start, end, d, nic, args)
                else:
                    self._builtin(callee, field, start, end, d, nic, dict(zip([p.name for p in callee.params], args)))
            else:
                raise NotImplementedError(k)

    def _builtin(self, callee, st, start, end, d, nic, a):
        L, n = self.L, callee.builtin_name
        o = po.fptr(d)
        sr = float(a.get("sample_rate", 0.0))
        enum = {p.name: p.param_type.enum for p in callee.params}
        if n == "SineOsc":
            L.zo_sineosc_paint(C.byref(st), start, end, o, sr, self._cob(a["freq"]), self._cob(a["phase"]))
        elif n == "PulseOsc":
            L.zo_pulseosc_paint(C.byref(st), start, end, o, sr, self._cob(a["freq"]), float(a["color"]))
        elif n == "TriSawOsc":
            L.zo_trisawosc_paint(C.byref(st), start, end, o, sr, self._cob(a["freq"]), float(a["color"]))
        elif n == "Noise":
            L.zo_noise_paint(C.byref(st), start, end, o, _enum_index(enum["color"], a["color"][0]))
        elif n == "Envelope":
            p = po.EnvelopeParams(sr, self._curve(enum["attack"], a["attack"]), self._curve(enum["decay"], a["decay"]),
                                  self._curve(enum["release"], a["release"]), float(a["sustain_volume"]), 1 if a["note_on"] else 0)
            L.zo_envelope_paint(C.byref(st), start, end, o, 1 if nic else 0, C.byref(p))
        elif n == "Gate":
            L.zo_gate_paint(start, end, o, 1 if a["note_on"] else 0)
        elif n == "Filter":
            L.zo_filter_paint(C.byref(st), start, end, o, po.fptr(a["input"]), _enum_index(enum["type"], a["type"][0]),
                              self._cob(a["cutoff"]), self._cob(a["res"]))
        elif n == "Decimator":
            L.zo_decimator_paint(C.byref(st), start, end, o, sr, po.fptr(a["input"]), float(a["fake_sample_rate"]))
        elif n == "Distortion":
            L.zo_distortion_paint(start, end, o, po.fptr(a["input"]), _enum_index(enum["type"], a["type"][0]),
                                  float(a["ingain"]), float(a["outgain"]), float(a["offset"]))
        elif n == "Cycle":
            L.zo_cycle_paint(C.byref(st), start, end, o, sr, self._cob(a["speed"]))
        elif n == "Portamento":
            L.zo_portamento_paint(C.byref(st), start, end, o, 1 if nic else 0, sr, self._curve(enum["curve"], a["curve"]),
                                  float(a["goal"]), 1 if a["note_on"] else 0, 1 if a["prev_note_on"] else 0)
        elif n == "Curve":
            nodes = (po.CurveNode * max(len(a["curve"]), 1))(*[po.CurveNode(float(v), float(t)) for t, v in a["curve"]])
            L.zo_curve_paint(C.byref(st), start, end, o, 1 if nic else 0, sr, _enum_index(enum["function"], a["function"][0]),
                             nodes, len(a["curve"]))
        else:
            raise NotImplementedError(n)


def noise_field_count(script, module_index):
    n = 0
    for f in script.module_results[module_index].fields:
        m = script.modules[f]
        n += noise_field_count(script, f) if m.scope is not None else (1 if m.builtin_name == "Noise" else 0)
    return n


def make_voices(script, name, n_voices, first_seed=0):
    """n_voices instances created voice by voice: Noise seeds first_seed + v*K + k (Noise.zig:25-29)."""
    mi = script.module_index(name)
    K = noise_field_count(script, mi)
    voices = []
    for v in range(n_voices):
        seeds = iter(range(first_seed + v * K, first_seed + (v + 1) * K))
        voices.append(Instance(script, mi, seeds))
    return voices
