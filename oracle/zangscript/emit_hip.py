"""zangscript -> HIP: one fused lane-per-voice kernel per exported module.

The reference's backend prints Zig whose paint() is a list of whole-span buffer operations over
numbered temps (codegen_zig.zig:110-457).  Every one of those operations is elementwise in the frame
index, or a module whose output at frame i reads its input buffers at frame i only, so the same list
can be evaluated one frame at a time with every temp buffer held in a register: lane = voice, the
per-voice recurrences (oscillator phase, filter, envelope) stay in VGPRs, nothing but the module's
output and its waveform/cob parameters touches HBM.  What does not depend on the frame -- temp floats,
per-paint prologues and epilogues of the builtin modules -- is hoisted out of the frame loop.

To keep the reference's bits the emitted code repeats its operation order literally:
  temps are zeroed and then `+=`-ed (`zang.zero` + `zang.multiplyScalar` is `0.0f + a*s`), outputs only
  `+=`, sub/div/pow/min/max and the unary functions assign, float (op) buffer swaps its operands for
  add/mul (codegen_zig.zig:205-206); and the library is compiled with contraction off.

Builtin modules come from csrc/voices.hip.h (the same lane objects the standalone kernels use).  Script
modules calling script modules are inlined.  `delay` keeps its ring in the per-voice state blob; a track
call (`from ... begin`) keeps NoteTracker + Trigger per voice and walks their sub-spans.  What the backend
cannot express (more than 16 params, a zero-sample delay) raises HipBackendError, reported per module."""
import os
import re
from dataclasses import dataclass

from .errors import ScriptError


class HipBackendError(Exception):
    pass


@dataclass
class Val:
    """A value as the emitted code sees it.  kind: buf (an expression valid inside the frame body),
    float / bool (a per-paint constant), enum (tag + optional payload), curve (pointer + count)."""
    kind: str
    expr: str = ""
    tag: object = None             # enum: an int when known at compile time, else a C++ expression
    payload: "Val" = None          # enum payload (float Val) or None
    count: str = ""                # curve node count expression
    enum: object = None            # BuiltinEnum
    computed: bool = False         # buf: derives from a module's output or a transcendental function (not just params / constants / + - * /)
    cob_b: str = ""                # buf that is exactly a constant_or_buffer param's value: its "is a buffer" flag ...
    cob_c: str = ""                # ... and its constant
    sines: int = 0                 # buf: the sine sources (bit per SineOsc call / sin(), _Kernel.nsines) whose results flow into it


def f32_literal(x):
    if x != x:
        return "__builtin_nanf(\"\")"
    if x in (float("inf"), float("-inf")):
        return "%s__builtin_inff()" % ("-" if x < 0 else "")
    return float.hex(float(x)) + "f"


# 32-bit state words per voice of each builtin, and how the lane object is loaded / stored
STATE_WORDS = {"SineOsc": 1, "PulseOsc": 1, "TriSawOsc": 2, "Noise": 8, "Envelope": 4, "Gate": 0, "Filter": 2,
               "Decimator": 2, "Distortion": 0, "Cycle": 1, "Portamento": 3, "Curve": 4}
TRACK_WORDS = 3                       # NoteTracker {next_song_event, t} + Trigger {note}
# the params of a builtin that its state recurrence reads every frame
STATE_INPUTS = {"SineOsc": ("freq",), "PulseOsc": ("freq",), "TriSawOsc": ("freq",), "Cycle": ("speed",),
                "Filter": ("input", "cutoff", "res"), "Decimator": ("input",)}
# ZH_PAINT_TOLERANT (include/zang_hip.h): a sine may be evaluated in f32 (zmath.hip.h zsinf_tol, within 2.4e-7 of musl's) when its
# error can only be scaled and added on its way to the output -- through + - * neg abs min max, copies, a Filter's or a Decimator's
# `input`, a delay ring written (not read: a ring's content is of unknown origin, ALL_SINES).  Every other place a value can go is a
# SINK that keeps the sines reaching it exact: any other builtin param (an oscillator's freq / phase: the error would be integrated
# or -- PMOscInstrument, profiles/r04/NOTES.md 5a -- multiplied by the carrier's slope at a large argument; a Distortion's input: gain up to
# 64), the argument of sin / cos / sqrt, a divisor, both operands of pow.
LINEAR_INPUTS = {("Filter", "input"), ("Decimator", "input")}
ALL_SINES = (1 << 64) - 1
MAX_SINES = 63                        # sources beyond this many in one kernel stay exact


class _Kernel:
    """Everything collected while one exported module is walked."""

    def __init__(self, name):
        self.name = name
        self.params, self.pro, self.frame, self.epi_ends, self.epi_stores, self.init = [], [], [], [], [], []
        self.tracks = set()            # track indices whose tables the kernel reads
        self.temps = []                # frame-scope float variables
        self.rows = []                 # exported param index of every frame-loop input row
        self.words = 0
        self.noise_fields = 0
        self.uid = 0
        self.rings = False             # a delay ring lives in the state blob and is read and written inside the frame body
        self.walk_reads_computed = False   # a builtin's frame-to-frame state is fed by a value computed in the frame body
        self.quiet_terms = []          # wave-uniform tests over the next `zs_n` frames: the chunk may run the body's ZS_Q forms
        self.nsines = 0                # sine sources met so far (SineOsc calls and sin() of a buffer)
        self.exact_sines = 0           # ... and the ones that reach a sink (LINEAR_INPUTS above)

    def fresh(self, stem):
        self.uid += 1
        return "%s%d" % (stem, self.uid)

    def alloc(self, n):
        w = self.words
        self.words += n
        return w

    def sine_source(self):
        """-> (id, bit) of a new sine source; (None, 0) = one too many: emitted exact"""
        if self.nsines >= MAX_SINES:
            return None, 0
        self.nsines += 1
        return self.nsines - 1, 1 << (self.nsines - 1)

    def sink(self, v):
        if v.kind == "buf":
            self.exact_sines |= v.sines


class _ModuleCtx:
    """One (possibly inlined) instance of a script module."""

    def __init__(self, k, module_index, env, outvar, nic, prefix, parent=None):
        self.k, self.module_index, self.env, self.outvar, self.nic, self.prefix = k, module_index, env, outvar, nic, prefix
        self.tnames, self.fnames = {}, {}
        self.heavy = {}                # temp index -> its current value derives from a module output / transcendental (Val.computed)
        self.cobsrc = {}               # temp index -> (Val.cob_b, Val.cob_c) while it holds a cob param's value
        self.srcs = {}                 # temp index -> Val.sines of its current value (unknown temp: ALL_SINES)
        self.outsines = 0              # Val.sines of everything added to this module's output so far
        # where the per-paint prologue / epilogue of builtin calls goes: the kernel's own prologue and
        # epilogue, or -- inside a `delay` body, which the reference paints chunk by chunk -- the
        # chunk's; `rel` / `length` are the frame index within, and the length of, that paint call
        self.begin_sink = parent.begin_sink if parent else k.pro
        self.end_sink = parent.end_sink if parent else k.epi_ends
        self.rel = parent.rel if parent else "(i - L.start)"
        self.length = parent.length if parent else "SPAN_LEN"
        self.track = None              # inside `from <track> ... begin`: {param index: Val} of the running note

    def tname(self, i):
        if i not in self.tnames:
            self.tnames[i] = "%st%d" % (self.prefix, i)
            self.k.temps.append(self.tnames[i])
        return self.tnames[i]

    def fname(self, i):
        return "%sf%d" % (self.prefix, i)


class HipEmitter:
    def __init__(self, script):
        self.s = script

    # ---- values
    def val(self, mc, r):
        k = r.kind
        if k == "temp_buffer":
            cb, cc = mc.cobsrc.get(r.index, ("", ""))
            return Val("buf", mc.tname(r.index), computed=mc.heavy.get(r.index, True), cob_b=cb, cob_c=cc, sines=mc.srcs.get(r.index, ALL_SINES))
        if k == "temp_float":
            return Val("float", mc.fname(r.index))
        if k == "literal_number":
            return Val("float", f32_literal(r.value.value))
        if k == "literal_boolean":
            return Val("bool", "true" if r.value else "false")
        if k == "literal_enum_value":
            return Val("enum", tag=r.value, payload=self.val(mc, r.payload) if r.payload is not None else None)
        if k == "literal_curve":
            return Val("curve", "zs_curve%d" % r.index, count=str(len(self.s.curves[r.index].points)))
        if k == "self_param":
            return mc.env[r.index]
        if k == "track_param":
            return mc.track[r.index]
        raise AssertionError(k)

    @staticmethod
    def enum_tag(v, enum):
        """C++ expression for the index of an enum value in its declaration order."""
        if isinstance(v.tag, str) and v.kind == "enum" and v.enum is None:
            for i, ev in enumerate(enum.values):
                if ev.label == v.tag:
                    return str(i)
            raise AssertionError(v.tag)
        return v.tag                                           # runtime expression (exported param)

    @staticmethod
    def enum_payload(v):
        if v.payload is not None:
            return v.payload.expr
        return "0.0f"

    # ---- destinations
    @staticmethod
    def put(mc, d, expr, zero_first, heavy=False, sines=0):
        """`dest (+)= expr` with the reference's zeroing: temps are assigned (after zang.zero when the
        op accumulates), outputs accumulate.  `heavy`: Val.computed of the value written, `sines`: its Val.sines."""
        if d.kind == "temp":
            mc.heavy[d.index] = heavy
            mc.srcs[d.index] = sines
            mc.cobsrc.pop(d.index, None)
            t = mc.tname(d.index)
            if zero_first:
                return ["%s = 0.0f;" % t, "%s = %s + (%s);" % (t, t, expr)]
            return ["%s = %s;" % (t, expr)]
        mc.outsines |= sines
        return ["%s = %s + (%s);" % (mc.outvar, mc.outvar, expr)]

    UN = {"abs": "fabsf(%s)", "cos": "zcosf(%s)", "neg": "-(%s)", "sin": "zsinf(%s)", "sqrt": "sqrtf(%s)"}
    BIN = {"add": "(%s) + (%s)", "sub": "(%s) - (%s)", "mul": "(%s) * (%s)", "div": "(%s) / (%s)", "pow": "zpowf(%s, %s)",
           "max": "zs_max(%s, %s)", "min": "zs_min(%s, %s)"}

    # ---- builtin module calls
    def call_builtin(self, mc, ins, callee, args):
        k = mc.k
        name = callee.builtin_name
        a = {p.name: self.val(mc, r) for p, r in zip(callee.params, args)}
        # Launching the kernel as frame ranges pays only when replaying the state walk is cheap: not when a computed
        # buffer (an oscillator's output, a filtered signal ...) feeds a builtin's state -- an oscillator's or a cycle's
        # frequency, a filter's or a decimator's input.
        for pname in STATE_INPUTS.get(name, ()):
            if a[pname].kind == "buf" and a[pname].computed:
                k.walk_reads_computed = True
        out_sines = 0                                            # Val.sines of the module's output
        for pname, v in a.items():
            if (name, pname) in LINEAR_INPUTS:
                out_sines |= v.sines if v.kind == "buf" else 0
            else:
                k.sink(v)
        o = k.fresh("m")
        w = k.alloc(STATE_WORDS[name])
        decl, pro, frame = k.pro, mc.begin_sink, []      # lane object + state loads | per-paint prologue | per frame
        ends, epi = mc.end_sink, k.epi_stores            # per-paint epilogue | state stores at kernel end
        painted, value = None, None

        def ld_f(field, word):
            decl.append("%s.%s = zs_ld_f(L.state, %d, V, v);" % (o, field, word))
            epi.append("zs_st_f(L.state, %d, V, v, %s.%s);" % (word, o, field))

        def ld_u(field, word, cast=""):
            decl.append("%s.%s = %szs_ld_u(L.state, %d, V, v);" % (o, field, cast, word))
            epi.append("zs_st_u(L.state, %d, V, v, (uint32_t)%s.%s);" % (word, o, field))

        def cob(v):                     # (is_buffer, constant expr, per-frame expr)
            if v.kind == "buf":
                return True, "0.0f", v.expr
            return False, v.expr, v.expr

        if name == "SineOsc":
            fb, fc, fi = cob(a["freq"])
            pb, pc, pi = cob(a["phase"])
            decl.append("SineOscLane %s;" % o)
            ld_f("t", w)
            pro.append("%s.begin(%s, %s);" % (o, a["sample_rate"].expr, fc))
            # frequency and phase constant over the span: the chunks in which no voice can reach the sine's rare path run the
            # kernel's second frame body (zscript_emit.hip, the same lines)
            fv = a["freq"]
            quiet = (not pb) and mc.begin_sink is k.pro and ((not fb) or fv.cob_b != "")
            if quiet and not fb:
                k.quiet_terms.append("%s.small_args(%s, (float)zs_n)" % (o, pc))
            if quiet and fb:
                k.quiet_terms.append("(!%s && %s.small_args_step(%s * %s.inv_sr, %s, (float)zs_n))" % (fv.cob_b, o, fv.cob_c, o, pc))
            sid, bit = k.sine_source()
            out_sines |= bit
            mode = (", !ZS_Q" if quiet else "") if sid is None else "\x01O%d|%s\x02" % (sid, "!ZS_Q" if quiet else "1")   # resolved by generate()
            value = "%s.frame<%s%s>(%s, %s)" % (o, "true" if fb else "false", mode, fi if fb else "0.0f", pi)
            ends.append("%s.end();" % o)
        elif name == "Cycle":
            sb, sc, si = cob(a["speed"])
            decl.append("CycleLane %s;" % o)
            ld_f("t", w)
            pro.append("%s.begin(%s, %s);" % (o, a["sample_rate"].expr, sc))
            value = "%s.frame<%s>(%s)" % (o, "true" if sb else "false", si if sb else "0.0f")
        elif name in ("PulseOsc", "TriSawOsc"):
            fb, fc, fi = cob(a["freq"])
            lane = name + "Lane"
            decl.append("%s %s;" % (lane, o))
            ld_u("cnt", w)
            if name == "TriSawOsc":
                ld_f("t", w + 1)
            if fb:
                pro.append("%s.begin_ctrl(%s, %s);" % (o, a["sample_rate"].expr, a["color"].expr))
                if name == "PulseOsc":
                    cv, cp = k.fresh("cv"), k.fresh("cp")
                    frame += ["float %s = 0.0f;" % cv, "const bool %s = %s.frame_ctrl(%s, %s);" % (cp, o, fi, cv)]
                    painted, value = cp, cv
                else:
                    value = "%s.frame_ctrl(%s)" % (o, fi)
                    ends.append("%s.end_ctrl();" % o)
            else:
                pro.append("%s.begin_const(%s, %s, %s);" % (o, a["sample_rate"].expr, fc, a["color"].expr))
                cv, cp = k.fresh("cv"), k.fresh("cp")
                frame += ["float %s = 0.0f;" % cv, "const bool %s = %s.frame_const(%s);" % (cp, o, cv)]
                painted, value = cp, cv
        elif name == "Noise":
            decl.append("NoiseLane %s;" % o)
            for j in range(4):
                decl.append("%s.r.s%d = zs_ld_u64(L.state, %d, V, v);" % (o, j, w + 2 * j))
                epi.append("zs_st_u64(L.state, %d, V, v, %s.r.s%d);" % (w + 2 * j, o, j))
            pro.append("%s.begin();" % o)
            k.init.append(("noise", w, k.noise_fields))
            k.noise_fields += 1
            tag = self.enum_tag(a["color"], callee.params[0].param_type.enum)
            if tag in ("0", "1"):
                value = "%s.frame<%s>()" % (o, "true" if tag == "1" else "false")
            else:
                value = "((%s) == 1u ? %s.frame<true>() : %s.frame<false>())" % (tag, o, o)
        elif name == "Envelope":
            # three literal curves with one tag (not instantaneous): the tag-specialised lane, whose per-frame curve has no selects
            tags = [self.enum_tag(a[st], callee.params[1 + i].param_type.enum) for i, st in enumerate(("attack", "decay", "release"))]
            literal = all(isinstance(a[st].tag, str) and a[st].kind == "enum" and a[st].enum is None for st in ("attack", "decay", "release"))
            one_tag = literal and tags[0] == tags[1] == tags[2] and tags[0] != "0"
            decl.append("EnvLaneT<1, %s> %s;" % (tags[0], o) if one_tag else "EnvLane %s;" % o)
            ld_u("state", w)
            ld_f("t", w + 1)
            ld_f("last_value", w + 2)
            ld_f("start", w + 3)
            pro.append("%s.sample_rate = %s; %s.sustain_volume = %s; %s.note_on = %s;" %
                       (o, a["sample_rate"].expr, o, a["sustain_volume"].expr, o, a["note_on"].expr))
            for i, stage in enumerate(("attack", "decay", "release")):
                enum = callee.params[1 + i].param_type.enum
                pro.append("%s.%s = CurveP{(uint32_t)(%s), %s};" % (o, stage, self.enum_tag(a[stage], enum), self.enum_payload(a[stage])))
            pro.append("%s.begin(%s);" % (o, mc.nic))
            cv, cp = k.fresh("cv"), k.fresh("cp")
            # begin() only in the kernel's prologue (not per delay chunk / track sub-span): the frames a frame range
            # replays step the clock and the stage ends only (envelope.hip.h frame_walk)
            step = "frame_sq<ZS_Q>(%s, zs_walk)" % cv if mc.begin_sink is k.pro else "frame(%s)" % cv
            if mc.begin_sink is k.pro:
                k.quiet_terms.append("%s.quiet(zs_n)" % o)
            frame += ["float %s = 0.0f;" % cv, "const bool %s = %s.%s;" % (cp, o, step)]
            painted, value = cp, cv
        elif name == "Gate":
            painted, value = a["note_on"].expr, "1.0f"          # Gate.zig:28-30
        elif name == "Filter":
            cb, cc, ci = cob(a["cutoff"])
            rb, rc, ri = cob(a["res"])
            decl.append("FilterLane %s;" % o)
            ld_f("l", w)
            ld_f("b", w + 1)
            pro.append("%s.begin((uint32_t)(%s), %s, %s);" % (o, self.enum_tag(a["type"], callee.params[1].param_type.enum), cc, rc))
            value = "%s.frame<%s, %s>(%s, %s, %s)" % (o, "true" if cb else "false", "true" if rb else "false", a["input"].expr,
                                                      ci if cb else "0.0f", ri if rb else "0.0f")
        elif name == "Decimator":
            decl.append("DecimatorLane %s;" % o)
            ld_f("dval", w)
            ld_f("dcount", w + 1)
            k.init.append(("f32", w + 1, 1.0))                   # Decimator.zig:14-19
            pro.append("%s.begin(%s, %s);" % (o, a["sample_rate"].expr, a["fake_sample_rate"].expr))
            cv, cp = k.fresh("cv"), k.fresh("cp")
            frame += ["float %s = 0.0f;" % cv, "const bool %s = %s.frame(%s, %s);" % (cp, o, a["input"].expr, cv)]
            painted, value = cp, cv
            ends.append("%s.end();" % o)
        elif name == "Distortion":
            decl.append("DistortionLane %s;" % o)
            pro.append("%s.begin((uint32_t)(%s), %s, %s, %s);" % (o, self.enum_tag(a["type"], callee.params[1].param_type.enum),
                                                                  a["ingain"].expr, a["outgain"].expr, a["offset"].expr))
            value = "%s.frame(%s)" % (o, a["input"].expr)
        elif name == "Portamento":
            decl.append("PortamentoLane %s;" % o)
            ld_f("t", w)
            ld_f("last", w + 1)
            ld_f("st", w + 2)
            enum = callee.params[1].param_type.enum
            pro.append("%s.begin(%s, (uint32_t)(%s), %s, %s, %s, %s, %s);" % (
                o, a["sample_rate"].expr, self.enum_tag(a["curve"], enum), self.enum_payload(a["curve"]), a["goal"].expr,
                a["note_on"].expr, a["prev_note_on"].expr, mc.nic))
            value = "%s.frame()" % o
        elif name == "Curve":
            decl.append("CurveLane %s;" % o)
            decl.append("CurveTable %s_tb;" % o)
            ld_f("t", w)
            ld_u("cur", w + 1)
            ld_u("off", w + 2, "(int32_t)")
            ld_u("next", w + 3)
            pro.append("%s.begin(%s_tb, %s, (uint32_t)(%s), %s, %s, %s, %s);" % (
                o, o, a["sample_rate"].expr, self.enum_tag(a["function"], callee.params[1].param_type.enum), a["curve"].expr,
                a["curve"].count, mc.length, mc.nic))
            cv, cp = k.fresh("cv"), k.fresh("cp")
            frame += ["float %s = 0.0f;" % cv, "const bool %s = %s.frame(%s_tb, %s, %s);" % (cp, o, o, mc.rel, cv)]
            painted, value = cp, cv
        else:
            raise HipBackendError("builtin module %s is not supported by the HIP backend" % name)

        # zang.zero(dest) for a temp, then the module's `+=` (codegen_zig.zig:284-291)
        d = ins.out
        if d.kind == "temp":
            mc.heavy[d.index] = True                                 # a module's output
            mc.srcs[d.index] = out_sines
            mc.cobsrc.pop(d.index, None)
            t = mc.tname(d.index)
            frame.insert(0, "%s = 0.0f;" % t)
            target = t
        else:
            mc.outsines |= out_sines
            target = mc.outvar
        add = "%s = %s + (%s);" % (target, target, value)
        frame.append("if (%s) %s" % (painted, add) if painted else add)
        k.frame += ["{"] + ["    " + l for l in frame] + ["}"]

    # ---- instructions
    def instruction(self, mc, mr, ins):
        k, kind = mc.k, ins.kind
        if kind == "copy_buffer":
            src = self.val(mc, ins.src)
            k.frame += self.put(mc, ins.out, src.expr, False, src.computed, src.sines)
            if ins.out.kind == "temp" and src.cob_b:
                mc.cobsrc[ins.out.index] = (src.cob_b, src.cob_c)
        elif kind == "float_to_buffer":
            k.frame += self.put(mc, ins.out, self.val(mc, ins.src).expr, False)
        elif kind == "cob_to_buffer":
            src = mc.env[ins.in_self_param]
            k.frame += self.put(mc, ins.out, src.expr, False, src.computed, src.sines)
            if ins.out.kind == "temp" and src.cob_b:
                mc.cobsrc[ins.out.index] = (src.cob_b, src.cob_c)
        elif kind in ("arith_float", "arith_float_float"):
            expr = (self.UN[ins.op] % self.val(mc, ins.a).expr if kind == "arith_float"
                    else self.BIN[ins.op] % (self.val(mc, ins.a).expr, self.val(mc, ins.b).expr))
            if mc.begin_sink is k.pro:
                k.pro.append("const float %s = %s;" % (mc.fname(ins.out), expr))
            else:                      # inside a delay / track body: evaluated once per chunk, like the Zig `const` in the loop
                k.pro.append("float %s = 0.0f;" % mc.fname(ins.out))
                mc.begin_sink.append("%s = %s;" % (mc.fname(ins.out), expr))
        elif kind == "arith_buffer":
            va = self.val(mc, ins.a)
            expr, sines = self.UN[ins.op] % va.expr, va.sines
            if ins.op in ("sin", "cos", "sqrt"):
                k.sink(va)
                sines = 0
                if ins.op == "sin":
                    sid, sines = k.sine_source()
                    if sid is not None:
                        expr = "\x01F%d|%s\x02" % (sid, va.expr)          # zsinf(...) / its tolerant form: resolved by generate()
            k.frame += self.put(mc, ins.out, expr, False, va.computed or ins.op in ("sin", "cos"), sines)
        elif kind in ("arith_float_buffer", "arith_buffer_float", "arith_buffer_buffer"):
            va, vb = self.val(mc, ins.a), self.val(mc, ins.b)
            a, b = va.expr, vb.expr
            heavy = va.computed or vb.computed or ins.op == "pow"
            sines = (va.sines if va.kind == "buf" else 0) | (vb.sines if vb.kind == "buf" else 0)
            if ins.op == "pow":
                k.sink(va)
                k.sink(vb)
                sines = 0
            elif ins.op == "div":
                k.sink(vb)
                sines = va.sines if va.kind == "buf" else 0
            if ins.op in ("add", "mul"):
                if kind == "arith_float_buffer":
                    a, b = b, a                                  # addScalar / multiplyScalar(dest, buffer, float)
                k.frame += self.put(mc, ins.out, self.BIN[ins.op] % (a, b), True, heavy, sines)
            else:
                k.frame += self.put(mc, ins.out, self.BIN[ins.op] % (a, b), False, heavy, sines)
        elif kind == "call":
            callee_index = mr.fields[ins.field_index]
            callee = self.s.modules[callee_index]
            if callee.scope is None:
                self.call_builtin(mc, ins, callee, ins.args)
            else:
                env = [self.val(mc, r) for r in ins.args]
                d = ins.out
                if d.kind == "temp":
                    mc.heavy[d.index] = True
                    mc.cobsrc.pop(d.index, None)
                    outvar = mc.tname(d.index)
                    k.frame.append("%s = 0.0f;" % outvar)
                else:
                    outvar = mc.outvar
                sub = _ModuleCtx(k, callee_index, env, outvar, mc.nic, k.fresh(mc.prefix + "c") + "_", parent=mc)
                self.module_body(sub)
                if d.kind == "temp":
                    mc.srcs[d.index] = sub.outsines
                else:
                    mc.outsines |= sub.outsines
        elif kind == "track_call":
            self.track_call(mc, mr, ins)
        elif kind == "delay":
            self.delay(mc, mr, ins)
        else:
            raise AssertionError(kind)

    def delay(self, mc, mr, ins):
        """`delay N begin ... end` (codegen_zig.zig:391-455).  The reference walks the span in chunks of
        samples_read = min(N, remaining) frames (delay.zig:28-57): read the ring into the feedback temp,
        paint the body over the chunk, write the body's `feedback` value back.  A slot is read before it
        is rewritten, so the per-frame form (read slot, body, write slot, advance) gives the same values;
        what must follow the chunks is the per-paint prologue / epilogue of the modules called in the
        body, which the reference runs once per chunk: they are emitted under `first / last frame of the
        chunk` conditions.  The ring is N state words per voice, [word][voice] like all state."""
        k = mc.k
        n = mr.delays[ins.delay_index]
        if n < 1:
            raise HipBackendError("delay of 0 samples")
        w_idx = k.alloc(1)
        w_ring = k.alloc(n)
        d = k.fresh("d")
        k.pro.append("uint32_t %s_idx = zs_ld_u(L.state, %d, V, v);" % (d, w_idx))
        k.epi_stores.append("zs_st_u(L.state, %d, V, v, %s_idx);" % (w_idx, d))
        if ins.out.kind == "temp":
            k.frame.append("%s = 0.0f;" % mc.tname(ins.out.index))   # zang.zero(span, dest) (:396-399)
        fb, fbout = mc.tname(ins.feedback_temp), mc.tname(ins.feedback_out_temp)
        mc.srcs[ins.feedback_temp] = ALL_SINES                       # what a ring holds is of unknown origin
        begins, ends, body = [], [], []
        saved = (mc.begin_sink, mc.end_sink, mc.rel, mc.length, k.frame)
        rel, length = "%s_rel" % d, "%s_len" % d
        head = ["const uint32_t %s = %s %% %du;" % (rel, saved[2], n),
                "const uint32_t %s = min(%du, %s - (%s - %s));" % (length, n, saved[3], saved[2], rel)]
        mc.begin_sink, mc.end_sink, mc.rel, mc.length, k.frame = begins, ends, rel, length, body
        try:
            for sub in ins.instructions:
                self.instruction(mc, mr, sub)
        finally:
            mc.begin_sink, mc.end_sink, mc.rel, mc.length, k.frame = saved
        k.rings = True
        slot = "L.state[(size_t)(%du + %s_idx) * V + v]" % (w_ring, d)
        k.frame += head
        if begins:
            k.frame += ["if (%s == 0u) {" % rel] + ["    " + l for l in begins] + ["}"]
        k.frame += ["%s = 0.0f;" % fbout, "%s = 0.0f;" % fb]
        if n >= 2:
            # the slot of the NEXT frame is a different slot, last written n-1 frames ago: it is loaded while this
            # frame computes, so that no frame waits out its own ring load (read slot / body / write slot per frame
            # exposes the full load latency: the write is an opaque store the next read cannot be hoisted above)
            k.pro.append("float %s_pre = zu2f(L.state[(size_t)(%du + %s_idx) * V + v]);" % (d, w_ring, d))
            k.frame += ["const float %s_cur = %s_pre;" % (d, d),
                        "%s_pre = zu2f(L.state[(size_t)(%du + (%s_idx + 1u == %du ? 0u : %s_idx + 1u)) * V + v]);" % (d, w_ring, d, n, d),
                        "%s = %s + %s_cur;" % (fb, fb, d)]           # readDelayBuffer: `+=` (delay.zig:39-42)
        else:
            k.frame += ["%s = %s + zu2f(%s);" % (fb, fb, slot)]
        k.frame += body
        k.frame += ["%s = zf2u(%s);" % (slot, fbout),             # writeDelayBuffer (delay.zig:62-89)
                    "%s_idx = %s_idx + 1u == %du ? 0u : %s_idx + 1u;" % (d, d, n, d)]
        if ends:
            k.frame += ["if (%s + 1u == %s) {" % (rel, length)] + ["    " + l for l in ends] + ["}"]

    def track_call(self, mc, mr, ins):
        """`from <track>, <speed> begin ... end` (codegen_zig.zig:359-389): NoteTracker.consume turns the
        track's notes that fall into this paint call into impulses, Trigger cuts the span into one
        sub-span per note, and the body is painted once per sub-span with that note's params and
        `_new_note`.  TrackLane (voices.hip.h) builds the sub-span list in the prologue; the frame loop
        runs the body's per-paint prologue at a sub-span's first frame, its epilogue at the last, and
        paints nothing outside the sub-spans (before the first note)."""
        k = mc.k
        ti = ins.track_index
        track = self.s.tracks[ti]
        module = self.s.modules[mc.module_index]
        w = k.alloc(TRACK_WORDS)
        t = k.fresh("trk")
        k.tracks.add(ti)
        # the running note's params: plain variables, reloaded at every sub-span start
        env, loads = {}, []
        for pi, p in enumerate(track.params):
            kind = p.param_type.kind
            if kind == "constant":
                k.pro.append("float %s_p%d = 0.0f;" % (t, pi))
                loads.append("%s_p%d = zs_track%d_p%d[%s_note];" % (t, pi, ti, pi, t))
                env[pi] = Val("float", "%s_p%d" % (t, pi))
            elif kind == "boolean":
                k.pro.append("bool %s_p%d = false;" % (t, pi))
                loads.append("%s_p%d = zs_track%d_p%d[%s_note] != 0;" % (t, pi, ti, pi, t))
                env[pi] = Val("bool", "%s_p%d" % (t, pi))
            elif kind == "one_of":
                k.pro.append("uint32_t %s_p%d = 0u; float %s_q%d = 0.0f;" % (t, pi, t, pi))
                loads.append("%s_p%d = zs_track%d_p%d[%s_note]; %s_q%d = zs_track%d_q%d[%s_note];" % (t, pi, ti, pi, t, t, pi, ti, pi, t))
                env[pi] = Val("enum", tag="%s_p%d" % (t, pi), payload=Val("float", "%s_q%d" % (t, pi)), enum=p.param_type.enum)
            elif kind == "curve":
                k.pro.append("const zh_curve_node *%s_p%d = nullptr; uint32_t %s_q%d = 0u;" % (t, pi, t, pi))
                loads.append("%s_p%d = zs_track%d_p%d[%s_note]; %s_q%d = zs_track%d_q%d[%s_note];" % (t, pi, ti, pi, t, t, pi, ti, pi, t))
                env[pi] = Val("curve", "%s_p%d" % (t, pi), count="%s_q%d" % (t, pi))
            else:
                raise HipBackendError("track param `%s`: type %s is not supported by the HIP backend" % (p.name, kind))
        has_note_on = [i for i, p in enumerate(module.params) if p.name == "note_on"]
        reset = "(%s && %s)" % (mc.env[has_note_on[0]].expr, mc.nic) if has_note_on else mc.nic    # codegen_zig.zig:362-371
        k.pro += ["TrackLane %s;" % t,
                  "%s.next = zs_ld_u(L.state, %d, V, v); %s.t = zs_ld_f(L.state, %d, V, v); %s.cur = zs_ld_u(L.state, %d, V, v);" % (t, w, t, w + 1, t, w + 2),
                  "uint32_t %s_k = 0u, %s_note = 0u; bool %s_new = false;" % (t, t, t)]
        k.epi_stores.append("zs_st_u(L.state, %d, V, v, %s.next); zs_st_f(L.state, %d, V, v, %s.t); zs_st_u(L.state, %d, V, v, %s.cur);" % (w, t, w + 1, t, w + 2, t))
        # tracker.consume(params.sample_rate / speed, span); trigger.counter(span, iap)
        mc.begin_sink += ["%s.begin(zs_track%d_t, %du, (%s) / (%s), %s, %s);" % (t, ti, len(track.notes), mc.env[0].expr, self.val(mc, ins.speed).expr, mc.length, reset),
                          "%s_k = 0u;" % t]
        begins, ends, body = [], [], []
        saved = (mc.begin_sink, mc.end_sink, mc.rel, mc.length, mc.nic, mc.track, k.frame)
        outer_rel = saved[2]
        mc.begin_sink, mc.end_sink, k.frame = begins, ends, body
        mc.rel = "(%s_rel - %s.s_start[%s_k])" % (t, t, t)
        mc.length = "(%s.s_end[%s_k] - %s.s_start[%s_k])" % (t, t, t, t)
        mc.nic, mc.track = "%s_new" % t, env
        try:
            for sub in ins.instructions:
                self.instruction(mc, mr, sub)
        finally:
            mc.begin_sink, mc.end_sink, mc.rel, mc.length, mc.nic, mc.track, k.frame = saved
        I = "    "
        k.frame += ["const uint32_t %s_rel = %s;" % (t, outer_rel),
                    "if (%s_k < %s.n && %s_rel == %s.s_start[%s_k]) {" % (t, t, t, t, t),
                    I + "%s_note = %s.s_note[%s_k];" % (t, t, t),
                    I + "%s_new = %s || %s.s_new[%s_k];" % (t, reset, t, t)]      # _new_note (:379-383)
        k.frame += [I + l for l in loads + begins] + ["}"]
        k.frame += ["const bool %s_on = %s_k < %s.n && %s_rel >= %s.s_start[%s_k];" % (t, t, t, t, t, t),
                    "if (%s_on) {" % t] + [I + l for l in body] + ["}"]
        k.frame += ["if (%s_on && %s_rel + 1u == %s.s_end[%s_k]) {" % (t, t, t, t)] + [I + l for l in ends] + [I + "%s_k++;" % t, "}"]

    def track_tables(self, ti):
        """The track's notes as device tables: times, and one table per param (values are literals: the
        notes are evaluated in the global context, codegen.zig:692-706)."""
        track = self.s.tracks[ti]
        notes = self.s.track_results[ti]
        n = max(len(track.notes), 1)
        out = ["__device__ const float zs_track%d_t[%d] = {%s};" % (ti, n, ", ".join(f32_literal(x.t.value) for x in track.notes) or "0.0f")]
        for pi, p in enumerate(track.params):
            kind = p.param_type.kind
            vals = [notes[ni][pi] for ni in range(len(track.notes))]
            if kind == "constant":
                out.append("__device__ const float zs_track%d_p%d[%d] = {%s};" % (ti, pi, n, ", ".join(f32_literal(r.value.value) for r in vals) or "0.0f"))
            elif kind == "boolean":
                out.append("__device__ const unsigned char zs_track%d_p%d[%d] = {%s};" % (ti, pi, n, ", ".join("1" if r.value else "0" for r in vals) or "0"))
            elif kind == "one_of":
                labels = [v.label for v in p.param_type.enum.values]
                out.append("__device__ const unsigned int zs_track%d_p%d[%d] = {%s};" % (ti, pi, n, ", ".join(str(labels.index(r.value)) for r in vals) or "0"))
                out.append("__device__ const float zs_track%d_q%d[%d] = {%s};" % (
                    ti, pi, n, ", ".join(f32_literal(r.payload.value.value) if r.payload is not None else "0.0f" for r in vals) or "0.0f"))
            elif kind == "curve":                                 # a note's curve is a `defcurve` literal (global context)
                out.append("__device__ const zh_curve_node *const zs_track%d_p%d[%d] = {%s};" % (ti, pi, n, ", ".join("zs_curve%d" % r.index for r in vals) or "nullptr"))
                out.append("__device__ const unsigned int zs_track%d_q%d[%d] = {%s};" % (
                    ti, pi, n, ", ".join(str(len(self.s.curves[r.index].points)) for r in vals) or "0"))
        return out

    def module_body(self, mc):
        mr = self.s.module_results[mc.module_index]
        for ins in mr.instructions:
            self.instruction(mc, mr, ins)

    # ---- one exported module
    def kernel(self, name, module_index):
        module = self.s.modules[module_index]
        if len(module.params) > 16:                             # ZH_SCRIPT_MAX_PARAMS (include/zang_hip.h)
            raise HipBackendError("module has %d params; the loader passes at most 16" % len(module.params))
        k = _Kernel(name)
        env = []
        for i, p in enumerate(module.params):
            kind = p.param_type.kind
            if kind == "constant":
                k.pro.append("const float P%d = zs_const(L.p[%d], v);" % (i, i))
                env.append(Val("float", "P%d" % i))
            elif kind == "boolean":
                k.pro.append("const bool P%d = zs_bool(L.p[%d], v);" % (i, i))
                env.append(Val("bool", "P%d" % i))
            elif kind == "buffer":
                j = len(k.rows)
                k.rows.append(i)
                env.append(Val("buf", "x[%d]" % j))
            elif kind == "constant_or_buffer":
                j = len(k.rows)
                k.rows.append(i)
                k.pro.append("const bool P%d_b = L.p[%d].is_buffer != 0; const float P%d_c = zs_const(L.p[%d], v);" % (i, i, i, i))
                env.append(Val("buf", "(P%d_b ? x[%d] : P%d_c)" % (i, j, i), cob_b="P%d_b" % i, cob_c="P%d_c" % i))     # cob_to_buffer's switch (codegen_zig.zig:130-143)
            elif kind == "curve":
                env.append(Val("curve", "reinterpret_cast<const zh_curve_node *>(L.p[%d].pf)" % i, count="L.p[%d].u" % i))
            else:
                k.pro.append("const uint32_t P%d_tag = L.p[%d].u; const float P%d_f = L.p[%d].f;" % (i, i, i, i))
                env.append(Val("enum", tag="P%d_tag" % i, payload=Val("float", "P%d_f" % i), enum=p.param_type.enum))
            k.params.append((p.name, kind, p.param_type.enum.name if p.param_type.enum else None))
        mc = _ModuleCtx(k, module_index, env, "o", "NIC", "")
        self.module_body(mc)
        return k

    def generate(self, only=None):
        out = ["// generated by zang_amd.zangscript (HIP backend) -- compile with zh_script_load / zh_script_compile",
               '#include "script_rt.hip.h"', ""]
        for ci, curve in enumerate(self.s.curves):
            pts = ", ".join("{%s, %s}" % (f32_literal(v.value), f32_literal(t.value)) for t, v in curve.points)   # {value, t}
            out.append("__device__ const zh_curve_node zs_curve%d[] = {%s};" % (ci, pts or "{0.0f, 0.0f}"))
        meta = {}
        used_tracks, table_at = set(), len(out)
        for name, mi in self.s.exported_modules:
            if only is not None and name not in only:
                continue
            try:
                k = self.kernel(name, mi)
            except HipBackendError as e:
                meta[name] = {"error": str(e)}
                out += ["", "// %s: %s" % (name, e)]
                continue
            meta[name] = {"state_words": k.words, "params": k.params, "noise_fields": k.noise_fields,
                          "num_temps": self.s.module_results[mi].num_temps}
            used_tracks |= k.tracks
            nin = len(k.rows)
            ni = max(nin, 1)
            # frames per unrolled chunk of the frame loop: the unroll exists to prefetch input rows; a body
            # that is large (many inlined module calls) is not replicated 8x -- the 64 KiB instruction cache
            # is shared by two CUs
            unroll = int(os.environ.get("ZH_SCRIPT_UNROLL", "0")) or (8 if len(k.frame) <= 40 else 4 if len(k.frame) <= 100 else 2)
            I = "    "
            # state words the epilogue stores (the loader takes the frame-range form only when it is every word: csrc/zscript_emit.hip)
            stored = set()
            for ln in k.epi_ends + k.epi_stores:
                for mt in re.finditer(r"zs_st_(f|u|u64)\(L\.state, ([0-9]+), V, v,", ln):
                    w = int(mt.group(2))
                    stored |= {w, w + 1} if mt.group(1) == "u64" else {w}
            stored = {w for w in stored if w < k.words}
            out += ["", 'extern "C" __device__ const uint32_t zs_ranges_ok_%s = %du;' % (name, 0 if (k.rings or k.walk_reads_computed) else 1),
                    'extern "C" __device__ const uint32_t zs_state_words_stored_%s = %du;' % (name, len(stored)),
                    'extern "C" __global__ void zs_init_%s(uint32_t *__restrict__ st, uint32_t V, uint64_t first_seed) {' % name,
                    I + "const uint32_t v = blockIdx.x * 64 + threadIdx.x;", I + "if (v >= V) return;",
                    I + "for (uint32_t w = 0; w < %du; w++) st[(size_t)w * V + v] = 0u;" % k.words]
            for item in k.init:
                if item[0] == "f32":
                    out.append(I + "zs_st_f(st, %d, V, v, %s);" % (item[1], f32_literal(item[2])))
                else:                                              # Noise.zig:25-32: seed = counter++ at init()
                    out += [I + "{ ZXoshiro r; zxoshiro_seed(r, first_seed + (uint64_t)v * %du + %du);" % (k.noise_fields, item[2]),
                            I + "  zs_st_u64(st, %d, V, v, r.s0); zs_st_u64(st, %d, V, v, r.s1); zs_st_u64(st, %d, V, v, r.s2); zs_st_u64(st, %d, V, v, r.s3); }"
                            % (item[1], item[1] + 2, item[1] + 4, item[1] + 6)]
            out += ["}", "",
                    'extern "C" __global__ void __launch_bounds__(64) zs_paint_%s(const ZsLaunch L) {' % name,
                    I + "const uint32_t v = blockIdx.x * 64 + threadIdx.x;", I + "const uint32_t V = L.V;", I + "if (v >= V) return;",
                    I + "const bool NIC = L.nic.get(v);", I + "const uint32_t SPAN_LEN = L.end - L.start;",
                    I + "(void)NIC; (void)SPAN_LEN;",
                    I + "const float *ins[%d] = {%s};" % (ni, ", ".join(["nullptr"] * ni)), I + "size_t istr[%d] = {%s};" % (ni, ", ".join(["0"] * ni)),
                    I + "uint32_t ivo[%d] = {%s};" % (ni, ", ".join(["0"] * ni))]
            for j, pi in enumerate(k.rows):
                out.append(I + "ins[%d] = zs_row(L.p[%d], v, istr[%d], ivo[%d]);" % (j, pi, j, j))
            out += [I + l for l in k.pro]
            out.append(I + "bool zs_walk = false; (void)zs_walk;")
            loop_call = I + "zs_frame_loop<%d, %d>(L.out, v, L.ostride, ins, istr, ivo, L.start, L.end, (L.flags & ZH_PAINT_ZERO_FIRST) != 0, zs_walk," % (unroll, nin)
            two_bodies = bool(k.quiet_terms)
            # the sine sources that reach no sink: under ZH_PAINT_TOLERANT their f32 form (a second instance of the frame body,
            # chosen once per paint: ZS_T).  A kernel without one reads as before.
            tolerant = set(i for i in range(k.nsines) if not (k.exact_sines >> i) & 1)
            frame = [resolve_sines(l, tolerant) for l in k.frame]
            lam = I + "                     [&](uint32_t i, const float (&x)[%d], float &o) ZH_INLINE_LAMBDA " % ni
            if tolerant:
                if two_bodies:
                    out.append(I + "auto zs_quiet = [&](int zs_n) ZH_INLINE_LAMBDA -> bool { return " + " && ".join(k.quiet_terms) + "; };")
                out.append(I + "auto zs_body = [&](auto zs_q, auto zs_t, uint32_t i, const float (&x)[%d], float &o) ZH_INLINE_LAMBDA {" % ni)
                out.append(I + I + "constexpr bool ZS_Q = decltype(zs_q)::value; (void)ZS_Q;")
                out.append(I + I + "constexpr bool ZS_T = decltype(zs_t)::value; (void)ZS_T;")
            elif two_bodies:
                out.append(I + "auto zs_quiet = [&](int zs_n) ZH_INLINE_LAMBDA -> bool { return " + " && ".join(k.quiet_terms) + "; };")
                out.append(I + "auto zs_body = [&](auto zs_q, uint32_t i, const float (&x)[%d], float &o) ZH_INLINE_LAMBDA {" % ni)
                out.append(I + I + "constexpr bool ZS_Q = decltype(zs_q)::value; (void)ZS_Q;")
            else:
                out.append(loop_call)
                out.append(lam + "{")
            out.append(I + I + "(void)i; (void)x;")
            if k.temps:
                out.append(I + I + "float " + ", ".join("%s = 0.0f" % t for t in k.temps) + ";")
            out += [I + I + l for l in frame]
            if tolerant:
                out.append(I + "};")
                for t in ("true", "false"):
                    out.append(I + ("if (L.flags & ZH_PAINT_TOLERANT) {" if t == "true" else "} else {"))
                    out.append(loop_call)
                    if two_bodies:
                        out.append(lam + "{ zs_body(zs_tag<false>{}, zs_tag<%s>{}, i, x, o); }, zs_quiet," % t)
                        out.append(lam + "{ zs_body(zs_tag<true>{}, zs_tag<%s>{}, i, x, o); });" % t)
                    else:
                        out.append(lam + "{ zs_body(zs_tag<false>{}, zs_tag<%s>{}, i, x, o); });" % t)
                out.append(I + "}")
            elif two_bodies:
                out.append(I + "};")
                out.append(loop_call)
                out.append(lam + "{ zs_body(zs_tag<false>{}, i, x, o); }, zs_quiet,")
                out.append(lam + "{ zs_body(zs_tag<true>{}, i, x, o); });")
            else:
                out.append(I + "});")
            out += [I + l for l in k.epi_ends + k.epi_stores]
            out.append("}")
        tables = []
        for ti in sorted(used_tracks):
            tables += self.track_tables(ti)
        out[table_at:table_at] = tables
        return "\n".join(out) + "\n", meta


_SINE_MARK = re.compile("\x01([OF])(\\d+)\\|([^\x02]*)\x02")


def resolve_sines(line, tolerant):
    """The sine placeholders of a frame-body line (call_builtin's SineOsc, instruction()'s sin) as code, now that the kernel's sinks are
    known: the exact text for a source that reaches a sink, the ZS_T-switched one otherwise."""
    def rep(m):
        kind, tol, payload = m.group(1), int(m.group(2)) in tolerant, m.group(3)
        if kind == "O":                                           # SineOscLane::frame's SINMODE argument (voices.hip.h)
            if payload == "1":
                return ", (ZS_T ? 2 : 1)" if tol else ""
            return ", (ZS_T ? 2 : (int)!ZS_Q)" if tol else ", !ZS_Q"
        return "(ZS_T ? zsinf_tol(%s) : zsinf(%s))" % (payload, payload) if tol else "zsinf(%s)" % payload
    return _SINE_MARK.sub(rep, line)


def generate_hip(script, only=None):
    return HipEmitter(script).generate(only)
