"""zangscript semantic analysis -> instruction list (src/zangscript/codegen.zig).

Every module body becomes a flat list of buffer-level instructions over numbered temp buffers and
temp floats, exactly as the reference allocates them (TempManager: buffers reuse the lowest free
slot, floats never reuse, codegen.zig:175-223, 788-792), so that `num_temps`, the temps handed to
callees and the generated text agree with `generateZig`."""
from dataclasses import dataclass, field
from typing import List, Optional

from .errors import ScriptError
from .parse import NumberLiteral


@dataclass
class Res:
    """ExpressionResult (codegen.zig:48-60)."""
    kind: str              # nothing | temp_buffer | temp_float | literal_boolean | literal_number |
    #                        literal_enum_value | literal_curve | literal_track | literal_module | self_param | track_param
    index: int = 0         # temp index / curve / track / module / param index
    weak: bool = False     # someone else owns the temp: do not release it
    value: object = None   # bool / NumberLiteral / enum label
    payload: Optional["Res"] = None
    track_index: int = 0


@dataclass
class Dest:
    """BufferDest (codegen.zig:66-69): a temp buffer (assigned, `=`) or an output (accumulated, `+=`)."""
    kind: str              # temp | output
    index: int


@dataclass
class Instr:
    """Instruction (codegen.zig:110-123); `kind` is the union tag."""
    kind: str
    out: object = None                 # Dest, or a temp-float index for arith_float / arith_float_float
    op: str = None
    a: Res = None
    b: Res = None
    src: Res = None                    # copy_buffer / float_to_buffer input
    in_self_param: int = -1            # cob_to_buffer
    field_index: int = -1              # call
    temps: list = None
    args: list = None
    track_index: int = -1              # track_call
    speed: Res = None
    trigger_index: int = -1
    note_tracker_index: int = -1
    delay_index: int = -1              # delay
    feedback_out_temp: int = -1
    feedback_temp: int = -1
    instructions: list = None          # nested (track_call / delay)


@dataclass
class ModuleResult:
    num_outputs: int
    num_temps: int
    num_temp_floats: int
    builtin: bool
    fields: list = None                # callee module index per call site, in call order
    delays: list = None                # num_samples
    note_trackers: list = None         # track index
    triggers: list = None              # track index
    instructions: list = None


@dataclass
class CompiledScript:
    source: object
    packages: tuple
    curves: list
    tracks: list
    modules: list
    track_results: list                # per track: per note: [Res per track param]
    module_results: list
    exported_modules: list             # [(name, module_index)]

    def module_index(self, name):
        for n, i in self.exported_modules:
            if n == name:
                return i
        raise KeyError(name)


class _Temps:
    def __init__(self, reuse):
        self.reuse, self.claimed = reuse, []

    def claim(self):
        if self.reuse:
            for i, used in enumerate(self.claimed):
                if not used:
                    self.claimed[i] = True
                    return i
        self.claimed.append(True)
        return len(self.claimed) - 1

    def release(self, i):
        assert self.claimed[i]
        self.claimed[i] = False

    def final_count(self):
        return len(self.claimed)


class _ModuleState:
    def __init__(self, module_index, module):
        self.module_index = module_index
        self.instructions = []
        self.temp_buffers = _Temps(True)
        self.temp_floats = _Temps(False)       # they become `const` in Zig: never reused
        self.local_results = [None] * len(module.locals)
        self.fields, self.delays, self.triggers, self.note_trackers = [], [], [], []
        self.current_delay = None              # (feedback_temp_index, instruction list)
        self.current_track_call = None         # (track_index, instruction list)


class CodeGen:
    def __init__(self, source, packages, pr):
        self.source, self.packages = source, packages
        self.globals, self.curves, self.tracks, self.modules = pr.globals, pr.curves, pr.tracks, pr.modules
        self.global_results = [None] * len(self.globals)
        self.global_visited = [False] * len(self.globals)
        self.track_results = [None] * len(self.tracks)
        self.module_results = [None] * len(self.modules)
        self.module_visiting = set()

    def fail(self, sr, msg):
        return ScriptError(self.source, sr, msg)

    # ---- result classification (codegen.zig:243-377)
    def _param_type(self, cms, r):
        if r.kind == "self_param":
            return self.modules[cms.module_index].params[r.index].param_type
        if r.kind == "track_param":
            return self.tracks[r.track_index].params[r.index].param_type
        return None

    def is_boolean(self, cms, r):
        pt = self._param_type(cms, r)
        return r.kind == "literal_boolean" or (pt is not None and pt.kind == "boolean")

    def is_float(self, cms, r):
        pt = self._param_type(cms, r)
        return r.kind in ("temp_float", "literal_number") or (pt is not None and pt.kind == "constant")

    def is_buffer(self, cms, r):
        pt = self._param_type(cms, r)
        return r.kind == "temp_buffer" or (pt is not None and pt.kind == "buffer")

    def is_curve(self, cms, r):
        pt = self._param_type(cms, r)
        return r.kind == "literal_curve" or (pt is not None and pt.kind == "curve")

    @staticmethod
    def _enum_allows(allowed, label, has_float):
        for v in allowed:
            if v.label == label:
                return (v.payload == "f32") == has_float
        return False

    def is_enum_value(self, cms, r, allowed):
        if r.kind == "literal_enum_value":
            has_float = r.payload is not None and self.is_float(cms, r.payload)
            return self._enum_allows(allowed, r.value, has_float)
        pt = self._param_type(cms, r)
        if pt is not None and pt.kind == "one_of":               # every possible value must be allowed
            return all(self._enum_allows(allowed, v.label, v.payload == "f32") for v in pt.enum.values)
        return False

    # ---- temps / destinations (:379-424)
    def release(self, cms, r):
        if r.kind == "temp_buffer" and not r.weak:
            cms.temp_buffers.release(r.index)
        elif r.kind == "temp_float" and not r.weak:
            cms.temp_floats.release(r.index)
        elif r.kind == "literal_enum_value" and r.payload is not None:
            self.release(cms, r.payload)

    @staticmethod
    def request_buffer_dest(cms, result_loc):
        return result_loc if result_loc is not None else Dest("temp", cms.temp_buffers.claim())

    @staticmethod
    def commit_buffer_dest(result_loc, dest):
        if result_loc is not None:
            return Res("nothing")
        assert dest.kind == "temp"
        return Res("temp_buffer", dest.index)

    @staticmethod
    def add(cms, instr):                                         # :415-423
        if cms.current_track_call is not None:
            cms.current_track_call[1].append(instr)
        elif cms.current_delay is not None:
            cms.current_delay[1].append(instr)
        else:
            cms.instructions.append(instr)

    # ---- arithmetic (:438-500)
    def gen_un_arith(self, cms, sr, result_loc, op, ea):
        ra = self.gen_expression(cms, ea, None)
        try:
            if self.is_float(cms, ra):
                idx = cms.temp_floats.claim()
                self.add(cms, Instr("arith_float", out=idx, op=op, a=ra))
                return Res("temp_float", idx)
            if self.is_buffer(cms, ra):
                dest = self.request_buffer_dest(cms, result_loc)
                self.add(cms, Instr("arith_buffer", out=dest, op=op, a=ra))
                return self.commit_buffer_dest(result_loc, dest)
            raise self.fail(sr, "arithmetic can only be performed on numeric types")
        finally:
            self.release(cms, ra)

    def gen_bin_arith(self, cms, sr, result_loc, op, ea, eb):
        ra = self.gen_expression(cms, ea, None)
        try:
            rb = self.gen_expression(cms, eb, None)
            try:
                fa, fb = self.is_float(cms, ra), self.is_float(cms, rb)
                ba, bb = self.is_buffer(cms, ra), self.is_buffer(cms, rb)
                if fa and fb:
                    idx = cms.temp_floats.claim()
                    self.add(cms, Instr("arith_float_float", out=idx, op=op, a=ra, b=rb))
                    return Res("temp_float", idx)
                kind = ("arith_float_buffer" if fa and bb else "arith_buffer_float" if ba and fb
                        else "arith_buffer_buffer" if ba and bb else None)
                if kind is None:
                    raise self.fail(sr, "arithmetic can only be performed on numeric types")
                dest = self.request_buffer_dest(cms, result_loc)
                self.add(cms, Instr(kind, out=dest, op=op, a=ra, b=rb))
                return self.commit_buffer_dest(result_loc, dest)
            finally:
                self.release(cms, rb)       # Zig defers run in reverse order: rb, then ra
        finally:
            self.release(cms, ra)

    # ---- calls (:502-620)
    def commit_callee_param(self, cms, sr, r, pt):
        k = pt.kind
        if k == "boolean":
            if self.is_boolean(cms, r):
                return r
            raise self.fail(sr, "expected boolean value")
        if k == "buffer":
            if self.is_buffer(cms, r):
                return r
            if self.is_float(cms, r):
                idx = cms.temp_buffers.claim()
                self.add(cms, Instr("float_to_buffer", out=Dest("temp", idx), src=r))
                return Res("temp_buffer", idx)
            raise self.fail(sr, "expected buffer value")
        if k == "constant_or_buffer":
            if self.is_buffer(cms, r) or self.is_float(cms, r):
                return r
            raise self.fail(sr, "expected float or buffer value")
        if k == "constant":
            if self.is_float(cms, r):
                return r
            raise self.fail(sr, "expected float value")
        if k == "curve":
            if self.is_curve(cms, r):
                return r
            raise self.fail(sr, "expected curve value")
        if self.is_enum_value(cms, r, pt.enum.values):
            return r
        names = ", ".join("'%s'%s" % (v.label, "(number)" if v.payload == "f32" else "") for v in pt.enum.values)
        raise self.fail(sr, "expected one of " + names)

    def gen_args(self, cms, sr, params, args):
        for a in args:
            if not any(p.name == a.param_name for p in params):
                raise self.fail(a.param_name_token.sr, "call target has no param called `%s`" % self.source.text(a.param_name_token.sr))
        results = []
        for p in params:
            arg = None
            for a in args:
                if a.param_name != p.name:
                    continue
                if arg is not None:
                    raise self.fail(a.param_name_token.sr, "param `%s` provided more than once" % self.source.text(a.param_name_token.sr))
                arg = a
            if cms is not None and arg is None and p.name == "sample_rate":      # passed implicitly
                self_params = self.modules[cms.module_index].params
                results.append(Res("self_param", next(j for j, sp in enumerate(self_params) if sp.name == "sample_rate")))
                continue
            if arg is None:
                raise self.fail(sr, "argument list is missing param `%s`" % p.name)
            r = self.gen_expression(cms, arg.value, None)
            results.append(self.commit_callee_param(cms, arg.value.sr, r, p.param_type))
        return results

    def gen_call(self, cms, sr, result_loc, call):
        fr = self.gen_expression(cms, call.a, None)
        if fr.kind != "literal_module":
            raise self.fail(call.a.sr, "not a module")
        callee_index = fr.index
        field_index = len(cms.fields)
        cms.fields.append(callee_index)
        callee = self.modules[callee_index]
        arg_results = self.gen_args(cms, sr, callee.params, call.args)
        temps = [cms.temp_buffers.claim() for _ in range(self.module_results[callee_index].num_temps)]
        dest = self.request_buffer_dest(cms, result_loc)
        self.add(cms, Instr("call", out=dest, field_index=field_index, temps=temps, args=arg_results))
        result = self.commit_buffer_dest(result_loc, dest)
        for t in temps:                                          # deferred releases, reverse order of declaration
            cms.temp_buffers.release(t)
        for r in arg_results:
            self.release(cms, r)
        return result

    def _gen_inner_statements(self, cms, scope, dest, feedback_dest):
        for st in scope.statements:
            if st.kind == "let_assignment":
                cms.local_results[st.local_index] = self.gen_expression(cms, st.expr, None)
            elif st.kind == "output":
                r = self.gen_expression(cms, st.expr, dest)
                self.commit_output(cms, st.expr.sr, r, dest)
                self.release(cms, r)
            else:
                if feedback_dest is None:
                    raise self.fail(st.expr.sr, "`feedback` can only be used within a `delay` operation")
                r = self.gen_expression(cms, st.expr, feedback_dest)
                self.commit_output(cms, st.expr.sr, r, feedback_dest)
                self.release(cms, r)

    def gen_track_call(self, cms, sr, result_loc, e):            # :558-626
        if cms.current_track_call is not None:
            raise self.fail(sr, "you cannot nest track calls")
        if cms.current_delay is not None:
            raise self.fail(sr, "you cannot use a track call inside a delay")
        tr = self.gen_expression(cms, e.a, None)
        if tr.kind != "literal_track":
            raise self.fail(e.a.sr, "not a track")
        speed = self.gen_expression(cms, e.b, None)
        if not self.is_float(cms, speed):
            raise self.fail(e.b.sr, "speed must be a constant value")
        trigger_index = len(cms.triggers)
        cms.triggers.append(tr.index)
        note_tracker_index = len(cms.note_trackers)
        cms.note_trackers.append(tr.index)
        dest = self.request_buffer_dest(cms, result_loc)
        cms.current_track_call = (tr.index, [])
        self._gen_inner_statements(cms, e.scope, dest, None)
        inner = cms.current_track_call[1]
        cms.current_track_call = None
        self.add(cms, Instr("track_call", out=dest, track_index=tr.index, speed=speed, trigger_index=trigger_index,
                            note_tracker_index=note_tracker_index, instructions=inner))
        self.release(cms, speed)
        return self.commit_buffer_dest(result_loc, dest)

    def gen_delay(self, cms, sr, result_loc, e):                 # :628-690
        if cms.current_delay is not None:
            raise self.fail(sr, "you cannot nest delay operations")
        if cms.current_track_call is not None:
            raise self.fail(sr, "you cannot use a delay inside a track call")
        delay_index = len(cms.delays)
        cms.delays.append(e.value)
        feedback_temp = cms.temp_buffers.claim()
        dest = self.request_buffer_dest(cms, result_loc)
        feedback_out_temp = cms.temp_buffers.claim()
        cms.current_delay = (feedback_temp, [])
        self._gen_inner_statements(cms, e.scope, dest, Dest("temp", feedback_out_temp))
        inner = cms.current_delay[1]
        cms.current_delay = None
        self.add(cms, Instr("delay", out=dest, delay_index=delay_index, feedback_out_temp=feedback_out_temp,
                            feedback_temp=feedback_temp, instructions=inner))
        result = self.commit_buffer_dest(result_loc, dest)
        cms.temp_buffers.release(feedback_out_temp)
        cms.temp_buffers.release(feedback_temp)
        return result

    def gen_track(self, track_index):                            # :692-706
        if self.track_results[track_index] is not None:
            return
        track = self.tracks[track_index]
        self.track_results[track_index] = [self.gen_args(None, n.args_sr, track.params, n.args) for n in track.notes]

    def gen_module(self, module_index, sr):                      # :708-767
        if self.module_results[module_index] is not None:
            return
        # a module that (directly or not) calls itself: the reference recurses until the stack ends
        if module_index in self.module_visiting:
            raise self.fail(sr, "circular reference in module")
        self.module_visiting.add(module_index)
        module = self.modules[module_index]
        cms = _ModuleState(module_index, module)
        for st in module.scope.statements:
            if st.kind == "let_assignment":
                cms.local_results[st.local_index] = self.gen_expression(cms, st.expr, None)
            elif st.kind == "output":
                loc = Dest("output", 0)
                r = self.gen_expression(cms, st.expr, loc)
                self.commit_output(cms, st.expr.sr, r, loc)
                self.release(cms, r)
            else:
                raise self.fail(st.expr.sr, "`feedback` can only be used within a `delay` operation")
        for r in cms.local_results:
            if r is not None:
                self.release(cms, r)
        self.module_results[module_index] = ModuleResult(1, cms.temp_buffers.final_count(), cms.temp_floats.final_count(), False,
                                                         cms.fields, cms.delays, cms.note_trackers, cms.triggers, cms.instructions)

    # ---- expressions (:775-911); cms is None in the global context
    @staticmethod
    def _weaken(r):
        if r.kind in ("temp_buffer", "temp_float"):
            return Res(r.kind, r.index, weak=True)
        return r

    def gen_expression(self, cms, e, result_loc):
        k = e.kind
        if k == "literal_boolean":
            return Res("literal_boolean", value=e.value)
        if k == "literal_number":
            return Res("literal_number", value=e.value)
        if k == "literal_enum_value":
            payload = self.gen_expression(cms, e.a, None) if e.a is not None else None
            return Res("literal_enum_value", value=e.value, payload=payload)
        if k == "literal_curve":
            return Res("literal_curve", e.value)
        if k == "literal_track":
            self.gen_track(e.value)
            return Res("literal_track", e.value)
        if k == "literal_module":
            if self.modules[e.value].scope is not None:
                self.gen_module(e.value, e.sr)
            return Res("literal_module", e.value)
        if k == "name":
            name = self.source.text(e.token.sr)
            if cms is not None:
                if cms.current_track_call is not None:
                    ti = cms.current_track_call[0]
                    for pi, p in enumerate(self.tracks[ti].params):
                        if p.name == name:
                            return Res("track_param", pi, track_index=ti)
                for pi, p in enumerate(self.modules[cms.module_index].params):
                    if p.name != name:
                        continue
                    if p.param_type.kind == "constant_or_buffer":      # unwrapped into a buffer at once (:843-848)
                        dest = self.request_buffer_dest(cms, result_loc)
                        self.add(cms, Instr("cob_to_buffer", out=dest, in_self_param=pi))
                        return self.commit_buffer_dest(result_loc, dest)
                    return Res("self_param", pi)
            for gi, g in enumerate(self.globals):
                if g.name == name:
                    break
            else:
                raise self.fail(e.token.sr, "use of undeclared identifier `%s`" % name)
            if self.global_results[gi] is None:
                if self.global_visited[gi]:
                    raise self.fail(e.token.sr, "circular reference in global")
                self.global_visited[gi] = True
                self.global_results[gi] = self.gen_expression(None, self.globals[gi].value, None)
            return self._weaken(self.global_results[gi])
        if k == "local":
            return self._weaken(cms.local_results[e.value])
        if k in ("un_arith", "bin_arith") and cms is None:
            raise self.fail(e.sr, "constant arithmetic is not supported")
        if k == "un_arith":
            return self.gen_un_arith(cms, e.sr, result_loc, e.op, e.a)
        if k == "bin_arith":
            return self.gen_bin_arith(cms, e.sr, result_loc, e.op, e.a, e.b)
        if k == "call":
            return self.gen_call(cms, e.sr, result_loc, e)
        if k == "track_call":
            return self.gen_track_call(cms, e.sr, result_loc, e)
        if k == "delay":
            return self.gen_delay(cms, e.sr, result_loc, e)
        if k == "feedback":
            if cms.current_delay is None:
                raise self.fail(e.sr, "`feedback` can only be used within a `delay` operation")
            return Res("temp_buffer", cms.current_delay[0], weak=True)
        raise AssertionError(k)

    def commit_output(self, cms, sr, r, dest):                   # :913-960
        k = r.kind
        if k == "nothing":
            return
        if k == "temp_buffer":
            self.add(cms, Instr("copy_buffer", out=dest, src=r))
        elif k in ("temp_float", "literal_number"):
            self.add(cms, Instr("float_to_buffer", out=dest, src=r))
        elif k in ("self_param", "track_param"):
            pt = self._param_type(cms, r).kind
            if pt in ("buffer", "constant_or_buffer"):
                self.add(cms, Instr("copy_buffer", out=dest, src=r))
            elif pt == "constant":
                self.add(cms, Instr("float_to_buffer", out=dest, src=r))
            else:
                what = {"boolean": "boolean", "curve": "curve", "one_of": "enum value"}[pt]
                raise self.fail(sr, "expected buffer value, found " + what)
        else:
            what = {"literal_boolean": "boolean", "literal_enum_value": "enum value", "literal_curve": "curve",
                    "literal_track": "track", "literal_module": "module"}[k]
            raise self.fail(sr, "expected buffer value, found " + what)

    def run(self):                                               # :1058-1161
        idx = 0
        for pkg in self.packages:
            for b in pkg.builtins:
                self.module_results[idx] = ModuleResult(b.num_outputs, b.num_temps, 0, True)
                idx += 1
        for gi, g in enumerate(self.globals):
            if self.global_visited[gi]:
                continue
            self.global_visited[gi] = True
            self.global_results[gi] = self.gen_expression(None, g.value, None)
        exported = []
        for gi, g in enumerate(self.globals):
            r = self.global_results[gi]
            if r.kind == "literal_module" and self.modules[r.index].scope is not None:
                exported.append((g.name, r.index))
        return CompiledScript(self.source, self.packages, self.curves, self.tracks, self.modules,
                              self.track_results, self.module_results, exported)
