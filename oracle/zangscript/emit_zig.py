"""The reference's own backend, `generateZig` (src/zangscript/codegen_zig.zig), restated so that the
front-end can be pinned against the reference's golden text (src/zangscript/tests.zig:44-92) and so
that the op sequence the HIP backend must reproduce is on record next to it."""
from .builtins import ModuleParam

ZIG_KEYWORDS = frozenset("""addrspace align allowzero and anyframe anytype asm async await break callconv catch comptime
const continue defer else enum errdefer error export extern fn for if inline linksection noalias noinline nosuspend opaque
or orelse packed pub resume return struct suspend switch test threadlocal try union unreachable usingnamespace var volatile
while""".split())


class _Out:
    """print_helper.zig:19-100: a chunk ending in "{\\n" indents what follows, a chunk starting with
    "}" dedents itself, blank lines are not indented."""

    def __init__(self):
        self.parts, self.indentation, self.indent_next = [], 0, False

    def p(self, text):
        if self.indent_next:
            self.indent_next = False
            if text.startswith("}"):
                self.indentation -= 1
            if not text.startswith("\n"):
                self.parts.append("    " * self.indentation)
        self.parts.append(text)
        if text.endswith("\n"):
            self.indent_next = True
            if text.endswith("{\n"):
                self.indentation += 1


def ident(s):
    return '@"%s"' % s if s in ZIG_KEYWORDS else s


def number(n):                                                   # print_helper.zig:71-81
    v = n.verbatim
    if "0" <= v[0] <= "9" and "." not in v:
        v += ".0"
    return v


class ZigEmitter:
    def __init__(self, script):
        self.s = script
        self.o = _Out()
        self.module = None

    def module_name(self, i):
        m = self.s.modules[i]
        return "%s.%s" % (ident(m.zig_package_name), ident(m.builtin_name)) if m.zig_package_name else "_module%d" % i

    def res(self, r):
        k = r.kind
        if k == "temp_buffer":
            return "temps[%d]" % r.index
        if k == "temp_float":
            return "temp_float%d" % r.index
        if k == "literal_boolean":
            return "true" if r.value else "false"
        if k == "literal_number":
            return number(r.value)
        if k == "literal_enum_value":
            return ".{ .%s = %s }" % (ident(r.value), self.res(r.payload)) if r.payload is not None else "." + ident(r.value)
        if k == "literal_curve":
            return "&_curve%d" % r.index
        if k == "literal_track":
            return "_track%d" % r.index
        if k == "literal_module":
            return self.module_name(r.index)
        if k == "self_param":
            return "params." + ident(self.module.params[r.index].name)
        if k == "track_param":
            return "_result.params." + ident(self.s.tracks[r.track_index].params[r.index].name)
        raise AssertionError(k)

    @staticmethod
    def dest(d):
        return ("temps[%d]" if d.kind == "temp" else "outputs[%d]") % d.index

    def param_decls(self, params, skip_sample_rate):
        for p in params:
            if skip_sample_rate and p.name == "sample_rate":
                continue
            t = p.param_type
            name = {"boolean": "bool", "buffer": "[]const f32", "constant": "f32", "constant_or_buffer": "zang.ConstantOrBuffer",
                    "curve": "[]const zang.CurveNode"}.get(t.kind) or t.enum.zig_name
            self.o.p("%s: %s,\n" % (ident(p.name), name))

    def _loop(self, span, out, rhs):
        p = self.o.p
        p("{\n")
        p("var i = %s.start;\n" % span)
        p("while (i < %s.end) : (i += 1) {\n" % span)
        p("%s[i] %s %s;\n" % (self.dest(out), "+=" if out.kind == "output" else "=", rhs))
        p("}\n")
        p("}\n")

    def _cob_arg(self, arg):
        k = arg.kind
        if k == "temp_buffer":
            return "zang.buffer(temps[%d])" % arg.index
        if k == "temp_float":
            return "zang.constant(temp_float%d)" % arg.index
        if k == "literal_number":
            return "zang.constant(%s)" % number(arg.value)
        if k in ("self_param", "track_param"):
            param = (self.module.params[arg.index] if k == "self_param" else self.s.tracks[arg.track_index].params[arg.index])
            prefix = "params." if k == "self_param" else "_result.params."
            pk = param.param_type.kind
            if pk == "buffer":
                return "zang.buffer(%s%s)" % (prefix, ident(param.name))
            if pk == "constant":
                return "zang.constant(%s%s)" % (prefix, ident(param.name))
            return prefix + ident(param.name)
        return ""

    def instruction(self, inner, ins, span, nic):
        p, k = self.o.p, ins.kind
        UN = {"abs": "std.math.fabs(%s)", "cos": "std.math.cos(%s)", "neg": "-%s", "sin": "std.math.sin(%s)", "sqrt": "std.math.sqrt(%s)"}
        BIN = {"add": "%s + %s", "sub": "%s - %s", "mul": "%s * %s", "div": "%s / %s", "pow": "std.math.pow(f32, %s, %s)",
               "max": "std.math.max(%s, %s)", "min": "std.math.min(%s, %s)"}
        if k == "copy_buffer":
            p("zang.%s(%s, %s, %s);\n" % ("addInto" if ins.out.kind == "output" else "copy", span, self.dest(ins.out), self.res(ins.src)))
        elif k == "float_to_buffer":
            p("zang.%s(%s, %s, %s);\n" % ("addScalarInto" if ins.out.kind == "output" else "set", span, self.dest(ins.out), self.res(ins.src)))
        elif k == "cob_to_buffer":
            p("switch (params.%s) {\n" % ident(self.module.params[ins.in_self_param].name))
            fc, fb = ("addScalarInto", "addInto") if ins.out.kind == "output" else ("set", "copy")
            p(".constant => |v| zang.%s(%s, %s, v),\n" % (fc, span, self.dest(ins.out)))
            p(".buffer => |v| zang.%s(%s, %s, v),\n" % (fb, span, self.dest(ins.out)))
            p("}\n")
        elif k == "arith_float":
            p("const temp_float%d = " % ins.out)
            p((UN[ins.op] % self.res(ins.a)) + ";\n")
        elif k == "arith_buffer":
            self._loop(span, ins.out, UN[ins.op] % (self.res(ins.a) + "[i]"))
        elif k == "arith_float_float":
            p("const temp_float%d = " % ins.out)
            p((BIN[ins.op] % (self.res(ins.a), self.res(ins.b))) + ";\n")
        elif k in ("arith_float_buffer", "arith_buffer_float", "arith_buffer_buffer"):
            a = self.res(ins.a) + ("[i]" if k != "arith_float_buffer" else "")
            b = self.res(ins.b) + ("[i]" if k != "arith_buffer_float" else "")
            if ins.op in ("add", "mul"):
                if ins.out.kind != "output":
                    p("zang.zero(%s, %s);\n" % (span, self.dest(ins.out)))
                if k == "arith_buffer_buffer":
                    fn, x, y = ("zang.add" if ins.op == "add" else "zang.multiply"), self.res(ins.a), self.res(ins.b)
                else:
                    fn = "zang.addScalar" if ins.op == "add" else "zang.multiplyScalar"
                    # float (op) buffer: operands swapped, the supported operators being commutative (:205-206)
                    x, y = (self.res(ins.b), self.res(ins.a)) if k == "arith_float_buffer" else (self.res(ins.a), self.res(ins.b))
                p(fn)
                p("(%s, %s, %s, %s);\n" % (span, self.dest(ins.out), x, y))
            else:
                self._loop(span, ins.out, BIN[ins.op] % (a, b))
        elif k == "call":
            callee = self.s.modules[inner.fields[ins.field_index]]
            if ins.out.kind != "output":
                p("zang.zero(%s, %s);\n" % (span, self.dest(ins.out)))
            p("self.field%d.paint(%s, .{" % (ins.field_index, span))
            p("%s}, .{" % self.dest(ins.out))
            p(", ".join("temps[%d]" % t for t in ins.temps))
            p("}, %s, .{\n" % ident(nic))
            for arg, cp in zip(ins.args, callee.params):
                p(".%s = " % ident(cp.name))
                p(self._cob_arg(arg) if cp.param_type.kind == "constant_or_buffer" else self.res(arg))
                p(",\n")
            p("});\n")
        elif k == "track_call":
            has_note_on = any(prm.name == "note_on" for prm in self.module.params)
            p(("if (params.note_on and %s) {\n" if has_note_on else "if (%s) {\n") % ident(nic))
            p("self.tracker%d.reset();\n" % ins.note_tracker_index)
            p("self.trigger%d.reset();\n" % ins.trigger_index)
            p("}\n")
            p("const _iap%d = self.tracker%d.consume(params.sample_rate / %s, %s);\n" % (ins.note_tracker_index, ins.note_tracker_index, self.res(ins.speed), span))
            p("var _ctr%d = self.trigger%d.counter(%s, _iap%d);\n" % (ins.trigger_index, ins.trigger_index, span, ins.note_tracker_index))
            p("while (self.trigger%d.next(&_ctr%d)) |_result| {\n" % (ins.trigger_index, ins.trigger_index))
            p(("const _new_note = (params.note_on and %s) or _result.note_id_changed;\n" if has_note_on
               else "const _new_note = %s or _result.note_id_changed;\n") % ident(nic))
            for sub in ins.instructions:
                self.instruction(inner, sub, "_result.span", "_new_note")
            p("}\n")
        elif k == "delay":
            if ins.out.kind != "output":
                p("zang.zero(%s, %s);\n" % (span, self.dest(ins.out)))
            p("{\n")
            p("var start = span.start;\n")
            p("const end = span.end;\n")
            p("while (start < end) {\n")
            p("// temps[%d] will be the destination for writing into the feedback buffer\n" % ins.feedback_out_temp)
            p("zang.zero(zang.Span.init(start, end), temps[%d]);\n" % ins.feedback_out_temp)
            p("// temps[%d] will contain the delay buffer's previous contents\n" % ins.feedback_temp)
            p("zang.zero(zang.Span.init(start, end), temps[%d]);\n" % ins.feedback_temp)
            p("const samples_read = self.delay%d.readDelayBuffer(temps[%d][start..end]);\n" % (ins.delay_index, ins.feedback_temp))
            p("const inner_span = zang.Span.init(start, start + samples_read);\n")
            p("\n")
            p("// inner expression\n")
            for sub in ins.instructions:
                self.instruction(inner, sub, "inner_span", nic)
            p("\n")
            p("// write expression result into the delay buffer\n")
            p("self.delay%d.writeDelayBuffer(temps[%d][start..start + samples_read]);\n" % (ins.delay_index, ins.feedback_out_temp))
            p("start += samples_read;\n")
            p("}\n")
            p("}\n")
        else:
            raise AssertionError(k)

    def generate(self):                                          # codegen_zig.zig:459-577
        s, p = self.s, self.o.p
        p("// THIS FILE WAS GENERATED BY THE ZANGC COMPILER\n\n")
        p('const std = @import("std");\n')
        p('const zang = @import("zang");\n')
        for pkg in s.packages:
            if pkg.zig_package_name != "zang":
                p('const %s = @import("%s");\n' % (pkg.zig_package_name, pkg.zig_import_path))
        if s.exported_modules:
            p("\n")
        for name, mi in s.exported_modules:
            p("pub const %s = %s;\n" % (ident(name), self.module_name(mi)))
        for ci, curve in enumerate(s.curves):
            p("\n")
            p("const _curve%d = [_]zang.CurveNode{\n" % ci)
            for t, v in curve.points:
                p(".{ .t = %s, .value = %s },\n" % (number(t), number(v)))
            p("};\n")
        for ti, track in enumerate(s.tracks):
            p("\n")
            p("const _track%d = struct {\n" % ti)
            p("const Params = struct {\n")
            self.param_decls(track.params, False)
            p("};\n")
            p("const notes = [_]zang.Notes(Params).SongEvent{\n")
            for ni, note in enumerate(track.notes):
                p(".{ .t = %s, .note_id = %d, .params = .{" % (number(note.t), ni + 1))
                for pi, prm in enumerate(track.params):
                    if pi > 0:
                        p(",")
                    p(" .%s = %s" % (prm.name, self.res(s.track_results[ti][ni][pi])))
                p(" } },\n")
            p("};\n")
            p("};\n")
        for mi, module in enumerate(s.modules):
            mr = s.module_results[mi]
            if mr.builtin:
                continue
            self.module = module
            p("\n")
            p("const _module%d = struct {\n" % mi)
            p("pub const num_outputs = %d;\n" % mr.num_outputs)
            p("pub const num_temps = %d;\n" % mr.num_temps)
            p("pub const Params = struct {\n")
            self.param_decls(module.params, False)
            p("};\n")
            p("pub const NoteParams = struct {\n")
            self.param_decls(module.params, True)
            p("};\n")
            p("\n")
            for j, f in enumerate(mr.fields):
                p("field%d: %s,\n" % (j, self.module_name(f)))
            for j, n in enumerate(mr.delays):
                p("delay%d: zang.Delay(%d),\n" % (j, n))
            for j, t in enumerate(mr.note_trackers):
                p("tracker%d: zang.Notes(_track%d.Params).NoteTracker,\n" % (j, t))
            for j, t in enumerate(mr.triggers):
                p("trigger%d: zang.Trigger(_track%d.Params),\n" % (j, t))
            p("\n")
            p("pub fn init() _module%d {\n" % mi)
            p("return .{\n")
            for j, f in enumerate(mr.fields):
                p(".field%d = %s.init(),\n" % (j, self.module_name(f)))
            for j, n in enumerate(mr.delays):
                p(".delay%d = zang.Delay(%d).init(),\n" % (j, n))
            for j, t in enumerate(mr.note_trackers):
                p(".tracker%d = zang.Notes(_track%d.Params).NoteTracker.init(&_track%d.notes),\n" % (j, t, t))
            for j, t in enumerate(mr.triggers):
                p(".trigger%d = zang.Trigger(_track%d.Params).init(),\n" % (j, t))
            p("};\n")
            p("}\n")
            p("\n")
            p("pub fn paint(self: *_module%d, span: zang.Span, outputs: [num_outputs][]f32, temps: [num_temps][]f32, note_id_changed: bool, params: Params) void {\n" % mi)
            for ins in mr.instructions:
                self.instruction(mr, ins, "span", "note_id_changed")
            p("}\n")
            p("};\n")
        assert self.o.indentation == 0
        return "".join(self.o.parts)


def generate_zig(script):
    return ZigEmitter(script).generate()
