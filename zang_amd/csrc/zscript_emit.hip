// zscript_emit.hip -- the two zangscript backends (host C++; see zscript.hpp) and the zh_zscript_* C ABI:
//   generate_zig : the reference's own backend (src/zangscript/codegen_zig.zig), whose output pins the
//                  front-end against the reference's golden text (src/zangscript/tests.zig:44-92);
//   generate_hip : one fused lane-per-voice kernel per exported module (see zang_amd/zangscript/emit_hip.py
//                  for the design notes; this file prints the same text).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/zang_hip.h"
#include "zscript.hpp"
#include <algorithm>
#include <regex>

namespace zs {
namespace {

std::string strf(const char *fmt, ...) {
    va_list ap, ap2;
    va_start(ap, fmt);
    va_copy(ap2, ap);
    const int n = vsnprintf(nullptr, 0, fmt, ap);
    va_end(ap);
    std::string s((size_t)n + 1, '\0');
    vsnprintf(&s[0], s.size(), fmt, ap2);
    va_end(ap2);
    s.resize((size_t)n);
    return s;
}
typedef std::vector<std::string> Lines;
void append(Lines &dst, const Lines &src) { dst.insert(dst.end(), src.begin(), src.end()); }
Lines indent(const Lines &src) { Lines o; for (const std::string &l : src) o.push_back("    " + l); return o; }

// ================================================================== Zig backend
const char *kZigKeywords[] = {"addrspace", "align", "allowzero", "and", "anyframe", "anytype", "asm", "async", "await", "break", "callconv", "catch",
                              "comptime", "const", "continue", "defer", "else", "enum", "errdefer", "error", "export", "extern", "fn", "for", "if",
                              "inline", "linksection", "noalias", "noinline", "nosuspend", "opaque", "or", "orelse", "packed", "pub", "resume",
                              "return", "struct", "suspend", "switch", "test", "threadlocal", "try", "union", "unreachable", "usingnamespace",
                              "var", "volatile", "while"};
std::string ident(const std::string &s) {
    for (const char *k : kZigKeywords) if (s == k) return "@\"" + s + "\"";
    return s;
}
std::string number(const NumberLiteral &n) {                       // print_helper.zig:71-81
    std::string v = n.verbatim;
    if (v[0] >= '0' && v[0] <= '9' && v.find('.') == std::string::npos) v += ".0";
    return v;
}

struct ZigOut {                                                    // print_helper.zig:19-100
    std::string text;
    int indentation = 0;
    bool indent_next = false;
    void p(const std::string &t) {
        if (indent_next) {
            indent_next = false;
            if (!t.empty() && t[0] == '}') indentation--;
            if (t.empty() || t[0] != '\n') for (int i = 0; i < indentation; i++) text += "    ";
        }
        text += t;
        if (!t.empty() && t.back() == '\n') {
            indent_next = true;
            if (t.size() >= 2 && t[t.size() - 2] == '{') indentation++;
        }
    }
};

class ZigEmitter {
public:
    const CompiledScript &s;
    ZigOut o;
    const Module *module = nullptr;
    explicit ZigEmitter(const CompiledScript &cs) : s(cs) {}

    std::string module_name(size_t i) const {
        const Module &m = s.pr.modules[i];
        return m.builtin ? ident(m.zig_package_name) + "." + ident(m.builtin_name) : strf("_module%zu", i);
    }
    std::string res(const Res &r) const {
        switch (r.kind) {
        case RK::temp_buffer: return strf("temps[%zu]", r.index);
        case RK::temp_float: return strf("temp_float%zu", r.index);
        case RK::literal_boolean: return r.bval ? "true" : "false";
        case RK::literal_number: return number(r.num);
        case RK::literal_enum_value: return r.payload ? ".{ ." + ident(r.label) + " = " + res(*r.payload) + " }" : "." + ident(r.label);
        case RK::literal_curve: return strf("&_curve%zu", r.index);
        case RK::literal_track: return strf("_track%zu", r.index);
        case RK::literal_module: return module_name(r.index);
        case RK::self_param: return "params." + ident(module->params[r.index].name);
        case RK::track_param: return "_result.params." + ident(s.pr.tracks[r.track_index].params[r.index].name);
        default: return "";
        }
    }
    static std::string dest(const Dest &d) { return strf(d.output ? "outputs[%zu]" : "temps[%zu]", d.index); }
    void param_decls(const std::vector<ModuleParam> &params, bool skip_sample_rate) {
        for (const ModuleParam &p : params) {
            if (skip_sample_rate && p.name == "sample_rate") continue;
            const char *t = "";
            switch (p.type.kind) {
            case PK::boolean: t = "bool"; break;
            case PK::buffer: t = "[]const f32"; break;
            case PK::constant: t = "f32"; break;
            case PK::constant_or_buffer: t = "zang.ConstantOrBuffer"; break;
            case PK::curve: t = "[]const zang.CurveNode"; break;
            case PK::one_of: t = p.type.en->zig_name.c_str(); break;
            }
            o.p(ident(p.name) + ": " + t + ",\n");
        }
    }
    void loop(const std::string &span, const Dest &out, const std::string &rhs) {
        o.p("{\n");
        o.p("var i = " + span + ".start;\n");
        o.p("while (i < " + span + ".end) : (i += 1) {\n");
        o.p(dest(out) + "[i] " + (out.output ? "+=" : "=") + " " + rhs + ";\n");
        o.p("}\n");
        o.p("}\n");
    }
    std::string cob_arg(const Res &arg) const {
        if (arg.kind == RK::temp_buffer) return strf("zang.buffer(temps[%zu])", arg.index);
        if (arg.kind == RK::temp_float) return strf("zang.constant(temp_float%zu)", arg.index);
        if (arg.kind == RK::literal_number) return "zang.constant(" + number(arg.num) + ")";
        if (arg.kind == RK::self_param || arg.kind == RK::track_param) {
            const ModuleParam &param = arg.kind == RK::self_param ? module->params[arg.index] : s.pr.tracks[arg.track_index].params[arg.index];
            const std::string prefix = arg.kind == RK::self_param ? "params." : "_result.params.";
            if (param.type.kind == PK::buffer) return "zang.buffer(" + prefix + ident(param.name) + ")";
            if (param.type.kind == PK::constant) return "zang.constant(" + prefix + ident(param.name) + ")";
            return prefix + ident(param.name);
        }
        return "";
    }
    static std::string un(const std::string &op, const std::string &a) {
        if (op == "abs") return "std.math.fabs(" + a + ")";
        if (op == "cos") return "std.math.cos(" + a + ")";
        if (op == "neg") return "-" + a;
        if (op == "sin") return "std.math.sin(" + a + ")";
        return "std.math.sqrt(" + a + ")";
    }
    static std::string bin(const std::string &op, const std::string &a, const std::string &b) {
        if (op == "add") return a + " + " + b;
        if (op == "sub") return a + " - " + b;
        if (op == "mul") return a + " * " + b;
        if (op == "div") return a + " / " + b;
        if (op == "pow") return "std.math.pow(f32, " + a + ", " + b + ")";
        if (op == "max") return "std.math.max(" + a + ", " + b + ")";
        return "std.math.min(" + a + ", " + b + ")";
    }
    void instruction(const ModuleResult &inner, const Instr &ins, const std::string &span, const std::string &nic) {
        switch (ins.kind) {
        case IK::copy_buffer:
            o.p(std::string("zang.") + (ins.out.output ? "addInto" : "copy") + "(" + span + ", " + dest(ins.out) + ", " + res(ins.src) + ");\n");
            break;
        case IK::float_to_buffer:
            o.p(std::string("zang.") + (ins.out.output ? "addScalarInto" : "set") + "(" + span + ", " + dest(ins.out) + ", " + res(ins.src) + ");\n");
            break;
        case IK::cob_to_buffer:
            o.p("switch (params." + ident(module->params[ins.in_self_param].name) + ") {\n");
            o.p(std::string(".constant => |v| zang.") + (ins.out.output ? "addScalarInto" : "set") + "(" + span + ", " + dest(ins.out) + ", v),\n");
            o.p(std::string(".buffer => |v| zang.") + (ins.out.output ? "addInto" : "copy") + "(" + span + ", " + dest(ins.out) + ", v),\n");
            o.p("}\n");
            break;
        case IK::arith_float:
            o.p(strf("const temp_float%zu = ", ins.out_float));
            o.p(un(ins.op, res(ins.a)) + ";\n");
            break;
        case IK::arith_buffer: loop(span, ins.out, un(ins.op, res(ins.a) + "[i]")); break;
        case IK::arith_float_float:
            o.p(strf("const temp_float%zu = ", ins.out_float));
            o.p(bin(ins.op, res(ins.a), res(ins.b)) + ";\n");
            break;
        case IK::arith_float_buffer: case IK::arith_buffer_float: case IK::arith_buffer_buffer: {
            const std::string a = res(ins.a) + (ins.kind != IK::arith_float_buffer ? "[i]" : "");
            const std::string b = res(ins.b) + (ins.kind != IK::arith_buffer_float ? "[i]" : "");
            if (ins.op == "add" || ins.op == "mul") {
                if (!ins.out.output) o.p("zang.zero(" + span + ", " + dest(ins.out) + ");\n");
                std::string fn, x = res(ins.a), y = res(ins.b);
                if (ins.kind == IK::arith_buffer_buffer) fn = ins.op == "add" ? "zang.add" : "zang.multiply";
                else {
                    fn = ins.op == "add" ? "zang.addScalar" : "zang.multiplyScalar";
                    if (ins.kind == IK::arith_float_buffer) std::swap(x, y);   // operands swapped: the operators are commutative (:205-206)
                }
                o.p(fn);
                o.p("(" + span + ", " + dest(ins.out) + ", " + x + ", " + y + ");\n");
            } else {
                loop(span, ins.out, bin(ins.op, a, b));
            }
            break;
        }
        case IK::call: {
            const Module &callee = s.pr.modules[inner.fields[ins.field_index]];
            if (!ins.out.output) o.p("zang.zero(" + span + ", " + dest(ins.out) + ");\n");
            o.p(strf("self.field%zu.paint(", ins.field_index) + span + ", .{");
            o.p(dest(ins.out) + "}, .{");
            std::string temps;
            for (size_t j = 0; j < ins.temps.size(); j++) temps += (j ? ", " : "") + strf("temps[%zu]", ins.temps[j]);
            o.p(temps);
            o.p("}, " + ident(nic) + ", .{\n");
            for (size_t j = 0; j < ins.args.size(); j++) {
                o.p("." + ident(callee.params[j].name) + " = ");
                o.p(callee.params[j].type.kind == PK::constant_or_buffer ? cob_arg(ins.args[j]) : res(ins.args[j]));
                o.p(",\n");
            }
            o.p("});\n");
            break;
        }
        case IK::track_call: {
            bool has_note_on = false;
            for (const ModuleParam &p : module->params) has_note_on |= p.name == "note_on";
            o.p((has_note_on ? "if (params.note_on and " : "if (") + ident(nic) + ") {\n");
            o.p(strf("self.tracker%zu.reset();\n", ins.note_tracker_index));
            o.p(strf("self.trigger%zu.reset();\n", ins.trigger_index));
            o.p("}\n");
            o.p(strf("const _iap%zu = self.tracker%zu.consume(params.sample_rate / ", ins.note_tracker_index, ins.note_tracker_index) + res(ins.speed) + ", " + span + ");\n");
            o.p(strf("var _ctr%zu = self.trigger%zu.counter(", ins.trigger_index, ins.trigger_index) + span + strf(", _iap%zu);\n", ins.note_tracker_index));
            o.p(strf("while (self.trigger%zu.next(&_ctr%zu)) |_result| {\n", ins.trigger_index, ins.trigger_index));
            o.p((has_note_on ? "const _new_note = (params.note_on and " + ident(nic) + ")" : "const _new_note = " + ident(nic)) + " or _result.note_id_changed;\n");
            for (const Instr &sub : ins.instructions) instruction(inner, sub, "_result.span", "_new_note");
            o.p("}\n");
            break;
        }
        case IK::delay:
            if (!ins.out.output) o.p("zang.zero(" + span + ", " + dest(ins.out) + ");\n");
            o.p("{\n");
            o.p("var start = span.start;\n");
            o.p("const end = span.end;\n");
            o.p("while (start < end) {\n");
            o.p(strf("// temps[%zu] will be the destination for writing into the feedback buffer\n", ins.feedback_out_temp));
            o.p(strf("zang.zero(zang.Span.init(start, end), temps[%zu]);\n", ins.feedback_out_temp));
            o.p(strf("// temps[%zu] will contain the delay buffer's previous contents\n", ins.feedback_temp));
            o.p(strf("zang.zero(zang.Span.init(start, end), temps[%zu]);\n", ins.feedback_temp));
            o.p(strf("const samples_read = self.delay%zu.readDelayBuffer(temps[%zu][start..end]);\n", ins.delay_index, ins.feedback_temp));
            o.p("const inner_span = zang.Span.init(start, start + samples_read);\n");
            o.p("\n");
            o.p("// inner expression\n");
            for (const Instr &sub : ins.instructions) instruction(inner, sub, "inner_span", nic);
            o.p("\n");
            o.p("// write expression result into the delay buffer\n");
            o.p(strf("self.delay%zu.writeDelayBuffer(temps[%zu][start..start + samples_read]);\n", ins.delay_index, ins.feedback_out_temp));
            o.p("start += samples_read;\n");
            o.p("}\n");
            o.p("}\n");
            break;
        }
    }
    std::string generate() {                                         // codegen_zig.zig:459-577
        o.p("// THIS FILE WAS GENERATED BY THE ZANGC COMPILER\n\n");
        o.p("const std = @import(\"std\");\n");
        o.p("const zang = @import(\"zang\");\n");
        for (const Package *pkg : s.packages)
            if (pkg->zig_package_name != "zang") o.p("const " + pkg->zig_package_name + " = @import(\"" + pkg->zig_import_path + "\");\n");
        if (!s.exported_modules.empty()) o.p("\n");
        for (const auto &em : s.exported_modules) o.p("pub const " + ident(em.first) + " = " + module_name(em.second) + ";\n");
        for (size_t ci = 0; ci < s.pr.curves.size(); ci++) {
            o.p("\n");
            o.p(strf("const _curve%zu = [_]zang.CurveNode{\n", ci));
            for (const auto &pt : s.pr.curves[ci].points) o.p(".{ .t = " + number(pt.first) + ", .value = " + number(pt.second) + " },\n");
            o.p("};\n");
        }
        for (size_t ti = 0; ti < s.pr.tracks.size(); ti++) {
            const Track &track = s.pr.tracks[ti];
            o.p("\n");
            o.p(strf("const _track%zu = struct {\n", ti));
            o.p("const Params = struct {\n");
            param_decls(track.params, false);
            o.p("};\n");
            o.p("const notes = [_]zang.Notes(Params).SongEvent{\n");
            for (size_t ni = 0; ni < track.notes.size(); ni++) {
                o.p(".{ .t = " + number(track.notes[ni].t) + strf(", .note_id = %zu, .params = .{", ni + 1));
                for (size_t pi = 0; pi < track.params.size(); pi++) {
                    if (pi > 0) o.p(",");
                    o.p(" ." + track.params[pi].name + " = " + res(s.track_results[ti][ni][pi]));
                }
                o.p(" } },\n");
            }
            o.p("};\n");
            o.p("};\n");
        }
        for (size_t mi = 0; mi < s.pr.modules.size(); mi++) {
            const ModuleResult &mr = s.module_results[mi];
            if (mr.builtin) continue;
            module = &s.pr.modules[mi];
            o.p("\n");
            o.p(strf("const _module%zu = struct {\n", mi));
            o.p(strf("pub const num_outputs = %zu;\n", mr.num_outputs));
            o.p(strf("pub const num_temps = %zu;\n", mr.num_temps));
            o.p("pub const Params = struct {\n");
            param_decls(module->params, false);
            o.p("};\n");
            o.p("pub const NoteParams = struct {\n");
            param_decls(module->params, true);
            o.p("};\n");
            o.p("\n");
            for (size_t j = 0; j < mr.fields.size(); j++) o.p(strf("field%zu: ", j) + module_name(mr.fields[j]) + ",\n");
            for (size_t j = 0; j < mr.delays.size(); j++) o.p(strf("delay%zu: zang.Delay(%zu),\n", j, mr.delays[j]));
            for (size_t j = 0; j < mr.note_trackers.size(); j++) o.p(strf("tracker%zu: zang.Notes(_track%zu.Params).NoteTracker,\n", j, mr.note_trackers[j]));
            for (size_t j = 0; j < mr.triggers.size(); j++) o.p(strf("trigger%zu: zang.Trigger(_track%zu.Params),\n", j, mr.triggers[j]));
            o.p("\n");
            o.p(strf("pub fn init() _module%zu {\n", mi));
            o.p("return .{\n");
            for (size_t j = 0; j < mr.fields.size(); j++) o.p(strf(".field%zu = ", j) + module_name(mr.fields[j]) + ".init(),\n");
            for (size_t j = 0; j < mr.delays.size(); j++) o.p(strf(".delay%zu = zang.Delay(%zu).init(),\n", j, mr.delays[j]));
            for (size_t j = 0; j < mr.note_trackers.size(); j++)
                o.p(strf(".tracker%zu = zang.Notes(_track%zu.Params).NoteTracker.init(&_track%zu.notes),\n", j, mr.note_trackers[j], mr.note_trackers[j]));
            for (size_t j = 0; j < mr.triggers.size(); j++) o.p(strf(".trigger%zu = zang.Trigger(_track%zu.Params).init(),\n", j, mr.triggers[j]));
            o.p("};\n");
            o.p("}\n");
            o.p("\n");
            o.p(strf("pub fn paint(self: *_module%zu, span: zang.Span, outputs: [num_outputs][]f32, temps: [num_temps][]f32, note_id_changed: bool, params: Params) void {\n", mi));
            for (const Instr &ins : mr.instructions) instruction(mr, ins, "span", "note_id_changed");
            o.p("}\n");
            o.p("};\n");
        }
        return o.text;
    }
};

// ================================================================== HIP backend
struct HipBackendError { std::string message; };

std::string f32_literal(float xf) {                                 // Python's float.hex(x) + "f"
    const double x = (double)xf;
    if (x != x) return "__builtin_nanf(\"\")";
    if (isinf(x)) return x < 0 ? "-__builtin_inff()" : "__builtin_inff()";
    const char *sign = signbit(x) ? "-" : "";
    if (x == 0.0) return std::string(sign) + "0x0.0p+0f";
    int e;
    const double m = frexp(fabs(x), &e);                             // m in [0.5, 1)
    const uint64_t frac = (uint64_t)ldexp(m * 2.0 - 1.0, 52);        // 52 fraction bits of the [1, 2) mantissa
    return strf("%s0x1.%013llxp%+df", sign, (unsigned long long)frac, e - 1);
}

struct Val {
    enum Kind { buf, flt, boolean, en, curve } kind = flt;
    std::string expr;              // buf: valid inside the frame body; flt / boolean: a per-paint constant; curve: pointer
    std::string tag;               // enum: label (literal) or a C++ expression (runtime)
    bool tag_literal = false;
    std::shared_ptr<Val> payload;  // enum payload (float) or null
    std::string count;             // curve node count expression
    bool computed = false;         // buf: derives from a module's output or a transcendental function (not just params / constants / + - * /)
    std::string cob_b, cob_c;      // buf that is exactly a constant_or_buffer param's value: its "is a buffer" flag and its constant
    uint64_t sines = 0;            // buf: the sine sources (bit per SineOsc call / sin(), Kernel::nsines) whose results flow into it
};

// ZH_PAINT_TOLERANT (include/zang_hip.h): a sine may be evaluated in f32 (zmath.hip.h zsinf_tol, within 2.4e-7 of musl's) when its
// error can only be scaled and added on its way to the output -- through + - * neg abs min max, copies, a Filter's or a Decimator's
// `input`, a delay ring written (not read: a ring's content is of unknown origin, kAllSines).  Every other place a value can go is a
// SINK that keeps the sines reaching it exact: any other builtin param (an oscillator's freq / phase: the error would be integrated
// or -- PMOscInstrument, profiles/r04/NOTES.md 5a -- multiplied by the carrier's slope at a large argument; a Distortion's input: gain up to
// 64), the argument of sin / cos / sqrt, a divisor, both operands of pow.
const uint64_t kAllSines = ~0ull;
const size_t kMaxSines = 63;          // sources beyond this many in one kernel stay exact
static bool linear_input(const std::string &module, const std::string &param) {
    return (module == "Filter" || module == "Decimator") && param == "input";
}

size_t state_words(const std::string &name) {
    static const std::map<std::string, size_t> w = {{"SineOsc", 1}, {"PulseOsc", 1}, {"TriSawOsc", 2}, {"Noise", 8}, {"Envelope", 4}, {"Gate", 0},
                                                    {"Filter", 2}, {"Decimator", 2}, {"Distortion", 0}, {"Cycle", 1}, {"Portamento", 3}, {"Curve", 4}};
    auto it = w.find(name);
    if (it == w.end()) throw HipBackendError{"builtin module " + name + " is not supported by the HIP backend"};
    return it->second;
}
const size_t kTrackWords = 3;                                       // NoteTracker {next_song_event, t} + Trigger {note}

struct InitItem { bool noise; size_t word; size_t k; float value; };

// One step of the frame body as the role-wave form deals it out (plan_roles below): a builtin's frame, one arithmetic
// instruction, or a whole delay / track construct (opaque).  `text` is what the lane form's body holds for it.
struct Part {                      // a unit's piece for the role-wave form
    Lines lines;
    int cost = 1, walk = 0;
    bool stateful = false;
    bool owner = false;            // the piece that owns the unit's state: its end / store / quiet lines go where it goes
};
struct Unit {
    Lines text;
    // role form: the unit in pieces that may go to different roles -- a plain add into the output as (value [, painted flag]) for
    // the writer role's add; a Filter as (input + offset) / the (l, b) recurrence / the mix.  Empty: `text` is the one piece.
    std::vector<Part> parts;
    std::vector<std::string> ztemps;   // frame-local names the pieces hand values on in
    bool opaque = false;           // a delay / track construct: conditionals around inner units, taken whole
    bool stateful = false;         // owns frame-to-frame state (a lane object, a ring, a track): lives in exactly one role
    int cost = 1;                  // ~VALU instructions per frame
    int walk = 0;                  // ... of which carry state from frame to frame (what a frame whose value nobody wants still costs)
    std::vector<size_t> ends, stores, quiets;   // the lines of Kernel::epi_ends / epi_stores / quiet_terms that belong to it
};
struct Mark { size_t ends, stores, quiets, units, frame; };

struct Kernel {
    std::string name;
    std::vector<HipParam> params;
    Lines pro, frame, epi_ends, epi_stores;
    std::vector<Unit> units;       // the frame body again, unit by unit (k.frame == the units' texts in order)
    Mark mark() const { return Mark{epi_ends.size(), epi_stores.size(), quiet_terms.size(), units.size(), frame.size()}; }
    void unit_done(const Mark &m, Unit u) {
        for (size_t i = m.ends; i < epi_ends.size(); i++) u.ends.push_back(i);
        for (size_t i = m.stores; i < epi_stores.size(); i++) u.stores.push_back(i);
        for (size_t i = m.quiets; i < quiet_terms.size(); i++) u.quiets.push_back(i);
        units.push_back(std::move(u));
    }
    // the units made since `m` (the inside of a delay / track construct) become one opaque unit: everything appended to `frame` since
    void collapse(const Mark &m) {
        Unit u;
        u.opaque = u.stateful = true;
        u.cost = 4;
        for (size_t i = m.units; i < units.size(); i++) u.cost += units[i].cost;
        u.walk = u.cost;
        units.resize(m.units);
        u.text.assign(frame.begin() + (long)m.frame, frame.end());
        unit_done(m, std::move(u));
    }
    std::vector<InitItem> init;
    std::set<size_t> tracks;
    Lines temps;
    std::vector<size_t> rows;
    size_t words = 0, noise_fields = 0, uid = 0;
    bool rings = false;            // a delay ring lives in the state blob and is read and written inside the frame body
    bool walk_reads_computed = false;   // a builtin's frame-to-frame state is fed by a value computed in the frame body
    Lines quiet_terms;             // wave-uniform tests over the next `zs_n` frames (a chunk): no SineOsc of constant freq / phase can
                                   // reach zsinf's rare path, no Envelope ends a stage -- the chunk then runs the body's ZS_Q forms
    size_t nsines = 0;             // sine sources met so far (SineOsc calls and sin() of a buffer)
    uint64_t exact_sines = 0;      // ... and the ones that reach a sink (linear_input above)
    std::string fresh(const std::string &stem) { uid++; return stem + std::to_string(uid); }
    size_t zuid = 0;               // (the role form's own names count apart: the lane form's text does not depend on them)
    std::string zfresh(const std::string &stem) { zuid++; return stem + std::to_string(zuid); }
    size_t alloc(size_t n) { const size_t w = words; words += n; return w; }
    // a new sine source: its id, or -1 = one too many (emitted exact); `bit` = its Val.sines bit
    long sine_source(uint64_t &bit) {
        if (nsines >= kMaxSines) { bit = 0; return -1; }
        bit = 1ull << nsines;
        return (long)nsines++;
    }
    void sink(const Val &v) { if (v.kind == Val::buf) exact_sines |= v.sines; }
};

struct ModuleCtx {
    Kernel &k;
    size_t module_index;
    std::vector<Val> env;
    std::string outvar, nic, prefix;
    std::map<size_t, std::string> tnames;
    std::map<size_t, bool> heavy;  // temp index -> its current value derives from a module output / transcendental (Val.computed)
    std::map<size_t, std::pair<std::string, std::string>> cobsrc;   // temp index -> (Val.cob_b, Val.cob_c) while it holds a cob param's value
    std::map<size_t, uint64_t> srcs;   // temp index -> Val.sines of its current value (unknown temp: kAllSines)
    uint64_t outsines = 0;             // Val.sines of everything added to this module's output so far
    Lines *begin_sink, *end_sink;
    std::string rel = "(i - L.start)", length = "SPAN_LEN";
    const std::map<size_t, Val> *track = nullptr;
    ModuleCtx(Kernel &k_, size_t mi, std::vector<Val> env_, std::string outvar_, std::string nic_, std::string prefix_, const ModuleCtx *parent)
        : k(k_), module_index(mi), env(std::move(env_)), outvar(std::move(outvar_)), nic(std::move(nic_)), prefix(std::move(prefix_)) {
        begin_sink = parent ? parent->begin_sink : &k.pro;
        end_sink = parent ? parent->end_sink : &k.epi_ends;
        if (parent) { rel = parent->rel; length = parent->length; }
    }
    std::string tname(size_t i) {
        auto it = tnames.find(i);
        if (it != tnames.end()) return it->second;
        const std::string n = prefix + "t" + std::to_string(i);
        tnames[i] = n;
        k.temps.push_back(n);
        return n;
    }
    std::string fname(size_t i) const { return prefix + "f" + std::to_string(i); }
};

// The sine placeholders of a frame-body line (call_builtin's SineOsc, instruction()'s sin) as code, now that the kernel's sinks are
// known: the exact text for a source that reaches a sink, the ZS_T-switched one otherwise.
static std::string resolve_sines(const std::string &line, uint64_t tolerant) {
    std::string out;
    size_t at = 0;
    for (;;) {
        const size_t b = line.find('\x01', at);
        if (b == std::string::npos) { out += line.substr(at); return out; }
        const size_t bar = line.find('|', b), e = line.find('\x02', b);
        out += line.substr(at, b - at);
        const char kind = line[b + 1];
        const unsigned long sid = strtoul(line.substr(b + 2, bar - b - 2).c_str(), nullptr, 10);
        const std::string payload = line.substr(bar + 1, e - bar - 1);
        const bool tol = (tolerant >> sid) & 1;
        if (kind == 'O') {                                              // SineOscLane::frame's SINMODE argument (voices.hip.h)
            if (payload == "1") out += tol ? ", (ZS_T ? 2 : 1)" : "";
            else out += tol ? ", (ZS_T ? 2 : (int)!ZS_Q)" : ", !ZS_Q";
        } else {
            out += tol ? "(ZS_T ? zsinf_tol(" + payload + ") : zsinf(" + payload + "))" : "zsinf(" + payload + ")";
        }
        at = e + 1;
    }
}

class HipEmitter {
public:
    const CompiledScript &s;
    explicit HipEmitter(const CompiledScript &cs) : s(cs) {}

    Val val(ModuleCtx &mc, const Res &r) {
        Val v;
        switch (r.kind) {
        case RK::temp_buffer: {
            v.kind = Val::buf; v.expr = mc.tname(r.index);
            auto it = mc.heavy.find(r.index);
            v.computed = it == mc.heavy.end() ? true : it->second;
            auto ss = mc.srcs.find(r.index);
            v.sines = ss == mc.srcs.end() ? kAllSines : ss->second;
            auto cs = mc.cobsrc.find(r.index);
            if (cs != mc.cobsrc.end()) { v.cob_b = cs->second.first; v.cob_c = cs->second.second; }
            return v;
        }
        case RK::temp_float: v.kind = Val::flt; v.expr = mc.fname(r.index); return v;
        case RK::literal_number: v.kind = Val::flt; v.expr = f32_literal(r.num.value); return v;
        case RK::literal_boolean: v.kind = Val::boolean; v.expr = r.bval ? "true" : "false"; return v;
        case RK::literal_enum_value:
            v.kind = Val::en; v.tag = r.label; v.tag_literal = true;
            if (r.payload) v.payload = std::make_shared<Val>(val(mc, *r.payload));
            return v;
        case RK::literal_curve: v.kind = Val::curve; v.expr = strf("zs_curve%zu", r.index); v.count = std::to_string(s.pr.curves[r.index].points.size()); return v;
        case RK::self_param: return mc.env[r.index];
        case RK::track_param: return mc.track->at(r.index);
        default: throw HipBackendError{"internal: value kind"};
        }
    }
    static std::string enum_tag(const Val &v, const BuiltinEnum &en) {
        if (v.tag_literal) {
            for (size_t i = 0; i < en.values.size(); i++) if (en.values[i].label == v.tag) return std::to_string(i);
            throw HipBackendError{"internal: enum label"};
        }
        return v.tag;
    }
    static std::string enum_payload(const Val &v) { return v.payload ? v.payload->expr : "0.0f"; }
    static Lines put_lines(ModuleCtx &mc, const Dest &d, const std::string &expr, bool zero_first, bool heavy, uint64_t sines) {
        if (!d.output) {
            mc.heavy[d.index] = heavy;
            mc.srcs[d.index] = sines;
            mc.cobsrc.erase(d.index);
            const std::string t = mc.tname(d.index);
            if (zero_first) return {t + " = 0.0f;", t + " = " + t + " + (" + expr + ");"};
            return {t + " = " + expr + ";"};
        }
        mc.outsines |= sines;
        return {mc.outvar + " = " + mc.outvar + " + (" + expr + ");"};
    }
    // one arithmetic / copy instruction into the frame body, and its unit (`cost`: ~VALU instructions of `expr`)
    static void put(ModuleCtx &mc, const Dest &d, const std::string &expr, bool zero_first, bool heavy = false, uint64_t sines = 0, int cost = 1) {
        Kernel &k = mc.k;
        const Mark mk = k.mark();
        Unit u;
        u.text = put_lines(mc, d, expr, zero_first, heavy, sines);
        u.cost = cost + (zero_first ? 1 : 0);
        if (d.output && mc.outvar == "o") {                           // `o = o + (expr);`: the value travels, the add is the writer's
            const std::string zo = k.zfresh("zo");
            u.ztemps.push_back(zo);
            Part a, w;
            a.lines = {zo + " = " + expr + ";"};
            a.cost = u.cost; a.owner = true;
            w.lines = {"o = o + (" + zo + ");"};
            u.parts = {a, w};
        }
        append(k.frame, u.text);
        k.unit_done(mk, std::move(u));
    }
    static int op_cost(const std::string &op) {                     // ~VALU instructions (zmath.hip.h)
        if (op == "sin" || op == "cos") return 36;
        if (op == "pow") return 90;
        if (op == "div") return 10;
        if (op == "sqrt") return 8;
        if (op == "max" || op == "min") return 2;
        return 1;
    }
    static std::string un(const std::string &op, const std::string &a) {
        if (op == "abs") return "fabsf(" + a + ")";
        if (op == "cos") return "zcosf(" + a + ")";
        if (op == "neg") return "-(" + a + ")";
        if (op == "sin") return "zsinf(" + a + ")";
        return "sqrtf(" + a + ")";
    }
    static std::string bin(const std::string &op, const std::string &a, const std::string &b) {
        if (op == "add") return "(" + a + ") + (" + b + ")";
        if (op == "sub") return "(" + a + ") - (" + b + ")";
        if (op == "mul") return "(" + a + ") * (" + b + ")";
        if (op == "div") return "(" + a + ") / (" + b + ")";
        if (op == "pow") return "zpowf(" + a + ", " + b + ")";
        if (op == "max") return "zs_max(" + a + ", " + b + ")";
        return "zs_min(" + a + ", " + b + ")";
    }

    void call_builtin(ModuleCtx &mc, const Instr &ins, const Module &callee, const std::vector<Res> &args) {
        Kernel &k = mc.k;
        const Mark mk = k.mark();
        const std::string &name = callee.builtin_name;
        std::map<std::string, Val> a;
        for (size_t i = 0; i < callee.params.size(); i++) a[callee.params[i].name] = val(mc, args[i]);
        // Launching the kernel as frame ranges pays only when replaying the state walk is cheap: not when a computed
        // buffer (an oscillator's output, a filtered signal ...) feeds a builtin's state -- an oscillator's or a cycle's
        // frequency, a filter's or a decimator's input.
        {
            static const std::map<std::string, std::vector<std::string>> state_inputs = {
                {"SineOsc", {"freq"}}, {"PulseOsc", {"freq"}}, {"TriSawOsc", {"freq"}}, {"Cycle", {"speed"}},
                {"Filter", {"input", "cutoff", "res"}}, {"Decimator", {"input"}}};
            auto si = state_inputs.find(name);
            if (si != state_inputs.end())
                for (const std::string &pn : si->second) {
                    auto it = a.find(pn);
                    if (it != a.end() && it->second.kind == Val::buf && it->second.computed) k.walk_reads_computed = true;
                }
        }
        uint64_t out_sines = 0;                                          // Val.sines of the module's output
        for (const auto &pv : a) {
            if (linear_input(name, pv.first)) out_sines |= pv.second.kind == Val::buf ? pv.second.sines : 0;
            else k.sink(pv.second);
        }
        const std::string o = k.fresh("m");
        const size_t w = k.alloc(state_words(name));
        Lines &decl = k.pro, &pro = *mc.begin_sink, &ends = *mc.end_sink, &epi = k.epi_stores;
        Lines frame;
        std::string painted, value;
        const char *oc = o.c_str();
        auto ld_f = [&](const char *field, size_t word) {
            decl.push_back(strf("%s.%s = zs_ld_f(L.state, %zu, V, v);", oc, field, word));
            epi.push_back(strf("zs_st_f(L.state, %zu, V, v, %s.%s);", word, oc, field));
        };
        auto ld_u = [&](const char *field, size_t word, const char *cast) {
            decl.push_back(strf("%s.%s = %szs_ld_u(L.state, %zu, V, v);", oc, field, cast, word));
            epi.push_back(strf("zs_st_u(L.state, %zu, V, v, (uint32_t)%s.%s);", word, oc, field));
        };
        struct Cob { bool is_buf; std::string c, i; };
        auto cob = [](const Val &v) { return v.kind == Val::buf ? Cob{true, "0.0f", v.expr} : Cob{false, v.expr, v.expr}; };
        auto tf = [](bool b) { return b ? "true" : "false"; };
        auto painted_pair = [&]() {
            const std::string cv = k.fresh("cv"), cp = k.fresh("cp");
            return std::make_pair(cv, cp);
        };
        const std::string sr = a.count("sample_rate") ? a["sample_rate"].expr : "";

        if (name == "SineOsc") {
            const Cob f = cob(a["freq"]), ph = cob(a["phase"]);
            decl.push_back("SineOscLane " + o + ";");
            ld_f("t", w);
            pro.push_back(o + ".begin(" + sr + ", " + f.c + ");");
            // frequency and phase constant over the span (the frequency: a constant, or a constant_or_buffer param that is a
            // constant this paint -- it reaches the oscillator as a temp, cob_to_buffer): the kernel gets a second frame body
            // whose sine has no rare-path branch, run for the chunks in which no voice of the wave can reach that path
            // (SineOscLane::small_args)
            const Val &fv = a["freq"];
            const bool quiet = !ph.is_buf && mc.begin_sink == &k.pro && (!f.is_buf || !fv.cob_b.empty());
            if (quiet && !f.is_buf) k.quiet_terms.push_back(o + ".small_args(" + ph.c + ", (float)zs_n)");
            if (quiet && f.is_buf) k.quiet_terms.push_back("(!" + fv.cob_b + " && " + o + ".small_args_step(" + fv.cob_c + " * " + o + ".inv_sr, " + ph.c + ", (float)zs_n))");
            uint64_t bit = 0;
            const long sid = k.sine_source(bit);
            out_sines |= bit;
            const std::string mode = sid < 0 ? std::string(quiet ? ", !ZS_Q" : "") : strf("\x01" "O%ld|%s\x02", sid, quiet ? "!ZS_Q" : "1");   // resolved by generate()
            value = o + ".frame<" + tf(f.is_buf) + mode + ">(" + (f.is_buf ? f.i : "0.0f") + ", " + ph.i + ")";
            ends.push_back(o + ".end();");
        } else if (name == "Cycle") {
            const Cob sp = cob(a["speed"]);
            decl.push_back("CycleLane " + o + ";");
            ld_f("t", w);
            pro.push_back(o + ".begin(" + sr + ", " + sp.c + ");");
            value = o + ".frame<" + tf(sp.is_buf) + ">(" + (sp.is_buf ? sp.i : "0.0f") + ")";
        } else if (name == "PulseOsc" || name == "TriSawOsc") {
            const Cob f = cob(a["freq"]);
            decl.push_back(name + "Lane " + o + ";");
            ld_u("cnt", w, "");
            if (name == "TriSawOsc") ld_f("t", w + 1);
            if (f.is_buf) {
                pro.push_back(o + ".begin_ctrl(" + sr + ", " + a["color"].expr + ");");
                if (name == "PulseOsc") {
                    auto pp = painted_pair();
                    frame.push_back("float " + pp.first + " = 0.0f;");
                    frame.push_back("const bool " + pp.second + " = " + o + ".frame_ctrl(" + f.i + ", " + pp.first + ");");
                    painted = pp.second; value = pp.first;
                } else {
                    value = o + ".frame_ctrl(" + f.i + ")";
                    ends.push_back(o + ".end_ctrl();");
                }
            } else {
                pro.push_back(o + ".begin_const(" + sr + ", " + f.c + ", " + a["color"].expr + ");");
                auto pp = painted_pair();
                frame.push_back("float " + pp.first + " = 0.0f;");
                frame.push_back("const bool " + pp.second + " = " + o + ".frame_const(" + pp.first + ");");
                painted = pp.second; value = pp.first;
            }
        } else if (name == "Noise") {
            decl.push_back("NoiseLane " + o + ";");
            for (size_t j = 0; j < 4; j++) {
                decl.push_back(strf("%s.r.s%zu = zs_ld_u64(L.state, %zu, V, v);", oc, j, w + 2 * j));
                epi.push_back(strf("zs_st_u64(L.state, %zu, V, v, %s.r.s%zu);", w + 2 * j, oc, j));
            }
            pro.push_back(o + ".begin();");
            k.init.push_back(InitItem{true, w, k.noise_fields, 0.0f});
            k.noise_fields++;
            const std::string tag = enum_tag(a["color"], *callee.params[0].type.en);
            if (tag == "0" || tag == "1") value = o + ".frame<" + tf(tag == "1") + ">()";
            else value = "((" + tag + ") == 1u ? " + o + ".frame<true>() : " + o + ".frame<false>())";
        } else if (name == "Envelope") {
            // three literal curves with one tag (not instantaneous): the tag-specialised lane, whose per-frame curve has no selects
            const std::string t0 = enum_tag(a["attack"], *callee.params[1].type.en), t1 = enum_tag(a["decay"], *callee.params[2].type.en),
                              t2 = enum_tag(a["release"], *callee.params[3].type.en);
            const bool one_tag = a["attack"].tag_literal && a["decay"].tag_literal && a["release"].tag_literal && t0 == t1 && t1 == t2 && t0 != "0";
            decl.push_back(one_tag ? "EnvLaneT<1, " + t0 + "> " + o + ";" : "EnvLane " + o + ";");
            ld_u("state", w, "");
            ld_f("t", w + 1);
            ld_f("last_value", w + 2);
            ld_f("start", w + 3);
            pro.push_back(o + ".sample_rate = " + sr + "; " + o + ".sustain_volume = " + a["sustain_volume"].expr + "; " + o + ".note_on = " + a["note_on"].expr + ";");
            const char *stages[3] = {"attack", "decay", "release"};
            for (int i = 0; i < 3; i++)
                pro.push_back(o + "." + stages[i] + " = CurveP{(uint32_t)(" + enum_tag(a[stages[i]], *callee.params[1 + i].type.en) + "), " + enum_payload(a[stages[i]]) + "};");
            pro.push_back(o + ".begin(" + mc.nic + ");");
            auto pp = painted_pair();
            frame.push_back("float " + pp.first + " = 0.0f;");
            // begin() only in the kernel's prologue (not per delay chunk / track sub-span): the frames a frame range
            // replays step the clock and the stage ends only (envelope.hip.h frame_walk)
            // ... and the chunks in which no voice ends a stage run frame_quiet (no clamp, no stage-end test: the kernel's
            // second frame body, see SineOsc above)
            const std::string step = mc.begin_sink == &k.pro ? "frame_sq<ZS_Q>(" + pp.first + ", zs_walk)" : "frame(" + pp.first + ")";
            if (mc.begin_sink == &k.pro) k.quiet_terms.push_back(o + ".quiet(zs_n)");
            frame.push_back("const bool " + pp.second + " = " + o + "." + step + ";");
            painted = pp.second; value = pp.first;
        } else if (name == "Gate") {
            painted = a["note_on"].expr; value = "1.0f";                                          // Gate.zig:28-30
        } else if (name == "Filter") {
            const Cob c = cob(a["cutoff"]), r = cob(a["res"]);
            decl.push_back("FilterLane " + o + ";");
            ld_f("l", w);
            ld_f("b", w + 1);
            pro.push_back(o + ".begin((uint32_t)(" + enum_tag(a["type"], *callee.params[1].type.en) + "), " + c.c + ", " + r.c + ");");
            value = o + ".frame<" + tf(c.is_buf) + ", " + tf(r.is_buf) + ">(" + a["input"].expr + ", " + (c.is_buf ? c.i : "0.0f") + ", " + (r.is_buf ? r.i : "0.0f") + ")";
        } else if (name == "Decimator") {
            decl.push_back("DecimatorLane " + o + ";");
            ld_f("dval", w);
            ld_f("dcount", w + 1);
            k.init.push_back(InitItem{false, w + 1, 0, 1.0f});                                    // Decimator.zig:14-19
            pro.push_back(o + ".begin(" + sr + ", " + a["fake_sample_rate"].expr + ");");
            auto pp = painted_pair();
            frame.push_back("float " + pp.first + " = 0.0f;");
            frame.push_back("const bool " + pp.second + " = " + o + ".frame(" + a["input"].expr + ", " + pp.first + ");");
            painted = pp.second; value = pp.first;
            ends.push_back(o + ".end();");
        } else if (name == "Distortion") {
            decl.push_back("DistortionLane " + o + ";");
            pro.push_back(o + ".begin((uint32_t)(" + enum_tag(a["type"], *callee.params[1].type.en) + "), " + a["ingain"].expr + ", " + a["outgain"].expr + ", " + a["offset"].expr + ");");
            value = o + ".frame(" + a["input"].expr + ")";
        } else if (name == "Portamento") {
            decl.push_back("PortamentoLane " + o + ";");
            ld_f("t", w);
            ld_f("last", w + 1);
            ld_f("st", w + 2);
            pro.push_back(o + ".begin(" + sr + ", (uint32_t)(" + enum_tag(a["curve"], *callee.params[1].type.en) + "), " + enum_payload(a["curve"]) + ", " + a["goal"].expr + ", " +
                          a["note_on"].expr + ", " + a["prev_note_on"].expr + ", " + mc.nic + ");");
            value = o + ".frame()";
        } else if (name == "Curve") {
            decl.push_back("CurveLane " + o + ";");
            decl.push_back("CurveTable " + o + "_tb;");
            ld_f("t", w);
            ld_u("cur", w + 1, "");
            ld_u("off", w + 2, "(int32_t)");
            ld_u("next", w + 3, "");
            pro.push_back(o + ".begin(" + o + "_tb, " + sr + ", (uint32_t)(" + enum_tag(a["function"], *callee.params[1].type.en) + "), " + a["curve"].expr + ", " + a["curve"].count + ", " +
                          mc.length + ", " + mc.nic + ");");
            auto pp = painted_pair();
            frame.push_back("float " + pp.first + " = 0.0f;");
            frame.push_back("const bool " + pp.second + " = " + o + ".frame(" + o + "_tb, " + mc.rel + ", " + pp.first + ");");
            painted = pp.second; value = pp.first;
        } else {
            throw HipBackendError{"builtin module " + name + " is not supported by the HIP backend"};
        }
        // zang.zero(dest) for a temp, then the module's `+=` (codegen_zig.zig:284-291)
        std::string target;
        if (!ins.out.output) {
            mc.heavy[ins.out.index] = true;                          // a module's output
            mc.srcs[ins.out.index] = out_sines;
            mc.cobsrc.erase(ins.out.index);
            target = mc.tname(ins.out.index);
            frame.insert(frame.begin(), target + " = 0.0f;");
        } else {
            mc.outsines |= out_sines;
            target = mc.outvar;
        }
        const std::string addl = target + " = " + target + " + (" + value + ");";
        Unit u;
        u.stateful = state_words(name) > 0;
        u.cost = builtin_cost(name, a, u.walk);
        // the role form's pieces.  The add into `o` is always the writer's: the value (and the painted flag of a builtin that can
        // paint nothing) travels to it.
        auto writer_add = [&](const std::string &val, const std::string &flag) {
            const std::string zo = k.zfresh("zo");
            u.ztemps.push_back(zo);
            Lines give = {zo + " = " + val + ";"};
            Part w;
            w.lines = {"o = o + (" + zo + ");"};
            if (!flag.empty()) {
                const std::string zp = k.zfresh("zp");
                u.ztemps.push_back(zp);
                give.push_back(zp + " = (" + flag + ") ? 1.0f : 0.0f;");
                w.lines = {"if (" + zp + " != 0.0f) o = o + (" + zo + ");"};
            }
            return std::make_pair(give, w);
        };
        if (name == "Filter") {
            // (input + offset), a function of the input sample alone, joins the role that makes the input; the (l, b) recurrence is a
            // role of its own; the mix goes with its consumer (FilterLane::pre / core / post: frame() in three parts)
            const Cob c = cob(a["cutoff"]), r = cob(a["res"]);
            const std::string fi = k.zfresh("zfi"), fl = k.zfresh("zfl"), fb = k.zfresh("zfb"), fh = k.zfresh("zfh");
            for (const std::string &t : {fi, fl, fb, fh}) u.ztemps.push_back(t);
            Part pa, pb, pc;
            pa.lines = {fi + " = " + o + ".pre(" + a["input"].expr + ");"};
            pa.cost = 1;
            pb.lines = {"{", "    const SvfOut zs_s = " + o + ".core<" + tf(c.is_buf) + ", " + tf(r.is_buf) + ">(" + fi + ", " + (c.is_buf ? c.i : "0.0f") + ", " + (r.is_buf ? r.i : "0.0f") + ");",
                        "    " + fl + " = zs_s.l;", "    " + fb + " = zs_s.b;", "    " + fh + " = zs_s.h;", "}"};      // (one statement per line: analyze())
            pb.cost = pb.walk = u.cost - 6; pb.stateful = pb.owner = true;
            // (a literal type other than bypass: the mix does not read the input sample)
            const bool never_bypass = a["type"].tag_literal && a["type"].tag != "bypass";
            const std::string mix = never_bypass ? o + ".mix(" + fl + ", " + fb + ", " + fh + ")" : o + ".post(" + a["input"].expr + ", " + fl + ", " + fb + ", " + fh + ")";
            pc.cost = 5;
            if (target == "o") {
                auto gw = writer_add(mix, "");
                pc.lines = gw.first;
                u.parts = {pa, pb, pc, gw.second};
            } else {
                // (zang.zero(dest) only for a temp of THIS module; an inlined module's output, a temp of its caller, accumulates -- codegen_zig.zig:284-291)
                if (!ins.out.output) pc.lines.push_back(target + " = 0.0f;");
                pc.lines.push_back(target + " = " + target + " + (" + mix + ");");
                u.parts = {pa, pb, pc};
            }
        } else if (target == "o") {
            auto gw = writer_add(value, painted);
            Lines body = frame;
            append(body, gw.first);
            Part pa;
            pa.lines.push_back("{");
            append(pa.lines, indent(body));
            pa.lines.push_back("}");
            pa.cost = u.cost; pa.walk = u.walk; pa.stateful = u.stateful; pa.owner = true;
            u.parts = {pa, gw.second};
        }
        frame.push_back(painted.empty() ? addl : "if (" + painted + ") " + addl);
        u.text.push_back("{");
        append(u.text, indent(frame));
        u.text.push_back("}");
        append(k.frame, u.text);
        k.unit_done(mk, std::move(u));
    }
    // ~VALU instructions of one frame of a builtin as a lone wave's role compiles it (tools/isa_loopstat.py style counts of generated
    // role kernels: TriSawOsc 42, Filter 21, Envelope 13 quiet / 39 around a stage end); `walk` = the part that carries its state
    static int builtin_cost(const std::string &name, std::map<std::string, Val> &a, int &walk) {
        auto is_buf = [&](const char *p) { auto it = a.find(p); return it != a.end() && it->second.kind == Val::buf; };
        auto ret = [&](int cost, int w) { walk = w; return cost; };
        if (name == "SineOsc") return is_buf("freq") ? ret(42, 3) : ret(40, 2);
        if (name == "PulseOsc") return is_buf("freq") ? ret(34, 34) : ret(16, 2);
        if (name == "TriSawOsc") {
            if (is_buf("freq")) return ret(24, 12);
            const std::string &c = a["color"].expr;                  // a literal 0: the sawtooth's two arms alone (voices.hip.h trisaw_sample_saw)
            return c == "0x0.0p+0f" || c == "-0x0.0p+0f" ? ret(17, 2) : ret(42, 2);
        }
        if (name == "Noise") return a["color"].tag_literal && a["color"].tag == "white" ? ret(24, 20) : ret(44, 20);
        if (name == "Envelope") return ret(16, 5);                   // (a walked group of four quiet frames: four clock steps, one curve)
        if (name == "Filter") { const int c = 21 + (is_buf("cutoff") ? 3 : 0) + (is_buf("res") ? 4 : 0); return ret(c, c); }
        if (name == "Decimator") return ret(8, 8);
        if (name == "Distortion") return a["type"].tag_literal && a["type"].tag == "clip" ? ret(6, 0) : ret(40, 0);
        if (name == "Portamento") return ret(14, 9);
        if (name == "Curve") return ret(10, 4);
        if (name == "Cycle") return ret(5, 5);
        return ret(2, 0);                                            // Gate
    }

    void instruction(ModuleCtx &mc, const ModuleResult &mr, const Instr &ins) {
        Kernel &k = mc.k;
        switch (ins.kind) {
        case IK::copy_buffer: {
            const Val src = val(mc, ins.src);
            put(mc, ins.out, src.expr, false, src.computed, src.sines);
            if (!ins.out.output && !src.cob_b.empty()) mc.cobsrc[ins.out.index] = {src.cob_b, src.cob_c};
            break;
        }
        case IK::float_to_buffer: put(mc, ins.out, val(mc, ins.src).expr, false); break;
        case IK::cob_to_buffer: {
            const Val &src = mc.env[ins.in_self_param];
            put(mc, ins.out, src.expr, false, src.computed, src.sines);
            if (!ins.out.output && !src.cob_b.empty()) mc.cobsrc[ins.out.index] = {src.cob_b, src.cob_c};
            break;
        }
        case IK::arith_float: case IK::arith_float_float: {
            const std::string expr = ins.kind == IK::arith_float ? un(ins.op, val(mc, ins.a).expr) : bin(ins.op, val(mc, ins.a).expr, val(mc, ins.b).expr);
            if (mc.begin_sink == &k.pro) {
                k.pro.push_back("const float " + mc.fname(ins.out_float) + " = " + expr + ";");
            } else {                   // inside a delay / track body: evaluated once per chunk, like the Zig `const` in the loop
                k.pro.push_back("float " + mc.fname(ins.out_float) + " = 0.0f;");
                mc.begin_sink->push_back(mc.fname(ins.out_float) + " = " + expr + ";");
            }
            break;
        }
        case IK::arith_buffer: {
            const Val va = val(mc, ins.a);
            std::string expr = un(ins.op, va.expr);
            uint64_t sines = va.sines;
            if (ins.op == "sin" || ins.op == "cos" || ins.op == "sqrt") {
                k.sink(va);
                sines = 0;
                if (ins.op == "sin") {
                    const long sid = k.sine_source(sines);
                    if (sid >= 0) expr = strf("\x01" "F%ld|", sid) + va.expr + "\x02";            // zsinf(...) / its tolerant form: resolved by generate()
                }
            }
            put(mc, ins.out, expr, false, va.computed || ins.op == "sin" || ins.op == "cos", sines, op_cost(ins.op));
            break;
        }
        case IK::arith_float_buffer: case IK::arith_buffer_float: case IK::arith_buffer_buffer: {
            const Val va = val(mc, ins.a), vb = val(mc, ins.b);
            std::string a = va.expr, b = vb.expr;
            const bool heavy = va.computed || vb.computed || ins.op == "pow";
            const uint64_t sa = va.kind == Val::buf ? va.sines : 0, sb = vb.kind == Val::buf ? vb.sines : 0;
            uint64_t sines = sa | sb;
            if (ins.op == "pow") { k.sink(va); k.sink(vb); sines = 0; }
            else if (ins.op == "div") { k.sink(vb); sines = sa; }
            if (ins.op == "add" || ins.op == "mul") {
                if (ins.kind == IK::arith_float_buffer) std::swap(a, b);       // addScalar / multiplyScalar(dest, buffer, float)
                put(mc, ins.out, bin(ins.op, a, b), true, heavy, sines, op_cost(ins.op));
            } else {
                put(mc, ins.out, bin(ins.op, a, b), false, heavy, sines, op_cost(ins.op));
            }
            break;
        }
        case IK::call: {
            const size_t callee_index = mr.fields[ins.field_index];
            const Module &callee = s.pr.modules[callee_index];
            if (callee.builtin) { call_builtin(mc, ins, callee, ins.args); break; }
            std::vector<Val> env;
            for (const Res &r : ins.args) env.push_back(val(mc, r));
            std::string outvar;
            if (!ins.out.output) {
                mc.heavy[ins.out.index] = true; mc.cobsrc.erase(ins.out.index); outvar = mc.tname(ins.out.index);
                const Mark mk = k.mark();
                Unit u;
                u.text = {outvar + " = 0.0f;"};
                append(k.frame, u.text);
                k.unit_done(mk, std::move(u));
            } else outvar = mc.outvar;
            ModuleCtx sub(k, callee_index, env, outvar, mc.nic, k.fresh(mc.prefix + "c") + "_", &mc);
            module_body(sub);
            if (!ins.out.output) mc.srcs[ins.out.index] = sub.outsines;
            else mc.outsines |= sub.outsines;
            break;
        }
        case IK::track_call: track_call(mc, mr, ins); break;
        case IK::delay: delay(mc, mr, ins); break;
        }
    }

    void delay(ModuleCtx &mc, const ModuleResult &mr, const Instr &ins) {
        Kernel &k = mc.k;
        const Mark mk = k.mark();
        const size_t n = mr.delays[ins.delay_index];
        if (n < 1) throw HipBackendError{"delay of 0 samples"};
        const size_t w_idx = k.alloc(1), w_ring = k.alloc(n);
        const std::string d = k.fresh("d");
        k.pro.push_back(strf("uint32_t %s_idx = zs_ld_u(L.state, %zu, V, v);", d.c_str(), w_idx));
        k.epi_stores.push_back(strf("zs_st_u(L.state, %zu, V, v, %s_idx);", w_idx, d.c_str()));
        if (!ins.out.output) k.frame.push_back(mc.tname(ins.out.index) + " = 0.0f;");             // zang.zero(span, dest) (:396-399)
        const std::string fb = mc.tname(ins.feedback_temp), fbout = mc.tname(ins.feedback_out_temp);
        mc.srcs[ins.feedback_temp] = kAllSines;                          // what a ring holds is of unknown origin
        Lines begins, ends, body;
        Lines *sb = mc.begin_sink, *se = mc.end_sink;
        const std::string srel = mc.rel, slen = mc.length;
        const std::string rel = d + "_rel", length = d + "_len";
        const Lines head = {"const uint32_t " + rel + " = " + srel + strf(" %% %zuu;", n),
                            "const uint32_t " + length + strf(" = min(%zuu, ", n) + slen + " - (" + srel + " - " + rel + "));"};
        Lines saved_frame;
        saved_frame.swap(k.frame);
        mc.begin_sink = &begins; mc.end_sink = &ends; mc.rel = rel; mc.length = length;
        try {
            for (const Instr &sub : ins.instructions) instruction(mc, mr, sub);
        } catch (...) {
            mc.begin_sink = sb; mc.end_sink = se; mc.rel = srel; mc.length = slen;
            throw;
        }
        body.swap(k.frame);
        k.frame.swap(saved_frame);
        mc.begin_sink = sb; mc.end_sink = se; mc.rel = srel; mc.length = slen;
        k.rings = true;
        const std::string slot = strf("L.state[(size_t)(%zuu + %s_idx) * V + v]", w_ring, d.c_str());
        append(k.frame, head);
        if (!begins.empty()) { k.frame.push_back("if (" + rel + " == 0u) {"); append(k.frame, indent(begins)); k.frame.push_back("}"); }
        k.frame.push_back(fbout + " = 0.0f;");
        k.frame.push_back(fb + " = 0.0f;");
        if (n >= 2) {
            // the slot of the NEXT frame is a different slot, last written n-1 frames ago: it is loaded while this
            // frame computes, so that no frame waits out its own ring load
            const char *dc = d.c_str();
            k.pro.push_back(strf("float %s_pre = zu2f(L.state[(size_t)(%zuu + %s_idx) * V + v]);", dc, w_ring, dc));
            k.frame.push_back(strf("const float %s_cur = %s_pre;", dc, dc));
            k.frame.push_back(strf("%s_pre = zu2f(L.state[(size_t)(%zuu + (%s_idx + 1u == %zuu ? 0u : %s_idx + 1u)) * V + v]);", dc, w_ring, dc, n, dc));
            k.frame.push_back(fb + " = " + fb + " + " + d + "_cur;");                             // readDelayBuffer: `+=` (delay.zig:39-42)
        } else {
            k.frame.push_back(fb + " = " + fb + " + zu2f(" + slot + ");");
        }
        append(k.frame, body);
        k.frame.push_back(slot + " = zf2u(" + fbout + ");");                                       // writeDelayBuffer (delay.zig:62-89)
        k.frame.push_back(strf("%s_idx = %s_idx + 1u == %zuu ? 0u : %s_idx + 1u;", d.c_str(), d.c_str(), n, d.c_str()));
        if (!ends.empty()) { k.frame.push_back("if (" + rel + " + 1u == " + length + ") {"); append(k.frame, indent(ends)); k.frame.push_back("}"); }
        k.collapse(mk);
    }

    void track_call(ModuleCtx &mc, const ModuleResult &mr, const Instr &ins) {
        Kernel &k = mc.k;
        const Mark mk = k.mark();
        const size_t ti = ins.track_index;
        const Track &track = s.pr.tracks[ti];
        const Module &module = s.pr.modules[mc.module_index];
        const size_t w = k.alloc(kTrackWords);
        const std::string t = k.fresh("trk");
        const char *tc = t.c_str();
        k.tracks.insert(ti);
        std::map<size_t, Val> env;
        Lines loads;
        for (size_t pi = 0; pi < track.params.size(); pi++) {
            const ModuleParam &p = track.params[pi];
            Val v;
            if (p.type.kind == PK::constant) {
                k.pro.push_back(strf("float %s_p%zu = 0.0f;", tc, pi));
                loads.push_back(strf("%s_p%zu = zs_track%zu_p%zu[%s_note];", tc, pi, ti, pi, tc));
                v.kind = Val::flt; v.expr = strf("%s_p%zu", tc, pi);
            } else if (p.type.kind == PK::boolean) {
                k.pro.push_back(strf("bool %s_p%zu = false;", tc, pi));
                loads.push_back(strf("%s_p%zu = zs_track%zu_p%zu[%s_note] != 0;", tc, pi, ti, pi, tc));
                v.kind = Val::boolean; v.expr = strf("%s_p%zu", tc, pi);
            } else if (p.type.kind == PK::one_of) {
                k.pro.push_back(strf("uint32_t %s_p%zu = 0u; float %s_q%zu = 0.0f;", tc, pi, tc, pi));
                loads.push_back(strf("%s_p%zu = zs_track%zu_p%zu[%s_note]; %s_q%zu = zs_track%zu_q%zu[%s_note];", tc, pi, ti, pi, tc, tc, pi, ti, pi, tc));
                v.kind = Val::en; v.tag = strf("%s_p%zu", tc, pi);
                v.payload = std::make_shared<Val>();
                v.payload->kind = Val::flt; v.payload->expr = strf("%s_q%zu", tc, pi);
            } else if (p.type.kind == PK::curve) {
                k.pro.push_back(strf("const zh_curve_node *%s_p%zu = nullptr; uint32_t %s_q%zu = 0u;", tc, pi, tc, pi));
                loads.push_back(strf("%s_p%zu = zs_track%zu_p%zu[%s_note]; %s_q%zu = zs_track%zu_q%zu[%s_note];", tc, pi, ti, pi, tc, tc, pi, ti, pi, tc));
                v.kind = Val::curve; v.expr = strf("%s_p%zu", tc, pi); v.count = strf("%s_q%zu", tc, pi);
            } else {
                const char *kn = p.type.kind == PK::buffer ? "buffer" : "constant_or_buffer";
                throw HipBackendError{"track param `" + p.name + "`: type " + kn + " is not supported by the HIP backend"};
            }
            env[pi] = v;
        }
        std::string reset = mc.nic;                                                               // codegen_zig.zig:362-371
        for (size_t i = 0; i < module.params.size(); i++)
            if (module.params[i].name == "note_on") { reset = "(" + mc.env[i].expr + " && " + mc.nic + ")"; break; }
        k.pro.push_back("TrackLane " + t + ";");
        k.pro.push_back(strf("%s.next = zs_ld_u(L.state, %zu, V, v); %s.t = zs_ld_f(L.state, %zu, V, v); %s.cur = zs_ld_u(L.state, %zu, V, v);", tc, w, tc, w + 1, tc, w + 2));
        k.pro.push_back(strf("uint32_t %s_k = 0u, %s_note = 0u; bool %s_new = false;", tc, tc, tc));
        k.epi_stores.push_back(strf("zs_st_u(L.state, %zu, V, v, %s.next); zs_st_f(L.state, %zu, V, v, %s.t); zs_st_u(L.state, %zu, V, v, %s.cur);", w, tc, w + 1, tc, w + 2, tc));
        mc.begin_sink->push_back(t + strf(".begin(zs_track%zu_t, %zuu, (", ti, track.notes.size()) + mc.env[0].expr + ") / (" + val(mc, ins.speed).expr + "), " + mc.length + ", " + reset + ");");
        mc.begin_sink->push_back(t + "_k = 0u;");
        Lines begins, ends, body;
        Lines *sb = mc.begin_sink, *se = mc.end_sink;
        const std::string srel = mc.rel, slen = mc.length, snic = mc.nic;
        const std::map<size_t, Val> *strack = mc.track;
        Lines saved_frame;
        saved_frame.swap(k.frame);
        mc.begin_sink = &begins; mc.end_sink = &ends;
        mc.rel = strf("(%s_rel - %s.s_start[%s_k])", tc, tc, tc);
        mc.length = strf("(%s.s_end[%s_k] - %s.s_start[%s_k])", tc, tc, tc, tc);
        mc.nic = t + "_new"; mc.track = &env;
        auto restore = [&]() { mc.begin_sink = sb; mc.end_sink = se; mc.rel = srel; mc.length = slen; mc.nic = snic; mc.track = strack; };
        try {
            for (const Instr &sub : ins.instructions) instruction(mc, mr, sub);
        } catch (...) { restore(); throw; }
        body.swap(k.frame);
        k.frame.swap(saved_frame);
        restore();
        const std::string I = "    ";
        k.frame.push_back(strf("const uint32_t %s_rel = ", tc) + srel + ";");
        k.frame.push_back(strf("if (%s_k < %s.n && %s_rel == %s.s_start[%s_k]) {", tc, tc, tc, tc, tc));
        k.frame.push_back(I + strf("%s_note = %s.s_note[%s_k];", tc, tc, tc));
        k.frame.push_back(I + t + "_new = " + reset + strf(" || %s.s_new[%s_k];", tc, tc));      // _new_note (:379-383)
        append(k.frame, indent(loads));
        append(k.frame, indent(begins));
        k.frame.push_back("}");
        k.frame.push_back(strf("const bool %s_on = %s_k < %s.n && %s_rel >= %s.s_start[%s_k];", tc, tc, tc, tc, tc, tc));
        k.frame.push_back(strf("if (%s_on) {", tc));
        append(k.frame, indent(body));
        k.frame.push_back("}");
        k.frame.push_back(strf("if (%s_on && %s_rel + 1u == %s.s_end[%s_k]) {", tc, tc, tc, tc));
        append(k.frame, indent(ends));
        k.frame.push_back(I + t + "_k++;");
        k.frame.push_back("}");
        k.collapse(mk);
    }

    Lines track_tables(size_t ti) {
        const Track &track = s.pr.tracks[ti];
        const auto &notes = s.track_results[ti];
        const size_t n = track.notes.empty() ? 1 : track.notes.size();
        auto join = [](const Lines &parts, const char *empty) {
            std::string o;
            for (size_t i = 0; i < parts.size(); i++) o += (i ? ", " : "") + parts[i];
            return o.empty() ? std::string(empty) : o;
        };
        Lines out, parts;
        for (const TrackNote &x : track.notes) parts.push_back(f32_literal(x.t.value));
        out.push_back(strf("__device__ const float zs_track%zu_t[%zu] = {", ti, n) + join(parts, "0.0f") + "};");
        for (size_t pi = 0; pi < track.params.size(); pi++) {
            const ModuleParam &p = track.params[pi];
            Lines a, b;
            if (p.type.kind == PK::constant) {
                for (size_t ni = 0; ni < track.notes.size(); ni++) a.push_back(f32_literal(notes[ni][pi].num.value));
                out.push_back(strf("__device__ const float zs_track%zu_p%zu[%zu] = {", ti, pi, n) + join(a, "0.0f") + "};");
            } else if (p.type.kind == PK::boolean) {
                for (size_t ni = 0; ni < track.notes.size(); ni++) a.push_back(notes[ni][pi].bval ? "1" : "0");
                out.push_back(strf("__device__ const unsigned char zs_track%zu_p%zu[%zu] = {", ti, pi, n) + join(a, "0") + "};");
            } else if (p.type.kind == PK::one_of) {
                for (size_t ni = 0; ni < track.notes.size(); ni++) {
                    const Res &r = notes[ni][pi];
                    size_t idx = 0;
                    while (p.type.en->values[idx].label != r.label) idx++;
                    a.push_back(std::to_string(idx));
                    b.push_back(r.payload ? f32_literal(r.payload->num.value) : "0.0f");
                }
                out.push_back(strf("__device__ const unsigned int zs_track%zu_p%zu[%zu] = {", ti, pi, n) + join(a, "0") + "};");
                out.push_back(strf("__device__ const float zs_track%zu_q%zu[%zu] = {", ti, pi, n) + join(b, "0.0f") + "};");
            } else if (p.type.kind == PK::curve) {               // a note's curve is a `defcurve` literal (global context)
                for (size_t ni = 0; ni < track.notes.size(); ni++) {
                    a.push_back(strf("zs_curve%zu", notes[ni][pi].index));
                    b.push_back(std::to_string(s.pr.curves[notes[ni][pi].index].points.size()));
                }
                out.push_back(strf("__device__ const zh_curve_node *const zs_track%zu_p%zu[%zu] = {", ti, pi, n) + join(a, "nullptr") + "};");
                out.push_back(strf("__device__ const unsigned int zs_track%zu_q%zu[%zu] = {", ti, pi, n) + join(b, "0") + "};");
            }
        }
        return out;
    }

    void module_body(ModuleCtx &mc) {
        const ModuleResult &mr = s.module_results[mc.module_index];
        for (const Instr &ins : mr.instructions) instruction(mc, mr, ins);
    }

    void kernel(Kernel &k, size_t module_index) {
        const Module &module = s.pr.modules[module_index];
        if (module.params.size() > ZH_SCRIPT_MAX_PARAMS)
            throw HipBackendError{strf("module has %zu params; the loader passes at most 16", module.params.size())};
        std::vector<Val> env;
        for (size_t i = 0; i < module.params.size(); i++) {
            const ModuleParam &p = module.params[i];
            Val v;
            const char *kind = "";
            switch (p.type.kind) {
            case PK::constant:
                kind = "constant";
                k.pro.push_back(strf("const float P%zu = zs_const(L.p[%zu], v);", i, i));
                v.kind = Val::flt; v.expr = strf("P%zu", i);
                break;
            case PK::boolean:
                kind = "boolean";
                k.pro.push_back(strf("const bool P%zu = zs_bool(L.p[%zu], v);", i, i));
                v.kind = Val::boolean; v.expr = strf("P%zu", i);
                break;
            case PK::buffer: {
                kind = "buffer";
                const size_t j = k.rows.size();
                k.rows.push_back(i);
                v.kind = Val::buf; v.expr = strf("x[%zu]", j);
                break;
            }
            case PK::constant_or_buffer: {
                kind = "constant_or_buffer";
                const size_t j = k.rows.size();
                k.rows.push_back(i);
                k.pro.push_back(strf("const bool P%zu_b = L.p[%zu].is_buffer != 0; const float P%zu_c = zs_const(L.p[%zu], v);", i, i, i, i));
                v.kind = Val::buf; v.expr = strf("(P%zu_b ? x[%zu] : P%zu_c)", i, j, i);         // cob_to_buffer's switch (codegen_zig.zig:130-143)
                v.cob_b = strf("P%zu_b", i); v.cob_c = strf("P%zu_c", i);
                break;
            }
            case PK::curve:
                kind = "curve";
                v.kind = Val::curve; v.expr = strf("reinterpret_cast<const zh_curve_node *>(L.p[%zu].pf)", i); v.count = strf("L.p[%zu].u", i);
                break;
            case PK::one_of:
                kind = "one_of";
                k.pro.push_back(strf("const uint32_t P%zu_tag = L.p[%zu].u; const float P%zu_f = L.p[%zu].f;", i, i, i, i));
                v.kind = Val::en; v.tag = strf("P%zu_tag", i);
                v.payload = std::make_shared<Val>();
                v.payload->kind = Val::flt; v.payload->expr = strf("P%zu_f", i);
                break;
            }
            env.push_back(v);
            k.params.push_back(HipParam{p.name, kind, p.type.en ? p.type.en->name : ""});
        }
        ModuleCtx mc(k, module_index, env, "o", "NIC", "", nullptr);
        module_body(mc);
    }

    // ================================================================ the role-wave form (script_rt.hip.h zs_role_run)
    // The frame body's units dealt to waves.  plan: (1) every unit's exposed temp reads / writes, input rows and use of `o`, from its
    // text (the names are unique identifiers); (2) reaching definitions in body order; (3) roles: a unit that touches `o`, or reads
    // what such a unit wrote, is the WRITER's (the last role); pure cheap arithmetic on params / input rows alone FLOATS (copied into
    // every role that reads it); a heavy or stateful unit opens a new role while the budget lasts; everything else joins the
    // role of a producer it reads from -- one that none of its other producers depends on, the cheapest such.  The role graph
    // stays acyclic by construction; a role's lag is its longest path from the sources, then as late as its consumers allow
    // (fewer tile buffers).  A value that crosses roles is stored to its tile right after the defining unit and loaded right
    // before the first unit of the consumer role that reads it.
    struct RUnit {
        Lines lines;
        std::set<std::string> reads, writes;
        std::set<size_t> rows;
        bool touches_o = false, stateful = false, opaque = false, floating = false;
        int cost = 1, walk = 0, role = -1;
        std::vector<size_t> ends, stores, quiets;
    };
    struct Role {
        std::vector<size_t> items;         // unit indices in body order (floating units: copies)
        std::set<int> preds;
        int cost = 0, lag = 0;
        int walk = 0, rep = 1;             // the state-carrying part of `cost`; the waves the role runs in (zs_role_run K)
        int walk_sum = 0;                  // (while roles are being dealt: `walk` of the units so far)
        std::vector<size_t> tin, tout;     // transfer indices, in the order of zin[] / zout[]
    };
    struct Transfer {
        long def;                          // defining unit, or -1 - j: input row j, or kOIn: the live output row
        std::string var;
        int from = 0;                      // producing role (loader sources: -1)
        std::set<int> to;
        int depth = 2, src_lag = 0;
        size_t off = 0;
    };
    static constexpr long kOIn = -1000000;
    static constexpr int kMaxProducers = 5;

    template <class CB> static void scan_idents(const std::string &l, size_t from, size_t to, CB &&cb) {
        size_t i = from;
        while (i < to) {
            const char c = l[i];
            if ((c >= '0' && c <= '9')) {                                // a numeric literal (0x1.8p+0f ...): not identifiers
                size_t j = i + 1;
                while (j < to && (isalnum((unsigned char)l[j]) || l[j] == '.' || ((l[j] == '+' || l[j] == '-') && (l[j - 1] == 'p' || l[j - 1] == 'P' || l[j - 1] == 'e' || l[j - 1] == 'E')))) j++;
                i = j;
            } else if (isalpha((unsigned char)c) || c == '_') {
                size_t j = i + 1;
                while (j < to && (isalnum((unsigned char)l[j]) || l[j] == '_')) j++;
                const bool member = i > 0 && l[i - 1] == '.';
                if (!member) cb(l.substr(i, j - i), i, j);
                i = j;
            } else {
                i++;
            }
        }
    }
    static void analyze(RUnit &u, const std::set<std::string> &temps) {
        std::set<std::string> written;
        auto read = [&](const std::string &id, size_t, size_t e, const std::string &l) {
            if (id == "o") { u.touches_o = true; return; }
            if (id == "x" && e < l.size() && l[e] == '[') { u.rows.insert((size_t)strtoul(l.c_str() + e + 1, nullptr, 10)); return; }
            if (!temps.count(id)) return;
            if (u.opaque) { u.reads.insert(id); u.writes.insert(id); return; }
            if (!written.count(id)) u.reads.insert(id);
        };
        for (const std::string &l : u.lines) {
            size_t p = l.find_first_not_of(' ');
            if (p == std::string::npos) continue;
            size_t cond0 = 0, cond1 = 0;
            bool conditional = false;
            if (!u.opaque && l.compare(p, 4, "if (") == 0) {
                int depth = 0;
                size_t q = p + 3;
                for (; q < l.size(); q++) { if (l[q] == '(') depth++; else if (l[q] == ')' && --depth == 0) break; }
                cond0 = p + 4; cond1 = q; conditional = true;
                p = l.find_first_not_of(' ', q + 1);
                if (p == std::string::npos) p = l.size();
            }
            std::string target;
            size_t rhs = p;
            if (!u.opaque) {
                size_t j = p;
                while (j < l.size() && (isalnum((unsigned char)l[j]) || l[j] == '_')) j++;
                const std::string id = l.substr(p, j - p);
                if (l.compare(j, 3, " = ") == 0 && (temps.count(id) || id == "o")) { target = id; rhs = j + 3; }
            }
            scan_idents(l, cond0, cond1, [&](const std::string &id, size_t b, size_t e) { read(id, b, e, l); });
            scan_idents(l, rhs, l.size(), [&](const std::string &id, size_t b, size_t e) { read(id, b, e, l); });
            if (target == "o") u.touches_o = true;
            else if (!target.empty()) {
                if (conditional && !written.count(target)) u.reads.insert(target);   // keeps its old value where the condition fails
                written.insert(target);
                u.writes.insert(target);
            }
        }
    }

    // the text of zs_paint_pc_<name> and its launch record, or {} when the module has no use for the form
    Lines role_kernel(const Kernel &k, size_t nin, size_t ni, const Lines &preamble, size_t lds_budget, bool &worth) {
        // ---- (1) units
        std::set<std::string> temps(k.temps.begin(), k.temps.end());
        std::vector<RUnit> us;
        for (const Unit &u : k.units) {
            temps.insert(u.ztemps.begin(), u.ztemps.end());
            if (u.parts.empty()) {
                RUnit r;
                r.stateful = u.stateful; r.opaque = u.opaque; r.cost = u.cost; r.walk = u.walk;
                r.ends = u.ends; r.stores = u.stores; r.quiets = u.quiets;
                r.lines = u.text;
                us.push_back(std::move(r));
                continue;
            }
            for (const Part &p : u.parts) {
                RUnit r;
                r.stateful = p.stateful; r.cost = p.cost; r.walk = p.walk;
                if (p.owner) { r.ends = u.ends; r.stores = u.stores; r.quiets = u.quiets; }
                r.lines = p.lines;
                us.push_back(std::move(r));
            }
        }
        for (RUnit &u : us) analyze(u, temps);
        const size_t n = us.size();
        // ---- (2) reaching definitions: deps[u] = (defining unit, temp)
        std::vector<std::vector<std::pair<size_t, std::string>>> deps(n);
        {
            std::map<std::string, size_t> last;
            for (size_t i = 0; i < n; i++) {
                for (const std::string &t : us[i].reads) { auto it = last.find(t); if (it != last.end()) deps[i].push_back({it->second, t}); }
                for (const std::string &t : us[i].writes) last[t] = i;
            }
        }
        // ---- (3) roles.  Role 0 is the writer.
        std::vector<Role> roles(1);
        auto reaches = [&](int a, int b) {                               // a path a -> ... -> b in the role graph
            std::vector<int> stack = {b};
            std::set<int> seen;
            while (!stack.empty()) {
                const int r = stack.back(); stack.pop_back();
                if (r == a) return true;
                if (!seen.insert(r).second) continue;
                for (int p : roles[(size_t)r].preds) stack.push_back(p);
            }
            return false;
        };
        for (size_t i = 0; i < n; i++) {
            RUnit &u = us[i];
            std::set<int> P;                                             // the roles it reads from (through floating units: what those read from)
            {
                std::vector<size_t> stack = {i};
                std::set<size_t> seen;
                while (!stack.empty()) {
                    const size_t x = stack.back(); stack.pop_back();
                    for (const auto &d : deps[x]) {
                        if (!us[d.first].floating) P.insert(us[d.first].role);
                        else if (seen.insert(d.first).second) stack.push_back(d.first);
                    }
                }
            }
            // cheap pure arithmetic FLOATS -- it is copied into every role that reads its result -- when it reads params and input
            // rows alone, or when the one role it could join is a recurrence (every instruction there is on the span's critical
            // path: a Filter's mix goes to the role that wants the mix, at the price of handing on l, b and h instead of one value)
            auto recurrence = [&](int r) { return r > 0 && roles[(size_t)r].cost > 0 && roles[(size_t)r].walk_sum == roles[(size_t)r].cost; };
            bool floats = !u.stateful && !u.opaque && !u.touches_o && u.cost < 8 && !P.count(0);
            if (floats && !P.empty()) {
                floats = false;
                std::vector<int> sinks;
                for (int r : P) {
                    bool sink = true;
                    for (int q : P) if (q != r && reaches(r, q)) sink = false;
                    if (sink) sinks.push_back(r);
                }
                if (sinks.size() == 1 && recurrence(sinks[0])) floats = true;
            }
            if (u.touches_o || P.count(0)) { u.role = 0; }
            else if (floats) { u.floating = true; continue; }
            else {
                const bool anchor = u.cost >= 12 || (u.stateful && P.empty());
                if (anchor && (int)roles.size() - 1 < kMaxProducers) {
                    roles.emplace_back();
                    u.role = (int)roles.size() - 1;
                } else {
                    std::vector<int> cand;
                    if (P.empty()) { for (int r = 1; r < (int)roles.size(); r++) cand.push_back(r); }
                    else {
                        for (int r : P) {
                            bool sink = true;
                            for (int q : P) if (q != r && reaches(r, q)) sink = false;
                            if (sink) cand.push_back(r);
                        }
                    }
                    if (cand.empty()) { roles.emplace_back(); u.role = (int)roles.size() - 1; }
                    else {
                        int best = cand[0];
                        for (int r : cand) if (roles[(size_t)r].cost < roles[(size_t)best].cost) best = r;
                        u.role = best;
                    }
                }
            }
            Role &R = roles[(size_t)u.role];
            for (int p : P) if (p != u.role) R.preds.insert(p);
            R.cost += u.cost;
            R.walk_sum += u.walk;
        }
        const int NR = (int)roles.size();
        if (NR < 2) return {};                                           // nothing but the writer: the lane form is the same thing
        // items: the role's units in body order, each preceded by the floating units it reads (once per role)
        {
            std::vector<std::set<size_t>> have((size_t)NR);
            // (recursive lambda through a std::function-free trick: explicit stack)
            for (size_t i = 0; i < n; i++) {
                if (us[i].floating) continue;
                const int r = us[i].role;
                std::vector<size_t> need, stack = {i};
                while (!stack.empty()) {
                    const size_t x = stack.back(); stack.pop_back();
                    for (const auto &d : deps[x])
                        if (us[d.first].floating && !have[(size_t)r].count(d.first)) { have[(size_t)r].insert(d.first); need.push_back(d.first); stack.push_back(d.first); }
                }
                for (size_t f : need) { roles[(size_t)r].items.push_back(f); roles[(size_t)r].cost += us[f].cost; }
                roles[(size_t)r].items.push_back(i);
            }
            // BODY ORDER, copies included: temps are reused names, and a copy run later than its place in the body would overwrite a
            // newer value of the name it writes (Bell: `t0 = freq` after `t0 = 1 + 0.03 * sine` -- tests/test_zangscript.py)
            for (Role &R : roles) std::sort(R.items.begin(), R.items.end());
        }
        // lags: longest path from the sources (a role that reads an input row or the live output sits behind its loader), then
        // as late as the consumers allow
        std::vector<std::set<size_t>> role_rows((size_t)NR);
        for (int r = 0; r < NR; r++) for (size_t i : roles[(size_t)r].items) role_rows[(size_t)r].insert(us[i].rows.begin(), us[i].rows.end());
        {
            std::vector<int> order;                                      // topological (preds first)
            std::vector<int> state((size_t)NR, 0);
            std::vector<std::pair<int, bool>> stack;
            for (int r = 0; r < NR; r++) stack.push_back({r, false});
            while (!stack.empty()) {
                auto [r, done] = stack.back(); stack.pop_back();
                if (done) { order.push_back(r); continue; }
                if (state[(size_t)r]) continue;
                state[(size_t)r] = 1;
                stack.push_back({r, true});
                for (int p : roles[(size_t)r].preds) if (!state[(size_t)p]) stack.push_back({p, false});
            }
            for (int r : order) {
                int lag = (!role_rows[(size_t)r].empty() || r == 0) ? 1 : 0;
                for (int p : roles[(size_t)r].preds) lag = std::max(lag, roles[(size_t)p].lag + 1);
                roles[(size_t)r].lag = lag;
            }
            for (auto it = order.rbegin(); it != order.rend(); ++it) {
                const int r = *it;
                int latest = -1;
                for (int s = 0; s < NR; s++) if (roles[(size_t)s].preds.count(r)) latest = latest < 0 ? roles[(size_t)s].lag - 1 : std::min(latest, roles[(size_t)s].lag - 1);
                if (latest > roles[(size_t)r].lag) roles[(size_t)r].lag = latest;
            }
        }
        const int max_lag = roles[0].lag;
        // ---- transfers
        std::vector<Transfer> tr;
        auto transfer = [&](long def, const std::string &var, int from, int to) -> size_t {
            for (size_t t = 0; t < tr.size(); t++) if (tr[t].def == def && tr[t].var == var) { tr[t].to.insert(to); return t; }
            Transfer x; x.def = def; x.var = var; x.from = from; x.to.insert(to);
            tr.push_back(x);
            return tr.size() - 1;
        };
        for (int r = 0; r < NR; r++) {
            for (size_t j : role_rows[(size_t)r]) transfer(-1 - (long)j, strf("x[%zu]", j), -1, r);
            for (size_t i : roles[(size_t)r].items)
                for (const auto &d : deps[i])
                    if (!us[d.first].floating && us[d.first].role != r) transfer((long)d.first, d.second, us[d.first].role, r);
        }
        const size_t oin = transfer(kOIn, "o", -1, 0);                   // last: the launch leaves its buffers out when the paint zeroes first
        size_t bufs_no_oin = 0, bufs = 0;
        for (size_t t = 0; t < tr.size(); t++) {
            Transfer &x = tr[t];
            int first = 1 << 30, last = 0;
            for (int r : x.to) { first = std::min(first, roles[(size_t)r].lag); last = std::max(last, roles[(size_t)r].lag); }
            x.src_lag = x.from < 0 ? first - 1 : roles[(size_t)x.from].lag;
            x.depth = last - x.src_lag + 1;
            x.off = bufs;
            bufs += (size_t)x.depth;
            if (t != oin) bufs_no_oin = bufs;
        }
        int ch = 32;
        while (ch > 4 && bufs * (size_t)ch * 256 > lds_budget) ch /= 2;      // a buffer is [ch / 4][64] float4
        if (bufs * (size_t)ch * 256 > lds_budget) return {};
        const size_t tile = (size_t)ch / 4 * 64;
        for (int r = 0; r < NR; r++) {
            Role &R = roles[(size_t)r];
            for (size_t t = 0; t < tr.size(); t++) {
                if (t != oin && tr[t].to.count(r)) R.tin.push_back(t);
                if (tr[t].from == r) R.tout.push_back(t);
            }
            if (r == 0) R.tin.push_back(oin);
        }
        // loader waves: up to four sources each
        std::vector<std::vector<size_t>> loaders;
        for (size_t t = 0; t < tr.size(); t++) {
            if (tr[t].from >= 0) continue;
            if (loaders.empty() || loaders.back().size() == 4) loaders.emplace_back();
            loaders.back().push_back(t);
        }
        // a role whose frames cost far more to compute than to walk runs in several waves, each computing every K-th group of
        // four frames (zs_role_run): until it is no slower than the slowest role that can not be split
        size_t nwaves = loaders.size() + (size_t)NR;
        {
            int pace = 14;
            std::vector<bool> can((size_t)NR, false);
            for (int r = 0; r < NR; r++) {
                Role &R = roles[(size_t)r];
                bool ok = r != 0;
                for (size_t i : R.items) { R.walk += us[i].walk; if (us[i].opaque) ok = false; }
                can[(size_t)r] = ok && R.walk * 2 <= R.cost;
                pace = std::max(pace, can[(size_t)r] ? R.walk : R.cost);
            }
            for (int r = 1; r < NR; r++) {
                Role &R = roles[(size_t)r];
                if (!can[(size_t)r]) continue;
                // (2 or 4 waves: a tile's groups of four frames must divide evenly, the step waits for the slowest wave)
                while (R.rep < 4 && R.rep * 2 <= ch / 4 && nwaves + (size_t)R.rep <= 16 && R.walk + (R.cost - R.walk) / R.rep > pace + pace / 8) { nwaves += (size_t)R.rep; R.rep *= 2; }
            }
        }
        int total = 0, longest = 0;
        for (int r = 0; r < NR; r++) { total += roles[(size_t)r].cost; longest = std::max(longest, roles[(size_t)r].walk + (roles[(size_t)r].cost - roles[(size_t)r].walk) / roles[(size_t)r].rep); }
        // worth it where the lane form can not take frame ranges and the longest role is well below the whole body
        const bool hint = (k.rings || k.walk_reads_computed) && longest * 10 <= total * 7;
        worth = hint;
        // ... and still where a wave per SIMD or more of the lane form fits the chip (above half of script_pc_maxv: script.hip): there
        // the role form's WORK counts, not only its longest role -- a role walked by K waves repeats its state steps K times -- and every
        // wave more is one more at the step's barrier.  Measured at 65,536 voices (profiles/r06/role_ab_65536.txt): the modules with
        // work <= 1.25 x the body's, at most 9 waves and a longest role under 2/7 of the body win (FilteredSawtooth 181 -> 110 us, Sweep,
        // Glide, Buzz), the others lose or tie (Hiss 256 -> 357: its noise is walked four times).
        int work = 0;
        for (int r = 0; r < NR; r++) work += roles[(size_t)r].rep * roles[(size_t)r].walk + (roles[(size_t)r].cost - roles[(size_t)r].walk);
        const bool hint_many = hint && work * 4 <= total * 5 && nwaves <= 9 && longest * 7 <= total * 2;

        // ---- text
        const std::string I = "    ";
        const char *nc = k.name.c_str();
        Lines out;
        out.push_back(strf("// role-wave form: %zu waves (%zu loader, %d roles, the last the writer), %d frames per tile, %zu tile buffers", nwaves, loaders.size(), NR, ch, bufs));
        out.push_back(strf("extern \"C\" __device__ const uint32_t zs_pc_info_%s[4] = {%zuu, %zuu, %zuu, %uu};", nc, nwaves * 64, bufs * (size_t)ch * 256,
                           bufs_no_oin * (size_t)ch * 256, (hint ? 1u : 0u) | (hint_many ? 2u : 0u)));
        out.push_back(strf("extern \"C\" __global__ void __launch_bounds__(%zu) zs_paint_pc_%s(const ZsLaunch L) {", nwaves * 64, nc));
        out.push_back(I + "extern __shared__ float4 zs_lds[];");
        out.push_back(I + "const uint32_t zs_lane = threadIdx.x & 63u, zs_role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));");
        out.push_back(I + "const uint32_t V = L.V;");
        out.push_back(I + "const uint32_t zs_v0 = blockIdx.x * 64 + zs_lane;");
        out.push_back(I + "const bool zs_live = zs_v0 < V;");
        out.push_back(I + "const uint32_t v = zs_live ? zs_v0 : V - 1;      // a lane past the last voice runs voice V - 1 again");
        append(out, preamble);
        out.push_back(I + "const bool zs_walk = false; (void)zs_walk;");
        out.push_back(I + "const uint32_t zs_nfr = L.end - L.start;");
        out.push_back(I + strf("const uint32_t zs_steps = (zs_nfr + %du) / %du + %du;", ch - 1, ch, max_lag));
        out.push_back(I + "const bool zs_zf = (L.flags & ZH_PAINT_ZERO_FIRST) != 0;");
        auto tile_list = [&](const std::vector<size_t> &ts) {
            std::string s;
            for (size_t t : ts) s += (s.empty() ? "" : ", ") + strf("{%zuu, %du}", tr[t].off * tile, tr[t].depth);
            return s.empty() ? std::string("{0u, 1u}") : s;
        };
        size_t wave = 0;
        for (const auto &ld : loaders) {
            out.push_back(I + (wave ? "} else if" : "if") + strf(" (zs_role == %zuu) {", wave));
            std::string src, str, vof, lag;
            for (size_t t : ld) {
                const bool o = tr[t].def == kOIn;
                const size_t j = o ? 0 : (size_t)(-1 - tr[t].def);
                src += (src.empty() ? "" : ", ") + (o ? std::string("L.out") : strf("ins[%zu]", j));
                str += (str.empty() ? "" : ", ") + (o ? std::string("zs_zf ? (size_t)0 : (size_t)L.ostride") : strf("istr[%zu]", j));
                vof += (vof.empty() ? "" : ", ") + (o ? std::string("v * 4u") : strf("ivo[%zu]", j));
                lag += (lag.empty() ? "" : ", ") + strf("%du", tr[t].src_lag);
            }
            out.push_back(I + I + strf("const ZsTileRef zs_to[%zu] = {", ld.size()) + tile_list(ld) + "};");
            out.push_back(I + I + strf("const float *zs_src[%zu] = {", ld.size()) + src + "};");
            out.push_back(I + I + strf("const size_t zs_str[%zu] = {", ld.size()) + str + "};");
            out.push_back(I + I + strf("const uint32_t zs_vof[%zu] = {", ld.size()) + vof + "};");
            out.push_back(I + I + strf("const uint32_t zs_lag[%zu] = {", ld.size()) + lag + "};");
            out.push_back(I + I + strf("zs_loader_run<%d, %zu>(zs_lds, zs_lane, zs_steps, L.start, zs_nfr, zs_to, zs_src, zs_str, zs_vof, zs_lag);", ch, ld.size()));
            wave++;
        }
        // the producer roles in index order, the writer last
        std::vector<int> seq;
        for (int r = 1; r < NR; r++) seq.push_back(r);
        seq.push_back(0);
        for (int r : seq) {
            const Role &R = roles[(size_t)r];
            if (R.rep == 1) out.push_back(I + (wave ? "} else if" : "if") + strf(" (zs_role == %zuu) {", wave) + strf("      // lag %d, ~%d instructions per frame", R.lag, R.cost));
            else out.push_back(I + (wave ? "} else if" : "if") + strf(" (zs_role >= %zuu && zs_role < %zuu) {", wave, wave + (size_t)R.rep) +
                               strf("      // lag %d, ~%d instructions per frame, ~%d of them state: %d waves, each computes every %s group of four frames", R.lag, R.cost, R.walk, R.rep,
                                    R.rep == 2 ? "second" : R.rep == 3 ? "third" : "fourth"));
            out.push_back(I + I + strf("const uint32_t zs_rep = zs_role - %zuu; (void)zs_rep;", wave));
            wave += (size_t)R.rep;
            const size_t NIN = R.tin.size(), NOUT = R.tout.size();
            out.push_back(I + I + strf("const ZsTileRef zs_ti[%zu] = {", NIN ? NIN : 1) + tile_list(R.tin) + "};");
            out.push_back(I + I + strf("const ZsTileRef zs_to[%zu] = {", NOUT ? NOUT : 1) + tile_list(R.tout) + "};");
            Lines quiet;
            for (size_t i : R.items) for (size_t q : us[i].quiets) quiet.push_back(k.quiet_terms[q]);
            std::string all;
            for (const std::string &t : quiet) all += (all.empty() ? "" : " && ") + t;
            const bool two = !quiet.empty();
            if (two) out.push_back(I + I + "auto zs_quiet = [&](int zs_n) ZH_INLINE_LAMBDA -> bool { return " + all + "; };");
            else out.push_back(I + I + "auto zs_quiet = [](int) ZH_INLINE_LAMBDA -> bool { return false; };");
            out.push_back(I + I + strf("auto zs_body = [&](auto zs_q, uint32_t i, const float (&zs_in)[%zu], float (&zs_out)[%zu], float &o) ZH_INLINE_LAMBDA {", NIN ? NIN : 1, NOUT ? NOUT : 1));
            const std::string B = I + I + I;
            out.push_back(B + "constexpr bool ZS_Q = decltype(zs_q)::value; (void)ZS_Q;");
            out.push_back(B + "(void)i; (void)zs_in; (void)zs_out; (void)o;");
            out.push_back(B + strf("float x[%zu]; (void)x;", ni));
            {
                std::string decl;
                for (const std::string &t : temps) decl += (decl.empty() ? "float " : ", ") + t + " = 0.0f";
                if (!decl.empty()) out.push_back(B + decl + ";");
                std::string use;
                for (const std::string &t : temps) use += "(void)" + t + "; ";
                if (!use.empty()) out.push_back(B + use);
            }
            std::set<size_t> loaded;
            for (size_t a = 0; a < NIN; a++) if (tr[R.tin[a]].def < 0 && tr[R.tin[a]].def != kOIn) { out.push_back(B + tr[R.tin[a]].var + strf(" = zs_in[%zu];", a)); loaded.insert(R.tin[a]); }
            for (size_t i : R.items) {
                for (const auto &d : deps[i]) {
                    if (us[d.first].floating || us[d.first].role == r) continue;
                    for (size_t a = 0; a < NIN; a++) {
                        const Transfer &x = tr[R.tin[a]];
                        if (x.def == (long)d.first && x.var == d.second && loaded.insert(R.tin[a]).second) out.push_back(B + x.var + strf(" = zs_in[%zu];", a));
                    }
                }
                for (const std::string &l : us[i].lines) out.push_back(B + resolve_sines(l, 0));
                for (size_t b = 0; b < NOUT; b++) if (tr[R.tout[b]].def == (long)i) out.push_back(B + strf("zs_out[%zu] = ", b) + tr[R.tout[b]].var + ";");
            }
            out.push_back(I + I + "};");
            // quads per unrolled step of a tile by body size (the lane form's rule, scaled: a quad is four frames)
            size_t body_lines = 0;
            for (size_t i : R.items) body_lines += us[i].lines.size();
            const int uq = body_lines <= 12 ? ch / 4 : body_lines <= 40 ? 2 : 1;
            const std::string lam = strf("[&](uint32_t i, const float (&a)[%zu], float (&b)[%zu], float &o) ZH_INLINE_LAMBDA ", NIN ? NIN : 1, NOUT ? NOUT : 1);
            out.push_back(I + I + strf("zs_role_run<%d, %zu, %zu, %d, %s, %d>(zs_lds, zs_lane, %du, zs_rep, zs_steps, L.start, zs_nfr, zs_ti, zs_to, L.out, L.ostride, v * 4u, zs_zf,", ch, NIN, NOUT,
                                       R.rep > 1 ? std::max(uq / R.rep * R.rep, R.rep) : uq, r == 0 ? "true" : "false", R.rep, R.lag));
            out.push_back(I + I + I + lam + "{ zs_body(zs_tag<false>{}, i, a, b, o); }, zs_quiet,");
            out.push_back(I + I + I + lam + strf("{ zs_body(zs_tag<%s>{}, i, a, b, o); });", two ? "true" : "false"));
            Lines fin;
            for (size_t i : R.items) for (size_t e : us[i].ends) fin.push_back(k.epi_ends[e]);
            for (size_t i : R.items) for (size_t e : us[i].stores) fin.push_back(k.epi_stores[e]);
            if (!fin.empty()) {
                out.push_back(I + I + (R.rep > 1 ? "if (zs_live && zs_rep == 0u) {" : "if (zs_live) {"));
                for (const std::string &l : fin) out.push_back(I + I + I + l);
                out.push_back(I + I + "}");
            }
        }
        out.push_back(I + "}");
        out.push_back("}");
        return out;
    }

    std::string generate(const std::set<std::string> *only, std::vector<HipModuleMeta> &meta, int unroll_override, unsigned forms) {
        Lines out = {"// generated by zang_amd.zangscript (HIP backend) -- compile with zh_script_load / zh_script_compile",
                     "#include \"script_rt.hip.h\"", ""};
        for (size_t ci = 0; ci < s.pr.curves.size(); ci++) {
            std::string pts;
            for (size_t i = 0; i < s.pr.curves[ci].points.size(); i++) {
                const auto &p = s.pr.curves[ci].points[i];
                pts += (i ? ", " : "") + ("{" + f32_literal(p.second.value) + ", " + f32_literal(p.first.value) + "}");       // {value, t}
            }
            out.push_back(strf("__device__ const zh_curve_node zs_curve%zu[] = {", ci) + (pts.empty() ? "{0.0f, 0.0f}" : pts) + "};");
        }
        std::set<size_t> used_tracks;
        const size_t table_at = out.size();
        for (const auto &em : s.exported_modules) {
            const std::string &name = em.first;
            if (only && !only->count(name)) continue;
            Kernel k;
            k.name = name;
            HipModuleMeta m;
            m.name = name;
            m.num_temps = s.module_results[em.second].num_temps;       // what the generated Zig struct declares (codegen_zig.zig:518)
            try {
                kernel(k, em.second);
            } catch (const HipBackendError &e) {
                m.error = e.message;
                meta.push_back(m);
                out.push_back("");
                out.push_back("// " + name + ": " + e.message);
                continue;
            }
            m.state_words = k.words; m.noise_fields = k.noise_fields; m.params = k.params;
            meta.push_back(m);
            used_tracks.insert(k.tracks.begin(), k.tracks.end());
            const size_t nin = k.rows.size(), ni = nin ? nin : 1;
            const int unroll = unroll_override ? unroll_override : (k.frame.size() <= 40 ? 8 : k.frame.size() <= 100 ? 4 : 2);
            const std::string I = "    ";
            const char *nc = name.c_str();
            out.push_back("");
            // 1: the paint kernel may be launched as frame ranges (script_rt.hip.h zs_frame_loop); 0: its frame body writes memory
            // (a delay ring), or replaying its state walk would cost as much as painting (see call_builtin)
            out.push_back(strf("extern \"C\" __device__ const uint32_t zs_ranges_ok_%s = %uu;", nc, (k.rings || k.walk_reads_computed) ? 0u : 1u));
            {
                // how many of the module's state words the kernel's epilogue stores.  A launch as frame ranges writes the end state into
                // the OTHER blob and the host flips (script.hip): that is only right when EVERY word is stored, so the loader takes the
                // range form only where this count equals the module's state words (the invariant was pinned by a CPU test
                // of the emitter alone).
                std::vector<bool> stored(k.words, false);
                static const std::regex re("zs_st_(f|u|u64)\\(L\\.state, ([0-9]+), V, v,");
                auto scan = [&](const Lines &ls) {
                    for (const std::string &ln : ls)
                        for (std::sregex_iterator it(ln.begin(), ln.end(), re), end; it != end; ++it) {
                            const size_t w = (size_t)std::stoul((*it)[2].str());
                            if (w < stored.size()) stored[w] = true;
                            if ((*it)[1].str() == "u64" && w + 1 < stored.size()) stored[w + 1] = true;
                        }
                };
                scan(k.epi_ends); scan(k.epi_stores);
                size_t n_stored = 0;
                for (bool b : stored) n_stored += b ? 1 : 0;
                out.push_back(strf("extern \"C\" __device__ const uint32_t zs_state_words_stored_%s = %zuu;", nc, n_stored));
            }
            out.push_back(strf("extern \"C\" __global__ void zs_init_%s(uint32_t *__restrict__ st, uint32_t V, uint64_t first_seed) {", nc));
            out.push_back(I + "const uint32_t v = blockIdx.x * 64 + threadIdx.x;");
            out.push_back(I + "if (v >= V) return;");
            out.push_back(I + strf("for (uint32_t w = 0; w < %zuu; w++) st[(size_t)w * V + v] = 0u;", k.words));
            for (const InitItem &it : k.init) {
                if (!it.noise) {
                    out.push_back(I + strf("zs_st_f(st, %zu, V, v, ", it.word) + f32_literal(it.value) + ");");
                } else {                                          // Noise.zig:25-32: seed = counter++ at init()
                    out.push_back(I + strf("{ ZXoshiro r; zxoshiro_seed(r, first_seed + (uint64_t)v * %zuu + %zuu);", k.noise_fields, it.k));
                    out.push_back(I + strf("  zs_st_u64(st, %zu, V, v, r.s0); zs_st_u64(st, %zu, V, v, r.s1); zs_st_u64(st, %zu, V, v, r.s2); zs_st_u64(st, %zu, V, v, r.s3); }",
                                           it.word, it.word + 2, it.word + 4, it.word + 6));
                }
            }
            out.push_back("}");
            out.push_back("");
            out.push_back(strf("extern \"C\" __global__ void __launch_bounds__(64) zs_paint_%s(const ZsLaunch L) {", nc));
            out.push_back(I + "const uint32_t v = blockIdx.x * 64 + threadIdx.x;");
            out.push_back(I + "const uint32_t V = L.V;");
            out.push_back(I + "if (v >= V) return;");
            Lines preamble;                                              // (shared with the role-wave form)
            preamble.push_back(I + "const bool NIC = L.nic.get(v);");
            preamble.push_back(I + "const uint32_t SPAN_LEN = L.end - L.start;");
            preamble.push_back(I + "(void)NIC; (void)SPAN_LEN;");
            std::string nulls, zeros;
            for (size_t j = 0; j < ni; j++) { nulls += j ? ", nullptr" : "nullptr"; zeros += j ? ", 0" : "0"; }
            preamble.push_back(I + strf("const float *ins[%zu] = {", ni) + nulls + "};");
            preamble.push_back(I + strf("size_t istr[%zu] = {", ni) + zeros + "};");
            preamble.push_back(I + strf("uint32_t ivo[%zu] = {", ni) + zeros + "};");
            for (size_t j = 0; j < k.rows.size(); j++) preamble.push_back(I + strf("ins[%zu] = zs_row(L.p[%zu], v, istr[%zu], ivo[%zu]);", j, k.rows[j], j, j));
            append(preamble, indent(k.pro));
            append(out, preamble);
            out.push_back(I + "bool zs_walk = false; (void)zs_walk;");
            const bool two_bodies = !k.quiet_terms.empty();
            const std::string loop_call = I +
                strf("zs_frame_loop<%d, %zu>(L.out, v, L.ostride, ins, istr, ivo, L.start, L.end, (L.flags & ZH_PAINT_ZERO_FIRST) != 0, zs_walk,", unroll, nin);
            // the sine sources that reach no sink: under ZH_PAINT_TOLERANT their f32 form (a second instance of the frame body,
            // chosen once per paint: ZS_T).  A kernel without one reads as before.
            uint64_t tolerant = 0;
            for (size_t i = 0; i < k.nsines; i++) if (!((k.exact_sines >> i) & 1)) tolerant |= 1ull << i;
            Lines frame_lines;
            for (const std::string &l : k.frame) frame_lines.push_back(resolve_sines(l, tolerant));
            const std::string lam = I + strf("                     [&](uint32_t i, const float (&x)[%zu], float &o) ZH_INLINE_LAMBDA ", ni);
            std::string all;
            for (const std::string &t : k.quiet_terms) all += (all.empty() ? "" : " && ") + t;
            if (tolerant) {
                if (two_bodies) out.push_back(I + "auto zs_quiet = [&](int zs_n) ZH_INLINE_LAMBDA -> bool { return " + all + "; };");
                out.push_back(I + strf("auto zs_body = [&](auto zs_q, auto zs_t, uint32_t i, const float (&x)[%zu], float &o) ZH_INLINE_LAMBDA {", ni));
                out.push_back(I + I + "constexpr bool ZS_Q = decltype(zs_q)::value; (void)ZS_Q;");
                out.push_back(I + I + "constexpr bool ZS_T = decltype(zs_t)::value; (void)ZS_T;");
            } else if (two_bodies) {
                out.push_back(I + "auto zs_quiet = [&](int zs_n) ZH_INLINE_LAMBDA -> bool { return " + all + "; };");
                out.push_back(I + strf("auto zs_body = [&](auto zs_q, uint32_t i, const float (&x)[%zu], float &o) ZH_INLINE_LAMBDA {", ni));
                out.push_back(I + I + "constexpr bool ZS_Q = decltype(zs_q)::value; (void)ZS_Q;");
            } else {
                out.push_back(loop_call);
                out.push_back(lam + "{");
            }
            out.push_back(I + I + "(void)i; (void)x;");
            if (!k.temps.empty()) {
                std::string decl = "float ";
                for (size_t j = 0; j < k.temps.size(); j++) decl += (j ? ", " : "") + k.temps[j] + " = 0.0f";
                out.push_back(I + I + decl + ";");
            }
            append(out, indent(indent(frame_lines)));
            if (tolerant) {
                out.push_back(I + "};");
                for (const char *t : {"true", "false"}) {
                    out.push_back(I + (std::string(t) == "true" ? "if (L.flags & ZH_PAINT_TOLERANT) {" : "} else {"));
                    out.push_back(loop_call);
                    if (two_bodies) {
                        out.push_back(lam + strf("{ zs_body(zs_tag<false>{}, zs_tag<%s>{}, i, x, o); }, zs_quiet,", t));
                        out.push_back(lam + strf("{ zs_body(zs_tag<true>{}, zs_tag<%s>{}, i, x, o); });", t));
                    } else {
                        out.push_back(lam + strf("{ zs_body(zs_tag<false>{}, zs_tag<%s>{}, i, x, o); });", t));
                    }
                }
                out.push_back(I + "}");
            } else if (two_bodies) {
                out.push_back(I + "};");
                out.push_back(loop_call);
                out.push_back(lam + "{ zs_body(zs_tag<false>{}, i, x, o); }, zs_quiet,");
                out.push_back(lam + "{ zs_body(zs_tag<true>{}, i, x, o); });");
            } else {
                out.push_back(I + "});");
            }
            append(out, indent(k.epi_ends));
            append(out, indent(k.epi_stores));
            out.push_back("}");
            if (forms & (ZH_ZSCRIPT_FORM_ROLES | ZH_ZSCRIPT_FORM_ROLES_WORTH)) {
                bool worth = false;
                const Lines pc = role_kernel(k, nin, ni, preamble, (forms & 4u) ? 131072 : 65536, worth);    // (bit 2: an experiment, tools/r06_roles2.sh)
                if (!pc.empty() && ((forms & ZH_ZSCRIPT_FORM_ROLES) || worth)) { out.push_back(""); append(out, pc); }
            }
        }
        Lines tables;
        for (size_t ti : used_tracks) append(tables, track_tables(ti));
        out.insert(out.begin() + (long)table_at, tables.begin(), tables.end());
        std::string text;
        for (const std::string &l : out) text += l + "\n";
        return text;
    }
};

}  // namespace

std::string generate_zig(const CompiledScript &cs) { return ZigEmitter(cs).generate(); }
std::string generate_hip(const CompiledScript &cs, const std::set<std::string> *only, std::vector<HipModuleMeta> &meta, int unroll_override, unsigned forms) {
    return HipEmitter(cs).generate(only, meta, unroll_override, forms);
}

}  // namespace zs

// ================================================================== C ABI
struct zh_zscript {
    std::unique_ptr<zs::CompiledScript> cs;
    std::vector<zs::HipModuleMeta> meta;         // of the last zh_zscript_generate_hip
};

static void put_text(char *dst, size_t cap, const std::string &text) {
    if (!dst || cap == 0) return;
    const size_t n = text.size() < cap - 1 ? text.size() : cap - 1;
    memcpy(dst, text.data(), n);
    dst[n] = 0;
}
static char *dup_text(const std::string &s) {
    char *p = (char *)malloc(s.size() + 1);
    if (p) memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

extern "C" {

int zh_zscript_compile(const char *text, const char *filename, uint32_t packages, zh_zscript **out, char *err, size_t err_cap) {
    if (!text || !out) return ZH_ERR_INVALID;
    *out = nullptr;
    std::vector<const zs::Package *> pk;
    if (packages & 1u) pk.push_back(&zs::zang_builtin_package());
    if (packages & 2u) pk.push_back(&zs::modules_builtin_package());
    try {
        zh_zscript *z = new zh_zscript();
        try {
            z->cs = zs::compile(text, filename ? filename : "script.txt", pk);
        } catch (...) { delete z; throw; }
        *out = z;
        return ZH_OK;
    } catch (const zs::ScriptError &e) {
        put_text(err, err_cap, e.rendered);
        return ZH_ERR_INVALID;
    } catch (const std::exception &e) {
        put_text(err, err_cap, e.what());
        return ZH_ERR_INVALID;
    }
}
int zh_zscript_destroy(zh_zscript *z) { delete z; return ZH_OK; }
void zh_zscript_free_text(char *text) { free(text); }

int zh_zscript_generate_zig(zh_zscript *z, char **text_out) {
    if (!z || !text_out) return ZH_ERR_INVALID;
    *text_out = dup_text(zs::generate_zig(*z->cs));
    return *text_out ? ZH_OK : ZH_ERR_INVALID;
}
int zh_zscript_generate_hip(zh_zscript *z, const char *only_csv, int unroll, char **text_out) {
    return zh_zscript_generate_hip_forms(z, only_csv, unroll, 0u, text_out);
}
int zh_zscript_generate_hip_forms(zh_zscript *z, const char *only_csv, int unroll, uint32_t forms, char **text_out) {
    if (!z || !text_out) return ZH_ERR_INVALID;
    std::set<std::string> only;
    if (only_csv) {
        std::string cur;
        for (const char *p = only_csv;; p++) {
            if (*p == ',' || *p == 0) { if (!cur.empty()) only.insert(cur); cur.clear(); if (!*p) break; }
            else cur += *p;
        }
    }
    z->meta.clear();
    *text_out = dup_text(zs::generate_hip(*z->cs, only_csv ? &only : nullptr, z->meta, unroll, forms));
    return *text_out ? ZH_OK : ZH_ERR_INVALID;
}
uint32_t zh_zscript_module_count(zh_zscript *z) { return z ? (uint32_t)z->meta.size() : 0; }
int zh_zscript_module_info(zh_zscript *z, uint32_t i, char *name, size_t name_cap, uint32_t *state_words, uint32_t *noise_fields,
                           uint32_t *n_params, char *error, size_t error_cap) {
    if (!z || i >= z->meta.size()) return ZH_ERR_INVALID;
    const zs::HipModuleMeta &m = z->meta[i];
    put_text(name, name_cap, m.name);
    put_text(error, error_cap, m.error);
    if (state_words) *state_words = (uint32_t)m.state_words;
    if (noise_fields) *noise_fields = (uint32_t)m.noise_fields;
    if (n_params) *n_params = (uint32_t)m.params.size();
    return ZH_OK;
}
int zh_zscript_module_num_temps(zh_zscript *z, uint32_t i, uint32_t *num_temps) {
    if (!z || i >= z->meta.size() || !num_temps) return ZH_ERR_INVALID;
    *num_temps = (uint32_t)z->meta[i].num_temps;
    return ZH_OK;
}
int zh_zscript_module_param(zh_zscript *z, uint32_t i, uint32_t p, char *name, size_t name_cap, char *kind, size_t kind_cap, char *enum_name, size_t enum_cap) {
    if (!z || i >= z->meta.size() || p >= z->meta[i].params.size()) return ZH_ERR_INVALID;
    const zs::HipParam &hp = z->meta[i].params[p];
    put_text(name, name_cap, hp.name);
    put_text(kind, kind_cap, hp.kind);
    put_text(enum_name, enum_cap, hp.enum_name);
    return ZH_OK;
}

}  // extern "C"
