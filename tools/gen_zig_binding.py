#!/usr/bin/env python3
"""gen_zig_binding.py -- writes bindings/zang_hip.zig, the reference-side binding of libzang_hip.so, FROM
include/zang_hip.h: every ZH_API function as `pub extern fn`, every struct as `extern struct`, every enum value and
#define as a constant, every opaque handle, and -- data-driven from the params structs -- for each module

  * `<Module>`      a batch of n GPU voices with the reference's module shape (src/modules/SineOsc.zig:8-31:
                    num_outputs, num_temps, Params, init, paint(span, outputs, temps, note_id_changed, params)),
                    outputs / temps being device images (zh_buf) instead of host []f32;
  * `<Module>Host`  the literal one-voice drop-in over zh_<module>_paint_host: host []f32 slices, `state` = the
                    Zig struct's fields -- for the ten north-star modules.

    python tools/gen_zig_binding.py            rewrite bindings/zang_hip.zig
    python tools/gen_zig_binding.py --check    exit 1 if the checked-in file differs from what would be written

The output is NOT compiled in this repository (no Zig toolchain in the image); tests/test_zig_binding.py parses it
back and holds it against the header: symbol sets, argument counts, struct field order, struct sizes (vs gcc).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zang_hip.h")
OUT = os.path.join(ROOT, "bindings", "zang_hip.zig")

SCALARS = {"int": "c_int", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32", "double": "f64",
           "size_t": "usize", "char": "u8", "long": "c_long"}
WORDS = {"sineosc": "SineOsc", "pulseosc": "PulseOsc", "trisawosc": "TriSawOsc", "pmosc": "PMOsc", "f32": "F32", "cob": "Cob",
         "hcob": "HCob", "hcurve": "HCurve", "iap": "Iap", "zscript": "ZScript", "ipc": "Ipc"}

# (module key, Zig wrapper name, num_temps of the reference module, reference file, extra create args)
MODULES = [
    ("sineosc", "SineOsc", 0, "src/modules/SineOsc.zig", []),
    ("pulseosc", "PulseOsc", 0, "src/modules/PulseOsc.zig", []),
    ("trisawosc", "TriSawOsc", 0, "src/modules/TriSawOsc.zig", []),
    ("noise", "Noise", 0, "src/modules/Noise.zig", [("first_seed", "u64")]),
    ("envelope", "Envelope", 0, "src/modules/Envelope.zig", []),
    ("gate", "Gate", 0, "src/modules/Gate.zig", []),
    ("filter", "Filter", 0, "src/modules/Filter.zig", []),
    ("sampler", "Sampler", 0, "src/modules/Sampler.zig", []),
    ("decimator", "Decimator", 0, "src/modules/Decimator.zig", []),
    ("distortion", "Distortion", 0, "src/modules/Distortion.zig", []),
    ("curve_module", "CurveModule", 0, "src/modules/Curve.zig", []),
    ("cycle", "Cycle", 0, "src/modules/Cycle.zig", []),
    ("portamento", "Portamento", 0, "src/modules/Portamento.zig", []),
    ("nice", "NiceInstrument", 2, "examples/modules.zig:189-248", [("color", "F32")]),
    ("pmosc", "PMOscInstrument", 3, "examples/modules.zig:80-128", [("release_duration", "F32")]),
    ("delay", "SimpleDelay", 0, "examples/modules.zig:341-386", [("delay_samples", "u32")]),
    ("filtered_echoes", "FilteredEchoes", 2, "examples/modules.zig:390-461", [("delay_samples", "u32")]),
    ("noise_filter", "NoiseFilter", 1, "examples/example_stereo.zig:71-82", [("first_seed", "u64")]),
]
HOST_MODULES = ["sineosc", "pulseosc", "trisawosc", "noise", "envelope", "gate", "filter", "sampler", "decimator", "distortion"]
# pointer parameters that address MANY elements (everything else of struct type points at one)
MANY_NAMES = {"outputs", "temps", "host", "filter", "impulses"}
MANY_SPECIAL = {("zh_script_module_paint", "params"), ("zh_polyphony_dispatcher_dispatch", "out"),
                # the batch call reads n_buffers elements of each
                ("zh_nice_paint_mix_stereo_batch", "note_id_changed"), ("zh_nice_paint_mix_stereo_batch", "params")}
# scalar pointer parameters that are a single out value, not an array
ONE_SCALAR = {"ms", "state_words", "noise_fields", "n_params", "num_temps", "code_size_out", "default_value", "current_value"}


def camel(name):
    assert name.startswith("zh_"), name
    return "".join(WORDS.get(w, w.capitalize()) for w in name[3:].split("_"))


# ------------------------------------------------------------------------------------------------ header -> model
def strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return text


def parse_header(path=HEADER):
    raw = open(path).read()
    text = strip(raw)
    defines = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(ZH_[A-Z0-9_]+)\s+(\d+)\s*$", text, flags=re.M)]
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = text.replace('extern "C" {', "")
    # statements at brace depth 0
    stmts, depth, cur = [], 0, ""
    for ch in text:
        cur += ch
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif ch == ";" and depth == 0:
            stmts.append(" ".join(cur.split()))
            cur = ""
    model = {"defines": defines, "enums": [], "opaques": [], "structs": [], "aliases": [], "functions": []}
    struct_names = set()
    for s in stmts:
        s = s.lstrip("} ").strip()
        if s.startswith("enum"):
            body = s[s.index("{") + 1:s.rindex("}")]
            val = -1
            for item in [x.strip() for x in body.split(",") if x.strip()]:
                if "=" in item:
                    name, v = [x.strip() for x in item.split("=")]
                    val = int(v, 0)
                else:
                    name, val = item, val + 1
                model["enums"].append((name, val))
        elif s.startswith("typedef struct") and "{" in s:
            m = re.match(r"typedef struct (\w+) \{(.*)\} (\w+) ?;", s)
            assert m and m.group(1) == m.group(3), s
            fields = []
            for decl in [d.strip() for d in m.group(2).split(";") if d.strip()]:
                fields += parse_decl(decl)
            model["structs"].append((m.group(1), fields))
            struct_names.add(m.group(1))
        elif s.startswith("typedef struct"):
            m = re.match(r"typedef struct (\w+) (\w+) ?;", s)
            assert m and m.group(1) == m.group(2), s
            model["opaques"].append(m.group(1))
        elif s.startswith("typedef"):
            m = re.match(r"typedef (\w+) (\w+) ?;", s)
            assert m, s
            model["aliases"].append((m.group(2), m.group(1)))
        elif s.startswith("ZH_API"):
            m = re.match(r"ZH_API (.*?)\b(zh_\w+) ?\((.*)\) ?;", s)
            assert m, s
            ret = m.group(1).strip()
            args = []
            if m.group(3).strip() not in ("", "void"):
                for a in split_args(m.group(3)):
                    f = parse_decl(a)
                    assert len(f) == 1, a
                    args.append(f[0])
            model["functions"].append((m.group(2), ret, args))
        elif s in ("", ";") or s.startswith("}"):
            continue
        else:
            raise SystemExit("gen_zig_binding: cannot parse: " + s[:120])
    model["struct_names"] = struct_names | {a for a, _ in model["aliases"]}
    return model


def split_args(s):
    return [a.strip() for a in s.split(",") if a.strip()]


def parse_decl(decl):
    """`const float *a, *b` / `uint64_t r[4]` / `float *const *outputs` -> [(name, ctype, array_len)] with ctype a
    normalised string like 'const float *' or 'float *const *'."""
    decl = " ".join(decl.replace("*", " * ").split())
    toks = decl.split(" ")
    # base type = leading tokens up to (not including) the first declarator
    i = 0
    base = []
    while i < len(toks) and (toks[i] in ("const", "unsigned", "struct") or not base or (base and base[-1] in ("const", "unsigned", "struct"))):
        base.append(toks[i])
        i += 1
    rest = " ".join(toks[i:])
    out = []
    for d in [x.strip() for x in rest.split(",")]:
        m = re.match(r"^((?:\* ?(?:const )?)*) ?(\w+) ?(?:\[ ?(\d+) ?\])?$", d)
        assert m, (decl, d)
        ptr = m.group(1).replace(" ", "").replace("const", " const ").strip()
        ptr = " ".join(ptr.replace("*", " * ").split()).replace("* const", "*const")
        ctype = " ".join(base) + ((" " + ptr) if ptr else "")
        out.append((m.group(2), ctype.strip(), int(m.group(3)) if m.group(3) else 0))
    return out


# ------------------------------------------------------------------------------------------------ C type -> Zig type
def zig_type(ctype, model, name="", func="", field=False):
    t = ctype.strip()
    if t == "void":
        return "void"
    if t in SCALARS:
        return SCALARS[t]
    if t in model["struct_names"]:
        return camel(t)
    if t == "float *const *":
        return "[*]const ?[*]f32"
    if t == "char * *":
        return "*?[*:0]u8"
    if t == "const char * *":
        return "?*?[*:0]const u8"
    if t == "void * *":
        return "*?*anyopaque"
    m = re.match(r"^(const )?(\w+) \*$", t)
    if m:
        const, base = bool(m.group(1)), m.group(2)
        c = "const " if const else ""
        if base == "void":
            return "?*" + c + "anyopaque"
        if base == "char":
            return "?[*:0]const u8" if const else "?[*]u8"
        if base in model["opaques_set"]:
            return "?*" + c + opaque_name(base)
        if base in SCALARS:
            one = name in ONE_SCALAR or name.endswith("_out")
            return ("?*" if one else "?[*]") + c + SCALARS[base]
        if base in model["struct_names"]:
            many = name in MANY_NAMES or (func, name) in MANY_SPECIAL or field
            return ("?[*]" if many else "?*") + c + camel(base)
    m = re.match(r"^(\w+) \* \*$", t)
    if m and m.group(1) in model["opaques_set"]:
        return "*?*" + opaque_name(m.group(1))
    raise SystemExit(f"gen_zig_binding: no Zig type for C type '{ctype}' ({func} {name})")


WRAPPED = {"zh_" + k for k, *_ in MODULES}


def opaque_name(c):
    return camel(c) + ("Handle" if c in WRAPPED else "")


def zig_default(ztype, n):
    if n:
        inner = zig_default(ztype, 0)
        return "[_]%s{%s} ** %d" % (ztype, inner, n)
    if ztype.startswith("?"):
        return "null"
    if ztype in ("u8", "u32", "i32", "u64", "usize", "c_int", "f32"):
        return "0"
    return "std.mem.zeroes(%s)" % ztype


# ------------------------------------------------------------------------------------------------ emit
def emit(model):
    model["opaques_set"] = set(model["opaques"])
    structs = dict(model["structs"])
    W = []
    w = W.append
    w("""// zang_hip.zig -- reference-side binding of libzang_hip.so (include/zang_hip.h).
//
// GENERATED by tools/gen_zig_binding.py from include/zang_hip.h -- do not edit; rerun the generator.
// tests/test_zig_binding.py holds this file against the header (symbols, argument counts, struct fields and sizes).
//
// This is the file a zang maintainer adds to the Zig tree (e.g. as src/zang_hip.zig, linked with
// `exe.linkSystemLibrary("zang_hip")`, INTEGRATION.md).  It has NEVER BEEN COMPILED: the build image of this
// repository has no Zig toolchain (DESIGN.md 1).  It is written for Zig 0.12 / 0.13, the reference's own versions.
//
// Three layers:
//   1. the C ABI verbatim: `pub extern fn zh_*`, `extern struct`s, constants, opaque handles;
//   2. `<Module>`: a BATCH of n voices on the GPU behind zang's module interface (src/modules/SineOsc.zig:8-31)
//          paint(self, span, outputs, temps, note_id_changed, params)
//      where outputs / temps are device images laid out [frame][voice] (Buf) instead of host []f32 slices;
//   3. `<Module>Host`: the literal one-voice drop-in -- host []f32 slices, the Zig struct's own state fields --
//      over zh_<module>_paint_host (staged through the device, synchronous: for porting and checking, not speed).

const std = @import("std");
""")
    w("// ---------------------------------------------------------------- constants (enum values and #defines of the header)")
    for name, val in model["enums"]:
        ty = "c_int" if name.startswith("ZH_ERR") or name == "ZH_OK" else "u32"
        w(f"pub const {name[3:]}: {ty} = {val};")
    for name, val in model["defines"]:
        w(f"pub const {name[3:]}: u32 = {val};")
    w("")
    w("// ---------------------------------------------------------------- opaque handles")
    for o in model["opaques"]:
        w(f"pub const {opaque_name(o)} = opaque {{}};")
    w("")
    w("// ---------------------------------------------------------------- structs (C layout; every field defaulted to zero / null)")
    alias_at = {}
    for new, old in model["aliases"]:
        alias_at.setdefault(old, []).append(new)
    for sname, fields in model["structs"]:
        w(f"pub const {camel(sname)} = extern struct {{")
        for fname, ctype, n in fields:
            zt = zig_type(ctype, model, fname, sname, field=True)
            full = f"[{n}]{zt}" if n else zt
            w(f"    {zig_field(fname)}: {full} = {zig_default(zt, n)},")
        w("};")
        for new in alias_at.get(sname, []):
            w(f"pub const {camel(new)} = {camel(sname)};")
    w("")
    w("// ---------------------------------------------------------------- the C ABI")
    for fname, ret, args in model["functions"]:
        zargs = ", ".join(f"{zig_field(a)}: {zig_type(ct, model, a, fname)}" for a, ct, _ in args)
        zret = zig_type(ret, model, "", fname) if ret != "const char *" else "?[*:0]const u8"
        if ret == "void *":
            zret = "?*anyopaque"
        w(f"pub extern fn {fname}({zargs}) {zret};")
    w("")
    w(HELPERS)
    for key, zname, ntemps, ref, extra in MODULES:
        w(emit_module(model, structs, key, zname, ntemps, ref, extra))
    w("// ---------------------------------------------------------------- literal one-voice drop-ins (host []f32 slices)")
    w(HOST_HELPERS)
    for key in HOST_MODULES:
        zname = next(z for k, z, *_ in MODULES if k == key)
        ref = next(r for k, _, _, r, _ in MODULES if k == key)
        w(emit_host_module(model, structs, key, zname, ref))
    return "\n".join(W).rstrip("\n") + "\n"


ZIG_KEYWORDS = {"type", "error", "test", "var", "const", "fn", "struct", "enum", "union", "align", "export", "resume", "suspend"}


def zig_field(name):
    return '@"%s"' % name if name in ZIG_KEYWORDS else name


HELPERS = """// ---------------------------------------------------------------- helpers
/// zang.Span (src/zang/basics.zig:3-10)
pub const Span = struct {
    start: usize,
    end: usize,

    pub inline fn init(start: usize, end: usize) Span {
        return .{ .start = start, .end = end };
    }
};

/// zang.constant(x) (src/zang/constant_or_buffer.zig:9-11): one value for every voice
pub fn constant(x: f32) Cob {
    return .{ .tag = COB_CONSTANT, .constant = .{ .value = x } };
}
/// a constant that differs per voice: device float[n_voices]
pub fn constantPerVoice(xs: [*]const f32) Cob {
    return .{ .tag = COB_CONSTANT, .constant = .{ .per_voice = xs } };
}
/// zang.buffer(b) (constant_or_buffer.zig:13-15): a device image, indexed by absolute frame
pub fn buffer(b: Buf) Cob {
    return .{ .tag = COB_BUFFER, .buffer = b };
}
pub fn f32All(x: f32) F32 {
    return .{ .value = x };
}
pub fn f32PerVoice(xs: [*]const f32) F32 {
    return .{ .per_voice = xs };
}
pub fn boolAll(x: bool) Bool {
    return .{ .value = @intFromBool(x) };
}
pub fn boolPerVoice(xs: [*]const u8) Bool {
    return .{ .per_voice = xs };
}
/// zang.PaintCurve (src/zang/painter.zig:25-30)
pub const PaintCurve = struct {
    pub const instantaneous = Curve{ .tag = CURVE_INSTANTANEOUS };
    pub fn linear(duration: f32) Curve {
        return .{ .tag = CURVE_LINEAR, .duration = .{ .value = duration } };
    }
    pub fn squared(duration: f32) Curve {
        return .{ .tag = CURVE_SQUARED, .duration = .{ .value = duration } };
    }
    pub fn cubed(duration: f32) Curve {
        return .{ .tag = CURVE_CUBED, .duration = .{ .value = duration } };
    }
};

fn check(rc: c_int) void {
    // paint() cannot fail in zang (it returns void); an error here is a programming error
    if (rc != 0) std.debug.panic("zang_hip: error {d}: {s}", .{ rc, std.mem.span(zh_error_string(rc).?) });
}

// ---------------------------------------------------------------- modules: n voices on the GPU behind zang's module interface
"""

HOST_HELPERS = """/// host-side zang.ConstantOrBuffer over a []const f32 slice
pub fn hconstant(x: f32) HCob {
    return .{ .tag = COB_CONSTANT, .constant = x };
}
pub fn hbuffer(b: []const f32) HCob {
    return .{ .tag = COB_BUFFER, .buffer = b.ptr };
}
"""


def params_fields(model, structs, sname):
    """(zig field name, zig type) of the user-facing Params: the C params struct without its padding fields."""
    out = []
    for fname, ctype, n in structs[sname]:
        if fname.startswith("reserved"):
            continue
        zt = zig_type(ctype, model, fname, sname, field=True)
        out.append((fname, f"[{n}]{zt}" if n else zt))
    return out


def emit_module(model, structs, key, zname, ntemps, ref, extra):
    h = opaque_name("zh_" + key)
    P = camel(f"zh_{key}_params")
    fields = params_fields(model, structs, f"zh_{key}_params")
    has_state = f"zh_{key}_state" in structs
    S = camel(f"zh_{key}_state")
    L = []
    a = L.append
    a(f"/// `{zname}` ({ref}) for a batch of n voices on the GPU.")
    a(f"pub const {zname} = struct {{")
    a("    pub const num_outputs = 1;")
    a(f"    pub const num_temps = {ntemps};")
    a("    pub const Params = struct {")
    for f, t in fields:
        a(f"        {zig_field(f)}: {t},")
    a("    };")
    a("")
    a(f"    handle: *{h},")
    a("")
    ex_decl = "".join(f", {n}: {t}" for n, t in extra)
    ex_call = "".join(f", {n}" for n, _ in extra)
    a(f"    pub fn init(ctx: *Ctx, n_voices: u32{ex_decl}) {zname} {{")
    a(f"        var h: ?*{h} = null;")
    a(f"        check(zh_{key}_create(ctx, n_voices{ex_call}, &h));")
    a("        return .{ .handle = h.? };")
    a("    }")
    a(f"    pub fn deinit(self: *{zname}) void {{")
    a(f"        check(zh_{key}_destroy(self.handle));")
    a("    }")
    if has_state:
        a(f"    /// the n Zig structs' fields, voice by voice (host array of n)")
        a(f"    pub fn getState(self: *{zname}, host: [*]{S}) void {{")
        a(f"        check(zh_{key}_get_state(self.handle, host));")
        a("    }")
        a(f"    pub fn setState(self: *{zname}, host: [*]const {S}) void {{")
        a(f"        check(zh_{key}_set_state(self.handle, host));")
        a("    }")
    a(f"    pub fn paint(self: *{zname}, span: Span, outputs: [num_outputs]Buf, temps: [num_temps]Buf, note_id_changed: bool, params: Params) void {{")
    a("        self.paintFlags(span, outputs, temps, boolAll(note_id_changed), params, PAINT_ADD);")
    a("    }")
    a("    /// `note_id_changed` per voice, and the paint flags (PAINT_ZERO_FIRST fuses the caller's zang.zero)")
    a(f"    pub fn paintFlags(self: *{zname}, span: Span, outputs: [num_outputs]Buf, temps: [num_temps]Buf, note_id_changed: Bool, params: Params, flags: u32) void {{")
    a("        _ = temps; // the fused kernels keep their temps in registers")
    inits = ", ".join(f".{zig_field(f)} = params.{zig_field(f)}" for f, _ in fields)
    a(f"        const p = {P}{{ {inits} }};")
    a(f"        check(zh_{key}_paint(self.handle, @intCast(span.start), @intCast(span.end), &outputs, null, note_id_changed, &p, flags));")
    a("    }")
    a("};")
    a("")
    return "\n".join(L)


def emit_host_module(model, structs, key, zname, ref):
    sname = f"zh_{key}_host_params"
    real = dict(model["aliases"]).get(sname, sname)
    P = camel(sname)
    has_state = f"zh_{key}_state" in structs
    S = camel(f"zh_{key}_state")
    fields = []
    for fname, ctype, n in structs[real]:
        zt = zig_type(ctype, model, fname, real, field=True)
        if zt == "?[*]const f32":
            fields.append((fname, "[]const f32", f"params.{zig_field(fname)}.ptr"))
        elif zt == "?[*]const u8":
            fields.append((fname, "[]const u8", f"params.{zig_field(fname)}.ptr"))
        elif fname in ("note_on", "loop"):
            fields.append((fname, "bool", f"@intFromBool(params.{zig_field(fname)})"))
        else:
            fields.append((fname, zt, f"params.{zig_field(fname)}"))
    L = []
    a = L.append
    a(f"/// Literal drop-in for `mod.{zname}` ({ref}): one voice, host slices, the Zig struct's state.")
    a(f"pub const {zname}Host = struct {{")
    a("    pub const num_outputs = 1;")
    a("    pub const num_temps = 0;")
    a("    pub const Params = struct {")
    for f, t, _ in fields:
        a(f"        {zig_field(f)}: {t},")
    a("    };")
    a("")
    a("    ctx: *Ctx,")
    if has_state:
        a(f"    state: {S},")
    a("")
    if key == "noise":
        a(f"    pub fn init(ctx: *Ctx, seed: u64) {zname}Host {{")
        a(f"        var s = std.mem.zeroes({S});")
        a("        check(zh_noise_state_init(&s, seed));")
        a("        return .{ .ctx = ctx, .state = s };")
    elif key == "decimator":
        a(f"    pub fn init(ctx: *Ctx) {zname}Host {{")
        a(f"        var s = std.mem.zeroes({S});")
        a("        check(zh_decimator_state_init(&s));")
        a("        return .{ .ctx = ctx, .state = s };")
    elif has_state:
        a(f"    pub fn init(ctx: *Ctx) {zname}Host {{")
        a(f"        return .{{ .ctx = ctx, .state = std.mem.zeroes({S}) }};")
    else:
        a(f"    pub fn init(ctx: *Ctx) {zname}Host {{")
        a("        return .{ .ctx = ctx };")
    a("    }")
    a(f"    pub fn paint(self: *{zname}Host, span: Span, outputs: [num_outputs][]f32, temps: [num_temps][]f32, note_id_changed: bool, params: Params) void {{")
    a("        _ = temps;")
    a("        const outs = [_]?[*]f32{outputs[0].ptr};")
    inits = ", ".join(f".{zig_field(f)} = {expr}" for f, _, expr in fields)
    a(f"        const p = {P}{{ {inits} }};")
    st = "&self.state" if has_state else "null"
    a(f"        check(zh_{key}_paint_host(self.ctx, {st}, @intCast(span.start), @intCast(span.end), &outs, null, @intFromBool(note_id_changed), &p));")
    a("    }")
    a("};")
    a("")
    return "\n".join(L)


def main():
    text = emit(parse_header())
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            sys.stderr.write("bindings/zang_hip.zig is stale: rerun tools/gen_zig_binding.py\n")
            return 1
        return 0
    open(OUT, "w").write(text)
    print("wrote", os.path.relpath(OUT, ROOT), len(text.splitlines()), "lines")
    return 0


if __name__ == "__main__":
    sys.exit(main())
