"""zangc: compile a zangscript file (the reference's tools/zangc.zig:1-27 compiles to Zig source that
is built into the host program; this one also writes the fused HIP kernels for zh_script_load).

    python tools/zangc.py [options] -o <dest> <file>

A developer tool, not part of the product package: the kernels come from the library's C++ front-end
(zang_amd.zscript_native); the --dump-* listings come from the oracle-side Python front-end (oracle/zangscript).

      --backend hip|zig       what to write to <dest> (default: hip)
      --dump-codegen <file>   the instruction list of every script module
      --dump-builtins <file>  the builtin modules and enums the script can use
      --check                 stop after the front-end (tools/zangc.zig:6 TODO)
      --color always|never|auto   accepted for compatibility; errors are plain text
"""
import argparse
import sys

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zangscript as zs  # noqa: E402


def dump_codegen(script):
    lines = []

    def res(r):
        if r.kind in ("temp_buffer", "temp_float"):
            return "%s%d%s" % ("temp" if r.kind == "temp_buffer" else "temp_float", r.index, "(weak)" if r.weak else "")
        if r.kind == "literal_number":
            return r.value.verbatim
        if r.kind == "literal_enum_value":
            return ".%s%s" % (r.value, "(%s)" % res(r.payload) if r.payload is not None else "")
        if r.kind in ("self_param", "track_param"):
            return "%s[%d]" % (r.kind, r.index)
        return "%s(%s)" % (r.kind, r.value if r.value is not None else r.index)

    def dest(d):
        return d if isinstance(d, int) else "%s%d" % ("temp" if d.kind == "temp" else "output", d.index)

    def walk(instructions, indent):
        for i in instructions:
            parts = [i.kind]
            if i.out is not None:
                parts.append("out=%s" % dest(i.out))
            if i.op:
                parts.append("op=%s" % i.op)
            for name in ("a", "b", "src", "speed"):
                v = getattr(i, name)
                if v is not None:
                    parts.append("%s=%s" % (name, res(v)))
            if i.kind == "cob_to_buffer":
                parts.append("param=%d" % i.in_self_param)
            if i.kind == "call":
                parts.append("field=%d temps=%s args=[%s]" % (i.field_index, i.temps, ", ".join(res(a) for a in i.args)))
            lines.append("    " * indent + " ".join(parts))
            if i.instructions:
                walk(i.instructions, indent + 1)

    for name, mi in script.exported_modules:
        r = script.module_results[mi]
        lines.append("module %s: num_temps=%d num_temp_floats=%d fields=%s delays=%s" % (
            name, r.num_temps, r.num_temp_floats,
            [script.modules[f].builtin_name or "_module%d" % f for f in r.fields], r.delays))
        walk(r.instructions, 1)
    return "\n".join(lines) + "\n"


def dump_builtins(packages):
    lines = []
    for pkg in packages:
        for e in pkg.enums:
            lines.append("enum %s: %s" % (e.name, ", ".join(v.label + ("(f32)" if v.payload == "f32" else "") for v in e.values)))
        for b in pkg.builtins:
            lines.append("module %s(%s)" % (b.name, ", ".join("%s: %s" % (p.name, p.param_type.enum.name if p.param_type.enum else p.param_type.kind)
                                                              for p in b.params)))
    return "\n".join(lines) + "\n"


def main(argv=None):
    ap = argparse.ArgumentParser(prog="zangc", description="Compile zangscript to fused HIP kernels (or the reference's Zig).")
    ap.add_argument("file")
    ap.add_argument("-o", "--output")
    ap.add_argument("--backend", choices=("hip", "zig"), default="hip")
    ap.add_argument("--dump-codegen")
    ap.add_argument("--dump-builtins")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--color", default="auto")
    a = ap.parse_args(argv)
    from zang_amd import zscript_native as native
    text = open(a.file).read()
    try:                                                        # the C++ compiler in libzang_hip.so
        compiled = native.NativeScript(text, a.file)
    except native.NativeScriptError as e:
        sys.stderr.write(str(e) + "\n\n")
        return 1
    if a.dump_builtins:
        open(a.dump_builtins, "w").write(dump_builtins(zs.DEFAULT_PACKAGES))
    if a.dump_codegen:                                          # the instruction lists, from the Python front-end
        open(a.dump_codegen, "w").write(dump_codegen(zs.compile(text, a.file)))
    if a.check:
        return 0
    if not a.output:
        ap.error("-o <dest> is required unless --check is given")
    if a.backend == "zig":
        open(a.output, "w").write(compiled.generate_zig())
        return 0
    hip, meta = compiled.generate_hip()
    open(a.output, "w").write(hip)
    for name, m in meta.items():
        if "error" in m:
            sys.stderr.write("%s: module %s: %s\n" % (a.file, name, m["error"]))
        else:
            sys.stderr.write("%s: module %s: %d state words/voice, params %s\n" % (
                a.file, name, m["state_words"], ", ".join("%s:%s" % (p[0], p[2] or p[1]) for p in m["params"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
