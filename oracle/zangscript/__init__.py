"""TEST INFRASTRUCTURE (lives beside the oracle; the product never imports it).  A Python restatement of the
reference's zangscript front-end (src/zangscript/{tokenize,parse,codegen,codegen_zig}.zig), kept as the independent
second implementation the library's C++ front-end (zang_amd/csrc/zscript_front.hip, zscript_emit.hip) is held against
(differential fuzz, tests/test_zangscript_native.py) and as the source of the instruction lists the oracle-side
interpreter (oracle/zs_interp.py) executes.

    script = zangscript.compile(text)                  # tokenize -> parse -> codegen (compile.zig:40-64)
    zig_text = zangscript.generate_zig(script)         # the reference's backend, pinned by its golden test
    hip_text = zangscript.generate_hip(script)         # fused lane-per-voice kernels for libzang_hip's loader
"""
from .builtins import DEFAULT_PACKAGES, modules_builtin_package, zang_builtin_package
from .codegen import CodeGen, CompiledScript
from .errors import ScriptError, Source
from .parse import parse


def compile(contents, filename="script.txt", packages=DEFAULT_PACKAGES):
    source = Source(filename, contents)
    return CodeGen(source, packages, parse(source, packages)).run()


def generate_zig(script):
    from .emit_zig import generate_zig as g
    return g(script)


def generate_hip(script, **kw):
    from .emit_hip import generate_hip as g
    return g(script, **kw)
