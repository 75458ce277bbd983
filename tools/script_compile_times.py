#!/usr/bin/env python3
"""hiprtc compile time and code size per script module (tests/golden/script_modules.txt), lane form alone against lane + role-wave
form (ZH_ZSCRIPT_FORM_ROLES).  Needs no GPU.  usage: tools/script_compile_times.py [script.txt]"""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zang_amd import script, zscript_native as native

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "script_modules.txt")
text = open(path).read()
nat = native.NativeScript(text)
_, meta = nat.generate_hip()
print("# %s: hiprtc (--offload-arch=gfx950 -O3) per module; host: %d cores" % (os.path.relpath(path, ROOT), os.cpu_count()))
print("%-22s %12s %12s   %12s %12s   %s" % ("module", "lane: s", "bytes", "+ roles: s", "bytes", "role-wave form"))
for name in sorted(meta):
    if "error" in meta[name]:
        continue
    row = []
    for forms in (0, native.FORM_ROLES):
        src, _ = nat.generate_hip(only=[name], forms=forms)
        t0 = time.perf_counter()
        size = script.compile_hip(src)
        row += [time.perf_counter() - t0, size]
    m = re.search(r"// role-wave form: (.*)", src)
    print("%-22s %12.2f %12d   %12.2f %12d   %s" % (name, row[0], row[1], row[2], row[3], m.group(1) if m else "-"))
