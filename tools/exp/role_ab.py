#!/usr/bin/env python3
"""GPU box: every module of a script (default tests/golden/script_modules.txt) that has a role-wave kernel, lane form against role-wave
form (dispatch row script_pc = 0 / 1) at V voices, 20 paints per graph; prints the emitter's own estimate (`hint`) beside the result, so
that the default selection can be checked against measurement.  usage: role_ab.py [voices [script.txt]]"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ZH_ENV_LIVE"] = "1"
import numpy as np
import torch
import zang_amd
from zang_amd import script as zscript, zang, zscript_native as native

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "script_modules.txt")
F, SR, K = 1024, 48000.0, 20
torch.cuda.set_stream(torch.cuda.Stream())
ctx = zang_amd.Context(0)
dev = ctx.device
rng = np.random.default_rng(1)
text = open(path).read()
prog = zscript.ScriptProgram(text, ctx, forms=native.FORM_ROLES)
span = zang.Span(0, F)
out = [ctx.image(F, V) for _ in range(8)]
img = ctx.image(F, V); img.uniform_(-1.0, 1.0)
fimg = ctx.image(F, V); fimg.uniform_(100.0, 2000.0)


def value(name, kind, enum):
    if kind == "constant":
        lo, hi = (100.0, 2000.0) if "freq" in name or name == "pitch" else (0.05, 0.9)
        return torch.from_numpy(rng.uniform(lo, hi, V).astype(np.float32)).to(dev) if name != "sample_rate" else SR
    if kind == "boolean":
        return True
    if kind == "buffer":
        return fimg if "freq" in name else img
    if kind == "constant_or_buffer":
        return torch.from_numpy(rng.uniform(100.0, 2000.0, V).astype(np.float32)).to(dev) if "freq" in name else 0.4
    if kind == "curve":                                     # a device array of (value, t) pairs: nothing is uploaded while a capture records
        return torch.tensor([0.0, 0.0, 1.0, 0.01, 0.3, 0.2, 0.0, 1.0], dtype=torch.float32, device=dev)
    labels = native.ENUM_LABELS[enum]
    return (labels[1] if len(labels) > 1 else labels[0], 0.05)


print("# %s, %d voices x %d frames, %d paints per graph: us per paint" % (os.path.relpath(path, ROOT), V, F, K))
print("%-22s %10s %10s %8s  %s" % ("module", "lane", "role-wave", "hint", "role-wave form"))
for name in sorted(prog.meta):
    m_ = prog.meta[name]
    if "error" in m_ or ("zs_paint_pc_" + name + "(") not in prog.hip_source:
        continue
    hint = int(re.search(r"zs_pc_info_%s\[4\] = \{\d+u, \d+u, \d+u, (\d)u\}" % name, prog.hip_source).group(1))
    hint = (hint & 1) if V <= 32768 else (hint >> 1) & 1            # (bit 1: still worth it above half of script_pc_maxv)
    desc = re.search(r"// role-wave form: ([^\n]*)\nextern \"C\" __device__ const uint32_t zs_pc_info_%s\[" % name, prog.hip_source).group(1)
    res = []
    for pc in (0, 1):
        os.environ["ZH_FORMS"] = "script_pc=%d" % pc
        mod = prog.module(name, V, 0)
        params = {n: value(n, k, e) for n, k, e in mod.params}
        paint = lambda o: mod.paint(span, [o], None, False, params, zero_first=True)
        mod.paint(span, [out[0]], None, True, params, zero_first=True)
        for i in range(3):
            paint(out[i])
        ctx.sync()
        g = ctx.capture(lambda: [paint(out[i % 8]) for i in range(K)])
        g.launch(); ctx.sync()
        t0 = time.perf_counter(); g.launch(); ctx.sync(); dt = time.perf_counter() - t0
        g.close(); mod.close()
        res.append(dt * 1e6 / K)
    print("%-22s %10.1f %10.1f %8d  %s%s" % (name, res[0], res[1], hint, desc[:70], "   <-- the estimate is wrong" if (res[1] < res[0] * 0.95) != bool(hint) and abs(res[1] - res[0]) > 0.05 * res[0] else ""))
