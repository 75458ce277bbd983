#!/usr/bin/env python3
"""GPU box: where the time of a role-wave kernel goes -- FilteredSawtooth (tests/golden/script_modules.txt) at V voices with one
role's arithmetic replaced by a constant at a time (the generated text, edited).  usage: role_probe.py [voices]"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import zang_amd
from zang_amd import script as zscript, zang, workloads

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F, SR, K = 1024, 48000.0, 20
os.environ["ZH_FORMS"] = "script_pc=1"
torch.cuda.set_stream(torch.cuda.Stream())
ctx = zang_amd.Context(0)
dev = ctx.device
freq_h, _, _, _ = workloads.voice_params(5, 0, V)
freq = torch.from_numpy(freq_h).to(dev)
span = zang.Span(0, F)
out = [ctx.image(F, V) for _ in range(32)]
on_dev = torch.ones(V, dtype=torch.uint8, device=dev)
text = open(os.path.join(ROOT, "tests", "golden", "script_modules.txt")).read()

def pc_only(fn):
    def patch(src):
        i = src.index("// role-wave form")
        return src[:i] + fn(src[i:])
    return patch

variants = {
    "as generated": lambda s: s,
    "oscillator -> constant": lambda s: s.replace("const bool cp3 = m1.frame_const(cv2);", "const bool cp3 = true; cv2 = 0.25f;"),
    "envelope -> constant": lambda s: s.replace("const bool cp6 = m4.frame_sq<ZS_Q>(cv5, zs_walk);", "const bool cp6 = true; cv5 = 0.5f;"),
    "filter core -> copy": lambda s: re.sub(r"const SvfOut zs_s = m7\.core<false, false>\((\w+), 0\.0f, 0\.0f\);", r"const SvfOut zs_s = SvfOut{\1, \1, \1};", s),
    "all three": None,
}
def all3(s):
    for k in ("oscillator -> constant", "envelope -> constant", "filter core -> copy"):
        s = variants[k](s)
    return s
variants["all three"] = all3
if os.environ.get("ZS_PC_TRACE") == "1":          # one traced paint: per role wave, the cycles between barriers
    prog = zscript.ScriptProgram(text, ctx, only=["FilteredSawtooth"], hip_patch=lambda s: "#define ZS_PC_TRACE 1\n" + s)
    m = prog.module("FilteredSawtooth", V, 0)
    for i in range(2):
        m.paint(span, [out[0]], None, i == 0, {"sample_rate": SR, "freq": freq, "note_on": on_dev, "cutoff": 0.07}, zero_first=True)
        ctx.sync()
        print("--")
    prog.close()
    sys.exit(0)
print("# FilteredSawtooth, role-wave form, %d voices x %d frames, %d paints per graph" % (V, F, K))
for name, fn in variants.items():
    prog = zscript.ScriptProgram(text, ctx, only=["FilteredSawtooth"], hip_patch=pc_only(fn))
    m = prog.module("FilteredSawtooth", V, 0)
    paint = lambda o: m.paint(span, [o], None, False, {"sample_rate": SR, "freq": freq, "note_on": on_dev, "cutoff": 0.07}, zero_first=True)
    m.paint(span, [out[0]], None, True, {"sample_rate": SR, "freq": freq, "note_on": on_dev, "cutoff": 0.07}, zero_first=True)
    for i in range(3):
        paint(out[i])
    ctx.sync()
    g = ctx.capture(lambda: [paint(out[i % 32]) for i in range(K)])
    g.launch(); ctx.sync()
    t0 = time.perf_counter(); g.launch(); ctx.sync(); dt = time.perf_counter() - t0
    g.close()
    print("%-28s %8.1f us per paint   (%s)" % (name, dt * 1e6 / K, ",".join(ctx.last_form())))
    prog.close()
