import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
import zang_amd
from zang_amd import modules as mod, zang
from oracle import pyoracle as oracle
from tests import util
ctx = zang_amd.default_context()
L = oracle.lib()
V, F, SR = 256, 1024, 48000.0
rng = np.random.default_rng(99)
cut = np.array([L.zo_filter_cutoff_from_frequency(float(200.0 + 7800.0 * u), SR) for u in rng.random(V)], np.float32)
res = (0.9 * rng.random(V)).astype(np.float32)
inp = (util.rng_buffers(8, V, F) * np.float32(1e-30)).astype(np.float32)
ref = np.zeros((V, F), np.float32)
sts = []
for v in range(V):
    st = oracle.Filter(); L.zo_filter_init(C.byref(st)); sts.append(st)
m = mod.Filter(V, ctx)
gi = util.to_image(inp); dc, dr = util.dev(cut), util.dev(res)
for rep in range(2):
    for v in range(V):
        ref[v] = 0
        L.zo_filter_paint(C.byref(sts[v]), 0, F, oracle.fptr(ref[v]), oracle.fptr(inp[v]), 1, oracle.constant(cut[v]), oracle.constant(res[v]))
    out = ctx.image(F, V, fill=0.0)
    m.paint(zang.Span(0, F), [out], [], False, m.Params(gi, 1, zang.constant(dc), zang.constant(dr)), zero_first=True, tolerant=True)
    ctx.sync()
    got = util.from_image(out)
    ratio, dis, inside = util.peak_relative_error(got, ref)
    w = ratio.argmax()
    err = np.abs(got[w].astype(np.float64) - ref[w])
    print("paint", rep, "worst", ratio.max(), "voice", w, "peak", np.abs(ref[w]).max(), "cut", cut[w], "res", res[w], "first bad frame", np.argmax(err > 1e-5 * np.abs(ref[w]).max()), "err at chunk starts", [float(err[k]) for k in range(0, 1024, 64)][:6])
    print("   ref[w][60:70]", ref[w][60:70], "got", got[w][60:70])
    st = m.state(); st["l"] = np.array([t.l for t in sts], np.float32); st["b"] = np.array([t.b for t in sts], np.float32); m.set_state(st)
