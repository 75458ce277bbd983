import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import zang_amd
from zang_amd import modules as mod, zang
from oracle import pyoracle as oracle
from tests import util
F=1024
ctx = zang_amd.default_context(); L = oracle.lib()
for V, D, fbv in ((256, 600, 0.9), (256, 2000, 0.99), (256, 1024, 0.5)):
    rngf = np.random.default_rng(D)
    fb = np.full(V, fbv, np.float32); cutoff = rngf.uniform(0.05, 1.0, V).astype(np.float32)
    rings = np.zeros((V, D), np.float32); ds=[]; fls=[]
    for q in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[q]), D); ds.append(d)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
    m = mod.FilteredEchoes(V, D, ctx); gfb, gc = util.dev(fb), util.dev(cutoff)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for k in range(200):
        x = util.rng_buffers(900 + k, V, F) if k < 30 else np.zeros((V, F), np.float32)
        ref = np.zeros((V, F), np.float32)
        for q in range(V):
            L.zo_filtered_echoes_paint(C.byref(ds[q]), C.byref(fls[q]), 0, F, oracle.fptr(ref[q]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(x[q]), float(fb[q]), float(cutoff[q]))
        out = ctx.image(F, V)
        m.paint(zang.Span(0, F), [out], None, False, m.Params(util.to_image(x), gfb, gc), zero_first=True, tolerant=True); ctx.sync()
        rl = np.array([t.l for t in fls], np.float32); rb = np.array([t.b for t in fls], np.float32)
        ratio, _, _ = util.peak_relative_error(util.from_image(out), ref, scale_extra=np.maximum(np.abs(rl), np.abs(rb)))
        if k % 10 == 9 or k < 3: print(D, fbv, k + 1, "%.2e" % ratio.max(), "peak %.2e" % np.abs(ref).max())
