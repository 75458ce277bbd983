import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import zang_amd
from zang_amd import zang, modules as mod
from zang_amd.spans import SpanTable
ctx = zang_amd.default_context()
for V in (17, 64, 4096):
    F = 1024
    m = mod.NiceInstrument(V, 0.5, ctx)
    img = ctx.image(F, V)
    tab = SpanTable([[(0, F, 440.0, True, False)] for _ in range(V)], ctx.device)
    freq = torch.full((V,), 440.0, device=ctx.device)
    span = zang.Span(0, F)
    def t(fn, n=50):
        fn(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(n): fn()
        ctx.sync(); return (time.perf_counter() - t0) / n * 1e6
    a = t(lambda: m.paint(span, [img], [], False, m.Params(48000.0, freq, True), zero_first=True))
    b = t(lambda: m.paint_spans(span, [img], None, 48000.0, tab, zero_first=True))
    print(V, "paint %.1f us  paint_spans %.1f us" % (a, b))
