#!/usr/bin/env python3
"""GPU box: what does hipGraphLaunch itself cost on the driver's 20-step region?  The same 20 PulseOsc buffers (4,096 voices x 1,024 frames)
as (a) the coalesced graph bench.py replays (2 kernel nodes of 10 buffers) and (b) the same two launches made directly
(zh_pulseosc_paint_batch x 2, eager), each region = record, launch(es), record, synchronize from an idle GPU, 200 regions, medians."""
import ctypes as C
import statistics
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zang_amd
from zang_amd import abi, modules as mod, zang, workloads

V, F, K, SR = 4096, 1024, 20, 48000.0
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
ctx = zang_amd.Context(0)
lib = ctx.lib
freq, color, _, _ = workloads.voice_params(2, 0, V)
m = mod.PulseOsc(V, ctx)
fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
ring = [ctx.image(F, V) for _ in range(32)]
sp = zang.Span(0, F)
P = m.Params(SR, zang.constant(fr), col)
m.paint(sp, [ring[0]], [], False, P, zero_first=True)
ctx.sync()
bufs = [ring[i % 32] for i in range(K)]
g = ctx.capture(lambda: [m.paint(sp, [o], [], False, P, zero_first=True, params_unchanged=True) for o in bufs], coalesce=True)
print("graph:", g.info())
sync = getattr(torch._C, "_cuda_synchronize", None) or torch.cuda.synchronize
e0, e1 = C.c_void_p(), C.c_void_p()
abi.check(lib.zh_event_create(ctx.handle, C.byref(e0)), "ev"); abi.check(lib.zh_event_create(ctx.handle, C.byref(e1)), "ev")

def direct():
    m.paint_batch(sp, bufs[:10], P, zero_first=True, params_unchanged=True)
    m.paint_batch(sp, bufs[10:], P, zero_first=True, params_unchanged=True)

def region(fn):
    sync()
    t0 = time.perf_counter()
    lib.zh_event_record(ctx.handle, e0)
    fn()
    lib.zh_event_record(ctx.handle, e1)
    sync()
    wall = time.perf_counter() - t0
    ms = C.c_float()
    lib.zh_event_elapsed_ms(e0, e1, C.byref(ms))
    return wall * 1e6, ms.value * 1e3

for name, fn in (("graph replay (2 kernel nodes)", g.launch), ("two direct batch launches", direct), ("graph replay (2 kernel nodes)", g.launch), ("two direct batch launches", direct)):
    for _ in range(50):
        region(fn)
    r = [region(fn) for _ in range(200)]
    w = statistics.median(x[0] for x in r); e = statistics.median(x[1] for x in r)
    print("%-32s wall %.1f us  events %.1f us  -> %.3e voice-samples/s" % (name, w, e, V * F * K / (w * 1e-6)))
