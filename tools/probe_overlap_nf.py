#!/usr/bin/env python3
"""GPU box: would pass A of buffer n + 1 beside pass B of buffer n pay for the tolerant Noise -> Filter voice (config 3)?  Upper bound without
writing it: TWO independent 4,096-voice modules painted (a) one after the other on one stream and (b) on two streams at once (graphs of
50 paints each, replayed together).  If (b) is not much faster than (a), the two passes would not overlap either."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zang_amd
from zang_amd import modules as mod, zang, workloads

V, F, K, SR = 4096, 1024, 50, 48000.0
tol = len(sys.argv) < 2 or sys.argv[1] != "exact"
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs, mods, rings, graphs = [], [], [], []
_, _, u2, u3 = workloads.voice_params(3, 0, V)
for i, st in enumerate(streams):
    with torch.cuda.stream(st):
        c = zang_amd.Context(0)
        m = mod.NoiseFilter(V, c, first_seed=i * V)
        cutoff = mod.Filter.cutoffFromFrequency(torch.from_numpy((200.0 + 7800.0 * u2)).cuda(), SR, c)
        res = torch.from_numpy((0.9 * u3)).cuda()
        P = m.Params(0, 1, cutoff, res)
        ring = [c.image(F, V) for _ in range(16)]
        sp = zang.Span(0, F)
        for k in range(4):
            m.paint(sp, [ring[k]], None, False, P, zero_first=True, tolerant=tol)
        c.sync()
        g = c.capture(lambda: [m.paint(sp, [ring[k % 16]], None, False, P, zero_first=True, tolerant=tol) for k in range(K)])
        g.launch(); c.sync()
        ctxs.append(c); mods.append(m); rings.append(ring); graphs.append(g)
print("kernels:", ctxs[0].last_form())

def run(which):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in which:
        with torch.cuda.stream(streams[i]):
            graphs[i].launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6

for name, which in (("one module, one stream", [0]), ("two modules, one after the other", [0, 0, 1, 1][0:0] or None), ("two modules on two streams at once", [0, 1])):
    if which is None:
        ts = []
        for _ in range(8):
            a = run([0]); b = run([1]); ts.append(a + b)
        print("%-40s %.1f us per paint (two modules: %.1f us per pair)" % (name, min(ts) / (2 * K), min(ts) / K))
        continue
    ts = [run(which) for _ in range(8)]
    n = len(which) * K
    print("%-40s %.1f us per paint%s" % (name, min(ts) / n, "" if len(which) == 1 else "  (%.1f us per pair of paints)" % (min(ts) / K)))
