#!/usr/bin/env python3
"""GPU box: the pipelined recording of tolerant Noise -> Filter paints (ZH_CAPTURE_COALESCE, k_nf_tp_ba) against the same calls made one by
one on a twin module: random voice counts, random sequences of spans (whole buffers, sub-spans, spans too short for the two-pass form),
ZERO_FIRST and `+=`, exact paints and other modules' calls in between, two replays.  Bits of every image and of the states must agree.
usage: fuzz_nf_pipeline.py N [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zang_amd
from zang_amd import modules as mod, zang

F = 1024
n = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
bad = 0; held_total = 0; fused_total = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    V = int(rng.choice([64, 200, 256, 1000, 4096, 4100, 8192, 16384]))
    c = zang_amd.Context(0)
    try:
        cutoff = torch.from_numpy(rng.uniform(0.01, 0.9, V).astype(np.float32)).cuda(); res = torch.from_numpy(rng.uniform(0, 0.95, V).astype(np.float32)).cuda()
        ma, mb = mod.NoiseFilter(V, c, first_seed=seed), mod.NoiseFilter(V, c, first_seed=seed)
        g8 = mod.Gate(V, c)
        K = int(rng.integers(2, 9))
        ra = [c.image(F, V, fill=0.5) for _ in range(K)]; rb = [c.image(F, V, fill=0.5) for _ in range(K)]
        calls = []
        for k in range(K):
            kind = rng.random()
            if kind < 0.12: s, e = 0, int(rng.integers(1, 128))                       # too short for the two-pass form: exact
            elif kind < 0.4:
                s = int(rng.integers(0, 500)); e = int(rng.integers(s + 128, F + 1))
            else: s, e = 0, F
            calls.append((s, e, bool(rng.random() < 0.7), bool(rng.random() < 0.85), int(rng.integers(0, 6)) if rng.random() < 0.3 else 1, bool(rng.random() < 0.15)))
        def seq(m, ring):
            for k, (s, e, zf, tol, ftype, gate_after) in enumerate(calls):
                m.paint(zang.Span(s, e), [ring[k]], None, False, m.Params(0, ftype, cutoff, res), zero_first=zf, tolerant=tol)
                if gate_after:
                    g8.paint(zang.Span(0, 64), [ring[k]], [], False, g8.Params(True))
        seq(ma, ra); seq(mb, rb); c.sync()
        g = c.capture(lambda: seq(mb, rb), coalesce=True)
        nodes, held, launches = g.info(); held_total += held; fused_total += max(0, 2 * held - launches)   # (a chain of k paints: k + 1 launches)
        for rep in range(2):
            for im in ra + rb: im.fill_(0.5)
            seq(ma, ra); g.launch(); c.sync()
            for k in range(K):
                if not torch.equal(ra[k].view(torch.int32), rb[k].view(torch.int32)):
                    raise AssertionError(f"image {k} of replay {rep}: calls {calls}")
            sa, sb = ma.state(), mb.state()
            if sa["noise"]["r"].tobytes() != sb["noise"]["r"].tobytes() or sa["flt"].tobytes() != sb["flt"].tobytes():
                raise AssertionError(f"states after replay {rep}: calls {calls}")
        g.close()
    except AssertionError as ex:
        bad += 1; print("FAIL seed", seed, "V", V, str(ex)[:400])
    finally:
        c.close()
print("seeds", n, "from", first, "failures", bad, "| paints recorded pipelined:", held_total, "of them fused with their neighbour:", fused_total)
