#!/usr/bin/env python3
"""GPU box: the span-table parity tests (tests/test_gpu_spans.py) over further random tables.  usage: fuzz_spans.py N"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_spans as sp
import zang_amd
from oracle import pyoracle
ctx = zang_amd.Context(0)
orig = sp._random_tables
n = int(sys.argv[1]); bad = 0
for off in range(1, n + 1):
    sp._random_tables = lambda V, nbuf, seed, off=off: orig(V, nbuf, seed + 1000 * off)
    for V in (3, 20, 64, 65, 130):
        try:
            sp.test_nice_paint_spans(ctx, pyoracle, V)
            sp.test_pmosc_paint_spans(ctx, pyoracle, V, bool(off & 1))
        except AssertionError as e:
            bad += 1; print("FAIL", off, V, str(e)[:300])
print("tables", n, "x 5 voice counts x 2 instruments: failures", bad)
