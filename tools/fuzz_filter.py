"""GPU box: tests/test_gpu_fuzz.py::test_fuzz_filter_and_echoes alone over many seeds.  usage: fuzz_filter.py N [first]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tests.test_gpu_fuzz as fz
import zang_amd
from oracle import pyoracle
ctx = zang_amd.Context(0)
n = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(first, first + n):
    try:
        fz.test_fuzz_filter_and_echoes(ctx, pyoracle, seed)
    except AssertionError as e:
        bad += 1; print("FAIL", seed, str(e)[:300])
print("filter + echoes seeds", n, "from", first, "failures", bad)
