"""GPU box: the randomised parity cases of tests/test_gpu_fuzz.py over any number of further seeds.  usage: fuzz_many.py N"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytest
import tests.test_gpu_fuzz as fz
from tests import conftest
import zang_amd
from oracle import pyoracle
ctx = zang_amd.Context(0)
n = int(sys.argv[1]); bad = 0
for seed in range(5, 5 + n):
    for fn in (fz.test_fuzz_pulseosc, fz.test_fuzz_nice, fz.test_fuzz_pmosc, fz.test_fuzz_noise_filter, fz.test_fuzz_noise, fz.test_fuzz_sineosc, fz.test_fuzz_sampler,
               fz.test_fuzz_envelope, fz.test_fuzz_decimator_portamento, fz.test_fuzz_osc_control_images, fz.test_fuzz_filter_and_echoes):
        try:
            fn(ctx, pyoracle, seed)
        except AssertionError as e:
            bad += 1; print("FAIL", fn.__name__, seed, str(e)[:300])
print("seeds", n, "failures", bad)
