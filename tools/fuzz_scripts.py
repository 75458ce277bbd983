"""GPU box: tests/script_fuzz.py over any number of further seeds (odd seeds: 256-frame buffers as 3-7 frame ranges).  usage: fuzz_scripts.py N [first_seed]"""
import os, sys
os.environ["ZH_ENV_LIVE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zang_amd
from tests import script_fuzz
ctx = zang_amd.default_context()
n = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(first, first + n):
    try:
        if seed % 2:
            script_fuzz.run_case(ctx, seed, F=256, ranges=3 + seed % 5)
        else:
            script_fuzz.run_case(ctx, seed)
    except AssertionError as e:
        bad += 1; print("FAIL", str(e)[:3000])
    except Exception as e:
        bad += 1; print("ERROR seed", seed, type(e).__name__, str(e)[:2000]); print(script_fuzz.generate(seed)[0])
print("seeds", n, "from", first, "failures", bad)
