"""GPU box: tests/script_fuzz.py over any number of further seeds (odd seeds: 256-frame buffers as 3-7 frame ranges).  usage: fuzz_scripts.py N [first_seed [tolerant | roles | roles_tolerant]]
(tolerant: every paint with ZH_PAINT_TOLERANT, checked to 1e-5 of max(the voice's peak, 1) instead of bits;
 roles: every paint forced through the role-wave form, zs_paint_pc_<name>, 96- and 256-frame buffers; roles_tolerant: both)"""
import os, sys
os.environ["ZH_ENV_LIVE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zang_amd
from tests import script_fuzz
ctx = zang_amd.default_context()
n = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tol = len(sys.argv) > 3 and sys.argv[3] in ("tolerant", "roles_tolerant")
roles = len(sys.argv) > 3 and sys.argv[3] in ("roles", "roles_tolerant")
worst = [0.0]
bad = 0
for seed in range(first, first + n):
    try:
        if roles:
            script_fuzz.run_case(ctx, seed, roles=1, F=96 if seed % 2 else 256, tolerant=tol, worst=worst)
        elif seed % 2:
            script_fuzz.run_case(ctx, seed, F=256, ranges=3 + seed % 5, tolerant=tol, worst=worst)
        else:
            script_fuzz.run_case(ctx, seed, tolerant=tol, worst=worst)
    except AssertionError as e:
        bad += 1; print("FAIL", str(e)[:3000])
    except Exception as e:
        bad += 1; print("ERROR seed", seed, type(e).__name__, str(e)[:2000]); print(script_fuzz.generate(seed)[0])
print("seeds", n, "from", first, "failures", bad, ("worst error / max(peak, 1) %.2e" % worst[0]) if tol else "")
