import os, sys
os.environ["ZH_ENV_LIVE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zang_amd
from tests import script_fuzz
ctx = zang_amd.default_context()
for seed in (12032, 12155):
    for kw in ({}, {"roles": 0}, {"roles": 1}):
        F = 256 if seed % 2 else 96
        k = dict(kw)
        if seed % 2 and not kw: k["ranges"] = 3 + seed % 5
        try:
            script_fuzz.run_case(ctx, seed, F=F, **k)
            print("seed", seed, k, "ok")
        except AssertionError as e:
            print("seed", seed, k, "FAIL", str(e)[:200])
text, name = script_fuzz.generate(12032)
print(text)
