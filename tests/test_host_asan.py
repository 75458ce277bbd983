"""CPU: the host side of libzang_hip.so under AddressSanitizer + UBSan (SURVEY.md 5 asks for the host restatement
of the scheduler under sanitizers, the reference's own guards being the asserts at src/zang/trigger.zig:161 and src/zang/notes.zig:177).
tools/host_asan.sh builds every csrc/*.hip host-only against a HIP runtime that runs nothing (tools/host_asan/hip_stub.cpp) and runs
 - random sequences of begin_capture / paint / end / launch / destroy through the C ABI (tools/host_asan/harness.cpp): held-back batches,
   flips, the pipelined recording, modules destroyed before their graphs, the context destroyed before its graphs;
 - tests/test_scheduler.py (the reference's 8 scheduler cases as data, the 33-impulse overflow, out-of-order events) and tests/test_abi.py
   against the sanitized library.
Zero reports, and nothing left allocated on the fake device."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.timeout(900)
def test_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with the sanitizer runtimes")
    env = dict(os.environ, HOST_ASAN_DIR=str(tmp_path / "build"))
    env.pop("ZH_FORMS", None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_asan.sh"), "80", "11"], capture_output=True, text=True, env=env, timeout=850)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-4000:]
    m = re.search(r"(\d+) contexts, (\d+) captures, (\d+) paints \((\d+) held back -> (\d+) launches\), (\d+) replays.*left over: 0 device blocks, 0 graphs, 0 streams", out)
    assert m, out[-2000:]
    contexts, captures, paints, held, held_launches, replays = map(int, m.groups())
    assert contexts == 80 and captures > 100 and paints > 5000 and replays > 100
    assert held > 500 and held_launches < held                       # batches really formed (several paints per launch)
    assert re.search(r"\b20 passed\b", out), out[-1500:]               # test_scheduler.py + test_abi.py against the sanitized build
