"""The micro-benchmarks whose output DESIGN.md quotes this round still build for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["select_hazard", "data_power"])
def test_ubench_compiles(name, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / name
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-o", str(out), os.path.join(ROOT, "tools", "ubench", name + ".hip")],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()
