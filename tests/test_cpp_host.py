"""The C++ host-side API (include/zang_hip.hpp): compiles against the C ABI header here (CPU), and on a GPU
runs tests/cpp/host_parity.cpp -- examples/modules.zig's NiceInstrument written with the C++ mirror of zang's
namespaces, checked bit for bit against the oracle and against the fused kernel."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_parity.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_parity")
SCRIPT_SRC = os.path.join(ROOT, "tests", "cpp", "script_host.cpp")
SCRIPT_EXE = os.path.join(ROOT, "tests", "cpp", "script_host")


def _build(SRC=SRC, EXE=EXE):
    from oracle import pyoracle
    pyoracle.build()
    import zang_amd  # noqa: F401  (fails loudly if libzang_hip.so is missing)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), SRC,
           "-L" + os.path.join(ROOT, "zang_amd"), "-lzang_hip", os.path.join(ROOT, "oracle", "libzang_oracle.so"),
           "-Wl,-rpath," + os.path.join(ROOT, "zang_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-L" + rocm + "/lib", "-Wl,-rpath," + rocm + "/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_cpp_host_api_compiles_and_links():
    _build()
    _build(SCRIPT_SRC, SCRIPT_EXE)
    assert os.path.exists(EXE) and os.path.exists(SCRIPT_EXE)


@pytest.mark.gpu
def test_cpp_host_parity_program():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("PASS") and r.stdout.count("bit-exact") == 8


@pytest.mark.gpu
def test_cpp_script_host_offline_flow(tmp_path):
    """zangc offline -> script.hip -> loaded and painted by a compiled host through the C ABI alone."""
    import sys
    _build(SCRIPT_SRC, SCRIPT_EXE)
    hip = tmp_path / "script.hip"
    r = subprocess.run([sys.executable, "-m", "zang_amd.zangc", os.path.join(ROOT, "tests", "golden", "script_modules.txt"), "-o", str(hip)],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    words = [l for l in r.stderr.splitlines() if "module Pluck:" in l][0].split("module Pluck:")[1].split()[0]
    r = subprocess.run([SCRIPT_EXE, str(hip), words], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout + r.stderr
