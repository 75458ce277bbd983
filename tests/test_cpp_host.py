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
COMM_SRC = os.path.join(ROOT, "tests", "cpp", "comm_host.c")
COMM_EXE = os.path.join(ROOT, "tests", "cpp", "comm_host")


def _build_c(src, exe):
    import zang_amd  # noqa: F401
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "zang_amd"), "-lzang_hip", "-Wl,-rpath," + os.path.join(ROOT, "zang_amd"),
                           "-L" + rocm + "/lib", "-Wl,-rpath," + rocm + "/lib", "-o", exe])


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
    _build_c(COMM_SRC, COMM_EXE)
    assert os.path.exists(EXE) and os.path.exists(SCRIPT_EXE) and os.path.exists(COMM_EXE)


@pytest.mark.gpu
def test_cpp_host_parity_program():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("PASS") and r.stdout.count("bit-exact") == 12      # 8 paints + batch L/R + RCCL L/R
    assert "zh_nice_paint with ZH_PAINT_TOLERANT: worst sample" in r.stdout             # the opt-in flag from a compiled host


@pytest.mark.gpu
def test_cpp_script_host_without_python():
    """script text -> zh_zscript_compile -> HIP -> zh_script_load -> paint, all from a compiled host."""
    _build(SCRIPT_SRC, SCRIPT_EXE)
    r = subprocess.run([SCRIPT_EXE, os.path.join(ROOT, "tests", "golden", "script_modules.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout + r.stderr
    assert "compiled `Pluck`: 5 state words/voice, 3 params" in r.stdout


@pytest.mark.gpu
def test_c_comm_host_one_rank_without_python():
    """zh_comm_* from a plain C host: RCCL loaded by the library (the ROCm installation's copy here, not torch's), one
    rank per process, id handed over a pipe.  One GPU per box: one rank; `tests/cpp/comm_host 8` on a full node."""
    _build_c(COMM_SRC, COMM_EXE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([COMM_EXE, "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout + r.stderr
    assert "world 1" in r.stdout


@pytest.mark.gpu
def test_c_comm_host_two_processes_rendezvous_on_one_device():
    """Two rank PROCESSES on the one GPU of the box: rank 0's id travels over the pipe, both enter ncclCommInitRank and RCCL's
    bootstrap brings them together -- far enough to find that they name the same device, which RCCL refuses ("invalid usage":
    a communicator needs one GPU per rank).  Both ranks get that answer through the C ABI's error path (ZH_ERR_RCCL_BASE - 5,
    zh_comm_last_error) and the host exits cleanly: no hang, no crash.  The cross-process half of the N > 1 path, as far as
    one GPU can take it."""
    _build_c(COMM_SRC, COMM_EXE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", COMM_HOST_ONE_DEVICE="1")
    r = subprocess.run([COMM_EXE, "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 1 and r.stdout.strip().endswith("FAIL"), r.stdout + r.stderr
    for rank in (0, 1):
        assert f"rank {rank}: zh_comm_create" in r.stderr and "-> -105" in r.stderr, r.stderr
    assert r.stderr.count("ncclCommInitRank: invalid usage") == 2, r.stderr
