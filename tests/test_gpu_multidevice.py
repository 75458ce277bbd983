"""GPU, two or more devices: the multi-GPU path with more than one rank on more than one device.

These tests ARM THEMSELVES: on a box that shows >= 2 devices they run (`comm_host N`, the RCCL communicator of the library against
a host-side sum, the HIP-IPC slot exchange across devices, a 2-rank config-5 render against the 1-rank render); on the
one-GPU boxes they are collected and skipped.  The same worker (tests/multidev_worker.py) also runs here on ONE device with two
rank processes and gloo standing in for RCCL, so that its own logic is exercised by every GPU run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(case, world, emulate=False, timeout=540):
    port = _free_port()
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if emulate:
            env["MULTIDEV_EMULATE"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.multidev_worker", case], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    text = "\n".join(o + e[-1500:] for o, e in outs)
    assert all(p.returncode == 0 for p in procs), text
    assert f"PASS {case} world {world}" in outs[0][0], text


needs2 = pytest.mark.skipif(_devices() < 2, reason="needs two or more GPUs on the box (the driver's 8-GPU node): armed automatically there")


@needs2
@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", ["2", "all"])
def test_c_comm_host_n_ranks_on_n_devices(n):
    """`tests/cpp/comm_host N`: N rank processes from a plain C host, rank r on GPU r, RCCL through the C ABI, the id over pipes,
    all-reduce and reduce of the bench's block checked on every rank."""
    from tests.test_cpp_host import _build_c, COMM_SRC, COMM_EXE
    _build_c(COMM_SRC, COMM_EXE)
    world = 2 if n == "2" else _devices()
    r = subprocess.run([COMM_EXE, str(world)], capture_output=True, text=True, timeout=540, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), r.stdout + r.stderr
    assert f"world {world}" in r.stdout


@needs2
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 0])
def test_library_communicator_allreduce_against_the_host_sum(world):
    _run_ranks("comm", world or _devices())


@needs2
@pytest.mark.timeout(600)
def test_slot_exchange_across_two_devices():
    _run_ranks("slots", 2)


@needs2
@pytest.mark.timeout(900)
def test_two_rank_config5_render_equals_the_one_rank_render():
    _run_ranks("render", 2, timeout=840)


@needs2
@pytest.mark.timeout(900)
def test_bench_two_ranks_preflight_and_line():
    """bench.py's own N > 1 path with real ranks on real devices: the preflight, then the line (RCCL exchange in the timed region)."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ZH_BENCH_EMULATE")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--preflight"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "20", "--warmup", "5", "--voices", "16384"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["collective"]["world_size_seen"] == 2 and "rccl" in d["collective"]["backend"]
    assert d["parity"]["bitexact"] is True


# ---- the same worker on ONE device (every GPU box runs these): two rank processes, gloo instead of RCCL
@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", ["comm", "slots", "render"])
def test_worker_logic_on_one_device(case):
    _run_ranks(case, 2, emulate=True)
