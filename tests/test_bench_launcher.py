"""CPU: bench.py's launcher.  `--gpus N` must really start N ranks (it was a dead flag), relay rank 0's
line, and fail loudly when a rank fails or when the flag disagrees with the launcher's WORLD_SIZE.  On this box there is
no GPU: ZH_BENCH_EMULATE=1 turns the ranks into a dry run (rendezvous + the exchange step on host tensors over gloo;
nothing is painted, `value` is 0 and the line says so)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _gpu_here():
    import torch
    return torch.cuda.is_available()


@pytest.mark.timeout(300)
@pytest.mark.skipif(_gpu_here(), reason="the dry run is the no-GPU form of the emulation")
def test_gpus_flag_spawns_ranks_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5"], env=_env(ZH_BENCH_EMULATE="1"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 48 and line["steps_requested"] == 20 and line["warmup"] == 5   # whole note patterns
    assert line["collective"]["world_size_seen"] == 2 and line["collective"]["sum_correct"] is True
    assert line["value"] == 0.0 and "dry_run" in line            # nothing was painted and the line says so
    assert line["config"]["workload"].startswith("nice_mix")     # N > 1 defaults to config 5


@pytest.mark.timeout(300)
@pytest.mark.skipif(_gpu_here(), reason="the dry run is the no-GPU form of the emulation")
def test_under_torch_distributed_run_dry_run():
    """The driver's own command line for N > 1."""
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=_env(ZH_BENCH_EMULATE="1"), capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["collective"]["world_size_seen"] == 2


@pytest.mark.timeout(300)
@pytest.mark.skipif(_gpu_here(), reason="needs a box without a GPU")
def test_failed_rank_gives_nonzero_exit():
    """Without the emulation switch a rank on a GPU-less box must refuse (there is no CPU path to measure), and the
    parent must pass the failure on instead of printing a line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "0"], env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no HIP device" in r.stderr


@pytest.mark.timeout(120)
def test_gpus_flag_must_agree_with_world_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", ZH_BENCH_EMULATE="1"),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.timeout(120)
@pytest.mark.skipif(_gpu_here(), reason="needs a box without a GPU")
def test_single_rank_without_gpu_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--steps", "4", "--warmup", "0"], env=_env(), capture_output=True, text=True, timeout=100)
    assert r.returncode == 3 and "no HIP device" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
def test_two_emulated_ranks_on_one_gpu_run_the_whole_multi_gpu_path(exchange):
    """GPU box (one GPU): `ZH_BENCH_EMULATE=1 bench.py --gpus 2` = two rank processes on device 0 with the driver's step
    counts -- the N > 1 code path end to end (launcher, rendezvous, the 48-buffer pattern captured as ONE graph, the exchange inside
    the timed region, per-batch and per-buffer exchange timings, the shard without exchange, the scaling factor), gloo
    standing in for RCCL (two ranks on one device cannot form an RCCL communicator: tests/test_cpp_host.py) and, for
    `--exchange p2p`, the HIP-IPC slot exchange between the two processes."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--voices", "8192", "--exchange", exchange,
                        "--repeats", "2"], capture_output=True, text=True, timeout=540, env=_env(ZH_BENCH_EMULATE="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _last_json(r.stdout)
    # `--steps 20` on the note-pattern workload = one whole 48-buffer pattern (the release stage is inside the timed region)
    assert d["n_gpus"] == 2 and d["steps"] == 48 and d["steps_requested"] == 20 and d["config"]["launch"].startswith("hipGraph x48 steps")
    assert "pattern" in d["config"] and d["rehearsal_regions"] >= 3
    assert d["config"]["total_voices"] == 16384 and d["value"] > 0 and d["scaling"] == "weak"
    c = d["collective"]
    assert c["world_size_seen"] == 2 and c["in_timed_region"] and c["per"] == "48-buffer batch" and c["bytes"] == 48 * 2 * 1024 * 4
    assert c["reduce_us_spread"]["samples"] >= 20 and c["reduce_us_spread"]["min"] <= c["reduce_us"] <= c["reduce_us_spread"]["max"]
    if exchange == "rccl":
        assert "gloo" in c["backend"] and "ZH_BENCH_EMULATE" in c["note"]
        assert c["per_buffer_form"]["bytes"] == 2 * 1024 * 4 and c["per_buffer_form"]["reduce_us"] > 0
    else:
        assert c["backend"] == "host barriers + HIP IPC"
    # structure only: a wall-clock ratio of two ranks time-slicing one device is not a parity fact
    import math
    assert d["single_gpu_shard"]["value"] > 0 and math.isfinite(d["scaling_factor"]) and d["scaling_factor"] > 0
    assert d["single_gpu_shard"]["regions"] >= 5 and d["scaling_factor_spread"]["pairs"] >= 5          # medians, not one sample each
    assert d["parity"]["bitexact"] is True and d["parity"]["checked_voices"] == 64
    assert d["build"]["zh_version"].startswith("zang_hip")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_preflight_two_emulated_ranks():
    """`bench.py --gpus 2 --preflight` on the one GPU of the box: comm_host as a child process (ONE rank here: RCCL refuses two
    ranks on one device), then two rank processes with one HIP-IPC slot round trip per peer; one JSON line, exit code 0."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--preflight"], capture_output=True, text=True, timeout=540, env=_env(ZH_BENCH_EMULATE="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "comm_host 1: PASS" in r.stderr
    d = _last_json(r.stdout)["preflight"]
    assert d["ok"] is True and d["world"] == 2 and [x["ok"] for x in d["ranks"]] == [True, True]
    assert "zh_ipc_open" in d["ranks"][1]["ipc"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_preflight_one_rank():
    r = subprocess.run([sys.executable, BENCH, "--preflight"], capture_output=True, text=True, timeout=540, env=_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _last_json(r.stdout)["preflight"]
    assert d["ok"] is True and d["comm_host"]["ok"] is True and d["comm_host"]["ranks"] == 1


@pytest.mark.timeout(300)
@pytest.mark.skipif(_gpu_here(), reason="needs a box without a GPU")
def test_preflight_fails_with_its_own_exit_code_without_a_gpu():
    """No GPU here: comm_host's ranks cannot create a context, the rank's own check finds no device -> one line, exit code 5."""
    r = subprocess.run([sys.executable, BENCH, "--preflight"], capture_output=True, text=True, timeout=280, env=_env())
    assert r.returncode == 5, r.stdout[-2000:] + r.stderr[-2000:]
    d = _last_json(r.stdout)["preflight"]
    assert d["ok"] is False and d["ranks"][0]["ok"] is False and "no HIP device" in d["ranks"][0]["error"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_the_drivers_own_command_line_for_n_ranks_on_a_gpu_box():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2
    --steps 20 --warmup 5` -- the driver's launch form for N > 1 -- with the two ranks emulated on the box's one GPU: rank 0
    prints exactly one JSON line with the whole N > 1 structure, the other rank prints nothing on stdout, exit code 0."""
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--voices", "8192"],
                       env=_env(ZH_BENCH_EMULATE="1"), capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 48 and d["steps_requested"] == 20 and d["warmup"] == 5 and d["value"] > 0
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "collective",
                "single_gpu_shard", "scaling_factor", "parity", "build"):
        assert key in d, key
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["config"]["workload"].startswith("nice_mix")
    assert d["collective"]["world_size_seen"] == 2 and d["parity"]["bitexact"] is True


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_one_rank_takes_the_real_rccl_path_of_the_n_rank_line():
    """ZH_BENCH_ONE_RANK_DIST=1 under the driver's launch form with one process: the N > 1 code path as the scaling run takes it
    -- torch.distributed's nccl (= RCCL) process group with `device_id`, the gloo control group, the library's own RCCL
    communicator (zh_comm_*, librccl opened by libzang_hip.so), its all-reduce right behind the batch's kernels in the timed
    region, the paired regions, the parity check of the shard -- with a world of one.  (Two ranks on one device cannot form an
    RCCL communicator: that half runs over gloo in the tests above.)"""
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "20", "--warmup", "5", "--voices", "8192", "--no-cpu"],
                       env=_env(ZH_BENCH_ONE_RANK_DIST="1"), capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 48 and d["value"] > 0 and d["config"]["workload"].startswith("nice_mix")
    c = d["collective"]
    assert c["world_size_seen"] == 1 and c["reduce_us"] > 0 and "rccl" in (c["backend"] + c.get("kind", "")).lower(), c
    assert "zh_comm" in json.dumps(c) or "library" in json.dumps(c), c          # the library's communicator, not the torch fallback
    assert d["scaling_factor"] > 0 and d["parity"]["bitexact"] is True
