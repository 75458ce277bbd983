"""CPU, world_size 2, gloo: the N>1 path's host logic.  Each rank derives its shard's
parameters from GLOBAL voice indices, renders its partial mix (with the oracle standing in
for the GPU kernels -- this test checks the sharding + collective, not the kernels) and the
sum all-reduce must equal the unsharded render."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V, F, SR = 96, 256, 48000.0


def _render_partial(first, n):
    """Oracle NiceInstrument voices [first, first+n) + Noise voices seeded by global index, summed."""
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    from zang_amd import workloads
    L = po.lib()
    freq, color, _, _ = workloads.voice_params(5, first, n)
    mix = np.zeros(F, np.float64)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for v in range(n):
        st = po.NiceInstrument(); L.zo_nice_init(C.byref(st), float(color[v]))
        out = np.zeros(F, np.float32)
        L.zo_nice_paint(C.byref(st), 0, F, po.fptr(out), po.fptr(t0), po.fptr(t1), 1, SR, float(freq[v]), 1)
        nz = po.Noise(); L.zo_noise_init(C.byref(nz), first + v)
        L.zo_noise_paint(C.byref(nz), 0, F, po.fptr(out), po.NOISE_WHITE)
        mix += out
    return mix.astype(np.float32)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from zang_amd import sharding
    lo, hi = sharding.voice_range(V, rank, world)
    part = torch.from_numpy(_render_partial(lo, hi - lo))
    # bench.py exchanges a whole batch of buffers in one collective: a [buffers][frames] block
    block = torch.stack([part, part * 0.5, torch.zeros_like(part)])
    sharding.allreduce_mix(block)
    assert torch.equal(block[1], block[0] * 0.5) and not block[2].any()
    mix = block[0]
    if rank == 0:
        q.put(mix.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_voice_range_partitions():
    from zang_amd import sharding
    for total in (0, 1, 7, 96, 4096, 1048576):
        for world in (1, 2, 3, 8):
            r = [sharding.voice_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))


def test_voice_params_shard_consistently():
    from zang_amd import workloads
    whole = workloads.voice_params(2, 0, 1000)
    parts = [workloads.voice_params(2, lo, hi - lo) for lo, hi in ((0, 333), (333, 700), (700, 1000))]
    for k in range(4):
        assert np.array_equal(whole[k], np.concatenate([p[k] for p in parts]))


@pytest.mark.timeout(300)
def test_sharded_mix_allreduce_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _render_partial(0, V)
    # f32 partial sums in a different association: sqrt(V)*eps bound (mixdown parity)
    assert np.abs(got - ref).max() <= 8 * np.sqrt(V) * np.finfo(np.float32).eps * max(np.abs(ref).max(), 1.0)


class _FakeCommLib:
    """Stands in for libzang_hip.so's zh_comm_* in the CPU test of sharding.Comm's host logic (the id hand-over and the
    all-ranks-or-none set-up): the real entry points need a GPU and are covered by tests/test_gpu_comm.py."""

    def __init__(self, rank, fail):
        self.rank, self.fail, self.created, self.destroyed, self.seen_id = rank, fail, 0, 0, None

    def zh_comm_available(self):
        return 0 if self.fail == "unavailable" and self.rank == 1 else 1

    def zh_comm_last_error(self):
        return b"simulated"

    def zh_comm_unique_id(self, uid):
        if self.fail == "id":
            return -4
        for i in range(128):
            uid[i] = (7 * i + 3) & 255
        return 0

    def zh_comm_create(self, ctx, world, rank, uid, out):
        self.seen_id = bytes(uid)
        if self.fail == "create" and self.rank == 1:
            return -105
        self.created += 1
        out._obj.value = 1234 + rank
        return 0

    def zh_comm_destroy(self, h):
        self.destroyed += 1
        return 0


def _comm_worker(rank, world, port, fail, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from zang_amd import abi, sharding
    lib = _FakeCommLib(rank, fail)
    ctx = type("Ctx", (), {"lib": lib, "handle": None})()
    try:
        comm = sharding.Comm(ctx)
        outcome = ("ok", comm.world, comm.rank, lib.seen_id == bytes((7 * i + 3) & 255 for i in range(128)))
    except abi.ZangHipError as e:
        outcome = ("error", str(e)[:60], lib.created, lib.destroyed)
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fail", [None, "unavailable", "id", "create"])
def test_comm_setup_is_all_ranks_or_none(fail):
    """sharding.Comm over two gloo ranks with the library's zh_comm_* replaced by a fake: rank 0's 128-byte id reaches rank 1
    unchanged; and whatever goes wrong on ONE rank (librccl missing, the id, the communicator) raises on BOTH, so that
    nobody is left waiting in a collective -- the rank that did get a communicator destroys it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, fail, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if fail is None:
        assert got[0] == ("ok", 2, 0, True) and got[1] == ("ok", 2, 1, True)
    else:
        assert got[0][0] == "error" and got[1][0] == "error", got
        if fail == "create":
            assert got[0][2:] == (1, 1) and got[1][2:] == (0, 0)      # rank 0 made one and destroyed it
