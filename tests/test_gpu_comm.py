"""GPU: the collective form of the multi-GPU exchange step really runs RCCL.

One GPU per box, so the communicators here have ONE rank: RCCL is loaded, bootstrapped, and its reduce kernels are
launched on the context's stream; the sum over one rank is the identity, which is what is checked -- on the block
bench.py exchanges per batch ([48 buffers][2 channels][1024 frames]) filled by the config-5 mixdown kernel itself.
The N > 1 logic (id hand-over, rank-ordered shards) is covered on CPU by tests/test_sharding_gloo.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SR, F = 48000.0, 1024


def _mix_block(ctx, V=4096, buffers=48):
    """[buffers][2][F] partial mixes painted by zh_nice_paint_mix_stereo (the config-5 step of bench.py)."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    freq, color, u2, _ = workloads.voice_params(5, 0, V)
    dev = ctx.device
    m = mod.NiceInstrument(V, torch.from_numpy(color).to(dev), ctx)
    gl = torch.from_numpy((u2 * 0.5 + 0.25).astype(np.float32)).to(dev)
    gr = torch.from_numpy((0.75 - u2 * 0.5).astype(np.float32)).to(dev)
    fr = torch.from_numpy(freq).to(dev)
    mixes = torch.zeros((buffers, 2, F), dtype=torch.float32, device=dev)
    for b in range(buffers):
        m.paint_mix_stereo(zang.Span(0, F), mixes[b, 0], mixes[b, 1], gl, gr, b == 0, m.Params(SR, fr, b < buffers // 2), zero_first=True)
    ctx.sync()
    return mixes


def test_library_finds_rccl(ctx):
    lib = ctx.lib
    assert lib.zh_comm_available() == 1, lib.zh_comm_last_error().decode()
    assert lib.zh_comm_version() >= 20000
    assert b"rccl" in lib.zh_comm_library()


def test_one_rank_communicator_allreduce_and_reduce(ctx):
    import torch
    from zang_amd import sharding
    mixes = _mix_block(ctx)
    assert float(mixes.abs().max()) > 0
    want = mixes.clone()
    comm = sharding.Comm(ctx, world=1, rank=0)
    assert ctx.lib.zh_comm_world(comm.handle) == 1 and ctx.lib.zh_comm_rank(comm.handle) == 0
    comm.allreduce_mix(mixes)                                  # the per-batch exchange: 384 KiB in one collective
    ctx.sync()
    assert torch.equal(mixes.view(torch.int32), want.view(torch.int32))
    for b in range(48):                                        # the per-buffer exchange: 8 KiB each (write_wav.zig:58-93)
        comm.allreduce_mix(mixes[b])
    comm.reduce_mix(mixes, root=0)
    ctx.sync()
    assert torch.equal(mixes.view(torch.int32), want.view(torch.int32))
    # argument checks
    from zang_amd import abi
    assert ctx.lib.zh_reduce_mix(comm.handle, C.c_void_p(mixes.data_ptr()), mixes.numel(), 1) == abi.ZH_ERR_INVALID
    assert ctx.lib.zh_allreduce_mix(None, C.c_void_p(mixes.data_ptr()), 4) == abi.ZH_ERR_INVALID
    assert ctx.lib.zh_allreduce_mix(comm.handle, None, 0) == abi.ZH_OK
    comm.close()


def test_collective_after_graph_replay_on_one_stream(ctx):
    """bench.py's N > 1 region: a captured batch of mixdown paints, then the collective on the same stream, repeated --
    no host synchronisation in between, and the block equals the eager batch."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads, sharding
    V, B = 2048, 8
    freq, color, u2, _ = workloads.voice_params(5, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        dev = c2.device
        col, fr = torch.from_numpy(color).to(dev), torch.from_numpy(freq).to(dev)
        g1 = torch.from_numpy(u2).to(dev)
        me, mg = mod.NiceInstrument(V, col, c2), mod.NiceInstrument(V, col, c2)
        mix_e = torch.zeros((B, 2, F), dtype=torch.float32, device=dev); mix_g = torch.zeros_like(mix_e)
        sp = zang.Span(0, F)

        def batch(m, mixes):
            for b in range(B):
                m.paint_mix_stereo(sp, mixes[b, 0], mixes[b, 1], g1, g1, b == 0, m.Params(SR, fr, b < B // 2), zero_first=True)

        comm = sharding.Comm(c2, world=1, rank=0)
        batch(mg, mix_g); c2.sync()                       # lazy allocations outside the capture
        g = c2.capture(lambda: batch(mg, mix_g))
        batch(me, mix_e)
        for _ in range(3):
            batch(me, mix_e)
            g.launch()
            comm.allreduce_mix(mix_g)
        c2.sync()
        assert torch.equal(mix_e.view(torch.int32), mix_g.view(torch.int32))
        comm.close(); g.close(); c2.close()


def test_collective_recorded_inside_a_graph(ctx):
    """One collective per buffer (the reference mixes per buffer, write_wav.zig:58-93) recorded INTO the hipGraph next to the
    mixdown paints: zh_allreduce_mix on a capturing context becomes a graph node (RCCL supports stream capture), so a batch
    of buffers with its exchanges replays with one host call.  One rank: the replays equal the eager batch without exchange."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads, sharding
    V, B = 2048, 6
    freq, color, u2, _ = workloads.voice_params(5, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        dev = c2.device
        col, fr, g1 = torch.from_numpy(color).to(dev), torch.from_numpy(freq).to(dev), torch.from_numpy(u2).to(dev)
        me, mg = mod.NiceInstrument(V, col, c2), mod.NiceInstrument(V, col, c2)
        mix_e = torch.zeros((B, 2, F), dtype=torch.float32, device=dev); mix_g = torch.zeros_like(mix_e)
        sp = zang.Span(0, F)
        comm = sharding.Comm(c2, world=1, rank=0)

        def batch(m, mixes, exchange):
            for b in range(B):
                m.paint_mix_stereo(sp, mixes[b, 0], mixes[b, 1], g1, g1, b == 0, m.Params(SR, fr, b < B // 2), zero_first=True)
                if exchange:
                    comm.allreduce_mix(mixes[b])
        batch(mg, mix_g, True); c2.sync()                 # (lazy allocations and RCCL's first launch outside the capture)
        g = c2.capture(lambda: batch(mg, mix_g, True))
        batch(me, mix_e, False)
        for _ in range(3):
            batch(me, mix_e, False)
            g.launch()
        c2.sync()
        assert torch.equal(mix_e.view(torch.int32), mix_g.view(torch.int32))
        assert me.state().tobytes() == mg.state().tobytes()
        comm.close(); g.close(); c2.close()


_TORCH_NCCL = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(48 * 2 * 1024, dtype=torch.float32, device="cuda").reshape(48, 2, 1024) * 0.25
want = x.clone()
dist.all_reduce(x)                                 # torch.distributed backend "nccl" = RCCL: the form bench.py --exchange torch uses
torch.cuda.synchronize()
assert torch.equal(x, want)
# the library's communicator beside torch's in one process (one librccl: the copy torch loaded)
import zang_amd
from zang_amd import sharding
ctx = zang_amd.default_context()
comm = sharding.Comm(ctx)                          # world 1 taken from the process group
comm.allreduce_mix(x); ctx.sync()
assert torch.equal(x, want)
print("rccl_lib", ctx.lib.zh_comm_library().decode(), ctx.lib.zh_comm_version())
comm.close()
dist.barrier(); dist.destroy_process_group()
print("ok")
"""


def test_torch_nccl_backend_one_rank(ctx):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-c", _TORCH_NCCL % (ROOT, port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_comm_create_times_out_when_a_rank_never_arrives():
    """zh_comm_create is a rendezvous: rank 0 of a world of 2 whose partner never calls it.  The rendezvous is bounded BY DEFAULT
    (180 s; non-blocking ncclCommInitRankConfig polled through ncclCommGetAsyncError, ncclCommAbort at the limit): here the limit
    is set to 4 s through the API -- no environment variable -- and the call comes back with ZH_ERR_COMM and a message.  The
    same process then creates a one-rank communicator and runs a collective on it: nothing was left behind."""
    code = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, %r)
assert "ZH_COMM_TIMEOUT_S" not in os.environ
import torch
import zang_amd
from zang_amd import abi, sharding
ctx = zang_amd.default_context()
lib = ctx.lib
uid = (C.c_uint8 * abi.COMM_ID_BYTES)()
assert lib.zh_comm_unique_id(uid) == 0
assert lib.zh_comm_set_timeout(4.0) == 0
h = C.c_void_p()
t0 = time.time()
rc = lib.zh_comm_create(ctx.handle, 2, 0, uid, C.byref(h))
dt = time.time() - t0
print("rc", rc, "seconds %%.1f" %% dt, "message:", lib.zh_comm_last_error().decode())
ok = rc == abi.ZH_ERR_COMM and 3.0 < dt < 60.0 and not h.value
lib.zh_comm_set_timeout(120.0)
comm = sharding.Comm(ctx, world=1, rank=0)
x = torch.arange(1024, dtype=torch.float32, device=ctx.device)
comm.allreduce_mix(x); ctx.sync(); comm.check()
ok = ok and bool((x.cpu() == torch.arange(1024, dtype=torch.float32)).all())
comm.close()
print("after the timeout: one-rank communicator ok")
sys.stdout.flush()
os._exit(0 if ok else 1)
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k != "ZH_COMM_TIMEOUT_S"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "no rendezvous within 4 s" in r.stdout and "one-rank communicator ok" in r.stdout
