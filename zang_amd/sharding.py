"""Voice sharding across the GPUs of one node.

Voices share nothing (module state is per instance; Noise is seeded by GLOBAL voice index),
so painting needs no inter-GPU traffic.  The only exchange is the final mixdown: each rank
reduces its own voices to a [channels][frames] partial on its GPU, and the partials are summed.
Two forms of that one step:

* `Comm`           -- the library's own RCCL communicator (C ABI zh_comm_* / zh_allreduce_mix / zh_reduce_mix,
  csrc/comm.hip): the collective is enqueued on the context's stream; torch.distributed is only the host channel
  that carries rank 0's 128-byte id to the other ranks (a Zig / C++ host uses whatever channel it has);
* `allreduce_mix`  -- the same sum through torch.distributed (RCCL over xGMI on GPUs; gloo in the CPU tests);
* `SlotExchange`   -- the root GPU owns one slot per rank in its own HBM, every rank's mixdown kernel stores its
  partial STRAIGHT into its slot (peer stores over xGMI, HIP IPC mapping, csrc/xchg.hip), and the root adds the
  slots in rank order: a fixed order, so the mix is reproducible bit for bit whatever the link timing.
"""
import ctypes as C

import torch.distributed as dist


def voice_range(total_voices, rank, world):
    """Contiguous shard [lo, hi) of `total_voices` for `rank` of `world`."""
    lo = total_voices * rank // world
    hi = total_voices * (rank + 1) // world
    return lo, hi


def allreduce_mix(mix, group=None):
    """Sum the per-rank partial mixes in place (float32 [..., frames], on the rank's device)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if mix.is_cuda and dist.get_backend(group) == "gloo":     # CPU-only backend (tests, dry runs): stage through host
            host = mix.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            mix.copy_(host)
        else:
            dist.all_reduce(mix, op=dist.ReduceOp.SUM, group=group)
    return mix


class Comm:
    """RCCL communicator of libzang_hip.so (include/zang_hip.h zh_comm_*), one rank per process.

    `Comm(ctx)` with an initialised torch.distributed process group: rank 0 makes the id, `control_group` (any backend
    that can broadcast Python objects; default: the world group) carries it, every rank creates its communicator.
    `Comm(ctx, world=1, rank=0)` needs no process group (a one-rank communicator: RCCL init + launch, the sum is the
    identity).  Every rank first learns whether EVERY rank found librccl, so that none waits alone in RCCL's bootstrap."""

    def __init__(self, ctx, world=None, rank=None, control_group=None):
        from . import abi
        self._abi, self.ctx, self.lib = abi, ctx, ctx.lib
        self.handle = None
        use_dist = world is None
        if use_dist:
            world, rank = dist.get_world_size(control_group), dist.get_rank(control_group)
        self.world, self.rank = int(world), int(rank)
        ok = int(self.lib.zh_comm_available())
        if use_dist and self.world > 1:
            import torch
            t = torch.tensor([ok], dtype=torch.int32)
            if dist.get_backend(control_group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group)
            ok = int(t.item())
        if not ok:
            raise abi.ZangHipError("zh_comm: librccl is not available on every rank: " + self.lib.zh_comm_last_error().decode())
        uid = (C.c_uint8 * abi.COMM_ID_BYTES)()
        payload = [None]
        err = None
        if self.rank == 0:
            rc = self.lib.zh_comm_unique_id(uid)
            if rc == abi.ZH_OK:
                payload = [bytes(uid)]
            else:
                err = "zh_comm_unique_id failed: %d (%s)" % (rc, self.lib.zh_comm_last_error().decode())
        if self.world > 1:
            if not use_dist:
                raise abi.ZangHipError("Comm(world > 1) needs torch.distributed as the host channel for the id")
            dist.broadcast_object_list(payload, src=0, group=control_group)         # None = rank 0 could not make the id
            if payload[0] is None:
                raise abi.ZangHipError("zh_comm: " + (err or "rank 0 could not make the communicator id"))
            C.memmove(uid, payload[0], abi.COMM_ID_BYTES)
        elif err:
            raise abi.ZangHipError("zh_comm: " + err)
        h = C.c_void_p()
        rc = self.lib.zh_comm_create(ctx.handle, self.world, self.rank, uid, C.byref(h))
        # every rank learns whether EVERY rank has its communicator (a rank that failed alone would leave the others waiting
        # for it in the first collective)
        good = 1 if rc == abi.ZH_OK else 0
        if use_dist and self.world > 1:
            import torch
            t = torch.tensor([good], dtype=torch.int32)
            if dist.get_backend(control_group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group)
            good_all = int(t.item())
        else:
            good_all = good
        if not good_all:
            if good:
                self.lib.zh_comm_destroy(h)
            raise abi.ZangHipError("zh_comm_create failed on %s: %d (%s)" % ("this rank" if not good else "another rank", rc,
                                                                            self.lib.zh_comm_last_error().decode()))
        self.handle = h

    def allreduce_mix(self, mix):
        """Sum `mix` (contiguous float32 tensor on the context's device) over the ranks in place, on the context's stream."""
        self._abi.check(self.lib.zh_allreduce_mix(self.handle, C.c_void_p(mix.data_ptr()), mix.numel()), "zh_allreduce_mix")
        return mix

    def reduce_mix(self, mix, root=0):
        """Like allreduce_mix, but only `root` ends up with the sum (the other ranks' `mix` is unspecified)."""
        self._abi.check(self.lib.zh_reduce_mix(self.handle, C.c_void_p(mix.data_ptr()), mix.numel(), root), "zh_reduce_mix")
        return mix

    def check(self):
        """Raise if RCCL has recorded an asynchronous error on this communicator (a peer that died, a failed transport):
        zh_comm_check = ncclCommGetAsyncError.  Cheap; call it between batches."""
        self._abi.check(self.lib.zh_comm_check(self.handle), "zh_comm_check: " + self.lib.zh_comm_last_error().decode())

    def close(self):
        if self.handle is not None:
            self.lib.zh_comm_destroy(self.handle)
            self.handle = None

    def abort(self):
        """End the communicator without the collective hand-shake of close() (peers may be gone)."""
        if self.handle is not None:
            self.lib.zh_comm_abort(self.handle)
            self.handle = None


class DevicePtr:
    """A raw device address with the one method the module wrappers use of a tensor."""

    def __init__(self, addr):
        self.addr = int(addr)

    def data_ptr(self):
        return self.addr

    def __add__(self, nbytes):
        return DevicePtr(self.addr + int(nbytes))


class SlotExchange:
    """Direct-write exchange of the partial mixes.

    `floats` = floats of one rank's partial block (e.g. 48 buffers x 2 channels x 1024 frames).  Rank 0 allocates
    world x floats in its HBM (zh_ipc_alloc) and hands the 64-byte IPC handle to the other ranks over the host
    control group (gloo); they map it (zh_ipc_open).  Per batch:

        paint ... into  self.slot()        (device address of this rank's slot; float offsets are the caller's)
        self.finish(dst)                   every rank: stream sync, host barrier; root: dst (+)= slot_0 + slot_1 + ...
                                           in rank order, stream sync; host barrier (the slots may be rewritten)

    Host barriers order the processes (two per batch); no collective library is involved in the data path.
    """

    def __init__(self, ctx, floats, control_group=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.floats = int(floats)
        self.stride = (self.floats + 63) // 64 * 64                      # slots 256-byte aligned
        self.ctl = control_group if control_group is not None else dist.new_group(backend="gloo")
        from . import abi
        self._abi = abi
        base = C.c_void_p()
        handle = (C.c_uint8 * 64)()
        payload = [None]
        err = None
        if self.rank == 0:
            rc = self.lib.zh_ipc_alloc(ctx.handle, self.world * self.stride * 4, C.byref(base), handle)
            if rc == abi.ZH_OK:
                payload = [bytes(handle)]
            else:
                err = "zh_ipc_alloc failed: %d" % rc
        dist.broadcast_object_list(payload, src=0, group=self.ctl)         # None = the root could not allocate
        if self.rank != 0:
            if payload[0] is None:
                err = "the root rank could not allocate the slot block"
            else:
                C.memmove(handle, payload[0], 64)
                rc = self.lib.zh_ipc_open(ctx.handle, handle, C.byref(base))
                if rc != abi.ZH_OK:
                    err = "zh_ipc_open failed: %d" % rc
                    base = C.c_void_p()
        # every rank learns whether EVERY rank is set up: a rank that failed alone would otherwise leave the others
        # waiting for it in the first host barrier of finish()
        import torch
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.ctl)
        self.base = base.value or 0
        self.owner = self.rank == 0
        if int(ok.item()) == 0:
            if self.base:
                if self.owner:
                    self.lib.zh_free(ctx.handle, C.c_void_p(self.base))
                else:
                    self.lib.zh_ipc_close(ctx.handle, C.c_void_p(self.base))
                self.base = 0
            raise abi.ZangHipError("SlotExchange: " + (err or "another rank could not map the root's slot block"))

    def slot(self, rank=None):
        r = self.rank if rank is None else rank
        return DevicePtr(self.base + r * self.stride * 4)

    def finish(self, dst=None, zero_first=True):
        """All ranks call it after enqueueing their paints.  On return the root's `dst` (float32 CUDA tensor of
        `floats` elements; root only) holds the rank-ordered sum and the slots may be written again."""
        abi = self._abi
        self.ctx.sync()                                                    # this rank's stores have landed in the root's HBM
        dist.barrier(group=self.ctl)
        if self.owner and dst is not None:
            abi.check(self.lib.zh_sum_slots(self.ctx.handle, dst.data_ptr(), C.c_void_p(self.base), self.world, self.stride,
                                            self.floats, abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD), "zh_sum_slots")
            self.ctx.sync()
        dist.barrier(group=self.ctl)

    def close(self):
        if self.base:
            if self.owner:
                dist.barrier(group=self.ctl)                               # every mapping is closed before the block is freed
                self.lib.zh_free(self.ctx.handle, C.c_void_p(self.base))
            else:
                self.lib.zh_ipc_close(self.ctx.handle, C.c_void_p(self.base))
                dist.barrier(group=self.ctl)
            self.base = 0
