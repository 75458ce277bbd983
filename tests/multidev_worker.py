"""One rank of the multi-device tests (tests/test_gpu_multidevice.py): started N times by the test with RANK / WORLD_SIZE /
MASTER_* in the environment, rank r on GPU r.  MULTIDEV_EMULATE=1: every rank on device 0 and gloo instead of RCCL for the
collective (two ranks on one device cannot form an RCCL communicator) -- what a one-GPU box can run of the same code.

    python -m tests.multidev_worker comm        sharding.Comm (zh_comm_* over RCCL): all-reduce and reduce of the bench's
                                                [48][2][1024] block against the host-side sum of every rank's block
    python -m tests.multidev_worker slots       SlotExchange: peer IPC mapping, stores over xGMI into the root's slots,
                                                zh_sum_slots in rank order, bit-exact against the same order on the host
    python -m tests.multidev_worker render      config 5 sharded: each rank renders its contiguous range of 16,384 voices
                                                (fused NiceInstrument + stereo mixdown), the partial mixes are summed by the
                                                collective; rank 0 also renders all 16,384 voices alone.  Per-voice images of
                                                the shard bit-exact against the same voices of the one-rank render; the
                                                summed mix within sqrt(V) * eps * sum|x| of the one-rank mix (another order
                                                of the same f32 additions).

Exit code 0 = this rank's checks passed AND every other rank's (agreed over gloo at the end)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, F = 48000.0, 1024


def main():
    case = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    emulate = os.environ.get("MULTIDEV_EMULATE") == "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")                       # the host channel: ids, handles, verdicts
    dev = 0 if emulate else rank
    ok, why = True, ""
    try:
        if dev >= torch.cuda.device_count():
            raise RuntimeError(f"rank {rank} needs GPU {dev}, the node shows {torch.cuda.device_count()}")
        torch.cuda.set_device(dev)
        import zang_amd
        from zang_amd import abi, modules as mod, sharding, workloads, zang
        ctx = zang_amd.Context(dev)
        ctx.lib.zh_comm_set_timeout(120.0)

        def gather_host(t):
            """every rank's tensor, on the host, in rank order"""
            parts = [None] * world
            dist.all_gather_object(parts, t.cpu().numpy())
            return parts

        class GlooSum:                                   # MULTIDEV_EMULATE: the same interface over torch.distributed gloo
            def allreduce_mix(self, t):
                sharding.allreduce_mix(t); return t

            def reduce_mix(self, t, root=0):
                sharding.allreduce_mix(t); return t

            def check(self):
                pass

            def close(self):
                pass

        def make_comm():
            return GlooSum() if emulate else sharding.Comm(ctx, control_group=dist.group.WORLD)

        if case == "comm":
            n = 48 * 2 * F
            mine = torch.from_numpy(((np.arange(n) % 251) * (rank + 1) + 0.25 * rank).astype(np.float32)).to(ctx.device)
            parts = gather_host(mine)
            comm = make_comm()
            x = mine.clone()
            comm.allreduce_mix(x); ctx.sync(); comm.check()
            want = parts[0].astype(np.float64)
            for p in parts[1:]:
                want = want + p                           # small multiples of 0.25: exact in any order
            if not np.array_equal(x.cpu().numpy().astype(np.float64), want):
                raise RuntimeError("all-reduce != host sum")
            for b in range(0, 48, 7):                     # the per-buffer granularity too
                y = mine.view(48, 2 * F)[b].clone()
                comm.allreduce_mix(y); ctx.sync()
                if not np.array_equal(y.cpu().numpy().astype(np.float64), want.reshape(48, 2 * F)[b]):
                    raise RuntimeError("per-buffer all-reduce != host sum")
            root = world - 1
            z = mine.clone()
            comm.reduce_mix(z, root=root); ctx.sync(); comm.check()
            if rank == root and not np.array_equal(z.cpu().numpy().astype(np.float64), want):
                raise RuntimeError("reduce != host sum on the root")
            comm.close()
        elif case == "slots":
            n = 2 * F * 3
            ex = sharding.SlotExchange(ctx, n, control_group=dist.group.WORLD)
            rng = np.random.default_rng(100 + rank)
            mine = rng.uniform(-1, 1, n).astype(np.float32)   # arbitrary floats: the ORDER of the sum shows in the bits
            abi.check(ctx.lib.zh_upload(ctx.handle, C.c_void_p(ex.slot().addr), mine.ctypes.data, n * 4), "zh_upload into the root's slot")
            dst = torch.zeros(n, dtype=torch.float32, device=ctx.device) if rank == 0 else None
            ex.finish(dst)
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            if rank == 0:
                want = parts[0].copy()
                for p in parts[1:]:
                    want = (want + p).astype(np.float32)  # ((slot_0 + slot_1) + slot_2) + ...: zh_sum_slots' order
                got = dst.cpu().numpy()
                if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                    raise RuntimeError("rank-ordered slot sum differs from the same order on the host")
            ex.close()
        elif case == "render":
            V_total, B = 16384, 6
            lo, hi = sharding.voice_range(V_total, rank, world)

            def render(first, n, images=False):
                freq, color, u2, _ = workloads.voice_params(5, first, n)
                m = mod.NiceInstrument(n, torch.from_numpy(color).to(ctx.device), ctx)
                fr = torch.from_numpy(freq).to(ctx.device)
                pan = (2.0 * u2 - 1.0).astype(np.float32)
                gl = (np.float32(0.0) + ((np.float32(0.0) + pan * np.float32(0.5)) + np.float32(0.5))).astype(np.float32)
                gr = (np.float32(0.0) + ((np.float32(0.0) + gl * np.float32(-1.0)) + np.float32(1.0))).astype(np.float32)
                tgl, tgr = torch.from_numpy(gl).to(ctx.device), torch.from_numpy(gr).to(ctx.device)
                mixes = torch.zeros((B, 2, F), dtype=torch.float32, device=ctx.device)
                imgs = []
                for b in range(B):
                    on, new = b < B // 2, b == 0
                    m.paint_mix_stereo(zang.Span(0, F), mixes[b, 0], mixes[b, 1], tgl, tgr, new, m.Params(SR, fr, on), zero_first=True)
                ctx.sync()
                if images:
                    m2 = mod.NiceInstrument(n, torch.from_numpy(color).to(ctx.device), ctx)
                    for b in range(B):
                        img = ctx.image(F, n)
                        m2.paint(zang.Span(0, F), [img], [], b == 0, m2.Params(SR, fr, b < B // 2), zero_first=True)
                        imgs.append(img)
                    ctx.sync()
                return mixes, imgs, (gl, gr)

            part, imgs, _ = render(lo, hi - lo, images=True)
            comm = make_comm()
            summed = part.clone()
            comm.allreduce_mix(summed); ctx.sync(); comm.check()
            comm.close()
            # the one-rank render of ALL voices, on this rank's own GPU (every rank checks its own slice of it)
            whole, whole_imgs, (gl, gr) = render(0, V_total, images=True)
            for b in range(B):
                a = imgs[b].cpu().numpy(); w = whole_imgs[b][:, lo:hi].cpu().numpy()
                if not np.array_equal(a.view(np.uint32), w.view(np.uint32)):
                    raise RuntimeError(f"buffer {b}: the shard's voice images differ from the same voices of the one-rank render")
            # the mix: another association of the same f32 additions -- |difference| <= sqrt(V) * eps * sum_v |x_v * gain_v| per sample
            bound = np.zeros((B, 2, F))
            for b in range(B):
                x = np.abs(whole_imgs[b].cpu().numpy().astype(np.float64))        # [F][V]
                bound[b, 0] = x @ np.abs(gl.astype(np.float64)); bound[b, 1] = x @ np.abs(gr.astype(np.float64))
            eps = float(np.finfo(np.float32).eps)
            diff = np.abs(summed.cpu().numpy().astype(np.float64) - whole.cpu().numpy().astype(np.float64))
            tol = np.sqrt(V_total) * eps * bound + 1e-30
            if not (diff <= tol).all():
                raise RuntimeError("the summed shard mixes differ from the one-rank mix by %.3g of the bound" % float((diff / tol).max()))
            if float(np.abs(whole.cpu().numpy()).max()) == 0.0:
                raise RuntimeError("silent render")
        else:
            raise RuntimeError("unknown case " + case)
        ctx.close()
    except Exception as e:      # noqa: BLE001
        ok, why = False, f"{type(e).__name__}: {e}"
        sys.stderr.write(f"multidev_worker rank {rank} ({case}): {why}\n")
    verdicts = [None] * world
    try:
        dist.all_gather_object(verdicts, (ok, why))
        dist.destroy_process_group()
    except Exception:           # noqa: BLE001
        ok = False
    good = ok and all(v and v[0] for v in verdicts)
    if rank == 0:
        print(("PASS " if good else "FAIL ") + case + f" world {world}" + (" (emulated on one device, gloo)" if emulate else ""), flush=True)
    os._exit(0 if good else 1)


if __name__ == "__main__":
    main()
