#!/usr/bin/env python3
"""bench.py -- the BASELINE.json headline metric: voice-samples/sec of the paint() hot path.

One "step" = what the reference does per 1024-frame buffer for every voice of the
workload: `zang.zero(span, out)` then `Module.paint(span, {out}, ...)` (e.g.
examples/modules.zig:220-225), with state carried from buffer to buffer.

Default workload at N=1: BASELINE.json configs[1] -- 4096 PulseOsc voices x 1024 frames,
constant per-voice frequency, 48 kHz (the configuration the HBM-roofline target is quoted on).
Default workload at N>1: BASELINE.json configs[4] -- the NiceInstrument swarm (Osc+Env+Filter
fused, voice mixdown in the same kernel), 131,072 voices per GPU (1,048,576 at N=8), two output
channels, with the one exchange step of the multi-GPU path in the timed region: the GPUs'
partial mixes are summed by an RCCL all-reduce over xGMI, one collective per 48-buffer batch.
The N=1 line carries the same 131,072-voice shard without the collective in `config5_shard`,
and the N>1 line times it again on every rank (`single_gpu_shard`) for the scaling factor.

Launching: `python bench.py --gpus N` starts N ranks itself (fresh child processes, started
before anything touches a GPU; rank 0's line is relayed); under `torch.distributed.run`
(RANK / WORLD_SIZE in the environment) the process is one of the ranks.

Output images rotate through a ring larger than the 256 MiB Infinity Cache so the stores
really reach HBM (MI355X_MICROARCH.md "Infinity Cache").  Inputs (per-voice params, module
state) are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
HBM_STORE_GBS = 6200.0  # same guide: measured plain-store rate ("achievable" write bandwidth)
SR = 48000.0


PATTERN = 48            # buffers of the note on/off pattern of the nice / nice_mix / script workloads


PREFLIGHT_FAILED = 5    # exit code of `--preflight` when a check fails (distinct from a crash (1), a bad flag (2), no device (3), the p2p watchdog (4))


def pattern_steps(workload, steps):
    """Timed steps for `--steps steps`: the note-pattern workloads time whole patterns (the next multiple of 48), every other
    workload exactly `steps`.  The line reports the steps really timed as `steps` and the flag's value as `steps_requested`."""
    if workload in ("nice", "nice_mix", "script"):
        return max(1, -(-steps // PATTERN)) * PATTERN
    return steps


SCRIPT_MODULE = os.environ.get("ZH_SCRIPT_MODULE", "Lead")    # any module of tests/golden/script_modules.txt with (freq, note_on) params


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); without WORLD_SIZE in the environment N > 1 spawns them")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 1000; 960 for the 48-buffer-batch workloads)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default=None, choices=["pulseosc", "noise_filter", "noise_filter_fused", "nice", "nice_mix", "script"],
                    help="default: pulseosc (config 2) on one GPU, nice_mix (config 5) on several")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU (default 4096; 131072 for nice_mix on several GPUs)")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=2, choices=[1, 2], help="nice_mix: output channels of the mixdown (north_star: stereo)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch", "p2p"],
                    help="nice_mix on several GPUs: RCCL all-reduce through the library's own communicator (C ABI zh_comm_* / "
                         "zh_allreduce_mix, on the launch stream), the same through torch.distributed, or direct stores into the "
                         "root's slots + fixed-order sum")
    ap.add_argument("--ring-mib", type=int, default=512, help="bytes of distinct output images to rotate through")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU-baseline sample length")
    ap.add_argument("--repeats", type=int, default=None, help="further K-step regions timed after the first (default: up to 30 when the region is short)")
    ap.add_argument("--no-rehearsal", action="store_true", help="skip the untimed rehearsal regions before the timed one")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-run oracle comparison of one buffer")
    ap.add_argument("--no-config5", action="store_true", help="N=1: skip the extra config-5 shard measurement")
    ap.add_argument("--no-config4", action="store_true", help="N=1: skip the extra config-4 measurement (60 s of the generated song, GPU and oracle)")
    ap.add_argument("--p2p-compare", action="store_true",
                    help="N>1: after the line is complete, also measure the direct-write exchange beside the RCCL one (opt-in: a stalled "
                         "peer mapping ends the run with exit code 4 after --p2p-timeout, the line printed first)")
    ap.add_argument("--p2p-timeout", type=float, default=180.0, help="N>1 with --p2p-compare: seconds the comparison may take")
    ap.add_argument("--preflight", action="store_true",
                    help="no timing: check the multi-GPU plumbing first -- `tests/cpp/comm_host N` (N rank processes form an RCCL "
                         "communicator through the C ABI and sum a block) as a child process, then one HIP-IPC slot round trip per peer "
                         "(zh_ipc_alloc / zh_ipc_open / peer store / zh_sum_slots); prints pass/fail per rank as one JSON line; exit code "
                         "%d on any failure" % PREFLIGHT_FAILED)
    ap.add_argument("--tolerant", action="store_true",
                    help="noise_filter / noise_filter_fused / nice / nice_mix (few voices): paint with ZH_PAINT_TOLERANT (opt-in time-parallel Filter forms, 1e-5 of the "
                         "signal's peak instead of bits; csrc/filter_tp.hip.h); script: the module's sines that may be f32; the line says so in config.tolerant")
    ap.add_argument("--eager", action="store_true", help="one host launch per step instead of hipGraph replay")
    ap.add_argument("--pad-voices", type=int, default=None, help="row padding of the output images in voices (default: the library's choice, Context.image)")
    return ap.parse_args(argv)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never imports torch or
    touches a GPU), relay rank 0's JSON line, exit non-zero if any rank fails."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        # rank 0's stdout carries the line; the other ranks print nothing there
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))
    rc, out = 0, b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                if r == 0:
                    try:
                        o, _ = procs[0].communicate(timeout=0.2)
                        out += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                elif procs[r].poll() is None:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode or 1
            if rc:
                break
            time.sleep(0.05)
    finally:
        for pr in procs:                         # a failed rank leaves the others waiting in a rendezvous / collective
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    if rc:
        sys.stderr.write(f"bench.py: a rank exited with code {rc}\n")
    return rc


def comm_host_exe():
    """tests/cpp/comm_host, built from tests/cpp/comm_host.c against the library beside it when missing or stale (gcc only)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "comm_host.c")
    exe = os.path.join(ROOT, "tests", "cpp", "comm_host")
    libso = os.path.join(ROOT, "zang_amd", "libzang_hip.so")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(libso)):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), src,
                               "-L" + os.path.join(ROOT, "zang_amd"), "-lzang_hip", "-Wl,-rpath," + os.path.join(ROOT, "zang_amd"),
                               "-L" + rocm + "/lib", "-Wl,-rpath," + rocm + "/lib", "-o", exe])
    return exe


def preflight_comm_host(n):
    """`comm_host n` as a fresh child process (this process need not have touched a GPU): n rank processes, rank r on GPU r,
    RCCL through the C ABI, all-reduce and reduce of the bench's [48][2][1024] block checked on every rank."""
    import subprocess
    try:
        r = subprocess.run([comm_host_exe(), str(n)], capture_output=True, text=True, timeout=240,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        ok = r.returncode == 0 and r.stdout.strip().endswith("PASS")
        return {"ranks": n, "ok": ok, "exit_code": r.returncode, "output": (r.stdout + r.stderr).strip()[-600:]}
    except Exception as e:      # noqa: BLE001  (a hung rendezvous ends in TimeoutExpired: reported, not raised)
        return {"ranks": n, "ok": False, "exit_code": None, "output": f"{type(e).__name__}: {e}"[:600]}


def preflight_rank(args, world, rank, local_rank, emulate):
    """The per-rank half of `--preflight`: gloo rendezvous, a context on this rank's GPU, one SlotExchange round trip (the root
    allocates a slot per rank, every peer maps it with zh_ipc_open and stores a rank-dependent pattern into its slot through
    the mapping, the root adds the slots in rank order), pass/fail gathered per rank.  Returns the process exit code."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rec = {"rank": rank, "device": None, "ok": False, "error": None}
    comm_rec = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rank == 0 and os.environ.get("ZH_PREFLIGHT_COMM_DONE") != "1":
        # under torch.distributed.run there is no parent of ours: rank 0 starts comm_host before it touches a GPU itself
        n = min(world, torch.cuda.device_count()) if emulate else world
        comm_rec = preflight_comm_host(max(n, 1))
    if world > 1:
        dist.init_process_group("gloo")
    try:
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device")
        dev = 0 if (world == 1 or emulate) else local_rank
        rec["device"] = dev
        if dev >= torch.cuda.device_count():
            raise RuntimeError(f"needs GPU {dev}, this node shows {torch.cuda.device_count()}")
        torch.cuda.set_device(dev)
        import zang_amd
        from zang_amd import abi, sharding
        ctx = zang_amd.Context(dev)
        rec["zh_comm_available"] = int(ctx.lib.zh_comm_available())
        if world > 1:
            n = 2 * args.frames
            ex = sharding.SlotExchange(ctx, n, control_group=dist.group.WORLD)       # raises on EVERY rank if any rank failed
            mine = (np.arange(n, dtype=np.float32) % 251) * (rank + 1)
            abi.check(ctx.lib.zh_upload(ctx.handle, C.c_void_p(ex.slot().addr), mine.ctypes.data, n * 4), "zh_upload into the root's slot")
            dst = torch.zeros(n, dtype=torch.float32, device=ctx.device) if rank == 0 else None
            ex.finish(dst)
            if rank == 0:
                want = (np.arange(n, dtype=np.float32) % 251) * (world * (world + 1) // 2)
                if not np.array_equal(dst.cpu().numpy(), want):
                    raise RuntimeError("the rank-ordered sum of the slots is wrong")
            ex.close()
            rec["ipc"] = "root of the slot block" if rank == 0 else "zh_ipc_open + peer store: summed correctly by the root"
        rec["ok"] = True
    except Exception as e:      # noqa: BLE001
        rec["error"] = f"{type(e).__name__}: {e}"[:300]
    recs = [rec]
    if world > 1:
        recs = [None] * world
        dist.all_gather_object(recs, rec)
    code = 0
    if rank == 0:
        ok = all(r["ok"] for r in recs) and (comm_rec is None or comm_rec["ok"])
        line = {"preflight": {"ok": ok, "world": world, "ranks": recs}}
        if comm_rec is not None:
            line["preflight"]["comm_host"] = comm_rec
        print(json.dumps(line), flush=True)
        code = 0 if ok else PREFLIGHT_FAILED
    if world > 1:
        flag = [code]
        dist.broadcast_object_list(flag, src=0)
        code = flag[0]
        dist.destroy_process_group()
    return code


class Workload:
    """Builds the module(s), resident params and the per-step callable for one rank."""

    def __init__(self, name, ctx, V, F, first_voice, ring_bytes, world=1, pad=None, channels=1, exchange="rccl", tolerant=False, multi=None):
        import torch
        self.world = world
        self.multi = (world > 1) if multi is None else bool(multi)     # the exchange step runs (ZH_BENCH_ONE_RANK_DIST: with a world of one)
        self.tolerant = bool(tolerant)
        self.channels, self.exchange_kind = channels, exchange
        self.slots = None
        self.comm = None                # sharding.Comm (the library's RCCL communicator) when --exchange rccl
        self.batch_rows = 48            # rows of self.mixes one batch fills (a graph shorter than the note pattern fills fewer)
        from zang_amd import modules as mod, zang, workloads
        self.name, self.V, self.F = name, V, F
        self.ctx = ctx
        self.span = zang.Span(0, F)
        cfg = {"pulseosc": 2, "noise_filter": 3, "noise_filter_fused": 3, "nice": 5, "nice_mix": 5, "script": 5}[name]
        freq, color, u2, u3 = workloads.voice_params(cfg, first_voice, V)
        self.freq_h, self.color_h, self.u2_h, self.u3_h = freq, color, u2, u3
        dev = ctx.device
        self.freq = torch.from_numpy(freq).to(dev)
        self.color = torch.from_numpy(color).to(dev)
        img_bytes = V * F * 4
        nring = max(2, min(64, (ring_bytes + img_bytes - 1) // img_bytes))
        self.bytes_per_step = img_bytes            # algorithmic: 4 B written per voice-sample
        self.kernel = None
        self.kernels_launched = []
        if name == "pulseosc":
            self.m = mod.PulseOsc(V, ctx)
            self.ring = [ctx.image(F, V, pad=pad) for _ in range(nring)]
            self.params = self.m.Params(SR, zang.constant(self.freq), self.color)
            self.kernel = "k_osc_const4<PulseOscP>"
            self.step = self._step_pulse
        elif name == "noise_filter":
            self.noise = mod.Noise(V, ctx, first_seed=first_voice)
            self.flt = mod.Filter(V, ctx)
            cutoff_f = torch.from_numpy((200.0 + 7800.0 * u2)).to(dev)
            self.cutoff = mod.Filter.cutoffFromFrequency(cutoff_f, SR, ctx)
            self.res = torch.from_numpy((0.9 * u3)).to(dev)
            self.ring = [ctx.image(F, V, pad=pad) for _ in range(nring)]
            self.temp = ctx.image(F, V, pad=pad)
            self.kernel = "k_filter"                # (placeholder: Runner asks the library which kernels the step launched, zh_last_form)
            self.step = self._step_noise_filter
        elif name == "noise_filter_fused":
            self.m = mod.NoiseFilter(V, ctx, first_seed=first_voice)
            cutoff_f = torch.from_numpy((200.0 + 7800.0 * u2)).to(dev)
            self.cutoff = mod.Filter.cutoffFromFrequency(cutoff_f, SR, ctx)
            self.res = torch.from_numpy((0.9 * u3)).to(dev)
            self.ring = [ctx.image(F, V, pad=pad) for _ in range(nring)]
            self.params = self.m.Params(self.m_white(), mod.Filter.low_pass, self.cutoff, self.res)
            self.kernel = "k_noise_filter"
            self.step = self._step_noise_filter_fused
        elif name == "script":
            # a zangscript module compiled to ONE fused kernel at start-up (hiprtc): `Lead` of the repo's test
            # script = 5 SineOsc + 3 Envelope + arithmetic, two inlined script modules
            from zang_amd import script as zscript
            text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "script_modules.txt")).read()
            self.program = zscript.ScriptProgram(text, ctx, only=[SCRIPT_MODULE])
            self.m = self.program.module(SCRIPT_MODULE, V)
            self.ring = [ctx.image(F, V, pad=pad) for _ in range(nring)]
            self.kernel = "zs_paint_" + SCRIPT_MODULE
            self.step = self._step_script
            self.nsteps = 0
        elif name == "nice":
            self.m = mod.NiceInstrument(V, self.color, ctx)
            self.ring = [ctx.image(F, V, pad=pad) for _ in range(nring)]
            self.kernel = "k_nice"
            self.step = self._step_nice
            self.nsteps = 0
        else:
            self.m = mod.NiceInstrument(V, self.color, ctx)
            # one [channels][frames] partial mix per buffer of a 48-buffer batch; with several GPUs the batch is exchanged at once
            Cn = self.channels
            self.mixes = torch.zeros((48, Cn, F), dtype=torch.float32, device=dev)
            if Cn == 2:
                # a constant pan per voice through the reference's scaleWave(min 0, max 1) / invertWaveInPlace
                # (examples/example_stereo.zig:20-39,84-98): left = pan * 0.5 + 0.5, right = left * -1 + 1
                import numpy as np
                pan = (2.0 * u2 - 1.0).astype(np.float32)
                gl = (np.float32(0.0) + ((np.float32(0.0) + pan * np.float32(0.5)) + np.float32(0.5))).astype(np.float32)
                gr = (np.float32(0.0) + ((np.float32(0.0) + gl * np.float32(-1.0)) + np.float32(1.0))).astype(np.float32)
                self.gain_l, self.gain_r = torch.from_numpy(gl).to(dev), torch.from_numpy(gr).to(dev)
            self.targets = [[self.mixes[b, c] for c in range(Cn)] for b in range(48)]
            self.ring = []
            self.kernel = "k_nice_mix<%d>" % Cn
            self.step = self._step_nice_mix
            self.nsteps = 0
        self.nring = len(self.ring)
        self.i = 0
        self.steps_done = 0

    def graph_steps(self, steps=0):
        """Steps per captured graph.  The K timed steps as ONE graph when K is not huge (each graph launch costs a
        ~4 us bubble on the stream); otherwise a count between one and four ring rotations that divides K, so that
        no remainder of the timed steps has to be launched one by one from Python (~2x slower per step at the
        16 MiB size).  (Any count works: the oscillators' double-buffered state is reconciled by zh_graph_launch.)"""
        if self.name in ("nice", "nice_mix", "script"):
            # the note on/off pattern repeats every 48 buffers (note on for 0-23: attack, decay, sustain; off for 24-47:
            # release, idle) and the release buffers cost ~20 % more than the sustain ones: the graph is always the WHOLE
            # pattern, and a timed region is a whole number of patterns (pattern_steps) -- round 3 captured only buffers
            # 0..K-1 for K < 48, which kept the release stage out of the driver's 20-step region
            return PATTERN
        if os.environ.get("ZH_BENCH_G"):
            return int(os.environ["ZH_BENCH_G"])   # experiments
        g = max(self.nring, 2)
        if 2 <= steps <= 2049:
            return steps
        for cand in range(g, 4 * g + 1):
            if steps and steps % cand == 0:
                return cand
        return g

    def _next(self):
        o = self.ring[self.i]
        self.i = (self.i + 1) % self.nring
        return o

    def _step_pulse(self):
        # the per-voice params are resident and never change: every paint after the first says so
        # (ZH_PAINT_PARAMS_UNCHANGED: the kernel loads the per-voice constants the first paint stored; same bits)
        self.m.paint(self.span, [self._next()], [], False, self.params, zero_first=True,
                     params_unchanged=self.steps_done > 0 and os.environ.get("ZH_BENCH_NO_TABLE") != "1")
        self.steps_done += 1

    def _step_noise_filter(self):
        from zang_amd import zang
        mod_n, mod_f = self.noise, self.flt
        mod_n.paint(self.span, [self.temp], [], False, mod_n.Params(mod_n.white), zero_first=True)
        mod_f.paint(self.span, [self._next()], [], False,
                    mod_f.Params(self.temp, mod_f.low_pass, zang.constant(self.cutoff), zang.constant(self.res)),
                    zero_first=True, tolerant=self.tolerant)

    @staticmethod
    def m_white():
        return 0                                # Noise.Color.white

    def _step_noise_filter_fused(self):
        self.m.paint(self.span, [self._next()], None, False, self.params, zero_first=True, tolerant=self.tolerant)

    def _note_on(self):
        # config 5: note on for buffers 0-23, then off (attack -> decay -> sustain -> release), repeating
        k = self.nsteps % 48
        self.nsteps += 1
        return k < 24, k == 0

    def _step_nice(self):
        on, new = self._note_on()
        self.m.paint(self.span, [self._next()], [], new, self.m.Params(SR, self.freq, on), zero_first=True, tolerant=self.tolerant)

    def _step_script(self):
        on, new = self._note_on()
        self.m.paint(self.span, [self._next()], None, new, {"sample_rate": SR, "freq": self.freq, "note_on": on}, zero_first=True, tolerant=self.tolerant)

    def _step_nice_mix(self):
        row = self.nsteps % 48
        on, new = self._note_on()                                  # advances self.nsteps
        t = self.targets[row]
        P = self.m.Params(SR, self.freq, on)
        if self.channels == 2:
            self.m.paint_mix_stereo(self.span, t[0], t[1], self.gain_l, self.gain_r, new, P, zero_first=True, tolerant=self.tolerant)
        else:
            self.m.paint_mix(self.span, t[0], new, P, zero_first=True, tolerant=self.tolerant)

    def step_batch(self, B):
        """B consecutive steps of the stereo mixdown workload as ONE launch (zh_nice_paint_mix_stereo_batch: what an offline
        renderer that knows its params ahead would call; same bits as B single steps)."""
        rows = [(self.nsteps + j) % 48 for j in range(B)]
        flags = [self._note_on() for _ in range(B)]                # advances self.nsteps
        Ps = [self.m.Params(SR, self.freq, on) for (on, _) in flags]
        self.m.paint_mix_stereo_batch(self.span, [self.targets[r][0] for r in rows], [self.targets[r][1] for r in rows], self.gain_l, self.gain_r,
                                      [new for (_, new) in flags], Ps, zero_first=True)

    def use_slots(self, slots):
        """Direct-write exchange: the mixdown kernels store into this rank's slot of the root's block instead of
        self.mixes (zang_amd.sharding.SlotExchange); call before capturing the graph."""
        self.slots = slots
        base, F, Cn = slots.slot(), self.F, self.channels
        self.targets = [[base + ((b * Cn + c) * F * 4) for c in range(Cn)] for b in range(48)]

    def exchange(self):
        """config 5's one exchange step: the GPUs' partial mixes are summed over RCCL/xGMI.  A 4 KiB
        all-reduce per buffer would be pure latency (~20-40 us against 170 us of rendering), so the 48 buffers of a
        batch go in ONE all-reduce of [48][channels][frames] (192 KiB per channel) after the batch's launches -- an
        offline renderer only needs the mixed audio once the batch is done."""
        if self.name != "nice_mix" or not self.multi:
            return
        block = self.mixes[:self.batch_rows]
        if self.slots is not None:
            self.slots.finish(self.mixes)
        elif self.comm is not None:
            self.comm.allreduce_mix(block)          # zh_allreduce_mix: RCCL on the launch stream, right behind the batch's kernels
        else:
            from zang_amd import sharding
            sharding.allreduce_mix(block)

    def exchange_per_buffer(self):
        """The other granularity SURVEY.md 8e asks about: one collective per buffer ([channels][frames], 4-8 KiB) the way
        the reference's loop mixes per buffer (examples/write_wav.zig:58-93) -- `batch_rows` collectives instead of one."""
        for b in range(self.batch_rows):
            if self.comm is not None:
                self.comm.allreduce_mix(self.mixes[b])
            else:
                from zang_amd import sharding
                sharding.allreduce_mix(self.mixes[b])


def zig_probe():
    """probe `zig version` at run time and record the result; nothing depends on it (the reference
    sources never travel to the GPU box, so even with a toolchain the baseline stays the oracle)."""
    import shutil
    import subprocess
    exe = shutil.which("zig")
    if not exe:
        return "zig: not found on PATH"
    try:
        return subprocess.run([exe, "version"], capture_output=True, text=True, timeout=10).stdout.strip() or "zig: no output"
    except Exception as e:          # noqa: BLE001
        return f"zig: {type(e).__name__}"


def cpu_baseline(args, wl):
    """The oracle (C restatement of the reference loops; the Zig reference cannot be built
    in this image) timed on this host, one thread -- the reference's execution model --
    on a bounded sample: the first min(V, 4096) voices for about --cpu-seconds."""
    import ctypes as C
    import numpy as np
    from oracle import pyoracle as po
    L = po.lib()
    V, F = min(wl.V, 4096), wl.F
    if wl.name == "pulseosc":
        scratch = np.zeros(F, np.float32)
        states = (po.PulseOsc * V)()
        run = lambda n: L.zo_bench_pulseosc(V, F, n, SR, po.fptr(wl.freq_h), po.fptr(wl.color_h), states, po.fptr(scratch))
        what = "zero + PulseOsc.paint per voice"
    elif wl.name in ("noise_filter", "noise_filter_fused"):
        scratch = np.zeros(2 * F, np.float32)
        noise = (po.Noise * V)(); flt = (po.Filter * V)()
        for v in range(V):
            L.zo_noise_init(C.byref(noise[v]), v); L.zo_filter_init(C.byref(flt[v]))
        cutoff = np.array([L.zo_filter_cutoff_from_frequency(float(200.0 + 7800.0 * u), SR) for u in wl.u2_h[:V]], np.float32)
        res = (0.9 * wl.u3_h[:V]).astype(np.float32)
        run = lambda n: L.zo_bench_noise_filter(V, F, n, po.fptr(cutoff), po.fptr(res), noise, flt, po.fptr(scratch))
        what = "zero + Noise.paint + zero + Filter.paint per voice"
    elif wl.name == "script":
        # the same instruction list run the way the generated Zig would run it: buffer-level ops through temps
        from oracle import zs_interp
        V = min(V, 64)
        from oracle import zangscript as ozs              # the oracle-side front-end (test infrastructure), for the interpreter
        voices = zs_interp.make_voices(ozs.compile(wl.program.text, wl.program.filename), SCRIPT_MODULE, V)
        out = np.zeros(F, np.float32)
        state = {"n": 0}

        def run(n):
            for _ in range(n):
                k = state["n"] % 48
                state["n"] += 1
                for v in range(V):
                    out[:] = 0
                    voices[v].paint(0, F, out, k == 0, [np.float32(SR), np.float32(wl.freq_h[v]), k < 24])
        what = "script module %s through the oracle-side interpreter (oracle paints + buffer ops; Python dispatch per op)" % SCRIPT_MODULE
    else:
        scratch = np.zeros(3 * F, np.float32)
        inst = (po.NiceInstrument * V)()
        for v in range(V):
            L.zo_nice_init(C.byref(inst[v]), float(wl.color_h[v]))
        freq = np.ascontiguousarray(wl.freq_h[:V])
        run = lambda n: L.zo_bench_nice(V, F, n, SR, po.fptr(freq), inst, po.fptr(scratch))
        what = "NiceInstrument.paint (unfused module sequence through temps) per voice"
    t0 = time.perf_counter()
    run(4)                      # calibrate, then size the sample for ~cpu_seconds
    per = (time.perf_counter() - t0) / 4
    nbuf = max(4, int(args.cpu_seconds / per))
    t0 = time.perf_counter()
    run(nbuf)
    dt = time.perf_counter() - t0
    res = {"value": V * F * nbuf / dt, "unit": "voice-samples/s", "cores": 1, "kind": "port", "zig_version_probe": zig_probe(),
           "sample": f"{nbuf} consecutive buffers of {V} voices x {F} frames ({what}), {dt:.1f} s on 1 thread"}
    if wl.name == "pulseosc":
        # the same loops with the voices sharded over every host core (one thread each,
        # the ctypes calls release the GIL); reported beside the 1-thread figure, which stays `value`
        import threading
        T = os.cpu_count() or 1
        per = (V + T - 1) // T
        shards = [(a, min(a + per, V)) for a in range(0, V, per)]
        st = [(po.PulseOsc * (b - a))() for a, b in shards]
        sc = [np.zeros(F, np.float32) for _ in shards]
        fq = [np.ascontiguousarray(wl.freq_h[a:b]) for a, b in shards]
        cl = [np.ascontiguousarray(wl.color_h[a:b]) for a, b in shards]
        def run_all(nb):
            def work(i):
                a, b = shards[i]
                L.zo_bench_pulseosc(b - a, F, nb, SR, po.fptr(fq[i]), po.fptr(cl[i]), st[i], po.fptr(sc[i]))
            threads = [threading.Thread(target=work, args=(i,)) for i in range(len(shards))]
            t0 = time.perf_counter()
            for t in threads: t.start()
            for t in threads: t.join()
            return time.perf_counter() - t0
        d0 = run_all(64)                                     # calibrate, then ~cpu_seconds / 3 of work
        nb = max(64, min(200000, int(64 * (args.cpu_seconds / 3.0) / max(d0, 1e-3))))
        dtm = run_all(nb)
        res["all_cores"] = {"value": V * F * nb / dtm, "cores": len(shards), "sample": f"{nb} buffers, voices sharded over {len(shards)} threads, {dtm:.1f} s"}
    return res


def parity_check(wl, ctx):
    """SURVEY.md 8d "parity check in the bench": after the timed region, paint ONE more buffer eagerly from
    the state the timed steps left behind and compare it, bit for bit, with the oracle started from that same
    state -- every voice for V <= 65,536, a 4,096-voice stride sample above.  Checker only: nothing here is
    timed.  Returns None for workloads without a per-voice image (nice_mix) or without an oracle driver here."""
    import ctypes as C
    import numpy as np
    import torch
    if wl.name not in ("pulseosc", "nice", "noise_filter_fused"):
        return None
    from oracle import pyoracle as po
    L = po.lib()
    V, F = wl.V, wl.F
    sample = np.arange(V) if V <= 65536 else np.arange(0, V, V // 4096)[:4096]
    st = wl.m.state()
    out = wl.ring[0]
    if wl.name == "pulseosc":
        wl.m.paint(wl.span, [out], [], False, wl.params, zero_first=True)
    elif wl.name == "noise_filter_fused":
        wl.m.paint(wl.span, [out], None, False, wl.params, zero_first=True, tolerant=wl.tolerant)
        cut_h, res_h = wl.cutoff.cpu().numpy(), wl.res.cpu().numpy()
    else:
        wl.m.paint(wl.span, [out], [], True, wl.m.Params(SR, wl.freq, True), zero_first=True, tolerant=wl.tolerant)
    ctx.sync()
    got = out[:, torch.from_numpy(sample).to(out.device)].cpu().numpy().T
    ref = np.zeros((len(sample), F), np.float32)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for k, v in enumerate(sample):
        if wl.name == "pulseosc":
            o = po.PulseOsc(int(st["cnt"][v]))
            L.zo_pulseosc_paint(C.byref(o), 0, F, po.fptr(ref[k]), SR, po.constant(float(wl.freq_h[v])), float(wl.color_h[v]))
        elif wl.name == "noise_filter_fused":
            nz = po.Noise(); fl = po.Filter()
            for i in range(4):
                nz.r[i] = int(st["noise"]["r"][v][i])
            for i in range(7):
                nz.b[i] = float(st["noise"]["b"][v][i])
            fl.l, fl.b = float(st["flt"]["l"][v]), float(st["flt"]["b"][v])
            t0[:] = 0
            L.zo_noise_paint(C.byref(nz), 0, F, po.fptr(t0), 0)
            L.zo_filter_paint(C.byref(fl), 0, F, po.fptr(ref[k]), po.fptr(t0), 1, po.constant(float(cut_h[v])), po.constant(float(res_h[v])))
        else:
            o = po.NiceInstrument()
            L.zo_nice_init(C.byref(o), float(wl.color_h[v]))
            o.osc.cnt = int(st["osc"]["cnt"][v])
            o.flt.l, o.flt.b = float(st["flt"]["l"][v]), float(st["flt"]["b"][v])
            e = st["env"]
            o.env.state = int(e["state"][v])
            o.env.painter.t, o.env.painter.last_value, o.env.painter.start = float(e["t"][v]), float(e["last_value"][v]), float(e["start"][v])
            L.zo_nice_paint(C.byref(o), 0, F, po.fptr(ref[k]), po.fptr(t0), po.fptr(t1), 1, SR, float(wl.freq_h[v]), 1)
    same = got.view(np.uint32) == ref.view(np.uint32)
    rec = {"checked_voices": int(len(sample)), "frames": F, "bitexact": bool(same.all()),
           "mismatching_samples": int((~same).sum()), "against": "oracle from the GPU's own carried state"}
    if wl.tolerant and wl.name in ("nice", "noise_filter_fused"):
        # ZH_PAINT_TOLERANT's contract (include/zang_hip.h): every sample within 1e-5 of the voice's PEAK over the paint, from the
        # same start state; beside it the share of samples inside SURVEY.md 8d's per-sample metric |err| <= 1e-5 max(|ref|, 1e-3)
        #  -- near zero crossings that metric asks for less than one ulp of the filter's state
        g64, r64 = got.astype(np.float64), ref.astype(np.float64)
        err = np.abs(g64 - r64)
        peak = np.maximum(np.abs(r64).max(axis=1), 1e-300)
        rec["tolerant"] = {"worst_error_over_voice_peak": float((err.max(axis=1) / peak).max()), "allowed": 1e-5,
                           "within": bool(((err.max(axis=1) / peak) <= 1e-5).all()),
                           "fraction_inside_per_sample_metric": float((err <= 1e-5 * np.maximum(np.abs(r64), 1e-3)).mean()),
                           "fraction_bitexact": float(same.mean())}
    return rec


def config4_record(ctx, seconds=60.0, with_oracle=True):
    """BASELINE configs[3] beside the headline (extra key): `seconds` of the repo-authored song (tools/gen_song.py: the reference song's
    grammar and statistics; 17 sub-voices) rendered offline on the GPU -- scheduling on the host, one launch per 1..N buffers, mixDown to
    s16 -- and, as its CPU baseline, by the oracle-side renderer on one host thread; the two s16 payloads compared byte for byte."""
    import contextlib
    import importlib.util
    import io
    from zang_amd import song
    spec = importlib.util.spec_from_file_location("gen_song", os.path.join(ROOT, "tools", "gen_song.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gen.main(2900, 20240915)
    text = buf.getvalue()
    nbuf = int(seconds * SR) // 1024
    audio_s = nbuf * 1024 / SR
    r = song.SongRenderer(text, ctx)
    r.render_buffer(); ctx.sync()                        # first-launch costs
    r = song.SongRenderer(text, ctx)
    t0 = time.perf_counter()
    got = r.render(audio_s)
    gpu_s = time.perf_counter() - t0
    rec = {"workload": "example_song-like offline render (tools/gen_song.py, 2,900 rows), 17 sub-voices (3 PMOsc + 14 NiceInstrument), %d buffers = %.1f s of audio" % (nbuf, audio_s),
           "gpu_seconds": gpu_s, "x_realtime_gpu": audio_s / gpu_s, "payload_bytes": len(got)}
    if with_oracle:
        from oracle import pyoracle as po                 # (the checker and the CPU baseline of this key; nothing above used it)
        from tests.test_song import _oracle_song_render
        t0 = time.perf_counter()
        ref = _oracle_song_render(po, r.notes, song.EXAMPLE_SONG_INSTRUMENTS, nbuf)
        cpu_s = time.perf_counter() - t0
        rec.update({"oracle_seconds_1_thread": cpu_s, "x_realtime_oracle": audio_s / cpu_s, "payload_identical": got == ref})
    return rec


def parity_check_graph(run, ctx, max_voices=256):
    """The timed graph itself against the oracle (parity_check paints eagerly and never ran the coalesced or the pipelined
    launches): the state before one more replay, the replay, then buffer G - 1 of it (the last image it painted) against the oracle
    walked G buffers from that state -- `max_voices` voices on a stride (the oracle paints every buffer up to the last).
    Bit-exact, or for ZH_PAINT_TOLERANT within 1e-5 of the voice's peak per buffer carried."""
    import ctypes as C
    import numpy as np
    import torch
    wl, G = run.wl, run.G
    if run.graph is None or wl.name not in ("pulseosc", "noise_filter_fused") or not hasattr(wl, "graph_first_image"):
        return None
    from oracle import pyoracle as po
    L = po.lib()
    V, F = wl.V, wl.F
    sample = np.arange(V) if V <= max_voices else np.unique(np.linspace(0, V - 1, max_voices).astype(np.int64))
    st = wl.m.state()
    run.graph.launch()
    ctx.sync()
    out = wl.ring[(wl.graph_first_image + G - 1) % wl.nring]
    got = out[:, torch.from_numpy(sample).to(out.device)].cpu().numpy().T
    ref = np.zeros((len(sample), F), np.float32)
    t0 = np.zeros(F, np.float32)
    if wl.name == "noise_filter_fused":
        cut_h, res_h = wl.cutoff.cpu().numpy(), wl.res.cpu().numpy()
    for k, v in enumerate(sample):
        if wl.name == "pulseosc":
            o = po.PulseOsc(int(st["cnt"][v]))
            for _ in range(G):
                ref[k] = 0
                L.zo_pulseosc_paint(C.byref(o), 0, F, po.fptr(ref[k]), SR, po.constant(float(wl.freq_h[v])), float(wl.color_h[v]))
        else:
            nz = po.Noise(); fl = po.Filter()
            for i in range(4):
                nz.r[i] = int(st["noise"]["r"][v][i])
            for i in range(7):
                nz.b[i] = float(st["noise"]["b"][v][i])
            fl.l, fl.b = float(st["flt"]["l"][v]), float(st["flt"]["b"][v])
            for _ in range(G):
                t0[:] = 0; ref[k] = 0
                L.zo_noise_paint(C.byref(nz), 0, F, po.fptr(t0), 0)
                L.zo_filter_paint(C.byref(fl), 0, F, po.fptr(ref[k]), po.fptr(t0), 1, po.constant(float(cut_h[v])), po.constant(float(res_h[v])))
    same = got.view(np.uint32) == ref.view(np.uint32)
    rec = {"checked_voices": int(len(sample)), "buffer_of_the_replay": G - 1, "frames": F, "bitexact": bool(same.all()), "mismatching_samples": int((~same).sum()),
           "kernels": ["%s x%d" % kn for kn in run.graph_kernels],
           "against": "oracle walked %d buffers from the state before the replay" % G}
    if wl.tolerant:
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        peak = np.maximum(np.abs(ref.astype(np.float64)).max(axis=1), 1e-300)
        worst = float((err.max(axis=1) / peak).max())
        rec["tolerant"] = {"worst_error_over_voice_peak": worst, "allowed": 1e-5 * G, "within": worst <= 1e-5 * G,
                           "what": "carried over the replay's %d buffers from one start state: 1e-5 of the peak per buffer" % G}
    return rec


def parity_check_mix(wl, ctx, n=64):
    """The same for the mixdown workload, which has no per-voice image: `n` voices on a stride through the
    shard, two per launch -- zh_nice_paint_mix_stereo with ONE-HOT channel gains (left = voice a, right = voice b: every other
    voice enters the sum as x * 0 = +-0, the chosen one as x * 1 = x, so the channel IS that voice's samples whatever the
    summation order), every launch from the state the timed steps left behind (set_state), against the oracle started from
    that state.  Values compared (a +-0 sum has no sign to keep); the run ends on the carried state again."""
    import ctypes as C
    import numpy as np
    import torch
    if wl.name != "nice_mix" or wl.channels != 2:
        return None
    from oracle import pyoracle as po
    L = po.lib()
    V, F = wl.V, wl.F
    idx = np.arange(0, V, max(1, V // n))[:n]
    if len(idx) % 2:
        idx = idx[:-1]
    ctx.sync()
    st = wl.m.state()
    left = torch.zeros(F, dtype=torch.float32, device=ctx.device); right = torch.zeros_like(left)
    P = wl.m.Params(SR, wl.freq, True)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32); ref = np.zeros(F, np.float32)
    bad, checked, peak, worst, inside = 0, 0, 0.0, 0.0, 0
    for a, b in idx.reshape(-1, 2):
        gl = torch.zeros(V, dtype=torch.float32, device=ctx.device); gr = torch.zeros_like(gl)
        gl[int(a)] = 1.0; gr[int(b)] = 1.0
        wl.m.set_state(st)
        wl.m.paint_mix_stereo(wl.span, left, right, gl, gr, True, P, zero_first=True, tolerant=wl.tolerant)
        ctx.sync()
        for v, got in ((int(a), left.cpu().numpy()), (int(b), right.cpu().numpy())):
            o = po.NiceInstrument()
            L.zo_nice_init(C.byref(o), float(wl.color_h[v]))
            o.osc.cnt = int(st["osc"]["cnt"][v])
            o.flt.l, o.flt.b = float(st["flt"]["l"][v]), float(st["flt"]["b"][v])
            e = st["env"]
            o.env.state = int(e["state"][v])
            o.env.painter.t, o.env.painter.last_value, o.env.painter.start = float(e["t"][v]), float(e["last_value"][v]), float(e["start"][v])
            ref[:] = 0
            L.zo_nice_paint(C.byref(o), 0, F, po.fptr(ref), po.fptr(t0), po.fptr(t1), 1, SR, float(wl.freq_h[v]), 1)
            bad += int((got != ref).sum())
            peak = max(peak, float(np.abs(ref).max()))
            checked += 1
            err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
            vpeak = max(float(np.abs(ref).max()), 1e-300)
            worst = max(worst, float(err.max()) / vpeak)
            inside += int((err <= 1e-5 * np.maximum(np.abs(ref.astype(np.float64)), 1e-3)).sum())
    wl.m.set_state(st)
    rec = {"checked_voices": checked, "frames": F, "bitexact": bad == 0, "mismatching_samples": bad, "peak_abs_sample": peak,
           "against": "oracle from the GPU's own carried state; each voice isolated by one-hot channel gains of the fused mixdown kernel"}
    if wl.tolerant:
        # ZH_PAINT_TOLERANT's contract (include/zang_hip.h): every sample within 1e-5 of the voice's peak over the paint
        rec["tolerant"] = {"worst_error_over_voice_peak": worst, "allowed": 1e-5, "within": worst <= 1e-5,
                           "fraction_inside_per_sample_metric": inside / max(1, checked * F), "fraction_bitexact": 1.0 - bad / max(1, checked * F)}
    return rec


def traffic_record(args, V, F, kernel_hint=None, batch=False, driver_form=False, pipelined=False):
    """HBM bytes per launch of the workload's dominant kernel from the committed rocprofv3 --pmc passes of this same workload
    (counters cannot be read from inside the process; tools/pmc_traffic.sh collects them: one pass per counter, FETCH_SIZE
    doubled on gfx950): value + where it came from.  profiles/r03/pmc_traffic_<workload><voices>.json holds every kernel of a
    step; `traffic` is the kernel that moves the most bytes, the step's total goes into the source record."""
    if F != 1024:
        return None, None
    tol = "_tolerant" if getattr(args, "tolerant", False) else ""
    # the PMC passes of the form that ran: `_driver_args` = the recorded, coalesced graph at the driver's arguments (tools/pmc_traffic.sh
    # with PMC_GRAPH=1: counters of the launches a replay makes), else one launch per buffer (--eager)
    names = ([f"pmc_traffic_{args.workload}{V}{tol}_driver_args.json"] if (driver_form or batch or pipelined) else []) + [f"pmc_traffic_{args.workload}{V}{tol}.json"]
    path = next((p for p in (os.path.join(ROOT, "profiles", rnd, fn) for rnd in ("r06", "r05", "r04", "r03") for fn in names) if os.path.exists(p)),
                os.path.join(ROOT, "profiles", "r05", names[-1]))
    if os.path.exists(path):
        rec = json.load(open(path))
        ks = rec.get("kernels", {})
        if ks:
            hint = (kernel_hint or "").split("<")[0].split("[")[0]
            named = {n: v for n, v in ks.items() if hint and hint in n}
            if hint in BATCH_SUFFIX and named:
                suffix = BATCH_SUFFIX[hint][0 if batch else 1].split("(")[0]
                inst = {n: v for n, v in named.items() if n.rstrip().endswith(suffix)}
                if not inst and batch:
                    named = {}              # only one-buffer launches were counted: the caller scales them (see `traffic` below)
                else:
                    named = inst or named
            total = lambda v: v["write_bytes_per_launch"] + v["fetch_bytes_per_launch_corrected"]
            if pipelined and hint.endswith("_tp_ba") and not named:
                # the fused launch (pass B of buffer n + pass A of buffer n + 1) moves what the two kernels move: their sum
                parts = {n: v for n, v in ks.items() if "_tp_a" in n or "_tp_b" in n}
                if parts:
                    per_launch = sum(total(v) for v in parts.values())
                    src = {"file": os.path.relpath(path, ROOT), "collected_at_commit": rec.get("commit"), "kernel": hint + " = " + " + ".join(sorted(parts)),
                           "how": "tools/pmc_traffic.sh (separate --pmc passes of WRITE_SIZE / FETCH_SIZE, FETCH_SIZE x2 on gfx950)",
                           "kernels_per_step": {n: v["hbm_bytes_per_step"] for n, v in ks.items()},
                           "note": "a constant read from that file, not a counter of this run; pass A + pass B counted as separate launches and added"}
                    return per_launch, src
            if not named and batch:
                named = {n: v for n, v in ks.items() if hint and hint in n}
            name, k = max((named or ks).items(), key=lambda kv: total(kv[1]))
            per_launch = total(k)
            src = {"file": os.path.relpath(path, ROOT), "collected_at_commit": rec.get("commit"), "kernel": name,
                   "how": "tools/pmc_traffic.sh (separate --pmc passes of WRITE_SIZE / FETCH_SIZE, FETCH_SIZE x2 on gfx950)",
                   "hbm_bytes_per_step_all_kernels": rec.get("hbm_bytes_per_step"),
                   "kernels_per_step": {n: v["hbm_bytes_per_step"] for n, v in ks.items()},
                   "buffers_per_counted_launch": k.get("buffers_per_launch", 1),
                   "note": "a constant read from that file, not a counter of this run"}
            return per_launch, src
    if not (args.workload == "pulseosc" and V == 4096):
        return None, None
    for name in ("r02_pmc_pulseosc4096.json", "r01_pmc_pulseosc4096.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            rec = json.load(open(pmc))
            src = {"file": "profiles/" + name, "collected_at_commit": rec.get("commit"), "how": "tools/collect_pmc_traffic.sh (separate --pmc passes of WRITE_SIZE / FETCH_SIZE, FETCH_SIZE x2 on gfx950)",
                   "note": "a constant read from that file, not a counter of this run"}
            return rec["hbm_bytes_per_launch"], src
    return None, None


# kernels whose instantiations differ in a trailing template argument that zh_graph_kernels reports as "[batch]": the suffix of the
# rocprofv3 name of the instantiation that paints several buffers per launch / one buffer
BATCH_SUFFIX = {"k_osc_const4": ("true>(OscArgs)", "false>(OscArgs)")}


def select_kernel_row(rows, kernel, batch=False, launch_us=None):
    """Which row of a rocprofv3 kernel_stats.csv is the kernel the timed launches ran.  `kernel`: the name the library reported
    (zh_graph_kernels / zh_last_form: no template arguments); `batch`: the several-buffers-per-launch instantiation; `launch_us`:
    the HIP-event time per launch of this run.  Rows of that kernel, then the instantiation by its suffix where the family is known
    (BATCH_SUFFIX), else the row whose average is nearest the HIP-event time; -> (row, how) or (None, why)."""
    base = (kernel or "").split("<")[0].split("[")[0]
    if not base:
        return None, "no kernel name"
    named = [r for r in rows if (" " + base + "<") in (" " + r.get("Name", "")) or (" " + base + "(") in (" " + r.get("Name", ""))]
    if not named:
        return None, "no row of " + base
    if base in BATCH_SUFFIX:
        suffix = BATCH_SUFFIX[base][0 if batch else 1]
        inst = [r for r in named if r["Name"].rstrip().endswith(suffix)]
        if inst:
            return max(inst, key=lambda r: float(r.get("TotalDurationNs", 0) or 0)), "the %s instantiation of %s (..%s)" % ("batch" if batch else "one-buffer", base, suffix)
    if len(named) > 1 and launch_us:
        return min(named, key=lambda r: abs(float(r.get("AverageNs", 0) or 0) / 1e3 - launch_us)), "the instantiation of %s nearest this run's HIP-event time per launch" % base
    return max(named, key=lambda r: float(r.get("TotalDurationNs", 0) or 0)), "the only / largest row of " + base


def rocprof_record(args, V, kernel=None, batch=False, launch_us=None, driver_form=False):
    """The rocprofv3 --kernel-trace --stats average of the kernel the timed launches ran (select_kernel_row), from the committed summary of
    this workload -- collected at the driver's arguments (`*_driver_args_kernel_stats.csv`) when this run is the driver's form -- the
    figure `roofline.launch_ms_hip_events` should agree with and `roofline.frac_kernel` is made from."""
    import csv
    tag = {4096: "4096", 65536: "65536", 131072: "131072", 1048576: "1M"}.get(V)
    if tag is None:
        return None
    if getattr(args, "tolerant", False):
        tag += "_tolerant"
    names = ([f"{args.workload}{tag}_driver_args_kernel_stats.csv"] if driver_form else []) + [f"{args.workload}{tag}_kernel_stats.csv"]
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        for fn in names:
            path = os.path.join(ROOT, "profiles", rnd, fn)
            if not os.path.exists(path):
                continue
            rows = list(csv.DictReader(open(path)))
            best, how = select_kernel_row(rows, kernel, batch, launch_us)
            if best is None:                                   # (no name from the library: the row with the largest total, as before)
                best = max(rows, key=lambda r: float(r.get("TotalDurationNs", 0) or 0)) if rows else None
                how = "the row with the largest total duration (" + how + ")"
            if best:
                return {"file": f"profiles/{rnd}/{fn}", "kernel": best.get("Name", "")[:96], "selected": how,
                        "calls": int(float(best.get("Calls", 0) or 0)), "average_us": float(best.get("AverageNs", 0) or 0) / 1e3}
    return None


def build_record(lib):
    """What was measured: the library's version string, the commit and flags __graft_entry__.build() recorded beside the
    library (zang_amd/build_info.json -- .git does not travel to the GPU box), whether an alternative library was forced."""
    rec = {"zh_version": lib.zh_version().decode(), "build_mode": "in-tree make (zang_amd/csrc/Makefile: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize)"}
    if os.environ.get("ZANG_HIP_LIB"):
        rec["build_mode"] = "ZANG_HIP_LIB=" + os.environ["ZANG_HIP_LIB"] + " (an alternative build forced through the environment)"
    info = os.path.join(ROOT, "zang_amd", "build_info.json")
    if os.path.exists(info):
        try:
            rec.update(json.load(open(info)))
        except ValueError:
            pass
    switches = sorted(k for k in os.environ if k.startswith("ZH_") and k not in ("ZH_BENCH_EMULATE",))
    if switches:
        rec["switches_in_environment"] = {k: os.environ[k] for k in switches}
    return rec


def dry_run(args, world, rank):
    """ZH_BENCH_EMULATE=1 on a box WITHOUT a GPU: the launcher, the rendezvous and the exchange step with host tensors
    (gloo).  Nothing is painted and nothing is measured: `value` is 0 and the line says so.  Used by the CPU tests."""
    import torch
    import torch.distributed as dist
    from zang_amd import sharding
    dist.init_process_group("gloo")
    F, Cn = args.frames, args.channels
    mixes = torch.full((48, Cn, F), float(rank + 1), dtype=torch.float32)
    t0 = time.perf_counter()
    sharding.allreduce_mix(mixes)
    dt = time.perf_counter() - t0
    ok = bool((mixes == world * (world + 1) / 2).all())
    seen = torch.ones(1); dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "voice-samples/sec", "value": 0.0, "unit": "voice-samples/s", "n_gpus": world,
                          "steps": pattern_steps(args.workload, args.steps), "steps_requested": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "none", "dry_run": "no GPU on this box: launcher + rendezvous + exchange step only, nothing painted",
                          "config": {"workload": f"{args.workload}: dry run", "parallelism": f"voices sharded x{world}"},
                          "collective": {"backend": "gloo", "world_size_seen": int(seen.item()), "bytes": mixes.numel() * 4,
                                         "reduce_us": dt * 1e6, "sum_correct": ok}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.preflight and env_world is None and (args.gpus or 1) > 1:
        # parent: comm_host first (its own process tree), then the rank processes for the IPC round trips
        emu = os.environ.get("ZH_BENCH_EMULATE") == "1"
        n = args.gpus
        if emu:
            import torch
            n = max(1, min(n, torch.cuda.device_count()))        # counting devices does not initialise the GPU
        rec = preflight_comm_host(n)
        sys.stderr.write("bench.py --preflight: comm_host %d: %s\n" % (n, "PASS" if rec["ok"] else "FAIL " + rec["output"][-300:]))
        os.environ["ZH_PREFLIGHT_COMM_DONE"] = "1"
        rc = spawn_ranks(args.gpus, argv)
        sys.exit(rc or (0 if rec["ok"] else PREFLIGHT_FAILED))
    if env_world is None and (args.gpus or 1) > 1:
        sys.exit(spawn_ranks(args.gpus, argv))       # parent: nothing below runs in it
    world = int(env_world or 1)
    # ZH_BENCH_ONE_RANK_DIST=1 (under torch.distributed.run with ONE process): this rank takes the N > 1 code path as it is --
    # torch.distributed's nccl (= RCCL) process group and the gloo control group, the library's own RCCL communicator, the
    # exchange inside the timed region, the paired regions -- with a world of one.  What a one-GPU box can run of the path the
    # driver's scaling run takes (tests/test_bench_launcher.py); ZH_BENCH_EMULATE is the other half (two ranks, gloo).
    dist_on = world > 1 or os.environ.get("ZH_BENCH_ONE_RANK_DIST") == "1"
    if args.gpus is not None and args.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         f"or under torch.distributed.run with --nproc-per-node equal to --gpus\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.preflight:
        sys.exit(preflight_rank(args, world, rank, local_rank, os.environ.get("ZH_BENCH_EMULATE") == "1"))
    if args.workload is None:
        args.workload = "nice_mix" if dist_on else "pulseosc"
    if args.voices is None:
        args.voices = 131072 if (args.workload == "nice_mix" and dist_on) else 4096
    if args.steps is None:
        args.steps = 960 if args.workload in ("nice", "nice_mix", "script") else 1000
    import torch
    import torch.distributed as dist
    # ZH_BENCH_EMULATE=1: multi-process dry run on a single GPU (every rank on device 0, gloo instead
    # of RCCL) -- used only to exercise the N>1 code path where one GPU is available.
    emulate = os.environ.get("ZH_BENCH_EMULATE") == "1"
    if emulate and world > 1 and not torch.cuda.is_available():
        sys.exit(dry_run(args, world, rank))
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no HIP device (torch.cuda.is_available() is False); there is no CPU path to measure\n")
        sys.exit(3)
    device_index = 0 if (world == 1 or emulate) else local_rank
    if device_index >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} needs GPU {device_index} but this node shows {torch.cuda.device_count()}\n")
        sys.exit(3)
    ctl = None
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(device_index)
        if emulate:
            dist.init_process_group("gloo")
            ctl = dist.group.WORLD
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
            ctl = dist.new_group(backend="gloo")        # host-side control channel (handle exchange, host barriers)
    else:
        torch.cuda.set_device(0)
    import zang_amd
    from zang_amd import abi
    # a side stream (graph capture is not allowed on the default stream); every torch
    # allocation / copy / sync below happens with it current, so ordering is by stream
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    ctx = zang_amd.Context(device_index)
    V, F = args.voices, args.frames
    lib = ctx.lib
    # --exchange rccl: the library's own communicator (zh_comm_*, librccl opened by libzang_hip.so); torch.distributed only
    # carries rank 0's 128-byte id.  Two ranks emulated on one device cannot form an RCCL communicator: gloo stays there.
    comm, comm_note = None, None
    if dist_on and args.exchange == "rccl" and args.workload == "nice_mix":
        if emulate:
            comm_note = "ZH_BENCH_EMULATE: ranks share one device, no RCCL communicator; torch.distributed gloo instead"
        else:
            from zang_amd import sharding
            try:
                lib.zh_comm_set_timeout(90.0)            # the rendezvous is bounded (180 s by default): a bench run fails over sooner
                comm = sharding.Comm(ctx, control_group=ctl)
            except Exception as e:      # noqa: BLE001  (every rank raises together: Comm agrees on availability first)
                comm_note = f"zh_comm unavailable ({type(e).__name__}: {e}); torch.distributed nccl instead"[:300]
                sys.stderr.write(f"bench.py rank {rank}: {comm_note}\n")

    def barrier():
        if dist_on:
            dist.barrier()

    # torch.cuda.synchronize() without its Python wrapper (lazy-init check + a device context manager around the same call)
    _device_sync = getattr(torch._C, "_cuda_synchronize", None) or torch.cuda.synchronize

    def make_event():
        h = C.c_void_p()
        abi.check(lib.zh_event_create(ctx.handle, C.byref(h)), "zh_event_create")
        return h

    def max_over_ranks(x):
        if dist_on:
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if emulate else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    class Runner:
        """One workload prepared for timing: eager pre-pass (lazy allocations), the captured graph of G steps, and
        `region(K)` = exactly K steps bracketed by barrier + synchronize on both sides, timed by the wall clock
        (max over ranks) and by HIP events on the launch stream."""

        def __init__(self, name, voices, steps, exchange="rccl", slots=False, coalesce=None, tolerant=None):
            self.wl = Workload(name, ctx, voices, F, first_voice=rank * voices, ring_bytes=args.ring_mib << 20, world=world, multi=dist_on,
                               pad=args.pad_voices, channels=args.channels, exchange=exchange, tolerant=args.tolerant if tolerant is None else tolerant)
            wl = self.wl
            wl.comm = comm if exchange == "rccl" else None
            self.with_exchange = True
            if slots:
                from zang_amd import sharding
                wl.use_slots(sharding.SlotExchange(ctx, wl.mixes.numel(), control_group=ctl))
            # The per-buffer loop is launch-bound at 4096 voices (a 16 MiB paint takes ~4 us on the device, a
            # Python->C->hipLaunchKernel call about as long), so G consecutive steps are captured once into a
            # hipGraph and replayed.
            self.G = wl.graph_steps(steps) if not args.eager else 0
            self.graph = None
            self.graph_kernels = []
            self.graph_nodes, self.graph_held, self.graph_launches = 0, 0, 0
            if self.G:
                for _ in range(self.G):          # one eager pass first: lazy allocations happen outside capture
                    wl.step()
                torch.cuda.synchronize()
                # which kernels the step's (last) paint launched: the library's answer (zh_last_form), not a guess from the voice
                # count; the dominant one: pass B of a two-kernel time-parallel form, else the first
                wl.kernels_launched = ctx.last_form()
                if wl.kernels_launched:
                    tp_b = [k for k in wl.kernels_launched if k.endswith("_tp_b")]
                    wl.kernel = (tp_b or wl.kernels_launched)[0]
                if hasattr(wl, "nsteps"):
                    wl.nsteps = 0               # the graph holds buffers 0..G-1 of the note pattern
                wl.batch_rows = min(self.G, 48)
                # pulseosc: ZH_CAPTURE_COALESCE -- the steps' paints (params unchanged: the phase at any frame is the entry counter +
                # frames * ifreq exactly, PulseOsc.zig:111) are held back while recording and become one launch per <= 32 buffers.  The
                # step is still one zero+paint CALL per buffer; what the graph replays is fewer, larger launches (ZH_BENCH_IN_ORDER=1: one kernel node per step, the round-4 form)
                # noise_filter_fused --tolerant: under the same flag consecutive paints are recorded PIPELINED (pass A of buffer n + 1 beside pass B of
                # buffer n, on the context's side stream: csrc/composite.hip zh_noise_filter_paint); nothing is held back, so graph_held stays 0
                # nice_mix (stereo): the same flag holds back zh_nice_paint_mix_stereo calls -- up to 8 consecutive buffers per launch, the launch
                # zh_nice_paint_mix_stereo_batch makes (state words in registers from buffer to buffer, one second pass; same bits)
                wl.graph_first_image = wl.i                   # buffer j of a replay lands in ring[(this + j) % nring]
                self.graph = ctx.capture(lambda: [wl.step() for _ in range(self.G)], coalesce=(((name == "pulseosc" or (name == "nice_mix" and args.channels == 2) or (name == "noise_filter_fused" and self.wl.tolerant)) and os.environ.get("ZH_BENCH_IN_ORDER") != "1") if coalesce is None else coalesce))
                self.graph_nodes, self.graph_held, self.graph_launches = self.graph.info()
                # what a replay RUNS: the kernels launched while recording (zh_graph_kernels) -- a paint that was held back went out
                # later, in another shape, than the eager pass above showed (the line named k_nf_tp_b for k_nf_tp_ba launches)
                self.graph_kernels = self.graph.kernels()
                if self.graph_kernels:
                    wl.kernels_launched = ["%s x%d" % kn for kn in self.graph_kernels]
                    dom = max(self.graph_kernels, key=lambda kn: (kn[0].endswith("_tp_ba"), kn[1] * (3 if "[batch]" in kn[0] else 1)))
                    wl.kernel, wl.kernel_batch = dom[0].split("[")[0], "[batch]" in dom[0]
            # (event records captured INTO the graph would take two host calls off the timed path, but hipEventElapsedTime
            # refuses events recorded by graph nodes on this ROCm: "invalid resource handle", profiles/r04/probe_region.txt)
            self.ev0, self.ev1 = make_event(), make_event()

        def run_steps(self, n):
            wl, G, done = self.wl, self.G, 0
            if self.graph is not None:
                while n - done >= G:
                    self.graph.launch()
                    if self.with_exchange:
                        wl.exchange()
                    done += G
            for _ in range(n - done):
                wl.step()
            if n - done and self.with_exchange:
                wl.exchange()

        def warm(self, n):
            self.run_steps(n)
            if self.graph is not None and n < self.G:
                self.graph.launch()                      # untimed: the first replay of a graph uploads it
                if self.with_exchange:
                    self.wl.exchange()
            torch.cuda.synchronize()

        def region(self, K):
            sync = _device_sync
            if self.wl.comm is not None:
                self.wl.comm.check()                  # zh_comm_check: an asynchronous RCCL error (a dead peer) ends the run here, not in a hang
            sync()
            barrier()
            sync()
            one_launch = self.graph is not None and K == self.G and not dist_on
            rec, h, e0, e1 = lib.zh_event_record, ctx.handle, self.ev0, self.ev1
            launch, gh = lib.zh_graph_launch, (self.graph.handle if self.graph is not None else None)
            # HIP events on the launch stream bracket exactly the K timed steps: device time per
            # launch = elapsed / K (includes the ~1 us inter-kernel boundaries, so it can only
            # under-state the kernel's own rate; profiles/ holds the rocprofv3 per-kernel average).
            if one_launch:
                # the driver's short region: K steps = one graph replay.  Nothing but the four calls between the two clock
                # reads (their return codes are looked at after the clock stops): the interpreter's own work on the timed
                # path was 4 us of a 100 us region (tools/probe_region.py, profiles/r04/probe_region.txt)
                t0 = time.perf_counter()
                r0 = rec(h, e0)
                r1 = launch(h, gh)
                r2 = rec(h, e1)
                sync()
                elapsed = time.perf_counter() - t0
                for r, what in ((r0, "zh_event_record"), (r1, "zh_graph_launch"), (r2, "zh_event_record")):
                    abi.check(r, what)
            else:
                t0 = time.perf_counter()
                abi.check(rec(h, e0), "zh_event_record")
                self.run_steps(K)
                abi.check(rec(h, e1), "zh_event_record")
                sync()
                barrier()
                if dist_on:
                    sync()
                elapsed = time.perf_counter() - t0
            ms = C.c_float()
            abi.check(lib.zh_event_elapsed_ms(self.ev0, self.ev1, C.byref(ms)), "zh_event_elapsed_ms")
            return max_over_ranks(elapsed), ms.value

        def close(self):
            lib.zh_event_destroy(self.ev0)
            lib.zh_event_destroy(self.ev1)
            if self.wl.slots is not None:
                self.wl.slots.close()

    def time_batched(run, nsteps):
        """The same `nsteps` buffers of a stereo nice_mix Runner painted B per launch (extra key only: a step of `value` stays
        one buffer per launch)."""
        w = run.wl
        if w.name != "nice_mix" or w.channels != 2 or args.eager or w.slots is not None:
            return None
        B = max(b for b in (8, 6, 5, 4, 3, 2, 1) if nsteps % b == 0)
        if B == 1:
            return None
        def steps():
            for _ in range(0, nsteps, B):
                w.step_batch(B)
        w.nsteps = 0
        steps(); torch.cuda.synchronize()
        w.nsteps = 0
        g = ctx.capture(steps)
        g.launch(); torch.cuda.synchronize()
        e0, e1 = make_event(), make_event()
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        abi.check(lib.zh_event_record(ctx.handle, e0), "zh_event_record")
        g.launch()
        abi.check(lib.zh_event_record(ctx.handle, e1), "zh_event_record")
        torch.cuda.synchronize()
        wall = max_over_ranks(time.perf_counter() - t0)
        ms = C.c_float()
        abi.check(lib.zh_event_elapsed_ms(e0, e1, C.byref(ms)), "zh_event_elapsed_ms")
        lib.zh_event_destroy(e0); lib.zh_event_destroy(e1)
        g.close()
        return {"buffers_per_launch": B, "buffers": nsteps, "ms_per_buffer_wall": wall / nsteps * 1e3, "ms_per_buffer_hip_events": ms.value / nsteps,
                "value_per_gpu": w.V * F * nsteps / wall,
                "what": "zh_nice_paint_mix_stereo_batch: %d consecutive buffers per launch, state in registers between them, one second pass (same bits; no exchange)" % B}

    K_req = args.steps
    K = pattern_steps(args.workload, K_req)          # the note-pattern workloads time whole 48-buffer patterns
    mixdown = args.workload == "nice_mix"
    main_run = Runner(args.workload, V, K, exchange=args.exchange, slots=(args.exchange == "p2p" and dist_on and mixdown))
    wl, G, graph = main_run.wl, main_run.G, main_run.graph
    main_run.warm(args.warmup)
    # Untimed rehearsals of the timed region, same code path: the W warm-up steps of a 4 us kernel are 20 us of
    # device work after seconds of set-up, and the first region then ran on a GPU and a host path still coming out of idle -- 7 %
    # slower than the median of the repeats, 111 against 98 us after 200 ms of idle in tools/probe_region.py.  At least 3 regions
    # and 25 ms; nothing from them enters the line except their count.
    def rehearse(run, k):
        if args.no_rehearsal:
            return 0
        e3 = sum(run.region(k)[0] for _ in range(3))               # max over ranks: the same number on every rank,
        n = 3 + min(997, max(0, int((0.025 - e3) / (e3 / 3))))      # so every rank runs the same count of regions
        for _ in range(n - 3):
            run.region(k)
        return n
    rehearsals = rehearse(main_run, K)
    elapsed, ev_ms = main_run.region(K)              # THE timed region of the contract: exactly K steps
    step_ms_events = ev_ms / K

    # Further regions of exactly K steps, each bracketed the same way: a short region (the driver's 20 steps of a
    # 4 us kernel = 0.1 ms) is dominated by the launch ramp and the synchronize, and one sample of it is noisy.
    # With several GPUs every repeat is a PAIR -- the region with the exchange, then the same K steps without it -- so that
    # `single_gpu_shard` and `scaling_factor` are medians of interleaved regions, never one sample each.
    import statistics
    paired = dist_on and mixdown
    R = args.repeats if args.repeats is not None else (min(30, max(0, int(0.5 / max(elapsed, 1e-4)))) if elapsed < 0.1 else 0)
    if paired:
        R = max(R, 5)
    reps = None
    walls, evs, walls_nox = [], [], []
    for _ in range(R):
        e, m = main_run.region(K)
        walls.append(e / K * 1e3); evs.append(m / K)
        if paired:
            main_run.with_exchange = False
            e, _ = main_run.region(K)
            main_run.with_exchange = True
            walls_nox.append(e / K * 1e3)

    def spread(xs):
        return {"median": statistics.median(xs), "min": min(xs), "max": max(xs)}
    if R > 0:
        reps = {"regions": R, "steps_per_region": K, "ms_per_step_wall": spread(walls), "ms_per_step_hip_events": spread(evs),
                "ms_per_step_wall_all": [round(w, 6) for w in walls],
                "note": "`value`, `ms_per_step` and `roofline` come from the first region only"}

    total_units = world * V * F * K
    value = total_units / elapsed
    achieved = wl.bytes_per_step / (step_ms_events * 1e-3) / 1e9
    # kernel launches of the dominant kernel inside the timed region (a coalesced graph replays fewer, larger launches)
    launches = K
    pipelined = graph is not None and main_run.graph_held > 0 and args.workload == "noise_filter_fused"     # (one fused launch per buffer + the last pass B)
    if graph is not None and main_run.graph_held and K % G == 0 and not pipelined:
        launches = (K // G) * main_run.graph_launches
    batch = bool(getattr(wl, "kernel_batch", False))
    driver_form = (K_req, args.warmup) == (20, 5)
    traffic, traffic_src = traffic_record(args, V, F, wl.kernel, batch=batch, driver_form=driver_form, pipelined=pipelined)
    if traffic is not None and launches != K:
        per = traffic_src.get("buffers_per_counted_launch", 1)
        if per != K / launches:
            # counted on launches of `per` buffers; this run's launches paint K / launches each (the 114 KiB constants table and the
            # counters are read once per launch: scaling by buffers is an upper bound by < 1 %)
            traffic = traffic * (K / launches) / per
            traffic_src["note"] += "; counted per launch of %g buffer(s) and scaled to the %g buffers of this run's launches" % (per, K / launches)
    rp = rocprof_record(args, V, kernel=wl.kernel, batch=batch, launch_us=ev_ms / launches * 1e3, driver_form=driver_form)
    out = {
        "metric": "voice-samples/sec", "value": value, "unit": "voice-samples/s",
        # what one step of `value` is on the device (the form changed in round 5 under the same name): rounds 1-4 = "one launch per step"
        "value_form": ("eager: one launch per step" if graph is None else
                       "hipGraph, ZH_CAPTURE_COALESCE: %g buffers per launch" % (K / launches) if launches != K else
                       "hipGraph, ZH_CAPTURE_COALESCE: pass B of buffer n + pass A of buffer n + 1 per launch" if pipelined else "hipGraph: one kernel node per step"),
        "n_gpus": world, "steps": K, "warmup": args.warmup, "rehearsal_regions": rehearsals,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {V} voices/GPU x {F} frames, " +
                               (f"fused Osc+Env+Filter voices + {args.channels}-channel voice mixdown per {F}-frame buffer, 48 kHz (BASELINE configs[4])" if mixdown
                                else f"zero+paint per {F}-frame buffer, 48 kHz"),
                   "voices_per_gpu": V, "total_voices": V * world, "frames": F, "ring_images": wl.nring,
                   "launch": "eager" if graph is None else (f"hipGraph x{G} steps" + (
                       f" recorded with ZH_CAPTURE_COALESCE: the {main_run.graph_held} paint calls became {main_run.graph_launches} kernel launches of up to " +
                       ("8 consecutive buffers each (k_nice_mix_batch: the voices' state stays in registers from buffer to buffer)" if mixdown else
                        "32 buffers each (grid.z; an even number, so that a replay ends on the counter buffer it began on)") +
                       f"; {main_run.graph_nodes} nodes" if (main_run.graph_held and not pipelined) else
                       (f" recorded with ZH_CAPTURE_COALESCE, pipelined: pass B of buffer n and pass A of buffer n + 1 in one launch (k_nf_tp_ba), "
                        f"{main_run.graph_launches} launches for the {main_run.graph_held} paint calls" if pipelined else ""))),
                   "graph_nodes": main_run.graph_nodes,
                   "parallelism": f"voices sharded x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": wl.kernel, "kernels_launched_per_step": wl.kernels_launched,
                     "frac_of_measured_store_rate": achieved / HBM_STORE_GBS,   # also quote / 6200 "achievable"
                     "algorithmic_bytes_per_launch": wl.bytes_per_step * K / launches, "launch_ms_hip_events": ev_ms / launches,
                     "launches_in_region": launches, "buffers_per_launch": K / launches,
                     # the same bytes over the rocprofv3 per-kernel average of the instantiation that ran (no inter-launch gaps): the kernel's own rate
                     "frac_kernel": (wl.bytes_per_step * K / launches / (rp["average_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp and rp.get("average_us") else None,
                     "rocprofv3_kernel_average": rp},
        "equiv_write_GBs_whole_job": value * 4 / 1e9,
    }
    if args.tolerant:
        out["config"]["tolerant"] = ("ZH_PAINT_TOLERANT: the sines that reach the output through scaling and adding alone in f32 (not bit-exact)" if args.workload == "script"
                                     else "ZH_PAINT_TOLERANT: the mixdown kernel with multiply-adds fused (k_nice_mix_fma), samples within 1e-5 of the voice's peak (not bit-exact)"
                                     if (mixdown and "k_nice_mix_fma" in (wl.kernels_launched or [])) else
                                     "ZH_PAINT_TOLERANT: the Filter as chunks at once, samples within 1e-5 of the voice's peak (not bit-exact)")
    if K != K_req:
        out["steps_requested"] = K_req
        out["config"]["pattern"] = (f"--steps {K_req} -> {K} timed steps: the note pattern is {PATTERN} buffers (note on 0-23: attack, decay, "
                                    f"sustain; off 24-47: release, idle) and a timed region is whole patterns, buffers 0..{PATTERN - 1} in order; "
                                    "`steps`, `ms_per_step` and `value` are of the steps really timed")
    if mixdown:
        out["config"]["channels"] = args.channels
        out["roofline"]["note"] = ("the mixdown workload writes 4 KiB per channel per buffer: `achieved` is the EQUIVALENT write rate "
                                   "(4 B per voice-sample that the unfused path would store); the kernel is VALU-issue-bound (SURVEY.md 8d)")
    if reps:
        out["repeats"] = reps

    if dist_on and mixdown:
        # ---- the exchange step on its own, and the same shard without it (scaling factor) ----
        rows = wl.batch_rows
        nbytes = wl.mixes[:rows].numel() * 4
        n_ex = 24

        def time_exchange(fn):
            """`n_ex` samples of one exchange each (stream drained and ranks aligned before every sample, slowest rank per
            sample): median / min / max in microseconds -- a single figure hid a first-call outlier in round 3."""
            fn(); torch.cuda.synchronize()
            per = []
            for _ in range(n_ex):
                torch.cuda.synchronize(); barrier()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                per.append(time.perf_counter() - t0)
            if dist_on:
                t = torch.tensor(per, dtype=torch.float64, device="cpu" if emulate else "cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                per = t.tolist()
            per = [x * 1e6 for x in per]
            return {"median": statistics.median(per), "min": min(per), "max": max(per), "samples": n_ex}
        reduce_stat = time_exchange(wl.exchange)
        reduce_us = reduce_stat["median"]
        seen = torch.ones(1, device="cpu" if emulate else "cuda"); dist.all_reduce(seen)
        if wl.slots is not None:
            backend, kind = "host barriers + HIP IPC", "direct stores into the root's slots + rank-ordered sum (zh_sum_slots), host barriers"
        elif wl.comm is not None:
            backend = "rccl %d via zh_allreduce_mix (libzang_hip.so opened %s)" % (lib.zh_comm_version(), lib.zh_comm_library().decode())
            kind = "all_reduce(sum) on the launch stream (C ABI zh_comm_*; torch.distributed carried the 128-byte id only)"
        else:
            backend, kind = "torch.distributed " + dist.get_backend(), "all_reduce(sum)"
        out["collective"] = {"backend": backend, "kind": kind, "world_size_seen": int(seen.item()),
                             "bytes": nbytes, "per": f"{rows}-buffer batch", "reduce_us": reduce_us, "reduce_us_spread": reduce_stat,
                             "in_timed_region": True, "reduce_us_per_buffer": reduce_us / rows}
        if comm_note:
            out["collective"]["note"] = comm_note
        if wl.slots is None:
            # the same sum as ONE COLLECTIVE PER BUFFER (4-8 KiB each, the reference's per-buffer mix) -- which
            # granularity is latency-bound: `rows` small collectives against one of rows x the bytes
            pb = time_exchange(wl.exchange_per_buffer)
            per_buf_us = pb["median"] / rows
            out["collective"]["per_buffer_form"] = {"bytes": nbytes // rows, "reduce_us": per_buf_us,
                                                    "reduce_us_spread": {k: (v / rows if k != "samples" else v) for k, v in pb.items()},
                                                    "x_batch_form_per_buffer": per_buf_us / (reduce_us / rows),
                                                    "what": f"one all-reduce per buffer, {rows} back to back on the launch stream, per collective"}
            if wl.comm is not None:
                # and the torch.distributed form of the batch collective beside the library's
                keep = wl.comm
                wl.comm = None
                out["collective"]["torch_distributed_form_reduce_us"] = time_exchange(wl.exchange)["median"]
                wl.comm = keep
        # medians of the R interleaved region pairs above (with the exchange / without it), slowest rank per region
        med_with, med_nox = statistics.median(walls), statistics.median(walls_nox)
        single = V * F / (med_nox * 1e-3)
        out["single_gpu_shard"] = {"value": single, "ms_per_step": med_nox, "ms_per_step_spread": spread(walls_nox), "regions": R,
                                   "what": "the same K steps on every rank without the exchange (slowest rank), median of the regions "
                                           "interleaved with the `repeats` regions"}
        ratios = sorted(world * b / a for a, b in zip(walls, walls_nox))
        out["scaling_factor"] = world * med_nox / med_with
        out["scaling_factor_spread"] = {"min": ratios[0], "max": ratios[-1], "pairs": R,
                                        "what": "world x (ms without exchange) / (ms with exchange), pair by pair; `scaling_factor` uses the two medians"}
        out["value_median_of_repeats"] = world * V * F / (med_with * 1e-3)
        out["realtime_voices_48k"] = value / SR
        if args.p2p_compare and wl.slots is None:
            # the alternative SURVEY.md 8e asks to measure beside the collective.  It is the comparison, not the
            # measurement, and opt-in (--p2p-compare): if it stalls (a peer mapping or a host barrier that never returns on some
            # machine), a watchdog on EVERY rank ends the run after `--p2p-timeout` seconds with exit code 4 -- rank 0 prints the
            # line it already has first.
            import threading

            def give_up():
                if rank == 0:
                    out["p2p_direct"] = {"error": f"no result within {args.p2p_timeout:.0f} s; the run was ended by the watchdog"}
                    print(json.dumps(out), flush=True)
                os._exit(4)                      # distinct from a crash: the line is complete, the opt-in comparison stalled
            dog = threading.Timer(args.p2p_timeout, give_up)
            dog.daemon = True
            dog.start()
            try:
                alt = Runner("nice_mix", V, K, exchange="p2p", slots=True)
                alt.warm(48)
                e2, _ = alt.region(K)
                torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_ex):
                    alt.wl.exchange()
                torch.cuda.synchronize()
                p2p_us = max_over_ranks((time.perf_counter() - t0) / n_ex) * 1e6
                out["p2p_direct"] = {"value": world * V * F * K / e2, "ms_per_step": e2 / K * 1e3, "reduce_us": p2p_us, "bytes": nbytes,
                                     "what": "mixdown kernels store into the root GPU's per-rank slots (HIP IPC mapping); per batch: stream sync + host "
                                             "barrier, root adds the slots in rank order, host barrier; bit-reproducible"}
                alt.close()
            except Exception as e:          # noqa: BLE001  (reported, never fatal)
                out["p2p_direct"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            dog.cancel()

    if mixdown:
        bt = time_batched(main_run, G if G else 48)
        if bt:
            out["batched_launches"] = bt

    if world == 1 and args.workload == "pulseosc" and not args.eager and main_run.graph_held:
        # The same K steps recorded WITHOUT ZH_CAPTURE_COALESCE: one kernel node per step, each a launch of one 16 MiB buffer that
        # hands its phase counters to the next (the round-4 headline form).  Extra key, measured the same way as `value`.
        io = Runner("pulseosc", V, K, coalesce=False)
        io.warm(args.warmup)
        rehearse(io, K)
        io_regs = [io.region(K) for _ in range(1 + min(R, 10))]
        e_io, m_io = io_regs[0]
        out["one_launch_per_step"] = {"value": V * F * K / e_io, "ms_per_step": e_io / K * 1e3, "launch_ms_hip_events": m_io / K,
                                      "frac": wl.bytes_per_step / (m_io / K * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "ms_per_step_wall": spread([e / K * 1e3 for e, _ in io_regs]), "regions": len(io_regs),
                                      "launch": f"hipGraph x{io.G} steps", "graph_nodes": io.graph_nodes,
                                      "what": "the same steps recorded without ZH_CAPTURE_COALESCE (`value` of rounds 1-4 was this form)"}
        io.close()

    if world == 1 and not args.no_config5 and args.workload == "pulseosc" and V == 4096:
        # the 1-GPU shard of config 5 (what every rank of the N>1 run renders), for the scaling ratio
        c5 = Runner("nice_mix", 131072, 96)
        c5.warm(48)
        c5_rehearsals = rehearse(c5, 96)
        e5s = [c5.region(96) for _ in range(5)]
        e5, m5 = e5s[0]
        out["config5_shard"] = {"workload": f"nice_mix: 131072 voices x {F} frames, {args.channels}-channel mixdown, no exchange (one GPU)",
                                "value": 131072 * F * 96 / e5, "ms_per_step": e5 / 96 * 1e3, "steps": 96, "launch_ms_hip_events": m5 / 96,
                                "pattern": "two whole 48-buffer note patterns per region (attack, decay, sustain, release, idle)",
                                "ms_per_step_5_regions": spread([e / 96 * 1e3 for e, _ in e5s]), "rehearsal_regions": c5_rehearsals,
                                "realtime_voices_48k": 131072 * F * 96 / e5 / SR}
        bt = time_batched(c5, 96)
        if bt:
            out["config5_shard"]["batched_launches"] = bt
        if not args.no_parity:
            out["config5_shard"]["parity"] = parity_check_mix(c5.wl, ctx)
        c5.close()
        if not args.tolerant and args.channels == 2:
            # beside it (extra key only): the same shard painted with ZH_PAINT_TOLERANT -- above nice_tp_max voices that is the same kernel
            # compiled with multiply-adds fused (csrc/nice_mix_fma.hip): fewer instructions on an issue-bound kernel, not the reference's bits
            c5t = Runner("nice_mix", 131072, 96, tolerant=True)
            c5t.warm(48)
            rehearse(c5t, 96)
            e5t = [c5t.region(96) for _ in range(3)]
            out["config5_shard"]["tolerant_form"] = {"flag": "ZH_PAINT_TOLERANT", "kernels": c5t.wl.kernels_launched, "value": 131072 * F * 96 / e5t[0][0],
                                                     "ms_per_step": e5t[0][0] / 96 * 1e3, "launch_ms_hip_events": e5t[0][1] / 96,
                                                     "ms_per_step_3_regions": spread([e / 96 * 1e3 for e, _ in e5t])}
            if not args.no_parity:
                out["config5_shard"]["tolerant_form"]["parity"] = parity_check_mix(c5t.wl, ctx)
            c5t.close()

    if world == 1 and not args.no_config4 and not args.no_config5 and args.workload == "pulseosc" and V == 4096:
        try:
            out["config4"] = config4_record(ctx, with_oracle=not args.no_cpu)
        except Exception as e:          # noqa: BLE001  (an extra key: reported, never fatal)
            out["config4"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0 and not args.no_parity and (world == 1 or mixdown):
        # (N > 1: rank 0's shard of the mixdown workload; local launches only, the other ranks wait at the last barrier)
        out["parity"] = parity_check_mix(wl, ctx) if mixdown else parity_check(wl, ctx)
        if world == 1 and graph is not None and not mixdown:
            # ... and the LAST buffer of one more replay of the timed graph itself -- the coalesced / pipelined launches the line is about
            pg = parity_check_graph(main_run, ctx)
            if pg:
                out["parity_of_a_graph_replay"] = pg
    if rank == 0 and world == 1 and not args.no_cpu:
        cb = cpu_baseline(args, wl)
        if cb:
            out["cpu_baseline"] = cb
    # whoever reads SCALE first: the N = 1 line is config 2, the N > 1 lines are config 5 -- their ratio is not a scaling figure
    if world == 1 and "config5_shard" in out:
        out["scaling_anchor"] = ("config5_shard.value of THIS line (one GPU's shard of config 5, no exchange) is what the N > 1 lines' `value` / N compares with; "
                                 "`value` here is config 2 (4,096 PulseOsc voices), a different workload: value(N) / value(1) is meaningless")
    elif dist_on and mixdown:
        out["scaling_anchor"] = ("this line's own `scaling_factor` (world x time without the exchange / time with it) and `single_gpu_shard`, or the N = 1 line's "
                                 "`config5_shard.value`; the N = 1 line's `value` is config 2, a different workload")
    main_run.close()
    if comm is not None:
        comm.close()
    out["build"] = build_record(lib)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
