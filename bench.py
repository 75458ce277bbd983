#!/usr/bin/env python3
"""bench.py -- the BASELINE.json headline metric: voice-samples/sec of the paint() hot path.

One "step" = what the reference does per 1024-frame buffer for every voice of the
workload: `zang.zero(span, out)` then `Module.paint(span, {out}, ...)` (e.g.
examples/modules.zig:220-225), with state carried from buffer to buffer.

Default workload (N=1): BASELINE.json configs[1] -- 4096 PulseOsc voices x 1024 frames,
constant per-voice frequency, 48 kHz.  With --gpus N each rank renders its own 4096-voice
shard (weak scaling; voices are independent, no collective on this path: per-voice output
images stay resident on the GPU that painted them).

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


SCRIPT_MODULE = os.environ.get("ZH_SCRIPT_MODULE", "Lead")    # any module of tests/golden/script_modules.txt with (freq, note_on) params


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="pulseosc", choices=["pulseosc", "noise_filter", "noise_filter_fused", "nice", "nice_mix", "script"])
    ap.add_argument("--voices", type=int, default=4096, help="voices per GPU")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--ring-mib", type=int, default=512, help="bytes of distinct output images to rotate through")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU-baseline sample length")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-run oracle comparison of one buffer")
    ap.add_argument("--eager", action="store_true", help="one host launch per step instead of hipGraph replay")
    ap.add_argument("--pad-voices", type=int, default=None, help="row padding of the output images in voices (default: the library's choice, Context.image)")
    return ap.parse_args()


class Workload:
    """Builds the module(s), resident params and the per-step callable for one rank."""

    def __init__(self, name, ctx, V, F, first_voice, ring_bytes, world=1, pad=None):
        import torch
        self.world = world
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
            self.kernel = "k_filter"
            self.step = self._step_noise_filter
        elif name == "noise_filter_fused":
            self.m = mod.NoiseFilter(V, ctx, first_seed=first_voice)
            cutoff_f = torch.from_numpy((200.0 + 7800.0 * u2)).to(dev)
            self.cutoff = mod.Filter.cutoffFromFrequency(cutoff_f, SR, ctx)
            self.res = torch.from_numpy((0.9 * u3)).to(dev)
            self.ring = [ctx.image(F, V, pad=pad) for _ in range(nring)]
            self.params = self.m.Params(self.m_white(), mod.Filter.low_pass, self.cutoff, self.res)
            self.kernel = "k_noise_filter_pc" if V <= 65536 else "k_noise_filter"     # the library's choice by voice count
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
            self.kernel = "k_nice_pc" if V <= 65536 else "k_nice"
            self.step = self._step_nice
            self.nsteps = 0
        else:
            self.m = mod.NiceInstrument(V, self.color, ctx)
            # one [frames] partial mix per buffer of a 48-buffer batch; with several GPUs the batch is exchanged at once
            self.mixes = torch.zeros((48, F), dtype=torch.float32, device=dev)
            self.mix = self.mixes[0]
            self.ring = []
            self.kernel = "k_nice_mix"
            self.step = self._step_nice_mix
            self.nsteps = 0
        self.nring = len(self.ring)
        self.i = 0
        self.steps_done = 0

    def graph_steps(self, steps=0):
        """Steps per captured graph: at least one ring rotation, even (see bench main).  The K timed steps as ONE
        graph when K is even and not huge (each graph launch costs a ~4 us bubble on the stream); otherwise an even
        count between one and four rotations that divides K, so that no remainder of the timed steps has to be
        launched one by one from Python (~2x slower per step at the 16 MiB size)."""
        if self.name in ("nice", "nice_mix", "script"):
            return 48                       # the note on/off pattern repeats every 48 buffers
        if os.environ.get("ZH_BENCH_G"):
            return int(os.environ["ZH_BENCH_G"])   # experiments
        g = max(self.nring, 2)
        g = g if g % 2 == 0 else g + 1
        if 2 <= steps <= 2049:
            return steps - steps % 2        # the whole timed region is one graph launch (measured: 4.35 vs 4.47 us/step at 40);
                                            # an odd K leaves one step to launch by hand
        for cand in range(g, 4 * g + 1, 2):
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
                    zero_first=True)

    @staticmethod
    def m_white():
        return 0                                # Noise.Color.white

    def _step_noise_filter_fused(self):
        self.m.paint(self.span, [self._next()], None, False, self.params, zero_first=True)

    def _note_on(self):
        # config 5: note on for buffers 0-23, then off (attack -> decay -> sustain -> release), repeating
        k = self.nsteps % 48
        self.nsteps += 1
        return k < 24, k == 0

    def _step_nice(self):
        on, new = self._note_on()
        self.m.paint(self.span, [self._next()], [], new, self.m.Params(SR, self.freq, on), zero_first=True)

    def _step_script(self):
        on, new = self._note_on()
        self.m.paint(self.span, [self._next()], None, new, {"sample_rate": SR, "freq": self.freq, "note_on": on}, zero_first=True)

    def _step_nice_mix(self):
        row = self.nsteps % 48
        on, new = self._note_on()                                  # advances self.nsteps
        self.m.paint_mix(self.span, self.mixes[row], new, self.m.Params(SR, self.freq, on), zero_first=True)

    def exchange(self):
        """config 5's one exchange step: the GPUs' partial mixes are summed over RCCL/xGMI.  A 4 KiB
        all-reduce per buffer would be pure latency (~20-40 us against 170 us of rendering), so the 48 buffers of a
        batch go in ONE all-reduce of [48][frames] (192 KiB) after the batch's launches -- an offline renderer only
        needs the mixed audio once the batch is done."""
        if self.name == "nice_mix" and self.world > 1:
            from zang_amd import sharding
            sharding.allreduce_mix(self.mixes)


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
        voices = zs_interp.make_voices(wl.program.script, SCRIPT_MODULE, V)
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
    res = {"value": V * F * nbuf / dt, "unit": "voice-samples/s", "cores": 1, "kind": "port",
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
    if wl.name not in ("pulseosc", "nice"):
        return None
    from oracle import pyoracle as po
    L = po.lib()
    V, F = wl.V, wl.F
    sample = np.arange(V) if V <= 65536 else np.arange(0, V, V // 4096)[:4096]
    st = wl.m.state()
    out = wl.ring[0]
    if wl.name == "pulseosc":
        wl.m.paint(wl.span, [out], [], False, wl.params, zero_first=True)
    else:
        wl.m.paint(wl.span, [out], [], True, wl.m.Params(SR, wl.freq, True), zero_first=True)
    ctx.sync()
    got = out[:, torch.from_numpy(sample).to(out.device)].cpu().numpy().T
    ref = np.zeros((len(sample), F), np.float32)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for k, v in enumerate(sample):
        if wl.name == "pulseosc":
            o = po.PulseOsc(int(st["cnt"][v]))
            L.zo_pulseosc_paint(C.byref(o), 0, F, po.fptr(ref[k]), SR, po.constant(float(wl.freq_h[v])), float(wl.color_h[v]))
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
    return {"checked_voices": int(len(sample)), "frames": F, "bitexact": bool(same.all()),
            "mismatching_samples": int((~same).sum()), "against": "oracle from the GPU's own carried state"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ZH_BENCH_EMULATE=1: multi-process dry run on a single GPU (every rank on device 0, gloo instead
    # of RCCL) -- used only to exercise the N>1 code path where one GPU is available.
    emulate = os.environ.get("ZH_BENCH_EMULATE") == "1"
    device_index = 0 if (world == 1 or emulate) else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(device_index)
        if emulate:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
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
    wl = Workload(args.workload, ctx, V, F, first_voice=rank * V, ring_bytes=args.ring_mib << 20, world=world, pad=args.pad_voices)
    lib = ctx.lib

    def barrier():
        if world > 1:
            dist.barrier()

    def make_event():
        h = C.c_void_p()
        abi.check(lib.zh_event_create(ctx.handle, C.byref(h)), "zh_event_create")
        return h

    # The per-buffer loop is launch-bound at 4096 voices (a 16 MiB paint takes ~6 us on the
    # device, a Python->C->hipLaunchKernel call about as long), so G consecutive steps are
    # captured once into a hipGraph and replayed; G = the output ring length (even, so the
    # oscillator's double-buffered state ends where it started in the capture).
    G = wl.graph_steps(args.steps) if not args.eager else 0
    graph = None
    if G:
        for _ in range(G):          # one eager pass first: lazy allocations happen outside capture
            wl.step()
        torch.cuda.synchronize()
        graph = ctx.capture(lambda: [wl.step() for _ in range(G)])

    def run_steps(n):
        done = 0
        if graph is not None:
            while n - done >= G:
                graph.launch()
                wl.exchange()
                done += G
        for _ in range(n - done):
            wl.step()
        if n - done:
            wl.exchange()

    ev0, ev1 = make_event(), make_event()
    run_steps(args.warmup)
    if graph is not None and args.warmup < G:
        graph.launch()                      # untimed: the first replay of a graph uploads it
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # HIP events on the launch stream bracket exactly the K timed steps: device time per
    # launch = elapsed / K (includes the ~1 us inter-kernel boundaries, so it can only
    # under-state the kernel's own rate; profiles/ holds the rocprofv3 per-kernel average).
    abi.check(lib.zh_event_record(ctx.handle, ev0), "zh_event_record")
    run_steps(args.steps)
    abi.check(lib.zh_event_record(ctx.handle, ev1), "zh_event_record")
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = C.c_float()
    abi.check(lib.zh_event_elapsed_ms(ev0, ev1, C.byref(ms)), "zh_event_elapsed_ms")
    step_ms_events = ms.value / args.steps
    lib.zh_event_destroy(ev0)
    lib.zh_event_destroy(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if emulate else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HBM traffic per launch from the committed rocprofv3 --pmc passes of this same workload
    # (tools/summarize_pmc.py; counters cannot be read from inside the process)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_pulseosc4096.json")
    if args.workload == "pulseosc" and V == 4096 and F == 1024 and os.path.exists(pmc):
        traffic = json.load(open(pmc))["hbm_bytes_per_launch"]

    total_units = world * V * F * args.steps
    value = total_units / elapsed
    achieved = wl.bytes_per_step / (step_ms_events * 1e-3) / 1e9
    out = {
        "metric": "voice-samples/sec", "value": value, "unit": "voice-samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {V} voices/GPU x {F} frames, zero+paint per {F}-frame buffer, 48 kHz",
                   "voices_per_gpu": V, "frames": F, "ring_images": wl.nring, "launch": "eager" if graph is None else f"hipGraph x{G} steps", "parallelism": f"voices sharded x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": wl.kernel,
                     "frac_of_measured_store_rate": achieved / HBM_STORE_GBS,   # also quote / 6200 "achievable"
                     "algorithmic_bytes_per_launch": wl.bytes_per_step, "launch_ms_hip_events": step_ms_events},
        "equiv_write_GBs_whole_job": value * 4 / 1e9,
    }
    if rank == 0 and world == 1 and not args.no_parity:
        out["parity"] = parity_check(wl, ctx)
    if rank == 0 and world == 1 and not args.no_cpu:
        cb = cpu_baseline(args, wl)
        if cb:
            out["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
