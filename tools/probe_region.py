#!/usr/bin/env python3
"""Where the fixed cost of bench.py's short timed region goes: 20 steps of the 4,096-voice PulseOsc paint are
~95 us of device time, and the wall-clock region around them was ~117 us.  Times, 300 samples each, on one box:
  empty        synchronize only
  graph        graph.launch + synchronize
  ev+graph     event record, graph.launch, event record, synchronize     (bench.py through round 3)
  graph(ev)    the two event records captured INTO the graph
and prints median / min / p90 in microseconds plus the device time between the events.  Environment knobs of the HIP runtime are
tried by running this script again with them set (ROC_ACTIVE_WAIT_TIMEOUT ...): `--env K=V`."""
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    steps = int(os.environ.get("PROBE_STEPS", "20"))
    import torch
    import bench
    import zang_amd
    from zang_amd import abi
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    ctx = zang_amd.Context(0)
    lib = ctx.lib
    wl = bench.Workload("pulseosc", ctx, 4096, 1024, first_voice=0, ring_bytes=512 << 20)
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()

    def ev():
        h = C.c_void_p()
        abi.check(lib.zh_event_create(ctx.handle, C.byref(h)), "zh_event_create")
        return h
    e0, e1, g0, g1 = ev(), ev(), ev(), ev()
    graph = ctx.capture(lambda: [wl.step() for _ in range(steps)])

    def with_events():
        abi.check(lib.zh_event_record(ctx.handle, g0), "rec")
        for _ in range(steps):
            wl.step()
        abi.check(lib.zh_event_record(ctx.handle, g1), "rec")
    graph_ev = ctx.capture(with_events)
    for g in (graph, graph_ev):
        g.launch()
    torch.cuda.synchronize()
    # clock warm: ~50 ms of the same work
    for _ in range(int(os.environ.get("PROBE_WARM", "500"))):
        graph.launch()
    torch.cuda.synchronize()

    def sample(fn, n=300):
        ws = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ws.append((time.perf_counter() - t0) * 1e6)
        ws.sort()
        return {"median": round(statistics.median(ws), 2), "min": round(ws[0], 2), "p90": round(ws[int(0.9 * len(ws))], 2)}

    def elapsed(a, b):
        ms = C.c_float()
        abi.check(lib.zh_event_elapsed_ms(a, b, C.byref(ms)), "elapsed")
        return ms.value * 1e3

    def ev_graph():
        abi.check(lib.zh_event_record(ctx.handle, e0), "rec")
        graph.launch()
        abi.check(lib.zh_event_record(ctx.handle, e1), "rec")
    out = {"steps": steps, "env": {k: os.environ[k] for k in os.environ if k.startswith(("ROC_", "HIP_", "GPU_", "DEBUG_HIP", "AMD_"))}}
    out["empty"] = sample(lambda: None)
    out["graph"] = sample(graph.launch)
    out["ev+graph"] = sample(ev_graph)
    out["ev+graph device_us"] = round(elapsed(e0, e1), 2)
    try:
        out["graph(ev)"] = sample(graph_ev.launch)
        out["graph(ev) device_us"] = round(elapsed(g0, g1), 2)
    except Exception as e:      # noqa: BLE001
        out["graph(ev)"] = "failed: %s" % e
    # cold start: idle 200 ms, then one region
    cold = []
    for _ in range(10):
        torch.cuda.synchronize(); time.sleep(0.2)
        t0 = time.perf_counter(); graph.launch(); torch.cuda.synchronize()
        cold.append((time.perf_counter() - t0) * 1e6)
    out["graph after 200 ms idle"] = {"median": round(statistics.median(cold), 2), "min": round(min(cold), 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
