#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of one bench command, then THE SAME command without the profiler on the same box
# -> gpurun_out/<tag>/<name>_kernel_stats.csv (rocprofv3's own summary) and <name>_summary.json = {profiled line, unprofiled
# line, per kernel: calls, average, min, p50, p90, max from the per-dispatch trace}.
# usage: tools/prof_one.sh <tag> <name> [bench args...]
tag=$1; name=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=$out/tmp_$name
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-config5 --repeats 0 > $out/${name}_bench.json 2>/dev/null
cp $d/*/*kernel_stats.csv $out/${name}_kernel_stats.csv
tail -1 $out/${name}_bench.json > $out/${name}_bench.json.tmp; mv $out/${name}_bench.json.tmp $out/${name}_bench.json
python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-config5 2>/dev/null | tail -1 > $out/${name}_bench_unprofiled.json
python3 - $d $out $name <<'PY'
import csv, glob, json, statistics, sys
d, out, name = sys.argv[1:4]
per = {}
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        per.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ks = {}
for k, v in per.items():
    v.sort()
    ks[k] = {"calls": len(v), "average_us": statistics.mean(v), "min_us": v[0], "p50_us": v[len(v) // 2], "p90_us": v[int(len(v) * 0.9)], "max_us": v[-1]}
def line(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None
res = {"profiled_line": line(f"{out}/{name}_bench.json"), "unprofiled_line_same_box": line(f"{out}/{name}_bench_unprofiled.json"), "kernels": ks}
# the lines embed `roofline.rocprofv3_kernel_average`, which bench.py reads from the COMMITTED profiles -- one collection behind
# the file written here.  Put this collection's own figure (the kernel_stats.csv beside the summary) in its place, and say so.
try:
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_sel", os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rows = list(csv.DictReader(open(f"{out}/{name}_kernel_stats.csv")))
    def own_for(d):
        # the row of THIS collection for the kernel the line says ran (bench.select_kernel_row: name from zh_graph_kernels, the
        # several-buffers instantiation when the launches were coalesced), and frac_kernel from it
        rl = d["roofline"]
        top, how = bench.select_kernel_row(rows, rl.get("kernel"), batch=(rl.get("buffers_per_launch") or 1) > 1, launch_us=(rl.get("launch_ms_hip_events") or 0) * 1e3 or None)
        if top is None:
            top, how = max(rows, key=lambda r: float(r.get("TotalDurationNs", 0) or 0)), "the row with the largest total duration (" + how + ")"
        own = {"file": f"{name}_kernel_stats.csv (this collection)", "kernel": top.get("Name", "")[:96], "selected": how, "calls": int(float(top.get("Calls", 0) or 0)),
               "average_us": float(top.get("AverageNs", 0) or 0) / 1e3}
        rl["rocprofv3_kernel_average"] = own
        if own["average_us"] and rl.get("algorithmic_bytes_per_launch") and rl.get("peak"):
            rl["frac_kernel"] = rl["algorithmic_bytes_per_launch"] / (own["average_us"] * 1e-6) / 1e9 / rl["peak"]
    for key in ("profiled_line", "unprofiled_line_same_box"):
        if res[key] and "roofline" in res[key]:
            own_for(res[key])
    for fn in (f"{out}/{name}_bench.json", f"{out}/{name}_bench_unprofiled.json"):
        d = line(fn)
        if d and "roofline" in d:
            own_for(d)
            open(fn, "w").write(json.dumps(d) + "\n")
except Exception as e:      # noqa: BLE001
    res["rocprofv3_kernel_average_note"] = f"not rewritten: {e}"
json.dump(res, open(f"{out}/{name}_summary.json", "w"), indent=1)
for k, s in sorted(ks.items(), key=lambda kv: -kv[1]["average_us"] * kv[1]["calls"])[:4]:
    print("%-70s calls %6d  avg %8.2f  min %8.2f  p50 %8.2f  p90 %8.2f  max %8.2f us" % (k[:70], s["calls"], s["average_us"], s["min_us"], s["p50_us"], s["p90_us"], s["max_us"]))
PY
rm -rf $d
