#!/bin/bash
# GPU box: HBM traffic of every kernel of one bench.py workload from the PMC counters -- one counter per rocprofv3 pass
# (--kernel-trace only), as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at
# 64 bytes).  usage: tools/pmc_traffic.sh <name> <commit> <steps> [bench args...]  -> gpurun_out/pmc_traffic/<name>.json
# PMC_GRAPH=1: the form the driver's run takes instead of --eager -- `--steps <steps> --warmup 5`, the recorded (coalesced) graph replayed: the
# counters of the launches a replay makes; the kernel bench.py's line names gets `buffers_per_launch` from that line.
name=$1; commit=$2; steps=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "${PMC_GRAPH:-0}" = "1" ]; then form="--warmup 5"; else form="--warmup 0 --eager"; fi
for c in WRITE_SIZE FETCH_SIZE; do
  d=/tmp/pmct_${name}_$c; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps $steps $form --no-cpu --no-parity --no-config5 --repeats 0 --no-rehearsal > $out/${name}_run_$c.log 2>&1
done
python3 - $name $commit $steps "$*" $out "${PMC_GRAPH:-0}" <<'PY'
import csv, glob, json, statistics, sys
name, commit, steps, args, out, graph = sys.argv[1:7]
steps = int(steps)
graph = graph == "1"
line = None
for l in open(f"{out}/{name}_run_WRITE_SIZE.log"):
    if l.startswith("{") and '"roofline"' in l:
        line = json.loads(l)
per = {}
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob(f"/tmp/pmct_{name}_{c}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        per.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
kernels = {}
total = 0.0
for k, d in per.items():
    n = max(len(v) for v in d.values())
    if n < (2 if graph else steps // 2) or k.startswith("__amd_rocclr") or "at::native" in k:    # set-up kernels (fills, seeds, tables, torch's own): not part of a step
        continue
    w = statistics.mean(d.get("WRITE_SIZE", [0.0])) * 1024.0          # the counters are in KiB
    f = statistics.mean(d.get("FETCH_SIZE", [0.0])) * 1024.0 * 2.0    # gfx950 correction
    if graph:
        # a replayed graph: launches counted as they come; the kernel the bench line names paints `buffers_per_launch` buffers each
        bpl = 1.0
        rl = (line or {}).get("roofline", {})
        named = (rl.get("rocprofv3_kernel_average") or {}).get("kernel", "")
        base = (rl.get("kernel") or "").split("<")[0].split("[")[0]
        if base and base in k and (not named or k.strip() in named or named.split("(")[0].strip() == k.strip()):
            bpl = float(rl.get("buffers_per_launch") or 1.0)
        kernels[k] = {"launches_counted": n, "buffers_per_launch": bpl, "write_bytes_per_launch": w, "fetch_bytes_per_launch_corrected": f,
                      "fetch_bytes_per_launch_raw": f / 2.0, "hbm_bytes_per_step": (w + f) / bpl}
        continue
    per_step = n / steps
    kernels[k] = {"launches_per_step": per_step, "write_bytes_per_launch": w, "fetch_bytes_per_launch_corrected": f, "fetch_bytes_per_launch_raw": f / 2.0,
                  "hbm_bytes_per_step": (w + f) * per_step}
    total += (w + f) * per_step
res = {"name": name, "commit": commit, "command": f"bench.py {args} --steps {steps} " + ("--warmup 5 (the recorded graph replayed)" if graph else "--warmup 0 --eager") + ", one rocprofv3 --pmc pass per counter",
       "corrections": "FETCH_SIZE x 2 on gfx950 (128-byte read requests tallied at 64 bytes: exact for wide coalesced reads, an upper bound for kernels that read 4 bytes per lane); WRITE_SIZE as reported; counters in KiB (MI355X_MICROARCH.md, HBM)",
       "kernels": kernels, "hbm_bytes_per_step": total if not graph else None}
if graph and line:
    res["bench_line"] = {"value_form": line.get("value_form"), "kernels_launched_per_step": line["roofline"].get("kernels_launched_per_step"),
                         "buffers_per_launch": line["roofline"].get("buffers_per_launch")}
json.dump(res, open(f"{out}/{name}.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1800])
PY
