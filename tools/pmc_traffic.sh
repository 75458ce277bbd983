#!/bin/bash
# GPU box: HBM traffic of every kernel of one bench.py workload from the PMC counters -- one counter per rocprofv3 pass
# (--kernel-trace only), as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at
# 64 bytes).  usage: tools/pmc_traffic.sh <name> <commit> <steps> [bench args...]  -> gpurun_out/pmc_traffic/<name>.json
name=$1; commit=$2; steps=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  d=/tmp/pmct_${name}_$c; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps $steps --warmup 0 --eager --no-cpu --no-parity --no-config5 --repeats 0 --no-rehearsal > $out/${name}_run_$c.log 2>&1
done
python3 - $name $commit $steps "$*" $out <<'PY'
import csv, glob, json, statistics, sys
name, commit, steps, args, out = sys.argv[1:6]
steps = int(steps)
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
    if n < steps // 2:                # set-up kernels (fills, seeds, tables): not part of a step
        continue
    w = statistics.mean(d.get("WRITE_SIZE", [0.0])) * 1024.0          # the counters are in KiB
    f = statistics.mean(d.get("FETCH_SIZE", [0.0])) * 1024.0 * 2.0    # gfx950 correction
    per_step = n / steps
    kernels[k] = {"launches_per_step": per_step, "write_bytes_per_launch": w, "fetch_bytes_per_launch_corrected": f, "fetch_bytes_per_launch_raw": f / 2.0,
                  "hbm_bytes_per_step": (w + f) * per_step}
    total += (w + f) * per_step
res = {"name": name, "commit": commit, "command": f"bench.py {args} --steps {steps} --warmup 0 --eager, one rocprofv3 --pmc pass per counter",
       "corrections": "FETCH_SIZE x 2 on gfx950 (128-byte read requests tallied at 64 bytes: exact for wide coalesced reads, an upper bound for kernels that read 4 bytes per lane); WRITE_SIZE as reported; counters in KiB (MI355X_MICROARCH.md, HBM)",
       "kernels": kernels, "hbm_bytes_per_step": total}
json.dump(res, open(f"{out}/{name}.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1800])
PY
