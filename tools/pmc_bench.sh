#!/bin/bash
# GPU box: SQ counters (one --pmc pass, --kernel-trace only) of a bench.py workload's kernels.
# usage: tools/pmc_bench.sh <name> "<counters>" [bench args...]  -> gpurun_out/pmc_<name>.json
name=$1; counters=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmcb_$name; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-parity --no-config5 --repeats 0 --eager > $out/pmc_${name}_bench.txt 2>&1
python3 - $d $out/pmc_$name.json <<'PY'
import csv, glob, json, statistics, sys
d, outp = sys.argv[1:3]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
byk = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    byk.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, c in byk.items():
    m = {n: statistics.mean(v) for n, v in c.items()}
    m["dispatches"] = len(next(iter(c.values())))
    res[k] = m
json.dump(res, open(outp, "w"), indent=1)
for k, m in res.items():
    if m["dispatches"] >= 8:
        print(k[:90], {n: round(v, 1) for n, v in m.items()})
PY
