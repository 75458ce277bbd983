#!/bin/bash
# GPU box: SQ counters (one --pmc pass, --kernel-trace only) of the kernels of tools/bench_modules.py cases.
# usage: tools/pmc_modules.sh <name> <voices> "<case filter>"  -> gpurun_out/pmc_<name>.json (per kernel: mean counters + derived)
name=$1; v=$2; only=$3
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmcm_$name; rm -rf $d
export ZH_BENCH_ONLY="$only" ZH_BENCH_EAGER=1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/bench_modules.py $v > $out/pmc_${name}_modules.txt 2>&1
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
    waves = m.get("SQ_WAVES", 0)
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0                     # the csv holds the sum over the 8 XCDs
    if waves and gui:
        m["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / waves
        m["salu_insts_per_wave"] = m.get("SQ_INSTS_SALU", 0) / waves
        m["lds_insts_per_wave"] = m.get("SQ_INSTS_LDS", 0) / waves
        m["cycles_per_xcd"] = gui
        m["us_at_2.4GHz"] = gui / 2400.0
        m["valu_busy_pct"] = 100.0 * m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / gui
        m["cycles_per_valu_inst_per_simd"] = gui * 1024 / m["SQ_INSTS_VALU"] if m.get("SQ_INSTS_VALU") else None
    res[k] = m
json.dump(res, open(outp, "w"), indent=1)
for k, m in res.items():
    if m.get("dispatches", 0) >= 10 and "valu_busy_pct" in m:
        print("%-70s waves %7d  VALU/wave %8.0f  SALU/wave %7.0f  LDS/wave %6.0f  cycles %9.0f  VALUbusy %5.1f%%  cyc/VALU/SIMD %.2f" % (
            k[:70], m["SQ_WAVES"], m["valu_insts_per_wave"], m["salu_insts_per_wave"], m["lds_insts_per_wave"], m["cycles_per_xcd"], m["valu_busy_pct"], m["cycles_per_valu_inst_per_simd"] or 0))
PY
