#!/bin/bash
# GPU box, round 5: SQ counters of the config-5 shard's kernel, exact and ZH_PAINT_TOLERANT (multiply-adds fused), and at half the voices
# (own --pmc passes, --kernel-trace only) -> gpurun_out/pmc_valu_<tag>/
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  name=$1; shift
  d=/tmp/pmc_$name
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-parity --eager > /dev/null 2>&1
  cp $d/*/*counter_collection.csv $out/${name}_counters.csv 2>/dev/null
}
run2() {  # the second counter set of the same workload (its own pass): wave cycles, any-instruction activity, waits, LDS
  name=$1; shift
  d=/tmp/pmcb_$name
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-parity --eager > /dev/null 2>&1
  cp $d/*/*counter_collection.csv $out/${name}_counters_b.csv 2>/dev/null
}
run nice_mix131072 --workload nice_mix --voices 131072 --steps 48 --warmup 0 --no-rehearsal --repeats 0
run2 nice_mix131072 --workload nice_mix --voices 131072 --steps 48 --warmup 0 --no-rehearsal --repeats 0
run nice_mix131072_tolerant --workload nice_mix --voices 131072 --steps 48 --warmup 0 --no-rehearsal --repeats 0 --tolerant
run2 nice_mix131072_tolerant --workload nice_mix --voices 131072 --steps 48 --warmup 0 --no-rehearsal --repeats 0 --tolerant
run nice_mix65536 --workload nice_mix --voices 65536 --steps 48 --warmup 0 --no-rehearsal --repeats 0
python3 - $out <<'PY'
import csv, glob, json, statistics, sys, os
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + "/*_counters.csv")):
    rows = list(csv.DictReader(open(f)))
    fb = f.replace("_counters.csv", "_counters_b.csv")
    if os.path.exists(fb):
        rows += list(csv.DictReader(open(fb)))
    byk = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if not any(s in k for s in ("k_nice", "k_noise_filter")) or "seed" in k:
            continue
        byk.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, c in byk.items():
        m = {n: statistics.mean(v) for n, v in c.items()}
        waves = m.get("SQ_WAVES", 0)
        m["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / waves if waves else None
        m["salu_insts_per_wave"] = m.get("SQ_INSTS_SALU", 0) / waves if waves else None
        gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0                     # the csv holds the sum over the 8 XCDs
        m["cycles_per_xcd"] = gui
        m["valu_busy_pct"] = 100.0 * m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / gui if gui else None     # rocprof's VALUBusy, SIMD_NUM = 1024
        m["cycles_per_valu_inst_per_simd"] = gui * 1024 / m["SQ_INSTS_VALU"] if m.get("SQ_INSTS_VALU") else None
        if m.get("SQ_WAVE_CYCLES"):
            m["wave_cycles_waiting_pct"] = 100.0 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
            m["wave_cycles_issuing_pct"] = 100.0 * m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
            m["lds_insts_per_wave"] = m.get("SQ_INSTS_LDS", 0) / waves if waves else None
        res[os.path.basename(f).replace("_counters.csv", "") + ":" + k] = m
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
