#!/bin/bash
# GPU box: per-dispatch durations of one bench command in launch order (rocprofv3 --kernel-trace) -> gpurun_out/series_<name>.txt
# usage: tools/prof_series.sh <name> <kernel substring> [bench args...]
name=$1; pat=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=/tmp/series_$name
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-config5 --no-parity --repeats 0 > /dev/null 2>&1
python3 - $d "$pat" > $out/series_$name.txt <<'PY'
import csv, glob, sys
d, pat = sys.argv[1:3]
rows = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
prev_end = None
for k, (s, e) in enumerate(rows):
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%5d  dur %8.2f us  gap_from_prev_end %9.2f us" % (k, (e - s) / 1e3, gap))
    prev_end = e
PY
