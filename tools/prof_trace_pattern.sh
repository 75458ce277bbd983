#!/bin/bash
# GPU box: per-dispatch durations of the headline kernel (rocprofv3 --kernel-trace), summarised by ring position
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02k}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-config5 --no-parity --repeats 0 ${@:2} > /dev/null 2>&1
python3 - <<PY
import csv,glob,statistics as st
f=glob.glob("$out/tr/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_osc_const4' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
gap=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in range(len(rows)-1)]
print("n",len(d),"mean",st.mean(d),"median",st.median(d),"p10",sorted(d)[len(d)//10],"p90",sorted(d)[9*len(d)//10])
print("gap mean",st.mean(gap),"median",st.median(gap))
print("start-to-start median", st.median([int(rows[i+1]['Start_Timestamp'])-int(rows[i]['Start_Timestamp']) for i in range(len(rows)-1)]))
steady=d[200:]
by=[[] for _ in range(32)]
for i,x in enumerate(steady): by[i%32].append(x)
print("by ring slot (mean):",[round(st.mean(b)) for b in by])
print("first 40 of steady:",steady[:40])
PY
rm -rf $out/tr
