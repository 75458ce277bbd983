#!/bin/bash
# GPU box: tools/bench_modules.py with the tree's library and another build, alternating (B A B A), per case the better of each side's two runs.
# usage: tools/ab_modules2.sh <other.so> <voices> ["case name filter"]  -> gpurun_out/ab2_modules_<voices>.txt
other=$1; v=$2; only=$3
out=gpurun_out/ab2_modules_$v.txt; : > $out
for rep in 1 2; do
  echo "== other" >> $out; ZANG_HIP_LIB=$other ZH_BENCH_ONLY="$only" python tools/bench_modules.py $v 2>/dev/null >> $out
  echo "== tree" >> $out; ZH_BENCH_ONLY="$only" python tools/bench_modules.py $v 2>/dev/null >> $out
done
python - "$out" <<'PY'
import re, sys
d = {"other": {}, "tree": {}}; cur = None
for line in open(sys.argv[1]):
    if line.startswith("== "):
        cur = d[line[3:].strip()]; continue
    m = re.match(r"(.+?)\s{2,}([\d.]+)\s+[\d.e+]+\s+[\d.]+\s*$", line)
    if m and not line.startswith("#") and not line.startswith("module"): cur.setdefault(m.group(1).strip(), []).append(float(m.group(2)))
for k in d["tree"]:
    if k in d["other"]:
        a, b = min(d["tree"][k]), min(d["other"][k])
        print(f"{k:52s} tree {a:8.1f}  other {b:8.1f} us  (other {b/a-1:+.1%})   runs tree {d['tree'][k]} other {d['other'][k]}")
PY
