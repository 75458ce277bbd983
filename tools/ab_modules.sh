#!/bin/bash
# GPU box: tools/bench_modules.py with the library in the tree and with another build, on the same box.
# usage: tools/ab_modules.sh <other.so> <voices> ["case name filter"]  -> gpurun_out/ab_modules_<voices>.txt
other=$1; v=$2; only=$3
out=gpurun_out/ab_modules_$v.txt
{ echo "== this tree"; ZH_BENCH_ONLY="$only" python tools/bench_modules.py $v 2>/dev/null; echo "== $other"; ZANG_HIP_LIB=$other ZH_BENCH_ONLY="$only" python tools/bench_modules.py $v 2>/dev/null; } > $out
python - "$out" <<'PY'
import re, sys
a, b, cur = {}, {}, None
for line in open(sys.argv[1]):
    if line.startswith("== "):
        cur = a if cur is None else b
        continue
    m = re.match(r"(.+?)\s{2,}([\d.]+)\s+[\d.e+]+\s+[\d.]+\s*$", line)
    if m and not line.startswith("#") and not line.startswith("module"): cur[m.group(1).strip()] = float(m.group(2))
for k in a:
    if k in b: print(f"{k:52s} {b[k]:9.1f} -> {a[k]:9.1f} us  ({a[k]/b[k]-1:+.1%})")
PY
