#!/bin/bash
# GPU box: tools/bench_modules.py over a ladder of voice counts -> one table (us per 1024-frame paint), for the voice-count
# limits of the frame-range / pipeline forms.  usage: tools/scan_modules.sh <out.txt>
out=${1:-gpurun_out/modules_scan.txt}
Vs="4096 8192 16384 24576 32768 40960 49152 65536 98304 131072"
mkdir -p gpurun_out/scan
for V in $Vs; do python tools/bench_modules.py $V > gpurun_out/scan/scan_$V.txt 2>/dev/null; done
python - "$out" $Vs <<'PY'
import re, sys
out, Vs = sys.argv[1], [int(v) for v in sys.argv[2:]]
rows = {}
for V in Vs:
    for l in open(f"gpurun_out/scan/scan_{V}.txt"):
        m = re.match(r"^(.{46})\s+([\d.]+)\s", l)
        if m and not l.startswith(("#", "module")):
            rows.setdefault(m.group(1).strip(), {})[V] = float(m.group(2))
with open(out, "w") as f:
    f.write("# us per zero+paint of one 1024-frame buffer (20 paints per graph), one MI355X, by voice count\n")
    f.write("%-44s " % "module" + " ".join("%7d" % v for v in Vs) + "\n")
    for k, d in rows.items():
        f.write("%-44s " % k[:44] + " ".join("%7.1f" % d.get(v, 0) for v in Vs) + "\n")
print(open(out).read())
PY
