#!/bin/bash
# GPU box: every bench workload WITHOUT a profiler attached (the numbers DESIGN.md quotes), one JSON line each,
# plus the config-4 song timing -> gpurun_out/bench_all_<tag>/
tag=${1:-r01}
out=gpurun_out/bench_all_$tag
mkdir -p $out
: > $out/bench_unprofiled.jsonl
run() { python bench.py "$@" 2>/dev/null | tail -1 >> $out/bench_unprofiled.jsonl; }
run --cpu-seconds 10
run --voices 65536 --steps 100 --warmup 10 --no-cpu
run --voices 524288 --steps 40 --warmup 4 --no-cpu
run --voices 1048576 --steps 40 --warmup 4 --no-cpu
run --workload noise_filter --no-cpu
run --workload noise_filter --voices 131072 --steps 50 --warmup 10 --no-cpu
run --workload noise_filter_fused --no-cpu
run --workload noise_filter_fused --voices 131072 --steps 50 --warmup 10 --cpu-seconds 5
run --workload nice --steps 96 --warmup 48 --no-cpu
run --workload nice --voices 131072 --steps 96 --warmup 48 --cpu-seconds 5
run --workload nice_mix --voices 131072 --steps 96 --warmup 48 --no-cpu
run --workload nice_mix --voices 1048576 --steps 48 --warmup 48 --no-cpu
run --workload script --voices 4096 --steps 96 --warmup 48 --no-cpu
run --workload script --voices 131072 --steps 96 --warmup 48 --cpu-seconds 5
python tools/gen_song.py > /tmp/song.txt
python tools/time_song.py /tmp/song.txt 60 > $out/song_60s.txt 2>&1
python - $out/bench_unprofiled.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print("%-70s %.3e /s  %9.2f us/step  frac %.3f" % (d["config"]["workload"][:70], d["value"], d["ms_per_step"] * 1e3, d["roofline"]["frac"]))
PY
tail -1 $out/song_60s.txt
