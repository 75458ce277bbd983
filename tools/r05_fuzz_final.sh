#!/bin/bash
# GPU box: the fuzzers on the round's last build.  -> gpurun_out/r05/fuzz_final.txt
out=gpurun_out/r05/fuzz_final.txt
mkdir -p gpurun_out/r05; : > $out
run() { echo "== $*" >> $out; ( time timeout 1500 python "$@" ) 2>&1 | grep -v "amdgpu.ids" | tail -6 >> $out; }
run tools/fuzz_many.py 60
run tools/fuzz_tolerant.py 1500 20000
run tools/fuzz_scripts.py 150 4000
run tools/fuzz_scripts.py 60 5000 tolerant
run tools/fuzz_spans.py 40
run tools/fuzz_filter.py 400 12000
cat $out
