#!/bin/bash
# GPU box: a longer campaign of every fuzzer on the round's last kernels -> gpurun_out/r05/fuzz_long.txt
out=gpurun_out/r05/fuzz_long.txt
mkdir -p gpurun_out/r05; : > $out
run() { echo "== $*" >> $out; ( time timeout 1700 python "$@" ) 2>&1 | grep -v "amdgpu.ids" | tail -5 >> $out; }
run tools/fuzz_many.py 1500
run tools/fuzz_tolerant.py 3000 40000
run tools/fuzz_tolerant.py 300 50000 echoes
run tools/fuzz_scripts.py 500 8000
run tools/fuzz_scripts.py 150 9000 tolerant
run tools/fuzz_spans.py 100
run tools/fuzz_filter.py 1500 20000
cat $out
