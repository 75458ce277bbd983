#!/bin/bash
# GPU box: the fuzzers on the round's last build.  -> gpurun_out/r06_fuzz/fuzz_final.txt
out=gpurun_out/r06_fuzz/fuzz_final.txt
mkdir -p gpurun_out/r06_fuzz; : > $out
run() { echo "== $*" >> $out; ( time timeout 1500 python "$@" ) 2>&1 | grep -v "amdgpu.ids" | tail -6 >> $out; }
run tools/fuzz_many.py 200
run tools/fuzz_scripts.py 200 12000
run tools/fuzz_scripts.py 400 13000 roles
run tools/fuzz_scripts.py 60 14000 tolerant
run tools/fuzz_spans.py 60
run tools/fuzz_filter.py 600 30000
run tools/fuzz_nf_pipeline.py 100
cat $out
