#!/bin/bash
# GPU box: one bench.py workload with the library in the tree and with another build, on the same box, alternating.
# usage: tools/ab_bench.sh <other.so> [bench args...]   -> "us per step: other | tree" three times
other=$1; shift
run() { python bench.py "$@" --no-cpu --no-parity --no-config5 --repeats 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f (events %.2f)' % (d['ms_per_step']*1e3, d['roofline']['launch_ms_hip_events']*1e3))"; }
for i in 1 2 3; do b=$(ZANG_HIP_LIB=$other run "$@"); a=$(run "$@"); echo "$* : other $b us | tree $a us"; done
