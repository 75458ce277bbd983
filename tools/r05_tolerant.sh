#!/bin/bash
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests/test_gpu_tolerant.py -x -q -m gpu -s -k "carried or in_a_graph" 2>&1 | grep -v "^$" | tail -12 | tee gpurun_out/r05/test_carried.log
python bench.py --workload noise_filter_fused --tolerant --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps(d['parity']))
" | tee gpurun_out/r05/bench_tolerant_parity.txt
python bench.py --workload nice --tolerant --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps(d['parity']))
" | tee -a gpurun_out/r05/bench_tolerant_parity.txt
