#!/bin/bash
# round 5: the two-kernel tolerant forms after the single-launch experiment: chunk j on XCD j % 8, no per-sample noise test
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_tolerant.py -x -q -m gpu -k "noise_filter or pink" 2>&1 | tail -5 | tee gpurun_out/r05/test_nftp2.log
for rep in 1 2 3; do
python bench.py --workload noise_filter_fused --tolerant --no-cpu --no-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('nf fused tolerant: ms/step %.6f ev %.6f value %.4g frac %.4f' % (d['ms_per_step'], d['roofline']['launch_ms_hip_events'], d['value'], d['roofline']['frac']))
"
done | tee gpurun_out/r05/bench_nftp2.txt
python bench.py --workload noise_filter_fused --tolerant --voices 16384 --no-cpu --no-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('nf fused tolerant 16384: ms/step %.6f ev %.6f' % (d['ms_per_step'], d['roofline']['launch_ms_hip_events']))
" | tee -a gpurun_out/r05/bench_nftp2.txt
