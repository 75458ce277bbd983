#!/bin/bash
# round 5: the single-launch time-parallel Noise -> Filter form (k_nf_tp1): parity, a soak of it, and its time
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_tolerant.py -x -q -m gpu -k "noise_filter" 2>&1 | tail -15 | tee gpurun_out/r05/test_nftp1.log
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_tolerant.py -x -q -m gpu -k "test_noise_filter_tolerant or jump_tables" 2>&1 | tail -1; done | tee gpurun_out/r05/soak_nftp1.log
for rep in 1 2 3; do
python bench.py --workload noise_filter_fused --tolerant --no-cpu --no-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('nf fused tolerant: ms/step %.6f ev %.6f value %.4g frac %.4f' % (d['ms_per_step'], d['roofline']['launch_ms_hip_events'], d['value'], d['roofline']['frac']))
"
done | tee gpurun_out/r05/bench_nftp1.txt
python bench.py --workload noise_filter_fused --no-cpu --no-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('nf fused exact: ms/step %.6f ev %.6f' % (d['ms_per_step'], d['roofline']['launch_ms_hip_events']))
" | tee -a gpurun_out/r05/bench_nftp1.txt
python bench.py --workload noise_filter_fused --tolerant --voices 16384 --no-cpu --no-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('nf fused tolerant 16384: ms/step %.6f ev %.6f' % (d['ms_per_step'], d['roofline']['launch_ms_hip_events']))
" | tee -a gpurun_out/r05/bench_nftp1.txt
