#!/bin/bash
mkdir -p gpurun_out/r05c
cd "$GRAFT_REPO_ROOT" || exit 1
commit=$(python3 -c "import json; print(json.load(open('zang_amd/build_info.json')).get('commit') or 'unknown')" 2>/dev/null || echo unknown)
timeout 1500 python -m pytest tests/test_gpu_tolerant.py -x -q -m gpu -k "noise_filter or pink" 2>&1 | tail -3
bash tools/pmc_traffic.sh pmc_traffic_noise_filter_fused4096_tolerant $commit 32 --workload noise_filter_fused --tolerant > /dev/null 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic_noise_filter_fused4096_tolerant.json gpurun_out/r05c/
python3 -c "
import json
d=json.load(open('gpurun_out/r05c/pmc_traffic_noise_filter_fused4096_tolerant.json')); print('%.2f MB/step'%(d['hbm_bytes_per_step']/1e6), {k.split('<')[0][-24:]: (round(v['write_bytes_per_launch']/1e6,2), round(v['fetch_bytes_per_launch_corrected']/1e6,2)) for k,v in d['kernels'].items()})
"
bash tools/prof_one.sh r05c noise_filter_fused4096_tolerant --workload noise_filter_fused --tolerant | head -3
