#!/bin/bash
cd $GRAFT_REPO_ROOT
export ZH_NICE_MIX_WG8_MIN=0
bash tools/pmc_traffic.sh pmc_traffic_nice_mix131072_wg8 $(git rev-parse --short HEAD 2>/dev/null || echo 77fe4c1) 48 --workload nice_mix --voices 131072 > /dev/null 2>&1
python3 -c "import json; d=json.load(open('gpurun_out/pmc_traffic/pmc_traffic_nice_mix131072_wg8.json')); print('%.2f MB per step' % (d['hbm_bytes_per_step'] / 1e6), {k.split('<')[0][-28:]: round(v['hbm_bytes_per_step'] / 1e6, 2) for k, v in d['kernels'].items()})"
