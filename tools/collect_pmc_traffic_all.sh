#!/bin/bash
# GPU box: tools/pmc_traffic.sh for every workload whose bench line quotes `roofline.traffic` -> gpurun_out/pmc_traffic/pmc_traffic_<workload><voices>[_tolerant].json
# (copy into profiles/<tag>/).  usage: tools/collect_pmc_traffic_all.sh <commit>
commit=${1:-unknown}
t() { name=$1; steps=$2; shift 2; bash $GRAFT_REPO_ROOT/tools/pmc_traffic.sh pmc_traffic_$name $commit $steps "$@" > /dev/null 2>&1; echo "$name: $(python3 -c "import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic/pmc_traffic_$name.json')); print('%.2f MB per step' % (d['hbm_bytes_per_step'] / 1e6), {k.split('<')[0][-28:]: round(v['hbm_bytes_per_step'] / 1e6, 2) for k, v in d['kernels'].items()})" 2>&1)"; }
t pulseosc4096 64
t nice_mix131072 48 --workload nice_mix --voices 131072
t nice_mix4096 48 --workload nice_mix --voices 4096
t nice131072 48 --workload nice --voices 131072
t nice4096 48 --workload nice
t noise_filter4096 32 --workload noise_filter
t noise_filter4096_tolerant 32 --workload noise_filter --tolerant
t noise_filter_fused4096 32 --workload noise_filter_fused
t noise_filter_fused4096_tolerant 32 --workload noise_filter_fused --tolerant
t noise_filter_fused131072 32 --workload noise_filter_fused --voices 131072
