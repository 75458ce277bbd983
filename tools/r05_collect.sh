#!/bin/bash
# GPU box, round 5: the measurements DESIGN.md and profiles/r05/ quote, from ONE box.  usage: tools/r05_collect.sh  -> gpurun_out/r05c/
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r05c
mkdir -p $out
cd $root
commit=$(python3 -c "import json; print(json.load(open('zang_amd/build_info.json')).get('commit') or 'unknown')" 2>/dev/null || echo unknown)
# 1. the driver's command line, five times
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 > $out/bench_driver_args_$i.json 2>/dev/null; done
python bench.py > $out/bench_default.json 2>/dev/null
# 2. rocprofv3 kernel statistics (profiled + unprofiled line per workload)
p() { name=$1; shift; bash tools/prof_one.sh r05c $name "$@" > $out/prof_$name.txt 2>&1; }
p pulseosc4096_driver_args --steps 20 --warmup 5
p pulseosc4096
p pulseosc65536 --voices 65536 --steps 100 --warmup 10
p pulseosc1M --voices 1048576 --steps 40 --warmup 4
p noise_filter_fused4096 --workload noise_filter_fused
p noise_filter_fused4096_tolerant --workload noise_filter_fused --tolerant
p noise_filter4096_tolerant --workload noise_filter --tolerant
p nice4096 --workload nice --steps 96 --warmup 48
p nice4096_tolerant --workload nice --tolerant --steps 96 --warmup 48
p nice_mix131072 --workload nice_mix --voices 131072 --steps 96 --warmup 48
p nice_mix65536 --workload nice_mix --voices 65536 --steps 96 --warmup 48
p script131072 --workload script --voices 131072 --steps 96 --warmup 48
# 3. HBM traffic from the PMC counters (separate passes per counter)
t() { name=$1; steps=$2; shift 2; bash tools/pmc_traffic.sh pmc_traffic_$name $commit $steps "$@" > /dev/null 2>&1; cp gpurun_out/pmc_traffic/pmc_traffic_$name.json $out/ 2>/dev/null; }
t pulseosc4096 64
t noise_filter_fused4096 32 --workload noise_filter_fused
t noise_filter_fused4096_tolerant 32 --workload noise_filter_fused --tolerant
t nice_mix131072 48 --workload nice_mix --voices 131072
# 4. every module on its own
python tools/bench_modules.py 4096 > $out/modules_4096.txt 2>&1
python tools/bench_modules.py 131072 > $out/modules_131072.txt 2>&1
ls $out
