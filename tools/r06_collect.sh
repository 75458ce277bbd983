#!/bin/bash
# GPU box, round 6: the measurements DESIGN.md and profiles/r06/ quote.  Two calls: `profiles` (rocprofv3 summaries, PMC traffic, module tables,
# config 4) -- copy what it leaves into profiles/r06/ -- then `lines` (the bench lines, which cite the committed summaries).  -> gpurun_out/r06c/
what=${1:-profiles}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06c
mkdir -p $out
cd $root
commit=$(python3 -c "import json; print(json.load(open('zang_amd/build_info.json')).get('commit') or 'unknown')" 2>/dev/null || echo unknown)
if [ "$what" = "profiles" ]; then
# 1. rocprofv3 kernel statistics (profiled + unprofiled line per workload) -- first: the lines of step 3 cite them through profiles/r06 only
#    once they are committed; prof_one.sh rewrites this collection's own lines with its own CSV
p() { name=$1; shift; bash tools/prof_one.sh r06c $name "$@" > $out/prof_$name.txt 2>&1; }
p pulseosc4096_driver_args --steps 20 --warmup 5
p pulseosc4096
p noise_filter_fused4096 --workload noise_filter_fused
p noise_filter_fused4096_tolerant --workload noise_filter_fused --tolerant
p nice_mix131072 --workload nice_mix --voices 131072 --steps 96 --warmup 48
p script4096 --workload script --voices 4096 --steps 96 --warmup 48
# 2. HBM traffic from the PMC counters (separate passes per counter): one launch per buffer, and the recorded (coalesced / pipelined) graphs
t() { name=$1; steps=$2; shift 2; bash tools/pmc_traffic.sh pmc_traffic_$name $commit $steps "$@" > /dev/null 2>&1; cp gpurun_out/pmc_traffic/pmc_traffic_$name.json $out/ 2>/dev/null; }
t pulseosc4096 64
PMC_GRAPH=1 t pulseosc4096_driver_args 20
t noise_filter_fused4096_tolerant 32 --workload noise_filter_fused --tolerant
PMC_GRAPH=1 t noise_filter_fused4096_tolerant_driver_args 20 --workload noise_filter_fused --tolerant
# 3. every module on its own
python tools/bench_modules.py 4096 2>&1 | grep -v amdgpu.ids > $out/modules_4096.txt
python tools/bench_modules.py 131072 2>&1 | grep -v amdgpu.ids > $out/modules_131072.txt
# 4. config 4
python tools/gen_song.py > /tmp/song.txt
python tools/time_song.py /tmp/song.txt 60 2>&1 | grep -v amdgpu.ids > $out/song_60s.txt
else
# 5. the driver's command line, five times, and the default line
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 > $out/bench_driver_args_$i.json 2>/dev/null; done
python bench.py > $out/bench_default.json 2>/dev/null
python bench.py --workload noise_filter_fused --tolerant --steps 20 --warmup 5 > $out/bench_nf_tolerant_driver_args.json 2>/dev/null
fi
ls $out
