#!/bin/bash
# GPU box: everything under profiles/<tag>/ in one call (copy gpurun_out/refresh_<tag>/ into profiles/<tag>/ afterwards).
# usage: tools/refresh_profiles.sh r01
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/refresh_$tag
mkdir -p $out/ubench
cd $root
bash tools/bench_all.sh $tag > $out/bench_all.log 2>&1
cp gpurun_out/bench_all_$tag/bench_unprofiled.jsonl gpurun_out/bench_all_$tag/song_60s.txt $out/
python bench.py > $out/bench_default_with_cpu_baseline.json 2> /dev/null
bash tools/collect_profiles.sh $tag > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_$tag/* $out/
bash tools/collect_pmc_valu.sh $tag > $out/collect_pmc_valu.log 2>&1
mkdir -p $out/pmc_valu && cp gpurun_out/pmc_valu_$tag/*.csv gpurun_out/pmc_valu_$tag/summary.json $out/pmc_valu/
bash tools/collect_pmc_traffic_all.sh $(python3 -c "import json; print(json.load(open('zang_amd/build_info.json')).get('commit') or 'unknown')" 2>/dev/null || echo unknown) > $out/collect_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic_*.json $out/
cd tools/ubench
for b in valu_ops sclk_probe; do hipcc --offload-arch=gfx950 -O3 $b.hip -o /tmp/$b && timeout 300 /tmp/$b > $out/ubench/${b}_raw.txt 2>&1; done
ls -R $out | head -60
