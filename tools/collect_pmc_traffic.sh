#!/bin/bash
# GPU box: HBM traffic of the headline kernel from PMC counters, one counter per pass (--kernel-trace only), then
# tools/summarize_pmc.py -> gpurun_out/pmc_traffic/{pulseosc4096_pmc_*.csv, <tag>_pmc_pulseosc4096.json}
# usage: tools/collect_pmc_traffic.sh <tag> <commit>   (the commit is recorded in the json: .git does not travel to the GPU box)
tag=${1:-r02}; commit=${2:-unknown}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --eager --no-cpu --no-parity --no-config5 --repeats 0 > $out/run_$c.log 2>&1
  cp $d/*/*counter_collection.csv $out/pulseosc4096_pmc_$c.csv
done
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py k_osc_const4 $out/${tag}_pmc_pulseosc4096.json /tmp/pmc_WRITE_SIZE /tmp/pmc_FETCH_SIZE $commit
