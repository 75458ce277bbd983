#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of tools/bench_modules.py for the cases named by ZH_BENCH_ONLY
# usage: ZH_BENCH_ONLY="Curve" tools/prof_modules.sh <tag> <name> [voices]   -> gpurun_out/<tag>/<name>_kernel_stats.csv
tag=$1; name=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=$out/tmp_$name
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/bench_modules.py "$@" > $out/${name}_modules.txt 2>/dev/null
cp $d/*/*kernel_stats.csv $out/${name}_kernel_stats.csv
rm -rf $d
head -8 $out/${name}_kernel_stats.csv | cut -c1-160
