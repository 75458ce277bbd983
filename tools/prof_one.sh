#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of one bench command -> gpurun_out/<tag>/<name>_kernel_stats.csv + the line it printed
# usage: tools/prof_one.sh <tag> <name> [bench args...]
tag=$1; name=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=$out/tmp_$name
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu --no-config5 --repeats 0 > $out/${name}_bench.json 2>/dev/null
cp $d/*/*kernel_stats.csv $out/${name}_kernel_stats.csv
rm -rf $d
tail -1 $out/${name}_bench.json > $out/${name}_bench.json.tmp; mv $out/${name}_bench.json.tmp $out/${name}_bench.json
head -4 $out/${name}_kernel_stats.csv
