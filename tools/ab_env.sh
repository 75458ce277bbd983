#!/bin/bash
# GPU box: one bench.py workload with an environment switch of the library at two values, alternating on the same box.
# usage: tools/ab_env.sh NAME A B [bench args...]   -> "us per step" (median of the repeats' regions, wall and HIP events) four times each
name=$1; a=$2; b=$3; shift 3
run() { python bench.py "$@" --no-cpu --no-parity --no-config5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats') or {}
print('%.2f us/step (first region), median of %s regions: wall %.2f events %.2f' % (d['ms_per_step']*1e3, r.get('regions'), (r.get('ms_per_step_wall') or {}).get('median',0)*1e3, (r.get('ms_per_step_hip_events') or {}).get('median',0)*1e3))"; }
for i in 1 2 3 4; do
  echo "$name=$a : $(env $name=$a bash -c "$(declare -f run); run $*")"
  echo "$name=$b : $(env $name=$b bash -c "$(declare -f run); run $*")"
done
