#!/bin/bash
# repeats of the frames-per-lane sweep where the first pass showed a gap
out=gpurun_out/r05/bigv_sweep2.txt; mkdir -p gpurun_out/r05; : > $out
one() { v=$1; fc=$2; sm=$3; steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v --steps $steps --warmup 4 --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f'%r['frac'], end=' ')"; }
for v in 65536 262144 1048576; do for fc in 2 3 4 5 6; do for sm in 1 2; do
  echo -n "voices $v osc_fc=$fc store_mode=$sm: " >> $out
  for rep in 1 2 3; do one $v $fc $sm >> $out; done; echo >> $out
done; done; done
cat $out
