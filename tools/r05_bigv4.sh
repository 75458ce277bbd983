#!/bin/bash
out=gpurun_out/r05/bigv_sweep4.txt; mkdir -p gpurun_out/r05; : > $out
one() { v=$1; fc=$2; sm=$3; steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v --steps $steps --warmup 4 --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f'%r['frac'], end=' ')"; }
for v in 49152 65536 98304 131072 196608 262144 524288 786432; do for cfg in "3 1" "4 2" "3 2" "4 1" "2 1"; do
  set -- $cfg
  echo -n "voices $v osc_fc=$1 store_mode=$2: " >> $out
  for rep in 1 2 3; do one $v $1 $2 >> $out; done; echo >> $out
done; done
cat $out
