#!/bin/bash
out=gpurun_out/r05/bigv_sweep5.txt; mkdir -p gpurun_out/r05; : > $out
one() { v=$1; fc=$2; sm=$3; pad=$4; steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v --pad-voices $pad --steps $steps --warmup 4 --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f'%r['frac'], end=' ')"; }
for v in 65536 131072 262144; do for pad in 0 64 256 512 1024 4096; do for cfg in "3 1" "4 2"; do
  set -- $cfg
  echo -n "voices $v pad $pad osc_fc=$1 store_mode=$2: " >> $out
  for rep in 1 2; do one $v $1 $2 $pad >> $out; done; echo >> $out
done; done; done
cat $out
