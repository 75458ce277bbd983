#!/bin/bash
# the setup form (paints not flagged ZH_PAINT_PARAMS_UNCHANGED: constants computed per lane): frames per lane x store mode
out=gpurun_out/r05/bigv_setup.txt; mkdir -p gpurun_out/r05; : > $out
one() { v=$1; fc=$2; sm=$3; steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  ZH_BENCH_NO_TABLE=1 ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v --steps $steps --warmup 4 --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f'%r['frac'], end=' ')"; }
for v in 4096 16384 65536 131072 524288 1048576; do for cfg in "4 2" "3 1" "4 1" "3 2" "6 2" "8 2"; do
  set -- $cfg
  echo -n "setup form, voices $v osc_fc=$1 store_mode=$2: " >> $out
  for rep in 1 2; do one $v $1 $2 >> $out; done; echo >> $out
done; done
cat $out
