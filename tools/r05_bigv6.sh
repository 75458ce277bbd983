#!/bin/bash
out=gpurun_out/r05/bigv_sweep6.txt; mkdir -p gpurun_out/r05; : > $out
one() { v=$1; fc=$2; sm=$3; pad=$4; steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v --pad-voices $pad --steps $steps --warmup 4 --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f'%r['frac'], end=' ')"; }
for v in 65536 131072 196608 262144 393216 524288 786432 1048576; do for pad in 0 256 1024 4096; do
  echo -n "voices $v pad $pad (3,NT): " >> $out
  for rep in 1 2; do one $v 3 1 $pad >> $out; done; echo >> $out
done; done
w() { name=$1; shift; echo -n "$name: " >> $out; for rep in 1 2; do python bench.py "$@" --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], end=' ')" >> $out; done; echo >> $out; }
for pad in 256 4096 0; do
  w "nice 131072 pad $pad" --workload nice --voices 131072 --steps 48 --warmup 48 --pad-voices $pad
  w "noise_filter_fused 65536 pad $pad" --workload noise_filter_fused --voices 65536 --steps 40 --warmup 5 --pad-voices $pad
  w "noise_filter 65536 pad $pad" --workload noise_filter --voices 65536 --steps 40 --warmup 5 --pad-voices $pad
  w "script 131072 pad $pad" --workload script --voices 131072 --steps 48 --warmup 48 --pad-voices $pad
done
cat $out
