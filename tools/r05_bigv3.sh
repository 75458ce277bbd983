#!/bin/bash
# the headline size: frames per lane x store mode on the coalesced graph, driver form and 1,000 steps
out=gpurun_out/r05/bigv_sweep3.txt; mkdir -p gpurun_out/r05; : > $out
one() { fc=$1; sm=$2; shift 2
  ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py "$@" --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f/%.3g'%(r['frac'], d['value']), end=' ')"; }
for fc in 2 3 4 6 8; do for sm in 1 2; do
  echo -n "4096 voices, --steps 20: osc_fc=$fc store_mode=$sm: " >> $out
  for rep in 1 2 3; do one $fc $sm --steps 20 --warmup 5 >> $out; done; echo >> $out
  echo -n "4096 voices, --steps 1000: osc_fc=$fc store_mode=$sm: " >> $out
  for rep in 1 2; do one $fc $sm --steps 1000 --warmup 100 >> $out; done; echo >> $out
done; done
for v in 16384 32768; do for fc in 3 4; do for sm in 1 2; do
  echo -n "$v voices, --steps 200: osc_fc=$fc store_mode=$sm: " >> $out
  for rep in 1 2 3; do one $fc $sm --voices $v --steps 200 --warmup 10 >> $out; done; echo >> $out
done; done; done
cat $out
