#!/bin/bash
# the one-kernel-node-per-step form at the headline size: store mode x frames per lane
one() { fc=$1; sm=$2; shift 2
  ZH_BENCH_IN_ORDER=1 ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py "$@" --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f/%.3g'%(r['frac'], d['value']), end=' ')"; }
for fc in 4 3 2; do for sm in 2 1 0; do
  echo -n "in order, osc_fc=$fc store_mode=$sm, --steps 20: "; for rep in 1 2 3; do one $fc $sm --steps 20 --warmup 5; done; echo
  echo -n "in order, osc_fc=$fc store_mode=$sm, --steps 1000: "; for rep in 1 2; do one $fc $sm --steps 1000 --warmup 100; done; echo
done; done
