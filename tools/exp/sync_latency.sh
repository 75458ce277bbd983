#!/bin/bash
# GPU box: the host's share of the driver's 20-step region -- the same bench line with the HSA runtime polling its completion signals
# (HSA_ENABLE_INTERRUPT=0) instead of sleeping on an interrupt; three alternating rounds
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f us/step wall, %.2f us/launch HIP events, value %.4g, one-launch %.3g' % (d['ms_per_step']*1e3, r['launch_ms_hip_events']*1e3, d['value'], d.get('one_launch_per_step',{}).get('value',0)))"; }
for round in 1 2 3; do
  echo -n "default:                 "; python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-parity 2>/dev/null | line
  echo -n "HSA_ENABLE_INTERRUPT=0:  "; HSA_ENABLE_INTERRUPT=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-parity 2>/dev/null | line
done
