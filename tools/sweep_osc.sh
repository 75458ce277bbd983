#!/bin/bash
# GPU-box sweep of the chunked oscillator kernel's frames-per-lane (run through gpurun)
for fc in 4 8 16 32; do
  for sc in 0 1; do
    echo "== ZH_OSC_FC=$fc ZH_OSC_SCALAR=$sc"
    ZH_OSC_FC=$fc ZH_OSC_SCALAR=$sc python bench.py --steps 400 --warmup 40 --no-cpu "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.3e  ms/step %.4f  ev_ms %.4f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms_hip_events'], d['roofline']['frac']))"
  done
done
