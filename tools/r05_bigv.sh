#!/bin/bash
# PulseOsc at many voices: frames per lane (osc_fc) and store mode, one box.  usage: r05_bigv.sh "<voices...>" "<fc...>" "<store modes...>"
out=gpurun_out/r05/bigv_sweep.txt; mkdir -p gpurun_out/r05; : > $out
for v in ${1:-65536 1048576}; do
  steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  for fc in ${2:-0 2 8 16}; do for sm in ${3:-2 0 1}; do
    r=$(ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v --steps $steps --warmup 4 --no-cpu --no-config5 --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g frac %.3f ev/buffer %.2f us %s'%(d['value'], r['frac'], r['launch_ms_hip_events']/r['buffers_per_launch']*1e3, r['kernels_launched_per_step']))")
    echo "voices $v osc_fc=$fc store_mode=$sm: $r" >> $out
  done; done
done
cat $out
