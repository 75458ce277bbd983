#!/bin/bash
# after the change: defaults (no switches) over the voice counts, PulseOsc and TriSawOsc untouched sizes; parity tests first
out=gpurun_out/r05/bigv_after.txt; mkdir -p gpurun_out/r05; : > $out
python -m pytest tests/test_gpu_osc.py tests/test_gpu_dispatch.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 >> $out
one() { v=$1; steps=$((6553600 / v)); [ $steps -lt 12 ] && steps=12
  python bench.py --voices $v --steps $steps --warmup 4 --no-cpu --no-config5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f/%s'%(r['frac'], d['parity']['bitexact']), end=' ')"; }
for v in 4096 16384 65536 131072 524288 655360 786432 1048576; do
  echo -n "voices $v defaults: " >> $out
  for rep in 1 2 3; do one $v >> $out; done; echo >> $out
done
cat $out
