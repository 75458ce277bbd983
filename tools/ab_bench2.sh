#!/bin/bash
# GPU box: bench.py lines with the tree's library and another build, alternating.  usage: ab_bench2.sh <other.so> -- <bench args>
other=$1; shift; shift
for rep in 1 2 3; do
  for side in other tree; do
    if [ $side = other ]; then export ZANG_HIP_LIB=$other; else unset ZANG_HIP_LIB; fi
    python bench.py "$@" --no-cpu --no-config5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$side', '%.4g'%d['value'], 'ms/step %.5f'%d['ms_per_step'], 'parity', (d.get('parity') or {}).get('bitexact'))"
  done
done
