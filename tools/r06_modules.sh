#!/bin/bash
# GPU box: the per-module table at 4,096 and 131,072 voices -> gpurun_out/r06_modules/modules_<V>.txt (profiles/r06/)
set -u
O=gpurun_out/r06_modules; mkdir -p $O
for V in ${VOICES:-4096 131072}; do
  ZH_BENCH_ONLY="${ONLY:-}" timeout 1500 python tools/bench_modules.py $V 2>&1 | grep -v amdgpu.ids > $O/modules_$V.txt
  tail -8 $O/modules_$V.txt
done
