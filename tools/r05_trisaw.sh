#!/bin/bash
out=gpurun_out/r05/trisaw_sweep.txt; mkdir -p gpurun_out/r05; : > $out
for v in 16384 65536 131072 524288; do
  for cfg in "0 -" "3 1" "5 1" "7 1" "9 1" "4 1" "8 1" "7 2" "16 1"; do
    set -- $cfg
    if [ "$2" = "-" ]; then r=$(ZH_BENCH_ONLY="TriSawOsc const|PulseOsc const" python tools/bench_modules.py $v 2>/dev/null | grep -E "TriSawOsc|PulseOsc" | awk '{print $(NF-2), $NF}' | tr '\n' ' ')
    else r=$(ZH_STORE_MODE=$2 ZH_FORMS=osc_fc=$1 ZH_BENCH_ONLY="TriSawOsc const" python tools/bench_modules.py $v 2>/dev/null | grep -E "TriSawOsc" | awk '{print $(NF-2), $NF}' | tr '\n' ' '); fi
    echo "voices $v osc_fc=$1 store_mode=$2: $r" >> $out
  done
done
cat $out
