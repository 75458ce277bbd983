#!/bin/bash
# GPU box: FilteredSawtooth in the role-wave form, 64 KiB LDS budget (16-frame tiles) against 128 KiB (32-frame tiles)
set -u
O=gpurun_out/r06_roles; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_script_composites.py -x -q -m gpu -k "4096" > $O/pytest2.log 2>&1
tail -3 $O/pytest2.log
for V in 4096 16384; do
for FORMS in 1 5; do
  echo "V=$V ZH_SCRIPT_FORMS=$FORMS"
  ZH_SCRIPT_FORMS=$FORMS ZH_BENCH_ONLY="FilteredSawtooth" ZH_FORMS="script_pc=1" timeout 600 python tools/bench_modules.py $V 2>&1 | grep -E "script|rror"
done; done
