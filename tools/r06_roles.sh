#!/bin/bash
# GPU box: the role-wave form of the generated kernels -- parity (composite recipes in every form, forced-role fuzz), then the
# script rows of tools/bench_modules.py with the form off / on at several voice counts.  -> gpurun_out/r06_roles/
set -u
O=gpurun_out/r06_roles; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_script_composites.py tests/test_gpu_script_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for V in ${VOICES:-4096 16384 32768 131072}; do
  for PC in 0 1; do
    ZH_BENCH_ONLY="script" ZH_FORMS="script_pc=$PC" timeout 600 python tools/bench_modules.py $V 2>&1 | grep -v amdgpu.ids > $O/modules_${V}_pc$PC.txt
  done
  paste -d'\n' $O/modules_${V}_pc0.txt $O/modules_${V}_pc1.txt | grep -E "script|voices" 
done
