#!/bin/bash
# GPU box: the round's new tests, then the A/Bs that set defaults: Distortion one voice per lane against the chunked form by voice count,
# every script module lane form against role-wave form.  -> gpurun_out/r06_checks/
set -u
O=gpurun_out/r06_checks; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tolerant.py tests/test_gpu_graph.py tests/test_gpu_dispatch.py tests/test_gpu_script_composites.py -x -q -m gpu -k "near_zero or dipping or refuses or distortion or stateless or filtered_sawtooth or boundaries" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
{
echo "# Distortion, one voice per lane (k_distortion: distortion_rows_min=2^30) against four voices per lane with the constants once per workgroup (k_distortion_chunks: =0); us per paint, HBM TB/s"
for V in 4096 16384 32768 65536 131072; do for R in 1073741824 0; do
  echo -n "voices $V distortion_rows_min=$R: "; ZH_FORMS=distortion_rows_min=$R ZH_BENCH_ONLY="Distortion" python tools/bench_modules.py $V 2>/dev/null | grep Distortion | awk '{printf "%s %s %s us %s TB/s;  ", $1, $2, $(NF-2), $NF}'; echo
done; done
} > $O/ab_distortion.txt 2>&1; cat $O/ab_distortion.txt
for V in 4096 32768; do python tools/exp/role_ab.py $V 2>&1 | grep -v amdgpu.ids > $O/role_ab_$V.txt; cat $O/role_ab_$V.txt; done
