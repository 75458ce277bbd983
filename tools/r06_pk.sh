#!/bin/bash
# GPU box: the filter recurrence with packed products (tools/ubench/pk_svf.hip), then config 3 exact (k_noise_filter_ring) with the
# library in the tree against another build (zang_amd/libzang_hip_base.so: `git stash; make -C zang_amd/csrc; cp zang_amd/libzang_hip.so
# zang_amd/libzang_hip_base.so; git stash pop; make -C zang_amd/csrc` before the call), alternating.  -> gpurun_out/r06_pk/
set -u
O=gpurun_out/r06_pk; mkdir -p $O
timeout 120 tools/ubench/pk_svf > $O/ubench_pk_svf.txt 2>&1; cat $O/ubench_pk_svf.txt
us() { python bench.py "$@" --no-cpu --no-config4 --no-config5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/step, parity %s' % (d['ms_per_step']*1e3, d.get('parity')))"; }
{
for round in 1 2 3; do
  echo -n "tree  noise_filter_fused 4096: "; us --workload noise_filter_fused --steps 200 --warmup 20
  echo -n "other noise_filter_fused 4096: "; ZANG_HIP_LIB=$PWD/zang_amd/libzang_hip_base.so us --workload noise_filter_fused --steps 200 --warmup 20
done
for round in 1 2; do
  echo -n "tree  nice 4096: "; us --workload nice --steps 96 --warmup 48
  echo -n "other nice 4096: "; ZANG_HIP_LIB=$PWD/zang_amd/libzang_hip_base.so us --workload nice --steps 96 --warmup 48
  echo -n "tree  nice 131072: "; us --workload nice --voices 131072 --steps 96 --warmup 48
  echo -n "other nice 131072: "; ZANG_HIP_LIB=$PWD/zang_amd/libzang_hip_base.so us --workload nice --voices 131072 --steps 96 --warmup 48
done
} > $O/ab_lib.txt 2>&1
cat $O/ab_lib.txt
