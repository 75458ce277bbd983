#!/bin/bash
# GPU box: rocprofv3 kernel-trace summaries for every bench workload -> gpurun_out/profiles_rNN/
# usage: tools/collect_profiles.sh r01
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...: profiled + unprofiled on this box, per-dispatch min / p50 (tools/prof_one.sh)
  name=$1; shift
  bash $GRAFT_REPO_ROOT/tools/prof_one.sh profiles_$tag $name "$@" > /dev/null
}
run pulseosc4096
run pulseosc4096_driver_args --steps 20 --warmup 5      # the driver's own arguments: the 20-step region, its rehearsals and the batched-launch extra
run pulseosc65536 --voices 65536 --steps 100 --warmup 10
run pulseosc1M --voices 1048576 --steps 40 --warmup 4
run noise_filter4096 --workload noise_filter
run noise_filter131072 --workload noise_filter --voices 131072 --steps 50 --warmup 10
run nice4096 --workload nice --steps 96 --warmup 48
run nice131072 --workload nice --voices 131072 --steps 96 --warmup 48
run nice_mix131072 --workload nice_mix --voices 131072 --steps 96 --warmup 48
run nice_mix1M --workload nice_mix --voices 1048576 --steps 48 --warmup 48
run script131072 --workload script --voices 131072 --steps 96 --warmup 48
run noise_filter_fused131072 --workload noise_filter_fused --voices 131072 --steps 50 --warmup 10
run noise_filter_fused4096 --workload noise_filter_fused
run noise_filter_fused4096_tolerant --workload noise_filter_fused --tolerant
run noise_filter4096_tolerant --workload noise_filter --tolerant
run nice_mix4096 --workload nice_mix --steps 96 --warmup 48
ls $out
