#!/bin/bash
# tools/probe_region.py under the HIP runtime's wait / graph knobs, one child process each (gpurun_out/r04/probe_region.txt)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/probe_region.txt
: > $out
run() { echo "## $*" >> $out; env "$@" python3 tools/probe_region.py >> $out 2>&1; }
run A=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run ROC_CPU_WAIT_FOR_SIGNAL=0
run ROC_CPU_WAIT_FOR_SIGNAL=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run AMD_DIRECT_DISPATCH=0
run PROBE_WARM=0
run PROBE_STEPS=40
cat $out
