#!/bin/bash
# GPU-box sweep: store flavour x frames-per-lane for the chunked PulseOsc kernel.
# Prints rocprofv3 average kernel duration (ns) per variant.
cd /tmp && export TMPDIR=/tmp
for sm in 0 1 2 3; do
  for fc in 8 16; do
    d=$GRAFT_REPO_ROOT/gpurun_out/sweep_sm${sm}_fc${fc}
    ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu "$@" > /dev/null 2>&1
    echo "sm=$sm fc=$fc $(grep pulseosc $d/*/*kernel_stats.csv | awk -F, '{print "calls="$(NF-6)" avg_ns="$(NF-4)" min="$(NF-2)" max="$(NF-1)}')"
    rm -rf $d
  done
done
