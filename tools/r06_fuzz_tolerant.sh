#!/bin/bash
# GPU box: tools/fuzz_tolerant.py over N seeds from FIRST (default 2,500 from 60001), then 150 FilteredEchoes seeds -> gpurun_out/r06_fuzz/
N=${1:-2500}; FIRST=${2:-60001}
O=gpurun_out/r06_fuzz; mkdir -p $O
( time timeout 2300 python tools/fuzz_tolerant.py $N $FIRST ) 2>&1 | grep -v amdgpu.ids | tail -25 > $O/fuzz_tolerant_${FIRST}.txt
if [ "${ECHOES:-1}" = "1" ]; then ( time timeout 400 python tools/fuzz_tolerant.py 100 $((FIRST + 500000)) echoes ) 2>&1 | grep -v amdgpu.ids | tail -8 >> $O/fuzz_tolerant_${FIRST}.txt; fi
cat $O/fuzz_tolerant_${FIRST}.txt
