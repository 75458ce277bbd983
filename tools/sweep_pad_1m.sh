#!/bin/bash
# GPU box: the headline kernel at 1 Mi voices (4 GiB images) under several row paddings, and the voice counts on the way there
# -> gpurun_out/r04/sweep_pad_1m.txt
mkdir -p gpurun_out/r04
out=gpurun_out/r04/sweep_pad_1m.txt
: > $out
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'us/step', round(d['ms_per_step']*1e3,1), 'frac', round(d['roofline']['frac'],3), 'ring images', d['config']['ring_images'])" >> $out; }
for pad in 0 64 256 320 1024 4352 16640; do
  python3 bench.py --voices 1048576 --steps 40 --warmup 4 --no-cpu --no-parity --no-config5 --repeats 3 --pad-voices $pad 2>/dev/null | show "1048576 voices, pad $pad:"
done
for v in 262144 524288 786432; do
  python3 bench.py --voices $v --steps 40 --warmup 4 --no-cpu --no-parity --no-config5 --repeats 3 2>/dev/null | show "$v voices, default pad:"
done
cat $out
