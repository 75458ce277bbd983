#!/bin/bash
# (the forked-stream form this script measured was removed after this run: commit 5cdbd17 holds it; result: profiles/r05/ab_capture_lanes.txt)
# round 5: the concurrent-capture form of the headline against the in-order form, at the driver's arguments
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/test_graph.log
python -m pytest tests/test_gpu_tolerant.py -x -q -m gpu -k "longer_than_one_launch or jump_tables" 2>&1 | tail -15 > gpurun_out/r05/test_tol_long.log
for lanes in 1 2 3 4; do
  for rep in 1 2 3; do
    ZH_CAPTURE_LANES=$lanes python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 2>>gpurun_out/r05/lanes.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('lanes', $lanes, 'value %.4g' % d['value'], 'ms/step %.6f' % d['ms_per_step'], 'ev %.6f' % d['roofline']['launch_ms_hip_events'], 'frac %.4f' % d['roofline']['frac'], 'median', d.get('repeats', {}).get('ms_per_step_wall', {}).get('median'), d['parity'], d['config']['launch'][:60])
" >> gpurun_out/r05/lanes.txt
  done
done
for rep in 1 2 3; do
ZH_BENCH_IN_ORDER=1 python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 2>>gpurun_out/r05/lanes.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('in-order', 'value %.4g' % d['value'], 'ms/step %.6f' % d['ms_per_step'], 'ev %.6f' % d['roofline']['launch_ms_hip_events'], 'frac %.4f' % d['roofline']['frac'], 'median', d.get('repeats', {}).get('ms_per_step_wall', {}).get('median'), d['parity'])
" >> gpurun_out/r05/lanes.txt
done
python bench.py --no-cpu --no-config5 > gpurun_out/r05/bench_1000.json 2>>gpurun_out/r05/lanes.err
cat gpurun_out/r05/test_graph.log gpurun_out/r05/test_tol_long.log gpurun_out/r05/lanes.txt
