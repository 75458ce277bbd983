#!/bin/bash
# round 5: the coalescing capture of the headline at the driver's arguments (and the default 1000 steps)
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_graph.py tests/test_gpu_osc.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/test_graph.log
cat gpurun_out/r05/test_graph.log
for rep in 1 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 2>>gpurun_out/r05/co.err > gpurun_out/r05/bench_co_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r05/bench_co_$rep.json"))
io = d.get("one_launch_per_step", {})
print("coalesce value %.4g ms/step %.6f ev/launch %.6f frac %.4f median %s | in-order value %.4g frac %.4f | %s | %s" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms_hip_events"], d["roofline"]["frac"], d.get("repeats", {}).get("ms_per_step_wall", {}).get("median"), io.get("value", 0), io.get("frac", 0), d["parity"]["bitexact"], d["config"]["launch"][:90]))
PY
done | tee gpurun_out/r05/coalesce.txt
python bench.py --no-cpu --no-config5 > gpurun_out/r05/bench_co_1000.json 2>>gpurun_out/r05/co.err
python -c "
import json
d = json.load(open('gpurun_out/r05/bench_co_1000.json'))
print('1000 steps: value %.4g frac %.4f' % (d['value'], d['roofline']['frac']), d['config']['launch'])
" | tee -a gpurun_out/r05/coalesce.txt
tail -5 gpurun_out/r05/co.err
