#!/bin/bash
# nice_mix stereo paints coalesced in a capture: tests, then the config-5 shard line both ways.
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_graph.py tests/test_gpu_composite.py tests/test_bench_launcher.py tests/test_gpu_comm.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/nicemix_tests.log
for i in 1 2 3; do
  python bench.py --workload nice_mix --voices 131072 --steps 96 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/nicemix_coalesce_$i.json
  ZH_BENCH_IN_ORDER=1 python bench.py --workload nice_mix --voices 131072 --steps 96 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/nicemix_inorder_$i.json
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/nicemix_headline.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/nicemix_*.json")):
    try:
        d=json.loads(open(f).read()); r=d["roofline"]
        print(f, "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "ev %.4f"%(r["launch_ms_hip_events"]/r["buffers_per_launch"]), d["config"]["launch"][:90], (d.get("config5_shard") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r05/nicemix_tests.log
