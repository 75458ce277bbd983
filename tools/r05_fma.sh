#!/bin/bash
# the fused-multiply-add mixdown (ZH_PAINT_TOLERANT above nice_tp_max voices): tests, then config 5's shard exact / tolerant, alternating
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_tolerant.py tests/test_gpu_graph.py tests/test_gpu_composite.py -x -q -m gpu -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | grep -E "passed|failed|Error|error|assert|fma mixdown" | tail -20 > gpurun_out/r05/fma_tests.log
for i in 1 2 3; do
  python bench.py --workload nice_mix --voices 131072 --steps 96 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/fma_exact_$i.json
  python bench.py --workload nice_mix --voices 131072 --steps 96 --warmup 5 --tolerant 2>/dev/null | tail -1 > gpurun_out/r05/fma_tolerant_$i.json
done
ZH_FORMS=nice_mix_fma=0 python bench.py --workload nice_mix --voices 131072 --steps 96 --warmup 5 --tolerant 2>/dev/null | tail -1 > gpurun_out/r05/fma_tolerant_off.json
python bench.py --workload nice_mix --voices 1048576 --steps 48 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/fma_exact_1M.json
python bench.py --workload nice_mix --voices 1048576 --steps 48 --warmup 5 --tolerant 2>/dev/null | tail -1 > gpurun_out/r05/fma_tolerant_1M.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/fma_*.json")):
    try:
        d=json.loads(open(f).read()); r=d["roofline"]
        print(f, "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "ev %.4f"%(r["launch_ms_hip_events"]/r["buffers_per_launch"]), r.get("kernels_launched_per_step"), json.dumps(d.get("parity"))[:300])
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r05/fma_tests.log
