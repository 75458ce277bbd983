#!/bin/bash
# GPU box: the tolerant contract after the near-zero-cutoff escape (kTpExactCutBelow) -- the gating tests, then config 3 tolerant pipelined
# with 32- / 64- / 128-frame chunks (nf_tp_pipe_frames), alternating.  -> gpurun_out/r06_tolerant/
set -u
O=gpurun_out/r06_tolerant; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tolerant.py tests/test_gpu_graph.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f us/step wall, %.2f us/launch HIP events, value %.3g, %s, parity %s' % (d['ms_per_step']*1e3, r['launch_ms_hip_events']*1e3, d['value'], r['kernels_launched_per_step'], (d.get('parity_of_a_graph_replay') or {}).get('tolerant')))"; }
{
echo "# config 3 tolerant, recorded pipelined (bench.py --workload noise_filter_fused --tolerant --steps 20 --warmup 5), three alternating rounds"
for round in 1 2 3; do for fr in 0 64 128 256; do
  echo -n "nf_tp_pipe_frames=$fr: "; ZH_FORMS=nf_tp_pipe_frames=$fr python bench.py --workload noise_filter_fused --tolerant --steps 20 --warmup 5 --no-cpu 2>/dev/null | line
done; done
} > $O/ab_nf_pipe_frames.txt 2>&1
cat $O/ab_nf_pipe_frames.txt
