#!/bin/bash
# round 5: the bounded non-blocking RCCL rendezvous, the self-arming multi-device tests (their one-device halves here), and
# `comm_host 1` a hundred times in a row (VERDICT r4 item 2 "done")
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_cpp_host.py tests/test_gpu_multidevice.py -x -q -m gpu -rs 2>&1 | tail -25 > gpurun_out/r05/test_comm.log
cat gpurun_out/r05/test_comm.log
: > gpurun_out/r05/comm_host_100.txt
pass=0
for i in $(seq 1 100); do
  t0=$(date +%s.%N)
  out=$(timeout 120 ./tests/cpp/comm_host 1 2>&1); rc=$?
  t1=$(date +%s.%N)
  echo "run $i rc $rc $(echo "$t1 - $t0" | bc) s: $(echo "$out" | tr '\n' ' ')" >> gpurun_out/r05/comm_host_100.txt
  [ $rc -eq 0 ] && pass=$((pass+1))
done
echo "comm_host 1: $pass / 100 passed" | tee -a gpurun_out/r05/comm_host_100.txt
timeout 900 python -m pytest tests/test_bench_launcher.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05/test_launcher.log
