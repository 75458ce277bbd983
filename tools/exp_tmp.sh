timeout 1200 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fuzz.py -m gpu -x -q -k "envelope or fuzz" 2>&1 | tail -3
export ZH_BENCH_ONLY="Envelope"
for V in 4096 32768 131072; do timeout 300 python tools/bench_modules.py $V 2>/dev/null | grep -v "^module"; done
