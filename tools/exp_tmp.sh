timeout 1200 python -m pytest tests/test_gpu_composite.py tests/test_gpu_fullsize.py tests/test_song.py -m gpu -x -q 2>&1 | tail -3
export ZH_BENCH_ONLY="PMOsc"
for V in 1024 4096 16384 32768 131072; do timeout 300 python tools/bench_modules.py $V 2>/dev/null | grep -v "^module"; done
for R in 16 64; do echo "ranges $R"; ZH_PMOSC_RANGES=$R timeout 300 python tools/bench_modules.py 4096 2>/dev/null | grep -v "^module\|^#"; done
