timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for V in 4096 24576 131072; do timeout 300 python tools/bench_modules.py $V 2>/dev/null | grep "^#\|SineOsc\|PMOsc\|Sampler\|PulseOsc freq"; done
