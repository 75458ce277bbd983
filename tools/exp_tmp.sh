timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export ZH_BENCH_ONLY="Curve|Envelope|Portamento|Decimator|TriSawOsc freq"
for V in 4096 32768 131072; do timeout 300 python tools/bench_modules.py $V 2>/dev/null | grep -v "^module"; done
