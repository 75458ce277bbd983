timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_modules.py -m gpu -x -q 2>&1 | tail -15
timeout 1200 python tools/fuzz_many.py 300 2>&1 | tail -5
