timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ZH_NICE_PC4_MAX=0 timeout 600 python -m pytest tests/test_gpu_composite.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
ZH_NICE_PC4_MAX=100000 timeout 600 python -m pytest tests/test_gpu_composite.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
