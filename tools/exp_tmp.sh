timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/bench_modules.py 4096 2>/dev/null > gpurun_out/modules_4096.txt
timeout 600 python tools/bench_modules.py 131072 2>/dev/null > gpurun_out/modules_131072.txt
timeout 600 python tools/bench_modules.py 32768 2>/dev/null > gpurun_out/modules_32768.txt
