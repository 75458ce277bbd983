timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -12
