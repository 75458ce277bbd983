timeout 1200 python -m pytest tests/test_zangscript.py tests/test_song.py -m gpu -x -q 2>&1 | tail -3
python bench.py --workload script --voices 131072 --steps 20 --warmup 4 --no-cpu --no-config5 --repeats 0 2>/dev/null | tail -1 | cut -c1-400
python bench.py --workload script --voices 4096 --steps 20 --warmup 4 --no-cpu --no-config5 --repeats 0 2>/dev/null | tail -1 | cut -c1-400
