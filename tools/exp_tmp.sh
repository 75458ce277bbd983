timeout 1200 python -m pytest tests/test_gpu_composite.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for V in 131072 1048576; do
python bench.py --workload nice_mix --voices $V --steps 96 --warmup 48 --no-cpu --no-config5 --repeats 0 2>/dev/null | tail -1 | cut -c1-200
ZANG_HIP_LIB=$PWD/zang_amd/libold.so python bench.py --workload nice_mix --voices $V --steps 96 --warmup 48 --no-cpu --no-config5 --repeats 0 2>/dev/null | tail -1 | cut -c1-200
done; done
