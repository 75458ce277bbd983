export ZH_BENCH_ONLY="NiceInstrument"
for i in 1 2; do for V in 131072 262144; do echo "V $V new"; timeout 300 python tools/bench_modules.py $V 2>/dev/null | grep -v "^module\|^#"; echo "V $V old"; ZANG_HIP_LIB=$PWD/zang_amd/libold.so timeout 300 python tools/bench_modules.py $V 2>/dev/null | grep -v "^module\|^#"; done; done
