#!/bin/bash
# CPU only: the HOST side of libzang_hip.so under AddressSanitizer + UBSan.  Every csrc/*.hip is compiled host-only
# (clang -x hip --cuda-host-only: the kernels become launch stubs) against tools/host_asan/hip_stub.cpp, a HIP runtime that tracks
# allocations, streams and graphs and runs nothing.  Then
#   1. tools/host_asan/harness.cpp: random sequences of begin_capture / paint / end / launch / destroy over the C ABI (held-back
#      batches, flips, the pipelined recording, modules destroyed before their graphs, the context before its graphs);
#   2. tests/test_scheduler.py (the reference's 8 scheduler cases, the 33-impulse overflow, out-of-order events) and tests/test_abi.py
#      against the sanitized library (ZANG_HIP_LIB), python running with the ASan runtime preloaded.
# usage: tools/host_asan.sh [rounds [seed]]     exit code 0 = no sanitizer report, nothing leaked on the fake device
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
work=${HOST_ASAN_DIR:-$(mktemp -d)}
mkdir -p "$work"
CXX=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math -w -I/opt/rocm/include -I$root/include"
pids=()
for f in "$root"/zang_amd/csrc/*.hip; do
  o="$work/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find "$root/zang_amd/csrc" "$root/include" -newer "$o" \( -name '*.h' -o -name '*.hpp' -o -name '*.inc' \) | head -1)" ]; then
    $CXX -x hip --cuda-host-only --offload-arch=gfx950 $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# (the host stubs of a TU's kernels name its device image, which a host-only compile does not have)
nm -u "$work"/*.o | awk '/__hip_fatbin_/{print $2}' | sort -u | sed 's/.*/char &[8];/' > "$work/fatbins.c"
$CXX $FLAGS -D__HIP_PLATFORM_AMD__ -c "$root/tools/host_asan/hip_stub.cpp" -o "$work/hip_stub.o"
$CXX -x c -fPIC -c "$work/fatbins.c" -o "$work/fatbins.o"
objs=$(ls "$work"/*.o | grep -v harness.o)
$CXX -shared -fsanitize=address,undefined -shared-libsan $objs -ldl -o "$work/libzang_hip_asan.so"
$CXX $FLAGS -D__HIP_PLATFORM_AMD__ -c "$root/tools/host_asan/harness.cpp" -o "$work/harness.o"
$CXX -fsanitize=address,undefined -shared-libsan "$work/harness.o" -L"$work" -lzang_hip_asan -Wl,-rpath,"$work" -o "$work/harness"
rt=$(dirname "$($CXX -print-file-name=libclang_rt.asan-x86_64.so)")
export LD_LIBRARY_PATH="$rt:${LD_LIBRARY_PATH:-}"
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
ASAN_OPTIONS=detect_leaks=1 "$work/harness" "${1:-300}" "${2:-1}"
# the scheduler's unit tests and the ABI checks against the same build (python leaks by design: leak detection off)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$rt/libclang_rt.asan-x86_64.so" ZANG_HIP_LIB="$work/libzang_hip_asan.so" \
  python3 -m pytest -q -x -p no:cacheprovider "$root/tests/test_scheduler.py" "$root/tests/test_abi.py" -m "not gpu" 2>&1 | tail -3
[ -n "${HOST_ASAN_DIR:-}" ] || rm -rf "$work"
