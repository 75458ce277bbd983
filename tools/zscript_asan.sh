#!/bin/bash
# CPU only: the C++ zangscript compiler (csrc/zscript_front.hip, zscript_emit.hip -- plain host C++) built with
# AddressSanitizer + UBSan and fed 6,000 random mutations of the test script through the C ABI.
# (GPU sanitizers are not available on this pool; the compiler has no device code.)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
work=$(mktemp -d)
python3 - "$root" "$work" <<'PY'
import random, sys
root, work = sys.argv[1], sys.argv[2]
text = open(root + '/tests/golden/script_modules.txt').read()
parts = text.split("\n\n")
rng = random.Random(99)
tokens = ["(", ")", ",", "=", "*", "+", "-", "/", ".", ":", "begin", "end", "out", "feedback", "delay", "from", "defmodule", "defcurve",
          "deftrack", "true", "false", "pi", "sin", "max", "SineOsc", "Envelope", "freq", "note_on", "0.5", "3", "x", ".cubed", ".low_pass",
          "cob", "constant", "waveform", "\n", " "]
cases = [text]
for _ in range(6000):
    src = "\n\n".join(rng.sample(parts, rng.randint(1, 3)))
    for _ in range(rng.randint(1, 4)):
        k, pos = rng.random(), rng.randrange(len(src) + 1)
        if k < 0.4: src = src[:pos] + rng.choice(tokens) + src[pos:]
        elif k < 0.7: src = src[:pos] + src[pos + rng.randint(1, 12):]
        else:
            a = rng.randrange(len(src)); src = src[:pos] + src[a:min(len(src), a + rng.randint(1, 30))] + src[pos:]
    cases.append(src)
open(work + '/cases.txt', 'w').write("\x01".join(cases))
PY
cat > "$work/harness.cpp" <<'CPP'
#include <stdio.h>
#include <fstream>
#include <sstream>
#include <string>
#include "zang_hip.h"
int main(int argc, char **argv) {
    std::ifstream f(argv[1]); std::stringstream ss; ss << f.rdbuf(); std::string all = ss.str();
    size_t pos = 0, n = 0, ok = 0;
    while (pos <= all.size()) {
        size_t e = all.find('\x01', pos); if (e == std::string::npos) e = all.size();
        std::string src = all.substr(pos, e - pos); pos = e + 1; n++;
        zh_zscript *z = nullptr; char err[4096];
        if (zh_zscript_compile(src.c_str(), "f.txt", 3, &z, err, sizeof err) == 0) {
            char *a = nullptr, *b = nullptr;
            zh_zscript_generate_zig(z, &a); zh_zscript_generate_hip(z, nullptr, 0, &b);
            char nm[64], er[256]; unsigned w, no, np;
            for (unsigned i = 0; i < zh_zscript_module_count(z); i++) zh_zscript_module_info(z, i, nm, sizeof nm, &w, &no, &np, er, sizeof er);
            zh_zscript_free_text(a); zh_zscript_free_text(b); zh_zscript_destroy(z); ok++;
        }
    }
    printf("%zu cases, %zu compiled, no sanitizer report\n", n, ok);
    return 0;
}
CPP
g++ -x c++ -std=c++17 -g -O1 -w -fsanitize=address,undefined -fno-omit-frame-pointer -I"$root/include" \
    "$root/zang_amd/csrc/zscript_front.hip" "$root/zang_amd/csrc/zscript_emit.hip" "$work/harness.cpp" -o "$work/harness"
ASAN_OPTIONS=detect_leaks=1 "$work/harness" "$work/cases.txt"
rm -rf "$work"
