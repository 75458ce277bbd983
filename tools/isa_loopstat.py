#!/usr/bin/env python3
"""Per kernel of a device assembly listing: the largest innermost loop's instruction count, how many of them are
exec-mask / branch instructions, VALU instructions and memory instructions -- the quickest way to see a per-lane `if` that
became an exec-mask region in a frame loop (profiles/r04/NOTES.md 5a).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -S --cuda-device-only \\
          -o /tmp/modules.s zang_amd/csrc/modules.hip
    tools/isa_loopstat.py /tmp/modules.s [substring of the mangled kernel name]

Frame loops are unrolled 8 times: divide by 8 for instructions per frame."""
import re
import subprocess
import sys

fn = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
txt = open(fn).read().split("\n")
names = [(m.group(1), n) for n, l in enumerate(txt) for m in [re.match(r"^(_Z\w+):\s+; @", l)] if m]
names.append(("END", len(txt)))
print("%-90s %s" % ("kernel", "(loop label, instructions, exec/branch, VALU, memory)"))
for (name, a), (_, b) in zip(names, names[1:]):
    if pat and pat not in name:
        continue
    body = txt[a:b]
    for k, l in enumerate(body):
        if "s_endpgm" in l:
            body = body[:k]
            break
    best = None
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", l)
        if not m:
            continue
        lab, last = m.group(1), None
        for k2 in range(k + 1, len(body)):
            if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", body[k2]):
                last = k2
        if last:
            ins = [x.split()[0] for x in body[k + 1:last + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
            if best is None or len(ins) > best[1]:
                best = (lab, len(ins), sum(1 for x in ins if "exec" in x or x.startswith("s_cbranch")),
                        sum(1 for x in ins if x.startswith("v_")), sum(1 for x in ins if x.startswith(("buffer_", "global_", "ds_", "scratch_"))))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%-90s %s" % (dem[:90], best))
