#!/usr/bin/env python3
"""Exhaustive check behind csrc/zmath.hip.h's folded sinf/cosf reduction: for EVERY float with |x| <= 9pi/4
(1.09e9 bit patterns), rint(|x| * 2/pi) in float64 equals k, the number of musl's ladder thresholds
(sinf.c: 0x3f490fda, 0x4016cbe3, 0x407b53d1, 0x40afeddf) that |x| exceeds.  Takes a minute or two; the test
suite checks the step points only (tests/test_oracle_math.py), which covers the range by monotonicity."""
import numpy as np

T = [0x3f490fda, 0x4016cbe3, 0x407b53d1, 0x40afeddf]
INVPIO2 = np.float64(6.36619772367581382433e-01)
END = 0x40e231d5 + 1


def main():
    bad = 0
    for lo in range(0, END, 1 << 24):
        ix = np.arange(lo, min(lo + (1 << 24), END), dtype=np.uint32)
        k = np.rint(ix.view(np.float32).astype(np.float64) * INVPIO2).astype(np.int64)
        bad += int((k != sum((ix > t).astype(np.int64) for t in T)).sum())
    print("floats checked: %d, mismatches: %d" % (END, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
