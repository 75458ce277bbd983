#!/usr/bin/env python3
"""Generate the 2/pi (24-bit chunks) and pi/2 (24-bit chunks) tables used by the
Payne-Hanek large-argument reduction (published musl/FreeBSD `__rem_pio2_large`
algorithm) from first principles with big-integer arithmetic (Machin's formula).

Usage: python tools/gen_pio2_tables.py > /tmp/tables.h
The output is pasted into oracle/zmath_ref.h and zang_amd/csrc/zmath.hip.h; a CPU
test (tests/test_oracle_math.py) re-derives the tables and compares.
"""

def arctan_inv(x, bits):
    one = 1 << bits
    term = one // x
    s = term
    x2 = x * x
    n = 1
    sign = -1
    while term:
        term //= x2
        n += 2
        s += sign * (term // n)
        sign = -sign
    return s


def tables(n_ipio2=66, n_pio2=8):
    bits = 4000
    pi = 4 * (4 * arctan_inv(5, bits) - arctan_inv(239, bits))  # pi * 2^bits
    N = 24 * n_ipio2
    twoopi = (2 << (bits + N)) // pi
    ipio2 = [(twoopi >> (N - 24 * (i + 1))) & 0xFFFFFF for i in range(n_ipio2)]
    rem = pi >> 1
    pio2 = []
    for k in range(n_pio2):
        shift = bits - 23 - 24 * k
        c = rem >> shift
        rem -= c << shift
        pio2.append(c * 2.0 ** (-23 - 24 * k))
    return ipio2, pio2


if __name__ == "__main__":
    ipio2, pio2 = tables()
    print("static const int32_t ZM_IPIO2[%d] = {" % len(ipio2))
    for i in range(0, len(ipio2), 6):
        print("  " + " ".join("0x%06X," % e for e in ipio2[i:i + 6]))
    print("};")
    print("static const double ZM_PIO2[%d] = {" % len(pio2))
    for v in pio2:
        print("  %s," % float.hex(v))
    print("};")
