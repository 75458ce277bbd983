"""CPU: the oracle's libm / PRNG restatements (oracle/zmath_ref.h) against correctly rounded
values computed in float64 / exact big-integer argument reduction.  The reference takes these
from the un-vendored Zig std library: parity unpinned, accuracy pinned here."""
import math
import os
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gen_pio2_tables  # noqa: E402


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _run(L, oracle, name, xs):
    y = np.zeros_like(xs)
    getattr(L, name)(oracle.fptr(xs), oracle.fptr(y), xs.size)
    return y


def test_sin_cos_medium_range(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-900, 900, 200000), rng.uniform(-8, 8, 100000), rng.uniform(-1e-3, 1e-3, 1000),
                         np.array([0.0, -0.0, np.pi, np.pi / 2, 3 * np.pi / 4, 7 * np.pi / 4])]).astype(np.float32)
    for name, fn in (("zo_math_sinf_n", np.sin), ("zo_math_cosf_n", np.cos)):
        y = _run(L, oracle, name, xs)
        ref = fn(xs.astype(np.float64)).astype(np.float32)
        d = _ulps(y, ref)
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (name, d.max(), (d != 0).mean())


def test_sin_cos_huge_arguments_exact_reduction(oracle):
    """|x| >= 2^28*pi/2 takes the Payne-Hanek path; reduce exactly with a 600-bit pi."""
    L = oracle.lib()
    bits = 600
    pi_int = 4 * (4 * gen_pio2_tables.arctan_inv(5, bits) - gen_pio2_tables.arctan_inv(239, bits))
    twopi = Fraction(2 * pi_int, 1 << bits)
    rng = np.random.default_rng(3)
    xs = (rng.uniform(1, 2, 1500) * 2.0 ** rng.integers(29, 127, 1500)).astype(np.float32)
    xs[::2] *= -1

    def reduce(x):
        fr = Fraction(float(x))
        return float(fr - math.floor(fr / twopi) * twopi)

    red = np.array([reduce(x) for x in xs])
    for name, fn in (("zo_math_sinf_n", np.sin), ("zo_math_cosf_n", np.cos)):
        y = _run(L, oracle, name, xs)
        assert _ulps(y, fn(red).astype(np.float32)).max() <= 1
    assert math.isnan(L.zo_math_sinf(float("inf"))) and math.isnan(L.zo_math_cosf(float("nan")))


def test_sin_reduction_rint_equals_musl_ladder():
    """csrc/zmath.hip.h replaces musl's magnitude ladder (k = how many of four thresholds |x| exceeds) by
    rint(|x| * 2/pi) in float64.  Both are non-decreasing step functions of |x|, so they agree on all of
    [0, 9pi/4] iff they agree at every step point and at the ends (tools/check_sin_reduction.py sweeps all
    1.09e9 floats); a random sample rides along."""
    T = [0x3f490fda, 0x4016cbe3, 0x407b53d1, 0x40afeddf]
    invpio2 = np.float64(6.36619772367581382433e-01)
    rng = np.random.default_rng(5)
    ix = np.concatenate([np.array([0, 1, 0x00800000, 0x40e231d5], np.uint32),
                         np.array([t + d for t in T for d in range(-3, 4)], np.uint32),
                         rng.integers(0, 0x40e231d6, 1 << 20, dtype=np.uint64).astype(np.uint32)])
    k = np.rint(ix.view(np.float32).astype(np.float64) * invpio2).astype(np.int64)
    assert np.array_equal(k, sum((ix > t).astype(np.int64) for t in T))
    # and the 1.5*2^52 trick of the medium leaf IS rint: |x * invpio2| < 2^28 there
    v = rng.uniform(-2.0 ** 28, 2.0 ** 28, 1 << 20) * invpio2
    toint = np.float64(1.5) / np.float64(2.220446049250313e-16)
    assert np.array_equal(v + toint - toint, np.rint(v))


def test_atan_pow_exp_log(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-30, 30, 200000), rng.uniform(-1e8, 1e8, 1000)]).astype(np.float32)
    assert _ulps(_run(L, oracle, "zo_math_atanf_n", xs), np.arctan(xs.astype(np.float64)).astype(np.float32)).max() <= 1
    ys = rng.uniform(-2.0, 6.0, 200000).astype(np.float32)          # Distortion: ingain*8-2, ingain in [0,1]
    got = _run(L, oracle, "zo_math_pow2f_n", ys)
    assert _ulps(got, np.exp2(ys.astype(np.float64)).astype(np.float32)).max() <= 1
    for y, want in ((0.0, 1.0), (1.0, 2.0), (0.5, np.float32(np.sqrt(np.float32(2)))), (3.0, 8.0), (-2.0, 0.25)):
        assert np.float32(L.zo_math_powf(2.0, y)) == np.float32(want)
    es = rng.uniform(-20, 20, 50000).astype(np.float32)
    ge = np.array([L.zo_math_expf(float(x)) for x in es[:5000]], np.float32)
    assert _ulps(ge, np.exp(es[:5000].astype(np.float64)).astype(np.float32)).max() <= 1
    ls = rng.uniform(1e-6, 1e6, 5000).astype(np.float32)
    gl = np.array([L.zo_math_logf(float(x)) for x in ls], np.float32)
    assert _ulps(gl, np.log(ls.astype(np.float64)).astype(np.float32)).max() <= 1


def test_pio2_tables_rederived():
    """The 2/pi and pi/2 chunk tables pasted into zmath_ref.h / zmath.hip.h equal a fresh derivation."""
    ipio2, pio2 = gen_pio2_tables.tables()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in ("oracle/zmath_ref.h", "zang_amd/csrc/zmath.hip.h"):
        text = open(os.path.join(root, path)).read()
        for e in ipio2:
            assert ("0x%06X" % e) in text, (path, hex(e))
        for v in pio2:
            assert float.hex(v).replace("0x1.", "0x1.")[:12] in text, (path, float.hex(v))
