"""GPU parity: the device's std.math.sin / cos (zh_sin, zh_cos) against the oracle's musl restatement, bit for bit.

The device folds musl's magnitude ladder into straight-line code (csrc/zmath.hip.h); this sweeps every leaf of
that ladder, both signs: the ladder's thresholds +- a few ulps, |x| < 2^-12, [-9pi/4, 9pi/4], the two-constant
medium range, the 2^28*pi/2 boundary, huge arguments, denormals, zeros, infinities and NaNs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THRESHOLDS = [0x39800000, 0x3f490fda, 0x4016cbe3, 0x407b53d1, 0x40afeddf, 0x40e231d5, 0x4dc90fdb, 0x7f800000]


def _inputs():
    rng = np.random.default_rng(20260102)
    near = np.array([t + d for t in THRESHOLDS for d in range(-4, 5)], np.uint32)
    bits = [near, near | np.uint32(0x80000000),
            rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64).astype(np.uint32),            # every exponent, both signs
            np.array([0, 0x80000000, 1, 0x80000001, 0x007fffff, 0x00800000, 0x7f7fffff, 0xff7fffff,
                      0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001], np.uint32)]
    xs = [b.view(np.float32) for b in bits]
    xs.append(rng.uniform(-7.1, 7.1, 1_000_000).astype(np.float32))                             # the ladder
    xs.append(rng.uniform(-900.0, 900.0, 1_000_000).astype(np.float32))                         # oscillator phases * 2pi
    xs.append((rng.uniform(-1, 1, 200_000) * 2.0 ** rng.integers(3, 29, 200_000)).astype(np.float32))   # medium
    xs.append((np.arange(1, 200_001, dtype=np.float64) * (np.pi / 2)).astype(np.float32))       # next to multiples of pi/2
    xs.append((np.arange(1, 200_001, dtype=np.float64) * (np.pi / 4)).astype(np.float32))       # next to the kernel boundaries
    return np.ascontiguousarray(np.concatenate(xs))


@pytest.mark.parametrize("name", ["sin", "cos"])
def test_sin_cos_bitexact(ctx, oracle, name):
    import torch
    from zang_amd import abi
    xs = _inputs()
    ref = np.zeros_like(xs)
    getattr(oracle.lib(), "zo_math_%sf_n" % name)(oracle.fptr(xs), oracle.fptr(ref), xs.size)
    x = torch.from_numpy(xs).to(ctx.device)
    out = torch.empty_like(x)
    abi.check(getattr(ctx.lib, "zh_" + name)(ctx.handle, xs.size, out.data_ptr(), x.data_ptr()), "zh_" + name)
    ctx.sync()
    got = out.cpu().numpy()
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    bad = np.nonzero((got.view(np.uint32) != ref.view(np.uint32)) & ~nan)[0]
    assert bad.size == 0, (name, bad.size, [(hex(int(xs[i:i + 1].view(np.uint32)[0])), float(got[i]), float(ref[i])) for i in bad[:8]])


@pytest.mark.parametrize("name", ["sin", "cos", "atan"])
def test_sin_cos_stratified_2_to_28(ctx, oracle, name):
    """(atan, round 4: zatanf takes its range's coefficients from a row instead of selects and divides with v_rcp_f32 + one residual
    correction -- forms tools/ubench/atan_exhaustive.hip holds against musl's branchy order for ALL 2^32 arguments on the device,
    profiles/r04/atan_exhaustive.txt; the same stratified check against the host oracle.)
    Round 3: the device's sinf / cosf fuse multiply-add pairs and take fn from musl's magic-number rounding (csrc/zmath.hip.h
    zsincos_kernels / zreduce_pio2f) -- forms that tools/ubench/sin_exhaustive.hip holds against musl's operation order for ALL
    2^32 arguments on the device.  Here a 2^28-argument stratified subset (every 16th bit pattern, the offset rotating so that
    all 16 residues are visited) against the ORACLE's musl restatement on the host: every exponent, both signs, every leaf."""
    import concurrent.futures as cf
    import torch
    from zang_amd import abi
    L = oracle.lib()
    fn = getattr(L, "zo_math_%sf_n" % name)
    n = 1 << 24
    base = torch.arange(n, dtype=torch.int64, device=ctx.device) * 16

    def chunk(c):
        bits = (base + (c << 28) + ((c * 7 + 3) % 16)).to(torch.int32)           # patterns [c * 2^28, (c + 1) * 2^28), stride 16
        x = bits.view(torch.float32)
        out = torch.empty_like(x)
        abi.check(getattr(ctx.lib, "zh_" + name)(ctx.handle, n, out.data_ptr(), x.data_ptr()), "zh_" + name)
        ctx.sync()
        return x.cpu().numpy(), out.cpu().numpy()

    def check(c, xs, got):
        ref = np.empty_like(xs)
        parts = 4
        for k in range(parts):                                                    # (ctypes releases the GIL: the pool runs these side by side)
            a, b = k * n // parts, (k + 1) * n // parts
            fn(oracle.fptr(xs[a:b]), oracle.fptr(ref[a:b]), b - a)
        nan = np.isnan(ref)
        assert np.array_equal(np.isnan(got), nan), (name, c)
        bad = np.nonzero((got.view(np.uint32) != ref.view(np.uint32)) & ~nan)[0]
        assert bad.size == 0, (name, c, bad.size, [(hex(int(xs[i:i + 1].view(np.uint32)[0])), float(got[i]), float(ref[i])) for i in bad[:8]])
        return n

    done = 0
    with cf.ThreadPoolExecutor(max_workers=8) as pool:
        futs = []
        for c in range(16):
            xs, got = chunk(c)
            futs.append(pool.submit(check, c, xs, got))
        for f in futs:
            done += f.result()
    assert done == 1 << 28


def test_sin_cos_rejects_null(ctx):
    assert ctx.lib.zh_sin(ctx.handle, 4, None, None) != 0
    assert ctx.lib.zh_cos(ctx.handle, 0, None, None) == 0


ATAN_THRESHOLDS = [0x39800000, 0x3ee00000, 0x3f300000, 0x3f980000, 0x401c0000, 0x4c800000, 0x7f800000]


def test_atan_bitexact(ctx, oracle):
    """zh_atan against the oracle's musl atanf over every range of the routine (both signs): the thresholds +- a few
    ulps, random bit patterns of every exponent, the signal range of Distortion's overdrive, zeros, infinities, NaNs."""
    import torch
    from zang_amd import abi
    rng = np.random.default_rng(7)
    near = np.array([t + d for t in ATAN_THRESHOLDS for d in range(-4, 5)], np.uint32)
    xs = np.ascontiguousarray(np.concatenate([
        near.view(np.float32), (near | np.uint32(0x80000000)).view(np.float32),
        rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        rng.uniform(-70.0, 70.0, 1_000_000).astype(np.float32), rng.uniform(-3.0, 3.0, 1_000_000).astype(np.float32),
        np.array([0, 0x80000000, 1, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00001], np.uint32).view(np.float32)]))
    ref = np.zeros_like(xs)
    oracle.lib().zo_math_atanf_n(oracle.fptr(xs), oracle.fptr(ref), xs.size)
    x = torch.from_numpy(xs).to(ctx.device)
    out = torch.empty_like(x)
    abi.check(ctx.lib.zh_atan(ctx.handle, xs.size, out.data_ptr(), x.data_ptr()), "zh_atan")
    ctx.sync()
    got = out.cpu().numpy()
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    bad = np.nonzero((got.view(np.uint32) != ref.view(np.uint32)) & ~nan)[0]
    assert bad.size == 0, (bad.size, [(hex(int(xs[i:i + 1].view(np.uint32)[0])), float(got[i]), float(ref[i])) for i in bad[:8]])
