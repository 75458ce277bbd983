"""GPU parity, randomised: random voice counts, random sequences of paint calls (random spans, empty spans,
ZERO_FIRST / +=, note on / off / retrigger, per-voice frequencies that change between calls, silent voices)
through the kernels that exist in several forms -- the chunked PulseOsc, the fused NiceInstrument and the fused
Noise->Filter voice -- against the oracle, bit for bit, outputs and carried state.  Seeded: failures reproduce."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024


def _calls(rng, n):
    out = []
    for _ in range(n):
        a, b = sorted(int(x) for x in rng.integers(0, F + 1, 2))
        if rng.random() < 0.15:
            b = a                                               # empty span: prologue / epilogue only
        elif rng.random() < 0.3:
            a, b = 0, F
        out.append((a, b, bool(rng.random() < 0.5)))           # (start, end, zero_first)
    return out


def _freqs(rng, V):
    f = (55.0 * 2.0 ** (6.7 * rng.random(V))).astype(np.float32)
    f[rng.random(V) < 0.05] = np.float32(7000.0)                # above sr/8: silent
    f[rng.random(V) < 0.05] = np.float32(-1.0)                  # negative: silent
    return f


@pytest.mark.parametrize("seed", range(5))
def test_fuzz_pulseosc(ctx, oracle, seed):
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.choice([1, 4, 63, 64, 65, 96, 200, 256, 260]))
    color = rng.uniform(-0.1, 1.1, V).astype(np.float32)
    L = oracle.lib()
    st = [oracle.PulseOsc() for _ in range(V)]
    for s in st:
        L.zo_pulseosc_init(C.byref(s))
    m = mod.PulseOsc(V, ctx)
    gcol = util.dev(color)
    img = util.rng_buffers(seed, V, F)
    for k, (a, b, zf) in enumerate(_calls(rng, 6)):
        freq = _freqs(rng, V)
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_pulseosc_paint(C.byref(st[v]), a, b, oracle.fptr(ref[v]), SR, oracle.constant(freq[v]), float(color[v]))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], [], False, m.Params(SR, zang.constant(util.dev(freq)), gcol), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"pulseosc seed {seed} call {k} V={V} span {(a, b)} zf={zf}")
        img = ref
    assert [int(x) for x in m.state()["cnt"]] == [s.cnt for s in st]


@pytest.mark.parametrize("seed", range(5))
def test_fuzz_nice(ctx, oracle, seed):
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(2000 + seed)
    V = int(rng.choice([1, 2, 63, 64, 66, 130, 200]))
    color = rng.uniform(0.0, 1.0, V).astype(np.float32)
    L = oracle.lib()
    st = [oracle.NiceInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_nice_init(C.byref(st[v]), float(color[v]))
    m = mod.NiceInstrument(V, util.dev(color), ctx)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    img = util.rng_buffers(seed + 50, V, F)
    on = False
    for k, (a, b, zf) in enumerate(_calls(rng, 7)):
        freq = _freqs(rng, V)
        nic = bool(rng.random() < 0.4)
        on = (not on) if rng.random() < 0.5 else on
        if nic:
            on = True                                           # a new note id arrives with note_on (Envelope.zig:45)
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_nice_paint(C.byref(st[v]), a, b, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1), int(nic), SR, float(freq[v]), int(on))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], None, nic, m.Params(SR, util.dev(freq), on), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"nice seed {seed} call {k} V={V} span {(a, b)} zf={zf} on={on} nic={nic}")
        img = ref
    gs = m.state()
    assert [int(x) for x in gs["osc"]["cnt"]] == [r.osc.cnt for r in st]
    util.assert_bitexact(gs["flt"]["l"].astype(np.float32), np.array([r.flt.l for r in st], np.float32), "flt.l")
    util.assert_bitexact(gs["env"]["t"].astype(np.float32), np.array([r.env.painter.t for r in st], np.float32), "env.t")
    assert [int(x) for x in gs["env"]["state"]] == [r.env.state for r in st]


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_noise_filter(ctx, oracle, seed):
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(3000 + seed)
    V = int(rng.choice([1, 3, 64, 70, 150]))
    first = int(rng.integers(0, 100000))
    L = oracle.lib()
    nzs, fls = [], []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v); nzs.append(nz)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
    m = mod.NoiseFilter(V, ctx, first_seed=first)
    temp = np.zeros(F, np.float32)
    img = util.rng_buffers(seed + 90, V, F)
    for k, (a, b, zf) in enumerate(_calls(rng, 6)):
        color = int(rng.integers(0, 2)); ftype = int(rng.integers(0, 6))
        cutoff = rng.uniform(-0.1, 1.1, V).astype(np.float32); res = rng.uniform(-0.1, 1.1, V).astype(np.float32)
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_zero(a, b, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nzs[v]), a, b, oracle.fptr(temp), color)
            L.zo_filter_paint(C.byref(fls[v]), a, b, oracle.fptr(ref[v]), oracle.fptr(temp), ftype, oracle.constant(cutoff[v]), oracle.constant(res[v]))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], None, False, m.Params(color, ftype, util.dev(cutoff), util.dev(res)), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"noise_filter seed {seed} call {k} V={V} span {(a, b)} zf={zf} color={color} type={ftype}")
        img = ref
    gs = m.state()
    util.assert_bitexact(gs["flt"]["b"].astype(np.float32), np.array([f.b for f in fls], np.float32), "flt.b")
    assert [[int(x) for x in row] for row in gs["noise"]["r"]] == [list(n.r) for n in nzs]


def test_fuzz_again_with_the_single_wave_forms():
    """The same random cases through k_nice / k_noise_filter (the forms used above 65,536 voices)."""
    import os
    import subprocess
    import sys
    if os.environ.get("ZH_FUZZ_CHILD"):
        pytest.skip("already the rerun")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZH_NICE_PC_MAX="0", ZH_NF_PC_MAX="0", ZH_FUZZ_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fuzz.py", "-q", "-m", "gpu"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "14 passed" in r.stdout
