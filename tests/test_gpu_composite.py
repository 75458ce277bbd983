"""GPU parity: the fused NiceInstrument / PMOscInstrument kernels vs the oracle's UNFUSED
composition of module paints through temps (examples/modules.zig:101-127, 212-247), and the
fused mixdown variant."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024

# (span, note_on, note_id_changed) per paint: attack -> decay -> sustain -> release -> retrigger
SCRIPT = [((0, 1024), 1, 1), ((0, 1024), 1, 0), ((0, 500), 1, 0), ((500, 1024), 0, 0), ((0, 1024), 0, 0),
          ((0, 300), 0, 0), ((300, 1024), 1, 1), ((0, 1024), 1, 0)]


def _oracle_nice(po, V, freq, color, script):
    L = po.lib()
    outs = []
    st = [po.NiceInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_nice_init(C.byref(st[v]), float(color[v]))
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for ((s, e), on, nic) in script:
        out = np.zeros((V, F), np.float32)
        for v in range(V):
            L.zo_nice_paint(C.byref(st[v]), s, e, po.fptr(out[v]), po.fptr(t0), po.fptr(t1), nic, SR, float(freq[v]), on)
        outs.append(out)
    return outs, st


def test_nice_fused_equals_unfused_oracle(ctx, oracle):
    from zang_amd import modules as mod, zang, workloads
    V = 300                                  # the last wave has idle lanes (and even, for the W = 2 variant)
    freq, color, _, _ = workloads.voice_params(5, 0, V)
    freq[:3] = [7000.0, -3.0, 0.5]          # silent / silent / very low
    refs, rst = _oracle_nice(oracle, V, freq, color, SCRIPT)
    m = mod.NiceInstrument(V, util.dev(color), ctx)
    gf = util.dev(freq)
    for k, ((s, e), on, nic) in enumerate(SCRIPT):
        out = ctx.image(F, V, fill=0.0)
        m.paint(zang.Span(s, e), [out], None, bool(nic), m.Params(SR, gf, bool(on)))
        ctx.sync()
        util.assert_bitexact(util.from_image(out), refs[k], f"nice paint {k}")
    st = m.state()
    assert [int(x) for x in st["osc"]["cnt"]] == [r.osc.cnt for r in rst]
    util.assert_bitexact(st["flt"]["l"].astype(np.float32), np.array([r.flt.l for r in rst], np.float32), "nice flt.l")
    util.assert_bitexact(st["flt"]["b"].astype(np.float32), np.array([r.flt.b for r in rst], np.float32), "nice flt.b")
    assert [int(x) for x in st["env"]["state"]] == [r.env.state for r in rst]
    util.assert_bitexact(st["env"]["last_value"].astype(np.float32), np.array([r.env.painter.last_value for r in rst], np.float32), "nice env")


@pytest.mark.parametrize("V", [1, 65])
@pytest.mark.parametrize("zero_first", [False, True])
def test_nice_few_voices_short_spans(ctx, oracle, V, zero_first):
    """One voice and one-past-a-wave, spans shorter than the pipeline's 32-frame tile, `+=` onto a non-zero image."""
    from zang_amd import modules as mod, zang, workloads
    freq, color, _, _ = workloads.voice_params(5, 3, V)
    script = [((0, 7), 1, 1), ((7, 40), 1, 0), ((40, 41), 1, 0), ((41, 1000), 0, 0), ((1000, 1024), 1, 1), ((3, 3), 0, 0), ((0, 1024), 0, 0)]
    refs, rst = _oracle_nice(oracle, V, freq, color, script)
    m = mod.NiceInstrument(V, util.dev(color), ctx)
    gf = util.dev(freq)
    base = util.rng_buffers(23, V, F)
    for k, ((s, e), on, nic) in enumerate(script):
        out = util.to_image(base)
        m.paint(zang.Span(s, e), [out], None, bool(nic), m.Params(SR, gf, bool(on)), zero_first=zero_first)
        ctx.sync()
        ref = base.copy()
        ref[:, s:e] = (0.0 if zero_first else base[:, s:e]) + refs[k][:, s:e]
        util.assert_bitexact(util.from_image(out), ref, f"nice V={V} paint {k}")
    st = m.state()
    assert [int(x) for x in st["osc"]["cnt"]] == [r.osc.cnt for r in rst]
    util.assert_bitexact(st["flt"]["l"].astype(np.float32), np.array([r.flt.l for r in rst], np.float32), "flt.l")
    assert [int(x) for x in st["env"]["state"]] == [r.env.state for r in rst]


def test_nice_equals_gpu_unfused_modules(ctx):
    """The fused kernel against the SAME recipe run as separate GPU module paints through
    device temps (what a reference user would write against this library)."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 256
    freq, color, _, _ = workloads.voice_params(5, 0, V)
    gf, gc = util.dev(freq), util.dev(color)
    fused = mod.NiceInstrument(V, gc, ctx)
    osc, flt, env = mod.PulseOsc(V, ctx), mod.Filter(V, ctx), mod.Envelope(V, ctx)
    cutoff = mod.Filter.cutoffFromFrequency(gf * 8.0, SR, ctx)
    t0, t1 = ctx.image(F, V), ctx.image(F, V)
    for ((s, e), on, nic) in SCRIPT:
        sp = zang.Span(s, e)
        a = ctx.image(F, V, fill=0.0); b = ctx.image(F, V, fill=0.0)
        fused.paint(sp, [a], [t0, t1], bool(nic), fused.Params(SR, gf, bool(on)))
        zang.zero(sp, t0, ctx=ctx)
        osc.paint(sp, [t0], [], bool(nic), osc.Params(SR, zang.constant(gf), gc))
        zang.multiplyWithScalar(sp, t0, 0.5, ctx=ctx)
        zang.zero(sp, t1, ctx=ctx)
        flt.paint(sp, [t1], [], bool(nic), flt.Params(t0, flt.low_pass, zang.constant(cutoff), zang.constant(0.7)))
        zang.zero(sp, t0, ctx=ctx)
        env.paint(sp, [t0], [], bool(nic), env.Params(SR, zang.PaintCurve.cubed(0.01), zang.PaintCurve.cubed(0.1),
                                                     zang.PaintCurve.cubed(0.5), 0.8, bool(on)))
        zang.multiply(sp, b, t0, t1, ctx=ctx)
        ctx.sync()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("wg_min", [None, "0"])      # None: the library's choice by voice count (per-wave rows here); "0": one row per workgroup
def test_nice_paint_mix(ctx, oracle, wg_min, monkeypatch):
    if wg_min is not None:
        util.set_form(monkeypatch, nice_mix_wg_min=wg_min)
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 1000          # not a multiple of 256: tail lanes
    freq, color, _, _ = workloads.voice_params(5, 7, V)
    m1 = mod.NiceInstrument(V, util.dev(color), ctx)
    m2 = mod.NiceInstrument(V, util.dev(color), ctx)
    gf = util.dev(freq)
    for ((s, e), on, nic) in SCRIPT:
        per_voice = ctx.image(F, V, fill=0.0)
        m1.paint(zang.Span(s, e), [per_voice], None, bool(nic), m1.Params(SR, gf, bool(on)))
        mix = torch.full((F,), 0.25, dtype=torch.float32, device="cuda")
        m2.paint_mix(zang.Span(s, e), mix, bool(nic), m2.Params(SR, gf, bool(on)))
        mix_b = torch.full((F,), 0.25, dtype=torch.float32, device="cuda")
        ctx.sync()
        pv = per_voice.cpu().numpy().astype(np.float64)
        ref = 0.25 + pv.sum(axis=1)
        ref[:s] = 0.25; ref[e:] = 0.25
        got = mix.cpu().numpy()
        bound = 4 * np.sqrt(V) * np.finfo(np.float32).eps * max(np.abs(pv).sum(axis=1).max(), 1.0)
        assert np.abs(got - ref).max() <= bound
        assert np.array_equal(got[:s], np.full(s, 0.25, np.float32))
    assert np.array_equal(m1.state(), m2.state())


@pytest.mark.parametrize("wg_min", [None, "0"])      # None: the library's choice by voice count (per-wave rows here); "0": one row per workgroup
def test_nice_paint_mix_stereo(ctx, oracle, wg_min, monkeypatch):
    """Two channels (examples/example_stereo.zig:84-98 with a constant pan per voice): every voice is added to the left
    channel times its left gain and to the right channel times (1 - left); the per-voice products are f32, their sum
    over the voices is checked against the f64 sum with the sqrt(V) * eps bound; with both gains 1.0 the two channels
    equal the mono mixdown bit for bit (x * 1.0 == x, same summation order)."""
    if wg_min is not None:
        util.set_form(monkeypatch, nice_mix_wg_min=wg_min)
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 1000
    freq, color, u2, _ = workloads.voice_params(5, 7, V)
    pan = (2.0 * u2 - 1.0).astype(np.float32)
    # scaleWave(min 0, max 1) and invertWaveInPlace of the reference, on the per-voice constant (example_stereo.zig:20-39)
    gl = (np.float32(0.0) + ((np.float32(0.0) + pan * np.float32(0.5)) + np.float32(0.5))).astype(np.float32)
    gr = (np.float32(0.0) + ((np.float32(0.0) + gl * np.float32(-1.0)) + np.float32(1.0))).astype(np.float32)
    m1 = mod.NiceInstrument(V, util.dev(color), ctx)
    m2 = mod.NiceInstrument(V, util.dev(color), ctx)
    m3 = mod.NiceInstrument(V, util.dev(color), ctx)
    m4 = mod.NiceInstrument(V, util.dev(color), ctx)
    gf, dgl, dgr = util.dev(freq), util.dev(gl), util.dev(gr)
    for ((s, e), on, nic) in SCRIPT:
        per_voice = ctx.image(F, V, fill=0.0)
        P = m1.Params(SR, gf, bool(on))
        m1.paint(zang.Span(s, e), [per_voice], None, bool(nic), P)
        left = torch.full((F,), 0.25, dtype=torch.float32, device="cuda"); right = torch.full((F,), -0.5, dtype=torch.float32, device="cuda")
        m2.paint_mix_stereo(zang.Span(s, e), left, right, dgl, dgr, bool(nic), P)
        mono = torch.zeros(F, device="cuda"); l1 = torch.zeros(F, device="cuda"); r1 = torch.zeros(F, device="cuda")
        m3.paint_mix(zang.Span(s, e), mono, bool(nic), P, zero_first=True)
        m4.paint_mix_stereo(zang.Span(s, e), l1, r1, 1.0, 1.0, bool(nic), P, zero_first=True)
        ctx.sync()
        pv = per_voice.cpu().numpy()                                       # [frames][voices]
        for got_t, g, base in ((left, gl, 0.25), (right, gr, -0.5)):
            prod = (pv * g[None, :]).astype(np.float32).astype(np.float64)  # f32 products, summed in f64
            ref = base + prod.sum(axis=1)
            ref[:s] = base; ref[e:] = base
            got = got_t.cpu().numpy()
            bound = 4 * np.sqrt(V) * np.finfo(np.float32).eps * max(np.abs(prod).sum(axis=1).max(), 1.0)
            assert np.abs(got - ref).max() <= bound
            assert np.array_equal(got[:s], np.full(s, base, np.float32)) and np.array_equal(got[e:], np.full(F - e, base, np.float32))
        assert torch.equal(l1.view(torch.int32), mono.view(torch.int32)) and torch.equal(r1.view(torch.int32), mono.view(torch.int32))
    assert np.array_equal(m1.state(), m2.state()) and np.array_equal(m1.state(), m4.state())


@pytest.mark.parametrize("form", ["ranges", "sequential"])
def test_pmosc_fused_equals_unfused_oracle(ctx, oracle, form, monkeypatch):
    """(At a small voice count a span is painted as frame ranges at once, each range replaying the phase / envelope walk of
    the earlier frames -- k_pmosc_ranges; ZH_PMOSC_RANGES=0 is the lane-per-voice walk k_pmosc.)"""
    if form == "sequential":
        util.set_form(monkeypatch, pmosc_ranges="0")
    from zang_amd import modules as mod, zang, workloads
    V = 192
    freq, _, u2, _ = workloads.voice_params(4, 0, V)
    freq = (freq * 0.5).astype(np.float32)
    rel = (0.05 + 0.4 * u2).astype(np.float32)
    L = oracle.lib()
    st = [oracle.PMOscInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_pmosc_init(C.byref(st[v]), float(rel[v]))
    m = mod.PMOscInstrument(V, util.dev(rel), ctx)
    gf = util.dev(freq)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32); t2 = np.zeros(F, np.float32)
    for k, ((s, e), on, nic) in enumerate(SCRIPT):
        ref = np.zeros((V, F), np.float32)
        for v in range(V):
            L.zo_pmosc_paint(C.byref(st[v]), s, e, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(t2), nic, SR, float(freq[v]), on)
        out = ctx.image(F, V, fill=0.0)
        m.paint(zang.Span(s, e), [out], None, bool(nic), m.Params(SR, gf, bool(on)))
        ctx.sync()
        got = util.from_image(out)
        util.assert_close(got, ref, f"pmosc paint {k}")
        util.assert_bitexact(got, ref, f"pmosc paint {k} (expected exact)")
    gs = m.state()
    util.assert_bitexact(gs["carrier"]["t"].astype(np.float32), np.array([r.carrier.t for r in st], np.float32), "carrier t")
    util.assert_bitexact(gs["modulator"]["t"].astype(np.float32), np.array([r.modulator.t for r in st], np.float32), "modulator t")


@pytest.mark.parametrize("color", [0, 1])
@pytest.mark.parametrize("ftype", [0, 1, 3, 5])
def test_noise_filter_fused_equals_unfused(ctx, oracle, color, ftype):
    """The fused Noise->Filter voice vs the oracle running zero/Noise.paint/zero/Filter.paint through a temp
    (examples/example_stereo.zig:71-82), three sub-spans, carried state; global seeds.  At this voice count the
    library picks the two-wave producer/consumer kernel; the test below reruns this one with the single-wave form."""
    from zang_amd import modules as mod, zang
    V, first = 150, 5000                                       # 150: the last wave has idle lanes
    rng = np.random.default_rng(17)
    cutoff = rng.uniform(-0.05, 1.05, V).astype(np.float32); res = rng.uniform(-0.05, 1.05, V).astype(np.float32)
    out0 = util.rng_buffers(18, V, F)
    L = oracle.lib()
    ref = out0.copy(); rl = np.zeros(V, np.float32); rb = np.zeros(V, np.float32); rs = []
    temp = np.zeros(F, np.float32)
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for (s, e) in util.SPANS_THREE:
            L.zo_zero(s, e, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nz), s, e, oracle.fptr(temp), color)
            L.zo_filter_paint(C.byref(fl), s, e, oracle.fptr(ref[v]), oracle.fptr(temp), ftype, oracle.constant(cutoff[v]), oracle.constant(res[v]))
        rl[v], rb[v] = fl.l, fl.b; rs.append(list(nz.r))
    m = mod.NoiseFilter(V, ctx, first_seed=first)
    out = util.to_image(out0)
    gc, gr = util.dev(cutoff), util.dev(res)
    for (s, e) in util.SPANS_THREE:
        m.paint(zang.Span(s, e), [out], None, False, m.Params(color, ftype, gc, gr))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"noise_filter color {color} type {ftype}")
    st = m.state()
    util.assert_bitexact(st["flt"]["l"].astype(np.float32), rl, "l"); util.assert_bitexact(st["flt"]["b"].astype(np.float32), rb, "b")
    assert [[int(x) for x in row] for row in st["noise"]["r"]] == rs


@pytest.mark.parametrize("zero_first", [True, False])
@pytest.mark.parametrize("V,ftype", [(300, 1), (4096, 4)])
def test_noise_filter_three_wave_form_with_multi_draw_voices(ctx, oracle, zero_first, V, ftype):
    """White noise -> Filter at small voice counts runs as two noise producer waves (alternate 32-frame tiles, T^32 jump
    between them) and one filter wave (k_noise_filter_pc2).  Every voice equals the oracle's zero / Noise.paint / Filter.paint
    sequence bit for bit over full buffers and shorter spans with carried state -- including voices crafted so that one of
    Random.float's multi-draw samples (2^-41 per sample) lands on a chosen frame of the first span (first tile, tile
    edges, a middle tile, the last frame): the consumer stops such a voice after the tile and k_noise_filter_fix
    continues it sequentially from the recorded generator / filter state."""
    from zang_amd import modules as mod, zang
    from tests.test_gpu_modules import _xoshiro_step_back
    first = 7000
    rng = np.random.default_rng(91)
    L = oracle.lib()
    cutoff = rng.uniform(0.02, 0.6, V).astype(np.float32); res = rng.uniform(0.0, 0.9, V).astype(np.float32)
    nzs, fls = [], []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v); nzs.append(nz)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
    crafted = {5 + 41 * i: k for i, k in enumerate([0, 31, 32, 63, 64, 500, 1023])}
    for v, k in crafted.items():
        back = _xoshiro_step_back([0, int(rng.integers(1, 1 << 63)), int(rng.integers(1, 1 << 63)), 1 << 41], k)
        for i in range(4):
            nzs[v].r[i] = back[i]
    m = mod.NoiseFilter(V, ctx, first_seed=first)
    st = m.state()
    for v in crafted:
        st["noise"]["r"][v] = [int(x) for x in nzs[v].r]
    m.set_state(st)
    gc, gr = util.dev(cutoff), util.dev(res)
    out0 = util.rng_buffers(13, V, F)
    temp = np.zeros(F, np.float32)
    for (s, e) in [(0, 1024), (0, 1024), (100, 612), (612, 1001), (0, 70)]:
        ref = out0.copy()
        if zero_first:
            ref[:, s:e] = 0.0
        for v in range(V):
            L.zo_zero(s, e, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nzs[v]), s, e, oracle.fptr(temp), 0)
            L.zo_filter_paint(C.byref(fls[v]), s, e, oracle.fptr(ref[v]), oracle.fptr(temp), ftype, oracle.constant(cutoff[v]), oracle.constant(res[v]))
        out = util.to_image(out0)
        m.paint(zang.Span(s, e), [out], None, False, m.Params(0, ftype, gc, gr), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"noise_filter three-wave form, span {(s, e)}")
        gs = m.state()
        assert [[int(x) for x in row] for row in gs["noise"]["r"]] == [list(n.r) for n in nzs], f"generator states after span {(s, e)}"
        util.assert_bitexact(gs["flt"]["l"].astype(np.float32), np.array([f.l for f in fls], np.float32), "l")
        util.assert_bitexact(gs["flt"]["b"].astype(np.float32), np.array([f.b for f in fls], np.float32), "b")


@pytest.mark.parametrize("V", [1, 65])
@pytest.mark.parametrize("zero_first", [False, True])
def test_noise_filter_few_voices_short_spans(ctx, oracle, V, zero_first):
    """One voice and one-past-a-wave through the two-wave pipeline: spans shorter than its 32-frame tile, a span that is
    not a multiple of it, an empty span; pink noise, low-pass; `+=` and ZERO_FIRST."""
    from zang_amd import modules as mod, zang
    first, color, ftype = 77, 1, 0
    rng = np.random.default_rng(29)
    cutoff = rng.uniform(0.0, 1.0, V).astype(np.float32); res = rng.uniform(0.0, 1.0, V).astype(np.float32)
    spans = [(0, 5), (5, 37), (37, 37), (37, 38), (38, 1001), (1001, 1024)]
    base = util.rng_buffers(31, V, F)
    L = oracle.lib()
    temp = np.zeros(F, np.float32)
    nzs, fls = [], []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v); nzs.append(nz)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
    m = mod.NoiseFilter(V, ctx, first_seed=first)
    gc, gr = util.dev(cutoff), util.dev(res)
    for (s, e) in spans:
        ref = base.copy()
        if zero_first:
            ref[:, s:e] = 0.0
        for v in range(V):
            L.zo_zero(s, e, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nzs[v]), s, e, oracle.fptr(temp), color)
            L.zo_filter_paint(C.byref(fls[v]), s, e, oracle.fptr(ref[v]), oracle.fptr(temp), ftype, oracle.constant(cutoff[v]), oracle.constant(res[v]))
        out = util.to_image(base)
        m.paint(zang.Span(s, e), [out], None, False, m.Params(color, ftype, gc, gr), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"noise_filter V={V} span {(s, e)}")
    st = m.state()
    util.assert_bitexact(st["flt"]["l"].astype(np.float32), np.array([f.l for f in fls], np.float32), "l")
    assert [[int(x) for x in row] for row in st["noise"]["r"]] == [list(n.r) for n in nzs]


@pytest.mark.gpu
def test_noise_filter_single_wave_form_is_bit_identical():
    """k_noise_filter (one wave does noise and filter: the form used above ZH_NF_PC_MAX voices) against the same
    oracle: rerun the fused Noise->Filter parity tests in a subprocess with the two-wave form switched off."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = util.forms_env(nf_pc_max=0)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_composite.py", "-q", "-m", "gpu", "-k",
                        "noise_filter_fused_equals_unfused or noise_filter_few_voices_short_spans"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    util.assert_rerun_green(r, 12)


@pytest.mark.gpu
def test_nice_single_wave_form_is_bit_identical():
    """k_nice (one wave runs oscillator, envelope and filter: the form used above ZH_NICE_PC_MAX voices) against the
    same oracle: rerun the NiceInstrument parity tests in a subprocess with the three-wave pipeline switched off."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = util.forms_env(nice_pc_max=0)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_composite.py", "-q", "-m", "gpu", "-k",
                        "nice_fused_equals_unfused_oracle or nice_equals_gpu_unfused_modules or nice_few_voices_short_spans"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    util.assert_rerun_green(r, 6)


@pytest.mark.parametrize("form", ["default", "three_waves", "one_wave"])
def test_nice_with_non_finite_and_huge_filter_states(ctx, oracle, form, monkeypatch):
    """The low-pass mix into the zeroed temp is computed as `0 + (l + b*0)` (dsp.hip.h svf_lowpass_into_zero): voices whose
    filter state is infinite, NaN, huge (overflowing inside the step), denormal or a signed zero must still give what the
    reference's `l*1 + b*0 + h*0` gives -- NaN where it gives NaN (payloads aside), the same bits everywhere else --
    in every kernel form, and in the mixdown form."""
    from zang_amd import modules as mod, zang, workloads
    if form == "three_waves":
        util.set_form(monkeypatch, nice_pc4_max="0")
    elif form == "one_wave":
        util.set_form(monkeypatch, nice_pc_max="0")
    V = 200
    freq, color, _, _ = workloads.voice_params(5, 9, V)
    L = oracle.lib()
    st = [oracle.NiceInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_nice_init(C.byref(st[v]), float(color[v]))
    vals = [np.inf, -np.inf, np.nan, 3.0e38, -3.0e38, 1e-45, -0.0, 0.0, 1e30, -1e25]
    for i, v in enumerate(range(3, V, 7)):
        st[v].flt.l = vals[i % len(vals)]
        st[v].flt.b = vals[(i // 3 + 1) % len(vals)]
    m = mod.NiceInstrument(V, util.dev(color), ctx)
    gs = m.state()
    for v in range(V):
        gs["flt"]["l"][v] = st[v].flt.l
        gs["flt"]["b"][v] = st[v].flt.b
    m.set_state(gs)
    gf = util.dev(freq)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)

    def same(got, ref, what):
        ng, nr = np.isnan(got), np.isnan(ref)
        assert np.array_equal(ng, nr), what + ": NaN positions"
        assert np.array_equal(got[~ng].view(np.uint32), ref[~nr].view(np.uint32)), what
        return int(nr.sum())

    nans = 0
    for k, ((s, e), on, nic) in enumerate([((0, 1024), 1, 1), ((0, 700), 1, 0), ((700, 1024), 0, 0)]):
        ref = np.zeros((V, F), np.float32)
        for v in range(V):
            L.zo_nice_paint(C.byref(st[v]), s, e, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1), nic, SR, float(freq[v]), on)
        out = ctx.image(F, V, fill=0.0)
        m.paint(zang.Span(s, e), [out], None, bool(nic), m.Params(SR, gf, bool(on)))
        ctx.sync()
        nans += same(util.from_image(out), ref, f"nice {form} paint {k}")
        g2 = m.state()
        same(g2["flt"]["l"].astype(np.float32), np.array([r.flt.l for r in st], np.float32), "flt.l")
        same(g2["flt"]["b"].astype(np.float32), np.array([r.flt.b for r in st], np.float32), "flt.b")
    assert nans > 1000


@pytest.mark.parametrize("V", [300, 4096])
@pytest.mark.parametrize("wg_min", [None, "0"])
def test_nice_paint_mix_stereo_batch_equals_separate_calls(ctx, V, wg_min, monkeypatch):
    """zh_nice_paint_mix_stereo_batch: n consecutive paints in one launch (state in registers from buffer to buffer, one
    second pass) -- bit for bit the mixes and the final state of the n separate calls, with notes going on and off, a new note
    and a frequency change between buffers, ZERO_FIRST and `+=`, and a sub-span."""
    if wg_min is not None:
        util.set_form(monkeypatch, nice_mix_wg_min=wg_min)
    import torch
    from zang_amd import modules as mod, zang, workloads
    freq, color, u2, _ = workloads.voice_params(5, 3, V)
    gl = util.dev((0.25 + 0.5 * u2).astype(np.float32)); gr = util.dev((0.75 - 0.5 * u2).astype(np.float32))
    f1, f2 = util.dev(freq), util.dev((freq * np.float32(1.25)).astype(np.float32))
    on_mix = util.dev((np.arange(V) % 3 != 0).astype(np.uint8))
    ma, mb = mod.NiceInstrument(V, util.dev(color), ctx), mod.NiceInstrument(V, util.dev(color), ctx)
    script = [(True, True, f1), (True, False, f1), (on_mix, False, f1), (False, False, f2), (True, True, f2), (False, False, f2), (on_mix, on_mix, f1)]
    for (span, zf) in ((zang.Span(0, F), True), (zang.Span(100, 900), False)):
        n = len(script)
        la = torch.full((n, F), 0.5, device="cuda"); ra = torch.full((n, F), -0.25, device="cuda")
        lb, rb = la.clone(), ra.clone()
        P = [ma.Params(SR, f, on) for (on, _, f) in script]
        for k, (on, nic, f) in enumerate(script):
            ma.paint_mix_stereo(span, la[k], ra[k], gl, gr, nic, P[k], zero_first=zf)
        mb.paint_mix_stereo_batch(span, [lb[k] for k in range(n)], [rb[k] for k in range(n)], gl, gr, [nic for (_, nic, _) in script], P, zero_first=zf)
        ctx.sync()
        assert float(la.abs().max()) > 0.5
        assert torch.equal(la.view(torch.int32), lb.view(torch.int32)) and torch.equal(ra.view(torch.int32), rb.view(torch.int32))
        assert ma.state().tobytes() == mb.state().tobytes()
    # argument checks
    from zang_amd import abi
    with pytest.raises(abi.ZangHipError):
        mb.paint_mix_stereo_batch(zang.Span(0, F), [lb[0], lb[1]], [rb[0], rb[1]], gl, gr, [False, False], [ma.Params(SR, f1, True), ma.Params(44100.0, f1, True)])
