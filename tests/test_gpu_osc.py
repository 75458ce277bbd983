"""GPU parity: PulseOsc / TriSawOsc kernels vs the oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0


def _oracle_osc(po, kind, V, F, spans, freq, color, freq_buf=None, out0=None, state0=None):
    L = po.lib()
    out = np.zeros((V, F), np.float32) if out0 is None else out0.copy()
    states = []
    for v in range(V):
        st = po.PulseOsc() if kind == "pulse" else po.TriSawOsc()
        (L.zo_pulseosc_init if kind == "pulse" else L.zo_trisawosc_init)(C.byref(st))
        if state0 is not None:
            st.cnt = int(state0[v][0])
            if kind != "pulse":
                st.t = float(state0[v][1])
        for (s, e) in spans:
            f = po.constant(freq[v]) if freq_buf is None else po.buffer(freq_buf[v])
            fn = L.zo_pulseosc_paint if kind == "pulse" else L.zo_trisawosc_paint
            fn(C.byref(st), s, e, po.fptr(out[v]), SR, f, float(color[v]))
        states.append((st.cnt,) if kind == "pulse" else (st.cnt, st.t))
    return out, states


def _gpu_osc(ctx, kind, V, F, spans, freq, color, freq_buf=None, out0=None, zero_first=False):
    import torch
    from zang_amd import modules as mod, zang
    m = (mod.PulseOsc if kind == "pulse" else mod.TriSawOsc)(V, ctx)
    out = util.to_image(np.zeros((V, F), np.float32) if out0 is None else out0)
    fb = util.to_image(freq_buf) if freq_buf is not None else None
    fr = util.dev(freq) if freq_buf is None else None
    col = util.dev(color)
    P = m.Params
    for (s, e) in spans:
        f = zang.constant(fr) if fb is None else zang.buffer(fb)
        m.paint(zang.Span(s, e), [out], [], False, P(SR, f, col), zero_first=zero_first)
    ctx.sync()
    return util.from_image(out), m.state()


@pytest.mark.parametrize("kind", ["pulse", "trisaw"])
@pytest.mark.parametrize("spans", [util.SPANS_ONE, util.SPANS_THREE])
def test_const_freq_bitexact(ctx, oracle, kind, spans):
    from zang_amd import workloads
    V, F = 640, 1024
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    # edge voices: silent (freq > sr/8), negative, color 0 / 1 / out of range, tiny freq
    freq[:8] = [6000.0, 6000.5, -1.0, 0.0, 440.0, 440.0, 440.0, 0.001]
    color[:8] = [0.5, 0.5, 0.5, 0.5, 0.0, 1.0, 1.7, 0.3]
    out0 = util.rng_buffers(5, V, F)
    ref, rst = _oracle_osc(oracle, kind, V, F, spans, freq, color, out0=out0)
    got, gst = _gpu_osc(ctx, kind, V, F, spans, freq, color, out0=out0)
    util.assert_bitexact(got, ref, f"{kind} const freq")
    assert [int(x) for x in gst["cnt"]] == [s[0] for s in rst]


@pytest.mark.parametrize("V", [1024, 67])
@pytest.mark.parametrize("zero_first", [False, True])
def test_trisaw_sawtooth_waves(ctx, oracle, V, zero_first):
    """color <= 0 (brpt == 0): a wave whose voices are ALL sawtooths takes trisaw_sample_saw (csrc/voices.hip.h: the two flat arms
    alone); here every voice is one, or whole waves are and others are mixed (one lane group of the four-voices-per-lane kernel holds 256
    voices, the one-voice-per-lane kernel 64), with silent voices, wraps every other sample (freq = sr / 8) and a tiny frequency."""
    from zang_amd import workloads
    F = 1024
    freq, color, _, _ = workloads.voice_params(2, 7, V)
    for colors in (np.zeros(V, np.float32), None):
        c = color.copy() if colors is None else colors
        if colors is None:
            c[:512] = 0.0                                               # two all-sawtooth lane groups, then the generator's colors
            c[5] = -0.3; c[6] = -0.0
        f = freq.copy()
        f[:6] = [6000.0, 6000.5, -1.0, 0.0, 0.001, 5999.9]
        out0 = None if zero_first else util.rng_buffers(8, V, F)        # (the three spans cover the buffer: ZERO_FIRST over garbage = `+=` over zeros)
        ref, rst = _oracle_osc(oracle, "trisaw", V, F, util.SPANS_THREE, f, c, out0=out0)
        got, gst = _gpu_osc(ctx, "trisaw", V, F, util.SPANS_THREE, f, c, out0=util.rng_buffers(9, V, F) if zero_first else out0, zero_first=zero_first)
        util.assert_bitexact(got, ref, f"trisaw sawtooth V={V} zf={zero_first}")
        assert [int(x) for x in gst["cnt"]] == [s[0] for s in rst]


@pytest.mark.parametrize("kind", ["pulse", "trisaw"])
def test_const_freq_zero_first(ctx, oracle, kind):
    from zang_amd import workloads
    V, F = 256, 1024
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    freq[3] = 7000.0   # silent voice: ZERO_FIRST must still leave zeros
    ref, _ = _oracle_osc(oracle, kind, V, F, util.SPANS_ONE, freq, color)
    garbage = util.rng_buffers(6, V, F)
    got, _ = _gpu_osc(ctx, kind, V, F, util.SPANS_ONE, freq, color, out0=garbage, zero_first=True)
    util.assert_bitexact(got, ref, f"{kind} zero_first")


@pytest.mark.parametrize("form", ["ranges", "sequential"])
@pytest.mark.parametrize("kind,colors", [("pulse", [0.0, 0.3, 0.5, 1.0]), ("trisaw", [0.1, 0.5, 0.9, 0.25])])
@pytest.mark.parametrize("spans", [util.SPANS_ONE, util.SPANS_THREE])
def test_controlled_freq(ctx, oracle, kind, colors, spans, form, monkeypatch):
    """(Both oscillators at a small voice count paint a controlled-frequency span as frame ranges at once, each range first
    summing the earlier frames' phase increments; ZH_PULSE_CTRL_RANGES=0 / ZH_TRISAW_CTRL_RANGES=0 is the lane-per-voice walk.)"""
    if form == "sequential":
        util.set_form(monkeypatch, pulse_ctrl_ranges="0")
        util.set_form(monkeypatch, trisaw_ctrl_ranges="0")
    V, F = 192, 1024
    rng = np.random.default_rng(11)
    freq_buf = rng.uniform(-200.0, 7000.0, (V, F)).astype(np.float32)   # includes out-of-range samples
    color = np.resize(np.array(colors, np.float32), V)
    out0 = util.rng_buffers(7, V, F)
    ref, rst = _oracle_osc(oracle, kind, V, F, spans, None, color, freq_buf=freq_buf, out0=out0)
    got, gst = _gpu_osc(ctx, kind, V, F, spans, None, color, freq_buf=freq_buf, out0=out0)
    util.assert_bitexact(got, ref, f"{kind} controlled freq")
    if kind == "pulse":
        assert [int(x) for x in gst["cnt"]] == [s[0] for s in rst]
    else:
        util.assert_bitexact(gst["t"].astype(np.float32), np.array([s[1] for s in rst], np.float32), "t")


def test_pulseosc_config2_full(ctx, oracle):
    """BASELINE config 2: 4096 PulseOsc voices x 1024 frames, two consecutive buffers."""
    from zang_amd import workloads
    V, F = 4096, 1024
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    import torch
    from zang_amd import modules as mod, zang
    m = mod.PulseOsc(V, ctx)
    fr, col = util.dev(freq), util.dev(color)
    outs = [ctx.image(F, V) for _ in range(2)]
    for o in outs:
        m.paint(zang.Span(0, F), [o], [], False, m.Params(SR, zang.constant(fr), col), zero_first=True)
    ctx.sync()
    L = oracle.lib()
    ref = np.zeros((2, V, F), np.float32)
    for v in range(V):
        st = oracle.PulseOsc(); L.zo_pulseosc_init(C.byref(st))
        for b in range(2):
            L.zo_pulseosc_paint(C.byref(st), 0, F, oracle.fptr(ref[b, v]), SR, oracle.constant(freq[v]), float(color[v]))
    for b in range(2):
        util.assert_bitexact(util.from_image(outs[b]), ref[b], f"config2 buffer {b}")


@pytest.mark.parametrize("kind", ["pulse", "trisaw"])
def test_const_freq_scalar_lane_fallback(ctx, oracle, kind):
    """67 voices: not a multiple of 4, so the one-voice-per-lane kernel runs instead of the float4 one."""
    from zang_amd import workloads
    V, F = 67, 1024
    freq, color, _, _ = workloads.voice_params(2, 100, V)
    out0 = util.rng_buffers(15, V, F)
    ref, rst = _oracle_osc(oracle, kind, V, F, util.SPANS_THREE, freq, color, out0=out0)
    got, gst = _gpu_osc(ctx, kind, V, F, util.SPANS_THREE, freq, color, out0=out0)
    util.assert_bitexact(got, ref, f"{kind} scalar-lane")
    assert [int(x) for x in gst["cnt"]] == [s[0] for s in rst]


def _oracle_buffers(po, kind, V, F, span, nbuf, freq, color, out0=None):
    """nbuf consecutive paints over `span`, each into its own buffer (a host loop over 1024-frame buffers)."""
    L = po.lib()
    outs = np.zeros((nbuf, V, F), np.float32) if out0 is None else np.repeat(out0[None], nbuf, 0).copy()
    cnts = []
    for v in range(V):
        st = po.PulseOsc() if kind == "pulse" else po.TriSawOsc()
        (L.zo_pulseosc_init if kind == "pulse" else L.zo_trisawosc_init)(C.byref(st))
        fn = L.zo_pulseosc_paint if kind == "pulse" else L.zo_trisawosc_paint
        for b in range(nbuf):
            fn(C.byref(st), span[0], span[1], po.fptr(outs[b, v]), SR, po.constant(freq[v]), float(color[v]))
        cnts.append(st.cnt)
    return outs, cnts


@pytest.mark.parametrize("kind", ["pulse", "trisaw"])
@pytest.mark.parametrize("nbuf,span,zf", [(5, (0, 1024), True), (40, (0, 256), True), (3, (100, 901), False)])
def test_paint_batch_equals_consecutive_paints(ctx, oracle, kind, nbuf, span, zf):
    """zh_*_paint_batch: one launch for nbuf buffers (more than kOscMaxBatch = 32 splits into two) == nbuf paints."""
    from zang_amd import modules as mod, zang, workloads
    V, F = 512, 1024
    freq, color, _, _ = workloads.voice_params(2, 7, V)
    freq[:3] = [6000.5, -1.0, 440.0]                     # silent voices in a batch too
    out0 = util.rng_buffers(21, V, F)
    ref, rcnt = _oracle_buffers(oracle, kind, V, F, span, nbuf, freq, color, out0=None if zf else out0)
    m = (mod.PulseOsc if kind == "pulse" else mod.TriSawOsc)(V, ctx)
    imgs = [util.to_image(out0) for _ in range(nbuf)]
    fr, col = util.dev(freq), util.dev(color)
    m.paint_batch(zang.Span(*span), imgs, m.Params(SR, zang.constant(fr), col), zero_first=zf)
    ctx.sync()
    for b in range(nbuf):
        got = util.from_image(imgs[b])[:, span[0]:span[1]]
        util.assert_bitexact(got, ref[b][:, span[0]:span[1]], f"{kind} batch buffer {b}")
        # frames outside the span are untouched
        util.assert_bitexact(util.from_image(imgs[b])[:, :span[0]], out0[:, :span[0]], "before span")
        util.assert_bitexact(util.from_image(imgs[b])[:, span[1]:], out0[:, span[1]:], "after span")
    assert [int(x) for x in m.state()["cnt"]] == rcnt


@pytest.mark.parametrize("kind", ["pulse", "trisaw"])
def test_params_unchanged_flag(ctx, oracle, kind):
    """ZH_PAINT_PARAMS_UNCHANGED: the table of per-voice constants written by the previous paint gives the same bits;
    a flagged paint whose scalar params / pointers differ from the stored ones falls back to computing them."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    V, F = 1024, 1024
    freq, color, _, _ = workloads.voice_params(2, 3, V)
    freq[5] = 6100.0
    freq2 = (freq * 1.5).astype(np.float32)
    m = (mod.PulseOsc if kind == "pulse" else mod.TriSawOsc)(V, ctx)
    fr, col, fr2 = util.dev(freq), util.dev(color), util.dev(freq2)
    imgs = [ctx.image(F, V) for _ in range(6)]
    sp = zang.Span(0, F)
    P = m.Params
    m.paint(sp, [imgs[0]], [], False, P(SR, zang.constant(fr), col), zero_first=True)                       # stores the table
    m.paint(sp, [imgs[1]], [], False, P(SR, zang.constant(fr), col), zero_first=True, params_unchanged=True)  # loads it
    m.paint(sp, [imgs[2]], [], False, P(SR, zang.constant(fr), col), zero_first=True, params_unchanged=True)
    # flagged, but another array: the stored table does not match -> computed (and stored) again
    m.paint(sp, [imgs[3]], [], False, P(SR, zang.constant(fr2), col), zero_first=True, params_unchanged=True)
    m.paint(sp, [imgs[4]], [], False, P(SR, zang.constant(fr2), col), zero_first=True, params_unchanged=True)
    # flagged, another sample rate
    m.paint(sp, [imgs[5]], [], False, P(44100.0, zang.constant(fr2), col), zero_first=True, params_unchanged=True)
    ctx.sync()
    L = oracle.lib()
    ref = np.zeros((6, V, F), np.float32)
    plan = [(SR, freq), (SR, freq), (SR, freq), (SR, freq2), (SR, freq2), (44100.0, freq2)]
    for v in range(V):
        st = oracle.PulseOsc() if kind == "pulse" else oracle.TriSawOsc()
        (L.zo_pulseosc_init if kind == "pulse" else L.zo_trisawosc_init)(C.byref(st))
        fn = L.zo_pulseosc_paint if kind == "pulse" else L.zo_trisawosc_paint
        for b, (sr, f) in enumerate(plan):
            fn(C.byref(st), 0, F, oracle.fptr(ref[b, v]), sr, oracle.constant(f[v]), float(color[v]))
    for b in range(6):
        util.assert_bitexact(util.from_image(imgs[b]), ref[b], f"{kind} params_unchanged buffer {b}")
