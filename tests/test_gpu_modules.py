"""GPU parity: every stateful/stateless module kernel vs the oracle, through the C ABI.

Bar (BASELINE.json north_star): 1e-5 relative f32 (tests/util.py FLOOR documents the
zero-crossing floor); Gate and Decimator bit-exact.  In practice every module here is
bit-exact because the device runs the same published algorithms without FMA contraction;
the tests assert bit-exactness wherever that holds by construction and additionally state
the tolerance for the libm-dependent modules (SineOsc, Distortion).
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024
SPANS = [util.SPANS_ONE, util.SPANS_THREE]


def _cob(po, kind, const_v, buf_v):
    return po.constant(const_v) if kind == "c" else po.buffer(buf_v)


def _gcob(zang, kind, const_t, buf_t):
    return zang.constant(const_t) if kind == "c" else zang.buffer(buf_t)


# ------------------------------------------------------------------ SineOsc
RANGE_SWITCHES = ("sine_ranges", "sampler_ranges", "decimator_ranges", "envelope_ranges",
                  "portamento_ranges", "trisaw_ctrl_ranges", "pulse_ctrl_ranges", "cycle_ranges", "curve_ranges")


@pytest.fixture(params=["ranges", "sequential"])
def replay_form(request, monkeypatch):
    """SineOsc / Sampler at a small voice count paint a span as many frame ranges at once (each range replays the f32 state
    additions of the frames before it); ZH_*_RANGES=0 selects the lane-per-voice walk instead.  Both forms must give the
    oracle's bits (the library reads the variables at every paint)."""
    if request.param == "sequential":
        for name in RANGE_SWITCHES:
            util.set_form(monkeypatch, **{name: 0})
    return request.param


@pytest.mark.parametrize("fk,pk", [("c", "c"), ("c", "b"), ("b", "c"), ("b", "b")])
@pytest.mark.parametrize("spans", SPANS)
def test_sineosc(ctx, oracle, fk, pk, spans, replay_form):
    from zang_amd import modules as mod, zang
    V = 192
    rng = np.random.default_rng(21)
    freq = rng.uniform(20, 6000, V).astype(np.float32)
    phase = rng.uniform(-1, 1, V).astype(np.float32)
    fbuf = rng.uniform(20, 8000, (V, F)).astype(np.float32)
    pbuf = rng.uniform(-2, 2, (V, F)).astype(np.float32)
    out0 = util.rng_buffers(3, V, F)
    L = oracle.lib()
    ref = out0.copy(); rt = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st))
        for (s, e) in spans:
            L.zo_sineosc_paint(C.byref(st), s, e, oracle.fptr(ref[v]), SR, _cob(oracle, fk, freq[v], fbuf[v]), _cob(oracle, pk, phase[v], pbuf[v]))
        rt[v] = st.t
    m = mod.SineOsc(V, ctx)
    out = util.to_image(out0); gf, gp = util.to_image(fbuf), util.to_image(pbuf)
    cf, cp = util.dev(freq), util.dev(phase)
    for (s, e) in spans:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(SR, _gcob(zang, fk, cf, gf), _gcob(zang, pk, cp, gp)))
    ctx.sync()
    got = util.from_image(out)
    util.assert_close(got, ref, "sineosc")
    util.assert_bitexact(got, ref, "sineosc (same algorithm, expected exact)")
    util.assert_bitexact(m.state()["t"].astype(np.float32), rt, "sineosc t")


def test_sineosc_large_phase_and_nan(ctx, oracle):
    """Arguments beyond 2^28*pi/2 take the Payne-Hanek path; inf/NaN give NaN."""
    from zang_amd import modules as mod, zang
    V = 64
    phase = (np.random.default_rng(5).uniform(1, 2, V) * 2.0 ** np.random.default_rng(6).integers(20, 100, V)).astype(np.float32)
    phase[0], phase[1], phase[2] = np.inf, np.nan, -1e30
    L = oracle.lib()
    ref = np.zeros((V, 16), np.float32)
    for v in range(V):
        st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st))
        L.zo_sineosc_paint(C.byref(st), 0, 16, oracle.fptr(ref[v]), SR, oracle.constant(440.0), oracle.constant(phase[v]))
    m = mod.SineOsc(V, ctx)
    out = ctx.image(16, V)
    m.paint(zang.Span(0, 16), [out], [], False, m.Params(SR, zang.constant(440.0), zang.constant(util.dev(phase))), zero_first=True)
    ctx.sync()
    got = util.from_image(out)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    util.assert_bitexact(got[ok], ref[ok], "sineosc huge args")


# ------------------------------------------------------------------ Noise
@pytest.mark.parametrize("color", [0, 1])
def test_noise(ctx, oracle, color):
    """Three consecutive paints: pink must restart its taps each paint (Noise.zig:68 quirk)."""
    from zang_amd import modules as mod, zang
    V, first = 256, 1000
    out0 = util.rng_buffers(8, V, F)
    L = oracle.lib()
    ref = out0.copy()
    rstate = []
    for v in range(V):
        st = oracle.Noise(); L.zo_noise_init(C.byref(st), first + v)
        for (s, e) in util.SPANS_THREE:
            L.zo_noise_paint(C.byref(st), s, e, oracle.fptr(ref[v]), color)
        rstate.append(list(st.r))
    m = mod.Noise(V, ctx, first_seed=first)
    out = util.to_image(out0)
    for (s, e) in util.SPANS_THREE:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(color))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"noise color {color}")
    gs = m.state()
    assert [[int(x) for x in row] for row in gs["r"]] == rstate
    assert not gs["b"].any()


@pytest.mark.parametrize("fused", [False, True])
def test_noise_rare_float_paths(ctx, oracle, fused):
    """Random.float(f32)'s rare branches, reached by crafting the xoshiro state (next() = rotl(s0+s3, 23) + s0):
    a draw whose high word is 0 (probability 2^-32: the device takes its general leading-zero form behind a
    wave-uniform test) with 32..40 leading zeros, and a draw below 2^23 (2^-41: a second draw refines the exponent),
    0 included -- mixed with ordinary lanes in the same wave, in Noise and in the fused Noise->Filter voice."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(41)
    s3 = [1, 2, 255, 256, 511, 512,                       # rnd = s3 << 23: 40, 39, 33, 32, 32 leading zeros; 512 is ordinary
          1 << 41, 5 << 41, (2 ** 23 - 1) << 41, 0,       # rnd = s3 >> 41 < 2^23: second draw; rnd == 0
          (1 << 41) | 1]                                  # rnd = 2^23 + 1: back to the first rare form
    V = 64 + len(s3)
    L = oracle.lib()
    nzs = []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), 900 + v)
        if v >= 64:
            nz.r[0] = 0; nz.r[3] = s3[v - 64]
            nz.r[1] = int(rng.integers(1, 1 << 63)); nz.r[2] = int(rng.integers(1, 1 << 63))
        nzs.append(nz)
    nframes = 40
    if fused:
        m = mod.NoiseFilter(V, ctx, first_seed=900)
        st = m.state()
        for v in range(64, V):
            st["noise"]["r"][v] = [int(x) for x in nzs[v].r]
    else:
        m = mod.Noise(V, ctx, first_seed=900)
        st = m.state()
        for v in range(64, V):
            st["r"][v] = [int(x) for x in nzs[v].r]
    m.set_state(st)
    ref = np.zeros((V, F), np.float32)
    temp = np.zeros(F, np.float32)
    cutoff = np.full(V, 0.3, np.float32); res = np.full(V, 0.2, np.float32)
    for v in range(V):
        if fused:
            fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
            L.zo_zero(0, nframes, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nzs[v]), 0, nframes, oracle.fptr(temp), 0)
            L.zo_filter_paint(C.byref(fl), 0, nframes, oracle.fptr(ref[v]), oracle.fptr(temp), 0, oracle.constant(cutoff[v]), oracle.constant(res[v]))
        else:
            L.zo_noise_paint(C.byref(nzs[v]), 0, nframes, oracle.fptr(ref[v]), 0)
    out = ctx.image(F, V, fill=0.0)
    if fused:
        m.paint(zang.Span(0, nframes), [out], None, False, m.Params(0, 0, util.dev(cutoff), util.dev(res)))
    else:
        m.paint(zang.Span(0, nframes), [out], [], False, m.Params(0))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, "noise rare float paths")
    gs = m.state()
    got_r = gs["noise"]["r"] if fused else gs["r"]
    assert [[int(x) for x in row] for row in got_r] == [list(n.r) for n in nzs]      # the extra draws advanced the generator
    first = util.from_image(out)[64:, 0] if not fused else None
    if first is not None:
        assert (np.abs(first[:5]) > 0.99).all()            # tiny random floats map to white ~ -1


def _xoshiro_step_back(st, k):
    """The state k transitions BEFORE `st` (xoshiro256++'s transition is invertible: a = rotr(n3, 45) = s3 ^ s1,
    s0 = n0 ^ a, n1 ^ n2 = s1 ^ (s1 << 17), ...)."""
    M = (1 << 64) - 1
    n0, n1, n2, n3 = st
    for _ in range(k):
        a = ((n3 >> 45) | (n3 << 19)) & M
        s0 = n0 ^ a
        x = n1 ^ n2
        s1 = (x ^ (x << 17) ^ (x << 34) ^ (x << 51)) & M
        s3 = a ^ s1
        s2 = n1 ^ s1 ^ s0
        n0, n1, n2, n3 = s0, s1, s2, s3
    return [n0, n1, n2, n3]


@pytest.mark.parametrize("zero_first", [True, False])
@pytest.mark.parametrize("V", [300, 4096])
def test_noise_white_frame_ranges(ctx, oracle, zero_first, V):
    """Few voices: white noise is painted as many frame ranges at once, each from a state jumped ahead with the
    T^(32 j) tables (csrc/noise_jump.hip).  Bit for bit the oracle's sequential walk, states included -- also for voices
    whose span holds one of Random.float's multi-draw samples (2^-41 per sample; crafted here by stepping a state with
    s0 = 0, s3 = 2^41 BACK k transitions, so that the sample lands on frame k of the first span: in the first range, at
    a range boundary, in a middle range, on the last frame), which the ranges flag and k_noise_fix repaints."""
    from zang_amd import modules as mod, zang
    first = 4242
    rng = np.random.default_rng(77)
    L = oracle.lib()
    ks = [0, 31, 32, 500, 777, 1023]
    nzs = []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v)
        nzs.append(nz)
    crafted = {10 + 37 * i: k for i, k in enumerate(ks)}
    for v, k in crafted.items():
        target = [0, int(rng.integers(1, 1 << 63)), int(rng.integers(1, 1 << 63)), 1 << 41]     # next(): rnd = 1 -> 63 leading zeros
        back = _xoshiro_step_back(target, k)
        for i in range(4):
            nzs[v].r[i] = back[i]
    m = mod.Noise(V, ctx, first_seed=first)
    st = m.state()
    for v in crafted:
        st["r"][v] = [int(x) for x in nzs[v].r]
    m.set_state(st)
    out0 = util.rng_buffers(12, V, F)
    spans = [(0, 1024), (0, 1024), (100, 612), (612, 1000)]          # full buffers, then two shorter spans (carried state)
    for (s, e) in spans:
        ref = out0.copy()
        if zero_first:
            ref[:, s:e] = 0.0
        for v in range(V):
            L.zo_noise_paint(C.byref(nzs[v]), s, e, oracle.fptr(ref[v]), 0)
        out = util.to_image(out0)
        m.paint(zang.Span(s, e), [out], [], False, m.Params(m.white), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"white noise ranges, span {(s, e)}")
        gs = m.state()
        assert [[int(x) for x in row] for row in gs["r"]] == [list(n.r) for n in nzs], f"states after span {(s, e)}"


def test_noise_seed_known_answer(ctx):
    """K1 of SURVEY.md 8c: seed 0 / seed 1 white samples."""
    from zang_amd import modules as mod, zang
    m = mod.Noise(2, ctx, first_seed=0)
    out = ctx.image(4, 2)
    m.paint(zang.Span(0, 4), [out], [], False, m.Params(m.white), zero_first=True)
    ctx.sync()
    got = util.from_image(out)
    k = np.array([[-0.4564839005470276, -0.4967494606971741, -0.3965456485748291, -0.9767617583274841],
                  [0.02937638759613037, 0.49904024600982666, -0.8330492973327637, 0.6042125225067139]], np.float32)
    util.assert_bitexact(got, k, "noise K1")


@pytest.mark.parametrize("form", ["taps", "taps16", "chain", "sequential"])
@pytest.mark.parametrize("zero_first", [True, False])
@pytest.mark.parametrize("V", [300, 4096])
def test_noise_pink_pipeline(ctx, oracle, zero_first, V, form, monkeypatch):
    """Few voices: pink noise = the white samples painted as frame ranges into a module-owned image, then Kellett's filter
    as a seven-wave pipeline per 64 voices (six taps adding into a running sum through LDS, a final wave; k_pink_pipe).
    Bit for bit the oracle's one loop -- with non-zero taps in the state (the reference starts every paint from self.b and
    never writes it back, Noise.zig:55/68), a voice whose span holds a multi-draw sample, ragged spans; ZH_PINK_PIPE_MAX=0 is
    the lane-per-voice kernel."""
    from zang_amd import modules as mod, zang
    if form == "sequential":
        util.set_form(monkeypatch, pink_pipe_max="0")
    elif form == "chain":
        util.set_form(monkeypatch, pink_taps="0")                   # the seven-stage chain k_pink_pipe instead of the four-wave k_pink_taps
    elif form == "taps16":
        util.set_form(monkeypatch, pink_taps="16")                  # k_pink_taps with the 16-frame tiles it takes above 16,384 voices
    first = 777
    rng = np.random.default_rng(78)
    L = oracle.lib()
    nzs = []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v)
        nzs.append(nz)
    target = [0, int(rng.integers(1, 1 << 63)), int(rng.integers(1, 1 << 63)), 1 << 41]
    back = _xoshiro_step_back(target, 333)
    for i in range(4):
        nzs[17].r[i] = back[i]
    taps = rng.uniform(-0.5, 0.5, (V, 7)).astype(np.float32)
    taps[::3] = 0.0
    for v in range(V):
        for j in range(7):
            nzs[v].b[j] = float(taps[v, j])
    m = mod.Noise(V, ctx, first_seed=first)
    st = m.state()
    st["r"][17] = [int(x) for x in nzs[17].r]
    st["b"][:] = taps
    m.set_state(st)
    out0 = util.rng_buffers(13, V, F)
    for (s, e) in [(0, 1024), (0, 1024), (100, 612), (612, 1000), (5, 170)]:
        ref = out0.copy()
        if zero_first:
            ref[:, s:e] = 0.0
        for v in range(V):
            L.zo_noise_paint(C.byref(nzs[v]), s, e, oracle.fptr(ref[v]), 1)
        out = util.to_image(out0)
        m.paint(zang.Span(s, e), [out], [], False, m.Params(m.pink), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"pink noise {form}, span {(s, e)}")
        gs = m.state()
        assert [[int(x) for x in row] for row in gs["r"]] == [list(n.r) for n in nzs], f"states after span {(s, e)}"
        util.assert_bitexact(gs["b"].astype(np.float32), taps, "taps are never written back")


# ------------------------------------------------------------------ Envelope
def _env_case(oracle, ctx, V, curves, sustain, script, dur_scale=1.0):
    """script: list of (span, note_on[V] bool array, note_id_changed[V] bool array)."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(31)
    dur = [(rng.uniform(0.0005, 0.02, V) * dur_scale).astype(np.float32) for _ in range(3)]
    sus = np.full(V, sustain, np.float32)
    out0 = util.rng_buffers(9, V, F)
    L = oracle.lib()
    ref = out0.copy(); rst = []
    for v in range(V):
        st = oracle.Envelope(); L.zo_envelope_init(C.byref(st))
        for ((s, e), on, nic) in script:
            p = oracle.EnvelopeParams(SR, oracle.curve(curves[0], dur[0][v]), oracle.curve(curves[1], dur[1][v]),
                                      oracle.curve(curves[2], dur[2][v]), float(sus[v]), int(on[v]))
            L.zo_envelope_paint(C.byref(st), s, e, oracle.fptr(ref[v]), int(nic[v]), C.byref(p))
        rst.append((st.state, st.painter.t, st.painter.last_value, st.painter.start))
    m = mod.Envelope(V, ctx)
    out = util.to_image(out0)
    mk = [None, zang.PaintCurve.linear, zang.PaintCurve.squared, zang.PaintCurve.cubed]
    gc = [zang.PaintCurve.instantaneous if curves[i] == 0 else mk[curves[i]](util.dev(dur[i])) for i in range(3)]
    gs = util.dev(sus)
    for ((s, e), on, nic) in script:
        m.paint(zang.Span(s, e), [out], [], util.dev(nic.astype(np.uint8)),
                m.Params(SR, gc[0], gc[1], gc[2], gs, util.dev(on.astype(np.uint8))))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"envelope {curves}")
    st = m.state()
    assert [int(x) for x in st["state"]] == [r[0] for r in rst]
    for name, k in (("t", 1), ("last_value", 2), ("start", 3)):
        util.assert_bitexact(st[name].astype(np.float32), np.array([r[k] for r in rst], np.float32), f"envelope {name}")


@pytest.mark.parametrize("curves", [(1, 1, 1), (2, 2, 2), (3, 3, 3), (1, 2, 3), (0, 3, 3), (3, 0, 3), (3, 3, 0), (0, 0, 0)])
@pytest.mark.parametrize("sustain", [0.5, 1.0])
def test_envelope_stages(ctx, oracle, curves, sustain, replay_form):
    V = 128
    rng = np.random.default_rng(41)
    on = np.ones(V, bool); off = np.zeros(V, bool)
    new = np.ones(V, bool); same = np.zeros(V, bool)
    mixed_on = rng.random(V) < 0.5
    # note_on voices whose envelope is in `release` need a new note id (the reference asserts it)
    script = [((0, 200), on, new), ((200, 777), on, same), ((777, 1024), mixed_on, same),
              ((0, 1024), off, same), ((0, 300), mixed_on, mixed_on), ((300, 300), on, new), ((300, 1024), off, same)]
    _env_case(oracle, ctx, V, curves, sustain, script)


def test_envelope_off_while_idle_and_release_note_on(ctx, oracle):
    """paintOff while idle paints nothing (Envelope.zig:78-80); note_on during release without a
    new note id is the reference's assert case -- defined as painting nothing."""
    V = 64
    on = np.ones(V, bool); off = np.zeros(V, bool); new = np.ones(V, bool); same = np.zeros(V, bool)
    script = [((0, 512), off, same), ((0, 100), on, new), ((100, 150), off, same), ((150, 400), on, same), ((400, 1024), on, new)]
    _env_case(oracle, ctx, V, (3, 3, 3), 0.8, script, dur_scale=4.0)


# ------------------------------------------------------------------ Gate
@pytest.mark.parametrize("V", [200, 201, 1028])     # 201: one voice per lane (k_gate); the others four per lane (k_gate4)
def test_gate_bitexact(ctx, oracle, V):
    from zang_amd import modules as mod, zang
    on = np.random.default_rng(2).random(V) < 0.5
    out0 = util.rng_buffers(4, V, F)
    ref = out0.copy()
    L = oracle.lib()
    for v in range(V):
        L.zo_gate_paint(10, 1000, oracle.fptr(ref[v]), int(on[v]))
    m = mod.Gate(V, ctx)
    out = util.to_image(out0)
    m.paint(zang.Span(10, 1000), [out], [], False, m.Params(util.dev(on.astype(np.uint8))))
    out2 = util.to_image(out0)
    m.paint(zang.Span(10, 1000), [out2], [], False, m.Params(util.dev(on.astype(np.uint8))), zero_first=True)
    ctx.sync()
    assert ctx.last_form() == ["k_gate4" if V % 4 == 0 else "k_gate"]
    util.assert_bitexact(util.from_image(out), ref, "gate")
    ref2 = out0.copy(); ref2[:, 10:1000] = on[:, None].astype(np.float32)
    util.assert_bitexact(util.from_image(out2), ref2, "gate zero_first")


# ------------------------------------------------------------------ Filter
@pytest.mark.parametrize("form", ["pipeline", "pipeline16", "walk"])
@pytest.mark.parametrize("zero_first", [False, True])
@pytest.mark.parametrize("ftype", range(6))
def test_filter_const_params_both_forms(ctx, oracle, ftype, zero_first, form, monkeypatch):
    """Constant cutoff / resonance at a small voice count: the three-wave pipeline k_filter_pc (loader, recurrence, writer)
    with 32-frame tiles, with the 16-frame tiles it takes between 32,768 and 65,536 voices (forced here: ZH_FILTER_PC_MAX=1)
    and the one-wave walk (ZH_FILTER_PC_MAX=0) against the oracle -- every filter type, += and ZERO_FIRST,
    ragged spans (a span shorter than 64 frames takes the walk anyway), a voice count that is not a multiple of 64, carried state."""
    from zang_amd import modules as mod, zang
    if form == "walk":
        util.set_form(monkeypatch, filter_pc_max="0")
    elif form == "pipeline16":
        util.set_form(monkeypatch, filter_pc_max="1")
    V = 200
    rng = np.random.default_rng(54)
    cut = rng.uniform(-0.1, 1.1, V).astype(np.float32)
    res = rng.uniform(-0.1, 1.1, V).astype(np.float32)
    inp = util.rng_buffers(55, V, F)
    out0 = util.rng_buffers(56, V, F)
    spans = [(0, 1024), (0, 1024), (100, 612), (612, 1000), (5, 170), (170, 200), (200, 329)]
    L = oracle.lib()
    ref = out0.copy(); rl = np.zeros(V, np.float32); rb = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.Filter(); L.zo_filter_init(C.byref(st))
        for (s, e) in spans:
            if zero_first:
                ref[v][s:e] = 0.0
            L.zo_filter_paint(C.byref(st), s, e, oracle.fptr(ref[v]), oracle.fptr(inp[v]), ftype, oracle.constant(cut[v]), oracle.constant(res[v]))
        rl[v], rb[v] = st.l, st.b
    m = mod.Filter(V, ctx)
    out = util.to_image(out0); gi = util.to_image(inp)
    dc, dr = util.dev(cut), util.dev(res)
    for (s, e) in spans:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(gi, ftype, zang.constant(dc), zang.constant(dr)), zero_first=zero_first)
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"filter type {ftype} {form}")
    st = m.state()
    util.assert_bitexact(st["l"].astype(np.float32), rl, "filter l")
    util.assert_bitexact(st["b"].astype(np.float32), rb, "filter b")


@pytest.mark.parametrize("ftype", range(6))
@pytest.mark.parametrize("ck,rk", [("c", "c"), ("c", "b"), ("b", "c"), ("b", "b")])
def test_filter(ctx, oracle, ftype, ck, rk):
    from zang_amd import modules as mod, zang
    V = 128
    rng = np.random.default_rng(51)
    cut = rng.uniform(-0.1, 1.1, V).astype(np.float32)       # outside [0,1]: clamped
    res = rng.uniform(-0.1, 1.1, V).astype(np.float32)
    cbuf = rng.uniform(-0.1, 1.1, (V, F)).astype(np.float32)
    rbuf = rng.uniform(-0.1, 1.1, (V, F)).astype(np.float32)
    inp = util.rng_buffers(52, V, F)
    out0 = util.rng_buffers(53, V, F)
    L = oracle.lib()
    ref = out0.copy(); rl = np.zeros(V, np.float32); rb = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.Filter(); L.zo_filter_init(C.byref(st))
        for (s, e) in util.SPANS_THREE:
            L.zo_filter_paint(C.byref(st), s, e, oracle.fptr(ref[v]), oracle.fptr(inp[v]), ftype,
                              _cob(oracle, ck, cut[v], cbuf[v]), _cob(oracle, rk, res[v], rbuf[v]))
        rl[v], rb[v] = st.l, st.b
    m = mod.Filter(V, ctx)
    out = util.to_image(out0); gi = util.to_image(inp); gc, gr = util.to_image(cbuf), util.to_image(rbuf)
    dc, dr = util.dev(cut), util.dev(res)
    for (s, e) in util.SPANS_THREE:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(gi, ftype, _gcob(zang, ck, dc, gc), _gcob(zang, rk, dr, gr)))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"filter type {ftype}")
    st = m.state()
    util.assert_bitexact(st["l"].astype(np.float32), rl, "filter l")
    util.assert_bitexact(st["b"].astype(np.float32), rb, "filter b")


@pytest.mark.parametrize("form", ["pipeline", "walk"])
@pytest.mark.parametrize("zero_first", [False, True])
@pytest.mark.parametrize("ck,rk", [("b", "c"), ("c", "b"), ("b", "b")])
@pytest.mark.parametrize("ftype", [1, 3, 5])
def test_filter_control_images_both_forms(ctx, oracle, ftype, ck, rk, zero_first, form, monkeypatch):
    """Cutoff and / or resonance as control images at a small voice count: the three-wave pipeline with the images' rows as tiles of
    their own (k_filter_pc_ctl: 32-frame tiles with one image, 16 with both) and the one-wave walk (ZH_FILTER_PC_CTL_MAX=0) against
    the oracle -- per-frame values outside [0, 1] (clamped), ragged spans, a span under 64 frames (the walk anyway), a voice count
    that is not a multiple of 64, += and ZERO_FIRST, carried state."""
    from zang_amd import modules as mod, zang
    if form == "walk":
        util.set_form(monkeypatch, filter_pc_ctl_max="0")
    V = 200
    rng = np.random.default_rng(57)
    cut = rng.uniform(-0.1, 1.1, V).astype(np.float32); res = rng.uniform(-0.1, 1.1, V).astype(np.float32)
    cbuf = rng.uniform(-0.1, 1.1, (V, F)).astype(np.float32); rbuf = rng.uniform(-0.1, 1.1, (V, F)).astype(np.float32)
    cbuf[:, 300:340] = np.linspace(0.0, 1.0, 40, dtype=np.float32)[None, :]          # a sweep inside the noise
    inp = util.rng_buffers(58, V, F); out0 = util.rng_buffers(59, V, F)
    spans = [(0, 1024), (0, 1024), (100, 612), (612, 1000), (5, 170), (170, 200), (200, 329), (0, 1007)]
    L = oracle.lib()
    ref = out0.copy(); rl = np.zeros(V, np.float32); rb = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.Filter(); L.zo_filter_init(C.byref(st))
        for (s, e) in spans:
            if zero_first:
                ref[v][s:e] = 0.0
            L.zo_filter_paint(C.byref(st), s, e, oracle.fptr(ref[v]), oracle.fptr(inp[v]), ftype, _cob(oracle, ck, cut[v], cbuf[v]), _cob(oracle, rk, res[v], rbuf[v]))
        rl[v], rb[v] = st.l, st.b
    m = mod.Filter(V, ctx)
    out = util.to_image(out0); gi = util.to_image(inp); gc, gr = util.to_image(cbuf), util.to_image(rbuf)
    dc, dr = util.dev(cut), util.dev(res)
    for (s, e) in spans:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(gi, ftype, _gcob(zang, ck, dc, gc), _gcob(zang, rk, dr, gr)), zero_first=zero_first)
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"filter type {ftype} cutoff {ck} res {rk} {form}")
    st = m.state()
    util.assert_bitexact(st["l"].astype(np.float32), rl, "filter l")
    util.assert_bitexact(st["b"].astype(np.float32), rb, "filter b")


def test_filter_and_distortion_in_place(ctx, oracle):
    """The input image IS the output image (`out += f(out)`): every frame's input is read before its output is written,
    in the oracle's scalar loop and in the device's chunked loops (a chunk's loads precede its stores; the next chunk's
    prefetched rows are not yet written) alike."""
    from zang_amd import modules as mod, zang
    V = 100
    buf0 = util.rng_buffers(61, V, F)
    L = oracle.lib()
    ref = buf0.copy()
    for v in range(V):
        st = oracle.Filter(); L.zo_filter_init(C.byref(st))
        for (s, e) in util.SPANS_THREE:
            L.zo_filter_paint(C.byref(st), s, e, oracle.fptr(ref[v]), oracle.fptr(ref[v]), mod.Filter.low_pass, oracle.constant(0.3), oracle.constant(0.4))
        L.zo_distortion_paint(0, F, oracle.fptr(ref[v]), oracle.fptr(ref[v]), mod.Distortion.clip, 0.5, 0.5, 0.0)
    img = util.to_image(buf0)
    m = mod.Filter(V, ctx)
    for (s, e) in util.SPANS_THREE:
        m.paint(zang.Span(s, e), [img], [], False, m.Params(img, m.low_pass, zang.constant(0.3), zang.constant(0.4)))
    d = mod.Distortion(V, ctx)
    d.paint(zang.Span(0, F), [img], [], False, d.Params(img, d.clip, 0.5, 0.5, 0.0))
    ctx.sync()
    util.assert_bitexact(util.from_image(img), ref, "in-place filter + distortion")


def test_filter_known_answer_and_cutoff(ctx, oracle):
    """K3 of SURVEY.md 8c and Filter.cutoffFromFrequency vs the oracle."""
    from zang_amd import modules as mod, zang
    m = mod.Filter(1, ctx)
    inp = util.to_image(np.array([[1, 0, 0, 0, 0, 0]], np.float32)); out = ctx.image(6, 1)
    m.paint(zang.Span(0, 6), [out], [], False, m.Params(inp, m.low_pass, zang.constant(0.5), zang.constant(0.7)), zero_first=True)
    ctx.sync()
    k3 = np.array([[0.2499980926513672, 0.8275015354156494, 0.5310308337211609, -0.1411801278591156,
                    -0.5050814747810364, -0.3323642313480377]], np.float32)
    util.assert_bitexact(util.from_image(out), k3, "filter K3")
    f = np.random.default_rng(1).uniform(0, 30000, 4096).astype(np.float32)
    got = mod.Filter.cutoffFromFrequency(util.dev(f), SR, ctx).cpu().numpy()
    L = oracle.lib()
    ref = np.array([L.zo_filter_cutoff_from_frequency(float(x), SR) for x in f], np.float32)
    util.assert_bitexact(got, ref, "cutoffFromFrequency")


# ------------------------------------------------------------------ Sampler
def _pcm(fmt, nframes, channels, seed):
    rng = np.random.default_rng(seed)
    bps = fmt + 1
    return rng.integers(0, 256, nframes * channels * bps, dtype=np.uint8)


@pytest.mark.parametrize("fmt", range(4))
@pytest.mark.parametrize("loop", [False, True])
def test_sampler(ctx, oracle, fmt, loop, replay_form):
    """Per-voice output rates cover: ratio ~ 1 (integer copy), up/down-sampling, negative ratio."""
    from zang_amd import modules as mod, zang
    V, channels, in_rate = 96, 2, 44100
    data = _pcm(fmt, 700, channels, 60 + fmt)
    rng = np.random.default_rng(61)
    rate = rng.uniform(8000, 96000, V).astype(np.float32)
    rate[:8] = [44100.0, 44100.5, 44099.0, -44100.0, -22050.0, 22050.0, 88200.0, 44100.0]
    nic_script = [np.zeros(V, bool), rng.random(V) < 0.3, np.zeros(V, bool)]
    out0 = util.rng_buffers(62, V, F)
    L = oracle.lib()
    ref = out0.copy(); rt = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.Sampler(); L.zo_sampler_init(C.byref(st))
        for k, (s, e) in enumerate(util.SPANS_THREE):
            p = oracle.SamplerParams(float(rate[v]), channels, in_rate, fmt, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, 1, int(loop))
            L.zo_sampler_paint(C.byref(st), s, e, oracle.fptr(ref[v]), int(nic_script[k][v]), C.byref(p))
        rt[v] = st.t
    m = mod.Sampler(V, ctx)
    out = util.to_image(out0)
    smp = m.Sample(channels, in_rate, fmt, util.dev(data))
    gr = util.dev(rate)
    for k, (s, e) in enumerate(util.SPANS_THREE):
        m.paint(zang.Span(s, e), [out], [], util.dev(nic_script[k].astype(np.uint8)), m.Params(gr, smp, 1, loop))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"sampler fmt {fmt} loop {loop}")
    util.assert_bitexact(m.state()["t"].astype(np.float32), rt, "sampler t")


@pytest.mark.parametrize("fmt,channels", [(0, 1), (0, 2), (0, 4), (1, 1), (1, 2), (3, 1)])
@pytest.mark.parametrize("loop", [False, True])
@pytest.mark.parametrize("nframes", [2, 3, 5, 64, 700])
def test_sampler_pair_loads(ctx, oracle, fmt, channels, loop, nframes, replay_form):
    """Frames of 1, 2 or 4 bytes: the interpolation's two samples come from one load of two frames (k_sampler's pair path).
    Sample lengths down to two frames (wraps every frame; |ratio| + 1 >= n sends a voice's wave through the general body),
    ratios from 1/12 to 6 and negative ones, play positions that start beyond the end and before the start, an s16 base
    that is not 2-byte aligned (general body), the last channel of a frame, a retrigger in the middle buffer."""
    from zang_amd import modules as mod, zang
    V, in_rate = 134, 44100
    raw = _pcm(fmt, nframes + 1, channels, 160 + fmt)
    rng = np.random.default_rng(nframes * 7 + fmt)
    rate = (in_rate / rng.uniform(1.0 / 12.0, 6.0, V)).astype(np.float32)
    rate[:6] = [-44100.0, -9000.0, 44100.0, 44100.5, 500000.0, 7000.0]
    t_start = rng.uniform(-3.0, nframes + 3.0, V).astype(np.float32)
    t_start[70:] = 0.0
    nic_script = [np.zeros(V, bool), rng.random(V) < 0.3, np.zeros(V, bool)]
    for misaligned in ([False, True] if fmt == 1 else [False]):
        data = np.ascontiguousarray(raw[1 if misaligned else 0:][: nframes * channels * (fmt + 1)])
        out0 = util.rng_buffers(62, V, F)
        L = oracle.lib()
        ref = out0.copy(); rt = np.zeros(V, np.float32)
        for v in range(V):
            st = oracle.Sampler(); L.zo_sampler_init(C.byref(st)); st.t = float(t_start[v])
            for k, (s, e) in enumerate(util.SPANS_THREE):
                p = oracle.SamplerParams(float(rate[v]), channels, in_rate, fmt, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, channels - 1, int(loop))
                L.zo_sampler_paint(C.byref(st), s, e, oracle.fptr(ref[v]), int(nic_script[k][v]), C.byref(p))
            rt[v] = st.t
        m = mod.Sampler(V, ctx)
        st = m.state(); st["t"] = t_start; m.set_state(st)
        out = util.to_image(out0)
        if misaligned:                                 # a device buffer whose sample base sits at an odd address
            import torch
            holder = torch.zeros(data.size + 1, dtype=torch.uint8, device="cuda")
            holder[1:] = torch.from_numpy(data).cuda()
            dev_data = holder[1:]
            assert dev_data.data_ptr() % 2 == 1
        else:
            dev_data = util.dev(data)
        smp = m.Sample(channels, in_rate, fmt, dev_data)
        gr = util.dev(rate)
        for k, (s, e) in enumerate(util.SPANS_THREE):
            m.paint(zang.Span(s, e), [out], [], util.dev(nic_script[k].astype(np.uint8)), m.Params(gr, smp, channels - 1, loop))
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"sampler fmt {fmt} x {channels} loop {loop} n {nframes} misaligned {misaligned}")
        util.assert_bitexact(m.state()["t"].astype(np.float32), rt, "sampler t")


@pytest.mark.parametrize("fmt", [1, 3])
def test_sampler_huge_play_positions(ctx, oracle, fmt, replay_form):
    """Play positions beyond the i32 range (set through the state): floor(t) converts to the saturation value, t0 + 1 wraps
    to INT32_MIN -- the looped second tap's index is NOT the first one's + 1 there (the device derives it that way in every
    other case) -- and positions just below, negative ones, and ordinary ones in the same wave."""
    from zang_amd import modules as mod, zang
    V, channels, in_rate = 70, 1, 44100
    data = _pcm(fmt, 333, channels, 77)
    t0 = np.linspace(-5000.0, 5000.0, V).astype(np.float32)
    t0[:10] = [3.0e9, 2.2e9, 2147483520.0, 2147483648.0, -3.0e9, -2147483648.0, 4.0e9, 1.0e10, -1.0e10, 2147483392.0]
    rate = np.full(V, 30000.0, np.float32); rate[::3] = 50000.0
    out0 = util.rng_buffers(91, V, F)
    L = oracle.lib()
    ref = out0.copy(); rt = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.Sampler(); L.zo_sampler_init(C.byref(st)); st.t = float(t0[v])
        p = oracle.SamplerParams(float(rate[v]), channels, in_rate, fmt, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, 0, 1)
        for (s, e) in [(0, 1024), (100, 700)]:
            L.zo_sampler_paint(C.byref(st), s, e, oracle.fptr(ref[v]), 0, C.byref(p))
        rt[v] = st.t
    m = mod.Sampler(V, ctx)
    st = m.state(); st["t"] = t0; m.set_state(st)
    out = util.to_image(out0)
    smp = m.Sample(channels, in_rate, fmt, util.dev(data))
    for (s, e) in [(0, 1024), (100, 700)]:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(util.dev(rate), smp, 0, True))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"sampler huge positions fmt {fmt}")
    util.assert_bitexact(m.state()["t"].astype(np.float32), rt, "t")


def test_sampler_channel_out_of_range(ctx):
    from zang_amd import modules as mod, zang
    m = mod.Sampler(8, ctx)
    data = util.dev(_pcm(1, 64, 1, 3))
    out = ctx.image(32, 8, fill=2.0)
    st = m.state(); st["t"] = 5.0
    m.set_state(st)
    m.paint(zang.Span(0, 32), [out], [], True, m.Params(44100.0, m.Sample(1, 44100, 1, data), 1, False))
    ctx.sync()
    assert float(out.min()) == 2.0 and float(out.max()) == 2.0
    assert (m.state()["t"] == 5.0).all()           # Sampler.zig:87-89 returns before the reset


# ------------------------------------------------------------------ Decimator
def test_decimator_bitexact(ctx, oracle, replay_form):
    from zang_amd import modules as mod, zang
    V = 160
    rng = np.random.default_rng(71)
    fake = rng.uniform(100, 47000, V).astype(np.float32)
    fake[:6] = [48000.0, 96000.0, 0.0, -5.0, 24000.0, 6000.0]
    inp = util.rng_buffers(72, V, F); out0 = util.rng_buffers(73, V, F)
    L = oracle.lib()
    ref = out0.copy(); rs = np.zeros((V, 2), np.float32)
    for v in range(V):
        st = oracle.Decimator(); L.zo_decimator_init(C.byref(st))
        for (s, e) in util.SPANS_THREE:
            L.zo_decimator_paint(C.byref(st), s, e, oracle.fptr(ref[v]), SR, oracle.fptr(inp[v]), float(fake[v]))
        rs[v] = (st.dval, st.dcount)
    m = mod.Decimator(V, ctx)
    out = util.to_image(out0); gi = util.to_image(inp); gf = util.dev(fake)
    for (s, e) in util.SPANS_THREE:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(SR, gi, gf))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, "decimator")
    st = m.state()
    util.assert_bitexact(st["dval"].astype(np.float32), rs[:, 0].copy(), "dval")
    util.assert_bitexact(st["dcount"].astype(np.float32), rs[:, 1].copy(), "dcount")


def test_decimator_unusual_states(ctx, oracle, replay_form):
    """States the module never produces itself (set through set_state): dcount above 1, negative, huge -- the frame-range
    replay's one-instruction wrap (dc - floor(dc)) does not apply and the wave takes the reference's compare-and-subtract."""
    from zang_amd import modules as mod, zang
    V = 96
    rng = np.random.default_rng(74)
    fake = rng.uniform(300, 47000, V).astype(np.float32)
    dcount0 = rng.uniform(0, 1, V).astype(np.float32)
    dcount0[:8] = [5.0, -3.0, 1.0, 0.0, 2.5, 1.0e6, -0.25, 1.5]
    dval0 = rng.uniform(-1, 1, V).astype(np.float32)
    inp = util.rng_buffers(75, V, F); out0 = util.rng_buffers(76, V, F)
    L = oracle.lib()
    ref = out0.copy(); rs = np.zeros((V, 2), np.float32)
    for v in range(V):
        st = oracle.Decimator(); L.zo_decimator_init(C.byref(st)); st.dval = float(dval0[v]); st.dcount = float(dcount0[v])
        for (s, e) in [(0, 1024), (100, 700)]:
            L.zo_decimator_paint(C.byref(st), s, e, oracle.fptr(ref[v]), SR, oracle.fptr(inp[v]), float(fake[v]))
        rs[v] = (st.dval, st.dcount)
    m = mod.Decimator(V, ctx)
    st = m.state(); st["dval"] = dval0; st["dcount"] = dcount0; m.set_state(st)
    out = util.to_image(out0); gi = util.to_image(inp); gf = util.dev(fake)
    for (s, e) in [(0, 1024), (100, 700)]:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(SR, gi, gf))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, "decimator, unusual states")
    st = m.state()
    util.assert_bitexact(st["dval"].astype(np.float32), rs[:, 0].copy(), "dval")
    util.assert_bitexact(st["dcount"].astype(np.float32), rs[:, 1].copy(), "dcount")


# ------------------------------------------------------------------ Distortion
@pytest.mark.parametrize("dtype_", [0, 1])
def test_distortion(ctx, oracle, dtype_):
    from zang_amd import modules as mod, zang
    V = 192
    rng = np.random.default_rng(81)
    ingain = rng.uniform(0, 1, V).astype(np.float32); outgain = rng.uniform(0, 1, V).astype(np.float32)
    offset = rng.uniform(-1, 1, V).astype(np.float32)
    ingain[:4] = [0.25, 0.3125, 0.1875, 0.375]     # pow special cases: y = 0, 0.5, -0.5, 1
    inp = util.rng_buffers(82, V, F, -3, 3); out0 = util.rng_buffers(83, V, F)
    L = oracle.lib()
    ref = out0.copy()
    for v in range(V):
        L.zo_distortion_paint(5, 1000, oracle.fptr(ref[v]), oracle.fptr(inp[v]), dtype_, float(ingain[v]), float(outgain[v]), float(offset[v]))
    m = mod.Distortion(V, ctx)
    out = util.to_image(out0)
    m.paint(zang.Span(5, 1000), [out], [], False, m.Params(util.to_image(inp), dtype_, util.dev(ingain), util.dev(outgain), util.dev(offset)))
    ctx.sync()
    got = util.from_image(out)
    util.assert_close(got, ref, "distortion")
    util.assert_bitexact(got, ref, "distortion (same algorithm, expected exact)")


# ------------------------------------------------------------------ Cycle / Portamento
@pytest.mark.parametrize("kind", ["c", "b"])
def test_cycle(ctx, oracle, kind, replay_form):
    from zang_amd import modules as mod, zang
    V = 128
    rng = np.random.default_rng(91)
    speed = rng.uniform(-50, 4000, V).astype(np.float32)
    sbuf = rng.uniform(-50, 4000, (V, F)).astype(np.float32)
    out0 = util.rng_buffers(92, V, F)
    L = oracle.lib()
    ref = out0.copy(); rt = np.zeros(V, np.float32)
    for v in range(V):
        st = oracle.Cycle(); L.zo_cycle_init(C.byref(st))
        for (s, e) in util.SPANS_THREE:
            L.zo_cycle_paint(C.byref(st), s, e, oracle.fptr(ref[v]), SR, _cob(oracle, kind, speed[v], sbuf[v]))
        rt[v] = st.t
    m = mod.Cycle(V, ctx)
    out = util.to_image(out0); gb = util.to_image(sbuf); gs = util.dev(speed)
    for (s, e) in util.SPANS_THREE:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(SR, _gcob(zang, kind, gs, gb)))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, "cycle")
    util.assert_bitexact(m.state()["t"].astype(np.float32), rt, "cycle t")


@pytest.mark.parametrize("curve", [0, 1, 2, 3])
def test_portamento(ctx, oracle, curve, replay_form):
    from zang_amd import modules as mod, zang
    V = 128
    rng = np.random.default_rng(93)
    dur = rng.uniform(0.001, 0.03, V).astype(np.float32)
    script = []
    for (s, e) in [(0, 300), (300, 1024), (0, 1024), (0, 0), (0, 512)]:
        script.append(((s, e), rng.uniform(100, 2000, V).astype(np.float32), rng.random(V) < 0.7, rng.random(V) < 0.7, rng.random(V) < 0.4))
    out0 = util.rng_buffers(94, V, F)
    L = oracle.lib()
    ref = out0.copy(); rst = np.zeros((V, 3), np.float32)
    for v in range(V):
        st = oracle.Portamento(); L.zo_portamento_init(C.byref(st))
        for ((s, e), goal, on, prev, nic) in script:
            L.zo_portamento_paint(C.byref(st), s, e, oracle.fptr(ref[v]), int(nic[v]), SR, oracle.curve(curve, dur[v]), float(goal[v]), int(on[v]), int(prev[v]))
        rst[v] = (st.painter.t, st.painter.last_value, st.painter.start)
    m = mod.Portamento(V, ctx)
    out = util.to_image(out0)
    mk = [None, zang.PaintCurve.linear, zang.PaintCurve.squared, zang.PaintCurve.cubed]
    gcurve = zang.PaintCurve.instantaneous if curve == 0 else mk[curve](util.dev(dur))
    u8 = lambda a: util.dev(a.astype(np.uint8))
    for ((s, e), goal, on, prev, nic) in script:
        m.paint(zang.Span(s, e), [out], [], u8(nic), m.Params(SR, gcurve, util.dev(goal), u8(on), u8(prev)))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"portamento curve {curve}")
    st = m.state()
    for k, name in enumerate(("t", "last_value", "start")):
        util.assert_bitexact(st[name].astype(np.float32), rst[:, k].copy(), f"portamento {name}")


@pytest.mark.parametrize("function", [0, 1])
def test_curve(ctx, oracle, function, replay_form):
    """Curve.zig: shared node list, per-voice progress; voices are desynchronised by per-voice
    note_id_changed (restart) at different buffers; sub-span paints; runs past the last node."""
    from zang_amd import modules as mod, zang
    V = 96
    rng = np.random.default_rng(97)
    ts = np.cumsum(rng.uniform(0.0004, 0.02, 24)).astype(np.float32); ts[0] = 0.0
    vals = rng.uniform(-1, 1, 24).astype(np.float32)
    ts[5] = ts[4]                                    # two nodes on the same frame (:163-167)
    nodes = np.stack([vals, ts], axis=1).astype(np.float32)
    carr = (oracle.CurveNode * len(nodes))(*[oracle.CurveNode(float(v), float(t)) for v, t in nodes])
    script = []
    for b in range(6):
        for (s, e) in (util.SPANS_THREE if b % 2 else util.SPANS_ONE):
            script.append(((s, e), rng.random(V) < (1.0 if (b == 0 and s == 0) else 0.08)))
    script.append(((10, 10), rng.random(V) < 0.5))   # empty span still resets on note_id_changed
    script.append(((0, 1024), np.zeros(V, bool)))
    out0 = util.rng_buffers(98, V, F)
    L = oracle.lib()
    ref = out0.copy(); rst = []
    for v in range(V):
        st = oracle.CurveModule(); L.zo_curve_init(C.byref(st))
        for ((s, e), nic) in script:
            L.zo_curve_paint(C.byref(st), s, e, oracle.fptr(ref[v]), int(nic[v]), SR, function, carr, len(nodes))
        rst.append((st.t, st.current_song_note, st.current_song_note_offset, st.next_song_note))
    m = mod.Curve(V, ctx)
    out = util.to_image(out0); gnodes = util.dev(nodes)
    for ((s, e), nic) in script:
        m.paint(zang.Span(s, e), [out], [], util.dev(nic.astype(np.uint8)), m.Params(SR, function, gnodes))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, f"curve fn {function}")
    gs = m.state()
    util.assert_bitexact(gs["t"].astype(np.float32), np.array([r[0] for r in rst], np.float32), "curve t")
    assert [int(x) for x in gs["current_song_note"]] == [r[1] for r in rst]
    assert [int(x) for x in gs["current_song_note_offset"]] == [r[2] for r in rst]
    assert [int(x) for x in gs["next_song_note"]] == [r[3] for r in rst]
