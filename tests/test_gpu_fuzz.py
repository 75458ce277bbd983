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
def test_fuzz_pmosc(ctx, oracle, seed):
    """PMOscInstrument: random sub-spans, per-voice frequencies and release times (down to stages a few frames long, so that
    chunks with a stage end sit between quiet ones and, with a shared note pattern, chunks in which every voice is inside a
    stage), note on / off / retrigger; both launch forms (frame ranges for few voices, the lane-per-voice walk)."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(2600 + seed)
    V = int(rng.choice([1, 63, 64, 66, 130, 200]))
    rel = rng.choice([0.0004, 0.002, 0.05, 0.4], V).astype(np.float32) * rng.uniform(0.5, 1.5, V).astype(np.float32)
    L = oracle.lib()
    st = [oracle.PMOscInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_pmosc_init(C.byref(st[v]), float(rel[v]))
    m = mod.PMOscInstrument(V, util.dev(rel), ctx)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32); t2 = np.zeros(F, np.float32)
    img = util.rng_buffers(seed + 350, V, F)
    on = False
    for k, (a, b, zf) in enumerate(_calls_long(rng, 7)):
        freq = (_freqs(rng, V) * np.float32(0.5)).astype(np.float32)
        if rng.random() < 0.2:
            freq[int(rng.integers(V))] = np.float32(rng.choice([0.0, -220.0, 3.0e12, 1.0e5]))
        nic = bool(rng.random() < 0.4)
        on = (not on) if rng.random() < 0.5 else on
        if nic:
            on = True
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_pmosc_paint(C.byref(st[v]), a, b, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(t2), int(nic), SR, float(freq[v]), int(on))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], None, nic, m.Params(SR, util.dev(freq), on), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"pmosc seed {seed} call {k} V={V} span {(a, b)} zf={zf} on={on} nic={nic}")
        img = ref
    gs = m.state()
    util.assert_bitexact(gs["carrier"]["t"].astype(np.float32), np.array([r.carrier.t for r in st], np.float32), "carrier t")
    util.assert_bitexact(gs["modulator"]["t"].astype(np.float32), np.array([r.modulator.t for r in st], np.float32), "modulator t")


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


def _calls_long(rng, n):
    """Like _calls, biased towards spans of 128 frames and more (the frame-range forms need them) with short and empty ones mixed in."""
    out = []
    for _ in range(n):
        r = rng.random()
        if r < 0.35:
            a, b = 0, F
        elif r < 0.75:
            a = int(rng.integers(0, F - 128)); b = int(rng.integers(a + 128, F + 1))
        elif r < 0.9:
            a, b = sorted(int(x) for x in rng.integers(0, F + 1, 2))
        else:
            a = b = int(rng.integers(0, F + 1))
        out.append((a, b, bool(rng.random() < 0.5)))
    return out


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_noise(ctx, oracle, seed):
    """Noise module: white (frame ranges with jump-ahead for long spans, sequential for short ones) and pink, += and ZERO_FIRST."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(4000 + seed)
    V = int(rng.choice([1, 5, 64, 100, 257, 300]))
    first = int(rng.integers(0, 100000))
    L = oracle.lib()
    nzs = []
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + v); nzs.append(nz)
    m = mod.Noise(V, ctx, first_seed=first)
    img = util.rng_buffers(seed + 120, V, F)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 7)):
        color = int(rng.random() < 0.3)
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_noise_paint(C.byref(nzs[v]), a, b, oracle.fptr(ref[v]), color)
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], [], False, m.Params(color), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"noise seed {seed} call {k} V={V} span {(a, b)} zf={zf} color={color}")
        img = ref
    assert [[int(x) for x in row] for row in m.state()["r"]] == [list(n.r) for n in nzs]


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_sineosc(ctx, oracle, seed):
    """SineOsc: the four param paths, per-voice frequencies that change between calls, long and short spans (frame ranges /
    sequential), the phase wrapped once per paint."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(5000 + seed)
    V = int(rng.choice([1, 7, 64, 130, 260]))
    L = oracle.lib()
    sts = []
    for v in range(V):
        st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st)); sts.append(st)
    m = mod.SineOsc(V, ctx)
    img = util.rng_buffers(seed + 150, V, F)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 7)):
        fb, pb = bool(rng.random() < 0.4), bool(rng.random() < 0.4)
        freq = rng.uniform(-50.0, 6000.0, V).astype(np.float32)
        fbuf = rng.uniform(0.0, 4000.0, (V, F)).astype(np.float32); pbuf = rng.uniform(-2.0, 2.0, (V, F)).astype(np.float32)
        ph = float(rng.uniform(-1, 1))
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_sineosc_paint(C.byref(sts[v]), a, b, oracle.fptr(ref[v]), SR, oracle.buffer(fbuf[v]) if fb else oracle.constant(freq[v]),
                               oracle.buffer(pbuf[v]) if pb else oracle.constant(ph))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], [], False,
                m.Params(SR, zang.buffer(util.to_image(fbuf)) if fb else zang.constant(util.dev(freq)),
                         zang.buffer(util.to_image(pbuf)) if pb else zang.constant(ph)), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"sineosc seed {seed} call {k} V={V} span {(a, b)} zf={zf} fb={fb} pb={pb}")
        img = ref
    util.assert_bitexact(m.state()["t"].astype(np.float32), np.array([s.t for s in sts], np.float32), "t")


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_sampler(ctx, oracle, seed):
    """Sampler: formats, loop / no loop, per-voice output rates (ratio ~ 1, resampling both ways, negative), note restarts."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(6000 + seed)
    V = int(rng.choice([1, 9, 64, 96, 200]))
    fmt = int(rng.integers(0, 4)); loop = bool(rng.random() < 0.5); channels = int(rng.integers(1, 3)); in_rate = 44100
    nbytes = (fmt + 1) * channels * int(rng.integers(200, 1500))
    data = rng.integers(0, 256, nbytes, dtype=np.uint8)
    L = oracle.lib()
    sts = []
    for v in range(V):
        st = oracle.Sampler(); L.zo_sampler_init(C.byref(st)); sts.append(st)
    m = mod.Sampler(V, ctx)
    smp = m.Sample(channels, in_rate, fmt, util.dev(data))
    img = util.rng_buffers(seed + 170, V, F)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 6)):
        rate = rng.uniform(8000, 96000, V).astype(np.float32)
        rate[rng.random(V) < 0.2] = np.float32(44100.0)
        rate[rng.random(V) < 0.1] = np.float32(-30000.0)
        nic = rng.random(V) < 0.25
        ch = int(rng.integers(0, channels))
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            p = oracle.SamplerParams(float(rate[v]), channels, in_rate, fmt, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, ch, int(loop))
            L.zo_sampler_paint(C.byref(sts[v]), a, b, oracle.fptr(ref[v]), int(nic[v]), C.byref(p))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], [], util.dev(nic.astype(np.uint8)), m.Params(util.dev(rate), smp, ch, loop), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"sampler seed {seed} call {k} V={V} span {(a, b)} zf={zf} fmt={fmt} loop={loop}")
        img = ref
    util.assert_bitexact(m.state()["t"].astype(np.float32), np.array([s.t for s in sts], np.float32), "t")


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_envelope(ctx, oracle, seed):
    """Envelope: random curve tags (shared-tag kernels, the generic one, instantaneous), per-voice durations from a few frames
    to several buffers (stages end anywhere inside an 8-frame chunk, or not at all), random note scripts; ragged spans."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(7000 + seed)
    V = int(rng.choice([1, 64, 100, 257]))
    tags = [int(x) for x in (rng.integers(0, 4, 3) if rng.random() < 0.5 else [int(rng.integers(1, 4))] * 3)]
    dur = [np.exp(rng.uniform(np.log(0.0001), np.log(0.08), V)).astype(np.float32) for _ in range(3)]
    sus = rng.choice(np.array([0.0, 0.3, 0.8, 1.0], np.float32), V).astype(np.float32)
    L = oracle.lib()
    sts = []
    for v in range(V):
        st = oracle.Envelope(); L.zo_envelope_init(C.byref(st)); sts.append(st)
    m = mod.Envelope(V, ctx)
    mk = [None, zang.PaintCurve.linear, zang.PaintCurve.squared, zang.PaintCurve.cubed]
    gc = [zang.PaintCurve.instantaneous if tags[i] == 0 else mk[tags[i]](util.dev(dur[i])) for i in range(3)]
    img = util.rng_buffers(seed + 200, V, F)
    on_prev = np.zeros(V, bool)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 8)):
        on = np.where(rng.random(V) < 0.3, ~on_prev, on_prev)
        nic = on & (~on_prev | (rng.random(V) < 0.2))            # a note that starts gets a new id (the reference asserts it)
        on_prev = on
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            p = oracle.EnvelopeParams(SR, oracle.curve(tags[0], dur[0][v]), oracle.curve(tags[1], dur[1][v]),
                                      oracle.curve(tags[2], dur[2][v]), float(sus[v]), int(on[v]))
            L.zo_envelope_paint(C.byref(sts[v]), a, b, oracle.fptr(ref[v]), int(nic[v]), C.byref(p))
        out = util.to_image(img)
        m.paint(zang.Span(a, b), [out], [], util.dev(nic.astype(np.uint8)),
                m.Params(SR, gc[0], gc[1], gc[2], util.dev(sus), util.dev(on.astype(np.uint8))), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"envelope seed {seed} call {k} V={V} span {(a, b)} zf={zf} tags={tags}")
        img = ref
    st = m.state()
    assert [int(x) for x in st["state"]] == [s.state for s in sts]
    util.assert_bitexact(st["t"].astype(np.float32), np.array([s.painter.t for s in sts], np.float32), "envelope t")
    util.assert_bitexact(st["last_value"].astype(np.float32), np.array([s.painter.last_value for s in sts], np.float32), "envelope last_value")


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_decimator_portamento(ctx, oracle, seed):
    """Decimator (all three modes in one wave, rates from 1/200 to above the sample rate; frame ranges for long spans) and
    Portamento (random curves / durations / goals, glides that arrive mid-chunk, waves that are flat throughout)."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(8000 + seed)
    V = int(rng.choice([1, 64, 96, 200]))
    L = oracle.lib()
    fake = np.exp(rng.uniform(np.log(240.0), np.log(60000.0), V)).astype(np.float32)
    fake[rng.random(V) < 0.1] = np.float32(0.0)
    fake[rng.random(V) < 0.1] = np.float32(48000.0)
    inp = util.rng_buffers(seed + 230, V, F)
    dsts, psts = [], []
    for v in range(V):
        d = oracle.Decimator(); L.zo_decimator_init(C.byref(d)); dsts.append(d)
        q = oracle.Portamento(); L.zo_portamento_init(C.byref(q)); psts.append(q)
    md, mp = mod.Decimator(V, ctx), mod.Portamento(V, ctx)
    gi = util.to_image(inp)
    tag = int(rng.integers(0, 4))
    dur = np.exp(rng.uniform(np.log(0.0002), np.log(0.05), V)).astype(np.float32)
    mk = [None, zang.PaintCurve.linear, zang.PaintCurve.squared, zang.PaintCurve.cubed]
    gcurve = zang.PaintCurve.instantaneous if tag == 0 else mk[tag](util.dev(dur))
    img_d = util.rng_buffers(seed + 231, V, F); img_p = util.rng_buffers(seed + 232, V, F)
    u8 = lambda x: util.dev(x.astype(np.uint8))
    for k, (a, b, zf) in enumerate(_calls_long(rng, 7)):
        goal = rng.uniform(50, 3000, V).astype(np.float32)
        on, prev, nic = rng.random(V) < 0.7, rng.random(V) < 0.7, rng.random(V) < 0.3
        ref_d, ref_p = img_d.copy(), img_p.copy()
        if zf:
            ref_d[:, a:b] = 0.0; ref_p[:, a:b] = 0.0
        for v in range(V):
            L.zo_decimator_paint(C.byref(dsts[v]), a, b, oracle.fptr(ref_d[v]), SR, oracle.fptr(inp[v]), float(fake[v]))
            L.zo_portamento_paint(C.byref(psts[v]), a, b, oracle.fptr(ref_p[v]), int(nic[v]), SR, oracle.curve(tag, dur[v]), float(goal[v]), int(on[v]), int(prev[v]))
        od, op = util.to_image(img_d), util.to_image(img_p)
        md.paint(zang.Span(a, b), [od], [], False, md.Params(SR, gi, util.dev(fake)), zero_first=zf)
        mp.paint(zang.Span(a, b), [op], [], u8(nic), mp.Params(SR, gcurve, util.dev(goal), u8(on), u8(prev)), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(od), ref_d, f"decimator seed {seed} call {k} V={V} span {(a, b)} zf={zf}")
        util.assert_bitexact(util.from_image(op), ref_p, f"portamento seed {seed} call {k} V={V} span {(a, b)} zf={zf} tag={tag}")
        img_d, img_p = ref_d, ref_p
    st = md.state()
    util.assert_bitexact(st["dval"].astype(np.float32), np.array([d.dval for d in dsts], np.float32), "dval")
    util.assert_bitexact(st["dcount"].astype(np.float32), np.array([d.dcount for d in dsts], np.float32), "dcount")
    st = mp.state()
    util.assert_bitexact(st["t"].astype(np.float32), np.array([q.painter.t for q in psts], np.float32), "portamento t")


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_osc_control_images(ctx, oracle, seed):
    """PulseOsc / TriSawOsc with a frequency image (out-of-range samples included): frame ranges for long spans, the walk for
    short ones, += and ZERO_FIRST, state carried from call to call."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(9000 + seed)
    V = int(rng.choice([1, 64, 130, 192]))
    L = oracle.lib()
    color = rng.uniform(0.0, 1.0, V).astype(np.float32)
    ps, ts = [], []
    for v in range(V):
        a_ = oracle.PulseOsc(); L.zo_pulseosc_init(C.byref(a_)); ps.append(a_)
        b_ = oracle.TriSawOsc(); L.zo_trisawosc_init(C.byref(b_)); ts.append(b_)
    mp_, mt_ = mod.PulseOsc(V, ctx), mod.TriSawOsc(V, ctx)
    img_p = util.rng_buffers(seed + 260, V, F); img_t = util.rng_buffers(seed + 261, V, F)
    gc = util.dev(color)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 6)):
        fbuf = rng.uniform(-300.0, 7000.0, (V, F)).astype(np.float32)
        ref_p, ref_t = img_p.copy(), img_t.copy()
        if zf:
            ref_p[:, a:b] = 0.0; ref_t[:, a:b] = 0.0
        for v in range(V):
            L.zo_pulseosc_paint(C.byref(ps[v]), a, b, oracle.fptr(ref_p[v]), SR, oracle.buffer(fbuf[v]), float(color[v]))
            L.zo_trisawosc_paint(C.byref(ts[v]), a, b, oracle.fptr(ref_t[v]), SR, oracle.buffer(fbuf[v]), float(color[v]))
        gf = util.to_image(fbuf)
        op, ot = util.to_image(img_p), util.to_image(img_t)
        mp_.paint(zang.Span(a, b), [op], [], False, mp_.Params(SR, zang.buffer(gf), gc), zero_first=zf)
        mt_.paint(zang.Span(a, b), [ot], [], False, mt_.Params(SR, zang.buffer(gf), gc), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(op), ref_p, f"pulseosc image seed {seed} call {k} V={V} span {(a, b)} zf={zf}")
        util.assert_bitexact(util.from_image(ot), ref_t, f"trisawosc image seed {seed} call {k} V={V} span {(a, b)} zf={zf}")
        img_p, img_t = ref_p, ref_t
    assert [int(x) for x in mp_.state()["cnt"]] == [x.cnt for x in ps]
    util.assert_bitexact(mt_.state()["t"].astype(np.float32), np.array([x.t for x in ts], np.float32), "trisaw t")


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_filter_and_echoes(ctx, oracle, seed):
    """Filter (every type; constant parameters = the three-wave pipeline, a cutoff and / or a resonance image = the same pipeline with
    the images' rows as tiles, k_filter_pc_ctl; spans under 64 frames = the one-wave walk; inputs with huge and tiny samples) and
    FilteredEchoes (delays either side of the pipeline's 192-frame minimum) over random spans."""
    from zang_amd import modules as mod, zang
    rng = np.random.default_rng(9000 + seed)
    V = int(rng.choice([1, 3, 64, 70, 150]))
    L = oracle.lib()
    fls = []
    for v in range(V):
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
    m = mod.Filter(V, ctx)
    img = util.rng_buffers(seed + 190, V, F)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 6)):
        ftype = int(rng.integers(0, 6))
        inp = rng.uniform(-1, 1, (V, F)).astype(np.float32)
        inp[rng.random((V, F)) < 0.01] *= np.float32(1e30)
        inp[rng.random((V, F)) < 0.01] *= np.float32(1e-30)
        cut_image = rng.random() < 0.3
        cutoff = rng.uniform(-0.1, 1.1, (V, F) if cut_image else V).astype(np.float32)
        res_image = seed >= 4 and rng.random() < 0.3                  # (seeds 0-3 keep the draws they had before round 4)
        res = rng.uniform(-0.1, 1.1, (V, F) if res_image else V).astype(np.float32)
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        with np.errstate(all="ignore"):
            for v in range(V):
                L.zo_filter_paint(C.byref(fls[v]), a, b, oracle.fptr(ref[v]), oracle.fptr(inp[v]), ftype,
                                  oracle.buffer(cutoff[v]) if cut_image else oracle.constant(cutoff[v]),
                                  oracle.buffer(res[v]) if res_image else oracle.constant(res[v]))
        out = util.to_image(img)
        gc = zang.buffer(util.to_image(cutoff)) if cut_image else zang.constant(util.dev(cutoff))
        gr = zang.buffer(util.to_image(res)) if res_image else zang.constant(util.dev(res))
        m.paint(zang.Span(a, b), [out], [], False, m.Params(util.to_image(inp), ftype, gc, gr), zero_first=zf)
        ctx.sync()
        got = util.from_image(out)
        nan = np.isnan(ref)
        assert np.array_equal(np.isnan(got), nan), f"filter seed {seed} call {k}: NaN positions"
        util.assert_bitexact(np.where(nan, np.float32(0), got), np.where(nan, np.float32(0), ref), f"filter seed {seed} call {k} V={V} span {(a, b)} zf={zf} type={ftype} images={cut_image, res_image}")
        img = np.where(nan, np.float32(0), ref)                       # (the next call adds onto finite values again)
        fl_l = np.array([f.l for f in fls], np.float32)
        if not np.all(np.isfinite(fl_l)) or not np.all(np.isfinite(np.array([f.b for f in fls], np.float32))):
            for v in range(V):                                        # a voice that blew up starts over, on both sides
                L.zo_filter_init(C.byref(fls[v]))
            m.close(); m = mod.Filter(V, ctx)
    D = int(rng.choice([5, 191, 192, 193, 300, 777]))
    fb = rng.uniform(0.1, 0.9, V).astype(np.float32); cutoff = rng.uniform(0.05, 1.0, V).astype(np.float32)
    rings = np.zeros((V, D), np.float32)
    ds, fl2 = [], []
    for v in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[v]), D); ds.append(d)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fl2.append(fl)
    e = mod.FilteredEchoes(V, D, ctx)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    img = util.rng_buffers(seed + 290, V, F)
    for k, (a, b, zf) in enumerate(_calls_long(rng, 6)):
        inp = rng.uniform(-1, 1, (V, F)).astype(np.float32)
        ref = img.copy()
        if zf:
            ref[:, a:b] = 0.0
        for v in range(V):
            L.zo_filtered_echoes_paint(C.byref(ds[v]), C.byref(fl2[v]), a, b, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1),
                                       oracle.fptr(inp[v]), float(fb[v]), float(cutoff[v]))
        out = util.to_image(img)
        e.paint(zang.Span(a, b), [out], None, False, e.Params(util.to_image(inp), util.dev(fb), util.dev(cutoff)), zero_first=zf)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"echoes seed {seed} call {k} V={V} D={D} span {(a, b)} zf={zf}")
        img = ref
    grings, gidx, gflt = e.state()
    util.assert_bitexact(grings, rings, "ring")
    assert [int(x) for x in gidx] == [d.index for d in ds]


def test_fuzz_again_with_the_single_wave_forms():
    """The same random cases through the lane-per-voice sequential forms (k_nice, k_noise_filter, k_noise, k_sineosc, the
    one-range k_sampler / k_decimator / oscillator control kernels: what runs above the voice-count limits of the pipelined /
    frame-range forms)."""
    import os
    import subprocess
    import sys
    if os.environ.get("ZH_FUZZ_CHILD"):
        pytest.skip("already the rerun")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = util.forms_env(nice_pc_max=0, nf_pc_max=0, nf_ring_max=0, noise_ranges=0, sine_ranges=0, sampler_ranges=0, pink_taps=0, decimator_ranges=0,
                         envelope_ranges=0, portamento_ranges=0, pulse_ctrl_ranges=0, trisaw_ctrl_ranges=0, pink_pipe_max=0, filter_pc_max=0, echoes_pc_max=0)
    env["ZH_FUZZ_CHILD"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fuzz.py", "-q", "-m", "gpu"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    util.assert_rerun_green(r, 40)
