"""CPU: an INDEPENDENT numpy-float32 re-derivation of the stateful modules, written from the
cited reference lines without looking at oracle/zang_oracle.c's structure, compared with the
C oracle.  Two restatements that agree bit for bit are the strongest pin available while
the reference itself cannot be built.  Pure-Python loops: small cases only.
"""
import ctypes as C

import numpy as np

from tests import util

f32 = np.float32
u32 = lambda x: int(x) & 0xFFFFFFFF
SR = f32(48000.0)


def utof23(x):                       # PulseOsc.zig:19-21
    return np.array([(u32(x) >> 9) | 0x3F800000], np.uint32).view(np.float32)[0] - f32(1)


def ftou32(v):                       # PulseOsc.zig:24-26
    return int(f32(f32(f32(v) * f32(4294967296.0)) * f32(0.99995)))


def clamp01(v):
    return f32(0) if v < 0 else (f32(1) if v > 1 else f32(v))


def np_pulse(cnt, n, freq, color):   # PulseOsc.zig:75-114
    out = np.zeros(n, np.float32)
    if freq < 0 or freq > SR / f32(8):
        return out, cnt
    ifreq = int(f32(f32(4294967296.0) / SR) * f32(freq))
    brpt = ftou32(clamp01(color))
    gain = f32(0.7)
    with np.errstate(divide="ignore"):
        gdf = gain / utof23(ifreq)
    col = utof23(brpt)
    cc121 = gdf * f32(2) * (col - f32(1)) + gain
    cc212 = gdf * f32(2) * col - gain
    state = 3 if u32(cnt - ifreq) < brpt else 0
    for i in range(n):
        p = utof23(cnt)
        state = ((state << 1) | (1 if cnt < brpt else 0)) & 3
        tr = state | ((1 if cnt < ifreq else 0) << 2)
        out[i] = {3: gain, 0: -gain, 2: gdf * f32(2) * (col - p) + gain, 5: gdf * f32(2) * p - gain,
                  7: cc121, 4: cc212}.get(tr, f32(0))
        cnt = u32(cnt + ifreq)
    return out, cnt


def np_trisaw(cnt, n, freq, color):  # TriSawOsc.zig:77-118
    out = np.zeros(n, np.float32)
    if freq < 0 or freq > SR / f32(8):
        return out, cnt
    ifreq = int(f32(f32(4294967296.0) / SR) * f32(freq))
    brpt = ftou32(clamp01(color))
    gain = f32(0.7)
    f = utof23(ifreq); omf = f32(1) - f; rcpf = f32(1) / f
    col = utof23(brpt)
    with np.errstate(divide="ignore"):
        c1 = gain / col
        c2 = -gain / (f32(1) - col)
    state = 3 if u32(cnt - ifreq) < brpt else 0
    sq = lambda v: v * v
    for i in range(n):
        p = utof23(cnt) - col
        state = ((state << 1) | (1 if cnt < brpt else 0)) & 3
        s = state | ((1 if cnt < ifreq else 0) << 2)
        if s == 3: v = c1 * (p + p - f)
        elif s == 0: v = c2 * (p + p - f)
        elif s == 2: v = rcpf * (c2 * sq(p) - c1 * sq(p - f))
        elif s == 5: v = -rcpf * (gain + c2 * sq(p + omf) - c1 * sq(p))
        elif s == 7: v = -rcpf * (gain + c1 * omf * (p + p + omf))
        elif s == 4: v = -rcpf * (gain + c2 * omf * (p + p + omf))
        else: v = f32(0)
        out[i] = gain + v
        cnt = u32(cnt + ifreq)
    return out, cnt


def test_pulse_and_trisaw_const(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(7)
    for trial in range(12):
        freq = f32(rng.uniform(20, 6000)); color = f32(rng.uniform(0, 1)); cnt0 = int(rng.integers(0, 2 ** 32)); n = 300
        if trial == 0: color = f32(0.0)
        if trial == 1: color = f32(1.0)
        if trial == 2: freq = f32(6000.5)
        ref, cnt = np_pulse(cnt0, n, freq, color)
        st = oracle.PulseOsc(cnt0); out = np.zeros(n, np.float32)
        L.zo_pulseosc_paint(C.byref(st), 0, n, oracle.fptr(out), SR, oracle.constant(freq), float(color))
        util.assert_bitexact(out, ref, f"pulse trial {trial}"); assert st.cnt == cnt
        ref, cnt = np_trisaw(cnt0, n, freq, color)
        st = oracle.TriSawOsc(cnt0, 0.0); out = np.zeros(n, np.float32)
        L.zo_trisawosc_paint(C.byref(st), 0, n, oracle.fptr(out), SR, oracle.constant(freq), float(color))
        util.assert_bitexact(out, ref, f"trisaw trial {trial}"); assert st.cnt == cnt


def test_filter_all_types(oracle):   # Filter.zig:98-146
    L = oracle.lib()
    rng = np.random.default_rng(8)
    n = 400
    inp = rng.uniform(-1, 1, n).astype(np.float32)
    dc = f32(3.814697265625e-6)
    muls = {1: (1, 0, 0), 2: (0, 1, 0), 3: (0, 0, 1), 4: (1, 0, 1), 5: (1, 1, 1)}
    for ftype, (lm, bm, hm) in muls.items():
        cutoff = f32(rng.uniform(0, 1)); resp = f32(rng.uniform(0, 1))
        cut = min(max(cutoff, f32(0)), f32(1)); res = f32(1) - min(max(resp, f32(0)), f32(1))
        l = f32(0); b = f32(0); ref = np.zeros(n, np.float32)
        for i in range(n):
            x = inp[i] + dc
            l = l + (cut * b - dc)
            b = b + cut * (x - b * res - l)
            l = l + cut * b
            h = x - b * res - l
            b = b + cut * h
            ref[i] = l * f32(lm) + b * f32(bm) + h * f32(hm)
        st = oracle.Filter(); L.zo_filter_init(C.byref(st)); out = np.zeros(n, np.float32)
        L.zo_filter_paint(C.byref(st), 0, n, oracle.fptr(out), oracle.fptr(inp), ftype, oracle.constant(cutoff), oracle.constant(resp))
        util.assert_bitexact(out, ref, f"filter type {ftype}")
        assert f32(st.l) == l and f32(st.b) == b


class NpEnvelope:                    # Envelope.zig + painter.zig, stage-by-stage like the reference
    IDLE, ATTACK, DECAY, SUSTAIN, RELEASE = range(5)

    def __init__(self):
        self.state = 0; self.t = f32(0); self.last = f32(0); self.start = f32(0)

    def change(self, s):
        self.state = s; self.start = self.last; self.t = f32(0)

    def toward(self, buf, i, curve, goal):
        tag, dur = curve
        if self.t >= 1: return True, i
        if tag == 0:
            self.t = f32(1); self.last = f32(goal); return True, i
        with np.errstate(divide="ignore"):
            step = f32(1) / (f32(dur) * SR)
        fin = False
        while not fin and i < len(buf):
            self.t = self.t + step
            if self.t >= 1: self.t = f32(1); fin = True
            it = f32(1) - self.t
            tp = self.t if tag == 1 else (f32(1) - it * it if tag == 2 else f32(1) - it * it * it)
            self.last = self.start + tp * (f32(goal) - self.start)
            buf[i] += self.last; i += 1
        return fin, i

    def paint(self, buf, new_note, attack, decay, release, sustain, note_on):
        i = 0
        if note_on:
            if new_note: self.change(self.ATTACK)
            if self.state == self.IDLE: self.change(self.ATTACK)
            if self.state == self.ATTACK:
                fin, i = self.toward(buf, i, attack, 1.0)
                if fin: self.change(self.DECAY if sustain < 1 else self.SUSTAIN)
            if self.state == self.DECAY:
                fin, i = self.toward(buf, i, decay, sustain)
                if fin: self.change(self.SUSTAIN)
            if self.state == self.SUSTAIN:
                buf[i:] += f32(sustain)
        else:
            if self.state == self.IDLE: return
            if self.state != self.RELEASE: self.change(self.RELEASE)
            fin, i = self.toward(buf, i, release, 0.0)
            if fin: self.change(self.IDLE)


def test_envelope_scripts(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(9)
    for trial in range(10):
        curves = [(int(rng.integers(0, 4)), f32(rng.uniform(0.0005, 0.01))) for _ in range(3)]
        sustain = f32(1.0) if trial % 3 == 0 else f32(rng.uniform(0.2, 0.9))
        script = [(200, 1, 1), (577, 1, 0), (247, 0, 0), (1024, 0, 0), (300, 1, 1), (0, 1, 1), (724, 0, 0)]
        ne = NpEnvelope(); st = oracle.Envelope(); L.zo_envelope_init(C.byref(st))
        for (n, on, nic) in script:
            ref = np.zeros(n, np.float32); out = np.zeros(max(n, 1), np.float32)
            ne.paint(ref, nic, curves[0], curves[1], curves[2], sustain, on)
            p = oracle.EnvelopeParams(SR, oracle.curve(*curves[0]), oracle.curve(*curves[1]), oracle.curve(*curves[2]), float(sustain), on)
            L.zo_envelope_paint(C.byref(st), 0, n, oracle.fptr(out), nic, C.byref(p))
            util.assert_bitexact(out[:n], ref, f"envelope trial {trial}")
            assert (st.state, f32(st.painter.t), f32(st.painter.last_value), f32(st.painter.start)) == (ne.state, ne.t, ne.last, ne.start)


def test_decimator_and_pink(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(10)
    n = 500
    inp = rng.uniform(-1, 1, n).astype(np.float32)
    for fake in (f32(11025.0), f32(47999.0), f32(123.0)):
        ratio = fake / SR; dcount = f32(1); dval = f32(0); ref = np.zeros(n, np.float32)
        for i in range(n):                                     # Decimator.zig:44-52
            dcount = dcount + ratio
            if dcount >= 1: dval = inp[i]; dcount = dcount - f32(1)
            ref[i] = dval
        st = oracle.Decimator(); L.zo_decimator_init(C.byref(st)); out = np.zeros(n, np.float32)
        L.zo_decimator_paint(C.byref(st), 0, n, oracle.fptr(out), SR, oracle.fptr(inp), fake)
        util.assert_bitexact(out, ref, "decimator"); assert (f32(st.dval), f32(st.dcount)) == (dval, dcount)
    # pink = Kellett filter over the same white stream (Noise.zig:58-66); white taken from the oracle
    w = oracle.Noise(); L.zo_noise_init(C.byref(w), 42); white = np.zeros(n, np.float32)
    L.zo_noise_paint(C.byref(w), 0, n, oracle.fptr(white), oracle.NOISE_WHITE)
    b = [f32(0)] * 7; ref = np.zeros(n, np.float32)
    k = [(0.99886, 0.0555179), (0.99332, 0.0750759), (0.96900, 0.1538520), (0.86650, 0.3104856), (0.55000, 0.5329522)]
    for i in range(n):
        wh = white[i]
        for j, (a, c) in enumerate(k): b[j] = f32(a) * b[j] + wh * f32(c)
        b[5] = f32(-0.7616) * b[5] - wh * f32(0.0168980)
        ref[i] = b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + wh * f32(0.5362)
        b[6] = wh * f32(0.115926)
    p = oracle.Noise(); L.zo_noise_init(C.byref(p), 42); out = np.zeros(n, np.float32)
    L.zo_noise_paint(C.byref(p), 0, n, oracle.fptr(out), oracle.NOISE_PINK)
    util.assert_bitexact(out, ref, "pink")
    assert list(p.r) == list(w.r) and not any(p.b)             # taps not written back (Noise.zig:68)


def test_sineosc_phase_accumulation(oracle):
    """t accumulates add by add and wraps once per paint (SineOsc.zig:40,51); sin is compared
    with float64 numpy at 1 ulp."""
    L = oracle.lib()
    n, freq, phase = 700, f32(1234.5), f32(0.25)
    step = freq / SR; t = f32(0.37); args = np.zeros(n, np.float32)
    for i in range(n):
        args[i] = (t + phase) * f32(np.pi) * f32(2.0)
        t = t + step
    st = oracle.SineOsc(0.37); out = np.zeros(n, np.float32)
    L.zo_sineosc_paint(C.byref(st), 0, n, oracle.fptr(out), SR, oracle.constant(freq), oracle.constant(phase))
    ref = np.sin(args.astype(np.float64)).astype(np.float32)
    assert np.abs(out.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max() <= 1
    assert f32(st.t) == t - np.trunc(t)


def test_sampler_linear_interp(oracle):
    """Sampler.zig:116-130: tfrac = (t0+1) - t, s = s0*(1-tfrac) + s1*tfrac (reference quirk kept)."""
    L = oracle.lib()
    rng = np.random.default_rng(12)
    pcm = rng.integers(-32768, 32767, 400).astype("<i2")
    data = pcm.view(np.uint8).copy()
    n, out_rate = 300, f32(30000.0)
    ratio = f32(44100.0) / out_rate; t = f32(0); ref = np.zeros(n, np.float32)
    get = lambda i: f32(pcm[i]) / f32(32768.0) if 0 <= i < len(pcm) else f32(0)
    for i in range(n):
        t0 = int(np.floor(t)); tfrac = f32(t0 + 1) - t
        ref[i] = get(t0) * (f32(1) - tfrac) + get(t0 + 1) * tfrac
        t = t + ratio
    st = oracle.Sampler(); L.zo_sampler_init(C.byref(st)); out = np.zeros(n, np.float32)
    p = oracle.SamplerParams(float(out_rate), 1, 44100, oracle.SAMPLE_S16, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, 0, 0)
    L.zo_sampler_paint(C.byref(st), 0, n, oracle.fptr(out), 0, C.byref(p))
    util.assert_bitexact(out, ref, "sampler interp"); assert f32(st.t) == t


def test_curve_linear_matches_closed_form(oracle):
    """Curve.zig linear interpolation between nodes, one whole-buffer paint: out[i] follows
    start + (i - f0)/(f1 - f0) * delta (f64 closed form, 1e-5), exact node values at node frames."""
    L = oracle.lib()
    nodes = [(0.0, 0.0), (0.5, 0.004), (-0.25, 0.0125), (1.0, 0.02)]
    arr = (oracle.CurveNode * 4)(*[oracle.CurveNode(v, t) for v, t in nodes])
    st = oracle.CurveModule(); L.zo_curve_init(C.byref(st))
    n = 1024; out = np.zeros(n, np.float32)
    L.zo_curve_paint(C.byref(st), 0, n, oracle.fptr(out), 1, SR, 0, arr, 4)
    frames = [int(np.float32((np.float32(t) - np.float32(0)) / (np.float32(n) / SR)) * np.float32(n)) for _, t in nodes]
    assert frames == [0, 192, 600, 960]
    for (f0, f1), ((v0, _), (v1, _)) in zip(zip(frames, frames[1:]), zip(nodes, nodes[1:])):
        i = np.arange(f0, f1)
        want = v0 + (i - f0) / (f1 - f0) * (v1 - v0)
        assert np.abs(out[f0:f1] - want).max() < 1e-5
        assert out[f0] == np.float32(v0)
    assert not out[960:].any()                      # after the last node: silent gap (:238-244)
    assert (st.current_song_note, st.next_song_note, st.current_song_note_offset) == (3, 4, -1024)      # :175,:181: offset is reset to 0 on advance, then -= out_len


def test_lowpass_mix_into_a_zeroed_temp_in_three_operations():
    """dsp.hip.h svf_lowpass_into_zero: `0 + (l*1 + b*0 + h*0)` (Filter.zig:146 with the low-pass multipliers, added into a
    zeroed temp as NiceInstrument and FilteredEchoes do) equals `0 + (l + b*0)` bit for bit for every (l, b, h) that a
    filter step (Filter.zig:135-144) can leave behind -- vectorised numpy float32 over random and extreme states, inputs
    and coefficients: zeros of both signs, denormals, huge values that overflow inside the step, infinities, NaNs."""
    rng = np.random.default_rng(5)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, 1.0, -1.0, 3e38, -3e38, 3.4028235e38, -3.4028235e38,
                        np.inf, -np.inf, np.nan, 1e19, -1e19, 1e30, -1e30], np.float32)
    n = 400000

    def draw():
        x = rng.uniform(-2, 2, n).astype(np.float32)
        x = np.where(rng.random(n) < 0.15, x * f32(1e37), x)
        pick = rng.random(n) < 0.35
        return np.where(pick, special[rng.integers(0, len(special), n)], x).astype(np.float32)

    l, b, x = draw(), draw(), draw()
    cut = np.where(rng.random(n) < 0.2, rng.choice(np.array([0.0, 1.0], np.float32), n), rng.uniform(0, 1, n)).astype(np.float32)
    res = np.where(rng.random(n) < 0.2, rng.choice(np.array([0.0, 1.0], np.float32), n), rng.uniform(0, 1, n)).astype(np.float32)
    dc = f32(3.814697265625e-6)
    zero = f32(0.0)
    with np.errstate(all="ignore"):
        for _ in range(4):                                              # a few steps: states the recurrence itself produces
            inp = x + dc
            l = l + cut * b - dc
            b = b + cut * (inp - b * res - l)
            l = l + cut * b
            h = inp - b * res - l
            b = b + cut * h
            six = zero + (l * f32(1.0) + b * f32(0.0) + h * f32(0.0))
            three = zero + (l + b * f32(0.0))
            nan6, nan3 = np.isnan(six), np.isnan(three)
            assert np.array_equal(nan6, nan3)
            assert np.array_equal(six[~nan6].view(np.uint32), three[~nan3].view(np.uint32))
            assert nan6.any() and (six == 0).any() and np.isinf(l[~np.isnan(l)]).any()   # the corners are really visited
            x = draw()


# ---------------------------------------------------------------------------------------------- round 3: the remaining modules
def np_painter_toward(p, buf, i0, curve, goal):
    """painter.zig:63-120 for one call: p = [t, last_value, start]; returns (i, finished)."""
    if p[0] >= f32(1.0):
        return i0, True
    tag, dur = curve
    if tag == 0:                                               # .instantaneous
        p[0] = f32(1.0); p[1] = f32(goal)
        return i0, True
    t_step = f32(1.0) / (f32(dur) * SR)
    finished, i = False, i0
    while not finished and i < len(buf):
        p[0] = p[0] + t_step
        if p[0] >= f32(1.0):
            p[0] = f32(1.0); finished = True
        it = f32(1.0) - p[0]
        tp = p[0] if tag == 1 else (f32(1.0) - it * it if tag == 2 else f32(1.0) - it * it * it)
        p[1] = p[2] + tp * (f32(goal) - p[2])
        buf[i] = buf[i] + p[1]
        i += 1
    return i, finished


def test_portamento_over_the_painter(oracle):
    """Portamento.zig:21-48: the curve only while the note stays on, newCurve on a new note, paintFlat once the goal is
    reached -- written from those lines over a re-derived Painter; sub-span paints, all four curve tags."""
    L = oracle.lib()
    rng = np.random.default_rng(21)
    for tag in (0, 1, 2, 3):
        dur = f32(0.004) if tag else f32(0.0)
        p = [f32(0), f32(0), f32(0)]
        st = oracle.Portamento(); L.zo_portamento_init(C.byref(st))
        ref = rng.uniform(-1, 1, 600).astype(np.float32); out = ref.copy()
        script = [((0, 200), 440.0, True, False, True), ((200, 600), 440.0, True, True, False), ((0, 600), 660.0, True, True, True),
                  ((0, 0), 100.0, True, True, False), ((50, 400), 220.0, False, True, False), ((0, 600), 330.0, True, False, True)]
        for (s, e), goal, on, prev, nic in script:
            curve = (tag, dur) if (on and prev) else (0, f32(0))
            if on and nic:
                p[2] = p[1]; p[0] = f32(0.0)                    # newCurve
            view = ref[s:e]
            i, fin = np_painter_toward(p, view, 0, curve, goal)
            if fin:
                view[i:] = view[i:] + f32(goal)                # paintFlat -> addScalarInto
            L.zo_portamento_paint(C.byref(st), s, e, oracle.fptr(out), int(nic), SR, oracle.curve(tag, dur), goal, int(on), int(prev))
            util.assert_bitexact(out, ref, f"portamento tag {tag} span {(s, e)}")
            assert (f32(st.painter.t), f32(st.painter.last_value), f32(st.painter.start)) == (p[0], p[1], p[2])


def test_cycle_gate_and_clip(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(22)
    n = 700
    # Cycle.zig:36-58: out += t; t += step (or speed[i] * isr); t -= trunc(t)
    speeds = rng.uniform(-30, 5000, n).astype(np.float32)
    for buf in (False, True):
        t = f32(0.3); ref = np.zeros(n, np.float32)
        step = f32(777.0) / SR; isr = f32(1.0) / SR
        for i in range(n):
            ref[i] = ref[i] + t
            t = t + (speeds[i] * isr if buf else step)
            t = t - np.trunc(t)
        st = oracle.Cycle(0.3); out = np.zeros(n, np.float32)
        L.zo_cycle_paint(C.byref(st), 0, n, oracle.fptr(out), SR, oracle.buffer(speeds) if buf else oracle.constant(777.0))
        util.assert_bitexact(out, ref, "cycle"); assert f32(st.t) == t
    # Gate.zig:27-29
    base = rng.uniform(-1, 1, n).astype(np.float32)
    for on in (0, 1):
        out = base.copy(); L.zo_gate_paint(100, 600, oracle.fptr(out), on)
        ref = base.copy()
        if on: ref[100:600] = ref[100:600] + f32(1.0)
        util.assert_bitexact(out, ref, "gate")
    # Distortion.zig:41, 54-63 (clip): gain1 = pow(2, ingain * 8 - 2) from the oracle's pow (its own tests pin it); the loop here
    inp = rng.uniform(-2, 2, n).astype(np.float32)
    for ingain, outgain, offset in ((0.25, 0.8, 0.0), (0.6, 0.5, -0.3), (0.05, 1.0, 0.9)):
        gain1 = f32(L.zo_math_powf(2.0, float(f32(ingain) * f32(8.0) - f32(2.0))))
        offs = gain1 * f32(offset)
        a = inp * gain1 + offs
        b = np.where(a < f32(-1.0), f32(-1.0), np.where(a > f32(1.0), f32(1.0), a)).astype(np.float32)
        ref = (base + f32(outgain) * b).astype(np.float32)
        out = base.copy()
        L.zo_distortion_paint(0, n, oracle.fptr(out), oracle.fptr(inp), oracle.DISTORTION_CLIP, ingain, outgain, offset)
        util.assert_bitexact(out, ref, "distortion clip")
        # overdrive (:44-52): atan in f64 numpy, 2 ulp
        gain2 = f32(outgain) / f32(np.arctan(np.float64(gain1)))
        refo = base.astype(np.float64) + np.float64(gain2) * np.arctan((inp * gain1 + offs).astype(np.float64))
        out = base.copy()
        L.zo_distortion_paint(0, n, oracle.fptr(out), oracle.fptr(inp), oracle.DISTORTION_OVERDRIVE, ingain, outgain, offset)
        assert np.abs(out - refo).max() <= 4e-7 * max(1.0, np.abs(refo).max())


def test_curve_smoothstep_and_linear_bit_for_bit(oracle):
    """Curve.zig:90-121 for one whole-buffer paint from a new note (every curve span starts at its node's frame, start_x = 0):
    linear `y += y_step`, smoothstep `x*x*(3 - 2x)*delta` with `x += x_step`, re-derived and compared bit for bit."""
    L = oracle.lib()
    nodes = [(0.0, 0.0), (0.5, 0.004), (-0.25, 0.0125), (1.0, 0.02)]
    arr = (oracle.CurveNode * 4)(*[oracle.CurveNode(v, t) for v, t in nodes])
    frames = [0, 192, 600, 960]                                # (pinned by test_curve_linear_matches_closed_form)
    n = 1024
    for fn in (0, 1):
        ref = np.zeros(n, np.float32)
        for (f0, f1), ((v0, _), (v1, _)) in zip(zip(frames, frames[1:]), zip(nodes, nodes[1:])):
            start_x = f32(0) / f32(f1 - f0); x_step = f32(1.0) / f32(f1 - f0)
            start_value, delta = f32(v0), f32(v1) - f32(v0)
            if fn == 0:
                y = start_value + start_x * delta; y_step = x_step * delta
                for i in range(f0, f1):
                    ref[i] = ref[i] + y; y = y + y_step
            else:
                x = start_x
                for i in range(f0, f1):
                    v = x * x * (f32(3.0) - f32(2.0) * x) * delta
                    ref[i] = ref[i] + (start_value + v); x = x + x_step
        st = oracle.CurveModule(); L.zo_curve_init(C.byref(st)); out = np.zeros(n, np.float32)
        L.zo_curve_paint(C.byref(st), 0, n, oracle.fptr(out), 1, SR, fn, arr, 4)
        util.assert_bitexact(out, ref, f"curve function {fn}")


def test_white_noise_from_published_xoshiro_and_float_conversion(oracle):
    """Noise.zig:22-53 over the un-vendored Zig std: DefaultPrng = xoshiro256++ seeded through SplitMix64 (Vigna's published
    reference code), Random.float(f32) = 23 mantissa bits of one u64 draw under an exponent of 126 - clz(draw) (a second draw
    when the first has >= 41 leading zeros), white = float * 2 - 1.  Python integers here; the oracle in C."""
    M = (1 << 64) - 1
    rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & M

    def splitmix(state):
        state = (state + 0x9e3779b97f4a7c15) & M
        z = state
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
        return state, z ^ (z >> 31)

    def seed(s0):
        out = []
        for _ in range(4):
            s0, v = splitmix(s0)
            out.append(v)
        return out

    def nxt(s):
        r = (rotl((s[0] + s[3]) & M, 23) + s[0]) & M
        t = (s[1] << 17) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        return r

    clz64 = lambda x: 64 - x.bit_length()

    def rfloat(s):
        rand = nxt(s)
        lz = clz64(rand)
        if lz >= 41:
            lz = 41 + clz64(nxt(s))
        bits = ((126 - lz) << 23) | (rand & 0x7FFFFF)
        return np.array([bits], np.uint32).view(np.float32)[0]

    L = oracle.lib()
    n = 400
    for sd in (0, 1, 12345, (1 << 40) + 7):
        s = seed(sd)
        ref = np.array([rfloat(s) * f32(2.0) - f32(1.0) for _ in range(n)], np.float32)
        st = oracle.Noise(); L.zo_noise_init(C.byref(st), sd); out = np.zeros(n, np.float32)
        L.zo_noise_paint(C.byref(st), 0, n, oracle.fptr(out), oracle.NOISE_WHITE)
        util.assert_bitexact(out, ref, f"white noise seed {sd}")
        assert [int(x) for x in st.r] == s
    # the second-draw branch: a state whose next draw is below 2^23 (found by walking the generator backwards in the GPU tests;
    # here: force it by construction -- s0 = 0, s3 = k << 41 gives rotl(k << 41, 23) + 0 = k)
    s = [0, 0x123456789abcdef, 0xfedcba987654321, 5 << 41]
    st = oracle.Noise(); L.zo_noise_init(C.byref(st), 0)
    for i in range(4):
        st.r[i] = s[i]
    out = np.zeros(3, np.float32)
    L.zo_noise_paint(C.byref(st), 0, 3, oracle.fptr(out), oracle.NOISE_WHITE)
    ref = np.array([rfloat(s) * f32(2.0) - f32(1.0) for _ in range(3)], np.float32)
    util.assert_bitexact(out, ref, "second-draw branch"); assert [int(x) for x in st.r] == s
