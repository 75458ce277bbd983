"""GPU: every kernel form that is selected BY VOICE COUNT, against the ORACLE at the voice counts that select it.

The library picks a form per paint from the voice count (frame ranges with a state replay, wave pipelines with 32- or
16-frame tiles, one-wave walks ...; limits at 16,384 / 32,768 / 40,960 / 65,536 / 131,072 voices).  The small-voice-count
parity tests (test_gpu_modules.py ...) hold each form against the oracle where it can be forced through a switch; here every
module is painted in its DEFAULT form at each boundary voice count (and just past the last one) -- two carried buffers, the
first with ZERO_FIRST, the second adding -- and the image columns and final states of sampled voices (a stride through the
whole range, the first and the last voice, voices next to wave and block edges) are compared bit for bit with the oracle.
A second test forces the geometries that only large voice counts reach (2-3 white-noise ranges, 16-frame filter tiles,
2-4 Decimator / Curve / Envelope / ... ranges) at a small voice count, every voice checked.
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024
D = 300                 # delay samples of the echo modules


class Shared:
    """Inputs shared by the cases of one voice count: per-voice params, control / input images, and their sampled columns."""

    def __init__(self, ctx, V, idx):
        import torch
        from zang_amd import workloads
        self.V, self.idx = V, idx
        self.freq, self.color, self.u2, self.u3 = workloads.voice_params(5, 0, V)
        self.gf, self.gc = util.dev(self.freq), util.dev(self.color)
        g = torch.Generator(device="cuda"); g.manual_seed(1234 + V)
        wob = 1.0 + 0.25 * torch.rand(F, 1, device="cuda", generator=g)
        self.fbuf = ctx.image(F, V); self.fbuf.copy_(self.gf[None, :].expand(F, V)); self.fbuf.mul_(wob)      # a frequency image
        self.ibuf = ctx.image(F, V); self.ibuf.copy_(torch.rand(F, V, device="cuda", generator=g) * 2.0 - 1.0)   # an input signal
        self.cbuf = ctx.image(F, V); self.cbuf.copy_(torch.rand(F, V, device="cuda", generator=g) * 0.9 + 0.02)  # cutoffs in (0, 1)
        tidx = torch.from_numpy(idx).cuda()
        self.tidx = tidx
        self.fcol = np.ascontiguousarray(self.fbuf[:, tidx].cpu().numpy().T)
        self.icol = np.ascontiguousarray(self.ibuf[:, tidx].cpu().numpy().T)
        self.ccol = np.ascontiguousarray(self.cbuf[:, tidx].cpu().numpy().T)
        self.pcm = np.random.default_rng(4).integers(-20000, 20000, 9000, dtype=np.int16).view(np.uint8).copy()
        self.gpcm = util.dev(self.pcm)

    def cols(self, img):
        return np.ascontiguousarray(img[:, self.tidx].cpu().numpy().T)


def sample_voices(V, n=48):
    idx = set(range(0, V, max(1, V // n)))
    idx |= {0, 1, 63, 64, 65, 255, 256, V - 1, V - 2, V - 64, V - 65, V // 2 - 1, V // 2}
    return np.array(sorted(i for i in idx if 0 <= i < V), dtype=np.int64)


def _crafted_noise_state(k_frames, seed):
    """A xoshiro256++ state whose k_frames-th draw from now is one of Random.float's multi-draw samples (2^-41 per sample)."""
    from tests.test_gpu_modules import _xoshiro_step_back
    rng = np.random.default_rng(seed)
    return _xoshiro_step_back([0, int(rng.integers(1, 1 << 63)), int(rng.integers(1, 1 << 63)), 1 << 41], k_frames)


# ---------------------------------------------------------------------------------------------- the cases
# each: case(ctx, oracle, sh) -> None; paints on the GPU, runs the oracle for sh.idx, asserts bit-exact images and states

def _two_paints(paint):
    for k in range(2):
        paint(k)


def _check(sh, ctx, name, out, ref, state_pairs):
    ctx.sync()
    util.assert_bitexact(sh.cols(out), ref, f"{name} at {sh.V} voices: image columns of sampled voices")
    for what, got, want in state_pairs():
        g = np.asarray(got)
        if g.dtype.kind == "f":
            util.assert_bitexact(g.astype(np.float32), np.asarray(want, dtype=np.float32), f"{name} at {sh.V} voices: state {what}")
        else:                                               # (u64 generator words: never through a float array)
            flat = [int(x) for row in want for x in (row if isinstance(row, (list, tuple)) else [row])]
            assert [int(x) for x in g.ravel()] == flat, f"{name} at {sh.V} voices: state {what}"


def case_sineosc(ctx, oracle, sh, image):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.SineOsc(sh.V, ctx); o = ctx.image(F, sh.V)
    fr = zang.buffer(sh.fbuf) if image else zang.constant(sh.gf)
    ph = 0.25 if image else 0.0
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(SR, fr, zang.constant(ph)), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rt = []
    for j, v in enumerate(sh.idx):
        st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st))
        for _ in range(2):
            L.zo_sineosc_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), SR, oracle.buffer(sh.fcol[j]) if image else oracle.constant(sh.freq[v]), oracle.constant(ph))
        rt.append(st.t)
    _check(sh, ctx, "SineOsc " + ("freq image" if image else "const"), o, ref, lambda: [("t", m.state()["t"][sh.idx], rt)])


def case_osc(ctx, oracle, sh, which, image):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    cls, ocls, init, paint = ((mod.PulseOsc, oracle.PulseOsc, L.zo_pulseosc_init, L.zo_pulseosc_paint) if which == "pulse" else
                              (mod.TriSawOsc, oracle.TriSawOsc, L.zo_trisawosc_init, L.zo_trisawosc_paint))
    m = cls(sh.V, ctx); o = ctx.image(F, sh.V)
    fr = zang.buffer(sh.fbuf) if image else zang.constant(sh.gf)
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(SR, fr, sh.gc), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        st = ocls(); init(C.byref(st))
        for _ in range(2):
            paint(C.byref(st), 0, F, oracle.fptr(ref[j]), SR, oracle.buffer(sh.fcol[j]) if image else oracle.constant(sh.freq[v]), float(sh.color[v]))
        rs.append((st.cnt, st.t if which == "trisaw" else 0.0))
    def states():
        gs = m.state()
        out = [("cnt", gs["cnt"][sh.idx], [r[0] for r in rs])]
        if which == "trisaw":
            out.append(("t", gs["t"][sh.idx], [r[1] for r in rs]))
        return out
    _check(sh, ctx, f"{which} osc " + ("freq image" if image else "const"), o, ref, states)


def case_sampler(ctx, oracle, sh):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.Sampler(sh.V, ctx); o = ctx.image(F, sh.V)
    smp = m.Sample(1, 44100, m.signed16_lsb, sh.gpcm)
    rate = (sh.freq * np.float32(40.0)).astype(np.float32)
    gr = util.dev(rate)
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(gr, smp, 0, True), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rt = []
    for j, v in enumerate(sh.idx):
        st = oracle.Sampler(); L.zo_sampler_init(C.byref(st))
        p = oracle.SamplerParams(float(rate[v]), 1, 44100, oracle.SAMPLE_S16, sh.pcm.ctypes.data_as(C.POINTER(C.c_uint8)), sh.pcm.size, 0, 1)
        for _ in range(2):
            L.zo_sampler_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), 0, C.byref(p))
        rt.append(st.t)
    _check(sh, ctx, "Sampler", o, ref, lambda: [("t", m.state()["t"][sh.idx], rt)])


def _env_params(oracle, on):
    return oracle.EnvelopeParams(SR, oracle.curve(3, 0.004), oracle.curve(3, 0.02), oracle.curve(3, 0.03), 0.6, int(on))


def case_envelope(ctx, oracle, sh):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.Envelope(sh.V, ctx); o = ctx.image(F, sh.V)
    on1 = (np.arange(sh.V) % 5 != 0)                     # second buffer: most voices stay on (decay -> sustain), every fifth releases
    g_on1 = util.dev(on1.astype(np.uint8))
    P = lambda on: m.Params(SR, zang.PaintCurve.cubed(0.004), zang.PaintCurve.cubed(0.02), zang.PaintCurve.cubed(0.03), 0.6, on)
    m.paint(zang.Span(0, F), [o], [], True, P(True), zero_first=True)
    m.paint(zang.Span(0, F), [o], [], False, P(g_on1))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        st = oracle.Envelope(); L.zo_envelope_init(C.byref(st))
        L.zo_envelope_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), 1, C.byref(_env_params(oracle, True)))
        L.zo_envelope_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), 0, C.byref(_env_params(oracle, on1[v])))
        rs.append((st.state, st.painter.t, st.painter.last_value, st.painter.start))
    def states():
        gs = m.state()
        return [(n, gs[n][sh.idx], [r[i] for r in rs]) for i, n in enumerate(("state", "t", "last_value", "start"))]
    _check(sh, ctx, "Envelope", o, ref, states)


def case_decimator(ctx, oracle, sh):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.Decimator(sh.V, ctx); o = ctx.image(F, sh.V)
    fake = (sh.freq * np.float32(8.0)).astype(np.float32)
    gfake = util.dev(fake)
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(SR, sh.ibuf, gfake), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        st = oracle.Decimator(); L.zo_decimator_init(C.byref(st))
        for _ in range(2):
            L.zo_decimator_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), SR, oracle.fptr(sh.icol[j]), float(fake[v]))
        rs.append((st.dval, st.dcount))
    def states():
        gs = m.state()
        return [("dval", gs["dval"][sh.idx], [r[0] for r in rs]), ("dcount", gs["dcount"][sh.idx], [r[1] for r in rs])]
    _check(sh, ctx, "Decimator", o, ref, states)


def case_filter(ctx, oracle, sh, ftype, cutoff_image):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.Filter(sh.V, ctx); o = ctx.image(F, sh.V)
    cut = zang.buffer(sh.cbuf) if cutoff_image else zang.constant(sh.gc)
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(sh.ibuf, ftype, cut, zang.constant(0.4)), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        st = oracle.Filter(); L.zo_filter_init(C.byref(st))
        for _ in range(2):
            L.zo_filter_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), oracle.fptr(sh.icol[j]), ftype,
                              oracle.buffer(sh.ccol[j]) if cutoff_image else oracle.constant(sh.color[v]), oracle.constant(0.4))
        rs.append((st.l, st.b))
    def states():
        gs = m.state()
        return [("l", gs["l"][sh.idx], [r[0] for r in rs]), ("b", gs["b"][sh.idx], [r[1] for r in rs])]
    _check(sh, ctx, f"Filter type {ftype} " + ("cutoff image" if cutoff_image else "const"), o, ref, states)


def case_echoes(ctx, oracle, sh, filtered):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    o = ctx.image(F, sh.V)
    n = len(sh.idx)
    ref = np.zeros((n, F), np.float32); rings = np.zeros((n, D), np.float32); rs = []
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    if filtered:
        m = mod.FilteredEchoes(sh.V, D, ctx)
        _two_paints(lambda k: m.paint(zang.Span(0, F), [o], None, False, m.Params(sh.ibuf, 0.5, 0.2), zero_first=(k == 0)))
    else:
        m = mod.SimpleDelay(sh.V, D, ctx)
        _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(sh.ibuf), zero_first=(k == 0)))
    for j in range(n):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[j]), D)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for _ in range(2):
            if filtered:
                L.zo_filtered_echoes_paint(C.byref(d), C.byref(fl), 0, F, oracle.fptr(ref[j]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(sh.icol[j]), 0.5, 0.2)
            else:
                L.zo_simple_delay_paint(C.byref(d), 0, F, oracle.fptr(ref[j]), oracle.fptr(sh.icol[j]))
        rs.append((d.index, fl.l, fl.b))
    def states():
        st = m.state()
        out = [("ring", st[0][sh.idx], rings), ("index", np.asarray(st[1])[sh.idx], [r[0] for r in rs])]
        if filtered:
            out += [("l", st[2]["l"][sh.idx], [r[1] for r in rs]), ("b", st[2]["b"][sh.idx], [r[2] for r in rs])]
        return out
    _check(sh, ctx, "FilteredEchoes" if filtered else "SimpleDelay", o, ref, states)


def case_nice(ctx, oracle, sh):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.NiceInstrument(sh.V, sh.gc, ctx); o = ctx.image(F, sh.V)
    m.paint(zang.Span(0, F), [o], None, True, m.Params(SR, sh.gf, True), zero_first=True)
    m.paint(zang.Span(0, F), [o], None, False, m.Params(SR, sh.gf, False))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for j, v in enumerate(sh.idx):
        st = oracle.NiceInstrument(); L.zo_nice_init(C.byref(st), float(sh.color[v]))
        L.zo_nice_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), oracle.fptr(t0), oracle.fptr(t1), 1, SR, float(sh.freq[v]), 1)
        L.zo_nice_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), oracle.fptr(t0), oracle.fptr(t1), 0, SR, float(sh.freq[v]), 0)
        rs.append((st.osc.cnt, st.flt.l, st.flt.b, st.env.state, st.env.painter.t, st.env.painter.last_value, st.env.painter.start))
    def states():
        gs = m.state()
        return [("osc.cnt", gs["osc"]["cnt"][sh.idx], [r[0] for r in rs]), ("flt.l", gs["flt"]["l"][sh.idx], [r[1] for r in rs]),
                ("flt.b", gs["flt"]["b"][sh.idx], [r[2] for r in rs]), ("env.state", gs["env"]["state"][sh.idx], [r[3] for r in rs]),
                ("env.t", gs["env"]["t"][sh.idx], [r[4] for r in rs]), ("env.last_value", gs["env"]["last_value"][sh.idx], [r[5] for r in rs]),
                ("env.start", gs["env"]["start"][sh.idx], [r[6] for r in rs])]
    _check(sh, ctx, "NiceInstrument", o, ref, states)


def case_pmosc(ctx, oracle, sh):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    rel = (0.1 + 0.4 * sh.u2).astype(np.float32)
    m = mod.PMOscInstrument(sh.V, util.dev(rel), ctx); o = ctx.image(F, sh.V)
    m.paint(zang.Span(0, F), [o], None, True, m.Params(SR, sh.gf, True), zero_first=True)
    m.paint(zang.Span(0, F), [o], None, False, m.Params(SR, sh.gf, False))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    t = [np.zeros(F, np.float32) for _ in range(3)]
    for j, v in enumerate(sh.idx):
        st = oracle.PMOscInstrument(); L.zo_pmosc_init(C.byref(st), float(rel[v]))
        for k in range(2):
            L.zo_pmosc_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), oracle.fptr(t[0]), oracle.fptr(t[1]), oracle.fptr(t[2]), int(k == 0), SR, float(sh.freq[v]), int(k == 0))
        rs.append((st.carrier.t, st.modulator.t, st.env.state, st.env.painter.t, st.env.painter.last_value, st.env.painter.start))
    def states():
        gs = m.state()
        return [("carrier.t", gs["carrier"]["t"][sh.idx], [r[0] for r in rs]), ("modulator.t", gs["modulator"]["t"][sh.idx], [r[1] for r in rs]),
                ("env.state", gs["env"]["state"][sh.idx], [r[2] for r in rs]), ("env.t", gs["env"]["t"][sh.idx], [r[3] for r in rs]),
                ("env.last_value", gs["env"]["last_value"][sh.idx], [r[4] for r in rs]), ("env.start", gs["env"]["start"][sh.idx], [r[5] for r in rs])]
    _check(sh, ctx, "PMOscInstrument", o, ref, states)


def case_noise(ctx, oracle, sh, color, first_zf=True):
    """White / pink: ZERO_FIRST then ADD (the ADD form of the white frame ranges goes through a module-owned image).  Two
    sampled voices start from crafted generator states: a multi-draw sample inside the first and inside the last range."""
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    first = 5000
    m = mod.Noise(sh.V, ctx, first_seed=first); o = ctx.image(F, sh.V, fill=0.0)       # (the first paint may be an ADD)
    crafted = {int(sh.idx[3]): _crafted_noise_state(77, 1), int(sh.idx[-3]): _crafted_noise_state(1001, 2), int(sh.idx[len(sh.idx) // 2]): _crafted_noise_state(1024 + 515, 3)}
    st = m.state()
    for v, r in crafted.items():
        st["r"][v] = r
    taps = np.random.default_rng(9).uniform(-0.5, 0.5, (sh.V, 7)).astype(np.float32)
    taps[::3] = 0.0
    st["b"][:] = taps
    m.set_state(st)
    m.paint(zang.Span(0, F), [o], [], False, m.Params(color), zero_first=first_zf)
    m.paint(zang.Span(0, F), [o], [], False, m.Params(color))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + int(v))
        if int(v) in crafted:
            for i in range(4):
                nz.r[i] = crafted[int(v)][i]
        for i in range(7):
            nz.b[i] = float(taps[v, i])
        for _ in range(2):
            L.zo_noise_paint(C.byref(nz), 0, F, oracle.fptr(ref[j]), color)
        rs.append(list(nz.r))
    def states():
        gs = m.state()
        return [("r", gs["r"][sh.idx], rs), ("b (never written back, Noise.zig:68)", gs["b"][sh.idx], taps[sh.idx])]
    _check(sh, ctx, "Noise " + ("pink" if color else "white"), o, ref, states)


def case_noise_filter(ctx, oracle, sh, color):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    first = 9000
    cutoff = (0.02 + 0.5 * sh.u2).astype(np.float32); res = (0.9 * sh.u3).astype(np.float32)
    m = mod.NoiseFilter(sh.V, ctx, first_seed=first); o = ctx.image(F, sh.V)
    crafted = {int(sh.idx[5]): _crafted_noise_state(300, 4), int(sh.idx[-2]): _crafted_noise_state(1024 + 9, 5)}
    st = m.state()
    for v, r in crafted.items():
        st["noise"]["r"][v] = r
    m.set_state(st)
    gcut, gres = util.dev(cutoff), util.dev(res)
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], None, False, m.Params(color, mod.Filter.low_pass, gcut, gres), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    temp = np.zeros(F, np.float32)
    for j, v in enumerate(sh.idx):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), first + int(v))
        if int(v) in crafted:
            for i in range(4):
                nz.r[i] = crafted[int(v)][i]
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for _ in range(2):
            L.zo_zero(0, F, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nz), 0, F, oracle.fptr(temp), color)
            L.zo_filter_paint(C.byref(fl), 0, F, oracle.fptr(ref[j]), oracle.fptr(temp), oracle.FILTER_LOW_PASS, oracle.constant(cutoff[v]), oracle.constant(res[v]))
        rs.append((list(nz.r), fl.l, fl.b))
    def states():
        gs = m.state()
        return [("noise.r", gs["noise"]["r"][sh.idx], [r[0] for r in rs]), ("flt.l", gs["flt"]["l"][sh.idx], [r[1] for r in rs]),
                ("flt.b", gs["flt"]["b"][sh.idx], [r[2] for r in rs])]
    _check(sh, ctx, "Noise->Filter fused, " + ("pink" if color else "white"), o, ref, states)


def case_curve(ctx, oracle, sh, function):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    rng = np.random.default_rng(97)
    ts = np.cumsum(rng.uniform(0.0004, 0.006, 24)).astype(np.float32); ts[0] = 0.0
    vals = rng.uniform(-1, 1, 24).astype(np.float32)
    ts[5] = ts[4]
    nodes = np.stack([vals, ts], axis=1).astype(np.float32)
    carr = (oracle.CurveNode * len(nodes))(*[oracle.CurveNode(float(v), float(t)) for v, t in nodes])
    nic1 = (np.arange(sh.V) % 7 == 0)
    m = mod.Curve(sh.V, ctx); o = ctx.image(F, sh.V)
    gn = util.dev(nodes)
    m.paint(zang.Span(0, F), [o], [], True, m.Params(SR, function, gn), zero_first=True)
    m.paint(zang.Span(0, F), [o], [], util.dev(nic1.astype(np.uint8)), m.Params(SR, function, gn))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        st = oracle.CurveModule(); L.zo_curve_init(C.byref(st))
        L.zo_curve_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), 1, SR, function, carr, len(nodes))
        L.zo_curve_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), int(nic1[v]), SR, function, carr, len(nodes))
        rs.append((st.t, st.current_song_note, st.current_song_note_offset, st.next_song_note))
    def states():
        gs = m.state()
        return [(n, gs[n][sh.idx], [r[i] for r in rs]) for i, n in enumerate(("t", "current_song_note", "current_song_note_offset", "next_song_note"))]
    _check(sh, ctx, f"Curve fn {function}", o, ref, states)


def case_cycle(ctx, oracle, sh, image):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    m = mod.Cycle(sh.V, ctx); o = ctx.image(F, sh.V)
    sp = zang.buffer(sh.fbuf) if image else zang.constant(sh.gf)
    _two_paints(lambda k: m.paint(zang.Span(0, F), [o], [], False, m.Params(SR, sp), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32); rt = []
    for j, v in enumerate(sh.idx):
        st = oracle.Cycle(); L.zo_cycle_init(C.byref(st))
        for _ in range(2):
            L.zo_cycle_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), SR, oracle.buffer(sh.fcol[j]) if image else oracle.constant(sh.freq[v]))
        rt.append(st.t)
    _check(sh, ctx, "Cycle " + ("speed image" if image else "const"), o, ref, lambda: [("t", m.state()["t"][sh.idx], rt)])


def case_portamento(ctx, oracle, sh):
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    dur = (0.002 + 0.03 * sh.u2).astype(np.float32)
    goal0 = sh.freq; goal1 = (sh.freq * np.float32(1.5)).astype(np.float32)
    m = mod.Portamento(sh.V, ctx); o = ctx.image(F, sh.V)
    gcurve = zang.PaintCurve.cubed(util.dev(dur))
    m.paint(zang.Span(0, F), [o], [], True, m.Params(SR, gcurve, util.dev(goal0), True, False), zero_first=True)
    m.paint(zang.Span(0, F), [o], [], True, m.Params(SR, gcurve, util.dev(goal1), True, True))
    ref = np.zeros((len(sh.idx), F), np.float32); rs = []
    for j, v in enumerate(sh.idx):
        st = oracle.Portamento(); L.zo_portamento_init(C.byref(st))
        L.zo_portamento_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), 1, SR, oracle.curve(3, dur[v]), float(goal0[v]), 1, 0)
        L.zo_portamento_paint(C.byref(st), 0, F, oracle.fptr(ref[j]), 1, SR, oracle.curve(3, dur[v]), float(goal1[v]), 1, 1)
        rs.append((st.painter.t, st.painter.last_value, st.painter.start))
    def states():
        gs = m.state()
        return [(n, gs[n][sh.idx], [r[i] for r in rs]) for i, n in enumerate(("t", "last_value", "start"))]
    _check(sh, ctx, "Portamento", o, ref, states)


def case_stateless(ctx, oracle, sh):
    """Gate (bit-exact index arithmetic) and Distortion (both types), frame-chunked kernels without state."""
    from zang_amd import modules as mod, zang
    L = oracle.lib()
    on = (np.arange(sh.V) % 3 != 1)
    g = mod.Gate(sh.V, ctx); o = ctx.image(F, sh.V)
    _two_paints(lambda k: g.paint(zang.Span(0, F), [o], [], False, g.Params(util.dev(on.astype(np.uint8))), zero_first=(k == 0)))
    ref = np.zeros((len(sh.idx), F), np.float32)
    for j, v in enumerate(sh.idx):
        for _ in range(2):
            L.zo_gate_paint(0, F, oracle.fptr(ref[j]), int(on[v]))
    _check(sh, ctx, "Gate", o, ref, lambda: [])
    for dtype in (0, 1):
        d = mod.Distortion(sh.V, ctx); o = ctx.image(F, sh.V)
        ing = (0.1 + 0.85 * sh.u2).astype(np.float32); outg = (0.2 + 0.7 * sh.u3).astype(np.float32)
        _two_paints(lambda k: d.paint(zang.Span(0, F), [o], [], False, d.Params(sh.ibuf, dtype, util.dev(ing), util.dev(outg), 0.1), zero_first=(k == 0)))
        ref = np.zeros((len(sh.idx), F), np.float32)
        for j, v in enumerate(sh.idx):
            for _ in range(2):
                L.zo_distortion_paint(0, F, oracle.fptr(ref[j]), oracle.fptr(sh.icol[j]), dtype, float(ing[v]), float(outg[v]), 0.1)
        _check(sh, ctx, f"Distortion type {dtype}", o, ref, lambda: [])


CASES = {
    "sineosc_const": lambda c, o, s: case_sineosc(c, o, s, False),
    "sineosc_image": lambda c, o, s: case_sineosc(c, o, s, True),
    "pulse_const": lambda c, o, s: case_osc(c, o, s, "pulse", False),
    "pulse_image": lambda c, o, s: case_osc(c, o, s, "pulse", True),
    "trisaw_const": lambda c, o, s: case_osc(c, o, s, "trisaw", False),
    "trisaw_image": lambda c, o, s: case_osc(c, o, s, "trisaw", True),
    "sampler": case_sampler,
    "envelope": case_envelope,
    "decimator": case_decimator,
    "filter_lowpass_const": lambda c, o, s: case_filter(c, o, s, 1, False),
    "filter_bandpass_const": lambda c, o, s: case_filter(c, o, s, 2, False),
    "filter_notch_image": lambda c, o, s: case_filter(c, o, s, 4, True),
    "filtered_echoes": lambda c, o, s: case_echoes(c, o, s, True),
    "simple_delay": lambda c, o, s: case_echoes(c, o, s, False),
    "nice": case_nice,
    "pmosc": case_pmosc,
    "noise_white": lambda c, o, s: case_noise(c, o, s, 0),
    "noise_white_add": lambda c, o, s: case_noise(c, o, s, 0, first_zf=False),
    "noise_pink": lambda c, o, s: case_noise(c, o, s, 1),
    "noise_filter_white": lambda c, o, s: case_noise_filter(c, o, s, 0),
    "noise_filter_pink": lambda c, o, s: case_noise_filter(c, o, s, 1),
    "curve_linear": lambda c, o, s: case_curve(c, o, s, 0),
    "curve_smoothstep": lambda c, o, s: case_curve(c, o, s, 1),
    "cycle_const": lambda c, o, s: case_cycle(c, o, s, False),
    "cycle_image": lambda c, o, s: case_cycle(c, o, s, True),
    "portamento": case_portamento,
    "stateless": case_stateless,
}

# every limit a form is selected by (zh_range_frames callers, ZH_*_PC*_MAX defaults, noise_jump.hip), and just past the last
def _table_rows():
    """the library's dispatch table (csrc/dispatch.hip) through the C ABI: {name: default}; loads without a GPU"""
    from zang_amd import abi
    lib = abi.load()
    rows = {}
    for i in range(lib.zh_form_count()):
        name, doc, d, c = C.c_char_p(), C.c_char_p(), C.c_long(), C.c_long()
        assert lib.zh_form_info(i, C.byref(name), C.byref(d), C.byref(c), C.byref(doc)) == 0
        rows[name.value.decode()] = d.value
    return rows


def _boundary_voices():
    """Every voice-count threshold of the table (`*_max` / `*_min` rows between 16,384 and 131,072) and one wave past it, plus the
    voice limits of the frame-range forms (arguments of zh_range_frames at the call sites: 16,384 / 32,768 / 40,960 / 65,536 /
    131,072) and two counts between thresholds."""
    vs = set()
    for name, d in _table_rows().items():
        if (name.endswith("_max") or name.endswith("_min")) and 16384 <= d <= 131072:
            vs.update((d, d + 64))
    for d in (16384, 32768, 40960, 65536, 131072):
        vs.update((d, d + 64))
    vs.update((24576, 49152))
    return sorted(vs)


BOUNDARY_VOICES = _boundary_voices()


@pytest.mark.parametrize("V", BOUNDARY_VOICES)
def test_default_forms_equal_the_oracle_at_dispatch_boundaries(ctx, oracle, V, monkeypatch):
    import torch
    for name in list(__import__("os").environ):
        if name.startswith("ZH_") and name not in ("ZH_ENV_LIVE",):
            monkeypatch.delenv(name)                           # default dispatch, whatever the suite was started with
    idx = sample_voices(V)
    sh = Shared(ctx, V, idx)
    for name, case in CASES.items():
        case(ctx, oracle, sh)
        torch.cuda.empty_cache()


FORCED = [
    # (switches, cases they matter for): geometries that only large voice counts reach, at 333 voices, every voice checked
    ({"noise_ranges": "2"}, ["noise_white", "noise_white_add", "noise_pink"]),
    ({"noise_ranges": "3"}, ["noise_white", "noise_white_add", "noise_pink"]),
    ({"noise_ranges": "0"}, ["noise_white", "noise_pink", "noise_filter_white"]),
    ({"pink_taps": "16", "noise_ranges": "2"}, ["noise_pink"]),
    ({"filter_pc_max": "1"}, ["filter_lowpass_const", "filter_bandpass_const"]),           # 16-frame tiles (ZH_FILTER_PC16_MAX default)
    ({"filter_pc_max": "0"}, ["filter_lowpass_const", "filter_bandpass_const"]),           # the one-wave walk
    ({"decimator_ranges": "2"}, ["decimator"]), ({"decimator_ranges": "3"}, ["decimator"]),
    ({"curve_ranges": "2"}, ["curve_linear", "curve_smoothstep"]), ({"curve_ranges": "3"}, ["curve_linear", "curve_smoothstep"]),
    ({"envelope_ranges": "2"}, ["envelope"]), ({"envelope_ranges": "3"}, ["envelope"]),
    ({"portamento_ranges": "2"}, ["portamento"]), ({"portamento_ranges": "5"}, ["portamento"]),
    ({"cycle_ranges": "2"}, ["cycle_const"]), ({"cycle_ranges": "3"}, ["cycle_const", "cycle_image"]),
    ({"sine_ranges": "2"}, ["sineosc_const", "sineosc_image"]), ({"sine_ranges": "3"}, ["sineosc_const", "sineosc_image"]),
    ({"sampler_ranges": "2"}, ["sampler"]), ({"sampler_ranges": "3"}, ["sampler"]),
    ({"pulse_ctrl_ranges": "2"}, ["pulse_image"]), ({"pulse_ctrl_ranges": "3", "pulse_ctrl_sums": "0"}, ["pulse_image"]),
    ({"trisaw_ctrl_ranges": "2"}, ["trisaw_image"]), ({"trisaw_ctrl_ranges": "3", "trisaw_ctrl_quot": "0"}, ["trisaw_image"]),
    ({"pmosc_ranges": "2"}, ["pmosc"]), ({"pmosc_ranges": "3"}, ["pmosc"]),
    ({"nice_pc4_max": "0"}, ["nice"]), ({"nice_pc_max": "0"}, ["nice"]),
    ({"nf_ring_max": "0"}, ["noise_filter_white"]), ({"nf_ring_max": "0", "nf_pc_max": "0"}, ["noise_filter_white", "noise_filter_pink"]),
    ({"echoes_pc_max": "0"}, ["filtered_echoes"]), ({"delay_frames_max": "0"}, ["simple_delay"]),
]


@pytest.mark.parametrize("k", range(len(FORCED)))
def test_forced_large_voice_count_geometries_at_a_small_voice_count(ctx, oracle, k, monkeypatch):
    env, names = FORCED[k]
    for name in list(__import__("os").environ):
        if name.startswith("ZH_") and name not in ("ZH_ENV_LIVE",):
            monkeypatch.delenv(name)
    for n, v in env.items():
        util.set_form(monkeypatch, **{n: v})
    V = 333
    sh = Shared(ctx, V, np.arange(V, dtype=np.int64))
    for name in names:
        CASES[name](ctx, oracle, sh)


@pytest.mark.parametrize("V", [1280, 4100])
def test_distortion_chunked_form_forced_at_a_small_voice_count(ctx, oracle, V, monkeypatch):
    """k_distortion_chunks (four voices per lane, the per-voice constants once per workgroup through LDS; default from
    distortion_rows_min voices) forced at voice counts whose last workgroup is partial: both types, `+=` after ZERO_FIRST, every voice
    against the oracle bit for bit; the kernel that ran is checked."""
    for name in list(__import__("os").environ):
        if name.startswith("ZH_") and name not in ("ZH_ENV_LIVE",):
            monkeypatch.delenv(name)
    util.set_form(monkeypatch, distortion_rows_min=0)
    sh = Shared(ctx, V, np.arange(V, dtype=np.int64))
    case_stateless(ctx, oracle, sh)
    assert ctx.last_form() == ["k_distortion_chunks"], ctx.last_form()
    util.set_form(monkeypatch, distortion_rows_min=1 << 30)
    case_stateless(ctx, oracle, sh)
    assert ctx.last_form() == ["k_distortion"], ctx.last_form()
