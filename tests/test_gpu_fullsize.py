"""GPU: BASELINE.json's full sizes, checked through sampled voices (the oracle cannot render a
million voices in seconds) and size-independent properties; plus API edge cases."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024


def test_pulseosc_one_million_voices_sampled(ctx, oracle):
    """Config-5 scale for the config-2 kernel: 1,048,576 voices x 1024 frames (a 4 GiB image).
    Voices are independent, so a strided sample of 512 voices must equal the oracle bit for bit."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 1 << 20
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    m = mod.PulseOsc(V, ctx)
    out = ctx.image(F, V)
    fr, col = util.dev(freq), util.dev(color)
    for _ in range(2):                                   # second buffer: carried state
        m.paint(zang.Span(0, F), [out], [], False, m.Params(SR, zang.constant(fr), col), zero_first=True)
    ctx.sync()
    idx = np.arange(0, V, V // 512)
    got = out[:, torch.from_numpy(idx).cuda()].cpu().numpy().T
    L = oracle.lib()
    ref = np.zeros((len(idx), F), np.float32)
    for k, v in enumerate(idx):
        st = oracle.PulseOsc(); L.zo_pulseosc_init(C.byref(st))
        for _ in range(2):
            ref[k] = 0
            L.zo_pulseosc_paint(C.byref(st), 0, F, oracle.fptr(ref[k]), SR, oracle.constant(freq[v]), float(color[v]))
    util.assert_bitexact(got, ref, "1Mi voices sampled")
    # property: |sample| <= gain * (1 + 2*gdf headroom) and the image is fully written (no NaN)
    assert bool(torch.isfinite(out).all())
    del out


def test_nice_131072_voices_sampled_and_mix_property(ctx, oracle):
    """One GPU's shard of config 5: 131,072 NiceInstrument voices.  Sampled voices vs the oracle; the
    fused mix equals the sum of the per-voice image within the sqrt(V) eps bound; mix is linear:
    mix(voices A) + mix(voices B) == mix(A u B) within the same bound."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 131072
    freq, color, _, _ = workloads.voice_params(5, 0, V)
    gf, gc = util.dev(freq), util.dev(color)
    m = mod.NiceInstrument(V, gc, ctx); mm = mod.NiceInstrument(V, gc, ctx)
    out = ctx.image(F, V)
    mix = torch.zeros(F, dtype=torch.float32, device="cuda")
    for b, (on, nic) in enumerate([(True, True), (True, False), (False, False)]):
        m.paint(zang.Span(0, F), [out], None, nic, m.Params(SR, gf, on), zero_first=True)
        mm.paint_mix(zang.Span(0, F), mix, nic, mm.Params(SR, gf, on), zero_first=True)
    ctx.sync()
    idx = np.arange(0, V, V // 128)
    got = out[:, torch.from_numpy(idx).cuda()].cpu().numpy().T
    L = oracle.lib()
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    ref = np.zeros((len(idx), F), np.float32)
    for k, v in enumerate(idx):
        st = oracle.NiceInstrument(); L.zo_nice_init(C.byref(st), float(color[v]))
        for (on, nic) in [(1, 1), (1, 0), (0, 0)]:
            ref[k] = 0
            L.zo_nice_paint(C.byref(st), 0, F, oracle.fptr(ref[k]), oracle.fptr(t0), oracle.fptr(t1), nic, SR, float(freq[v]), on)
    util.assert_bitexact(got, ref, "nice 131072 sampled")
    total = out.double().sum(dim=1).cpu().numpy()
    bound = 8 * np.sqrt(V) * np.finfo(np.float32).eps * float(out.abs().double().sum(dim=1).max())
    assert np.abs(mix.cpu().numpy() - total).max() <= bound
    half = V // 2
    a = torch.zeros(F, dtype=torch.float32, device="cuda"); b = torch.zeros(F, dtype=torch.float32, device="cuda")
    zang.mixdownVoices(zang.Span(0, F), a, out[:, :half], ctx=ctx)
    zang.mixdownVoices(zang.Span(0, F), b, out[:, half:], ctx=ctx)
    ctx.sync()
    assert np.abs((a + b).cpu().numpy() - total).max() <= bound


def test_config3_full_size_unfused_and_fused(ctx, oracle):
    """BASELINE configs[2] at its stated size and parameters: 4,096 voices, Noise(white, seed = voice)
    -> temp -> Filter(low_pass, cutoff = cutoffFromFrequency(200 + 7800 u, 48000), res = 0.9 u, both constant), recipe
    examples/example_stereo.zig:71-82; two consecutive buffers with carried state; the unfused module pair and the
    fused NoiseFilter kernel, every voice against the oracle, bit for bit."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 4096
    _, _, u2, u3 = workloads.voice_params(3, 0, V)
    cutoff_hz = (200.0 + 7800.0 * u2).astype(np.float32)
    res = (0.9 * u3).astype(np.float32)
    L = oracle.lib()
    cutoff = np.array([L.zo_filter_cutoff_from_frequency(float(f), SR) for f in cutoff_hz], np.float32)
    g_cut = mod.Filter.cutoffFromFrequency(util.dev(cutoff_hz), SR, ctx)
    util.assert_bitexact(g_cut.cpu().numpy(), cutoff, "cutoffFromFrequency at 4096 voices")
    g_res = util.dev(res)
    # oracle: zero(temp); noise.paint(temp); zero(out); flt.paint(out) per voice and buffer
    ref = np.zeros((2, V, F), np.float32)
    temp = np.zeros(F, np.float32)
    for v in range(V):
        nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), v)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for b in range(2):
            L.zo_zero(0, F, oracle.fptr(temp))
            L.zo_noise_paint(C.byref(nz), 0, F, oracle.fptr(temp), 0)
            L.zo_filter_paint(C.byref(fl), 0, F, oracle.fptr(ref[b, v]), oracle.fptr(temp), 1, oracle.constant(cutoff[v]), oracle.constant(res[v]))
    sp = zang.Span(0, F)
    noise, flt = mod.Noise(V, ctx, first_seed=0), mod.Filter(V, ctx)
    fused = mod.NoiseFilter(V, ctx, first_seed=0)
    tmp = ctx.image(F, V)
    for b in range(2):
        out_u, out_f = ctx.image(F, V), ctx.image(F, V)
        noise.paint(sp, [tmp], [], False, noise.Params(noise.white), zero_first=True)
        flt.paint(sp, [out_u], [], False, flt.Params(tmp, flt.low_pass, zang.constant(g_cut), zang.constant(g_res)), zero_first=True)
        fused.paint(sp, [out_f], None, False, fused.Params(0, mod.Filter.low_pass, g_cut, g_res), zero_first=True)
        ctx.sync()
        util.assert_bitexact(util.from_image(out_u), ref[b], f"config 3 unfused, buffer {b}")
        util.assert_bitexact(util.from_image(out_f), ref[b], f"config 3 fused, buffer {b}")
    assert np.abs(ref[1]).max() > 0.05                                    # it makes sound


def test_config4_full_song_385s(ctx, oracle):
    """BASELINE configs[3] at its stated size: a repo-authored song in the reference's tracker grammar with the
    reference song's statistics (tools/gen_song.py: 2,900 rows, 385 s), 17 sub-voices (3 PMOsc + 10 + 4
    NiceInstrument), 385 s x 48 kHz = 18,480,000 frames = 18,047 buffers with a partial last one
    (examples/write_wav.zig:7,58-59): the GPU render's s16 payload equals the oracle render's, byte for byte."""
    import contextlib
    import hashlib
    import io
    import importlib.util
    import os
    from zang_amd import song
    from tests.test_song import _oracle_song_render
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_song", os.path.join(root, "tools", "gen_song.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gen.main(2900, 20240915)
    text = buf.getvalue()
    seconds = 385.0
    total = int(seconds * 48000)
    nbuf, last = (total + F - 1) // F, total - (total - 1) // F * F
    assert (nbuf, last) == (18047, 896)
    r = song.SongRenderer(text, ctx)
    got = r.render(seconds)
    assert len(got) == total * 2
    ref = _oracle_song_render(oracle, r.notes, song.EXAMPLE_SONG_INSTRUMENTS, nbuf, last_frames=last)
    assert len(ref) == total * 2
    if got != ref:
        a = np.frombuffer(got, "<i2").astype(np.int32); b = np.frombuffer(ref, "<i2").astype(np.int32)
        raise AssertionError(f"{(a != b).sum()} of {a.size} s16 samples differ (max {np.abs(a - b).max()} LSB), first at {np.argmax(a != b)}")
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(ref).hexdigest()
    assert np.abs(np.frombuffer(got, "<i2").astype(np.int32)).max() > 1000


def test_config5_shard_48_buffer_cycle(ctx, oracle):
    """BASELINE configs[4], one GPU's shard at its stated size: 131,072 NiceInstrument voices over the whole 48-buffer
    note cycle of the bench (note on for buffers 0-23, then off: attack -> decay -> sustain -> release), mono and
    stereo mixdown kernels against the f64 sum of the per-voice image (sqrt(V) eps bound), 256 sampled voices
    against the oracle bit for bit at four points of the cycle."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    V = 131072
    freq, color, u2, _ = workloads.voice_params(5, 0, V)
    pan = (2.0 * u2 - 1.0).astype(np.float32)
    gl = (np.float32(0.0) + ((np.float32(0.0) + pan * np.float32(0.5)) + np.float32(0.5))).astype(np.float32)
    gr = (np.float32(0.0) + ((np.float32(0.0) + gl * np.float32(-1.0)) + np.float32(1.0))).astype(np.float32)
    gf, gc, dgl, dgr = util.dev(freq), util.dev(color), util.dev(gl), util.dev(gr)
    m, mm, ms = mod.NiceInstrument(V, gc, ctx), mod.NiceInstrument(V, gc, ctx), mod.NiceInstrument(V, gc, ctx)
    out = ctx.image(F, V)
    mono = torch.zeros(F, device="cuda"); left = torch.zeros(F, device="cuda"); right = torch.zeros(F, device="cuda")
    idx = np.arange(0, V, V // 256)
    gidx = torch.from_numpy(idx).cuda()
    L = oracle.lib()
    sts = []
    for v in idx:
        st = oracle.NiceInstrument(); L.zo_nice_init(C.byref(st), float(color[v])); sts.append(st)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    ref = np.zeros((len(idx), F), np.float32)
    sp = zang.Span(0, F)
    eps = np.finfo(np.float32).eps
    peak = 0.0
    for b in range(48):
        on, nic = b < 24, b == 0
        P = m.Params(SR, gf, on)
        m.paint(sp, [out], None, nic, P, zero_first=True)
        mm.paint_mix(sp, mono, nic, P, zero_first=True)
        ms.paint_mix_stereo(sp, left, right, dgl, dgr, nic, P, zero_first=True)
        ctx.sync()
        for k, v in enumerate(idx):
            ref[k] = 0
            L.zo_nice_paint(C.byref(sts[k]), 0, F, oracle.fptr(ref[k]), oracle.fptr(t0), oracle.fptr(t1), int(nic), SR, float(freq[v]), int(on))
        if b in (0, 23, 24, 47):
            util.assert_bitexact(out[:, gidx].cpu().numpy().T, ref, f"config 5 sampled voices, buffer {b}")
        o64 = out.double()
        absmax = float(out.abs().double().sum(dim=1).max())
        bound = 8 * np.sqrt(V) * eps * max(absmax, 1.0)
        assert np.abs(mono.cpu().numpy() - o64.sum(dim=1).cpu().numpy()).max() <= bound, f"mono mix, buffer {b}"
        for got, g in ((left, dgl), (right, dgr)):
            prod = (out * g[None, :]).double()                                # f32 products (torch multiplies in f32), f64 sum
            assert np.abs(got.cpu().numpy() - prod.sum(dim=1).cpu().numpy()).max() <= bound, f"stereo mix, buffer {b}"
        peak = max(peak, float(mono.abs().max()))
    assert peak > 1.0
    assert np.array_equal(m.state(), mm.state()) and np.array_equal(m.state(), ms.state())


@pytest.mark.parametrize("V", [24576, 49152])
def test_frame_range_forms_at_mid_voice_counts(ctx, oracle, monkeypatch, V):
    """Between 16,384 and ~131,072 voices SineOsc, Sampler, the controlled-frequency oscillators, PMOscInstrument, Envelope and
    Decimator paint a span as a few frame ranges (zh_range_frames, ctx.hip: more waves per SIMD), and Filter, FilteredEchoes,
    NiceInstrument run as wave pipelines, SimpleDelay as independent frames -- each up to its own voice-count limit.  Two
    carried buffers in the default form equal the one-wave walks (every switch off) bit for bit, images and states; sampled
    SineOsc voices equal the oracle."""
    import torch
    from zang_amd import modules as mod, zang, workloads
    freq, color, u2, _ = workloads.voice_params(5, 0, V)
    gf, gc = util.dev(freq), util.dev(color)
    fbuf = ctx.image(F, V); fbuf.copy_(gf[None, :].expand(F, V)); fbuf.mul_(1.0 + 0.25 * torch.rand(F, 1, device="cuda"))
    pcm = util.dev(np.random.default_rng(4).integers(-20000, 20000, 9000, dtype=np.int16).view(np.uint8).copy())
    rel = util.dev((0.1 + 0.4 * u2).astype(np.float32))
    span = zang.Span(0, F)

    def render():
        outs, states = [], []
        m = mod.SineOsc(V, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(SR, zang.constant(gf), zang.constant(0.0)), zero_first=True)
        outs.append(o); states.append(m.state()["t"])
        m = mod.SineOsc(V, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), zang.constant(0.25)), zero_first=True)
        outs.append(o); states.append(m.state()["t"])
        m = mod.Sampler(V, ctx); o = ctx.image(F, V); smp = m.Sample(1, 44100, m.signed16_lsb, pcm)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(gf * 40.0, smp, 0, True), zero_first=True)
        outs.append(o); states.append(m.state()["t"])
        m = mod.PulseOsc(V, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), gc), zero_first=True)
        outs.append(o); states.append(m.state()["cnt"])
        m = mod.PMOscInstrument(V, rel, ctx); o = ctx.image(F, V)
        for k in range(2):
            m.paint(span, [o], None, k == 0, m.Params(SR, gf, k == 0), zero_first=True)
        outs.append(o); states.append(m.state())
        # the wave pipelines and the other frame-range forms of round 2, against their one-wave walks
        m = mod.Envelope(V, ctx); o = ctx.image(F, V)
        for k in range(2):
            m.paint(span, [o], [], k == 0, m.Params(SR, zang.PaintCurve.cubed(0.004), zang.PaintCurve.cubed(0.02), zang.PaintCurve.cubed(0.03), 0.6, k == 0), zero_first=True)
        outs.append(o); states.append(m.state())
        m = mod.Decimator(V, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(SR, fbuf, gf * 8.0), zero_first=True)
        outs.append(o); states.append(m.state())
        m = mod.Filter(V, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(fbuf, m.band_pass, zang.constant(gc), zang.constant(0.4)), zero_first=True)
        outs.append(o); states.append(m.state())
        m = mod.FilteredEchoes(V, 300, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], None, False, m.Params(fbuf, 0.5, 0.2), zero_first=True)
        outs.append(o); states.append(np.concatenate([np.asarray(x).ravel().view(np.uint8) for x in m.state()]))
        m = mod.SimpleDelay(V, 300, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(fbuf), zero_first=True)
        outs.append(o); states.append(np.concatenate([np.asarray(x).ravel().view(np.uint8) for x in m.state()]))
        m = mod.NiceInstrument(V, gc, ctx); o = ctx.image(F, V)
        for k in range(2):
            m.paint(span, [o], None, k == 0, m.Params(SR, gf, k == 0), zero_first=True)
        outs.append(o); states.append(m.state())
        m = mod.TriSawOsc(V, ctx); o = ctx.image(F, V)
        for _ in range(2):
            m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), gc), zero_first=True)
        outs.append(o); states.append(m.state())
        ctx.sync()
        return outs, states

    a_out, a_st = render()
    for name in ("sine_ranges", "sampler_ranges", "pulse_ctrl_ranges", "pmosc_ranges", "envelope_ranges", "decimator_ranges",
                 "trisaw_ctrl_ranges", "filter_pc_max", "echoes_pc_max", "delay_frames_max", "nice_pc_max", "nice_pc4_max"):
        util.set_form(monkeypatch, **{name: 0})
    b_out, b_st = render()
    names = ("sineosc const", "sineosc image", "sampler", "pulseosc image", "pmosc", "envelope", "decimator", "filter", "filtered echoes",
             "simple delay", "nice", "trisawosc image")
    assert len(a_out) == len(names)
    for name, x, y, sx, sy in zip(names, a_out, b_out, a_st, b_st):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), name
        assert np.asarray(sx).tobytes() == np.asarray(sy).tobytes(), name + " state"
    idx = np.arange(0, V, V // 64)
    got = a_out[0][:, torch.from_numpy(idx).cuda()].cpu().numpy().T
    L = oracle.lib()
    ref = np.zeros((len(idx), F), np.float32)
    for k, v in enumerate(idx):
        st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st))
        for _ in range(2):
            ref[k] = 0
            L.zo_sineosc_paint(C.byref(st), 0, F, oracle.fptr(ref[k]), SR, oracle.constant(freq[v]), oracle.constant(0.0))
    util.assert_bitexact(got, ref, "sineosc sampled vs oracle")


def test_zero_voices_and_empty_spans(ctx):
    """Empty inputs: a module with 0 voices and 0-length spans are no-ops, not errors."""
    import torch
    from zang_amd import modules as mod, zang
    img = ctx.image(16, 8, fill=1.0)
    for cls, params in ((mod.PulseOsc, lambda m: m.Params(SR, zang.constant(440.0), 0.5)),
                        (mod.SineOsc, lambda m: m.Params(SR, zang.constant(440.0), zang.constant(0.0))),
                        (mod.Noise, lambda m: m.Params(0)), (mod.Gate, lambda m: m.Params(True))):
        m0 = cls(0, ctx)
        m0.paint(zang.Span(0, 16), [img], [], False, params(m0))
        m8 = cls(8, ctx)
        m8.paint(zang.Span(5, 5), [img], [], False, params(m8))
    zang.mixdownVoices(zang.Span(3, 3), torch.zeros(16, device="cuda"), img, ctx=ctx)
    ctx.sync()
    assert float(img.min()) == 1.0 and float(img.max()) == 1.0


def test_bad_arguments_are_rejected(ctx):
    from zang_amd import modules as mod, zang, abi
    m = mod.PulseOsc(64, ctx)
    small = ctx.image(16, 32)                              # fewer voices than the module
    with pytest.raises(abi.ZangHipError):
        m.paint(zang.Span(0, 16), [small], [], False, m.Params(SR, zang.constant(440.0), 0.5))
    img = ctx.image(16, 64)
    with pytest.raises(abi.ZangHipError):
        m.paint(zang.Span(0, 17), [img], [], False, m.Params(SR, zang.constant(440.0), 0.5))   # span beyond the image
    with pytest.raises(abi.ZangHipError):
        m.paint(zang.Span(9, 3), [img], [], False, m.Params(SR, zang.constant(440.0), 0.5))    # end < start
    f = mod.Filter(64, ctx)
    with pytest.raises(abi.ZangHipError):
        f.paint(zang.Span(0, 16), [img], [], False, f.Params(img, 17, zang.constant(0.5), zang.constant(0.5)))  # bad Filter.Type
    n = mod.Noise(64, ctx)
    with pytest.raises(abi.ZangHipError):
        n.paint(zang.Span(0, 16), [img], [], False, n.Params(5))                                   # bad Noise.Color
