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
