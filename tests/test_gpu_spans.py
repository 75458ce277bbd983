"""GPU parity: span-table paints (one launch = every voice's Trigger sub-spans), the sequential
voice mix and mixDown, vs the oracle running the reference's per-sub-span paint sequence."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024


def _random_tables(V, nbuf, seed):
    """Per buffer, per voice: 0-3 ascending sub-spans (adjacent, edge-touching and gapped ones)."""
    rng = np.random.default_rng(seed)
    bufs = []
    for _ in range(nbuf):
        per_voice = []
        for v in range(V):
            k = int(rng.integers(0, 4))
            cuts = np.sort(rng.choice(np.arange(1, F), size=2 * k, replace=False)) if k else np.array([], int)
            spans = []
            for j in range(k):
                s, e = int(cuts[2 * j]), int(cuts[2 * j + 1])
                if rng.random() < 0.3: s = 0 if j == 0 else spans[-1][1]     # start at buffer start / adjacent to previous
                if rng.random() < 0.3 and j == k - 1: e = F                  # end with the buffer
                spans.append((s, e, float(np.float32(rng.uniform(30, 3000))), bool(rng.random() < 0.7), bool(rng.random() < 0.5)))
            per_voice.append(spans)
        bufs.append(per_voice)
    # voice 0: the carry-over pattern of trigger_test.zig:77-115
    bufs[0][0] = [(0, 200, 440.0, True, True), (200, 1024, 220.0, True, True)]
    if nbuf > 1: bufs[1][0] = [(0, 500, 220.0, True, False), (500, 600, 330.0, True, True), (600, 1024, 660.0, False, False)]
    if nbuf > 2: bufs[2][0] = [(0, 1024, 660.0, False, False)]
    return bufs


@pytest.mark.parametrize("V", [200, 20, 64, 65])
def test_nice_paint_spans(ctx, oracle, V):
    """V > 64: lane-per-voice segment walk (k_nice_spans); V <= 64: one wave per voice, lanes = frames
    (k_nice_spans_wave)."""
    from zang_amd import modules as mod, zang, workloads
    from zang_amd.spans import SpanTable
    nbuf = 4
    _, color, _, _ = workloads.voice_params(4, 0, V)
    bufs = _random_tables(V, nbuf, 1)
    L = oracle.lib()
    st = [oracle.NiceInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_nice_init(C.byref(st[v]), float(color[v]))
    m = mod.NiceInstrument(V, util.dev(color), ctx)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    out0 = util.rng_buffers(2, V, F)
    for b in range(nbuf):
        ref = out0.copy()
        for v in range(V):
            for (s, e, f, on, nic) in bufs[b][v]:
                L.zo_nice_paint(C.byref(st[v]), s, e, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1), int(nic), SR, f, int(on))
        out = util.to_image(out0)
        m.paint_spans(zang.Span(0, F), [out], None, SR, SpanTable(bufs[b], ctx.device))
        zf = ctx.image(F, V, fill=7.0)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"nice spans buffer {b}")
    gs = m.state()
    assert [int(x) for x in gs["osc"]["cnt"]] == [r.osc.cnt for r in st]
    assert [int(x) for x in gs["env"]["state"]] == [r.env.state for r in st]
    util.assert_bitexact(gs["flt"]["l"].astype(np.float32), np.array([r.flt.l for r in st], np.float32), "l")


@pytest.mark.parametrize("V,zero_first", [(128, True), (20, True), (20, False), (64, False), (65, False)])
def test_pmosc_paint_spans(ctx, oracle, V, zero_first):
    """V > 64: lane-per-voice walk (k_pmosc_spans); V <= 64: one wave per voice, lanes = frames
    (k_pmosc_spans_wave).  zero_first on garbage / ADD onto an existing image."""
    from zang_amd import modules as mod, zang
    from zang_amd.spans import SpanTable
    nbuf = 3
    rel = np.random.default_rng(3).uniform(0.05, 0.5, V).astype(np.float32)
    bufs = _random_tables(V, nbuf, 4)
    L = oracle.lib()
    st = [oracle.PMOscInstrument() for _ in range(V)]
    for v in range(V):
        L.zo_pmosc_init(C.byref(st[v]), float(rel[v]))
    m = mod.PMOscInstrument(V, util.dev(rel), ctx)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32); t2 = np.zeros(F, np.float32)
    base = util.rng_buffers(5, V, F)
    for b in range(nbuf):
        ref = np.zeros((V, F), np.float32) if zero_first else base.copy()
        for v in range(V):
            for (s, e, f, on, nic) in bufs[b][v]:
                L.zo_pmosc_paint(C.byref(st[v]), s, e, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(t2), int(nic), SR, f, int(on))
        out = util.to_image(base)                                # garbage: ZERO_FIRST must clear unpainted frames too
        m.paint_spans(zang.Span(0, F), [out], None, SR, SpanTable(bufs[b], ctx.device), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref, f"pmosc spans buffer {b}")
    gs = m.state()
    util.assert_bitexact(gs["carrier"]["t"].astype(np.float32), np.array([r.carrier.t for r in st], np.float32), "carrier t")


def test_sequential_mix_bitexact(ctx):
    import torch
    from zang_amd import zang
    V = 17
    src = util.rng_buffers(6, V, F)
    mix = torch.full((F,), 0.125, dtype=torch.float32, device="cuda")
    zang.mixdownVoices(zang.Span(5, 1000), mix, util.to_image(src), sequential=True, ctx=ctx)
    ctx.sync()
    ref = np.full(F, 0.125, np.float32)
    for v in range(V):                     # out += voice_v, in order (example_song.zig:340-346)
        ref[5:1000] = ref[5:1000] + src[v, 5:1000]
    util.assert_bitexact(mix.cpu().numpy(), ref, "sequential mix")


@pytest.mark.parametrize("fmt", [0, 1])
def test_mix_down_bitexact(ctx, oracle, fmt):
    import torch
    from zang_amd import zang
    n, ch = 3000, 2
    x = np.random.default_rng(7).uniform(-6, 6, n).astype(np.float32)
    x[:6] = [np.nan, 4.0, -4.0, 0.0, 3.99999, -3.99999]
    bps = 2 if fmt == 1 else 1
    ref = np.zeros(n * bps * ch, np.uint8)
    L = oracle.lib()
    fn = L.zo_mixdown_s16lsb if fmt == 1 else L.zo_mixdown_s8
    for c in range(ch):
        fn(ref.ctypes.data_as(C.POINTER(C.c_uint8)), oracle.fptr(x), n, ch, c, 0.25 * (c + 1))
    dst = torch.zeros(n * bps * ch, dtype=torch.uint8, device="cuda")
    gx = util.dev(x)
    for c in range(ch):
        zang.mixDown(dst, gx, fmt, ch, c, 0.25 * (c + 1), ctx=ctx)
    ctx.sync()
    assert np.array_equal(dst.cpu().numpy(), ref)
