"""GPU parity: Delay(n) modules.  The oracle runs the reference's CHUNKED read/write form
(delay.zig:28-89, examples/modules.zig:363-460); the device walks sample by sample -- equal bits
prove the two formulations equivalent, including delays shorter than the span (several chunks per
paint, a slot re-read within one paint) and ring wrap-around."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
F = 1024
SPANS = [(0, 200), (200, 777), (777, 1024), (0, 1024), (0, 1024)]


@pytest.mark.parametrize("form", ["frames", "walk"])
@pytest.mark.parametrize("zero_first", [False, True])
@pytest.mark.parametrize("D", [1, 7, 8, 9, 100, 300, 577, 1024, 3000])     # < 8: frame by frame; >= 8: chunks of 8 frames
def test_simple_delay(ctx, oracle, D, zero_first, form, monkeypatch):
    """(Few voices, delay >= 8, spans >= 64 frames: the frames of a span are painted independently, k_delay_frames +
    k_delay_store + k_delay_advance -- delays shorter than, equal to and longer than the span; ZH_DELAY_FRAMES_MAX=0 is the
    per-voice walk.)"""
    from zang_amd import modules as mod, zang
    if form == "walk":
        util.set_form(monkeypatch, delay_frames_max="0")
    V = 96
    inp = [util.rng_buffers(10 + k, V, F) for k in range(len(SPANS))]
    out0 = util.rng_buffers(3, V, F)
    L = oracle.lib()
    ref = [out0.copy() for _ in SPANS]
    rings = np.zeros((V, D), np.float32); ridx = []
    for v in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[v]), D)
        for k, (s, e) in enumerate(SPANS):
            if zero_first:
                ref[k][v][s:e] = 0.0
            L.zo_simple_delay_paint(C.byref(d), s, e, oracle.fptr(ref[k][v]), oracle.fptr(inp[k][v]))
        ridx.append(d.index)
    m = mod.SimpleDelay(V, D, ctx)
    for k, (s, e) in enumerate(SPANS):
        out = util.to_image(out0)
        m.paint(zang.Span(s, e), [out], [], False, m.Params(util.to_image(inp[k])), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref[k], f"simple delay D={D} paint {k}")
    grings, gidx = m.state()
    util.assert_bitexact(grings, rings, "ring"); assert [int(x) for x in gidx] == ridx
    m.reset()
    grings, gidx = m.state()
    assert not grings.any() and not gidx.any()


@pytest.mark.parametrize("D", [4, 64])
def test_simple_delay_in_place(ctx, oracle, D):
    """The input image IS the output image: the order of reads and writes inside a frame becomes observable (the delay
    line is fed the sample it has just added to), so the device must not hoist a chunk's input reads above its output
    writes -- it falls back to the frame-by-frame form.  The oracle runs the reference's chunked form on one array."""
    from zang_amd import modules as mod, zang
    V = 70
    buf0 = util.rng_buffers(31, V, F)
    L = oracle.lib()
    ref = buf0.copy()
    rings = np.zeros((V, D), np.float32)
    for v in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[v]), D)
        for (s, e) in SPANS[:3]:
            L.zo_simple_delay_paint(C.byref(d), s, e, oracle.fptr(ref[v]), oracle.fptr(ref[v]))
    m = mod.SimpleDelay(V, D, ctx)
    img = util.to_image(buf0)
    for (s, e) in SPANS[:3]:
        m.paint(zang.Span(s, e), [img], [], False, m.Params(img))
    ctx.sync()
    util.assert_bitexact(util.from_image(img), ref, f"in-place simple delay D={D}")
    util.assert_bitexact(m.state()[0], rings, "ring")


@pytest.mark.parametrize("form", ["pipeline", "walk"])
@pytest.mark.parametrize("zero_first", [False, True])
@pytest.mark.parametrize("D", [5, 8, 192, 200, 333, 1024, 2000])
def test_filtered_echoes(ctx, oracle, D, zero_first, form, monkeypatch):
    """(Few voices, delay >= 192: three waves per 64 voices -- loader, filter recurrence, writer -- k_filtered_echoes_pc;
    ZH_ECHOES_PC_MAX=0 is the one-wave walk.)"""
    from zang_amd import modules as mod, zang
    if form == "walk":
        util.set_form(monkeypatch, echoes_pc_max="0")
    V = 96
    rng = np.random.default_rng(4)
    fb = rng.uniform(0.1, 0.9, V).astype(np.float32); cutoff = rng.uniform(0.05, 1.0, V).astype(np.float32)
    inp = [util.rng_buffers(20 + k, V, F) for k in range(len(SPANS))]
    out0 = util.rng_buffers(5, V, F)
    L = oracle.lib()
    ref = [out0.copy() for _ in SPANS]
    rings = np.zeros((V, D), np.float32); rst = []
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for v in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[v]), D)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for k, (s, e) in enumerate(SPANS):
            if zero_first:
                ref[k][v][s:e] = 0.0
            L.zo_filtered_echoes_paint(C.byref(d), C.byref(fl), s, e, oracle.fptr(ref[k][v]), oracle.fptr(t0), oracle.fptr(t1),
                                       oracle.fptr(inp[k][v]), float(fb[v]), float(cutoff[v]))
        rst.append((d.index, fl.l, fl.b))
    m = mod.FilteredEchoes(V, D, ctx)
    gfb, gc = util.dev(fb), util.dev(cutoff)
    for k, (s, e) in enumerate(SPANS):
        out = util.to_image(out0)
        m.paint(zang.Span(s, e), [out], None, False, m.Params(util.to_image(inp[k]), gfb, gc), zero_first=zero_first)
        ctx.sync()
        util.assert_bitexact(util.from_image(out), ref[k], f"filtered echoes D={D} paint {k}")
    grings, gidx, gflt = m.state()
    util.assert_bitexact(grings, rings, "ring"); assert [int(x) for x in gidx] == [r[0] for r in rst]
    util.assert_bitexact(gflt["l"].astype(np.float32), np.array([r[1] for r in rst], np.float32), "l")


@pytest.mark.parametrize("D", [200, 333])
def test_delays_with_per_voice_ring_indices(ctx, oracle, D):
    """set_state may give every voice its own ring index and ring content: the frame-parallel SimpleDelay and the three-wave
    FilteredEchoes then take their per-lane slot arithmetic (no wave-uniform row addressing), wraps at different frames per
    voice included."""
    from zang_amd import abi, modules as mod, zang
    V = 130
    rng = np.random.default_rng(31)
    idx = rng.integers(0, D, V).astype(np.uint32)
    rings0 = rng.uniform(-1, 1, (V, D)).astype(np.float32)
    inp = util.rng_buffers(40, V, F); out0 = util.rng_buffers(41, V, F)
    fb = rng.uniform(0.1, 0.9, V).astype(np.float32); cutoff = rng.uniform(0.05, 1.0, V).astype(np.float32)
    L = oracle.lib()
    spans = [(0, 1024), (100, 612), (612, 1000)]
    # SimpleDelay
    ref = out0.copy(); rings = rings0.copy(); ridx = []
    for v in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[v]), D)
        rings[v] = rings0[v]; d.index = int(idx[v])
        for (s, e) in spans:
            L.zo_simple_delay_paint(C.byref(d), s, e, oracle.fptr(ref[v]), oracle.fptr(inp[v]))
        ridx.append(d.index)
    m = mod.SimpleDelay(V, D, ctx)
    abi.check(ctx.lib.zh_delay_set_state(m.handle, rings0.ctypes.data, idx.ctypes.data), "set_state")
    out = util.to_image(out0); gi = util.to_image(inp)
    for (s, e) in spans:
        m.paint(zang.Span(s, e), [out], [], False, m.Params(gi))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, "simple delay, per-voice indices")
    grings, gidx = m.state()
    util.assert_bitexact(grings, rings, "ring"); assert [int(x) for x in gidx] == ridx
    # FilteredEchoes
    ref = out0.copy(); rings = rings0.copy(); rst = []
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    for v in range(V):
        d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[v]), D)
        rings[v] = rings0[v]; d.index = int(idx[v])
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for (s, e) in spans:
            L.zo_filtered_echoes_paint(C.byref(d), C.byref(fl), s, e, oracle.fptr(ref[v]), oracle.fptr(t0), oracle.fptr(t1),
                                       oracle.fptr(inp[v]), float(fb[v]), float(cutoff[v]))
        rst.append((d.index, fl.l, fl.b))
    m = mod.FilteredEchoes(V, D, ctx)
    flt = np.zeros(V, dtype=np.dtype(abi.FilterState))
    abi.check(ctx.lib.zh_filtered_echoes_set_state(m.handle, rings0.ctypes.data, idx.ctypes.data, flt.ctypes.data), "set_state")
    out = util.to_image(out0)
    for (s, e) in spans:
        m.paint(zang.Span(s, e), [out], None, False, m.Params(gi, util.dev(fb), util.dev(cutoff)))
    ctx.sync()
    util.assert_bitexact(util.from_image(out), ref, "filtered echoes, per-voice indices")
    grings, gidx, gflt = m.state()
    util.assert_bitexact(grings, rings, "ring"); assert [int(x) for x in gidx] == [r[0] for r in rst]
    util.assert_bitexact(gflt["l"].astype(np.float32), np.array([r[1] for r in rst], np.float32), "l")


def test_stereo_echoes_composition(ctx, oracle):
    """StereoEchoes (examples/modules.zig:463-525) as a host-level composition of C-ABI calls:
    addInto x2, SimpleDelay, FilteredEchoes, addInto, SimpleDelay -- vs the same composition of oracle calls."""
    from zang_amd import modules as mod, zang
    V, MAIN = 64, 1500
    HALF = MAIN // 2
    rng = np.random.default_rng(8)
    inp = [util.rng_buffers(30 + k, V, F) for k in range(3)]
    L = oracle.lib()
    refL = [np.zeros((V, F), np.float32) for _ in range(3)]; refR = [np.zeros((V, F), np.float32) for _ in range(3)]
    t0, t1, t2, t3 = (np.zeros(F, np.float32) for _ in range(4))
    r0 = np.zeros((V, HALF), np.float32); r1 = np.zeros((V, HALF), np.float32); r2 = np.zeros((V, MAIN), np.float32)
    for v in range(V):
        d0 = oracle.Delay(); L.zo_delay_init(C.byref(d0), oracle.fptr(r0[v]), HALF)
        d1 = oracle.Delay(); L.zo_delay_init(C.byref(d1), oracle.fptr(r1[v]), HALF)
        de = oracle.Delay(); L.zo_delay_init(C.byref(de), oracle.fptr(r2[v]), MAIN)
        fl = oracle.Filter(); L.zo_filter_init(C.byref(fl))
        for k in range(3):
            L.zo_add_into(0, F, oracle.fptr(refL[k][v]), oracle.fptr(inp[k][v])); L.zo_add_into(0, F, oracle.fptr(refR[k][v]), oracle.fptr(inp[k][v]))
            L.zo_zero(0, F, oracle.fptr(t0)); L.zo_simple_delay_paint(C.byref(d0), 0, F, oracle.fptr(t0), oracle.fptr(inp[k][v]))
            L.zo_zero(0, F, oracle.fptr(t1))
            L.zo_filtered_echoes_paint(C.byref(de), C.byref(fl), 0, F, oracle.fptr(t1), oracle.fptr(t2), oracle.fptr(t3), oracle.fptr(t0), 0.6, 0.1)
            L.zo_add_into(0, F, oracle.fptr(refL[k][v]), oracle.fptr(t1))
            L.zo_simple_delay_paint(C.byref(d1), 0, F, oracle.fptr(refR[k][v]), oracle.fptr(t1))
    delay0, delay1, echoes = mod.SimpleDelay(V, HALF, ctx), mod.SimpleDelay(V, HALF, ctx), mod.FilteredEchoes(V, MAIN, ctx)
    sp = zang.Span(0, F)
    g0, g1 = ctx.image(F, V), ctx.image(F, V)
    for k in range(3):
        gin = util.to_image(inp[k]); outL = ctx.image(F, V, fill=0.0); outR = ctx.image(F, V, fill=0.0)
        zang.addInto(sp, outL, gin, ctx=ctx); zang.addInto(sp, outR, gin, ctx=ctx)
        delay0.paint(sp, [g0], [], False, delay0.Params(gin), zero_first=True)
        echoes.paint(sp, [g1], None, False, echoes.Params(g0, 0.6, 0.1), zero_first=True)
        zang.addInto(sp, outL, g1, ctx=ctx)
        delay1.paint(sp, [outR], [], False, delay1.Params(g1))
        ctx.sync()
        util.assert_bitexact(util.from_image(outL), refL[k], f"stereo echoes L {k}")
        util.assert_bitexact(util.from_image(outR), refR[k], f"stereo echoes R {k}")
