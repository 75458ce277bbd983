"""GPU: the single-voice host-pointer wrappers (zh_*_paint_host) -- zang's literal one-voice call
shape with host []f32 slices -- against the oracle, all ten north-star modules."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
N = 600
FP = C.POINTER(C.c_float)


def _outs(a):
    arr = (FP * 1)(a.ctypes.data_as(FP))
    return arr


def _hcob(abi, kind, c, buf):
    return abi.HCob(0, float(c), None) if kind == "c" else abi.HCob(1, 0.0, buf.ctypes.data_as(FP))


def test_all_modules_host_paint(ctx, oracle):
    from zang_amd import abi
    lib, h = ctx.lib, ctx.handle
    L = oracle.lib()
    rng = np.random.default_rng(123)
    fbuf = rng.uniform(50, 5000, N).astype(np.float32); pbuf = rng.uniform(-1, 1, N).astype(np.float32)
    inp = rng.uniform(-1, 1, N).astype(np.float32)
    spans = [(0, 150), (150, 600)]

    def both(name, gpu_call, oracle_call, state_g, state_o, state_eq):
        out_g = rng.uniform(-1, 1, N).astype(np.float32); out_o = out_g.copy()
        for k, (s, e) in enumerate(spans):
            abi.check(gpu_call(s, e, out_g, k), name)
            oracle_call(s, e, out_o, k)
        util.assert_bitexact(out_g, out_o, name)
        if state_eq:
            assert state_eq(state_g, state_o), name

    # SineOsc (freq buffer, phase const)
    sg = abi.SineOscState(0.0); so = oracle.SineOsc(0.0)
    p = abi.SineOscHostParams(SR, _hcob(abi, "b", 0, fbuf), _hcob(abi, "c", 0.25, None))
    both("sineosc", lambda s, e, o, k: lib.zh_sineosc_paint_host(h, C.byref(sg), s, e, _outs(o), None, 0, C.byref(p)),
         lambda s, e, o, k: L.zo_sineosc_paint(C.byref(so), s, e, oracle.fptr(o), SR, oracle.buffer(fbuf), oracle.constant(0.25)),
         sg, so, lambda a, b: np.float32(a.t) == np.float32(b.t))
    # PulseOsc / TriSawOsc (const freq)
    for name, fn, ofn, ost in (("pulseosc", lib.zh_pulseosc_paint_host, L.zo_pulseosc_paint, oracle.PulseOsc()),
                               ("trisawosc", lib.zh_trisawosc_paint_host, L.zo_trisawosc_paint, oracle.TriSawOsc())):
        gst = abi.PulseOscState(0) if name == "pulseosc" else abi.TriSawOscState(0, 0.0)
        pp = abi.PulseOscHostParams(SR, _hcob(abi, "c", 441.0, None), 0.37)
        both(name, lambda s, e, o, k, fn=fn, gst=gst, pp=pp: fn(h, C.byref(gst), s, e, _outs(o), None, 0, C.byref(pp)),
             lambda s, e, o, k, ofn=ofn, ost=ost: ofn(C.byref(ost), s, e, oracle.fptr(o), SR, oracle.constant(441.0), 0.37),
             gst, ost, lambda a, b: a.cnt == b.cnt)
    # Noise (pink), seeded by zh_noise_state_init
    ng = abi.NoiseState(); abi.check(lib.zh_noise_state_init(C.byref(ng), 77), "seed")
    no = oracle.Noise(); L.zo_noise_init(C.byref(no), 77)
    assert list(ng.r) == list(no.r)
    npar = abi.NoiseHostParams(1)
    both("noise", lambda s, e, o, k: lib.zh_noise_paint_host(h, C.byref(ng), s, e, _outs(o), None, 0, C.byref(npar)),
         lambda s, e, o, k: L.zo_noise_paint(C.byref(no), s, e, oracle.fptr(o), 1), ng, no, lambda a, b: list(a.r) == list(b.r))
    # Envelope: on (new note) then off
    eg = abi.EnvelopeState(0, 0, 0, 0); eo = oracle.Envelope(); L.zo_envelope_init(C.byref(eo))
    def env_g(s, e, o, k):
        ep = abi.EnvelopeHostParams(SR, abi.HCurve(3, 0.001), abi.HCurve(2, 0.002), abi.HCurve(1, 0.004), 0.6, 1 - k)
        return lib.zh_envelope_paint_host(h, C.byref(eg), s, e, _outs(o), None, 1 - k, C.byref(ep))
    def env_o(s, e, o, k):
        ep = oracle.EnvelopeParams(SR, oracle.curve(3, 0.001), oracle.curve(2, 0.002), oracle.curve(1, 0.004), 0.6, 1 - k)
        L.zo_envelope_paint(C.byref(eo), s, e, oracle.fptr(o), 1 - k, C.byref(ep))
    both("envelope", env_g, env_o, eg, eo, lambda a, b: (a.state, np.float32(a.t), np.float32(a.last_value)) == (b.state, np.float32(b.painter.t), np.float32(b.painter.last_value)))
    # Gate
    gp = abi.GateHostParams(1)
    both("gate", lambda s, e, o, k: lib.zh_gate_paint_host(h, None, s, e, _outs(o), None, 0, C.byref(gp)),
         lambda s, e, o, k: L.zo_gate_paint(s, e, oracle.fptr(o), 1), None, None, None)
    # Filter (band-pass, cutoff buffer)
    cbuf = rng.uniform(0, 1, N).astype(np.float32)
    fg = abi.FilterState(0, 0); fo = oracle.Filter(); L.zo_filter_init(C.byref(fo))
    fpar = abi.FilterHostParams(inp.ctypes.data_as(FP), 2, _hcob(abi, "b", 0, cbuf), _hcob(abi, "c", 0.3, None))
    both("filter", lambda s, e, o, k: lib.zh_filter_paint_host(h, C.byref(fg), s, e, _outs(o), None, 0, C.byref(fpar)),
         lambda s, e, o, k: L.zo_filter_paint(C.byref(fo), s, e, oracle.fptr(o), oracle.fptr(inp), 2, oracle.buffer(cbuf), oracle.constant(0.3)),
         fg, fo, lambda a, b: (np.float32(a.l), np.float32(a.b)) == (np.float32(b.l), np.float32(b.b)))
    # Sampler (s16, resampling, loop)
    data = rng.integers(0, 256, 800, dtype=np.uint8)
    sg2 = abi.SamplerState(0.0); so2 = oracle.Sampler(); L.zo_sampler_init(C.byref(so2))
    sp = abi.SamplerHostParams(30000.0, 1, 44100, 1, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, 0, 1)
    op = oracle.SamplerParams(30000.0, 1, 44100, 1, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, 0, 1)
    both("sampler", lambda s, e, o, k: lib.zh_sampler_paint_host(h, C.byref(sg2), s, e, _outs(o), None, 0, C.byref(sp)),
         lambda s, e, o, k: L.zo_sampler_paint(C.byref(so2), s, e, oracle.fptr(o), 0, C.byref(op)), sg2, so2,
         lambda a, b: np.float32(a.t) == np.float32(b.t))
    # Decimator
    dg = abi.DecimatorState(); abi.check(lib.zh_decimator_state_init(C.byref(dg)), "init")
    do = oracle.Decimator(); L.zo_decimator_init(C.byref(do))
    dp = abi.DecimatorHostParams(SR, inp.ctypes.data_as(FP), 7000.0)
    both("decimator", lambda s, e, o, k: lib.zh_decimator_paint_host(h, C.byref(dg), s, e, _outs(o), None, 0, C.byref(dp)),
         lambda s, e, o, k: L.zo_decimator_paint(C.byref(do), s, e, oracle.fptr(o), SR, oracle.fptr(inp), 7000.0), dg, do,
         lambda a, b: (np.float32(a.dval), np.float32(a.dcount)) == (np.float32(b.dval), np.float32(b.dcount)))
    # Distortion (overdrive)
    xp = abi.DistortionHostParams(inp.ctypes.data_as(FP), 0, 0.6, 0.8, 0.1)
    both("distortion", lambda s, e, o, k: lib.zh_distortion_paint_host(h, None, s, e, _outs(o), None, 0, C.byref(xp)),
         lambda s, e, o, k: L.zo_distortion_paint(s, e, oracle.fptr(o), oracle.fptr(inp), 0, 0.6, 0.8, 0.1), None, None, None)
