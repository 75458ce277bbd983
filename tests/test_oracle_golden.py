"""CPU: the oracle against the committed known answers (tests/golden/known_answers.json).

The reference holds no vector for any paint() result (parity unpinned); these
are the hand / numpy-float32 derivations made from the cited reference lines."""
import ctypes as C
import json
import os

import numpy as np

from tests import util

K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))
f32 = lambda xs: np.array([float(x) for x in xs], np.float32)


def test_xoshiro_seed0(oracle):
    L = oracle.lib()
    st = (C.c_uint64 * 4)(); out = (C.c_uint64 * 6)()
    L.zo_xoshiro_seq(0, st, out, 6)
    assert ["%016x" % x for x in st] == K["xoshiro_seed0_state"]
    assert ["%016x" % x for x in out] == K["xoshiro_seed0_next6"]


def test_noise_white_seeds(oracle):
    L = oracle.lib()
    for seed, key in ((0, "noise_white_seed0"), (1, "noise_white_seed1")):
        n = oracle.Noise(); L.zo_noise_init(C.byref(n), seed)
        b = np.zeros(4, np.float32)
        L.zo_noise_paint(C.byref(n), 0, 4, oracle.fptr(b), oracle.NOISE_WHITE)
        util.assert_bitexact(b, f32(K[key]), key)


def test_pulseosc_known(oracle):
    L = oracle.lib()
    for c in K["pulseosc"]:
        p = oracle.PulseOsc(); L.zo_pulseosc_init(C.byref(p))
        b = np.zeros(c["n"], np.float32)
        L.zo_pulseosc_paint(C.byref(p), 0, c["n"], oracle.fptr(b), c["sample_rate"], oracle.constant(c["freq"]), c["color"])
        util.assert_bitexact(b, f32(c["out"]), "pulseosc KA")
        assert p.cnt == c["cnt_after"]


def test_filter_impulse(oracle):
    L = oracle.lib()
    c = K["filter_lowpass_impulse"]
    f = oracle.Filter(); L.zo_filter_init(C.byref(f))
    inp = f32(c["input"]); out = np.zeros_like(inp)
    L.zo_filter_paint(C.byref(f), 0, len(inp), oracle.fptr(out), oracle.fptr(inp), oracle.FILTER_LOW_PASS,
                      oracle.constant(c["cutoff"]), oracle.constant(c["res"]))
    util.assert_bitexact(out, f32(c["out"]), "filter KA")
    assert np.float32(f.l) == np.float32(c["l"]) and np.float32(f.b) == np.float32(c["b"])


def test_decimator_known(oracle):
    L = oracle.lib()
    for c in K["decimator"]:
        d = oracle.Decimator(); L.zo_decimator_init(C.byref(d))
        inp = f32(c["input"]); out = np.zeros_like(inp)
        L.zo_decimator_paint(C.byref(d), 0, len(inp), oracle.fptr(out), c["sample_rate"], oracle.fptr(inp), c["fake"])
        util.assert_bitexact(out, f32(c["added"]), "decimator KA")
        assert (d.dval, d.dcount) == (c["dval"], c["dcount"])


def test_painter_linear_steps(oracle):
    """Envelope attack with a linear curve from 0 to goal 1: painted value == t (painter.zig:97-114)."""
    L = oracle.lib()
    c = K["painter_linear_t"]
    e = oracle.Envelope(); L.zo_envelope_init(C.byref(e))
    p = oracle.EnvelopeParams(c["sample_rate"], oracle.curve(oracle.CURVE_LINEAR, c["duration"]),
                              oracle.curve(oracle.CURVE_LINEAR, 1.0), oracle.curve(oracle.CURVE_LINEAR, 1.0), 1.0, 1)
    out = np.zeros(3, np.float32)
    L.zo_envelope_paint(C.byref(e), 0, 3, oracle.fptr(out), 1, C.byref(p))
    util.assert_bitexact(out, f32(c["t"]), "painter t")


def test_gate_and_bypass(oracle):
    L = oracle.lib()
    out = np.full(8, 0.25, np.float32)
    L.zo_gate_paint(2, 6, oracle.fptr(out), 1)
    assert out.tolist() == [0.25, 0.25, 1.25, 1.25, 1.25, 1.25, 0.25, 0.25]
    L.zo_gate_paint(0, 8, oracle.fptr(out), 0)
    assert out.tolist() == [0.25, 0.25, 1.25, 1.25, 1.25, 1.25, 0.25, 0.25]
    f = oracle.Filter(); L.zo_filter_init(C.byref(f))
    inp = np.arange(8, dtype=np.float32)
    L.zo_filter_paint(C.byref(f), 0, 8, oracle.fptr(out), oracle.fptr(inp), oracle.FILTER_BYPASS, oracle.constant(0.3), oracle.constant(0.3))
    assert out.tolist() == [0.25, 1.25, 3.25, 4.25, 5.25, 6.25, 6.25, 7.25] and (f.l, f.b) == (0.0, 0.0)


def test_mixdown_s16(oracle):
    L = oracle.lib()
    c = K["mixdown_s16"]
    x = f32(c["input"])
    dst = np.zeros(x.size * 2 * 2, np.uint8)       # 2 channels, write channel 1
    L.zo_mixdown_s16lsb(dst.ctypes.data_as(C.POINTER(C.c_uint8)), oracle.fptr(x), x.size, 2, 1, c["vol"])
    got = dst.view("<i2").reshape(-1, 2)
    assert got[:, 1].tolist() == c["expected_i16"]
    assert not got[:, 0].any()
