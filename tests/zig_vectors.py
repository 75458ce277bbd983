"""Vector files written by tools/zig_oracle/dump_vectors.zig (the REFERENCE's own paint() code run under Zig) and the
table that says what each case computes, restated with the oracle.  Format: "ZGV1", then records until EOF --
u32 name_len, name, u32 type (0 f32, 1 u32, 2 u64, 3 u8), u32 count, data (little endian).

`expected(name, rec)` recomputes a case with the ORACLE from the inputs stored in the file and returns the records the
file should hold; `compare(name, rec)` checks them.  The bar (BASELINE.json north_star): bit-exact wherever the
arithmetic is integer / plain f32 add-mul-compare, 1e-5 relative (floor 1e-3) where Zig's std math (sin, cos, atan,
pow) is involved -- and the bit-exact fraction is reported for those too.

`write_oracle_vectors(dir)` writes the same files from the oracle with the dumper's input generator: NOT reference
output -- it exists so that the reader, the table and the comparison are exercised on a machine without Zig."""
import ctypes as C
import os
import struct

import numpy as np

F = 1024
SR = 48000.0
DTYPES = {0: np.float32, 1: np.uint32, 2: np.uint64, 3: np.uint8}
CODES = {np.dtype(v): k for k, v in DTYPES.items()}
SPANS3 = [(0, 200), (200, 777), (777, 1024)]
NOTE_SCRIPT = [(0, 200, True, True), (200, 777, True, False), (777, 1024, False, False), (1024, 2048, False, False)]
RETRIGGER_SCRIPT = [(0, 300, True, True), (300, 600, True, True), (600, 1024, False, False), (1024, 2048, True, True)]
SCRIPTS = [NOTE_SCRIPT, RETRIGGER_SCRIPT]
# records compared with the 1e-5 tolerance instead of bit for bit: (case prefix, record name or None = every float record)
LIBM = [("sineosc_", None), ("pmosc_", None), ("distortion_overdrive", None), ("distortion_clip", None), ("math", None),
        ("filter_cutoff_from_frequency", None), ("nice_", None), ("fsaw_", None)]


def read(path):
    data = open(path, "rb").read()
    assert data[:4] == b"ZGV1", path
    rec, off = {}, 4
    while off < len(data):
        (n,) = struct.unpack_from("<I", data, off); off += 4
        name = data[off:off + n].decode(); off += n
        code, count = struct.unpack_from("<II", data, off); off += 8
        dt = np.dtype(DTYPES[code]).newbyteorder("<")
        rec[name] = np.frombuffer(data, dt, count, off).astype(DTYPES[code]); off += count * dt.itemsize
    return rec


def write(path, rec):
    with open(path, "wb") as f:
        f.write(b"ZGV1")
        for name, a in rec.items():
            a = np.ascontiguousarray(a)
            f.write(struct.pack("<I", len(name)) + name.encode() + struct.pack("<II", CODES[a.dtype], a.size) + a.astype(a.dtype.newbyteorder("<")).tobytes())


def fill(n, seed, lo, hi):
    """dump_vectors.zig `fill`: SplitMix64 -> top 24 bits / 2^24 -> lo + (hi - lo) * u, all in f32."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) / np.float32(16777216.0)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)


def splitmix_bytes(n, seed):
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z & np.uint64(0xFF)).astype(np.uint8)


def _po():
    from oracle import pyoracle as po
    return po, po.lib()


def _cob(po, is_buf, c, buf):
    return po.buffer(buf) if is_buf else po.constant(c)


def _curve(po, kind, dur):
    return po.curve(int(kind), dur if int(kind) else 0.0)


def _f32(*xs):
    return np.array(xs, np.float32)


# ------------------------------------------------------------------ the case table
def expected(name, rec):
    """-> {record name: array} the file must hold, computed by the oracle from the file's own inputs."""
    po, L = _po()
    out = rec["out0"].copy() if "out0" in rec else None
    exp = {}
    if name.startswith("sineosc_"):
        k = ["cc", "cb", "bc", "bb"].index(name.split("_")[1])
        m = po.SineOsc(); L.zo_sineosc_init(C.byref(m))
        for s, e in SPANS3:
            L.zo_sineosc_paint(C.byref(m), s, e, po.fptr(out), SR, _cob(po, k & 2, 440.0, rec["freq"]), _cob(po, k & 1, 0.25, rec["phase"]))
        exp.update(out=out, state_t=_f32(m.t))
    elif name.startswith(("pulseosc_", "trisawosc_")):
        pulse = name.startswith("pulseosc_")
        m = po.PulseOsc() if pulse else po.TriSawOsc()
        (L.zo_pulseosc_init if pulse else L.zo_trisawosc_init)(C.byref(m))
        fn = L.zo_pulseosc_paint if pulse else L.zo_trisawosc_paint
        if "_buf_" in name:
            sr, color = (float(x) for x in rec["params"])
            freq = po.buffer(rec["freq"])
        else:
            sr, f, color = (float(x) for x in rec["params"])
            freq = po.constant(f)
        for s, e in SPANS3:
            fn(C.byref(m), s, e, po.fptr(out), sr, freq, color)
        exp.update(out=out, state_cnt=np.array([m.cnt], np.uint32))
        if not pulse:
            exp["state_t"] = _f32(m.t)
    elif name.startswith("noise_"):
        color = po.NOISE_WHITE if "white" in name else po.NOISE_PINK
        m = po.Noise(); L.zo_noise_init(C.byref(m), int(name[-1]))
        a, b = out[:F].copy(), out[F:].copy()
        for s, e in SPANS3:
            L.zo_noise_paint(C.byref(m), s, e, po.fptr(a), color)
        L.zo_noise_paint(C.byref(m), 0, F, po.fptr(b), color)
        exp.update(out=np.concatenate([a, b]), state_s=np.array(list(m.r), np.uint64))
    elif name.startswith("envelope_"):
        p = rec["params"]
        script = SCRIPTS[int(name.split("_")[3])]
        m = po.Envelope(); L.zo_envelope_init(C.byref(m))
        halves = [out[:F].copy(), out[F:].copy()]
        for s, e, on, nic in script:
            h = 1 if s >= F else 0
            pr = po.EnvelopeParams(float(p[0]), _curve(po, p[1], float(p[2])), _curve(po, p[3], float(p[4])), _curve(po, p[5], float(p[6])), float(p[7]), int(on))
            L.zo_envelope_paint(C.byref(m), s - h * F, e - h * F, po.fptr(halves[h]), int(nic), C.byref(pr))
        exp.update(out=np.concatenate(halves), state_stage=np.array([m.state], np.uint32),
                   state_painter=_f32(m.painter.t, m.painter.last_value, m.painter.start))
    elif name == "gate":
        for (s, e), on in zip(SPANS3, (1, 0, 1)):
            L.zo_gate_paint(s, e, po.fptr(out), on)
        exp.update(out=out)
    elif name == "filter_cutoff_from_frequency":
        exp["out"] = np.array([L.zo_filter_cutoff_from_frequency(float(f), SR) for f in rec["freq"]], np.float32)
    elif name.startswith("filter_"):
        p = int(name.rsplit("_", 1)[1])
        m = po.Filter(); L.zo_filter_init(C.byref(m))
        for s, e in SPANS3:
            L.zo_filter_paint(C.byref(m), s, e, po.fptr(out), po.fptr(rec["input"]), int(rec["type"][0]),
                              _cob(po, p & 2, 0.3, rec["cutoff"]), _cob(po, p & 1, 0.5, rec["res"]))
        exp.update(out=out, state_lb=_f32(m.l, m.b))
    elif name.startswith("decimator_"):
        sr, fake = (float(x) for x in rec["params"])
        m = po.Decimator(); L.zo_decimator_init(C.byref(m))
        for s, e in SPANS3:
            L.zo_decimator_paint(C.byref(m), s, e, po.fptr(out), sr, po.fptr(rec["input"]), fake)
        exp.update(out=out, state=_f32(m.dval, m.dcount))
    elif name.startswith("distortion_"):
        ing, outg, offs = (float(x) for x in rec["params"])
        for s, e in SPANS3:
            L.zo_distortion_paint(s, e, po.fptr(out), po.fptr(rec["input"]), int(rec["type"][0]), ing, outg, offs)
        exp.update(out=out)
    elif name.startswith("sampler_"):
        nch, rate, fmt, channel, loop = (int(x) for x in rec["params"])
        pcm = np.ascontiguousarray(rec["pcm"])
        m = po.Sampler(); L.zo_sampler_init(C.byref(m))
        pr = po.SamplerParams(SR, nch, rate, fmt, pcm.ctypes.data_as(C.POINTER(C.c_uint8)), pcm.size, channel, loop)
        for i, (s, e) in enumerate(SPANS3):
            L.zo_sampler_paint(C.byref(m), s, e, po.fptr(out), int(i == 0), C.byref(pr))
        exp.update(out=out, state_t=_f32(m.t))
    elif name.startswith(("nice_", "pmosc_")):
        script = SCRIPTS[int(name.split("_")[1])]
        sr, freq, arg = (float(x) for x in rec["params"])
        t = [np.zeros(F, np.float32) for _ in range(3)]
        halves = [out[:F].copy(), out[F:].copy()]
        if name.startswith("nice_"):
            m = po.NiceInstrument(); L.zo_nice_init(C.byref(m), arg)
            for s, e, on, nic in script:
                h = 1 if s >= F else 0
                L.zo_nice_paint(C.byref(m), s - h * F, e - h * F, po.fptr(halves[h]), po.fptr(t[0]), po.fptr(t[1]), int(nic), sr, freq, int(on))
            exp.update(state_cnt=np.array([m.osc.cnt], np.uint32), state_lb=_f32(m.flt.l, m.flt.b))
        else:
            m = po.PMOscInstrument(); L.zo_pmosc_init(C.byref(m), arg)
            for s, e, on, nic in script:
                h = 1 if s >= F else 0
                L.zo_pmosc_paint(C.byref(m), s - h * F, e - h * F, po.fptr(halves[h]), po.fptr(t[0]), po.fptr(t[1]), po.fptr(t[2]), int(nic), sr, freq, int(on))
            exp.update(state_t=_f32(m.carrier.t, m.modulator.t))
        exp.update(out=np.concatenate(halves), state_stage=np.array([m.env.state], np.uint32),
                   state_painter=_f32(m.env.painter.t, m.env.painter.last_value, m.env.painter.start))
    elif name.startswith(("fsaw_", "hsq_")):
        script = SCRIPTS[int(name.split("_")[1])]
        sr, freq, _ = (float(x) for x in rec["params"])
        t = [np.zeros(F, np.float32) for _ in range(3)]
        halves = [out[:F].copy(), out[F:].copy()]
        if name.startswith("fsaw_"):
            m = po.FilteredSawtooth(); L.zo_filtered_sawtooth_init(C.byref(m))
            for s, e, on, nic in script:
                h = 1 if s >= F else 0
                L.zo_filtered_sawtooth_paint(C.byref(m), s - h * F, e - h * F, po.fptr(halves[h]), po.fptr(t[0]), po.fptr(t[1]), po.fptr(t[2]), int(nic), sr,
                                             po.constant(freq), int(on))
            exp.update(out=np.concatenate(halves), state_cnt=np.array([m.osc.cnt], np.uint32), state_lb=_f32(m.flt.l, m.flt.b),
                       state_stage=np.array([m.env.state], np.uint32), state_painter=_f32(m.env.painter.t, m.env.painter.last_value, m.env.painter.start))
        else:
            m = po.HardSquare(); L.zo_hard_square_init(C.byref(m))
            for s, e, on, nic in script:
                h = 1 if s >= F else 0
                L.zo_hard_square_paint(C.byref(m), s - h * F, e - h * F, po.fptr(halves[h]), po.fptr(t[0]), po.fptr(t[1]), int(nic), sr, freq, int(on))
            exp.update(out=np.concatenate(halves), state_cnt=np.array([m.osc.cnt], np.uint32))
    elif name == "basics":
        a, b, d = rec["a"], rec["b"], rec["dest0"]
        s, e = 100, 900
        def run(fn, *args):
            x = d.copy(); fn(s, e, po.fptr(x), *args); return x
        exp.update(multiply=run(L.zo_multiply, po.fptr(a), po.fptr(b)), add=run(L.zo_add, po.fptr(a), po.fptr(b)),
                   addScalar=run(L.zo_add_scalar, po.fptr(a), 0.37), multiplyScalar=run(L.zo_multiply_scalar, po.fptr(a), 0.37),
                   multiplyWith=run(L.zo_multiply_with, po.fptr(a)), multiplyWithScalar=run(L.zo_multiply_with_scalar, 0.37),
                   addInto=run(L.zo_add_into, po.fptr(a)))
    elif name == "mixdown":
        mix = rec["mix"]
        s16 = np.zeros(F * 4, np.uint8); s8 = np.zeros(F, np.uint8)
        L.zo_mixdown_s16lsb(s16.ctypes.data_as(C.POINTER(C.c_uint8)), po.fptr(mix), F, 2, 1, 0.25)
        L.zo_mixdown_s8(s8.ctypes.data_as(C.POINTER(C.c_uint8)), po.fptr(mix), F, 1, 0, 0.25)
        exp.update(s16_2ch_ch1=s16, s8_1ch=s8)
    elif name == "math":
        x = rec["sin_x"]
        y = np.zeros_like(x)
        L.zo_math_sinf_n(po.fptr(x), po.fptr(y), x.size); exp["sin"] = y.copy()
        x2 = (x * np.float32(0.08)).astype(np.float32)
        L.zo_math_cosf_n(po.fptr(x2), po.fptr(y), x.size); exp["cos_of_0p08x"] = y.copy()
        L.zo_math_atanf_n(po.fptr(x), po.fptr(y), x.size); exp["atan"] = y.copy()
        x3 = (x * np.float32(0.2)).astype(np.float32)
        L.zo_math_pow2f_n(po.fptr(x3), po.fptr(y), x.size); exp["pow2_of_0p2x"] = y.copy()
        w = np.zeros(8 * 512, np.float32)
        for k in range(8):
            n = po.Noise(); L.zo_noise_init(C.byref(n), 100 + k)
            L.zo_noise_paint(C.byref(n), 0, 512, po.fptr(w[k * 512:(k + 1) * 512]), po.NOISE_WHITE)
        exp["white_seeds100to107_x512"] = w
    elif name == "math2":
        x = rec["sin_pio2_x"]
        y = np.zeros_like(x)
        L.zo_math_sinf_n(po.fptr(x), po.fptr(y), x.size); exp["sin_pio2"] = y.copy()
        L.zo_math_cosf_n(po.fptr(x), po.fptr(y), x.size); exp["cos_pio2"] = y.copy()
        py = rec["pow2_y"]
        y = np.zeros_like(py)
        L.zo_math_pow2f_n(po.fptr(py), po.fptr(y), py.size); exp["pow2"] = y.copy()
        st = rec["rare_states"]
        w = np.zeros(st.size, np.float32)
        for c in range(st.size // 4):
            n = po.Noise()
            for q in range(4):
                n.r[q] = int(st[c * 4 + q])
            L.zo_noise_paint(C.byref(n), 0, 4, po.fptr(w[c * 4:c * 4 + 4]), po.NOISE_WHITE)
        exp["rare_white"] = w
    else:
        raise KeyError("no case named %r in tests/zig_vectors.py" % name)
    return exp


def _is_libm(name, record):
    return record not in ("white_seeds100to107_x512", "rare_white") and any(name.startswith(p) and (r is None or r == record) for p, r in LIBM)


def compare(name, rec, rtol=1e-5, floor=1e-3):
    """-> {record: fraction of bit-identical elements}; raises AssertionError on a mismatch beyond the bar."""
    exp = expected(name, rec)
    report = {}
    for key, want in exp.items():
        assert key in rec, f"{name}: the file has no record {key!r}"
        got = rec[key]
        assert got.shape == want.shape and got.dtype == want.dtype, (name, key, got.shape, want.shape, got.dtype, want.dtype)
        same = (got.view(np.uint32) == want.view(np.uint32)) if got.dtype == np.float32 else (got == want)
        # NaNs produced by the same operation may differ in payload between implementations: equal if both NaN
        if got.dtype == np.float32:
            same = same | (np.isnan(got) & np.isnan(want))
        report[key] = float(same.mean()) if same.size else 1.0
        if same.all():
            continue
        if got.dtype == np.float32 and _is_libm(name, key):
            g, w = got.astype(np.float64), want.astype(np.float64)
            bad = ~(np.abs(g - w) <= rtol * np.maximum(np.abs(w), floor)) & ~same
            assert not bad.any(), f"{name}.{key}: {int(bad.sum())} of {bad.size} beyond 1e-5 relative, first at {int(np.argmax(bad))}: zig={got[np.argmax(bad)]!r} oracle={want[np.argmax(bad)]!r}"
        else:
            i = int(np.argmax(~same))
            raise AssertionError(f"{name}.{key}: {int((~same).sum())} of {same.size} differ (bit-exact required), first at {i}: zig={got[i]!r} oracle={want[i]!r}")
    return report


# ------------------------------------------------------------------ oracle-made files (machinery check only)
def case_inputs():
    """name -> input records, generated exactly as dump_vectors.zig generates them."""
    cases = {}
    for k, nm in enumerate(["sineosc_cc", "sineosc_cb", "sineosc_bc", "sineosc_bb"]):
        cases[nm] = dict(out0=fill(F, 100 + k, -1, 1), freq=fill(F, 200 + k, 20, 2000), phase=fill(F, 300 + k, -1, 1))
    consts = [("const_c0", 440.0, 0.0), ("const_c03", 440.0, 0.3), ("const_c05", 1234.5, 0.5), ("const_c09", 97.0, 0.9), ("const_c1", 5999.0, 1.0),
              ("const_silent_hi", 6000.5, 0.5), ("const_silent_neg", -1.0, 0.5)]
    for k, (nm, f, c) in enumerate(consts):
        for pre in ("pulseosc_", "trisawosc_"):
            cases[pre + nm] = dict(out0=fill(F, 400 + k, -1, 1), params=_f32(SR, f, c))
    for k, c in enumerate((0.1, 0.5, 0.9)):
        for pre in ("pulseosc_buf", "trisawosc_buf"):
            cases[f"{pre}_{k}"] = dict(out0=fill(F, 500 + k, -1, 1), freq=fill(F, 600 + k, -200, 7000), params=_f32(SR, c))
    for k, nm in enumerate(["noise_white_seed0", "noise_pink_seed1", "noise_white_seed2", "noise_pink_seed3"]):
        cases[nm] = dict(out0=fill(2 * F, 700 + k, -1, 1))
    combos = [(3, 3, 3), (1, 2, 3), (0, 1, 1), (2, 0, 2), (1, 1, 0), (0, 0, 0), (3, 1, 2)]
    for ci, cmb in enumerate(combos):
        for si, sus in enumerate((0.6, 1.0)):
            for ki in range(2):
                cases[f"envelope_{ci}_{si}_{ki}"] = dict(out0=fill(2 * F, 800 + ci * 10 + si * 2 + ki, -1, 1),
                                                         params=_f32(SR, cmb[0], 0.002, cmb[1], 0.004, cmb[2], 0.003, sus))
    cases["gate"] = dict(out0=fill(F, 900, -1, 1))
    k = 0
    for t, tn in enumerate(["bypass", "low_pass", "band_pass", "high_pass", "notch", "all_pass"]):
        for p in range(4 if tn in ("low_pass", "notch") else 1):
            cases[f"filter_{tn}_{p}"] = dict(out0=fill(F, 1000 + k, -1, 1), input=fill(F, 1100 + k, -1, 1), cutoff=fill(F, 1200 + k, -0.1, 1.1),
                                             res=fill(F, 1300 + k, -0.1, 1.1), type=np.array([t], np.uint32))
            k += 1
    cases["filter_cutoff_from_frequency"] = dict(freq=fill(512, 1400, 0, 26000))
    for k, fake in enumerate((24000.0, 6000.0, 11025.0, 48000.0, 96000.0, 0.0, -5.0)):
        cases[f"decimator_{k}"] = dict(out0=fill(F, 1500 + k, -1, 1), input=fill(F, 1600 + k, -1, 1), params=_f32(SR, fake))
    for k, p in enumerate([(0.5, 0.7, 0.1), (0.25, 1.0, 0.0), (0.9, 0.3, -0.4)]):
        for t, tn in enumerate(("overdrive", "clip")):
            cases[f"distortion_{tn}_{k}"] = dict(out0=fill(F, 1700 + k, -1, 1), input=fill(F, 1800 + k, -1.5, 1.5), params=_f32(*p), type=np.array([t], np.uint32))
    pcm = splitmix_bytes(300 * 2 * 2, 1900)
    k = 0
    for rate in (48000, 44100):
        for loop in (0, 1):
            for ch in (0, 1):
                cases[f"sampler_{k}"] = dict(out0=fill(F, 2000 + k, -1, 1), pcm=pcm, params=np.array([2, rate, 1, ch, loop], np.uint32))
                k += 1
    for ki in range(2):
        for fi, freq in enumerate((440.0, 55.0, 2793.83)):
            color = np.float32(0.3) + np.float32(0.2) * np.float32(fi)
            cases[f"nice_{ki}_{fi}"] = dict(out0=fill(2 * F, 2100 + ki * 4 + fi, -1, 1), params=_f32(SR, freq, color))
            cases[f"pmosc_{ki}_{fi}"] = dict(out0=fill(2 * F, 2200 + ki * 4 + fi, -1, 1), params=_f32(SR, np.float32(freq) * np.float32(0.5), 0.4))
            cases[f"fsaw_{ki}_{fi}"] = dict(out0=fill(2 * F, 2600 + ki * 4 + fi, -1, 1), params=_f32(SR, freq, 0.0))      # examples/modules.zig:130-187
            cases[f"hsq_{ki}_{fi}"] = dict(out0=fill(2 * F, 2700 + ki * 4 + fi, -1, 1), params=_f32(SR, freq, 0.0))       # examples/modules.zig:250-289
    cases["basics"] = dict(a=fill(F, 2300, -2, 2), b=fill(F, 2301, -2, 2), dest0=fill(F, 2302, -2, 2))
    mix = fill(F, 2400, -6, 6); mix[3] = np.nan; mix[4] = np.inf; mix[5] = -np.inf
    cases["mixdown"] = dict(mix=mix)
    cases["math"] = dict(sin_x=fill(4096, 2500, -40, 40))
    cases["math2"] = math2_inputs()
    return cases


def splitmix_seq(seed, n):
    """dump_vectors.zig `splitmix(&st)` called n times from st = seed"""
    out, st, M = [], seed, (1 << 64) - 1
    for _ in range(n):
        st = (st + 0x9E3779B97F4A7C15) & M
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out.append(z ^ (z >> 31))
    return out


def math2_inputs():
    """dump_vectors.zig mathProbes2's inputs: multiples of pi/2 at -4..+4 ulps, pow(2, .) exponents, crafted generator states."""
    base = (np.arange(1, 456, dtype=np.float64) * 1.5707963267948966).astype(np.float32).view(np.uint32)
    x = np.zeros(4096, np.uint32)
    x[:4095] = (base[:, None].astype(np.int64) + np.arange(-4, 5, dtype=np.int64)[None, :]).reshape(-1).astype(np.uint32)
    py = fill(4096, 2600, -2.5, 6.5)
    n = 0
    for e in range(-2, 7):
        for q in range(4):
            off = np.float32(1.0) / np.float32(1 << (17 + q))
            py[n] = np.float32(e) + off; py[n + 1] = np.float32(e) - off
            n += 2
    sm = splitmix_seq(2700, 80)
    st = np.zeros(160, np.uint64)
    for c in range(40):
        kk = c % 10
        st[c * 4 + 1] = sm[2 * c] | 1; st[c * 4 + 2] = sm[2 * c + 1] | 1
        st[c * 4 + 3] = (1 << kk) if kk < 9 else (1 << 41)
    return dict(sin_pio2_x=x.view(np.float32), pow2_y=py, rare_states=st)


def write_oracle_vectors(directory):
    os.makedirs(directory, exist_ok=True)
    names = []
    for name, rec in case_inputs().items():
        rec = dict(rec)
        rec.update(expected(name, rec))
        write(os.path.join(directory, name + ".zgv"), rec)
        names.append(name)
    return names
