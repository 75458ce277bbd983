"""zangscript.

CPU: the front-end against the reference's golden generated text (src/zangscript/tests.zig:44-92 -- the
only reference-held vector for this subsystem), tokenizer / parser / codegen behaviours and error
messages, the instruction list of the repo-authored test script, the oracle-side interpreter against
plain numpy, and that the generated HIP compiles for gfx950 (hiprtc needs no GPU).
GPU: every module of the test script, fused kernel vs the oracle-side interpreter, bit for bit, over
several consecutive paints (carried state, sub-spans, note on/off, retrigger)."""
import os

import numpy as np
import pytest

from oracle import zangscript as zs
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = open(os.path.join(HERE, "golden", "script_modules.txt")).read()

GOLDEN_SOURCE = """Instrument = defmodule
    freq: cob,
begin
    out freq * 2
end"""
GOLDEN_ZIG = open(os.path.join(HERE, "golden", "zangscript_example_test.zig.txt")).read()


# ------------------------------------------------------------------------------------------ CPU
def test_reference_golden_generated_text():
    """tests.zig "example test": only the zang builtin package, like compileScript there (:5-8)."""
    got = zs.generate_zig(zs.compile(GOLDEN_SOURCE, packages=(zs.zang_builtin_package,)))
    assert got == GOLDEN_ZIG


def test_tokenizer():
    from oracle.zangscript.errors import Source
    from oracle.zangscript.tokenize import Tokenizer
    t = Tokenizer(Source("t", "a_1 = .cubed(0.5) // note\n  -3*pi begin end\r\nx"))
    kinds = []
    while True:
        tok = t.next()
        kinds.append(tok.tt)
        if tok.tt == "end_of_file":
            break
    assert kinds == ["name", "sym_equals", "enum_value", "sym_left_paren", "number", "sym_right_paren", "sym_minus", "number",
                     "sym_asterisk", "name", "kw_begin", "kw_end", "name", "end_of_file"]
    assert t.line == 2                                       # LF, CRLF


@pytest.mark.parametrize("src,msg", [
    ("X = defmodule\nbegin\n out .\nend", "dot must be followed by an identifier"),
    ("X = defmodule\nbegin\n out 1.2.3\nend", "malformatted number"),
    ("X = defmodule\n pi: constant,\nbegin\nend", "`pi` is a reserved name"),
    ("X = defmodule\n a: constant,\n a: cob,\nbegin\nend", "redeclaration of param `a`"),
    ("X = defmodule\n a: nothing,\nbegin\nend", "expected param type, found `nothing`"),
    ("X = defmodule\nbegin\n out 1\n", "expected local declaration, `out`, `feedback` or `end`, found end of file"),
    ("X = 1\nX = 2", "redeclaration of global `X`"),
    ("X = defmodule\nbegin\n out y\nend", "use of undeclared identifier `y`"),
    ("X = defmodule\nbegin\n out SineOsc(freq=1)\nend", "argument list is missing param `phase`"),
    ("X = defmodule\nbegin\n out SineOsc(freq=1, phase=0, foo=1)\nend", "call target has no param called `foo`"),
    ("X = defmodule\nbegin\n out SineOsc(freq=1, phase=0, phase=0)\nend", "param `phase` provided more than once"),
    ("X = defmodule\nbegin\n out SineOsc(freq=true, phase=0)\nend", "expected float or buffer value"),
    ("X = defmodule\nbegin\n out Noise(color=.purple)\nend", "expected one of 'white', 'pink'"),
    ("X = defmodule\nbegin\n out Envelope(attack=.cubed, decay=.linear(1), release=.linear(1), sustain_volume=1, note_on=true)\nend",
     "expected one of 'instantaneous', 'linear'(number), 'squared'(number), 'cubed'(number)"),
    ("X = defmodule\nbegin\n out true\nend", "expected buffer value, found boolean"),
    ("X = defmodule\nbegin\n out true + 1\nend", "arithmetic can only be performed on numeric types"),
    ("X = defmodule\nbegin\n feedback 1\nend", "`feedback` can only be used within a `delay` operation"),
    ("Y = 1 + 2", "constant arithmetic is not supported"),
    ("A = B\nB = A", "circular reference in global"),
    ("X = defmodule\nbegin\n out 3(a=1)\nend", "not a module"),
    ("C = defcurve 0 1 0 2 end", "time value must be greater than the previous time value"),
    ("T = deftrack f: cob, begin end", "track param cannot be cob or waveform"),
    ("X = defmodule\nbegin\n out delay 10 begin out delay 5 begin out 1 end end\nend", "you cannot nest delay operations"),
])
def test_compile_errors(src, msg):
    with pytest.raises(zs.ScriptError) as e:
        zs.compile(src)
    assert msg in str(e.value)
    assert "script.txt:" in str(e.value)


def test_error_location_and_carets():
    with pytest.raises(zs.ScriptError) as e:
        zs.compile("X = defmodule\nbegin\n    out foo * 2\nend")
    text = str(e.value)
    assert text.startswith("script.txt:3:9: use of undeclared identifier `foo`")
    assert "    out foo * 2\n        ^^^" in text


def test_instruction_lists_and_temps():
    s = zs.compile(SCRIPT)
    names = [n for n, _ in s.exported_modules]
    assert names == ["Doubler", "Pluck", "CycleSine", "Bell", "Lead", "Hiss", "Buzz", "Crush", "Glide", "Sweep", "Maths", "Echo", "EchoLead", "Coin", "Jingle", "LateJingle", "Trig", "Shapes", "FilteredSawtooth", "FilteredSawtoothCtl", "HardSquare"]
    r = s.module_results[s.module_index("Doubler")]
    assert (r.num_temps, r.num_temp_floats, [i.kind for i in r.instructions]) == (1, 0, ["cob_to_buffer", "arith_buffer_float"])
    assert r.instructions[1].out.kind == "output"           # written straight into the result location
    r = s.module_results[s.module_index("Pluck")]
    assert [i.kind for i in r.instructions] == ["cob_to_buffer", "call", "arith_buffer_float", "arith_float_buffer", "call",
                                                "arith_buffer_buffer", "arith_buffer_float"]
    assert r.num_temps == 3 and [s.modules[f].builtin_name for f in r.fields] == ["SineOsc", "Envelope"]
    # a script module calling script modules: the callee's temps are claimed from the caller (codegen.zig:540-543)
    lead = s.module_results[s.module_index("Lead")]
    call = [i for i in lead.instructions if i.kind == "call"][0]
    assert len(call.temps) == s.module_results[s.module_index("Bell")].num_temps
    # temp floats are never reused (they become `const` in Zig, codegen.zig:788-792)
    maths = s.module_results[s.module_index("Maths")]
    assert maths.num_temp_floats == len([i for i in maths.instructions if i.kind in ("arith_float", "arith_float_float")])
    # shadowing: `freq = freq * 0.5` reads the param, later uses read the local
    bell = s.module_results[s.module_index("Bell")]
    assert bell.instructions[0].kind == "cob_to_buffer" and bell.instructions[1].kind == "arith_buffer_float"


def test_generated_zig_of_test_script_is_stable():
    text = zs.generate_zig(zs.compile(SCRIPT))
    assert "pub const Lead = _module16;" in text
    assert "zang.multiplyScalar(span, temps[0], temps[1], 0.25);" in text
    assert ".type = params.ftype," in text                   # `type` is a primitive's name, not a Zig keyword token
    from oracle.zangscript.emit_zig import ident
    assert ident("error") == '@"error"' and ident("freq") == "freq"   # keyword escaping (codegen_zig.zig:40-46)
    assert "std.math.pow(f32, temps[" in text and "const temp_float" in text


def test_operator_precedence_and_negation():
    s = zs.compile("M = defmodule\n a: constant,\nbegin\n out -a + 2 * 3 - 4 / a\nend")
    ins = s.module_results[s.module_index("M")].instructions
    assert [(i.kind, i.op) for i in ins] == [("arith_float", "neg"), ("arith_float_float", "mul"), ("arith_float_float", "add"),
                                             ("arith_float_float", "div"), ("arith_float_float", "sub"), ("float_to_buffer", None)]


def _interp_paint(script, name, params, frames=64, nic=True, voices=1, first_seed=0):
    from oracle import zs_interp
    vs = zs_interp.make_voices(script, name, voices, first_seed)
    outs = []
    for v in vs:
        out = np.zeros(frames, np.float32)
        v.paint(0, frames, out, nic, params)
        outs.append(out)
    return outs


def test_interpreter_against_numpy(oracle):
    s = zs.compile(SCRIPT)
    f = np.linspace(100, 200, 64).astype(np.float32)
    assert np.array_equal(_interp_paint(s, "Doubler", [np.float32(48000), f])[0], f * np.float32(2))
    assert np.array_equal(_interp_paint(s, "Doubler", [np.float32(48000), np.float32(3)])[0], np.full(64, 6, np.float32))
    x = np.linspace(-1, 1, 64).astype(np.float32)
    k = np.float32(0.7)
    L = oracle.lib()
    small = zs.compile("""
A = defmodule x: waveform, k: constant, begin out x - k end
B = defmodule x: waveform, begin out min(x, 0.1) / (2 + cos(x)) end
C = defmodule x: waveform, k: constant, begin out pow(abs(x), 0.5) + sqrt(k * 4) + -k end
""")
    z = np.zeros(64, np.float32)
    same = lambda got, want: np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
    assert same(_interp_paint(small, "A", [np.float32(48000), x, k])[0], z + (x - k))
    cos = np.array([L.zo_math_cosf(float(v)) for v in x], np.float32)
    assert same(_interp_paint(small, "B", [np.float32(48000), x])[0],
                z + np.where(x < np.float32(0.1), x, np.float32(0.1)) / (z + (cos + np.float32(2))))
    pw = np.array([L.zo_math_powf(float(abs(v)), 0.5) for v in x], np.float32)
    f0 = np.float32(k * np.float32(4)); f1 = np.float32(np.sqrt(f0)); f2 = np.float32(-k)
    assert same(_interp_paint(small, "C", [np.float32(48000), x, k])[0], z + ((z + (pw + f1)) + f2))


def test_interpreter_matches_direct_oracle_calls(oracle):
    """Pluck through the interpreter == the same three oracle calls made by hand."""
    import ctypes as C
    s = zs.compile(SCRIPT)
    got = _interp_paint(s, "Pluck", [np.float32(48000), np.float32(440), True], frames=256)[0]
    L = oracle.lib()
    osc, env = oracle.SineOsc(), oracle.Envelope()
    L.zo_sineosc_init(C.byref(osc)); L.zo_envelope_init(C.byref(env))
    t = [np.zeros(256, np.float32) for _ in range(3)]
    L.zo_set(0, 256, oracle.fptr(t[0]), 440.0)
    L.zo_sineosc_paint(C.byref(osc), 0, 256, oracle.fptr(t[1]), 48000.0, oracle.buffer(t[0]), oracle.constant(0.0))
    t[0][:] = 0; L.zo_multiply_scalar(0, 256, oracle.fptr(t[0]), oracle.fptr(t[1]), 0.25)
    t[1] = np.where(np.float32(0) > t[0], np.float32(0), t[0]).astype(np.float32)
    t[0][:] = 0
    p = oracle.EnvelopeParams(48000.0, oracle.curve(3, 0.02), oracle.curve(2, 0.15), oracle.curve(1, 0.8), 0.6, 1)
    L.zo_envelope_paint(C.byref(env), 0, 256, oracle.fptr(t[0]), 1, C.byref(p))
    t[2][:] = 0; L.zo_multiply(0, 256, oracle.fptr(t[2]), oracle.fptr(t[1]), oracle.fptr(t[0]))
    out = np.zeros(256, np.float32)
    L.zo_multiply_scalar(0, 256, oracle.fptr(out), oracle.fptr(t[2]), float(np.float32(np.pi)))
    assert np.array_equal(got.view(np.uint32), out.view(np.uint32))


def test_generated_hip_compiles_for_gfx950():
    from zang_amd import script
    src, meta = zs.generate_hip(zs.compile(SCRIPT))
    assert all("error" not in m for m in meta.values())
    assert meta["Hiss"]["noise_fields"] == 2 and meta["Hiss"]["state_words"] == 18
    assert meta["Lead"]["state_words"] == meta["Bell"]["state_words"] + meta["Pluck"]["state_words"]
    assert script.compile_hip(src) > 10000


def test_delay_front_end_and_state_layout():
    s = zs.compile(SCRIPT)
    r = s.module_results[s.module_index("Echo")]
    assert r.delays == [37] and r.num_temps == 4             # feedback, result, feedback-out + one working temp (codegen.zig:628-690)
    d = r.instructions[0]
    assert d.kind == "delay" and d.out.kind == "output" and d.feedback_temp != d.feedback_out_temp
    assert "readDelayBuffer" in zs.generate_zig(s) and "writeDelayBuffer" in zs.generate_zig(s)
    _, meta = zs.generate_hip(s, only=["Echo"])
    assert meta["Echo"]["state_words"] == 1 + 37 + 2         # ring index, ring, Filter (l, b)


def test_track_call_front_end():
    s = zs.compile(SCRIPT)
    r = s.module_results[s.module_index("Jingle")]
    assert r.triggers == [0] and r.note_trackers == [0] and r.num_temp_floats == 1
    tc = [i for i in r.instructions if i.kind == "track_call"][0]
    assert tc.out.kind == "output" and tc.speed.kind == "literal_number"
    assert [i.kind for i in tc.instructions][:2] == ["arith_float_float", "arith_buffer_float"]      # scale = pitch / 1000, per sub-span
    text = zs.generate_zig(s)
    assert ".{ .t = 0.0004, .note_id = 2, .params = .{ .pitch = 1000.0, .note_on = true, .shape = .{ .cubed = 0.001 } } }," in text
    assert "const _new_note = (params.note_on and note_id_changed) or _result.note_id_changed;" in text
    assert "const _iap0 = self.tracker0.consume(params.sample_rate / 1.5, span);" in text
    assert "const _new_note = note_id_changed or _result.note_id_changed;" in text            # LateJingle has no note_on param
    _, meta = zs.generate_hip(s, only=["Jingle"])
    assert meta["Jingle"]["state_words"] == 3 + 1 + 4 + 4     # tracker+trigger, PulseOsc, two Envelopes


def test_oracle_trigger_against_reference_cases():
    """The interpreter's Trigger restatement against the reference's own unit tests
    (src/zang/trigger_test.zig, transcribed in tests/golden/scheduler_tests.json)."""
    import json
    from oracle import zs_interp
    G = json.load(open(os.path.join(HERE, "golden", "scheduler_tests.json")))
    for case in G["trigger"]:
        cur = None
        for step in case["steps"]:
            impulses = [(fr, nid, p) for (fr, nid, _), p in zip(step["impulses"], step["params"])]
            spans, cur = zs_interp.trigger_spans(cur, impulses, 0, 1024)
            assert [list(x) for x in spans] == step["expected"], case["name"]


def test_oracle_note_tracker_frames():
    from oracle import zs_interp
    st = {"next": 0, "t": np.float32(0)}
    times = [0.0, 0.01, 0.0213, 0.5]
    assert zs_interp.note_tracker_consume(st, times, np.float32(48000), 0, 1024) == [(0, 1, 0), (480, 2, 1), (1022, 3, 2)]
    assert st["next"] == 3 and st["t"] == np.float32(np.float32(1024) / np.float32(48000))
    assert zs_interp.note_tracker_consume(st, times, np.float32(48000), 100, 200) == []


def test_unsupported_constructs_are_reported_not_miscompiled():
    src = """Player = defmodule
    shape: curve,
begin
    out from deftrack
        c: curve,
    begin
        0.0 (c=shape)
    end, 1 begin
        out Curve(curve=c, function=.linear)
    end
end"""
    with pytest.raises(zs.ScriptError):                       # a track note cannot reach the module's params (global context)
        zs.compile(src)
    s = zs.compile("P = defmodule\nbegin\n out delay 0 begin out feedback feedback 1 end\nend")
    _, meta = zs.generate_hip(s)
    assert "delay of 0 samples" in meta["P"]["error"]


def test_hiprtc_errors_are_reported():
    from zang_amd import script
    with pytest.raises(script.ScriptCompileError) as e:
        script.compile_hip('#include "script_rt.hip.h"\nextern "C" __global__ void k() { undefined_fn(); }\n')
    assert "undefined_fn" in str(e.value)


# ------------------------------------------------------------------------------------------ GPU
F = 96
V = 70          # one full wave + a partial one


def _per_voice(value, v):
    if isinstance(value, np.ndarray) and value.ndim >= 1 and value.shape[0] == V:
        x = value[v]
        if isinstance(x, np.ndarray):
            return x
        return bool(x) if value.dtype == np.bool_ else np.float32(x)
    return value


def _device_value(kind, value):
    import torch
    from tests.util import to_image
    if isinstance(value, np.ndarray) and value.ndim == 2:
        return to_image(value)
    if isinstance(value, np.ndarray) and value.dtype == np.bool_:
        return torch.from_numpy(value.astype(np.uint8)).cuda()
    if isinstance(value, np.ndarray):
        return torch.from_numpy(value.astype(np.float32)).cuda()
    return value


def _parity(ctx, name, paints, first_seed=0, add_into=None):
    """paints: [(start, end, nic, {param: value})]; value = scalar | bool | [V] array | [V][F] array |
    (label, payload) | [(t, value)]; nic = bool or [V] bool array."""
    import torch
    from oracle import zs_interp
    from tests.util import assert_bitexact, from_image, to_image
    from zang_amd import script, zang
    prog = script.ScriptProgram(SCRIPT, ctx, only=[name])
    mod = prog.module(name, V, first_seed)
    voices = zs_interp.make_voices(zs.compile(prog.text, prog.filename), name, V, first_seed)
    base = np.zeros((V, F), np.float32) if add_into is None else add_into.copy()
    ref = base.copy()
    img = to_image(base)
    order = [p[0] for p in mod.params]
    for start, end, nic, params in paints:
        dev = {k: _device_value(None, v) for k, v in params.items()}
        nic_dev = torch.from_numpy(nic.astype(np.uint8)).cuda() if isinstance(nic, np.ndarray) else nic
        mod.paint(zang.Span(start, end), [img], None, nic_dev, dev)
        for v in range(V):
            voices[v].paint(start, end, ref[v], bool(nic[v]) if isinstance(nic, np.ndarray) else nic,
                            [_per_voice(params[k], v) for k in order])
    ctx.sync()
    assert_bitexact(from_image(img), ref, name)
    prog.close()
    return ref


def _freqs(seed=1):
    return np.random.default_rng(seed).uniform(60, 3000, V).astype(np.float32)


def _tolerant_parity(ctx, name, paints, expect_tolerant, first_seed=0):
    """The same paints on two instances of one loaded kernel, one with ZH_PAINT_TOLERANT: per voice and paint, every sample within
    1e-5 of the larger of the voice's peak over the span and 1 (a unit-amplitude sine's f32 form is within 2.4e-7 of musl's; the
    module scales it) -- and identical bits when the emitter found no sine that may be tolerant.  Returns the worst ratio."""
    import torch
    from tests.util import from_image, to_image
    from zang_amd import script, zang
    prog = script.ScriptProgram(SCRIPT, ctx, only=[name])
    assert ("ZS_T" in prog.hip_source) == expect_tolerant, name
    exact, tol = prog.module(name, V, first_seed), prog.module(name, V, first_seed)
    worst = 0.0
    for start, end, nic, params in paints:
        dev = {k: _device_value(None, v) for k, v in params.items()}
        nic_dev = torch.from_numpy(nic.astype(np.uint8)).cuda() if isinstance(nic, np.ndarray) else nic
        ie, it = to_image(np.zeros((V, F), np.float32)), to_image(np.zeros((V, F), np.float32))
        exact.paint(zang.Span(start, end), [ie], None, nic_dev, dev)
        tol.paint(zang.Span(start, end), [it], None, nic_dev, dev, tolerant=True)
        ctx.sync()
        a, b = from_image(ie).astype(np.float64), from_image(it).astype(np.float64)
        if not expect_tolerant:
            assert np.array_equal(from_image(ie).view(np.uint32), from_image(it).view(np.uint32)), name
            continue
        assert np.array_equal(a[:, :start], b[:, :start]) and np.array_equal(a[:, end:], b[:, end:]), name
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)), name
        if end > start:
            peak = np.maximum(np.where(fin, np.abs(a), 0.0)[:, start:end].max(axis=1), 1.0)
            ratio = (np.where(fin, np.abs(a - b), 0.0)[:, start:end].max(axis=1) / peak).max()
            assert ratio <= 1e-5, (name, start, end, ratio)
            worst = max(worst, ratio)
    prog.close()
    return worst


@pytest.mark.gpu
def test_gpu_doubler_const_and_buffer(ctx):
    rng = np.random.default_rng(0)
    buf = rng.uniform(-2, 2, (V, F)).astype(np.float32)
    _parity(ctx, "Doubler", [(0, F, False, {"sample_rate": 48000.0, "freq": buf})], add_into=rng.uniform(-1, 1, (V, F)).astype(np.float32))
    _parity(ctx, "Doubler", [(0, 40, False, {"sample_rate": 48000.0, "freq": _freqs()}), (40, F, False, {"sample_rate": 48000.0, "freq": 3.5})])


@pytest.mark.gpu
def test_gpu_pluck_note_cycle(ctx):
    on = np.random.default_rng(2).random(V) < 0.7
    f = _freqs()
    p = lambda note_on: {"sample_rate": 48000.0, "freq": f, "note_on": note_on}
    _parity(ctx, "Pluck", [(0, 50, True, p(on)), (50, 64, False, p(on)), (64, 80, False, p(~on | on & False)), (80, F, on, p(on))])


@pytest.mark.gpu
def test_gpu_sineosc_quiet_body_and_its_fallback(ctx):
    """A SineOsc whose frequency is a constant this paint runs a frame body without the sine's rare-path branch when no
    voice of its wave can reach that path; one voice with an absurd frequency (or a frequency buffer) sends the whole wave
    through the general body.  Both against the oracle, and a module with two oscillators (Lead: a literal-frequency one
    and the nested Pluck's)."""
    on = np.ones(V, bool)
    f = _freqs(5)
    wild = f.copy()
    wild[3] = 4.0e12                    # t moves 8.3e7 a frame: beyond kSineOscSmallT at once, Payne-Hanek arguments within the span
    wild[40] = -2.5e13
    wild2 = f.copy()
    wild2[66] = np.inf                  # only the second wave leaves the quiet body
    rng = np.random.default_rng(9)
    img = rng.uniform(100, 2000, (V, F)).astype(np.float32)
    for name in ("Pluck", "Lead"):
        for freq in (f, wild, wild2, 440.0, img):
            p = {"sample_rate": 48000.0, "freq": freq, "note_on": on}
            _parity(ctx, name, [(0, 50, True, p), (50, F, False, p)])
    # the Envelope's half of the quiet body: at 1.5 kHz the attack is 30 frames and the decay 225, so chunks with a stage
    # end (general body) sit between chunks without one (frame_quiet); voices released at different times
    off = np.random.default_rng(3).random(V) < 0.5
    for name in ("Pluck", "Lead", "Bell"):
        p = lambda note_on: {"sample_rate": 1500.0, "freq": f * np.float32(0.01), "note_on": note_on}
        _parity(ctx, name, [(0, 41, True, p(on)), (41, 70, False, p(on & ~off)), (70, F, False, p(~on))])


@pytest.mark.gpu
def test_gpu_empty_span_still_runs_prologues(ctx):
    """paint over an empty span is not a no-op in the reference: Envelope's note-on prologue (Envelope.zig:41-50)
    and Portamento's newCurve run, and the next paint continues from there."""
    on = np.ones(V, bool)
    f = _freqs(20)
    p = lambda note_on: {"sample_rate": 48000.0, "freq": f, "note_on": note_on}
    _parity(ctx, "Pluck", [(0, 30, True, p(on)), (30, 30, False, p(~on)), (30, 60, False, p(~on)), (60, 60, True, p(on)), (60, F, False, p(on))])
    g = lambda goal, a, b: {"sample_rate": 48000.0, "goal": goal, "note_on": a, "prev_note_on": b}
    _parity(ctx, "Glide", [(0, 20, True, g(_freqs(21), on, ~on)), (20, 20, True, g(_freqs(22), on, on)), (20, F, False, g(_freqs(22), on, on))])


@pytest.mark.gpu
def test_gpu_cycle_sine(ctx):
    rng = np.random.default_rng(3)
    ph = rng.uniform(0, 1, (V, F)).astype(np.float32)
    _parity(ctx, "CycleSine", [(0, 33, False, {"sample_rate": 44100.0, "freq": _freqs(), "phase": ph}),
                               (33, F, False, {"sample_rate": 44100.0, "freq": rng.uniform(50, 900, (V, F)).astype(np.float32), "phase": 0.25})])


@pytest.mark.gpu
def test_gpu_bell_and_lead_inlined_modules(ctx):
    on = np.ones(V, bool)
    f = _freqs(4)
    for name in ("Bell", "Lead"):
        p = lambda note_on: {"sample_rate": 48000.0, "freq": f, "note_on": note_on}
        _parity(ctx, name, [(0, 64, True, p(on)), (64, 80, False, p(~on)), (80, F, True, p(on))])


@pytest.mark.gpu
def test_gpu_hiss_noise_seeds_and_runtime_filter_type(ctx):
    cut = np.random.default_rng(5).uniform(0.05, 0.9, V).astype(np.float32)
    for ftype in ("low_pass", "notch", "bypass"):
        _parity(ctx, "Hiss", [(0, 48, False, {"sample_rate": 48000.0, "cut": cut, "ftype": (ftype, None)}),
                              (48, F, False, {"sample_rate": 48000.0, "cut": 0.2, "ftype": (ftype, None)})], first_seed=1234)


@pytest.mark.gpu
def test_gpu_buzz_oscillators_both_paths(ctx):
    f = _freqs(6)
    f[:5] = [-1.0, 7000.0, 0.0, 5999.0, 6000.5]                # silent voices: freq < 0 or > sr/8 paints nothing
    col = np.random.default_rng(6).uniform(0, 1, V).astype(np.float32)
    col[:4] = [0.0, 1.0, 0.5, 0.2]
    on = np.random.default_rng(7).random(V) < 0.5
    _parity(ctx, "Buzz", [(0, 70, False, {"sample_rate": 48000.0, "freq": f, "color": col, "note_on": on}),
                          (70, F, False, {"sample_rate": 48000.0, "freq": f, "color": col, "note_on": ~on})])


@pytest.mark.gpu
def test_gpu_crush_decimator_distortion(ctx):
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, (V, F)).astype(np.float32)
    rate = rng.uniform(-100, 60000, V).astype(np.float32)       # <= 0: silent, >= sr: bypass + reset
    drive = rng.uniform(0, 1, V).astype(np.float32)
    _parity(ctx, "Crush", [(0, 31, False, {"sample_rate": 48000.0, "input": x, "rate": rate, "drive": drive}),
                           (31, F, False, {"sample_rate": 48000.0, "input": x, "rate": 8000.0, "drive": drive})])


@pytest.mark.gpu
def test_gpu_glide_portamento(ctx):
    on = np.ones(V, bool)
    g1, g2 = _freqs(9), _freqs(10)
    p = lambda goal, a, b: {"sample_rate": 48000.0, "goal": goal, "note_on": a, "prev_note_on": b}
    _parity(ctx, "Glide", [(0, 30, True, p(g1, on, ~on)), (30, 60, True, p(g2, on, on)), (60, F, False, p(g2, on, on))])


@pytest.mark.gpu
def test_gpu_sweep_curves(ctx):
    shape = [(0.0, 440.0), (0.0005, 880.0), (0.001, 110.0), (0.0016, 660.0)]
    p = {"sample_rate": 48000.0, "freq_mul": np.random.default_rng(11).uniform(0.5, 2, V).astype(np.float32), "shape": shape}
    _parity(ctx, "Sweep", [(0, 40, True, p), (40, 64, False, p), (64, F, False, p)])


@pytest.mark.gpu
def test_gpu_maths(ctx):
    rng = np.random.default_rng(12)
    x = rng.uniform(-2, 2, (V, F)).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1.0, -1.0, 0.5, 2.0, -2.0, 1e-20]
    k = rng.uniform(0.1, 3, V).astype(np.float32)
    _parity(ctx, "Maths", [(0, F, False, {"sample_rate": 48000.0, "x": x, "k": k})])


@pytest.mark.gpu
def test_gpu_delay_chunks_and_nested_delays(ctx):
    """Spans longer than the ring (96 > 37, 20): the body's modules see one paint call per chunk."""
    rng = np.random.default_rng(13)
    x = rng.uniform(-1, 1, (V, F)).astype(np.float32)
    vol = rng.uniform(0.1, 0.9, V).astype(np.float32)
    for ftype in ("low_pass", "high_pass"):
        p = {"sample_rate": 48000.0, "input": x, "echo_volume": vol, "ftype": (ftype, None)}
        _parity(ctx, "Echo", [(0, F, False, p), (0, 30, False, p), (30, F, False, p)])
    on = np.ones(V, bool)
    f = _freqs(14)
    q = lambda note_on: {"sample_rate": 48000.0, "freq": f, "note_on": note_on}
    _parity(ctx, "EchoLead", [(0, 50, True, q(on)), (50, F, False, q(on)), (0, F, False, q(~on))])


@pytest.mark.gpu
def test_gpu_track_calls(ctx):
    """Sub-spans per note, per-note params, retrigger through note_id_changed and through the host's
    note_on && note_id_changed reset, carried notes across paints, a gap before the first note."""
    on = np.random.default_rng(15).random(V) < 0.8
    f = _freqs(16)
    p = lambda note_on: {"sample_rate": 48000.0, "freq": f, "note_on": note_on}
    nic = np.random.default_rng(17).random(V) < 0.5
    _parity(ctx, "Jingle", [(0, 40, True, p(on)), (40, 90, False, p(on)), (0, F, nic, p(on)), (0, F, False, p(~on))])
    speed = np.random.default_rng(18).uniform(0.5, 3.0, V).astype(np.float32)
    q = {"sample_rate": 44100.0, "speed": speed}
    _parity(ctx, "LateJingle", [(0, 50, False, q), (50, F, False, q), (0, F, True, q), (0, F, False, q)])
    # curve-typed track params: each note carries its own defcurve, played by a Curve module inside the sub-span
    r = {"sample_rate": 48000.0, "freq": _freqs(23)}
    _parity(ctx, "Shapes", [(0, 30, True, r), (30, F, False, r), (0, F, nic, r)])


@pytest.mark.gpu
def test_gpu_trig_all_argument_ranges(ctx):
    """sin / cos across musl's magnitude ranges, both signs, the range boundaries, tiny, huge, inf, nan."""
    edges = np.array([0x3f490fda, 0x4016cbe3, 0x407b53d1, 0x40afeddf, 0x40e231d5, 0x39800000, 0x4dc90fdb, 0x7f7fffff], np.uint32)
    near = np.concatenate([edges - 1, edges, edges + 1]).view(np.float32)
    special = np.array([0.0, -0.0, 1e-30, -1e-30, np.inf, -np.inf, np.nan, 1e10, -3e20, 12345.678], np.float32)
    rng = np.random.default_rng(19)
    x = rng.uniform(-8, 8, (V, F)).astype(np.float32)
    x[0, :len(near)] = near
    x[1, :len(near)] = -near
    x[2, :len(special)] = special
    x[3] = rng.uniform(-2000, 2000, F).astype(np.float32)
    x[4] = (rng.standard_normal(F) * 1e-3).astype(np.float32)
    _parity(ctx, "Trig", [(0, F, False, {"sample_rate": 48000.0, "x": x})])


@pytest.mark.gpu
def test_gpu_zero_first_and_state_roundtrip(ctx):
    from tests.util import from_image, to_image
    from zang_amd import script, zang
    prog = script.ScriptProgram(SCRIPT, ctx, only=["Pluck"])
    mod = prog.module("Pluck", V)
    assert mod.num_temps == 3 and mod.get_state().shape == (5, V)
    img = to_image(np.full((V, F), 9.0, np.float32))
    p = {"sample_rate": 48000.0, "freq": 330.0, "note_on": True}
    mod.paint(zang.Span(0, 48), [img], None, True, p, zero_first=True)
    st = mod.get_state()
    mod.paint(zang.Span(48, F), [img], None, False, p, zero_first=True)
    a = from_image(img).copy()
    assert np.all(a[:, :1] == 0.0) and np.any(a[:, 40:] != 0.0)          # the 9.0 fill is gone
    mod.set_state(st)
    img2 = to_image(np.zeros((V, F), np.float32))
    mod.paint(zang.Span(48, F), [img2], None, False, p)
    assert np.array_equal(from_image(img2)[:, 48:].view(np.uint32), a[:, 48:].view(np.uint32))
    with pytest.raises(KeyError):
        mod.paint(zang.Span(0, 8), [img], None, False, {"sample_rate": 48000.0, "freq": 1.0})
    prog.close()


def test_zangc_cli(tmp_path):
    """tools/zangc.py: both backends, the dumps, error exit code (tools/zangc.zig:8-27)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    src = os.path.join(HERE, "golden", "script_modules.txt")
    out, cg, bi = tmp_path / "o.hip", tmp_path / "cg.txt", tmp_path / "bi.txt"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "zangc.py"), src, "-o", str(out), "--dump-codegen", str(cg), "--dump-builtins", str(bi)],
                       cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert 'extern "C" __global__ void __launch_bounds__(64) zs_paint_Lead' in out.read_text()
    assert "module Lead: 17 state words/voice" in r.stderr
    assert "module Pluck: num_temps=3" in cg.read_text() and "enum FilterType: bypass, low_pass" in bi.read_text()
    z = tmp_path / "o.zig"
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "zangc.py"), src, "--backend", "zig", "-o", str(z)], cwd=root).returncode == 0
    assert z.read_text().startswith("// THIS FILE WAS GENERATED BY THE ZANGC COMPILER")
    bad = tmp_path / "bad.txt"
    bad.write_text("X = defmodule\nbegin\n out foo\nend\n")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "zangc.py"), str(bad), "--check"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 1 and ":3:6: use of undeclared identifier `foo`" in r.stderr


@pytest.mark.gpu
def test_gpu_loader_argument_checks(ctx):
    import ctypes as C
    from zang_amd import abi, script, zang
    prog = script.ScriptProgram("M = defmodule x: waveform, begin out x * 2 end", ctx)
    with pytest.raises(KeyError):
        prog.module("Nope", 4)
    h = C.c_void_p()
    assert prog.lib.zh_script_module_create(prog.handle, b"Nope", 4, 0, 0, C.byref(h)) != 0      # no such kernel in the hipModule
    mod = prog.module("M", 8)
    img = ctx.image(16, 8)
    small = ctx.image(16, 4)
    with pytest.raises(abi.ZangHipError):                     # the waveform image must cover the module's voices
        mod.paint(zang.Span(0, 16), [img], None, False, {"sample_rate": 48000.0, "x": small})
    with pytest.raises(abi.ZangHipError):                     # span beyond the output image
        mod.paint(zang.Span(0, 32), [img], None, False, {"sample_rate": 48000.0, "x": img})
    prog.close()


@pytest.mark.gpu
def test_gpu_paintcurve_param_takes_library_values(ctx):
    """A script module with a PaintCurve param painted with the library's own zang.PaintCurve.* values (ADVICE r1
    script.py:154: the enum branch unpacked the value before testing for abi.Curve) and with the (label, payload)
    form: both equal the built-in Envelope module given the same curve, bit for bit."""
    import torch
    from zang_amd import modules as mod, script as zscript, zang
    text = """
Shaped = defmodule
    attack: PaintCurve,
    note_on: boolean,
begin
    out Envelope(attack=attack, decay=.cubed(0.004), release=.linear(0.003), sustain_volume=0.6, note_on)
end
"""
    prog = zscript.ScriptProgram(text, ctx)
    n, frames = 192, 512
    on = torch.from_numpy((np.random.default_rng(3).random(n) < 0.6).astype(np.uint8)).cuda()
    sp = zang.Span(0, frames)
    for curve, tup in ((zang.PaintCurve.cubed(0.002), ("cubed", 0.002)), (zang.PaintCurve.linear(0.001), (".linear", 0.001)),
                       (zang.PaintCurve.instantaneous, "instantaneous")):
        ref_m = mod.Envelope(n, ctx)
        ref = ctx.image(frames, n, fill=0.0)
        ref_m.paint(sp, [ref], [], True, ref_m.Params(48000.0, curve, zang.PaintCurve.cubed(0.004), zang.PaintCurve.linear(0.003), 0.6, on))
        for value in (curve, tup):
            m = prog.module("Shaped", n)
            out = ctx.image(frames, n, fill=0.0)
            m.paint(sp, [out], None, True, {"sample_rate": 48000.0, "attack": value, "note_on": on})
            ctx.sync()
            assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (curve.tag, value)
            m.close()
    with pytest.raises(ValueError):                                   # a per-voice duration is not a script-module value
        per_voice = zang.PaintCurve.cubed(torch.full((n,), 0.002, device="cuda"))
        prog.module("Shaped", n).paint(sp, [ctx.image(frames, n, fill=0.0)], None, True, {"sample_rate": 48000.0, "attack": per_voice, "note_on": on})


def _random_params(mod, rng, nv, nf):
    """A plausible random value for every exported param of a script module (by declared kind)."""
    vals = {}
    for name, kind, enum in mod.params:
        if name == "sample_rate":
            vals[name] = 48000.0
        elif kind == "constant":
            vals[name] = rng.uniform(0.05, 0.95, nv).astype(np.float32) if name in ("color", "cut", "drive", "mix") else \
                rng.uniform(40.0, 3000.0, nv).astype(np.float32)
        elif kind == "boolean":
            vals[name] = rng.random(nv) < 0.6
        elif kind == "constant_or_buffer":
            vals[name] = rng.uniform(40.0, 3000.0, (nv, nf)).astype(np.float32) if rng.random() < 0.5 else rng.uniform(40.0, 3000.0, nv).astype(np.float32)
        elif kind == "buffer":
            vals[name] = rng.uniform(-1.0, 1.0, (nv, nf)).astype(np.float32)
        elif kind == "curve":
            vals[name] = [(0.0, 440.0), (0.001, 880.0), (0.003, 110.0), (0.005, 660.0)]
        else:                                                            # one_of
            label = {"PaintCurve": ("cubed", 0.002), "FilterType": ("low_pass", None), "NoiseColor": ("white", None),
                     "DistortionType": ("overdrive", None), "InterpolationFunction": ("smoothstep", None)}[enum]
            vals[name] = label
    return vals


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Pluck", "CycleSine", "Bell", "Lead", "Hiss", "Buzz", "Crush", "Glide", "Sweep", "Maths", "Jingle",
                                  "LateJingle", "Trig", "Shapes", "Echo", "EchoLead"])
def test_gpu_script_kernels_as_frame_ranges(ctx, name, monkeypatch):
    _script_ranges_equal_sequential(ctx, name, 200, monkeypatch)


@pytest.mark.gpu
def test_gpu_script_frame_ranges_at_a_mid_voice_count(ctx, monkeypatch):
    """40,000 voices = 625 waves: three frame ranges per 64 voices (the count is 2048 / waves up to 65,536 voices)."""
    _script_ranges_equal_sequential(ctx, "Pluck", 40000, monkeypatch)


def _script_ranges_equal_sequential(ctx, name, nv, monkeypatch):
    """A generated kernel at a small voice count is launched as frame ranges at once (gridDim.y > 1): every range runs the
    frame body over the earlier frames with the output discarded (the state walk survives, the output math is dead code)
    and then paints its own frames.  Same module, same calls, once with ZH_SCRIPT_RANGES=0 (the lane-per-voice walk that the
    other tests hold against the interpreter) and once ranged: images and state blobs must be identical.  Modules whose
    body writes a delay ring (Echo, EchoLead) or whose state walk reads computed signals (Bell's FM, Hiss's filter ...) are
    flagged by the emitter and always take the sequential launch (both paints are then the same launch)."""
    import torch
    from tests.util import to_image
    from zang_amd import script, zang
    nf = 416
    prog = script.ScriptProgram(SCRIPT, ctx, only=[name])
    a, b = prog.module(name, nv, 77), prog.module(name, nv, 77)
    # ranged: no delay ring, and no module output / transcendental feeding a builtin's state (an FM oscillator, a filter's
    # input): replaying such a walk costs as much as painting, so the emitter leaves those kernels sequential
    assert b.frame_ranges_ok == (name in ("Pluck", "CycleSine", "Crush", "Maths", "Jingle", "LateJingle", "Trig", "Shapes"))
    rng = np.random.default_rng(sum(ord(c) for c in name))
    base = rng.uniform(-1, 1, (nv, nf)).astype(np.float32)
    img_a, img_b = to_image(base), to_image(base)
    for k, (s, e, zf) in enumerate([(0, nf, True), (0, nf, False), (40, 300, False), (300, nf, True), (0, nf, False)]):
        vals = _random_params(a, rng, nv, nf)
        nic = rng.random(nv) < 0.3
        dev = {kk: _device_value(None, vv) for kk, vv in vals.items()}
        nic_dev = torch.from_numpy(nic.astype(np.uint8)).cuda()
        util.set_form(monkeypatch, script_ranges="0")
        a.paint(zang.Span(s, e), [img_a], None, nic_dev, dev, zero_first=zf)
        util.del_form(monkeypatch, "script_ranges")
        b.paint(zang.Span(s, e), [img_b], None, nic_dev, dev, zero_first=zf)
        ctx.sync()
        assert torch.equal(img_a.view(torch.int32), img_b.view(torch.int32)), f"{name}: image differs after paint {k}"
        assert np.array_equal(a.get_state(), b.get_state()), f"{name}: state differs after paint {k}"
    prog.close()


STAGES_SCRIPT = """
Stages = defmodule
    freq: cob,
    note_on: boolean,
begin
    a = Envelope(attack=.cubed(0.01), decay=.cubed(0.02), release=.cubed(0.03), sustain_volume=0.5, note_on)
    b = Envelope(attack=.linear(0.004), decay=.squared(0.05), release=.instantaneous, sustain_volume=0.25, note_on)
    c = Envelope(attack=.instantaneous, decay=.linear(0.015), release=.squared(0.011), sustain_volume=1, note_on)
    out SineOsc(freq, phase=0) * a + b * 0.5 + c * TriSawOsc(freq, color=0.3)
end
"""


@pytest.mark.gpu
@pytest.mark.parametrize("ranges", [None, "0", "3"])
def test_gpu_script_envelope_stage_ends_inside_replays(ctx, ranges, monkeypatch):
    """Three envelopes (one tag-specialised lane, two generic with instantaneous stages) whose stages last 32-400 frames at
    the 8 kHz this test paints at, over 512-frame paints with notes going on and off between them: a frame-range launch
    replays the earlier frames with the envelope's state-only step (envelope.hip.h frame_walk), stage ends included.
    Against the interpreter, voice by voice, image and (through the following paints) state."""
    import torch
    from oracle import zs_interp
    from tests.util import assert_bitexact, from_image, to_image
    from zang_amd import script, zang
    if ranges is None:
        util.del_form(monkeypatch, "script_ranges")
    else:
        util.set_form(monkeypatch, script_ranges=ranges)
    nv, nf = 70, 512
    prog = script.ScriptProgram(STAGES_SCRIPT, ctx, only=["Stages"])
    mod = prog.module("Stages", nv, 0)
    assert mod.frame_ranges_ok
    voices = zs_interp.make_voices(zs.compile(prog.text, prog.filename), "Stages", nv, 0)
    rng = np.random.default_rng(11)
    ref = np.zeros((nv, nf), np.float32)
    img = to_image(ref)
    order = [p[0] for p in mod.params]
    on = rng.random(nv) < 0.8
    for k, (s, e) in enumerate([(0, nf), (0, nf), (100, 420), (0, nf), (0, 130), (130, nf), (0, nf)]):
        nic = rng.random(nv) < (1.0 if k == 0 else 0.25)
        prev_on = on
        on = np.where(rng.random(nv) < 0.35, ~on, on)
        params = {"sample_rate": 8000.0, "freq": rng.uniform(60, 900, nv).astype(np.float32), "note_on": on}
        # Envelope.zig:45 asserts against note_on without a new note while releasing: a voice that comes back on is a new note
        nic = nic | (on & ~prev_on)
        dev = {kk: _device_value(None, vv) for kk, vv in params.items()}
        mod.paint(zang.Span(s, e), [img], None, torch.from_numpy(nic.astype(np.uint8)).cuda(), dev)
        for v in range(nv):
            voices[v].paint(s, e, ref[v], bool(nic[v]), [_per_voice(params[kk], v) for kk in order])
        ctx.sync()
        assert_bitexact(from_image(img), ref, "Stages paint %d" % k)
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ranges", [None, "0", "5"])
def test_gpu_script_in_place_param_image_keeps_one_walk(ctx, ranges, monkeypatch):
    """A cob param whose image IS the output image (an in-place use): a frame-range launch would replay input rows that
    another range is overwriting, so zh_script_module_paint must keep the one-walk form for such a paint (as the builtin
    modules do with bufs_alias).  Against the interpreter, which paints sample by sample like the reference."""
    import torch
    from oracle import zs_interp
    from tests.util import assert_bitexact, from_image, to_image
    from zang_amd import script, zang
    if ranges is None:
        util.del_form(monkeypatch, "script_ranges")
    else:
        util.set_form(monkeypatch, script_ranges=ranges)
    nv, nf = 200, 512
    prog = script.ScriptProgram(STAGES_SCRIPT, ctx, only=["Stages"])
    mod = prog.module("Stages", nv, 0)
    assert mod.frame_ranges_ok
    voices = zs_interp.make_voices(zs.compile(prog.text, prog.filename), "Stages", nv, 0)
    rng = np.random.default_rng(5)
    ref = rng.uniform(60, 900, (nv, nf)).astype(np.float32)          # the image holds the frequencies and receives `+=`
    img = to_image(ref)
    order = [p[0] for p in mod.params]
    for k, (s, e) in enumerate([(0, nf), (0, nf), (64, 400)]):
        on = rng.random(nv) < 0.7
        nic = np.ones(nv, bool)
        mod.paint(zang.Span(s, e), [img], None, torch.from_numpy(nic.astype(np.uint8)).cuda(),
                  {"sample_rate": 8000.0, "freq": img, "note_on": _device_value(None, on)})
        for v in range(nv):
            # the reference reads freq[i] before it adds into out[i] of the same frame: a copy of the row as it was
            vals = {"sample_rate": np.float32(8000.0), "freq": ref[v].copy(), "note_on": bool(on[v])}
            voices[v].paint(s, e, ref[v], True, [vals[kk] for kk in order])
        ctx.sync()
        assert_bitexact(from_image(img), ref, "in-place Stages paint %d" % k)
    prog.close()


@pytest.mark.gpu
def test_gpu_script_tolerant_sines(ctx):
    """ZH_PAINT_TOLERANT on generated kernels (csrc/zscript_emit.hip: the sines that reach the output through scaling and adding
    alone take their f32 form): Pluck's oscillator, CycleSine's sin(), Bell's and Lead's carriers -- their modulators reach a `phase`
    and stay exact --, the LFO and the plucked voice inside EchoLead's delay, a track's notes; kernels with no such sine (Buzz: its
    only sine is a vibrato on two oscillators' freq; Maths; Echo) answer bit for bit."""
    rng = np.random.default_rng(41)
    on = np.ones(V, bool)
    f = _freqs(4)
    p = lambda note_on: {"sample_rate": 48000.0, "freq": f, "note_on": note_on}
    cycle = [(0, 64, True, p(on)), (64, 80, False, p(~on)), (80, F, True, p(on)), (0, F, False, p(on))]
    worst = {}
    for name in ("Pluck", "Bell", "Lead", "EchoLead"):
        worst[name] = _tolerant_parity(ctx, name, cycle, True)
    ph = rng.uniform(0, 1, (V, F)).astype(np.float32)
    worst["CycleSine"] = _tolerant_parity(ctx, "CycleSine", [(0, 33, False, {"sample_rate": 44100.0, "freq": _freqs(), "phase": ph}),
                                                             (33, F, False, {"sample_rate": 44100.0, "freq": rng.uniform(50, 900, (V, F)).astype(np.float32), "phase": 0.25})], True)
    x = rng.uniform(-8, 8, (V, F)).astype(np.float32)
    x[3] = rng.uniform(-2000, 2000, F).astype(np.float32)
    x[2, :6] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 3e20], np.float32)
    worst["Trig"] = _tolerant_parity(ctx, "Trig", [(0, F, False, {"sample_rate": 48000.0, "x": x})], True)
    speed = rng.uniform(0.5, 3.0, V).astype(np.float32)
    q = {"sample_rate": 44100.0, "speed": speed}
    worst["LateJingle"] = _tolerant_parity(ctx, "LateJingle", [(0, 50, False, q), (50, F, False, q), (0, F, True, q)], True)
    assert max(worst.values()) > 0.0, "no tolerant sine ran"
    print("script kernels, tolerant against exact, worst error / max(peak, 1):", {k: "%.1e" % v for k, v in worst.items()})
    _tolerant_parity(ctx, "Buzz", [(0, F, True, {"sample_rate": 48000.0, "freq": f, "color": 0.3, "note_on": on})], False)
    xin = rng.uniform(-1, 1, (V, F)).astype(np.float32)
    _tolerant_parity(ctx, "Echo", [(0, F, False, {"sample_rate": 48000.0, "input": xin, "echo_volume": 0.5, "ftype": ("low_pass", None)})], False)


def _stored_words(hip, name):
    """the state words zs_paint_<name>'s epilogue stores"""
    import re
    i = hip.index("zs_paint_%s(" % name)
    body = hip[i:hip.index("\n}\n", i)]
    words = set()
    for m in re.finditer(r"zs_st_(f|u|u64)\(L\.state, (\d+), V, v,", body):
        w = int(m.group(2))
        words |= {w, w + 1} if m.group(1) == "u64" else {w}
    return words


def test_every_state_word_is_stored_by_a_range_capable_kernel():
    """A launch as frame ranges stores the end state into the OTHER blob and the host flips (csrc/script.hip: no copy of the start
    state): every state word must then be written by the epilogue -- all of the repo's script modules and 150 generated ones."""
    import re
    from tests import script_fuzz
    texts = [SCRIPT] + [script_fuzz.generate(seed)[0] for seed in range(3000, 3150)]
    checked = 0
    for text in texts:
        hip, meta = zs.generate_hip(zs.compile(text, "t"))
        for name, m in meta.items():
            if "error" in m:
                continue
            ok = re.search(r"zs_ranges_ok_%s = (\d)u;" % name, hip)
            if ok and ok.group(1) == "1":
                assert _stored_words(hip, name) == set(range(m["state_words"])), name
                checked += 1
    assert checked > 100
