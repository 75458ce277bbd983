"""The C++ zangscript compiler in libzang_hip.so (zh_zscript_*) against the reference's golden generated text
and against the Python front-end: both implementations must print the same Zig, the same HIP, the same
module metadata and the same compile errors."""
import os

import pytest

from oracle import zangscript as zs
from zang_amd import zscript_native as native

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = open(os.path.join(HERE, "golden", "script_modules.txt")).read()
REF_SCRIPT = "/root/reference/examples/script.txt"


def test_native_matches_reference_golden_text():
    src = "Instrument = defmodule\n    freq: cob,\nbegin\n    out freq * 2\nend"
    got = native.NativeScript(src, packages=(zs.zang_builtin_package,)).generate_zig()
    assert got == open(os.path.join(HERE, "golden", "zangscript_example_test.zig.txt")).read()


def _both(text, only=None):
    py = zs.compile(text)
    nat = native.NativeScript(text)
    return py, nat


def test_native_equals_python_on_test_script():
    py, nat = _both(SCRIPT)
    assert nat.generate_zig() == zs.generate_zig(py)
    hip_py, meta_py = zs.generate_hip(py)
    hip_nat, meta_nat = nat.generate_hip()
    assert hip_nat == hip_py
    assert meta_nat == meta_py
    for only in (["Lead"], ["Jingle", "Echo"], ["Doubler"]):
        a, ma = zs.generate_hip(py, only=only)
        b, mb = nat.generate_hip(only=only)
        assert a == b and ma == mb and sorted(mb) == sorted(only)


@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="the reference's example script only exists in the build container")
def test_native_equals_python_on_reference_example_script():
    text = open(REF_SCRIPT).read()
    py, nat = _both(text)
    assert nat.generate_zig() == zs.generate_zig(py)
    a, ma = zs.generate_hip(py)
    b, mb = nat.generate_hip()
    assert a == b and ma == mb
    assert [n for n, _ in py.exported_modules] == ["CurvePlayerInner", "CurvePlayer", "Square", "MySineOsc", "InnerInstrument", "Echoes",
                                                  "CoinInstrument", "TrackPlayer", "DemoPlayer"]


ERRORS = [
    "X = defmodule\nbegin\n out .\nend",
    "X = defmodule\nbegin\n out 1.2.3\nend",
    "X = defmodule\n pi: constant,\nbegin\nend",
    "X = defmodule\n a: constant,\n a: cob,\nbegin\nend",
    "X = defmodule\n a: nothing,\nbegin\nend",
    "X = defmodule\nbegin\n out 1\n",
    "X = 1\nX = 2",
    "X = defmodule\nbegin\n out y\nend",
    "X = defmodule\nbegin\n out SineOsc(freq=1)\nend",
    "X = defmodule\nbegin\n out SineOsc(freq=1, phase=0, foo=1)\nend",
    "X = defmodule\nbegin\n out SineOsc(freq=1, phase=0, phase=0)\nend",
    "X = defmodule\nbegin\n out SineOsc(freq=true, phase=0)\nend",
    "X = defmodule\nbegin\n out Noise(color=.purple)\nend",
    "X = defmodule\nbegin\n out Envelope(attack=.cubed, decay=.linear(1), release=.linear(1), sustain_volume=1, note_on=true)\nend",
    "X = defmodule\nbegin\n out true\nend",
    "X = defmodule\nbegin\n out true + 1\nend",
    "X = defmodule\nbegin\n feedback 1\nend",
    "Y = 1 + 2",
    "A = B\nB = A",
    "X = defmodule\nbegin\n out 3(a=1)\nend",
    "C = defcurve 0 1 0 2 end",
    "T = deftrack f: cob, begin end",
    "X = defmodule\nbegin\n out delay 10 begin out delay 5 begin out 1 end end\nend",
    "X = defmodule\r\nbegin\r\n    out foo * 2\r\nend",
    "X = defmodule\nbegin\n out from 3, 1 begin out 1 end\nend",
    "X = defmodule\n c: curve,\nbegin\n out c\nend",
    "X = defmodule\nbegin\n out Filter(input=1, type=.low_pass, cutoff=true, res=0)\nend",
    "X = @",
    "A = defmodule\nbegin\n out A()\nend",
    "A = defmodule\nbegin\n out B()\nend\nB = defmodule\nbegin\n out A() * 2\nend",
]


@pytest.mark.parametrize("src", ERRORS)
def test_native_reports_the_same_errors(src):
    with pytest.raises(zs.ScriptError) as e1:
        zs.compile(src)
    with pytest.raises(native.NativeScriptError) as e2:
        native.NativeScript(src)
    assert str(e2.value) == str(e1.value)


def test_native_unsupported_module_is_reported():
    src = "P = defmodule\nbegin\n out delay 0 begin out feedback feedback 1 end\nend"
    a, ma = zs.generate_hip(zs.compile(src))
    b, mb = native.NativeScript(src).generate_hip()
    assert a == b and ma == mb and "delay of 0 samples" in mb["P"]["error"]


def test_differential_fuzz_python_vs_native():
    """3,000 random mutations of the test script: the two front-ends must agree on everything -- the same
    compile error text, or the same Zig + HIP + metadata -- and the C++ one must survive malformed input."""
    import random
    parts = SCRIPT.split("\n\n")
    rng = random.Random(20241002)
    tokens = ["(", ")", ",", "=", "*", "+", "-", "/", ".", ":", "begin", "end", "out", "feedback", "delay", "from", "defmodule", "defcurve",
              "deftrack", "true", "false", "pi", "sin", "max", "SineOsc", "Envelope", "freq", "note_on", "0.5", "3", "x", ".cubed", ".low_pass",
              "cob", "constant", "waveform", "\n", " "]
    ok = 0
    for _ in range(3000):
        src = "\n\n".join(rng.sample(parts, rng.randint(1, 3)))
        for _ in range(rng.randint(1, 4)):
            k, pos = rng.random(), rng.randrange(len(src) + 1)
            if k < 0.4:
                src = src[:pos] + rng.choice(tokens) + src[pos:]
            elif k < 0.7:
                src = src[:pos] + src[pos + rng.randint(1, 12):]
            else:
                a = rng.randrange(len(src))
                src = src[:pos] + src[a:min(len(src), a + rng.randint(1, 30))] + src[pos:]
        try:
            py = zs.compile(src)
            r1 = ("ok", zs.generate_zig(py), zs.generate_hip(py))
        except zs.ScriptError as e:
            r1 = ("err", str(e))
        try:
            nat = native.NativeScript(src)
            r2 = ("ok", nat.generate_zig(), nat.generate_hip())
        except native.NativeScriptError as e:
            r2 = ("err", str(e))
        assert r1 == r2, src
        ok += r1[0] == "ok"
    assert ok > 50


# ---- the role-wave form (ZH_ZSCRIPT_FORM_ROLES: zs_paint_pc_<name>, csrc/zscript_emit.hip plan / role_kernel) ----
def _role_section(hip, name):
    i = hip.index("// role-wave form", hip.index("void __launch_bounds__(64) zs_paint_%s(" % name))
    j = hip.find("\nextern \"C\" __device__ const uint32_t zs_ranges_ok_", i)
    return hip[i:] if j < 0 else hip[i:j]


def test_role_form_leaves_the_lane_kernels_text_alone():
    """forms = ROLES adds kernels; every line of the lane-form text is still there, in order, and the metadata is the same"""
    nat = native.NativeScript(SCRIPT)
    a, ma = nat.generate_hip()
    b, mb = nat.generate_hip(forms=native.FORM_ROLES)
    assert ma == mb
    it = iter(b.split("\n"))
    assert all(any(l == m for m in it) for l in a.split("\n")), "a lane-form line is missing or out of order"
    assert "zs_paint_pc_" in b and "zs_paint_pc_" not in a


def test_role_form_of_the_reference_recipe():
    """FilteredSawtooth (examples/modules.zig:130-187): oscillator, envelope, the filter's recurrence and the writer are roles of
    their own; the oscillator (a sawtooth, color 0: 17 instructions a frame, 2 of them state) and the envelope run in two waves each;
    the Filter is dealt out in three parts (input + offset with the producer, the recurrence alone, the mix with the writer); every
    state word is stored once; the emitter marks it twice (worth it at few voices, and still with the chip nearly full: 7 waves, work
    within 1.25 x the body's)."""
    import re
    nat = native.NativeScript(SCRIPT)
    hip, _ = nat.generate_hip(only=["FilteredSawtooth"], forms=native.FORM_ROLES)
    sec = _role_section(hip, "FilteredSawtooth")
    head = sec.split("\n")[0]
    m = re.match(r"// role-wave form: (\d+) waves \((\d+) loader, (\d+) roles, the last the writer\), (\d+) frames per tile, (\d+) tile buffers", head)
    assert m, head
    waves, loaders, roles, ch, bufs = map(int, m.groups())
    assert roles == 4 and loaders == 1 and waves > loaders + roles and ch in (16, 32)
    info = re.search(r"zs_pc_info_FilteredSawtooth\[4\] = \{(\d+)u, (\d+)u, (\d+)u, (\d+)u\}", sec)
    threads, lds, lds_zf, hint = map(int, info.groups())
    assert threads == waves * 64 and lds == bufs * ch * 256 and lds_zf < lds <= 65536 and hint == 3 and waves == 7
    assert re.search(r"\.pre\(t\d+\);", sec) and ".core<false, false>(" in sec and ".mix(" in sec
    # the recurrence role holds the core and nothing else that computes
    core_role = [blk for blk in sec.split("} else if")[1:] if ".core<" in blk]
    assert len(core_role) == 1 and ".frame" not in core_role[0] and ".mix(" not in core_role[0] and ".pre(" not in core_role[0]
    stores = re.findall(r"zs_st_[fu]\(L\.state, (\d+), V, v,", sec)
    assert sorted(map(int, stores)) == list(range(8))


def test_role_form_compiles_for_gfx950():
    """every module of the test script and a few random ones (delays, tracks, nested modules: opaque units), through hiprtc"""
    from tests import script_fuzz
    from zang_amd import script
    nat = native.NativeScript(SCRIPT)
    hip, meta = nat.generate_hip(forms=native.FORM_ROLES)
    assert script.compile_hip(hip) > 10000
    n_pc = hip.count(") zs_paint_pc_")
    assert n_pc >= 5, n_pc
    for seed in (0, 2, 9):
        text, name = script_fuzz.generate(seed)
        f = native.NativeScript(text, "fuzz")
        src, _ = f.generate_hip(only=[name], forms=native.FORM_ROLES)
        f.close()
        assert "zs_paint_pc_" + name in src
        assert script.compile_hip(src) > 10000
