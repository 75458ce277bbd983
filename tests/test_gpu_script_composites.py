"""GPU: the reference's two remaining composite recipes as GENERATED kernels.

examples/modules.zig:130-187 (FilteredSawtoothInstrument: TriSawOsc * 1.5, Envelope, multiply, low-pass Filter into the output)
and :250-289 (HardSquareInstrument: PulseOsc times Gate, multiplied into the output) are written in zangscript in
tests/golden/script_modules.txt, compiled by the library's own front end + HIP emitter into ONE fused kernel each, and compared
bit for bit with the oracle's UNFUSED composition of the same module calls through temps (oracle/zang_oracle.c
zo_filtered_sawtooth_paint / zo_hard_square_paint, which follow the reference line by line) -- recipes the reference defines
and the emitter has never seen, at 4,096 and 131,072 voices, over three sub-spans per buffer and a note script (retrigger, release,
silent voices), with a constant and a buffer frequency, into zeroed and into live output."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SR, F = 48000.0, 1024
SCRIPT = open(os.path.join(ROOT, "tests", "golden", "script_modules.txt")).read()
SUBSPANS = util.SPANS_THREE                      # (0, 200), (200, 777), (777, 1024)


def _voices(V):
    """every voice up to 512, a stride sample above (the oracle is one CPU thread)"""
    return np.arange(V) if V <= 512 else np.unique(np.concatenate([np.arange(0, V, max(1, V // 384)), [0, 1, 2, 3, V - 1]]))


def _note_script(rng, V, buffers):
    """per buffer and sub-span: note_on [V] bool, note_id_changed [V] bool -- a note per voice starting in buffer 0, released in a
    random later sub-span, some voices retriggered, some never playing"""
    steps = buffers * len(SUBSPANS)
    off_at = rng.integers(1, steps + 2, V)              # the step at which the note is released (steps + 1: held throughout)
    retrig = np.where(rng.random(V) < 0.25, rng.integers(2, steps, V), steps + 5)
    silent = rng.random(V) < 0.05
    out = []
    for k in range(steps):
        on = (k < off_at) | (k >= retrig)
        on &= ~silent
        nic = (k == 0) | (k == retrig)
        out.append((on, nic))
    return out


def _check_form(ctx, roles, name):
    """the kernel the last paint launched is the form the case asked for (zh_last_form)"""
    ran = ctx.last_form()
    if roles is not None:
        assert any(k == "zs_paint_pc_" + name for k in ran) == bool(roles), (roles, ran)


def _check_default_form(ctx, name, V, span):
    """the library's own choice: the role-wave form for the reference's filtered recipe up to script_pc_maxv voices (spans of at
    least 64 frames) -- up to half of it for the variant with a frequency image, whose role form does 1.6 x the body's work (the
    emitter's second hint bit, csrc/zscript_emit.hip role_kernel) -- the lane form above and for a recipe without a chain (HardSquare:
    frame ranges do better)"""
    ran = ctx.last_form()
    maxv = ctx.forms()["script_pc_maxv"][1]
    want = name.startswith("FilteredSawtooth") and V <= (maxv if name == "FilteredSawtooth" else maxv // 2) and span[1] - span[0] >= 64
    assert any(k == "zs_paint_pc_" + name for k in ran) == want, (name, V, span, ran)


# roles: the role-wave form of the generated kernel (zs_paint_pc_<name>) forced on (1), off (0), or the library's own choice
@pytest.mark.parametrize("roles", [None, 0, 1])
@pytest.mark.parametrize("zero_first", [True, False])
@pytest.mark.parametrize("freq_kind", ["constant", "buffer"])
@pytest.mark.parametrize("V", [4096, 131072])
def test_filtered_sawtooth_generated_kernel_equals_the_unfused_reference_recipe(ctx, oracle, monkeypatch, V, freq_kind, zero_first, roles):
    if V == 131072 and ((freq_kind == "buffer") != zero_first or roles is not None):
        pytest.skip("at 131,072 voices: constant + live output and buffer + zeroed output (two 512 MiB images per case), the library's own form")
    import torch
    from zang_amd import script, zang
    if roles is not None:
        util.set_form(monkeypatch, script_pc=roles)
    rng = np.random.default_rng(V + (7 if freq_kind == "buffer" else 0))
    L = oracle.lib()
    idx = _voices(V)
    name = "FilteredSawtoothCtl" if freq_kind == "buffer" else "FilteredSawtooth"      # one script module per arm of the reference's ConstantOrBuffer
    from zang_amd import zscript_native as native
    prog = script.ScriptProgram(SCRIPT, ctx, only=[name], forms=native.FORM_ROLES if roles else native.FORM_ROLES_WORTH)
    m = prog.module(name, V)
    freq = rng.uniform(40.0, 5000.0, V).astype(np.float32)
    freq[:4] = [-3.0, 7000.0, 0.25, 5999.0]           # silent (freq < 0), silent (> sr / 8), very low, just in range
    cutoff = float(L.zo_filter_cutoff_from_frequency(float(np.float32(440.0) * np.float32(L.zo_note_c5())), SR))   # examples/modules.zig:179-183
    buffers = 2
    notes = _note_script(rng, V, buffers)
    base = rng.uniform(-1, 1, (len(idx), F)).astype(np.float32)
    insts = []
    for v in idx:
        st = oracle.FilteredSawtooth(); L.zo_filtered_sawtooth_init(C.byref(st)); insts.append(st)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32); t2 = np.zeros(F, np.float32)
    gfreq = util.dev(freq)
    k = 0
    for b in range(buffers):
        if freq_kind == "buffer":                       # a per-frame frequency image: vibrato around the voice's frequency
            fimg_h = (freq[idx, None] * (1.0 + 0.02 * np.sin(np.arange(F)[None, :] * 0.01 + b))).astype(np.float32)
            fimg = torch.empty((F, V), dtype=torch.float32, device=ctx.device)
            fimg[:] = gfreq[None, :]
            fimg[:, torch.from_numpy(idx).to(ctx.device)] = torch.from_numpy(np.ascontiguousarray(fimg_h.T)).to(ctx.device)
        ref = base.copy()
        out = torch.zeros((F, V), dtype=torch.float32, device=ctx.device)
        out[:, torch.from_numpy(idx).to(ctx.device)] = torch.from_numpy(np.ascontiguousarray(base.T)).to(ctx.device)
        for (s, e) in SUBSPANS:
            on, nic = notes[k]; k += 1
            if zero_first:
                ref[:, s:e] = 0.0
            for q, v in enumerate(idx):
                f = oracle.buffer(fimg_h[q]) if freq_kind == "buffer" else oracle.constant(freq[v])
                L.zo_filtered_sawtooth_paint(C.byref(insts[q]), s, e, oracle.fptr(ref[q]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(t2),
                                             int(nic[v]), SR, f, int(on[v]))
            m.paint(zang.Span(s, e), [out], None, torch.from_numpy(nic.astype(np.uint8)).to(ctx.device),
                    {"sample_rate": SR, "freq": fimg if freq_kind == "buffer" else gfreq, "note_on": torch.from_numpy(on.astype(np.uint8)).to(ctx.device),
                     "cutoff": cutoff}, zero_first=zero_first)
            _check_form(ctx, roles, name)
            if roles is None:
                _check_default_form(ctx, name, V, (s, e))
        ctx.sync()
        got = out[:, torch.from_numpy(idx).to(ctx.device)].cpu().numpy().T
        util.assert_bitexact(np.ascontiguousarray(got), ref, f"FilteredSawtooth V={V} freq {freq_kind} zf={zero_first} buffer {b}")
        assert float(np.abs(ref).max()) > 0.01
    prog.close()


@pytest.mark.parametrize("roles", [None, 1])
@pytest.mark.parametrize("zero_first", [True, False])
@pytest.mark.parametrize("V", [4096, 131072])
def test_hard_square_generated_kernel_equals_the_unfused_reference_recipe(ctx, oracle, monkeypatch, V, zero_first, roles):
    if V == 131072 and roles is not None:
        pytest.skip("at 131,072 voices: the library's own form")
    import torch
    from zang_amd import script, zang
    if roles is not None:
        util.set_form(monkeypatch, script_pc=roles)
    rng = np.random.default_rng(V + 1)
    L = oracle.lib()
    idx = _voices(V)
    from zang_amd import zscript_native as native
    prog = script.ScriptProgram(SCRIPT, ctx, only=["HardSquare"], forms=native.FORM_ROLES if roles else native.FORM_ROLES_WORTH)
    m = prog.module("HardSquare", V)
    freq = rng.uniform(40.0, 5000.0, V).astype(np.float32)
    freq[:4] = [-3.0, 7000.0, 0.25, 5999.0]
    buffers = 2
    notes = _note_script(rng, V, buffers)
    base = rng.uniform(-1, 1, (len(idx), F)).astype(np.float32)
    insts = []
    for v in idx:
        st = oracle.HardSquare(); L.zo_hard_square_init(C.byref(st)); insts.append(st)
    t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
    gfreq = util.dev(freq)
    sel = torch.from_numpy(idx).to(ctx.device)
    k = 0
    for b in range(buffers):
        ref = base.copy()
        out = torch.zeros((F, V), dtype=torch.float32, device=ctx.device)
        out[:, sel] = torch.from_numpy(np.ascontiguousarray(base.T)).to(ctx.device)
        for (s, e) in SUBSPANS:
            on, nic = notes[k]; k += 1
            if zero_first:
                ref[:, s:e] = 0.0
            for q, v in enumerate(idx):
                L.zo_hard_square_paint(C.byref(insts[q]), s, e, oracle.fptr(ref[q]), oracle.fptr(t0), oracle.fptr(t1), int(nic[v]), SR, float(freq[v]), int(on[v]))
            m.paint(zang.Span(s, e), [out], None, torch.from_numpy(nic.astype(np.uint8)).to(ctx.device),
                    {"sample_rate": SR, "freq": gfreq, "note_on": torch.from_numpy(on.astype(np.uint8)).to(ctx.device)}, zero_first=zero_first)
            _check_form(ctx, roles, "HardSquare")
            if roles is None:
                _check_default_form(ctx, "HardSquare", V, (s, e))
        ctx.sync()
        got = out[:, sel].cpu().numpy().T
        util.assert_bitexact(np.ascontiguousarray(got), ref, f"HardSquare V={V} zf={zero_first} buffer {b}")
        assert float(np.abs(ref).max()) > 0.5
    prog.close()


@pytest.mark.parametrize("V", [1, 63, 64, 65, 200, 4160, 65536])
def test_role_wave_form_equals_the_lane_form_at_odd_voice_counts_and_spans(ctx, monkeypatch, V):
    """zs_paint_pc_<name> against zs_paint_<name> (which the tests above hold against the oracle), bit for bit, where the workgroup
    geometry has edges: fewer voices than a wave, a partial last workgroup, spans shorter than a tile, ending inside a tile, starting
    off zero; `+=` onto live output; the frequency as an image; state carried from paint to paint in each form."""
    import torch
    from zang_amd import script, zang, zscript_native as native
    rng = np.random.default_rng(V)
    spans = [(0, 1024), (0, 5), (5, 37), (37, 100), (100, 1001), (1001, 1024), (3, 3), (0, 1024)]
    for name in ("FilteredSawtooth", "FilteredSawtoothCtl", "Bell"):
        prog = script.ScriptProgram(SCRIPT, ctx, only=[name], forms=native.FORM_ROLES)
        freq = util.dev(rng.uniform(40.0, 5000.0, V).astype(np.float32))
        fimg = torch.from_numpy(rng.uniform(40.0, 5000.0, (F, V)).astype(np.float32)).to(ctx.device)
        on = torch.from_numpy((rng.random(V) < 0.8).astype(np.uint8)).to(ctx.device)
        base = torch.from_numpy(rng.uniform(-1, 1, (F, V)).astype(np.float32)).to(ctx.device)
        outs, states = [], []
        for roles in (0, 1):
            util.set_form(monkeypatch, script_pc=roles)
            m = prog.module(name, V)
            out = base.clone()
            for k, (s, e) in enumerate(spans):
                p = {"sample_rate": SR, "note_on": on, "freq": fimg if name == "FilteredSawtoothCtl" else freq}
                if name != "Bell":
                    p["cutoff"] = 0.07
                m.paint(zang.Span(s, e), [out], None, k in (0, 4), p, zero_first=(k % 3 == 0))
                ran = ctx.last_form()
                assert (ran == ["zs_paint_pc_" + name]) == bool(roles), (roles, ran)
            ctx.sync()
            outs.append(out.cpu().numpy()); states.append(m.get_state())
            m.close()
        util.assert_bitexact(outs[1], outs[0], f"{name} V={V}: role-wave form against lane form")
        assert np.array_equal(states[0], states[1]), f"{name} V={V}: state words"
        assert float(np.abs(outs[0]).max()) > 0.01
        prog.close()


def test_the_library_picks_the_form_by_voice_count_and_the_emitters_two_hints(ctx):
    """Up to script_pc_maxv / 2 voices every module with the first hint bit paints in the role-wave form; from there to script_pc_maxv only
    those with the second one (FilteredSawtooth, Glide: 181 -> 110 us and 163 -> 122 us at 65,536 voices; Hiss, whose noise role is walked
    by four waves, and Bell with its twelve waves would lose there: profiles/r06/role_ab_65536.txt); above it none; a module without the
    first bit never."""
    import re
    import torch
    from zang_amd import script, zang
    prog = script.ScriptProgram(SCRIPT, ctx, only=["FilteredSawtooth", "Glide", "Hiss", "Bell", "Pluck"])
    bits = {n: int(re.search(r"zs_pc_info_%s\[4\] = \{\d+u, \d+u, \d+u, (\d)u\}" % n, prog.hip_source).group(1)) for n in ("FilteredSawtooth", "Glide", "Hiss", "Bell")}
    assert bits == {"FilteredSawtooth": 3, "Glide": 3, "Hiss": 1, "Bell": 1}, bits
    assert "zs_paint_pc_Pluck" not in prog.hip_source                  # (FORM_ROLES_WORTH: no role-wave kernel is generated for it at all)
    maxv = ctx.forms()["script_pc_maxv"][1]
    for V in (maxv // 2, maxv // 2 + 64, maxv, maxv + 64):
        out = ctx.image(256, V)
        on = torch.ones(V, dtype=torch.uint8, device=ctx.device)
        for name in ("FilteredSawtooth", "Glide", "Hiss", "Bell", "Pluck"):
            m = prog.module(name, V)
            p = {}
            for pname, kind, enum in m.params:                         # something valid for every param, whatever the module
                p[pname] = (SR if pname == "sample_rate" else on if kind == "boolean" else ".low_pass" if enum == "FilterType" else
                            220.0 if pname in ("freq", "goal") else 0.1)
            m.paint(zang.Span(0, 256), [out], None, True, p, zero_first=True)
            ran = ctx.last_form()
            want = name != "Pluck" and (V <= maxv // 2 or (bits[name] & 2 and V <= maxv))
            assert (ran == ["zs_paint_pc_" + name]) == bool(want), (name, V, ran)
            m.close()
    ctx.sync()
    prog.close()


@pytest.mark.parametrize("V", [64, 1000])
def test_role_wave_form_equals_the_lane_form_in_place_and_tolerant(ctx, monkeypatch, V):
    """Two uses the tests above do not make of zs_paint_pc_<name>: the frequency image IS the output image (its loader role reads a row
    tiles before the writer role adds into it -- as the one-walk form reads a sample before it adds to it), and ZH_PAINT_TOLERANT:
    the role-wave kernels carry no f32 sine (csrc/zscript_emit.hip role_kernel resolves every sine exactly), so a tolerant paint that
    takes them has the EXACT lane kernel's bits -- the flag is a permission (include/zang_hip.h), and the lane kernel that uses it
    stays within 1e-5 of them."""
    import torch
    from zang_amd import script, zang, zscript_native as native
    rng = np.random.default_rng(V + 5)
    spans = [(0, 1024), (0, 70), (70, 1024), (0, 1024)]
    for name, tolerant, in_place in (("FilteredSawtoothCtl", False, True), ("Bell", True, False), ("FilteredSawtoothCtl", True, True)):
        prog = script.ScriptProgram(SCRIPT, ctx, only=[name], forms=native.FORM_ROLES)
        freq = util.dev(rng.uniform(40.0, 5000.0, V).astype(np.float32))
        on = torch.from_numpy((rng.random(V) < 0.8).astype(np.uint8)).to(ctx.device)
        base = torch.from_numpy(rng.uniform(40.0, 5000.0, (F, V)).astype(np.float32)).to(ctx.device)
        outs, states = [], []
        for roles in (0, 1):
            util.set_form(monkeypatch, script_pc=roles)
            m = prog.module(name, V)
            out = base.clone()
            for k, (s, e) in enumerate(spans):
                p = {"sample_rate": SR, "note_on": on, "freq": out if in_place else freq}
                if name != "Bell":
                    p["cutoff"] = 0.07
                m.paint(zang.Span(s, e), [out], None, k == 0, p, tolerant=tolerant and bool(roles))   # lane form: exact
                ran = ctx.last_form()
                assert (ran == ["zs_paint_pc_" + name]) == bool(roles), (roles, ran)
                if in_place and k < len(spans) - 1:
                    ctx.sync()
                    out.copy_(base)                                    # (frequencies again, not frequencies + signal)
            ctx.sync()
            outs.append(out.cpu().numpy()); states.append(m.get_state())
            m.close()
        util.assert_bitexact(outs[1], outs[0], f"{name} V={V} tolerant={tolerant} in_place={in_place}: role-wave form against lane form")
        assert np.array_equal(states[0], states[1]), f"{name} V={V}: state words"
        assert float(np.abs(outs[0] - base.cpu().numpy()).max()) > 0.01
        if tolerant and name == "Bell":                                 # ... and the lane kernel's f32 sines, from the same start
            util.set_form(monkeypatch, script_pc=0)
            pair = []
            for tol in (True, False):
                m = prog.module(name, V)
                out = torch.zeros_like(base)
                m.paint(zang.Span(0, 1024), [out], None, True, {"sample_rate": SR, "note_on": on, "freq": freq}, tolerant=tol)
                ctx.sync()
                pair.append(out.cpu().numpy().astype(np.float64))
                m.close()
            peak = np.maximum(np.abs(pair[1]).max(axis=0), 1e-30)
            worst = float((np.abs(pair[0] - pair[1]).max(axis=0) / peak).max())
            assert 0.0 < worst <= 1e-5, worst
        prog.close()


def test_a_script_compiled_once_is_loaded_from_its_code_object(ctx, tmp_path):
    """zh_script_compile -> a gfx950 code object kept on disk -> zh_script_load_code: the reference's compile-once flow
    (examples/example_script.zig:6-8: zangc runs before the program is built).  The second program never calls hiprtc and paints the
    same bits, lane kernel and role-wave kernel; something that is not a code object is refused."""
    import ctypes
    import torch
    from zang_amd import abi, script, zang
    V = 300
    rng = np.random.default_rng(3)
    freq = util.dev(rng.uniform(40.0, 5000.0, V).astype(np.float32))
    outs = []
    for k in range(2):
        prog = script.ScriptProgram(SCRIPT, ctx, only=["FilteredSawtooth"], code_cache=str(tmp_path))
        assert prog.loaded_from_cache == (k == 1)
        m = prog.module("FilteredSawtooth", V)
        out = torch.zeros((F, V), dtype=torch.float32, device=ctx.device)
        for (s, e) in SUBSPANS:
            m.paint(zang.Span(s, e), [out], None, s == 0, {"sample_rate": SR, "freq": freq, "note_on": True, "cutoff": 0.07}, zero_first=True)
        assert ctx.last_form() == ["zs_paint_pc_FilteredSawtooth"]
        ctx.sync()
        outs.append(out.cpu().numpy())
        prog.close()
    util.assert_bitexact(outs[1], outs[0], "loaded from the cached code object")
    assert len(list(tmp_path.glob("zs_*.hsaco"))) == 1
    h = ctypes.c_void_p()
    junk = b"not a code object" * 8
    assert ctx.lib.zh_script_load_code(ctx.handle, junk, len(junk), ctypes.byref(h)) == abi.ZH_ERR_INVALID
