"""Generated kernels against the oracle-side interpreter on RANDOM zangscript modules (tests/script_fuzz.py): every builtin
module, arithmetic, nested modules, `cob` params handed down, delays; random sub-spans, per-voice note events, constant /
per-voice / image frequencies with an absurd voice now and then.  tools/fuzz_scripts.py runs any number of further seeds on
the GPU box.  CPU: both front ends and both emitters agree on the generated text of the same random modules."""
import pytest

from tests import script_fuzz


@pytest.mark.parametrize("seed", range(24))
def test_random_scripts_both_emitters_agree(seed):
    from oracle import zangscript as zs
    from zang_amd import zscript_native as native
    text, name = script_fuzz.generate(seed)
    hip_py, meta_py = zs.generate_hip(zs.compile(text, "fuzz"))
    nat = native.NativeScript(text, "fuzz")
    hip_nat, meta_nat = nat.generate_hip()
    nat.close()
    assert "error" not in meta_py[name], meta_py[name]
    assert hip_py == hip_nat


def test_random_script_compiles_for_gfx950():
    from oracle import zangscript as zs
    from zang_amd import script
    text, _ = script_fuzz.generate(3)
    src, _ = zs.generate_hip(zs.compile(text, "fuzz"))
    assert script.compile_hip(src) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_random_script_parity(ctx, seed):
    script_fuzz.run_case(ctx, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12, 20))
def test_gpu_random_script_parity_as_frame_ranges(ctx, seed):
    """256-frame buffers, five frame ranges wherever the kernel allows them (replay of the state walk, quiet chunks in it)."""
    script_fuzz.run_case(ctx, seed, F=256, ranges=5)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(20, 44))
def test_gpu_random_script_parity_role_waves(ctx, seed):
    """Every paint forced through the role-wave form (zs_paint_pc_<name>: the body's units dealt to producer / recurrence /
    writer waves, values handed on through LDS tiles): spans shorter than a tile, partial last tiles, live and zeroed output,
    input images, delays and nested modules as the generator makes them; 96- and 256-frame buffers."""
    script_fuzz.run_case(ctx, seed, roles=1, F=96 if seed % 2 else 256)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [12032, 12155])
def test_gpu_role_waves_filter_into_an_inlined_modules_output(ctx, seed):
    """Found by the round's last soak (tools/fuzz_scripts.py 200 12000, the library's own choice of form): `out -abs(x)` then
    `out Filter(...)` inside a helper module that is inlined into a temp of its caller -- the Filter's mix piece of the role-wave form
    zeroed that temp first, as for a temp of its own module (codegen_zig.zig:284-291 zeroes only those), and the first `out` was lost."""
    script_fuzz.run_case(ctx, seed, roles=1, F=256 if seed % 2 else 96)
    script_fuzz.run_case(ctx, seed, F=256 if seed % 2 else 96)          # ... and in the form the library picks itself


@pytest.mark.gpu
def test_gpu_seed_1015_select_hazard(ctx):
    """The case that exposed the inline-asm v_cndmask of round 2 (lanes.hip.h zsel_hard): with a literal TriSawOsc color its
    `"s"(ballot(true))` operand became the EXEC register itself in one place, and a v_cndmask_b32_e64 with EXEC as its mask
    operand takes the upper half-wave's bits as zero (tools/ubench/select_hazard.hip): lanes 32-63 of the first frame of every
    chunk painted the triangle branch's value."""
    text = """Main = defmodule
    freq: cob,
    x: waveform,
    k: constant,
    note_on: boolean,
    prev_note_on: boolean,
begin
    m1 = Gate(note_on)
    out (TriSawOsc(freq=(207.0 + 196.1 * (freq - Gate(note_on))), color=0.993) * (Noise(color=.pink) / (2 + cos((Envelope(attack=.linear(0.00209), decay=.linear(0.00117), release=.squared(0.00244), sustain_volume=1, note_on) + m1)))))
end
"""
    script_fuzz.run_case(ctx, 1015, text=text)
    script_fuzz.run_case(ctx, 1015, F=256, ranges=3, text=text)


def test_no_inline_asm_reads_an_sgpr():
    """For an "s" operand the compiler may hand an inline asm EXEC or VCC themselves (it did: see the test above), and what an
    instruction makes of EXEC as an ordinary scalar operand is not what its SGPR copy gives.  No inline asm in the kernels
    takes a scalar operand; selects and compares stay the compiler's."""
    import glob, os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zang_amd", "csrc")
    bad = []
    for fn in glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.hip")):
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\((.*?)\);', open(fn).read(), re.S):
            if re.search(r'"[=+]?s"\s*\(', m.group(1)):
                bad.append((os.path.basename(fn), m.group(0)[:80]))
    assert not bad, bad
