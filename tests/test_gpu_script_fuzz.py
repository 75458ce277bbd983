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
def test_gpu_seed_1015_select_hazard(ctx):
    """The case that exposed the inline-asm v_cndmask of round 2 (lanes.hip.h zsel_hard): TriSawOsc + pink Noise + Envelope
    in one kernel put the compare that writes the mask SGPRs right before the asm, which the hazard recognizer does not
    see into; lanes 32-63 of the first frame of every chunk took the stale mask."""
    script_fuzz.run_case(ctx, 1015)
    script_fuzz.run_case(ctx, 1015, F=256, ranges=3)


def test_no_inline_asm_reads_an_sgpr():
    """gfx950 needs wait states between a VALU that writes an SGPR and a VALU that reads it; the compiler inserts them for
    its own instructions only.  No inline asm in the kernels may take a scalar ("s") operand."""
    import glob, os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zang_amd", "csrc")
    bad = []
    for fn in glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.hip")):
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\((.*?)\);', open(fn).read(), re.S):
            if re.search(r'"[=+]?s"\s*\(', m.group(1)):
                bad.append((os.path.basename(fn), m.group(0)[:80]))
    assert not bad, bad
