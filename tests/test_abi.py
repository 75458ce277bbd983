"""CPU: the C-ABI library loads without a GPU and exports exactly what include/zang_hip.h
declares; the ctypes mirror (zang_amd/abi.py) lists the same set.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zang_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"ZH_API\s+[\w\s\*]+?\b(zh_\w+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 95
    for module in ("sineosc", "pulseosc", "trisawosc", "noise", "envelope", "gate", "filter", "sampler", "decimator", "distortion"):
        assert f"zh_{module}_paint" in syms and f"zh_{module}_create" in syms
    for op in ("zero", "set", "copy", "add", "add_into", "add_scalar", "add_scalar_into", "multiply",
               "multiply_with", "multiply_scalar", "multiply_with_scalar", "mixdown_voices"):
        assert f"zh_{op}" in syms


def test_library_exports_every_declared_symbol():
    from zang_amd import abi
    assert os.path.exists(abi.LIB_PATH), "build libzang_hip.so first (__graft_entry__.build())"
    lib = C.CDLL(abi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.zh_version is not None


def test_ctypes_mirror_matches_header():
    from zang_amd import abi
    assert sorted(abi.SIGNATURES) == declared_symbols()


def test_no_undeclared_exports():
    from zang_amd import abi
    out = subprocess.run(["nm", "-D", "--defined-only", abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("zh_")})
    assert exported == declared_symbols()


def test_struct_sizes_match_c_layout():
    """sizeof() of every ABI struct as the C compiler sees it == the ctypes mirror."""
    from zang_amd import abi
    names = {"zh_buf": abi.Buf, "zh_f32": abi.F32, "zh_bool": abi.Bool, "zh_cob": abi.Cob, "zh_curve": abi.Curve,
             "zh_sineosc_params": abi.SineOscParams, "zh_pulseosc_params": abi.PulseOscParams,
             "zh_trisawosc_params": abi.TriSawOscParams, "zh_noise_params": abi.NoiseParams, "zh_noise_state": abi.NoiseState,
             "zh_envelope_params": abi.EnvelopeParams, "zh_envelope_state": abi.EnvelopeState, "zh_gate_params": abi.GateParams,
             "zh_filter_params": abi.FilterParams, "zh_sample": abi.Sample, "zh_sampler_params": abi.SamplerParams,
             "zh_decimator_params": abi.DecimatorParams, "zh_distortion_params": abi.DistortionParams,
             "zh_nice_params": abi.NiceParams, "zh_nice_state": abi.NiceState, "zh_pmosc_params": abi.PMOscParams,
             "zh_pmosc_state": abi.PMOscState, "zh_trisawosc_state": abi.TriSawOscState, "zh_script_param": abi.ScriptParam}
    prog = '#include <stdio.h>\n#include "zang_hip.h"\nint main(void){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c"); exe = os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        n, sz = line.split()
        assert C.sizeof(names[n]) == int(sz), (n, C.sizeof(names[n]), sz)


def test_product_has_no_oracle_or_cpu_path():
    """zang_amd must never import the oracle (it is test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zang_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hip.h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "zang_oracle" not in text and "zmath_ref" not in text, f


def test_context_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zang_amd
    with pytest.raises(zang_amd.abi.ZangHipError):
        zang_amd.Context(0)


def test_noise_jump_tables_selftest():
    """Host-only: the xoshiro256++ jump tables behind the frame-range form of white Noise (csrc/noise_jump.hip) -- table j
    applied to a state equals 32 (j + 1) sequential transitions, for all 63 tables and 64 random states."""
    from zang_amd import abi
    assert abi.load().zh_selftest_noise_jump(20260517, 64) == 0
