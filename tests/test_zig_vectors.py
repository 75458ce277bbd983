"""The oracle against vectors made by the REFERENCE itself (tools/zig_oracle/dump_vectors.zig, run by someone with a
Zig toolchain; the files land in tests/golden/zig/).  This is the pin SURVEY.md 8c says the reference lacks: until such
files are committed the oracle's parity stays "unpinned" and the second test skips.  The first test runs the whole
machinery -- reader, case table, comparison -- on oracle-made files so that it is known to work the day real files
arrive."""
import glob
import os

import numpy as np
import pytest

from tests import zig_vectors as zv

ZIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zig")


def test_machinery_on_oracle_made_files(tmp_path, oracle):
    names = zv.write_oracle_vectors(str(tmp_path))
    assert len(names) == 119
    for name in names:
        rec = zv.read(os.path.join(str(tmp_path), name + ".zgv"))
        report = zv.compare(name, rec)
        assert report and all(v == 1.0 for v in report.values()), (name, report)
    # a flipped bit is caught where bit-exactness is the bar, a small libm-sized error is tolerated where it is not
    rec = zv.read(os.path.join(str(tmp_path), "decimator_0.zgv"))
    rec["out"] = rec["out"].copy(); rec["out"].view(np.uint32)[300] ^= 1
    with pytest.raises(AssertionError):
        zv.compare("decimator_0", rec)
    rec = zv.read(os.path.join(str(tmp_path), "sineosc_cc.zgv"))
    rec["out"] = rec["out"].copy(); rec["out"].view(np.uint32)[300] ^= 1
    assert zv.compare("sineosc_cc", rec)["out"] < 1.0
    rec["out"][301] += np.float32(1e-3)
    with pytest.raises(AssertionError):
        zv.compare("sineosc_cc", rec)


def test_oracle_matches_reference_vectors(oracle):
    """Skips ONLY while tests/golden/zig/ is empty.  Anything in it makes this test binding: every file there must be a readable
    vector file of a known case, every case of the table must be there (a partial drop is a failure, with the missing names), and
    every record must agree with the oracle -- a disagreement fails, it never skips."""
    present = [f for f in (os.listdir(ZIG_DIR) if os.path.isdir(ZIG_DIR) else []) if not f.startswith(".")]
    if not present:
        pytest.skip("no reference-made vectors in tests/golden/zig/ (needs a Zig toolchain: tools/zig_oracle/README.md) -- parity unpinned")
    files = sorted(glob.glob(os.path.join(ZIG_DIR, "*.zgv")))
    stray = sorted(set(present) - {os.path.basename(f) for f in files})
    assert not stray, f"tests/golden/zig/ holds files that are not vector files: {stray}"
    want = set(zv.case_inputs())
    have = {os.path.splitext(os.path.basename(f))[0] for f in files}
    assert have == want, f"missing cases: {sorted(want - have)}; unknown cases: {sorted(have - want)}"
    inexact = {}
    for path in files:
        name = os.path.splitext(os.path.basename(path))[0]
        report = zv.compare(name, zv.read(path))
        for key, frac in report.items():
            if frac < 1.0:
                inexact[f"{name}.{key}"] = frac
    print("records not bit-identical (within 1e-5):", inexact)


def test_reference_vector_test_is_binding_once_files_exist(tmp_path, oracle, monkeypatch):
    """What test_oracle_matches_reference_vectors does the day tests/golden/zig/ is filled, tried on oracle-made files in a
    temporary directory: green on a complete, agreeing set; a FAILURE (never a skip) for one flipped bit, a missing case, a
    stray file."""
    import sys
    me = sys.modules[__name__]
    d = tmp_path / "zig"
    zv.write_oracle_vectors(str(d))
    monkeypatch.setattr(me, "ZIG_DIR", str(d))
    test_oracle_matches_reference_vectors(oracle)                               # complete and agreeing: passes
    # one flipped bit in a bit-exact record
    path = os.path.join(str(d), "math2.zgv")
    rec = zv.read(path)
    bad = dict(rec); bad["rare_white"] = rec["rare_white"].copy(); bad["rare_white"].view(np.uint32)[5] ^= 1
    zv.write(path, bad)
    with pytest.raises(AssertionError):
        test_oracle_matches_reference_vectors(oracle)
    zv.write(path, rec)
    # a libm record off by more than 1e-5
    bad = dict(rec); bad["pow2"] = rec["pow2"].copy(); bad["pow2"][100] *= np.float32(1.0001)
    zv.write(path, bad)
    with pytest.raises(AssertionError):
        test_oracle_matches_reference_vectors(oracle)
    zv.write(path, rec)
    test_oracle_matches_reference_vectors(oracle)
    # a missing case, then a stray file
    away = str(tmp_path / "math2.away")
    os.rename(path, away)
    with pytest.raises(AssertionError, match="missing cases"):
        test_oracle_matches_reference_vectors(oracle)
    os.rename(away, path)
    open(os.path.join(str(d), "notes.txt"), "w").write("x")
    with pytest.raises(AssertionError, match="not vector files"):
        test_oracle_matches_reference_vectors(oracle)
