"""Documentation drift: every ZH_* environment switch the sources read is described in INTEGRATION.md, and every one it
describes is still read somewhere."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources():
    text = ""
    for pat in ("zang_amd/csrc/*.hip", "zang_amd/csrc/*.h", "zang_amd/csrc/*.hpp", "zang_amd/*.py", "bench.py", "include/*", "bindings/*"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            if os.path.isfile(f):
                text += open(f, errors="ignore").read()
    return text


def test_environment_switches_are_documented():
    doc = set(re.findall(r"ZH_[A-Z0-9_]+", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    src = _sources()
    read = set(re.findall(r'getenv\("(ZH_[A-Z0-9_]+)"\)', src)) | set(re.findall(r'environ(?:\.get\(|\[)"(ZH_[A-Z0-9_]+)"', src))
    assert read, "no getenv found: the scan is broken"
    undocumented = sorted(read - doc)
    assert not undocumented, f"read but not in INTEGRATION.md: {undocumented}"
    stale = sorted(x for x in doc if x not in src)
    assert not stale, f"in INTEGRATION.md but read nowhere: {stale}"


def test_dispatch_table_in_integration_md_is_the_librarys():
    """INTEGRATION.md's form table is GENERATED from the library's own rows (tools/gen_form_docs.py, csrc/dispatch.hip)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_form_docs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "INTEGRATION.md's dispatch table is stale: run tools/gen_form_docs.py\n" + r.stderr[-500:]


def test_at_most_thirty_environment_switches_in_the_library():
    """the library read 50 environment switches; it reads a handful now and one table."""
    names = set()
    for f in glob.glob(os.path.join(ROOT, "zang_amd", "csrc", "*")):
        if os.path.isfile(f) and f.endswith((".hip", ".h", ".hpp")):
            names |= set(re.findall(r'"(ZH_[A-Z0-9_]+)"', open(f, errors="ignore").read()))
    assert len(names) <= 30, sorted(names)


def test_design_and_readme_are_pages_not_notebooks():
    """DESIGN.md is the CURRENT state in at most 400 lines of at most 160 columns (history lives in
    profiles/rNN/NOTES.md), README.md a page with tables."""
    for name, max_lines in (("DESIGN.md", 400), ("README.md", 120)):
        lines = open(os.path.join(ROOT, name)).read().split("\n")
        assert len(lines) <= max_lines, (name, len(lines))
        wide = [(i + 1, len(ln)) for i, ln in enumerate(lines) if len(ln) > 160]
        assert not wide, (name, wide[:5])


def test_every_file_design_md_cites_exists():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = []
    for m in re.finditer(r"`([A-Za-z0-9_./*{},-]+)`", text):
        tok = m.group(1)
        if "/" not in tok and not tok.endswith((".md", ".py", ".txt", ".json", ".csv", ".hip", ".h", ".hpp", ".zig", ".c", ".cpp", ".so")):
            continue
        if tok.startswith(("src/", "examples/")) or "rNN" in tok or tok.startswith("/"):
            continue                                              # reference paths, placeholders
        tok = tok.split("::")[0]
        cands = [tok, os.path.join("profiles", "r06", tok), os.path.join("profiles", "r05", tok), os.path.join("zang_amd", "csrc", tok), os.path.join("zang_amd", tok),
                 os.path.join("tests", tok), os.path.join("tools", tok), os.path.join("include", tok)]
        if "{" in tok:                                            # zang_amd/{abi,zang,...}.py
            head, rest = tok.split("{", 1)
            alts, tail = rest.split("}", 1)
            ok = all(glob.glob(os.path.join(ROOT, head + a + tail)) or glob.glob(os.path.join(ROOT, "profiles", "r06", head + a + tail)) or
                     glob.glob(os.path.join(ROOT, "profiles", "r05", head + a + tail)) for a in alts.split(","))
        else:
            ok = any(glob.glob(os.path.join(ROOT, c)) for c in cands)
        if not ok:
            missing.append(tok)
    assert not missing, sorted(set(missing))


def test_inline_assembly_memory_instructions_carry_their_wait_states():
    """The compiler's hazard recognizer does not look inside inline assembly: on gfx940+ a store of more than 64 bits needs two wait
    states before a VALU write of its data registers (round 5: store4_sc1 without them sent the NEXT instruction's result out as the first
    component, in one kernel shape out of three).  Every inline-assembly store / load in csrc/ must be followed by an s_nop in the same
    statement."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zang_amd", "csrc")
    bad = []
    for path in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        for m in re.finditer(r'asm\s+volatile\s*\(\s*"([^"]*)"', open(path).read()):
            text = m.group(1)
            if re.search(r"\b(global|buffer|flat|scratch)_(store|load|atomic)", text) and "s_nop" not in text:
                bad.append((os.path.basename(path), text[:60]))
    assert not bad, bad


def test_entry_point_and_form_counts_in_the_docs_are_the_headers():
    """DESIGN.md said 214 entry points where the header had 215.  The counts quoted in DESIGN.md and README.md are
    checked against include/zang_hip.h and the library's own dispatch table."""
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "zang_hip.h")).read(), flags=re.S)
    n_api = len(re.findall(r"\bZH_API\b[^;{]*\(", text))
    from zang_amd import abi
    n_forms = abi.load().zh_form_count()
    for name in ("DESIGN.md", "README.md"):
        doc = open(os.path.join(ROOT, name)).read()
        for m in re.finditer(r"(\d+) entry points", doc):
            assert int(m.group(1)) == n_api, (name, m.group(0), n_api)
        for m in re.finditer(r"\((\d+) rows", doc):
            assert int(m.group(1)) == n_forms, (name, m.group(0), n_forms)
