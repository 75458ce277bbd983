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
