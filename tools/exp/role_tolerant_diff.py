"""GPU box: Bell with ZH_PAINT_TOLERANT, role-wave form against lane form against the exact kernel -- where and by how much they differ."""
import os, sys
os.environ["ZH_ENV_LIVE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import zang_amd
from zang_amd import script, zang, zscript_native as native
from tests import util
ctx = zang_amd.default_context()
SCRIPT = open(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "script_modules.txt")).read()
V, F, SR = 1000, 1024, 48000.0
rng = np.random.default_rng(V + 5)
name = sys.argv[1] if len(sys.argv) > 1 else "Bell"
prog = script.ScriptProgram(SCRIPT, ctx, only=[name], forms=native.FORM_ROLES)
freq = util.dev(rng.uniform(40.0, 5000.0, V).astype(np.float32))
on = torch.from_numpy((rng.random(V) < 0.8).astype(np.uint8)).to(ctx.device)
base = torch.from_numpy(rng.uniform(40.0, 5000.0, (F, V)).astype(np.float32)).to(ctx.device)
res = {}
for label, roles, tol, ranges in (("lane tol", 0, True, None), ("role tol", 1, True, None), ("lane tol 1 range", 0, True, 0), ("lane exact", 0, False, None), ("role exact", 1, False, None)):
    rows = {"script_pc": roles}
    if ranges is not None:
        rows["script_ranges"] = ranges
    os.environ["ZH_FORMS"] = util.forms_env(**rows)["ZH_FORMS"]
    m = prog.module(name, V)
    out = torch.zeros_like(base)
    m.paint(zang.Span(0, F), [out], None, True, {"sample_rate": SR, "note_on": on, "freq": freq}, tolerant=tol)
    ctx.sync()
    print(label, ctx.last_form())
    res[label] = out.cpu().numpy()
    m.close()
def cmp(a, b):
    x, y = res[a], res[b]
    bad = np.argwhere(x.view(np.uint32) != y.view(np.uint32))
    print("%s vs %s: %d differ" % (a, b, len(bad)), [(int(f), int(v), float(x[f, v]), float(y[f, v])) for f, v in bad[:6]])
cmp("lane tol", "role tol"); cmp("lane tol", "lane tol 1 range"); cmp("role tol", "lane tol 1 range"); cmp("lane exact", "role exact")
d = np.abs(res["lane tol"].astype(np.float64) - res["lane exact"]).max(); print("lane tol vs exact max abs", d)
d = np.abs(res["role tol"].astype(np.float64) - res["lane exact"]).max(); print("role tol vs exact max abs", d)
