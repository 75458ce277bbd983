"""GPU: one tests/script_fuzz.py seed with and without ZH_PAINT_TOLERANT on twin instances; where the two differ most.
usage: script_tolerant_case.py SEED [F [ranges]]"""
import os, sys
os.environ["ZH_ENV_LIVE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import zang_amd
from tests import script_fuzz as sf
from tests.util import from_image, to_image
from zang_amd import script, zang
seed = int(sys.argv[1]); F = int(sys.argv[2]) if len(sys.argv) > 2 else sf.F
if len(sys.argv) > 3:
    os.environ["ZH_SCRIPT_RANGES"] = sys.argv[3]
ctx = zang_amd.default_context()
text, name = sf.generate(seed)
prog = script.ScriptProgram(text, ctx, only=[name])
a, b = prog.module(name, sf.V, seed), prog.module(name, sf.V, seed)
order = [p[0] for p in a.params]
for buf in range(2):
    ia, ib = to_image(np.zeros((sf.V, F), np.float32)), to_image(np.zeros((sf.V, F), np.float32))
    for start, end, nic, params in sf.schedule(seed * 16 + buf, F):
        dev = {k: sf._device_value(v) for k, v in params.items() if k in order}
        nd = torch.from_numpy(nic.astype(np.uint8)).cuda() if isinstance(nic, np.ndarray) else nic
        a.paint(zang.Span(start, end), [ia], None, nd, dev)
        b.paint(zang.Span(start, end), [ib], None, nd, dev, tolerant=True)
        ctx.sync()
        x, y = from_image(ia).astype(np.float64), from_image(ib).astype(np.float64)
        d = np.where(np.isfinite(x) & np.isfinite(y), np.abs(x - y), 0.0)
        v, f = np.unravel_index(np.argmax(d), d.shape)
        print("buffer", buf, "span", (start, end), "max diff %.3e at voice %d frame %d: exact %r tolerant %r; peak %.3e" % (d[v, f], v, f, x[v, f], y[v, f], np.abs(np.where(np.isfinite(x), x, 0))[v].max()))
        for k, val in params.items():
            if k in order:
                pv = sf._per_voice(val, v)
                print("    ", k, "=", (pv[f] if isinstance(pv, np.ndarray) else pv))
i = prog.hip_source.find("zs_paint_" + name)
print(prog.hip_source[i:i + 200])
for l in prog.hip_source.splitlines():
    if "ZS_T ?" in l:
        print(l.strip())
