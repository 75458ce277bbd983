import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zang_amd
from zang_amd import song
text = open(sys.argv[1]).read()
ctx = zang_amd.Context(0)
r = song.SongRenderer(text, ctx); r.render(2.0, batch=128); ctx.sync()
for batch in (32, 128, 512):
    r = song.SongRenderer(text, ctx)
    total = 60 * 48000
    counts = [1024] * (total // 1024)
    groups = [counts[i:i + batch] for i in range(0, len(counts), batch)]
    tp = tl = tc = 0.0
    t00 = time.perf_counter()
    for g in groups:
        t0 = time.perf_counter(); p = r._prepare_batch(g); t1 = time.perf_counter()
        l = r._launch_batch(p); t2 = time.perf_counter()
        r._collect_batch(l); t3 = time.perf_counter()
        tp += t1 - t0; tl += t2 - t1; tc += t3 - t2
    print("batch %d: serial total %.3f s: prepare(host) %.3f, launch %.3f, collect(wait+copy) %.3f" % (batch, time.perf_counter() - t00, tp, tl, tc))
    r = song.SongRenderer(text, ctx)
    t0 = time.perf_counter(); r.render(60.0, batch=batch); print("   pipelined render: %.3f s" % (time.perf_counter() - t0))
