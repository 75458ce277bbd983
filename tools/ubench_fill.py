"""GPU-box micro-benchmark: how long does a pure 16 MiB store kernel take?  Run under rocprofv3."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, zang_amd
from zang_amd import zang
ctx = zang_amd.Context(0)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = 1024
ring = [ctx.image(F, V) for _ in range(max(2, (512 << 20) // (V * F * 4)))]
sp = zang.Span(0, F)
for it in range(200):
    zang.zero(sp, ring[it % len(ring)], ctx=ctx)
for it in range(200):
    zang.set(sp, ring[it % len(ring)], 1.5, ctx=ctx)
for it in range(200):
    ring[it % len(ring)].zero_()
for it in range(200):
    zang.addScalarInto(sp, ring[it % len(ring)], 1.0, ctx=ctx)
ctx.sync()
