#!/usr/bin/env python3
"""GPU box: every module kernel of SURVEY.md 8a timed on its own -- zero+paint of one 1024-frame buffer for V voices
(default 131,072), state carried, the K timed paints replayed as one hipGraph -- and, for the ones that read an
input image, the HBM bytes they move.  usage: tools/bench_modules.py [voices]  -> one line per module"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import zang_amd
from zang_amd import modules as mod, zang, workloads

V = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
F, SR, K = 1024, 48000.0, 20
torch.cuda.set_stream(torch.cuda.Stream())               # graph capture is not allowed on the default stream
ctx = zang_amd.Context(0)
dev = ctx.device
freq_h, color_h, u2, u3 = workloads.voice_params(5, 0, V)
freq, color = torch.from_numpy(freq_h).to(dev), torch.from_numpy(color_h).to(dev)
span = zang.Span(0, F)
# output images: a ring of 512 MiB (2 .. 32 images) so that a paint's stores go to HBM, not to a 256 MiB Infinity Cache that still holds
# the image from two paints ago (at 4,096 voices an image is 16 MiB)
NOUT = max(2, min(32, (512 << 20) // (F * V * 4)))
out = [ctx.image(F, V) for _ in range(NOUT)]
inp = ctx.image(F, V); inp.uniform_(-1.0, 1.0)
fbuf = ctx.image(F, V); fbuf.copy_(freq[None, :].expand(F, V))          # a frequency control image
pcm = torch.from_numpy(np.random.default_rng(1).integers(-20000, 20000, 48000, dtype=np.int16).view(np.uint8).copy()).to(dev)
cutoff = mod.Filter.cutoffFromFrequency(torch.from_numpy((200.0 + 7800.0 * u2)).to(dev), SR, ctx)
res = torch.from_numpy((0.9 * u3)).to(dev)
lin = lambda d: zang.PaintCurve.linear(d)
cases = []


def case(name, m, paint, reads=0):
    cases.append((name, m, paint, reads))


# basics.zig: pure bandwidth; `reads` = images read beside the one written (the destination of a `+=` counts)
inp2 = ctx.image(F, V); inp2.uniform_(-1.0, 1.0)
case("zero", None, lambda o: zang.zero(span, o, ctx), 0)
case("set (per-voice scalar)", None, lambda o: zang.set(span, o, color, ctx), 0)
case("copy", None, lambda o: zang.copy(span, o, inp, ctx), 1)
case("addInto (dest += a)", None, lambda o: zang.addInto(span, o, inp, ctx), 2)
case("add (dest += a + b)", None, lambda o: zang.add(span, o, inp, inp2, ctx), 3)
case("multiply (dest += a * b)", None, lambda o: zang.multiply(span, o, inp, inp2, ctx), 3)
case("multiplyWithScalar (dest *= s)", None, lambda o: zang.multiplyWithScalar(span, o, 0.5, ctx), 1)
m = mod.SineOsc(V, ctx); case("SineOsc const freq / const phase", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.constant(freq), zang.constant(0.0)), zero_first=True))
m = mod.SineOsc(V, ctx); case("SineOsc freq image", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), zang.constant(0.0)), zero_first=True), 1)
m = mod.SineOsc(V, ctx); case("SineOsc const / const, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.constant(freq), zang.constant(0.0)), zero_first=True, tolerant=True))
m = mod.SineOsc(V, ctx); case("SineOsc freq image, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), zang.constant(0.0)), zero_first=True, tolerant=True), 1)
m = mod.PulseOsc(V, ctx); case("PulseOsc const freq (chunked)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.constant(freq), color), zero_first=True))
m = mod.PulseOsc(V, ctx); case("PulseOsc freq image", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), color), zero_first=True), 1)
m = mod.TriSawOsc(V, ctx); case("TriSawOsc const freq (chunked)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.constant(freq), color), zero_first=True))
m = mod.TriSawOsc(V, ctx); case("TriSawOsc const freq, color 0 (sawtooth)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.constant(freq), 0.0), zero_first=True))
m = mod.TriSawOsc(V, ctx); case("TriSawOsc freq image", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.buffer(fbuf), color), zero_first=True), 1)
if V <= 16384:
    m = mod.Noise(V, ctx); case("Noise pink, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(m.pink), zero_first=True, tolerant=True))
m = mod.Noise(V, ctx); case("Noise white", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(m.white), zero_first=True))
m = mod.Noise(V, ctx); case("Noise pink", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(m.pink), zero_first=True))
m = mod.Envelope(V, ctx)
_env_k = [0]
def _env(o, m=m):
    k = _env_k[0] % 8; _env_k[0] += 1
    m.paint(span, [o], [], k == 0, m.Params(SR, zang.PaintCurve.cubed(0.01), zang.PaintCurve.cubed(0.1), zang.PaintCurve.cubed(0.05), 0.8, k < 4), zero_first=True)
case("Envelope (cubed, note on 4 buffers / off 4)", m, _env)
m = mod.Gate(V, ctx); case("Gate", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(True), zero_first=True))
cbuf = ctx.image(F, V); cbuf.copy_((cutoff[None, :] * (0.5 + 0.5 * torch.linspace(0, 1, F, device=dev)[:, None])).expand(F, V))      # a cutoff sweep (control image)
m = mod.Filter(V, ctx); case("Filter low-pass, cutoff image", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, m.low_pass, zang.buffer(cbuf), zang.constant(res)), zero_first=True), 2)
m = mod.Filter(V, ctx); case("Filter low-pass, const cutoff / res", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, m.low_pass, zang.constant(cutoff), zang.constant(res)), zero_first=True), 1)
m = mod.Sampler(V, ctx); smp = m.Sample(1, 44100, m.signed16_lsb, pcm)
case("Sampler s16 mono, resampled, loop", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, smp, 0, True), zero_first=True))
m = mod.Decimator(V, ctx); case("Decimator", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, inp, 6000.0), zero_first=True), 1)
m = mod.Distortion(V, ctx); case("Distortion overdrive", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, m.overdrive, 0.5, 0.5, 0.0), zero_first=True), 1)
m = mod.Distortion(V, ctx); case("Distortion clip", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, m.clip, 0.5, 0.5, 0.0), zero_first=True), 1)
m = mod.NiceInstrument(V, color, ctx); case("NiceInstrument (fused)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, freq, True), zero_first=True))
if V <= 16384:
    m = mod.NiceInstrument(V, color, ctx); case("NiceInstrument, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, freq, True), zero_first=True, tolerant=True))
rel = torch.full((V,), 0.3, dtype=torch.float32, device=dev)
m = mod.PMOscInstrument(V, rel, ctx); case("PMOscInstrument (fused)", m, lambda o, m=m: m.paint(span, [o], None, False, m.Params(SR, freq, True), zero_first=True))
m = mod.PMOscInstrument(V, rel, ctx); case("PMOscInstrument, ZH_PAINT_TOLERANT (carrier)", m, lambda o, m=m: m.paint(span, [o], None, False, m.Params(SR, freq, True), zero_first=True, tolerant=True))
if V <= 16384:
    m = mod.Filter(V, ctx); case("Filter low-pass, cutoff image, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, m.low_pass, zang.buffer(cbuf), zang.constant(res)), zero_first=True, tolerant=True), 2)
    m = mod.Filter(V, ctx); case("Filter low-pass const, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, m.low_pass, zang.constant(cutoff), zang.constant(res)), zero_first=True, tolerant=True), 1)

m = mod.SimpleDelay(V, 300, ctx); case("SimpleDelay(300)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp), zero_first=True), 3)
m = mod.FilteredEchoes(V, 300, ctx); case("FilteredEchoes(300)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, 0.6, 0.1), zero_first=True), 3)
if V <= 16384:
    m = mod.FilteredEchoes(V, 600, ctx); case("FilteredEchoes(600)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, 0.6, 0.1), zero_first=True), 3)
    m = mod.FilteredEchoes(V, 600, ctx); case("FilteredEchoes(600), ZH_PAINT_TOLERANT (two pieces)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, 0.6, 0.1), zero_first=True, tolerant=True), 3)
    m = mod.FilteredEchoes(V, 15000, ctx); case("FilteredEchoes(15000)", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, 0.6, 0.1), zero_first=True), 3)
    m = mod.FilteredEchoes(V, 15000, ctx); case("FilteredEchoes(15000), ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(inp, 0.6, 0.1), zero_first=True, tolerant=True), 3)
crv = torch.tensor([[0.0, 0.0], [1.0, 0.005], [0.3, 0.012], [0.8, 0.02], [0.0, 0.05]], dtype=torch.float32, device=dev)
m = mod.Curve(V, ctx)
_crv_k = [0]
def _curve(o, m=m):
    k = _crv_k[0] % 4; _crv_k[0] += 1
    m.paint(span, [o], [], k == 0, m.Params(SR, m.smoothstep, crv), zero_first=True)
case("Curve smoothstep, 5 nodes, retrigger every 4", m, _curve)
m = mod.Cycle(V, ctx); case("Cycle const speed", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.constant(3.0)), zero_first=True))
m = mod.Portamento(V, ctx); case("Portamento cubed", m, lambda o, m=m: m.paint(span, [o], [], False, m.Params(SR, zang.PaintCurve.cubed(0.2), freq, True, True), zero_first=True))

# generated script kernels (tests/golden/script_modules.txt): compiled at run time through hiprtc
from zang_amd import script as zscript
_prog = zscript.ScriptProgram(open(os.path.join(ROOT, "tests", "golden", "script_modules.txt")).read(), ctx, only=["Pluck", "CycleSine", "FilteredSawtooth", "HardSquare"])
on_dev = torch.ones(V, dtype=torch.uint8, device=dev)
off_dev = torch.zeros(V, dtype=torch.uint8, device=dev)
m = _prog.module("Pluck", V, 0)
_pl_k = [0]
def _pluck(o, m=m):
    k = _pl_k[0] % 8; _pl_k[0] += 1
    m.paint(span, [o], None, k == 0, {"sample_rate": SR, "freq": freq, "note_on": on_dev if k < 4 else off_dev}, zero_first=True)
case("script Pluck (note on 4 buffers / off 4)", m, _pluck)
m = _prog.module("Pluck", V, 0)
_pt_k = [0]
def _pluck_tol(o, m=m):
    k = _pt_k[0] % 8; _pt_k[0] += 1
    m.paint(span, [o], None, k == 0, {"sample_rate": SR, "freq": freq, "note_on": on_dev if k < 4 else off_dev}, zero_first=True, tolerant=True)
case("script Pluck, ZH_PAINT_TOLERANT", m, _pluck_tol)
# the reference's remaining composite recipes as generated kernels (examples/modules.zig:130-187, 250-289; tests/test_gpu_script_composites.py)
m = _prog.module("FilteredSawtooth", V, 0)
_fs_k = [0]
def _fsaw(o, m=m):
    k = _fs_k[0] % 8; _fs_k[0] += 1
    m.paint(span, [o], None, k == 0, {"sample_rate": SR, "freq": freq, "note_on": on_dev if k < 4 else off_dev, "cutoff": 0.07}, zero_first=True)
case("script FilteredSawtooth (reference recipe)", m, _fsaw)
m = _prog.module("HardSquare", V, 0)
_hs_k = [0]
def _hsq(o, m=m):
    k = _hs_k[0] % 8; _hs_k[0] += 1
    m.paint(span, [o], None, k == 0, {"sample_rate": SR, "freq": freq, "note_on": on_dev if k < 4 else off_dev}, zero_first=True)
case("script HardSquare (reference recipe)", m, _hsq)
m = _prog.module("CycleSine", V, 0); case("script CycleSine (sin of Cycle + phase)", m, lambda o, m=m: m.paint(span, [o], None, False, {"sample_rate": SR, "freq": 3.0, "phase": 0.25}, zero_first=True))
m = _prog.module("CycleSine", V, 0); case("script CycleSine, ZH_PAINT_TOLERANT", m, lambda o, m=m: m.paint(span, [o], None, False, {"sample_rate": SR, "freq": 3.0, "phase": 0.25}, zero_first=True, tolerant=True))

print("# %d voices x %d frames per paint, %d paints per graph, one MI355X" % (V, F, K))
print("%-46s %10s %12s %10s" % ("module", "us/paint", "v-samples/s", "HBM TB/s"))
ONLY = os.environ.get("ZH_BENCH_ONLY", "")                  # substring filter on the case names, "|"-separated
for name, m, paint, reads in cases:
    if ONLY and not any(k in name for k in ONLY.split("|")):
        continue
    for i in range(4):
        paint(out[i % NOUT])
    ctx.sync()
    if os.environ.get("ZH_BENCH_EAGER") == "1":            # counter collection (rocprofv3 --pmc) wants plain launches
        t0 = time.perf_counter()
        for i in range(K):
            paint(out[i % NOUT])
        ctx.sync(); dt = time.perf_counter() - t0
    else:
        g = ctx.capture(lambda: [paint(out[i % NOUT]) for i in range(K)])
        g.launch(); ctx.sync()
        t0 = time.perf_counter(); g.launch(); ctx.sync(); dt = time.perf_counter() - t0
        g.close()
    us = dt * 1e6 / K
    print("%-46s %10.1f %12.3e %10.2f" % (name, us, V * F / (us * 1e-6), (1 + reads) * V * F * 4 / (us * 1e-6) / 1e12))
