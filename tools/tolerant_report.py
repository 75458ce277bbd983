#!/usr/bin/env python3
"""GPU: how far the ZH_PAINT_TOLERANT forms (csrc/filter_tp.hip.h) are from the oracle, case by case -- the numbers profiles/r04/NOTES.md 5a
quotes (profiles/r04/tolerant_error.txt).  Per case: the worst voice's max |gpu - oracle| over the span relative to that voice's
peak (the form's contract: <= 1e-5), the median voice, the share of samples inside tests/util.py's PER-SAMPLE metric
|err| <= 1e-5 max(|ref|, 1e-3) -- which no re-association of an f32 recurrence meets near zero crossings: the counter-example the
verdict asked to have written down -- and the worst such sample (reference value, error)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

SR, F = 48000.0, 1024


def main():
    import zang_amd
    from zang_amd import modules as mod, zang
    from oracle import pyoracle as oracle
    from tests import util
    ctx = zang_amd.default_context()
    L = oracle.lib()
    print("%-58s %11s %11s %9s   %s" % ("case", "worst/peak", "median", "inside %", "worst sample outside the per-sample metric (ref, |err|, allowed)"))

    def show(name, got, ref, extra=None):
        ratio, dis, inside = util.peak_relative_error(got, ref, scale_extra=extra)
        g = got.astype(np.float64); r = ref.astype(np.float64)
        with np.errstate(invalid="ignore"):
            err = np.abs(g - r); tol = util.RTOL * np.maximum(np.abs(r), util.FLOOR)
            over = np.where(np.isfinite(err), err / tol, 0.0)
        k = np.unravel_index(np.argmax(over), over.shape)
        print("%-58s %11.2e %11.2e %9.3f   ref % .3e  err %.2e  allowed %.2e  (%d finite/non-finite disagreements)" %
              (name, ratio.max(), np.median(ratio), inside * 100, r[k], err[k], tol[k], dis))

    def filter_case(name, V, ftype, cut, res, inp, l0=None, b0=None, paints=2):
        m = mod.Filter(V, ctx)
        sts = []
        for v in range(V):
            st = oracle.Filter(); L.zo_filter_init(C.byref(st))
            if l0 is not None:
                st.l, st.b = float(l0[v]), float(b0[v])
            sts.append(st)
        if l0 is not None:
            s = m.state(); s["l"] = l0; s["b"] = b0; m.set_state(s)
        gi = util.to_image(inp); dc, dr = util.dev(cut), util.dev(res)
        for p in range(paints):
            ref = np.zeros((V, F), np.float32)
            for v in range(V):
                L.zo_filter_paint(C.byref(sts[v]), 0, F, oracle.fptr(ref[v]), oracle.fptr(inp[v]), ftype, oracle.constant(cut[v]), oracle.constant(res[v]))
            out = ctx.image(F, V)
            m.paint(zang.Span(0, F), [out], [], False, m.Params(gi, ftype, zang.constant(dc), zang.constant(dr)), zero_first=True, tolerant=True)
            ctx.sync()
            rl = np.array([t.l for t in sts], np.float32); rb = np.array([t.b for t in sts], np.float32)
            show(f"Filter {name}, paint {p + 1}", util.from_image(out), ref, np.maximum(np.abs(rl), np.abs(rb)))
            s = m.state(); s["l"] = rl; s["b"] = rb; m.set_state(s)

    rng = np.random.default_rng(3)
    cfg3 = lambda V: (np.array([L.zo_filter_cutoff_from_frequency(float(200.0 + 7800.0 * u), SR) for u in rng.random(V)], np.float32), (0.9 * rng.random(V)).astype(np.float32))
    for V in (4096, 16384):
        c, r = cfg3(V)
        filter_case(f"config 3 parameters, {V} voices, low-pass", V, 1, c, r, util.rng_buffers(7, V, F), paints=1)
    V = 1024
    c, r = cfg3(V)
    x = util.rng_buffers(8, V, F)
    filter_case("high-pass", V, 3, c, r, x, paints=1)
    filter_case("all-pass", V, 5, c, r, x, paints=1)
    filter_case("res 0.9 everywhere, band-pass", V, 2, c, np.full(V, 0.9, np.float32), x, paints=1)
    lowc = np.array([L.zo_filter_cutoff_from_frequency(float(50.0 + 500.0 * u), SR) for u in rng.random(V)], np.float32)
    filter_case("res 1.0 (no damping), 50-550 Hz", V, 1, lowc, np.ones(V, np.float32), x, paints=1)
    filter_case("cutoff 1.0, res 0-1, all-pass", V, 5, np.ones(V, np.float32), rng.random(V).astype(np.float32), x, paints=1)
    filter_case("cutoff 1e-4", V, 1, np.full(V, 1e-4, np.float32), np.full(V, 0.7, np.float32), x, paints=1)
    filter_case("inputs x 1e-30 (the dc offset alone)", V, 1, c, r, (x * np.float32(1e-30)).astype(np.float32))
    filter_case("inputs x 1e+30", V, 1, c, r, (x * np.float32(1e30)).astype(np.float32), paints=1)
    filter_case("states of 1e30 at span start", V, 1, c, r, x, l0=(rng.uniform(-1, 1, V) * 1e30).astype(np.float32), b0=(rng.uniform(-1, 1, V) * 1e30).astype(np.float32), paints=1)

    # the fused voice of config 3
    for V in (4096, 16384):
        c, r = cfg3(V)
        m = mod.NoiseFilter(V, ctx, first_seed=0)
        nzs, fls = [], []
        for v in range(V):
            nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), v); nzs.append(nz)
            fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
        temp = np.zeros(F, np.float32)
        gc, gr = util.dev(c), util.dev(r)
        for p in range(3):
            ref = np.zeros((V, F), np.float32)
            for v in range(V):
                L.zo_zero(0, F, oracle.fptr(temp))
                L.zo_noise_paint(C.byref(nzs[v]), 0, F, oracle.fptr(temp), 0)
                L.zo_filter_paint(C.byref(fls[v]), 0, F, oracle.fptr(ref[v]), oracle.fptr(temp), 1, oracle.constant(c[v]), oracle.constant(r[v]))
            out = ctx.image(F, V)
            m.paint(zang.Span(0, F), [out], None, False, m.Params(0, 1, gc, gr), zero_first=True, tolerant=True)
            ctx.sync()
            show(f"Noise->Filter, {V} voices, buffer {p + 1} (states carried by the GPU)", util.from_image(out), ref)

    # pink Noise (Kellett's taps as chunks at once over exact white noise)
    for V in (4096, 16384):
        m = mod.Noise(V, ctx, first_seed=0)
        nzs = []
        for v in range(V):
            nz = oracle.Noise(); L.zo_noise_init(C.byref(nz), v); nzs.append(nz)
        for p in range(2):
            ref = np.zeros((V, F), np.float32)
            for v in range(V):
                L.zo_noise_paint(C.byref(nzs[v]), 0, F, oracle.fptr(ref[v]), 1)
            out = ctx.image(F, V)
            m.paint(zang.Span(0, F), [out], None, False, m.Params(m.pink), zero_first=True, tolerant=True)
            ctx.sync()
            show(f"Noise pink, {V} voices, buffer {p + 1}", util.from_image(out), ref)

    # NiceInstrument at few voices: a note script, states carried by the GPU
    for V in (4096, 16384):
        from zang_amd import workloads
        freq, color, _, _ = workloads.voice_params(5, 0, V)
        idx = np.arange(0, V, max(1, V // 512))
        sts = []
        for v in idx:
            st = oracle.NiceInstrument(); L.zo_nice_init(C.byref(st), float(color[v])); sts.append(st)
        m = mod.NiceInstrument(V, util.dev(color), ctx)
        gf = util.dev(freq)
        t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
        for k, (on, nic, what) in enumerate([(1, 1, "attack + decay"), (1, 0, "sustain"), (0, 0, "release"), (0, 0, "release, idle"), (1, 1, "retrigger")]):
            ref = np.zeros((len(idx), F), np.float32)
            for q, v in enumerate(idx):
                L.zo_nice_paint(C.byref(sts[q]), 0, F, oracle.fptr(ref[q]), oracle.fptr(t0), oracle.fptr(t1), nic, SR, float(freq[v]), on)
            out = ctx.image(F, V)
            m.paint(zang.Span(0, F), [out], None, bool(nic), m.Params(SR, gf, bool(on)), zero_first=True, tolerant=True)
            ctx.sync()
            rl = np.array([r.flt.l for r in sts], np.float32); rb = np.array([r.flt.b for r in sts], np.float32)
            show(f"NiceInstrument, {V} voices, buffer {k + 1}: {what} (states carried by the GPU)", util.from_image(out)[idx], ref, np.maximum(np.abs(rl), np.abs(rb)))


    # FilteredEchoes: the ring carries a buffer's error into later ones (times the feedback, through the filter): 40 buffers in a
    # row on the GPU's own ring and filter state, the worst buffers shown
    for V, D, fbv in ((4096, 15000, 0.6), (1024, 600, 0.9), (1024, 2000, 0.99)):
        rngf = np.random.default_rng(D)
        fb = np.full(V, fbv, np.float32); cutoff = rngf.uniform(0.05, 1.0, V).astype(np.float32)
        idx = np.arange(0, V, max(1, V // 256))
        rings = np.zeros((len(idx), D), np.float32)
        ds, fls = [], []
        for q in range(len(idx)):
            d = oracle.Delay(); L.zo_delay_init(C.byref(d), oracle.fptr(rings[q]), D); ds.append(d)
            fl = oracle.Filter(); L.zo_filter_init(C.byref(fl)); fls.append(fl)
        m = mod.FilteredEchoes(V, D, ctx)
        gfb, gc = util.dev(fb), util.dev(cutoff)
        t0 = np.zeros(F, np.float32); t1 = np.zeros(F, np.float32)
        worst = []
        for k in range(40):
            x = util.rng_buffers(900 + k, V, F) if k < 30 else np.zeros((V, F), np.float32)     # then the echoes ring out
            ref = np.zeros((len(idx), F), np.float32)
            for q, v in enumerate(idx):
                L.zo_filtered_echoes_paint(C.byref(ds[q]), C.byref(fls[q]), 0, F, oracle.fptr(ref[q]), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(x[v]), float(fb[v]), float(cutoff[v]))
            out = ctx.image(F, V)
            m.paint(zang.Span(0, F), [out], None, False, m.Params(util.to_image(x), gfb, gc), zero_first=True, tolerant=True)
            ctx.sync()
            rl = np.array([t.l for t in fls], np.float32); rb = np.array([t.b for t in fls], np.float32)
            got = util.from_image(out)[idx]
            ratio, _, _ = util.peak_relative_error(got, ref, scale_extra=np.maximum(np.abs(rl), np.abs(rb)))
            worst.append((float(ratio.max()), k, got, ref, np.maximum(np.abs(rl), np.abs(rb))))
        for w, k, got, ref, extra in sorted(worst, key=lambda t: -t[0])[:2] + [worst[-1]]:
            show(f"FilteredEchoes({D}), feedback {fbv}, {V} voices, buffer {k + 1} of 40", got, ref, extra)


if __name__ == "__main__":
    main()
