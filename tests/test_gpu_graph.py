"""GPU: hipGraph capture/replay of paint sequences (what bench.py times) gives the same bits and the
same carried state as eager launches; and the bench workload's last image equals the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 48000.0
F = 1024


def test_graph_replay_equals_eager(ctx):
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V, G = 1024, 4
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)                      # binds to the side stream (capture needs a non-default stream)
        fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
        me, mg = mod.PulseOsc(V, c2), mod.PulseOsc(V, c2)
        flt_e, flt_g = mod.Filter(V, c2), mod.Filter(V, c2)
        ring_e = [c2.image(F, V) for _ in range(G)]; ring_g = [c2.image(F, V) for _ in range(G)]
        tmp_e, tmp_g = c2.image(F, V), c2.image(F, V)
        sp = zang.Span(0, F)

        def steps(m, flt, ring, tmp):
            for o in ring:                            # PulseOsc -> temp -> Filter(low_pass) -> ring image
                m.paint(sp, [tmp], [], False, m.Params(SR, zang.constant(fr), col), zero_first=True)
                flt.paint(sp, [o], [], False, flt.Params(tmp, flt.low_pass, zang.constant(0.3), zang.constant(0.5)), zero_first=True)

        for _ in range(3):
            steps(me, flt_e, ring_e, tmp_e)           # 12 eager buffers
        steps(mg, flt_g, ring_g, tmp_g)               # 4 eager (also warms up), then capture 4 and replay twice
        c2.sync()
        g = c2.capture(lambda: steps(mg, flt_g, ring_g, tmp_g))
        g.launch(); g.launch()
        c2.sync()
        for a, b in zip(ring_e, ring_g):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert np.array_equal(me.state(), mg.state()) and np.array_equal(flt_e.state(), flt_g.state())
        g.close()
        c2.close()


def test_bench_workload_last_image_matches_oracle(ctx, oracle):
    """The exact step bench.py times (zero+paint of 4096 PulseOsc voices, graph replay over a ring):
    after K steps the most recent image equals the oracle's K-th buffer."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V, K, R = 4096, 12, 4
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        m = mod.PulseOsc(V, c2)
        fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
        ring = [c2.image(F, V) for _ in range(R)]
        sp = zang.Span(0, F)
        step = lambda: [m.paint(sp, [o], [], False, m.Params(SR, zang.constant(fr), col), zero_first=True) for o in ring]
        step(); c2.sync()
        g = c2.capture(step)
        g.launch(); g.launch(); c2.sync()
        got = util.from_image(ring[-1])
        g.close(); c2.close()
    L = oracle.lib()
    ref = np.zeros((V, F), np.float32)
    for v in range(0, V, 16):                          # every 16th voice keeps the CPU side short
        st = oracle.PulseOsc(); L.zo_pulseosc_init(C.byref(st))
        for _ in range(K):
            ref[v] = 0
            L.zo_pulseosc_paint(C.byref(st), 0, F, oracle.fptr(ref[v]), SR, oracle.constant(freq[v]), float(color[v]))
    util.assert_bitexact(got[::16], ref[::16], "bench step after 12 buffers")


@pytest.mark.parametrize("kind", ["pulse", "trisaw", "envelope", "decimator", "curve", "portamento", "cycle", "pmosc", "sampler"])
def test_graph_with_odd_paint_count_and_eager_paints_between(ctx, kind):
    """The chunked oscillators -- and every module whose frame-range form writes its end state into the other half of a double
    buffer (1-6 words per voice) -- flip on the host at every such paint.  A graph holding an ODD number of paints, replayed
    back to back and mixed with eager paints, must still continue the state exactly (zh_graph_launch reconciles the
    buffers; ADVICE r1 osc.hip:356)."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V = 512
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
        sp = zang.Span(0, F)
        inp = c2.image(F, V); inp.uniform_(-1.0, 1.0)
        crv = torch.tensor([[0.0, 0.0], [1.0, 0.02], [0.3, 0.07], [0.8, 0.11], [0.0, 0.3]], dtype=torch.float32, device="cuda")
        pcm = torch.from_numpy(np.random.default_rng(2).integers(-20000, 20000, 5000, dtype=np.int16).view(np.uint8).copy()).cuda()
        rel = torch.full((V,), 0.3, dtype=torch.float32, device="cuda")
        make = {"pulse": lambda: mod.PulseOsc(V, c2), "trisaw": lambda: mod.TriSawOsc(V, c2), "envelope": lambda: mod.Envelope(V, c2),
                "decimator": lambda: mod.Decimator(V, c2), "curve": lambda: mod.Curve(V, c2), "portamento": lambda: mod.Portamento(V, c2),
                "cycle": lambda: mod.Cycle(V, c2), "pmosc": lambda: mod.PMOscInstrument(V, rel, c2), "sampler": lambda: mod.Sampler(V, c2)}[kind]
        me, mg = make(), make()

        def paint_one(m, o, n):
            """one paint (the same params at every position: a graph bakes them in and is replayed at several)"""
            if kind in ("pulse", "trisaw"):
                m.paint(sp, [o], [], False, m.Params(SR, zang.constant(fr), col), zero_first=True)
            elif kind == "envelope":
                m.paint(sp, [o], [], False, m.Params(SR, zang.PaintCurve.cubed(0.03), zang.PaintCurve.cubed(0.12), zang.PaintCurve.cubed(0.04), 0.6, True), zero_first=True)
            elif kind == "decimator":
                m.paint(sp, [o], [], False, m.Params(SR, inp, fr * 4.0), zero_first=True)
            elif kind == "curve":
                m.paint(sp, [o], [], False, m.Params(SR, m.smoothstep, crv), zero_first=True)
            elif kind == "portamento":
                m.paint(sp, [o], [], False, m.Params(SR, zang.PaintCurve.squared(0.2), fr, True, True), zero_first=True)
            elif kind == "cycle":
                m.paint(sp, [o], [], False, m.Params(SR, zang.constant(3.0)), zero_first=True)
            elif kind == "pmosc":
                m.paint(sp, [o], None, False, m.Params(SR, fr, True), zero_first=True)
            else:
                m.paint(sp, [o], [], False, m.Params(SR * 0.9, m.Sample(1, 44100, m.signed16_lsb, pcm), 0, True), zero_first=True)

        n_total = 1 + 3 + 3 + 1 + 3 + 2 + 3
        imgs_e = [c2.image(F, V) for _ in range(n_total)]
        imgs_g = [c2.image(F, V) for _ in range(n_total)]
        for n, o in enumerate(imgs_e):
            paint_one(me, o, n)
        # graph side: images are baked into the graph, so replays overwrite the same three; copy them out after each
        ring = [c2.image(F, V) for _ in range(3)]
        k = 0
        def eager(n):
            nonlocal k
            for _ in range(n):
                paint_one(mg, imgs_g[k], k); k += 1
        def replay(g):
            nonlocal k
            g.launch()
            for o in ring:
                imgs_g[k].copy_(o); k += 1
        eager(1)
        c2.sync()
        g = c2.capture(lambda: [paint_one(mg, o, 0) for o in ring])   # 3 paints: odd
        replay(g); replay(g); eager(1); replay(g); eager(2); replay(g)
        c2.sync()
        assert k == n_total
        for i, (a, b) in enumerate(zip(imgs_e, imgs_g)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{kind}: buffer {i} differs"
        se, sg = me.state(), mg.state()
        assert se.tobytes() == sg.tobytes()
        g.close(); c2.close()


def test_mixdown_scratch_cannot_grow_inside_capture(ctx):
    """ADVICE r1 basics.hip:248: growing the mixdown scratch inside a capture is refused (ZH_ERR_UNSUPPORTED), and a
    graph recorded with a smaller scratch stays valid after a later, larger eager mixdown grew it."""
    import torch
    import zang_amd
    from zang_amd import abi, zang
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        small = torch.rand((F, 4096), device="cuda"); big = torch.rand((F, 65536), device="cuda")
        mix_s = torch.zeros(F, device="cuda"); mix_b = torch.zeros(F, device="cuda"); mix_g = torch.zeros(F, device="cuda")
        sp = zang.Span(0, F)
        with pytest.raises(abi.ZangHipError):
            c2.capture(lambda: zang.mixdownVoices(sp, mix_s, small, zero_first=True, ctx=c2))     # nothing reserved yet
        zang.mixdownVoices(sp, mix_s, small, zero_first=True, ctx=c2)                              # reserves (eager)
        c2.sync()
        g = c2.capture(lambda: zang.mixdownVoices(sp, mix_g, small, zero_first=True, ctx=c2))
        zang.mixdownVoices(sp, mix_b, big, zero_first=True, ctx=c2)                                # grows: old block retired, not freed
        g.launch()
        c2.sync()
        assert torch.equal(mix_g.view(torch.int32), mix_s.view(torch.int32))
        ref = big.double().sum(1)
        assert torch.allclose(mix_b.double(), ref, rtol=0, atol=1e-2)
        g.close(); c2.close()


def test_sineosc_frame_ranges_in_graphs_and_eager(ctx, oracle):
    """SineOsc at a small voice count paints a span as many frame ranges at once (k_sineosc_ranges: every range replays the
    f32 phase additions of the frames before it) and double-buffers its phase like the chunked oscillators: an odd number
    of paints in a graph, replays mixed with eager paints, all four param paths -- equal to the oracle's sequential
    paints bit for bit, phase state included."""
    import ctypes as C
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang
    V = 320
    rng = np.random.default_rng(5)
    freq = rng.uniform(30.0, 3000.0, V).astype(np.float32)
    fbuf = rng.uniform(20.0, 2000.0, (V, F)).astype(np.float32)
    pbuf = rng.uniform(-1.0, 1.0, (V, F)).astype(np.float32)
    L = oracle.lib()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        for fb, pb in ((False, False), (True, False), (False, True), (True, True)):
            m = mod.SineOsc(V, c2)
            gf = zang.buffer(util.to_image(fbuf)) if fb else zang.constant(util.dev(freq))
            gp = zang.buffer(util.to_image(pbuf)) if pb else zang.constant(0.25)
            P = m.Params(SR, gf, gp)
            spans = [(0, F), (0, F), (0, F), (100, 900), (0, F), (0, F), (0, F), (900, 1024), (0, F), (0, F), (0, F)]
            ring = [c2.image(F, V) for _ in range(3)]
            got = []
            def eager(s, e):
                o = c2.image(F, V, fill=0.0)
                m.paint(zang.Span(s, e), [o], [], False, P)               # `+=` onto zeros
                got.append(o)
            eager(0, F)                                                    # allocate / warm
            c2.sync()
            g = c2.capture(lambda: [m.paint(zang.Span(0, F), [o], [], False, P, zero_first=True) for o in ring])   # 3 paints: odd
            def replay():
                g.launch()
                for o in ring:
                    got.append(o.clone())
            replay(); eager(100, 900); replay(); eager(900, 1024); replay()
            c2.sync()
            order = [(0, F)] + [(0, F)] * 3 + [(100, 900)] + [(0, F)] * 3 + [(900, 1024)] + [(0, F)] * 3
            ref_t = np.zeros(V, np.float32)
            for v in range(0, V, 7):                                       # every 7th voice keeps the CPU side short
                st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st))
                for k, (s, e) in enumerate(order):
                    buf = np.zeros(F, np.float32)
                    L.zo_sineosc_paint(C.byref(st), s, e, oracle.fptr(buf), SR,
                                       oracle.buffer(fbuf[v]) if fb else oracle.constant(freq[v]),
                                       oracle.buffer(pbuf[v]) if pb else oracle.constant(0.25))
                    g_col = got[k][:, v].cpu().numpy()
                    util.assert_bitexact(g_col[s:e], buf[s:e], f"sineosc ranges fb={fb} pb={pb} paint {k} voice {v}")
                ref_t[v] = st.t
            t = m.state()["t"].astype(np.float32)
            util.assert_bitexact(t[::7], ref_t[::7], "phase after the sequence")
            g.close(); m.close()
        c2.close()


def test_round2_forms_in_graphs_equal_eager(ctx):
    """The kernel forms added in round 2 that keep an end state aside or use module-owned scratch -- Decimator and the
    TriSawOsc control path as frame ranges (`next` buffer + commit kernel), pink noise (white image + tap pipeline),
    SimpleDelay as independent frames (store + advance kernels), PMOscInstrument ranges, Curve, Envelope, Portamento --
    captured once and replayed give the bits and the states of the same paints launched eagerly."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V = 512
    freq, color, u2, _ = workloads.voice_params(5, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        gf, gc = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
        inp = c2.image(F, V); inp.uniform_(-1.0, 1.0)
        fbuf = c2.image(F, V); fbuf.copy_(gf[None, :].expand(F, V))
        rel = torch.from_numpy((0.1 + 0.3 * u2).astype(np.float32)).cuda()
        crv = torch.tensor([[0.0, 0.0], [1.0, 0.005], [0.3, 0.012], [0.8, 0.02], [0.0, 0.05]], dtype=torch.float32, device="cuda")
        sp = zang.Span(0, F)

        def build():
            ms = dict(dec=mod.Decimator(V, c2), tri=mod.TriSawOsc(V, c2), pink=mod.Noise(V, c2, first_seed=5), dly=mod.SimpleDelay(V, 300, c2),
                      pm=mod.PMOscInstrument(V, rel, c2), crv=mod.Curve(V, c2), env=mod.Envelope(V, c2), por=mod.Portamento(V, c2))
            outs = {k: c2.image(F, V, fill=0.0) for k in ms}
            return ms, outs

        def step(ms, outs, k):
            ms["dec"].paint(sp, [outs["dec"]], [], False, ms["dec"].Params(SR, inp, 6000.0))
            ms["tri"].paint(sp, [outs["tri"]], [], False, ms["tri"].Params(SR, zang.buffer(fbuf), gc))
            ms["pink"].paint(sp, [outs["pink"]], [], False, ms["pink"].Params(ms["pink"].pink))
            ms["dly"].paint(sp, [outs["dly"]], [], False, ms["dly"].Params(inp))
            ms["pm"].paint(sp, [outs["pm"]], None, k == 0, ms["pm"].Params(SR, gf, k < 2))
            ms["crv"].paint(sp, [outs["crv"]], [], k == 0, ms["crv"].Params(SR, ms["crv"].smoothstep, crv))
            ms["env"].paint(sp, [outs["env"]], [], k == 0, ms["env"].Params(SR, zang.PaintCurve.cubed(0.01), zang.PaintCurve.cubed(0.03), zang.PaintCurve.cubed(0.02), 0.7, k < 2))
            ms["por"].paint(sp, [outs["por"]], [], k == 0, ms["por"].Params(SR, zang.PaintCurve.squared(0.03), gf, True, k > 0))

        me, oe = build(); mg, og = build()
        for ms, outs in ((me, oe), (mg, og)):                     # one eager pass first: module-owned scratch (the white image of
            for k in range(3):                                    # pink noise) is allocated outside a capture
                step(ms, outs, k)
        c2.sync()
        g = c2.capture(lambda: [step(mg, og, k) for k in range(3)])
        for _ in range(2):
            for k in range(3):
                step(me, oe, k)
        g.launch(); g.launch()
        c2.sync()
        for name in oe:
            assert torch.equal(oe[name].view(torch.int32), og[name].view(torch.int32)), name
            se, sg = me[name].state(), mg[name].state()
            if isinstance(se, tuple):
                assert all(np.array_equal(a, b) for a, b in zip(se, sg)), name
            else:
                assert se.tobytes() == sg.tobytes(), name
        g.close()
        c2.close()


def test_graph_of_in_place_paints_survives_flipping_paints_between_replays(ctx):
    """A graph that holds ONLY short-span paints of double-buffered modules (spans under 128 frames take the in-place
    one-walk form and do not flip the state buffers) bakes in the buffer the capture saw.  An eager 1024-frame paint
    in between takes the frame-range form and flips: the replay must first find the live state where it was baked
    (zh_flipper_used / zh_graph_launch).  Twin modules doing the same sequence eagerly are the reference."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V = 512
    freq, _, _, _ = workloads.voice_params(2, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        fr = torch.from_numpy(freq).cuda()
        on = torch.from_numpy((np.arange(V) % 3 != 0).astype(np.uint8)).cuda()
        off = torch.zeros(V, dtype=torch.uint8, device="cuda")
        short, full = zang.Span(0, 64), zang.Span(0, F)

        def make():
            return mod.SineOsc(V, c2), mod.Envelope(V, c2)

        def paint(ms, span, img, note_on, new):
            osc, env = ms
            osc.paint(span, [img], [], False, osc.Params(SR, zang.constant(fr), zang.constant(0.0)), zero_first=True)
            env.paint(span, [img], [], new, env.Params(SR, zang.PaintCurve.cubed(0.05), zang.PaintCurve.linear(0.2), zang.PaintCurve.squared(0.1), 0.6, note_on))

        me, mg = make(), make()
        img_e, img_g = [c2.image(F, V) for _ in range(2)], [c2.image(F, V) for _ in range(2)]
        paint(me, short, img_e[0], on, True); paint(mg, short, img_g[0], on, True)     # warm (and note on)
        c2.sync()
        g = c2.capture(lambda: paint(mg, short, img_g[0], on, False))                  # in-place forms only: no flips recorded
        for k in range(3):
            paint(me, short, img_e[0], on, False); g.launch()
            paint(me, full, img_e[1], on if k != 1 else off, False)                    # frame-range forms: flip the state buffers
            paint(mg, full, img_g[1], on if k != 1 else off, False)
            paint(me, short, img_e[0], on, False); g.launch()
            c2.sync()
            for a, b in zip(img_e, img_g):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), k
            for a, b in zip(me, mg):
                assert a.state().tobytes() == b.state().tobytes(), k
        g.close(); c2.close()


def test_graph_with_params_unchanged_paint_keeps_its_constants(ctx):
    """A recorded ZH_PAINT_PARAMS_UNCHANGED paint reads the module's constants table at every replay.  An eager paint with
    OTHER params between replays must not rewrite it (it used to: the replays then rendered with the new constants while every
    other param of the recorded call was the old one).  Twin module: the same sequence, all eager."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V = 1024
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
        fr2, col2 = (fr * 1.5).contiguous(), (1.0 - col).contiguous()
        sp = zang.Span(0, F)
        me, mg = mod.PulseOsc(V, c2), mod.PulseOsc(V, c2)
        a_e, a_g, b_e, b_g = (c2.image(F, V) for _ in range(4))
        P = lambda m, f, c: m.Params(SR, zang.constant(f), c)
        for m, img in ((me, a_e), (mg, a_g)):
            m.paint(sp, [img], [], False, P(m, fr, col), zero_first=True)              # unflagged: stores the constants
        c2.sync()
        g = c2.capture(lambda: mg.paint(sp, [a_g], [], False, P(mg, fr, col), zero_first=True, params_unchanged=True))
        for k in range(3):
            me.paint(sp, [a_e], [], False, P(me, fr, col), zero_first=True); g.launch()
            me.paint(sp, [b_e], [], False, P(me, fr2, col2), zero_first=True)          # other params, eager, unflagged
            mg.paint(sp, [b_g], [], False, P(mg, fr2, col2), zero_first=True)
            me.paint(sp, [a_e], [], False, P(me, fr, col), zero_first=True); g.launch()
            mg_flagged = c2.image(F, V); me_flagged = c2.image(F, V)
            me.paint(sp, [me_flagged], [], False, P(me, fr2, col2), zero_first=True, params_unchanged=(k > 0))
            mg.paint(sp, [mg_flagged], [], False, P(mg, fr2, col2), zero_first=True, params_unchanged=(k > 0))
            c2.sync()
            for x, y in ((a_e, a_g), (b_e, b_g), (me_flagged, mg_flagged)):
                assert torch.equal(x.view(torch.int32), y.view(torch.int32)), k
            assert me.state().tobytes() == mg.state().tobytes(), k
        g.close(); c2.close()


@pytest.mark.parametrize("kind", ["pulse", "trisaw"])
def test_coalescing_capture_equals_eager(ctx, kind):
    """ZH_CAPTURE_COALESCE: table-form oscillator paints are held back while recording and consecutive ones
    become one launch of several buffers (the zh_*_paint_batch launch: counters read once, written once, one flip) -- against
    the same calls made eagerly on a twin module.  The recorded
    sequence mixes: spans of different lengths (a new launch each), an image painted twice (ZERO_FIRST then `+=` by another
    module: must keep the recorded order), a paint WITHOUT the flag in the middle (ends the epoch: launched, published,
    recorded in order), another library call on the stream (zero: ends the epoch too), a run of equal spans (merged), a _batch
    call, voices with a bad frequency, and an odd number of flips.  Replayed three times with eager paints between replays;
    bits and carried state equal after each."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V, N = 2048, 9
    freq, color, u2, _ = workloads.voice_params(2, 0, V)
    freq[3] = -5.0; freq[700] = 9000.0                       # voices out of range: silent, counters not advanced
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
        fr2 = torch.from_numpy((freq * (1.0 + u2)).astype(np.float32)).cuda()
        Osc = mod.PulseOsc if kind == "pulse" else mod.TriSawOsc
        sets = []
        for _ in range(2):
            sets.append({"a": Osc(V, c2), "b": Osc(V, c2), "img": [c2.image(F, V, fill=0.25) for _ in range(N)]})
        spans = [zang.Span(0, F), zang.Span(0, 512), zang.Span(512, F), zang.Span(4, 1000), zang.Span(0, F), zang.Span(0, F), zang.Span(0, F), zang.Span(0, F), zang.Span(100, 104)]

        def seq(s, flagged):
            a, b, img = s["a"], s["b"], s["img"]
            Pa = a.Params(SR, zang.constant(fr), col); Pb = b.Params(SR, zang.constant(fr2), col)
            for i in range(N):
                a.paint(spans[i], [img[i]], [], False, Pa, zero_first=True, params_unchanged=flagged)
                if i in (0, 2):
                    b.paint(spans[i], [img[i]], [], False, Pb, zero_first=False, params_unchanged=flagged)     # += onto a's image
                if i == 3:
                    a.paint(zang.Span(0, 300), [img[0]], [], False, Pa, zero_first=False, params_unchanged=False)   # setup form: ordered
                if i == 4:
                    zang.zero(zang.Span(0, 64), img[1], c2)
            a.paint(zang.Span(0, F), [img[7]], [], False, Pa, zero_first=False, params_unchanged=flagged)           # img[7] again: not merged with its first paint
            a.paint_batch(zang.Span(0, F), [img[5], img[6]], Pa, zero_first=True, params_unchanged=flagged)

        e, g_ = sets
        seq(e, False); seq(g_, False)                    # unflagged first: the tables are stored
        c2.sync()
        graph = c2.capture(lambda: seq(g_, True), coalesce=True)
        nodes, held, launches = graph.info()
        assert held == N + 2 + 1 + 2 and launches < held and nodes < held + 6, (nodes, held, launches)
        for rep in range(3):
            seq(e, True); graph.launch()
            if rep == 1:                                  # eager paints between replays (flip the double buffers)
                for s in (e, g_):
                    s["a"].paint(zang.Span(0, 77), [s["img"][2]], [], False, s["a"].Params(SR, zang.constant(fr), col), zero_first=True)
            c2.sync()
            for q, (x, y) in enumerate(zip(e["img"], g_["img"])):
                assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (rep, q)
            for k in ("a", "b"):
                assert np.asarray(e[k].state()).tobytes() == np.asarray(g_[k].state()).tobytes(), (rep, k)
        graph.close(); c2.close()


def test_coalescing_capture_of_the_bench_step_matches_oracle(ctx, oracle):
    """bench.py's pulseosc graph as it is recorded now (20 zero+paint steps over distinct ring images, ZH_CAPTURE_COALESCE: two
    launches of 10 buffers, so that a replay ends on the counter buffer it began on): after three replays every image of the
    ring equals the oracle's buffer of that step, and the carried counters equal the oracle's -- on every 8th voice.  40 steps:
    32 + 8 buffers; 33 steps: 32 + 1; one step: an odd number of flips, reconciled by zh_graph_launch."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V = 4096
    freq, color, _, _ = workloads.voice_params(2, 0, V)
    L = oracle.lib()
    for K, want_launches in ((20, 2), (40, 2), (33, 2), (1, 1)):     # 10 + 10; 32 + 8; 32 + 1; one paint alone (odd: zh_graph_launch copies the counters)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            c2 = zang_amd.Context(0)
            m = mod.PulseOsc(V, c2)
            fr, col = torch.from_numpy(freq).cuda(), torch.from_numpy(color).cuda()
            ring = [c2.image(F, V) for _ in range(K)]
            sp = zang.Span(0, F)
            P = m.Params(SR, zang.constant(fr), col)
            m.paint(sp, [ring[0]], [], False, P, zero_first=True)                    # stores the constants
            c2.sync()
            g = c2.capture(lambda: [m.paint(sp, [o], [], False, P, zero_first=True, params_unchanged=True) for o in ring], coalesce=True)
            nodes, held, launches = g.info()
            assert (held, launches) == (K, want_launches) and nodes == want_launches, (nodes, held, launches)
            for _ in range(3):
                g.launch()
            c2.sync()
            got = [util.from_image(o)[::8] for o in ring]
            cnt = m.state()["cnt"][::8].copy()
            g.close(); c2.close()
        ref = np.zeros(F, np.float32)
        for q, v in enumerate(range(0, V, 8)):
            st = oracle.PulseOsc(); L.zo_pulseosc_init(C.byref(st))
            for step in range(1 + 3 * K):
                ref[:] = 0
                L.zo_pulseosc_paint(C.byref(st), 0, F, oracle.fptr(ref), SR, oracle.constant(freq[v]), float(color[v]))
                if step >= 1 + 2 * K:
                    util.assert_bitexact(got[step - 1 - 2 * K][q], ref, f"K={K} voice {v} step {step}")
            assert int(cnt[q]) == int(st.cnt), (K, v)

@pytest.mark.gpu
def test_coalescing_capture_refuses_a_mixdown_its_scratch_cannot_hold():
    """ADVICE r5 (composite.hip): in a ZH_CAPTURE_COALESCE capture zh_nice_paint_mix_stereo used to return ZH_OK as soon as it held
    the paint back, and the launch made later on its behalf dropped its error -- with no partial-sum scratch reserved (it cannot grow
    while a capture records) the recorded graph silently lacked the paint.  The paint call itself now answers, as in a capture
    without the flag; after an eager paint has sized the scratch the same capture records, and replays the eager result."""
    import torch
    import zang_amd
    from zang_amd import abi, modules as mod, zang, workloads
    V = 4096
    freq, color, u2, _ = workloads.voice_params(5, 1, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)                                         # a fresh context: no scratch yet
        gl = util.dev((0.25 + 0.5 * u2).astype(np.float32)); gr = util.dev((0.75 - 0.5 * u2).astype(np.float32))
        m = mod.NiceInstrument(V, util.dev(color), c2)
        sp = zang.Span(0, F)
        l = torch.zeros((2, F), device="cuda"); r = torch.zeros((2, F), device="cuda")
        P = m.Params(SR, util.dev(freq), True)
        two = lambda: [m.paint_mix_stereo(sp, l[k], r[k], gl, gr, k == 0, P, zero_first=True) for k in range(2)]
        for coalesce in (False, True):
            with pytest.raises(abi.ZangHipError):
                c2.capture(two, coalesce=coalesce)
        st0 = m.state()
        two(); c2.sync()                                                 # eager: reserves
        want_l, want_r = l.clone(), r.clone()
        m.set_state(st0)
        l.zero_(); r.zero_()
        g = c2.capture(two, coalesce=True)
        assert g.info()[1] == 2 and [k for k, _ in g.kernels()][0].startswith("k_nice_mix")
        g.launch(); c2.sync()
        assert torch.equal(l.view(torch.int32), want_l.view(torch.int32)) and torch.equal(r.view(torch.int32), want_r.view(torch.int32))
        g.close(); c2.close()



@pytest.mark.parametrize("V", [300, 4096])
def test_coalesced_nice_mixdowns_equal_separate_calls(V):
    """ZH_CAPTURE_COALESCE also holds back zh_nice_paint_mix_stereo: consecutive mixdowns of one instrument over one span with the
    same gains become launches of up to 8 buffers (the launch zh_nice_paint_mix_stereo_batch makes).  The replayed graph leaves
    bit for bit the mixes and the state of the same calls made one by one -- notes going on and off, a new note and a frequency
    change between buffers; a paint into a row the batch already holds, a change of gains and a different span each end the batch."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    freq, color, u2, _ = workloads.voice_params(5, 3, V)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = zang_amd.Context(0)
        gl = util.dev((0.25 + 0.5 * u2).astype(np.float32)); gr = util.dev((0.75 - 0.5 * u2).astype(np.float32))
        f1, f2 = util.dev(freq), util.dev((freq * np.float32(1.25)).astype(np.float32))
        on_mix = util.dev((np.arange(V) % 3 != 0).astype(np.uint8))
        ma, mb = mod.NiceInstrument(V, util.dev(color), c2), mod.NiceInstrument(V, util.dev(color), c2)
        script = [(True, True, f1), (True, False, f1), (on_mix, False, f1), (False, False, f2), (True, True, f2), (False, False, f2), (on_mix, on_mix, f1),
                  (True, False, f1), (True, True, f2), (on_mix, False, f2), (False, False, f2)]
        n = len(script)                                                  # 11 buffers: 8 + 3
        sp = zang.Span(0, F)
        la = torch.full((n + 3, F), 0.5, device="cuda"); ra = torch.full((n + 3, F), -0.25, device="cuda")
        lb, rb = la.clone(), ra.clone()
        P = [ma.Params(SR, f, on) for (on, _, f) in script]

        def seq(m, l, r):
            for k, (on, nic, f) in enumerate(script):
                m.paint_mix_stereo(sp, l[k], r[k], gl, gr, nic, P[k], zero_first=True)
            m.paint_mix_stereo(sp, l[n - 1], r[n - 1], gl, gr, False, P[0])                     # `+=` into a row of the open batch: a new batch
            m.paint_mix_stereo(sp, l[n], r[n], gr, gl, False, P[1], zero_first=True)            # other gains: a new batch
            m.paint_mix_stereo(zang.Span(100, 900), l[n + 1], r[n + 1], gr, gl, False, P[2])    # another span: a new batch
            m.paint_mix_stereo(zang.Span(100, 900), l[n + 2], r[n + 2], gr, gl, True, P[3])     # ... which this one joins

        seq(ma, la, ra)                                                  # eager, one by one (and sizes the scratch)
        seq(mb, lb, rb)
        c2.sync()
        sa, sb = ma.state().tobytes(), mb.state().tobytes()
        assert sa == sb
        g = c2.capture(lambda: seq(mb, lb, rb), coalesce=True)
        nodes, held, launches = g.info()
        assert held == n + 4 and launches == 5, (nodes, held, launches)   # 8 + 3, then 1, 1, 2
        for _ in range(2):
            seq(ma, la, ra)
            g.launch()
        c2.sync()
        assert float(la.abs().max()) > 0.5
        assert torch.equal(la.view(torch.int32), lb.view(torch.int32)) and torch.equal(ra.view(torch.int32), rb.view(torch.int32))
        assert ma.state().tobytes() == mb.state().tobytes() and ma.state().tobytes() != sa
        g.close(); c2.close()


def test_two_contexts_record_coalescing_captures_interleaved():
    """What is held back belongs to the context that records it: two contexts on one thread, their coalescing captures interleaved
    call by call (mixdowns on one, oscillator paints and mixdowns on the other), replay what the same calls made one by one leave."""
    import torch
    import zang_amd
    from zang_amd import modules as mod, zang, workloads
    V, K = 300, 5
    freq, color, u2, _ = workloads.voice_params(5, 3, V)
    sp = zang.Span(0, F)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cs, eager, rec, mixes = [], [], [], []
    for st in streams:
        with torch.cuda.stream(st):
            c = zang_amd.Context(0)
            cs.append(c)
            eager.append(mod.NiceInstrument(V, util.dev(color), c)); rec.append(mod.NiceInstrument(V, util.dev(color), c))
            mixes.append((torch.zeros((K, 2, F), device="cuda"), torch.zeros((K, 2, F), device="cuda")))
    gl = util.dev((0.25 + 0.5 * u2).astype(np.float32)); gr = util.dev((0.75 - 0.5 * u2).astype(np.float32)); gf = util.dev(freq)

    def call(i, m, k, which):
        with torch.cuda.stream(streams[i]):
            mx = mixes[i][which]
            m.paint_mix_stereo(sp, mx[k, 0], mx[k, 1], gl, gr, k == 0, m.Params(SR, gf, k < 3), zero_first=True)

    for i in (0, 1):                                                     # eager passes: the scratch is sized, both twins on the same state
        for k in range(K):
            call(i, eager[i], k, 0); call(i, rec[i], k, 1)
    for c in cs:
        c.sync()
    from zang_amd import abi
    for c in cs:
        abi.check(c.lib.zh_graph_begin_capture_flags(c.handle, abi.ZH_CAPTURE_COALESCE), "begin")
    for k in range(K):                                                   # interleaved: context 0, context 1, context 0, ...
        call(0, rec[0], k, 1); call(1, rec[1], k, 1)
    graphs = []
    for c in cs:
        g = C.c_void_p()
        abi.check(c.lib.zh_graph_end_capture(c.handle, C.byref(g)), "end")
        graphs.append(g)
    for i in (0, 1):
        for k in range(K):
            call(i, eager[i], k, 0)
        with torch.cuda.stream(streams[i]):
            abi.check(cs[i].lib.zh_graph_launch(cs[i].handle, graphs[i]), "launch")
    for c in cs:
        c.sync()
    for i in (0, 1):
        nodes, held, launches = C.c_uint32(), C.c_uint32(), C.c_uint32()
        abi.check(cs[i].lib.zh_graph_info(graphs[i], C.byref(nodes), C.byref(held), C.byref(launches)), "info")
        assert (held.value, launches.value) == (K, 1), (held.value, launches.value)
        assert float(mixes[i][0].abs().max()) > 0.5
        assert torch.equal(mixes[i][0].view(torch.int32), mixes[i][1].view(torch.int32)), i
        assert eager[i].state().tobytes() == rec[i].state().tobytes()
        cs[i].lib.zh_graph_destroy(graphs[i])
    for c in cs:
        c.close()


def test_graph_destroyed_after_its_context_is_harmless():
    """the documented order is graphs before their context, but a host written against the earlier rounds destroyed the
    context first -- zh_graph_destroy then dereferenced freed memory.  zh_destroy now makes its live graphs forget it: a late
    zh_graph_destroy frees the graph alone, and launching such a graph is refused (through the C ABI directly: the Python
    Context closes its children first)."""
    import torch
    from zang_amd import abi
    lib = abi.load()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        h = C.c_void_p()
        assert lib.zh_create(C.byref(h), 0) == 0
        assert lib.zh_set_stream(h, C.c_void_p(side.cuda_stream)) == 0
        img = torch.zeros((64, 256), dtype=torch.float32, device="cuda")
        buf = abi.Buf(img.data_ptr(), 256, 64, 256, 0)
        assert lib.zh_graph_begin_capture(h) == 0
        assert lib.zh_zero(h, 0, 64, buf) == 0
        g = C.c_void_p()
        assert lib.zh_graph_end_capture(h, C.byref(g)) == 0
        assert lib.zh_graph_launch(h, g) == 0
        assert lib.zh_sync(h) == 0
        h2 = C.c_void_p()
        assert lib.zh_create(C.byref(h2), 0) == 0
        assert lib.zh_graph_launch(h2, g) == abi.ZH_ERR_INVALID          # another context's graph
        assert lib.zh_destroy(h) == 0                                      # the context first ...
        assert lib.zh_graph_launch(h2, g) == abi.ZH_ERR_INVALID
        assert lib.zh_graph_destroy(g) == 0                                # ... then its graph: nothing of the context is touched
        assert lib.zh_destroy(h2) == 0
