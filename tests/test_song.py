"""Config 1 (plumbing) and config 4 (song pipeline): parser/event compiler on CPU, and the GPU
render (span tables -> fused voices -> sequential mix -> mixDown s16) vs an oracle-driven render
that makes the reference's per-sub-voice, per-sub-span paint calls in its accumulation order."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_SONG = "/root/reference/examples/example_song.txt"
SR = 48000.0
F = 1024


def _small():
    return open(os.path.join(GOLD, "song_small.txt")).read()


# ------------------------------------------------------------------ CPU
def test_parser_tokens_and_errors():
    from zang_amd import song
    p = song.Parser("# c\nrate 2.5 |C-4 off|    A#3\nfoo_1", 4)
    assert p.parse_token() == ("word", "rate")
    assert p.parse_token() == ("number", np.float32(2.5))
    assert p.parse_token() == ("notes", [-9, "off", None, -11])       # C-4 = 4*12-57+0 = -9, A#3 = 3*12-57+10 = -11
    assert p.parse_token() == ("word", "foo_1") and p.parse_token() is None and p.line_index == 2
    with pytest.raises(song.SongSyntaxError):
        song.Parser("|C-4 D-4 E-4", 2).parse_token()                    # too many columns (songparse1.zig:106-108)
    with pytest.raises(song.SongSyntaxError):
        song.Parser("|C-4x", 2).parse_token()                           # junk after a cell (:130-136)


def test_compile_small_song_event_tables():
    """Hand-checked against example_song.zig:129-262: rows before `start` are dropped but their columns'
    last notes are remembered (:152 TODO), a new note in a column first emits the old note's off,
    a row's events are sorted by note id (offs before ons), t advances 0.15/(rate*tempo) in f32."""
    from zang_amd import song
    notes = song.compile_song(_small())
    assert [len(n) for n in notes] == [11, 63, 12]
    ev = lambda e: (round(float(e.t), 4), e.note_id, e.semis, e.note_on)
    assert [ev(e) for e in notes[0][:4]] == [(0.0, 1, -33, False), (0.0, 5, -31, True), (0.3, 14, -36, True), (0.45, 5, -31, False)]
    assert [ev(e) for e in notes[1][:6]] == [(0.0, 2, -9, False), (0.0, 3, -5, False), (0.0, 4, -2, False),
                                             (0.0, 6, -7, True), (0.0, 7, -4, True), (0.0, 8, 0, True)]
    f32 = np.float32
    t = f32(0)
    for step in (f32(0.15) / (f32(2) * f32(1)),) * 2 + (f32(0.15) / (f32(2) * f32(0.5)),) * 2:
        t = t + step
    assert np.float32(notes[0][3].t) == t                                 # pedal `off` row
    for inst in notes:
        assert all(float(a.t) <= float(b.t) for a, b in zip(inst, inst[1:]))
    ons = {e.note_id for inst in notes for e in inst if e.note_on}
    offs = {e.note_id for inst in notes for e in inst if not e.note_on}
    assert offs - ons == {1, 2, 3, 4}                                     # the pre-`start` notes


@pytest.mark.skipif(not os.path.exists(REF_SONG), reason="reference song only exists in the build container")
def test_reference_song_aggregates():
    """In-container only: the parser on the real examples/example_song.txt against aggregate answers
    captured once.  The song itself never travels."""
    from zang_amd import song
    notes = song.compile_song(open(REF_SONG).read())
    assert [len(n) for n in notes] == [658, 6630, 280]
    assert sum(1 for n in notes for e in n if e.note_on) == 3784
    assert max(float(e.t) for n in notes for e in n) == 383.5047607421875
    h = hashlib.sha256()
    for k, n in enumerate(notes):
        for e in n:
            h.update(("%d %.9g %d %d %d\n" % (k, float(e.t), e.note_id, e.semis, int(e.note_on))).encode())
    assert h.hexdigest() == "1deaf3c20f10b096735d89af881e9f8ff798a00f881a6b97cd28b3897a0b4f8a"


def test_wav_header():
    from zang_amd import song
    h = song.wav_header(1, 48000, 2, 2048)
    assert len(h) == 44 and h[:4] == b"RIFF" and h[8:16] == b"WAVEfmt " and h[36:40] == b"data"
    assert int.from_bytes(h[4:8], "little") == 36 + 2048 and int.from_bytes(h[40:44], "little") == 2048
    assert int.from_bytes(h[24:28], "little") == 48000 and int.from_bytes(h[28:32], "little") == 96000
    assert int.from_bytes(h[22:24], "little") == 1 and int.from_bytes(h[34:36], "little") == 16


def _config1_oracle(oracle):
    L = oracle.lib()
    st = oracle.SineOsc(); L.zo_sineosc_init(C.byref(st))
    out = np.zeros(F, np.float32)
    L.zo_sineosc_paint(C.byref(st), 0, F, oracle.fptr(out), SR, oracle.constant(440.0), oracle.constant(0.0))
    dst = np.zeros(2 * F, np.uint8)
    L.zo_mixdown_s16lsb(dst.ctypes.data_as(C.POINTER(C.c_uint8)), oracle.fptr(out), F, 1, 0, 0.25)
    return dst.tobytes()


def test_config1_payload_matches_golden(oracle):
    """BASELINE config[0]: 1 SineOsc voice, 48 kHz, 1024-frame buffer -> s16 payload (CPU plumbing)."""
    got = _config1_oracle(oracle)
    assert got == open(os.path.join(GOLD, "config1_sine440_s16.bin"), "rb").read()
    s = np.frombuffer(got, "<i2")
    assert s[0] == 0 and abs(int(s.max()) - 8191) <= 1 and abs(int(s.min()) + 8191) <= 1   # 0.25 * 32767 peak


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_config1_on_gpu_matches_golden(ctx):
    import torch
    from zang_amd import modules as mod, zang
    m = mod.SineOsc(1, ctx)
    out = ctx.image(F, 1)
    m.paint(zang.Span(0, F), [out], [], False, m.Params(SR, zang.constant(440.0), zang.constant(0.0)), zero_first=True)
    mix = out[:, 0].contiguous()
    pcm = torch.zeros(2 * F, dtype=torch.uint8, device="cuda")
    zang.mixDown(pcm, mix, zang.AudioFormat.signed16_lsb, 1, 0, 0.25, ctx=ctx)
    ctx.sync()
    assert bytes(pcm.cpu().numpy()) == open(os.path.join(GOLD, "config1_sine440_s16.bin"), "rb").read()


def _oracle_song_render(oracle, notes, instruments, nbuf, last_frames=F):
    """The reference's MainModule.paint + write_wav loop with the oracle's module paints; the last of the nbuf
    buffers may be shorter (write_wav.zig:58-59)."""
    from zang_amd import song, zang
    L = oracle.lib()
    sched = song.SongScheduler(notes, instruments)
    mods = []
    for inst in instruments:
        subs = []
        for _ in range(inst.polyphony):
            if inst.kind == "pmosc":
                m = oracle.PMOscInstrument(); L.zo_pmosc_init(C.byref(m), inst.init_arg)
            else:
                m = oracle.NiceInstrument(); L.zo_nice_init(C.byref(m), inst.init_arg)
            subs.append(m)
        mods.append(subs)
    t0, t1, t2 = (np.zeros(F, np.float32) for _ in range(3))
    payload = []
    for b in range(nbuf):
        n = last_frames if b == nbuf - 1 else F
        out = np.zeros(F, np.float32)                                   # write_wav.zig:63-64
        tables = sched.buffer(zang.Span(0, n))
        for inst, subs, per_voice in zip(instruments, mods, tables):
            for m, spans in zip(subs, per_voice):
                for (s, e, f, on, nic) in spans:                         # example_song.zig:336-347
                    if inst.kind == "pmosc":
                        L.zo_pmosc_paint(C.byref(m), s, e, oracle.fptr(out), oracle.fptr(t0), oracle.fptr(t1), oracle.fptr(t2), int(nic), SR, f, int(on))
                    else:
                        L.zo_nice_paint(C.byref(m), s, e, oracle.fptr(out), oracle.fptr(t0), oracle.fptr(t1), int(nic), SR, f, int(on))
        dst = np.zeros(2 * n, np.uint8)
        L.zo_mixdown_s16lsb(dst.ctypes.data_as(C.POINTER(C.c_uint8)), oracle.fptr(out), n, 1, 0, 0.25)   # write_wav.zig:71-78
        payload.append(dst.tobytes())
    return b"".join(payload)


@pytest.mark.gpu
def test_song_render_matches_oracle_render(ctx, oracle):
    from zang_amd import song
    r = song.SongRenderer(_small(), ctx)
    nbuf = 80                                       # 1.7 s: the whole song plus release tails
    got = b"".join(r.render_buffer() for _ in range(nbuf))
    ref = _oracle_song_render(oracle, r.notes, song.EXAMPLE_SONG_INSTRUMENTS, nbuf)
    a = np.frombuffer(got, "<i2").astype(np.int32); b = np.frombuffer(ref, "<i2").astype(np.int32)
    assert np.abs(a).max() > 1000                   # it actually makes sound
    assert np.array_equal(a, b), f"{(a != b).sum()} of {a.size} s16 samples differ (max {np.abs(a - b).max()} LSB)"


@pytest.mark.gpu
def test_song_batched_render_equals_per_buffer(ctx, oracle):
    """Many buffers per launch (SongRenderer.render_batch) must not change a bit."""
    from zang_amd import song
    nbuf = 45
    ref = _oracle_song_render(oracle, song.resolve_frequencies(song.compile_song(_small()), ctx), song.EXAMPLE_SONG_INSTRUMENTS, nbuf)
    r = song.SongRenderer(_small(), ctx)
    got = r.render(nbuf * F / SR, batch=7)          # 6 batches of 7 + one of 3
    assert got == ref


def test_native_batch_scheduler_equals_per_buffer_scheduling():
    """zh_poly_voice (Voice(T)'s scheduling in C++, many buffers per call) == the reference's calls made one
    by one (NoteTracker.consume -> PolyphonyDispatcher.dispatch -> Trigger.next), sub-spans shifted per buffer."""
    from zang_amd import song, zang
    notes = song.compile_song(_small())
    for k, inst in enumerate(notes):
        for e in inst:
            e.freq = float(np.float32(100.0 + 7.5 * e.semis + k))
    counts = [1024] * 9 + [500, 1024, 3]
    ref = song.SongScheduler(notes)
    per_inst = [[[] for _ in range(i.polyphony)] for i in song.EXAMPLE_SONG_INSTRUMENTS]
    base = 0
    for n in counts:
        for k, per_voice in enumerate(ref.buffer(zang.Span(0, n))):
            for v, spans in enumerate(per_voice):
                per_inst[k][v].extend((s + base, e + base, np.float32(f), on, nic) for (s, e, f, on, nic) in spans)
        base += n
    nat = song.NativeSongScheduler(notes).batch(counts)
    total = 0
    for k, (count, start, end, freq, on, nic) in enumerate(nat):
        for v in range(len(count)):
            got = [(int(start[j, v]), int(end[j, v]), np.float32(freq[j, v]), bool(on[j, v]), bool(nic[j, v])) for j in range(int(count[v]))]
            assert got == per_inst[k][v], (k, v)
            total += len(got)
    assert total > 40
    # state carries across batch() calls exactly like across buffer() calls
    a = song.NativeSongScheduler(notes)
    first, second = a.batch(counts[:5]), a.batch(counts[5:])
    for k in range(len(nat)):
        for v in range(len(nat[k][0])):
            n1, n2 = int(first[k][0][v]), int(second[k][0][v])
            assert n1 + n2 == int(nat[k][0][v])
            off = sum(counts[:5])
            assert [int(x) + off for x in second[k][1][:n2, v]] == [int(x) for x in nat[k][1][n1:n1 + n2, v]]
