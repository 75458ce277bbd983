"""CPU: the host scheduler (Trigger / ImpulseQueue / NoteTracker / PolyphonyDispatcher in
libzang_hip.so) against the reference's OWN unit tests, transcribed as data in
tests/golden/scheduler_tests.json -- the only reference-authored golden vectors for this path
(SURVEY.md 4, 8c).  Plus hand-derived cases for NoteTracker and ImpulseQueue."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from zang_amd import zang

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scheduler_tests.json")))
SPAN = zang.Span(0, 1024)


class MyNoteParams(C.Structure):           # notes_test.zig:6-8
    _fields_ = [("note_on", C.c_bool)]


class FreqNoteParams(C.Structure):         # examples/example_song.zig MyNoteParams {freq, note_on}
    _fields_ = [("freq", C.c_float), ("note_on", C.c_bool)]


@pytest.mark.parametrize("case", G["trigger"], ids=lambda c: c["name"])
def test_trigger_reference_cases(case):
    trig = zang.Trigger(C.c_float).init()
    for step in case["steps"]:
        iap = zang.ImpulsesAndParamses([zang.Impulse(*i) for i in step["impulses"]], step["params"])
        ctr = trig.counter(SPAN, iap)
        for (start, end, params, changed) in step["expected"]:
            r = trig.next(ctr)
            assert r is not None
            assert (r.span.start, r.span.end, r.params, r.note_id_changed) == (start, end, params, changed)
        assert trig.next(ctr) is None
        assert trig.next(ctr) is None


@pytest.mark.parametrize("case", G["polyphony_dispatcher"], ids=lambda c: c["name"])
def test_polyphony_dispatcher_reference_cases(case):
    N = zang.Notes(MyNoteParams)
    pd = N.PolyphonyDispatcher(case["polyphony"]).init()
    iap = zang.ImpulsesAndParamses([zang.Impulse(*i) for i in case["impulses"]], [MyNoteParams(x) for x in case["note_on"]])
    result = pd.dispatch(iap)
    assert [[int(i.note_id) for i in r.impulses] for r in result] == case["expected_note_ids"]
    assert [len(r) for r in result] == [len(x) for x in case["expected_note_ids"]]


def test_trigger_reset_and_gap_then_note():
    trig = zang.Trigger(C.c_float).init()
    iap = zang.ImpulsesAndParamses([zang.Impulse(10, 7, 1)], [1.0])
    c = trig.counter(SPAN, iap)
    r = trig.next(c)
    assert (r.span.start, r.span.end, r.note_id_changed) == (10, 1024, True)
    trig.reset()                            # trigger.zig:62-64: forgets the note
    c = trig.counter(SPAN, zang.ImpulsesAndParamses([], []))
    assert trig.next(c) is None


def test_impulse_queue_order_capacity_and_event_ids():
    N = zang.Notes(C.c_float)
    q = N.ImpulseQueue.init()
    q.push(5, 1, 10.0)
    q.push(3, 2, 20.0)                      # out of order: dropped (notes.zig:112-118)
    q.push(5, 3, 30.0)
    got = q.consume()
    assert [(i.frame, i.note_id, i.event_id) for i in got.impulses] == [(5, 1, 1), (5, 3, 2)]
    assert got.paramses == [10.0, 30.0]
    assert len(q.consume()) == 0            # consume empties the queue (:94)
    for k in range(40):
        q.push(k, k, float(k))
    got = q.consume()
    assert len(got) == 32                   # notes.zig:73,108-111
    assert got.impulses[0].event_id == 3 and got.impulses[31].event_id == 34


def test_note_tracker_frames_and_time_accumulation():
    """NoteTracker.consume (notes.zig:162-206) re-derived with numpy float32."""
    N = zang.Notes(FreqNoteParams)
    f32 = np.float32
    sr = f32(48000.0)
    times = [0.0, 0.005, 0.0213, 0.02134, 0.05, 0.0999, 0.1, 0.25]
    song = [N.SongEvent(FreqNoteParams(100.0 + k, k % 2 == 0), t, k + 1) for k, t in enumerate(times)]
    nt = N.NoteTracker.init(song)
    t = f32(0.0); nxt = 0
    for buf in range(14):
        span = zang.Span(0, 1024) if buf % 3 else zang.Span(100, 900)
        got = nt.consume(float(sr), span)
        out_len = span.end - span.start
        buf_time = f32(out_len) / sr
        end_t = t + buf_time
        exp = []
        while nxt < len(times) and f32(times[nxt]) < end_t:
            f = (f32(times[nxt]) - t) / buf_time
            rel = min(int(f * f32(out_len)), out_len - 1)
            nxt += 1
            exp.append((span.start + rel, nxt, nxt))
        t = end_t
        assert [(i.frame, i.note_id, i.event_id) for i in got.impulses] == exp
        assert [p.freq for p in got.paramses] == [100.0 + (e[1] - 1) for e in exp]
    assert nxt == len(times)
    nt.reset()
    assert len(nt.consume(float(sr), SPAN)) == 3      # 0.0, 0.005, 0.0213 < 1024/48000


def test_song_to_spans_pipeline():
    """NoteTracker -> PolyphonyDispatcher -> Trigger, the Voice.paint loop of
    examples/example_song.zig:326-349, producing per-sub-voice (span, params, note_id_changed)."""
    N = zang.Notes(FreqNoteParams)
    song = [N.SongEvent(FreqNoteParams(440.0, True), 0.0, 1), N.SongEvent(FreqNoteParams(550.0, True), 0.004, 2),
            N.SongEvent(FreqNoteParams(440.0, False), 0.010, 1), N.SongEvent(FreqNoteParams(660.0, True), 0.012, 3)]
    nt = N.NoteTracker.init(song)
    pd = N.PolyphonyDispatcher(2).init()
    trigs = [zang.Trigger(FreqNoteParams).init() for _ in range(2)]
    per_voice = pd.dispatch(nt.consume(48000.0, SPAN))
    spans = []
    for v in range(2):
        c = trigs[v].counter(SPAN, per_voice[v])
        while True:
            r = trigs[v].next(c)
            if r is None:
                break
            spans.append((v, r.span.start, r.span.end, r.params.freq, bool(r.params.note_on), r.note_id_changed))
    assert spans == [(0, 0, 480, 440.0, True, True), (0, 480, 576, 440.0, False, False), (0, 576, 1024, 660.0, True, True),
                     (1, 192, 1024, 550.0, True, True)]
