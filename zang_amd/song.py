"""The caller side of BASELINE config 4: zang's example_song pipeline on the GPU.

    tracker text --Parser (examples/common/songparse1.zig:3-198)-->
    SongEvents per instrument --doParse (examples/example_song.zig:129-262)-->
    per buffer: NoteTracker.consume -> PolyphonyDispatcher.dispatch -> Trigger.next
                (Voice.paint, example_song.zig:326-349; host C++ behind the C ABI)
    -> span tables -> zh_pmosc_paint_spans / zh_nice_paint_spans (one launch per instrument)
    -> sequential voice mix -> zang.mixDown s16 -> WAV bytes (examples/write_wav.zig:14-96)

The parser and event compiler are plain Python host code; all arithmetic that reaches a
sample is f32 (numpy.float32) in the reference's order.  Note frequencies use
std.math.pow(f32, 2.0, e): computed on the device by zh_pow (the product has no CPU math).
"""
import ctypes as C
import struct
from dataclasses import dataclass

import numpy as np

from . import abi
from . import zang
from .spans import SpanTable

f32 = np.float32
A4 = f32(440.0)                       # example_song.zig:19
NOTE_DURATION = f32(0.15)             # :20
AUDIO_SAMPLE_RATE = 48000             # :7
AUDIO_BUFFER_SIZE = 1024              # :8


class MyNoteParams(C.Structure):      # example_song.zig:23-26
    _fields_ = [("freq", C.c_float), ("note_on", C.c_bool)]


@dataclass
class Instrument:                     # Pedal / RegularOrgan / WeirdOrgan, example_song.zig:28-74
    kind: str                         # "pmosc" | "nice"
    init_arg: float                   # release_duration | color
    freq_mul: float                   # makeParams: src.freq * 0.5 for the pedal
    polyphony: int
    num_columns: int


EXAMPLE_SONG_INSTRUMENTS = [Instrument("pmosc", 0.4, 0.5, 3, 2), Instrument("nice", 0.25, 1.0, 10, 8),
                            Instrument("nice", 0.1, 1.0, 4, 2)]


class SongSyntaxError(Exception):
    pass


SEMITONES = {"C-": 0, "C#": 1, "D-": 2, "D#": 3, "E-": 4, "F-": 5, "F#": 6, "G-": 7, "G#": 8, "A-": 9, "A#": 10, "B-": 11}


class Parser:
    """songparse1.zig Parser(num_columns).  Tokens: ("word", str) | ("number", f32) | ("notes", list) where a
    note is None (idle), "off", or an int semitone offset from a4 (the frequency is resolved later, in bulk)."""

    def __init__(self, contents, num_columns):
        self.contents = contents
        self.n = num_columns
        self.index = 0
        self.line_index = 0

    def eat(self, prefix):                                     # :65-72
        if self.contents.startswith(prefix, self.index):
            self.index += len(prefix)
            return True
        return False

    def parse_note(self):                                      # :29-63
        c = self.contents
        if self.index + 3 > len(c):
            return None
        letter_mod, octave = c[self.index:self.index + 2], c[self.index + 2]
        if not ("0" <= octave <= "9"):
            return None
        if letter_mod not in SEMITONES:
            return None
        self.index += 3
        return (ord(octave) - ord("0")) * 12 - 57 + SEMITONES[letter_mod]

    def parse_token(self):                                     # :74-185
        c = self.contents
        while True:
            if self.eat(" "):
                pass
            elif self.eat("\n"):
                self.line_index += 1
            elif self.eat("#"):
                pos = c.find("\n", self.index)
                if pos >= 0:
                    self.line_index += 1
                    self.index = pos + 1
                else:
                    self.index = len(c)
            else:
                break
        if self.index >= len(c):
            return None
        ch = c[self.index]
        if ch == "|":
            self.index += 1
            notes = [None] * self.n
            col = 0
            while True:
                if col >= self.n:
                    raise SongSyntaxError(f"too many columns on line {self.line_index + 1}")
                semis = self.parse_note()
                if semis is not None:
                    notes[col] = semis
                elif self.eat("off"):
                    notes[col] = "off"
                elif self.eat("   "):
                    pass
                else:
                    break
                if self.index < len(c) and c[self.index] in " |":
                    self.index += 1
                else:
                    break
                col += 1
            if self.index < len(c):
                if c[self.index] == "\n":
                    self.line_index += 1
                    self.index += 1
                else:
                    raise SongSyntaxError(f"syntax error on line {self.line_index + 1}")
            return ("notes", notes)
        if ch.isascii() and (ch.isalpha() or ch == "_"):
            start = self.index
            self.index += 1
            while self.index < len(c) and c[self.index].isascii() and (c[self.index].isalnum() or c[self.index] == "_"):
                self.index += 1
            return ("word", c[start:self.index])
        if "0" <= ch <= "9":
            start = self.index
            dot = False
            self.index += 1
            while self.index < len(c):
                ch2 = c[self.index]
                if ch2 == ".":
                    if dot:
                        break
                    dot = True
                    self.index += 1
                elif "0" <= ch2 <= "9":
                    self.index += 1
                else:
                    break
            return ("number", f32(c[start:self.index]))
        raise SongSyntaxError(f"syntax error on line {self.line_index + 1}")

    def require_number(self):                                  # :191-196
        tok = self.parse_token()
        if tok is None or tok[0] != "number":
            raise SongSyntaxError("expected number")
        return tok[1]


@dataclass
class SongEvent:                       # Notes(MyNoteParams).SongEvent
    t: float                           # f32 seconds
    note_id: int
    semis: int                         # semitone offset from a4 (frequency = a4 * pow(2, semis/12))
    note_on: bool
    freq: float = 0.0


def compile_song(text, instruments=EXAMPLE_SONG_INSTRUMENTS):
    """doParse (example_song.zig:129-262): per-instrument chronological SongEvent lists."""
    columns = [i.num_columns for i in instruments]
    parser = Parser(text, sum(columns))
    col_instr = [k for k, n in enumerate(columns) for _ in range(n)]
    last = [None] * sum(columns)       # column_last_note
    notes = [[] for _ in instruments]
    next_id = 1
    t, rate, tempo = f32(0.0), f32(1.0), f32(1.0)
    while True:
        tok = parser.parse_token()
        if tok is None:
            break
        if tok == ("word", "start"):   # :146-152
            t = f32(0.0)
            notes = [[] for _ in instruments]
        elif tok == ("word", "rate"):
            rate = parser.require_number()
        elif tok == ("word", "tempo"):
            tempo = parser.require_number()
        elif tok[0] == "notes":
            old = [len(n) for n in notes]
            for col, note in enumerate(tok[1]):
                dst = notes[col_instr[col]]
                if note is None:
                    continue
                if note == "off":                              # :201-212
                    if last[col] is not None:
                        dst.append(SongEvent(t, last[col][1], last[col][0], False))
                        last[col] = None
                else:                                          # :176-200
                    if last[col] is not None:
                        dst.append(SongEvent(t, last[col][1], last[col][0], False))
                    dst.append(SongEvent(t, next_id, note, True))
                    last[col] = (note, next_id)
                    next_id += 1
            t = t + NOTE_DURATION / (rate * tempo)             # :216
            for k in range(len(instruments)):                  # :218-238 stable sort of this row's events by note id
                notes[k][old[k]:] = sorted(notes[k][old[k]:], key=lambda e: e.note_id)
        else:
            raise SongSyntaxError(f"bad token {tok!r}")
    return notes


def resolve_frequencies(notes, ctx):
    """freq = a4 * pow(f32, 2.0, f32(semis) / 12.0) (songparse1.zig:61-62), on the device via zh_pow."""
    import torch
    semis = sorted({e.semis for inst in notes for e in inst})
    if not semis:
        return notes
    exps = np.array([f32(s) / f32(12.0) for s in semis], np.float32)
    x = torch.full((len(semis),), 2.0, dtype=torch.float32, device=ctx.device)
    y = torch.from_numpy(exps).to(ctx.device)
    out = torch.empty_like(y)
    abi.check(ctx.lib.zh_pow(ctx.handle, len(semis), out.data_ptr(), x.data_ptr(), y.data_ptr()), "zh_pow")
    pw = out.cpu().numpy()
    table = {s: float(A4 * pw[i]) for i, s in enumerate(semis)}
    for inst in notes:
        for e in inst:
            e.freq = table[e.semis]
    return notes


class SongScheduler:
    """Voice(T) x 3 (example_song.zig:288-350): per buffer, per sub-voice sub-span lists."""

    def __init__(self, notes, instruments=EXAMPLE_SONG_INSTRUMENTS):
        N = zang.Notes(MyNoteParams)
        self.instruments = instruments
        self.trackers = [N.NoteTracker.init([N.SongEvent(MyNoteParams(e.freq, e.note_on), float(e.t), e.note_id) for e in inst])
                         for inst in notes]
        self.dispatchers = [N.PolyphonyDispatcher(i.polyphony).init() for i in instruments]
        self.triggers = [[zang.Trigger(MyNoteParams).init() for _ in range(i.polyphony)] for i in instruments]

    def buffer(self, span):
        """-> per instrument, per sub-voice: [(start, end, freq*freq_mul, note_on, note_id_changed), ...]"""
        out = []
        for k, inst in enumerate(self.instruments):
            iap = self.trackers[k].consume(float(AUDIO_SAMPLE_RATE), span)
            poly = self.dispatchers[k].dispatch(iap)
            per_voice = []
            for v in range(inst.polyphony):
                trig = self.triggers[k][v]
                ctr = trig.counter(span, poly[v])
                spans = []
                while True:
                    r = trig.next(ctr)
                    if r is None:
                        break
                    spans.append((r.span.start, r.span.end, float(f32(r.params.freq) * f32(inst.freq_mul)),
                                  bool(r.params.note_on), r.note_id_changed))
                per_voice.append(spans)
            out.append(per_voice)
        return out


class NativeSongScheduler:
    """The same scheduling through zh_poly_voice (Voice(T)'s scheduling half in C++, one call per instrument
    per batch of buffers) -- what SongRenderer uses; SongScheduler above makes the reference's calls one by
    one from Python and is what the tests read."""
    _dtype = np.dtype({"names": ["freq", "note_on"], "formats": ["<f4", "u1"],
                       "offsets": [MyNoteParams.freq.offset, MyNoteParams.note_on.offset], "itemsize": C.sizeof(MyNoteParams)})

    def __init__(self, notes, instruments=EXAMPLE_SONG_INSTRUMENTS):
        self.lib = abi.load()
        self.instruments = instruments
        self.handles = []
        for inst, evs in zip(instruments, notes):
            n = len(evs)
            params = (MyNoteParams * max(n, 1))(*[MyNoteParams(e.freq, e.note_on) for e in evs])
            t = np.array([e.t for e in evs], np.float32)
            ids = np.array([e.note_id for e in evs], np.uint64)
            h = C.c_void_p()
            abi.check(self.lib.zh_poly_voice_create(inst.polyphony, C.sizeof(MyNoteParams), MyNoteParams.note_on.offset, n,
                                                    C.cast(params, C.c_void_p), t.ctypes.data, ids.ctypes.data, C.byref(h)),
                      "zh_poly_voice_create")
            self.handles.append(h)

    def batch(self, frame_counts):
        """-> per instrument: (count [P], start [K][P], end, freq*freq_mul, note_on, note_id_changed), sub-spans of
        buffer b shifted by the frames before it."""
        frames = np.asarray(frame_counts, np.uint32)
        out = []
        for inst, h in zip(self.instruments, self.handles):
            P = inst.polyphony
            cap = 34 * len(frames) + 1                         # <= 32 impulses + carry-over per buffer per sub-voice
            count = np.zeros(P, np.uint32)
            start = np.zeros((cap, P), np.uint32); end = np.zeros((cap, P), np.uint32)
            params = np.zeros((cap, P), self._dtype); nic = np.zeros((cap, P), np.uint8)
            abi.check(self.lib.zh_poly_voice_schedule(h, float(AUDIO_SAMPLE_RATE), frames.ctypes.data, len(frames), cap, count.ctypes.data,
                                                      start.ctypes.data, end.ctypes.data, params.ctypes.data, nic.ctypes.data),
                      "zh_poly_voice_schedule")
            K = int(count.max()) if P else 0
            freq = params["freq"][:K] * np.float32(inst.freq_mul)          # makeParams (example_song.zig:35-39 etc.), f32
            out.append((count, start[:K], end[:K], freq, params["note_on"][:K], nic[:K]))
        return out

    def close(self):
        for h in self.handles:
            self.lib.zh_poly_voice_destroy(h)
        self.handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SongRenderer:
    """MainModule + write_wav's buffer loop on the GPU."""

    def __init__(self, text, ctx, instruments=EXAMPLE_SONG_INSTRUMENTS, vol=0.25):
        import torch
        from . import modules as mod
        self.ctx = ctx
        self.instruments = instruments
        self.notes = resolve_frequencies(compile_song(text, instruments), ctx)
        self.sched = SongScheduler(self.notes, instruments)          # per-buffer path (render_buffer)
        self.native = NativeSongScheduler(self.notes, instruments)   # batched path (render / render_batch)
        self.total_voices = sum(i.polyphony for i in instruments)
        F = AUDIO_BUFFER_SIZE
        self.image = ctx.image(F, self.total_voices)           # one column per sub-voice, in painting order
        self.mix = torch.zeros(F, dtype=torch.float32, device=ctx.device)
        self.pcm = torch.zeros(F * 2, dtype=torch.uint8, device=ctx.device)
        # The instruments are independent until the mix (example_song.zig:340-346 adds them in order
        # afterwards), and a handful of sub-voices is one wave walking frames serially: each instrument
        # paints on its own stream so the three serial walks overlap instead of queueing.
        from .runtime import Context
        self.main_stream = getattr(ctx, "_stream", None) or torch.cuda.current_stream(ctx.device)
        self.ictx = []
        self.mods, self.views = [], []
        col = 0
        for inst in instruments:
            ic = Context(ctx.device.index, adopt_torch_stream=False)
            ic.use_stream(torch.cuda.Stream(device=ctx.device))
            self.ictx.append(ic)
            if inst.kind == "pmosc":
                self.mods.append(mod.PMOscInstrument(inst.polyphony, inst.init_arg, ic))
            else:
                self.mods.append(mod.NiceInstrument(inst.polyphony, inst.init_arg, ic))
            self.views.append(self.image[:, col:col + inst.polyphony])
            col += inst.polyphony
        self.vol = vol

    def render_buffer(self, nframes=AUDIO_BUFFER_SIZE):
        """One write_wav iteration (write_wav.zig:58-93): returns nframes*2 bytes of s16 mono PCM."""
        span = zang.Span(0, nframes)
        tables = self.sched.buffer(span)
        live = []                                                  # tables stay allocated until the main stream has joined
        for m, ic, view, per_voice in zip(self.mods, self.ictx, self.views, tables):
            live.append(SpanTable(per_voice, self.ctx.device))     # uploaded on the main stream
            ic._stream.wait_stream(self.main_stream)
            m.paint_spans(span, [view], None, float(AUDIO_SAMPLE_RATE), live[-1], zero_first=True)
        for ic in self.ictx:
            self.main_stream.wait_stream(ic._stream)
        # outputs[0] was zeroed (write_wav.zig:63-64); sub-voices accumulate in painting order
        zang.mixdownVoices(span, self.mix, self.image, zero_first=True, sequential=True, ctx=self.ctx)
        zang.mixDown(self.pcm[:nframes * 2], self.mix[:nframes], zang.AudioFormat.signed16_lsb, 1, 0, self.vol, ctx=self.ctx)
        return bytes(self.pcm[:nframes * 2].cpu().numpy())

    def _prepare_batch(self, frame_counts):
        """Host part of a batch: schedule every buffer exactly as write_wav does (NoteTracker quantises note
        times per 1024-frame buffer), shifting buffer b's sub-spans by its start frame.  Trigger's carry-over
        already splits a note at every buffer boundary, so the per-call prologue/epilogue structure -- and
        the bits -- are unchanged, while the device walks len(frame_counts)*1024 frames per launch."""
        return int(sum(frame_counts)), self.native.batch(frame_counts)

    def _launch_batch(self, prepared):
        """Device part: upload the span tables, paint the instruments (each on its own stream), mix, convert."""
        import torch
        total, per_inst = prepared
        if getattr(self, "_batch_frames", 0) < total:
            self._bimage = self.ctx.image(total, self.total_voices)
            self._bmix = torch.zeros(total, dtype=torch.float32, device=self.ctx.device)
            self._bpcm = torch.zeros(total * 2, dtype=torch.uint8, device=self.ctx.device)
            self._batch_frames = total
        span = zang.Span(0, total)
        col = 0
        live = []
        for m, ic, inst, per_voice in zip(self.mods, self.ictx, self.instruments, per_inst):
            view = self._bimage[:, col:col + inst.polyphony]
            col += inst.polyphony
            live.append(SpanTable.from_arrays(*per_voice, self.ctx.device))
            ic._stream.wait_stream(self.main_stream)
            m.paint_spans(span, [view], None, float(AUDIO_SAMPLE_RATE), live[-1], zero_first=True)
        for ic in self.ictx:
            self.main_stream.wait_stream(ic._stream)
        zang.mixdownVoices(span, self._bmix, self._bimage, zero_first=True, sequential=True, ctx=self.ctx)
        zang.mixDown(self._bpcm[:total * 2], self._bmix[:total], zang.AudioFormat.signed16_lsb, 1, 0, self.vol, ctx=self.ctx)
        return total, live

    def _collect_batch(self, launched):
        total, _live = launched                                  # tables stay allocated until the copy below has synchronised
        return bytes(self._bpcm[:total * 2].cpu().numpy())

    def render_batch(self, frame_counts):
        """Several consecutive write_wav iterations in ONE set of launches."""
        return self._collect_batch(self._launch_batch(self._prepare_batch(frame_counts)))

    def render(self, seconds, batch=256):
        """write_wav's loop (write_wav.zig:58-93), `batch` buffers per launch; the host schedules batch k+1
        while the device renders batch k."""
        total = int(seconds * AUDIO_SAMPLE_RATE)
        counts, start = [], 0
        while start < total:                                   # write_wav.zig:58-59
            n = min(AUDIO_BUFFER_SIZE, total - start)
            counts.append(n)
            start += n
        groups = [counts[i:i + batch] for i in range(0, len(counts), batch)]
        out = []
        prepared = self._prepare_batch(groups[0]) if groups else None
        for g in range(len(groups)):
            launched = self._launch_batch(prepared)
            prepared = self._prepare_batch(groups[g + 1]) if g + 1 < len(groups) else None
            out.append(self._collect_batch(launched))
        return b"".join(out)


def wav_header(num_channels, sample_rate, bytes_per_sample, data_bytes):
    """Canonical 44-byte RIFF/WAVE PCM header (zig-wav is un-vendored: SURVEY.md 8c)."""
    byte_rate = sample_rate * num_channels * bytes_per_sample
    return (b"RIFF" + struct.pack("<I", 36 + data_bytes) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, num_channels, sample_rate, byte_rate, num_channels * bytes_per_sample,
                        bytes_per_sample * 8) + b"data" + struct.pack("<I", data_bytes))
