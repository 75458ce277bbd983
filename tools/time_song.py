#!/usr/bin/env python3
"""Config 4 wall-clock: GPU song render vs the oracle-driven render (1 thread), same song, same
number of buffers; also checks the two payloads are byte-identical.  usage: time_song.py song.txt seconds"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zang_amd
from zang_amd import song
from oracle import pyoracle as po
from tests.test_song import _oracle_song_render

text = open(sys.argv[1]).read(); seconds = float(sys.argv[2])
nbuf = int(seconds * 48000) // 1024
ctx = zang_amd.Context(0)
r = song.SongRenderer(text, ctx)
r.render_buffer(); ctx.sync()                        # warm up (first-launch costs)
r = song.SongRenderer(text, ctx)
t0 = time.perf_counter(); got1 = b"".join(r.render_buffer() for _ in range(nbuf)); t_gpu1 = time.perf_counter() - t0
r = song.SongRenderer(text, ctx)
t0 = time.perf_counter(); got = r.render(nbuf * 1024 / 48000.0); t_gpu = time.perf_counter() - t0
print(f"per-buffer launches: {t_gpu1:.2f} s; identical to batched: {got1 == got}")
t0 = time.perf_counter(); ref = _oracle_song_render(po, r.notes, song.EXAMPLE_SONG_INSTRUMENTS, nbuf); t_cpu = time.perf_counter() - t0
same = got == ref
print(f"{nbuf} buffers ({nbuf*1024/48000:.1f} s of audio, 17 sub-voices): GPU path {t_gpu:.2f} s ({nbuf*1024/48000/t_gpu:.1f}x real time), "
      f"oracle 1 thread {t_cpu:.2f} s ({nbuf*1024/48000/t_cpu:.1f}x real time); payload identical: {same}")
if not same:
    a = np.frombuffer(got, '<i2').astype(int); b = np.frombuffer(ref, '<i2').astype(int)
    print("differing samples:", (a != b).sum(), "max LSB", np.abs(a - b).max())
