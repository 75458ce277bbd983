#!/usr/bin/env python3
"""examples/write_wav.zig on the GPU: render a tracker-text song offline to a 16-bit mono WAV.
usage: write_wav.py song.txt out.wav [seconds]   (needs an MI355X; see zang_amd/song.py)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import zang_amd
    from zang_amd import song
    text = open(sys.argv[1]).read()
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 385.0     # NUM_SECONDS, write_wav.zig:7
    ctx = zang_amd.Context(0)
    r = song.SongRenderer(text, ctx)
    t0 = time.perf_counter()
    payload = r.render(seconds)
    dt = time.perf_counter() - t0
    with open(sys.argv[2], "wb") as f:
        f.write(song.wav_header(1, song.AUDIO_SAMPLE_RATE, 2, len(payload)))
        f.write(payload)
    nb = (int(seconds * song.AUDIO_SAMPLE_RATE) + 1023) // 1024
    print(f"rendered {seconds:.0f} s ({nb} buffers, {r.total_voices} sub-voices) in {dt:.2f} s = {seconds / dt:.1f}x real time")


if __name__ == "__main__":
    main()
