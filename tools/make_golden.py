#!/usr/bin/env python3
"""Regenerate oracle-derived fixtures under tests/golden/ (run in the build container).
  config1_sine440_s16.bin : BASELINE config[0] -- 1 SineOsc voice, 440 Hz, phase 0, 48 kHz, one 1024-frame
                            buffer -> mixDown(s16, 1 channel, vol 0.25) -> 2048 payload bytes.
These are ORACLE outputs (the reference cannot be built here): they pin regressions and let the
GPU path be compared with a committed byte string, not the reference."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402


def config1_payload():
    L = po.lib()
    st = po.SineOsc(); L.zo_sineosc_init(C.byref(st))
    out = np.zeros(1024, np.float32)
    L.zo_sineosc_paint(C.byref(st), 0, 1024, po.fptr(out), 48000.0, po.constant(440.0), po.constant(0.0))
    dst = np.zeros(2048, np.uint8)
    L.zo_mixdown_s16lsb(dst.ctypes.data_as(C.POINTER(C.c_uint8)), po.fptr(out), 1024, 1, 0, 0.25)
    return dst.tobytes()


if __name__ == "__main__":
    open(os.path.join(ROOT, "tests", "golden", "config1_sine440_s16.bin"), "wb").write(config1_payload())
    print("wrote config1_sine440_s16.bin")
