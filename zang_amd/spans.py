"""Per-voice span tables: the device form of what Trigger yields for one buffer
(src/zang/trigger.zig:80-105; consumed like examples/example_song.zig:336-347)."""
import numpy as np
import torch

from . import abi


class SpanTable:
    """Build from per-voice lists of (start, end, freq, note_on, note_id_changed)."""

    def __init__(self, per_voice, device, max_spans=None):
        V = len(per_voice)
        K = max([len(s) for s in per_voice] + [1]) if max_spans is None else max_spans
        count = np.zeros(V, np.uint32)
        start = np.zeros((K, V), np.uint32); end = np.zeros((K, V), np.uint32)
        freq = np.zeros((K, V), np.float32)
        on = np.zeros((K, V), np.uint8); nic = np.zeros((K, V), np.uint8)
        for v, spans in enumerate(per_voice):
            assert len(spans) <= K
            count[v] = len(spans)
            for k, (s, e, f, o, c) in enumerate(spans):
                start[k, v], end[k, v], freq[k, v], on[k, v], nic[k, v] = s, e, f, 1 if o else 0, 1 if c else 0
        d = lambda a: torch.from_numpy(a).to(device)
        self.t = [d(count), d(start), d(end), d(freq), d(on), d(nic)]
        self.c = abi.SpanTable(K, 0, *[x.data_ptr() for x in self.t])
        self.max_spans = K
