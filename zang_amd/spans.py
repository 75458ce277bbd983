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
        self._upload(K, count, start, end, freq, on, nic, device)

    def _upload(self, K, count, start, end, freq, on, nic, device):
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.t = [d(count), d(start), d(end), d(freq), d(on), d(nic)]
        self.c = abi.SpanTable(K, 0, *[x.data_ptr() for x in self.t])
        self.max_spans = K

    @classmethod
    def from_arrays(cls, count, start, end, freq, note_on, nic, device):
        """count [V] u32; start/end [K][V] u32; freq [K][V] f32; note_on/nic [K][V] u8 -- the layout
        zh_poly_voice_schedule fills."""
        self = cls.__new__(cls)
        K = max(int(start.shape[0]), 1)
        if start.shape[0] == 0:
            V = len(count)
            start = np.zeros((1, V), np.uint32); end = np.zeros((1, V), np.uint32); freq = np.zeros((1, V), np.float32)
            note_on = np.zeros((1, V), np.uint8); nic = np.zeros((1, V), np.uint8)
        self._upload(K, count.astype(np.uint32), start.astype(np.uint32), end.astype(np.uint32), freq.astype(np.float32),
                     note_on.astype(np.uint8), nic.astype(np.uint8), device)
        return self
