"""Synthetic per-voice inputs for the BASELINE.json configs.

Parameters come from a SplitMix64 stream seeded with ASCII "zang" || config id, in voice
order, so that every rank / test / the CPU baseline derive identical values for a voice
from its GLOBAL index alone.
"""
import numpy as np

MASK = (1 << 64) - 1
SAMPLE_RATE = 48000.0
FRAMES = 1024


def splitmix64_stream(seed, n, skip=0):
    """n uniform doubles in [0,1) from SplitMix64(seed), skipping the first `skip` draws."""
    idx = np.arange(skip + 1, skip + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def config_seed(config_id):
    return 0x7A616E6700000000 | config_id


def voice_params(config_id, first_voice, n_voices):
    """freq (Hz, log-uniform 55*2^(6.7u) clipped to [20, 6000] = sr/8, PulseOsc.zig:82),
    color in [0.1, 0.9], two more uniforms u2,u3 -- float32 arrays for voices
    [first_voice, first_voice + n_voices)."""
    u = splitmix64_stream(config_seed(config_id), 4 * n_voices, skip=4 * first_voice).reshape(n_voices, 4)
    freq = np.clip(55.0 * np.exp2(6.7 * u[:, 0]), 20.0, 6000.0).astype(np.float32)
    color = (0.1 + 0.8 * u[:, 1]).astype(np.float32)
    return freq, color, u[:, 2].astype(np.float32), u[:, 3].astype(np.float32)
