#!/usr/bin/env python3
"""CPU experiment for the tolerant time-parallel Filter: what does chunking the 2x-oversampled SVF
(Filter.zig:130-146) cost in accuracy?  numpy float32, many voices at once.
  reference   the sequential f32 recurrence over the whole span
  chunked     per chunk of L frames: zero-state end state (f32), start states by s_j = A^L s_{j-1} + e_{j-1} in f64, then the SAME
              f32 recurrence replayed from that start state
Reports, per parameter set: max |err| / peak(|ref|) per voice (worst voice), and the fraction of samples inside the per-sample
test metric |err| <= 1e-5 * max(|ref|, 1e-3) (tests/util.py assert_close)."""
import sys
import numpy as np

f32 = np.float32
D = f32(3.814697265625e-6)


def step(l, b, x, c, r):
    i = x + D
    l = l + (c * b - D)
    b = b + c * (i - b * r - l)
    l = l + c * b
    h = i - b * r - l
    b = b + c * h
    return l, b, h


def run(l, b, x, c, r, mul):
    out = np.empty_like(x)
    for k in range(x.shape[1]):
        l, b, h = step(l, b, x[:, k], c, r)
        out[:, k] = l * mul[0] + b * mul[1] + h * mul[2]
    return out, l, b


def hom64(c, r):
    """A (2x2, f64) of the homogeneous step: columns = images of (1,0), (0,1)."""
    c = c.astype(np.float64); r = r.astype(np.float64)
    def hs(l, b):
        l = l + c * b
        b = b + c * (-b * r - l)
        l = l + c * b
        h = -b * r - l
        b = b + c * h
        return l, b
    a00, a10 = hs(np.ones_like(c), np.zeros_like(c))
    a01, a11 = hs(np.zeros_like(c), np.ones_like(c))
    return np.array([[a00, a01], [a10, a11]])      # [2][2][V]


def matpow(A, n):
    R = np.array([[np.ones_like(A[0, 0]), np.zeros_like(A[0, 0])], [np.zeros_like(A[0, 0]), np.ones_like(A[0, 0])]])
    P = A.copy()
    while n:
        if n & 1:
            R = np.einsum("ijv,jkv->ikv", P, R)
        P = np.einsum("ijv,jkv->ikv", P, P)
        n >>= 1
    return R


def chunked(l0, b0, x, c, r, mul, L):
    V, N = x.shape
    C = N // L
    M = matpow(hom64(c, r), L)
    z = np.zeros(V, f32)
    ends = []
    for j in range(C):
        _, el, eb = run(z.copy(), z.copy(), x[:, j * L:(j + 1) * L], c, r, mul)
        ends.append((el, eb))
    out = np.empty_like(x)
    sl, sb = l0.astype(np.float64), b0.astype(np.float64)
    for j in range(C):
        o, _, _ = run(sl.astype(f32), sb.astype(f32), x[:, j * L:(j + 1) * L], c, r, mul)
        out[:, j * L:(j + 1) * L] = o
        el, eb = ends[j]
        sl, sb = M[0, 0] * sl + M[0, 1] * sb + el, M[1, 0] * sl + M[1, 1] * sb + eb
    return out, sl.astype(f32), sb.astype(f32)


def cutoff_from_frequency(f, sr):
    v = 2.0 * (1.0 - np.cos(np.pi * f / sr))
    return np.sqrt(np.clip(v, 0.0, 1.0)).astype(f32)


def report(name, ref, got):
    with np.errstate(invalid="ignore", over="ignore"):
        fin = np.isfinite(ref) & np.isfinite(got)
        both_bad = ~np.isfinite(ref) & ~np.isfinite(got)
        err = np.where(fin, np.abs(got.astype(np.float64) - ref.astype(np.float64)), 0.0)
        peak = np.max(np.where(np.isfinite(ref), np.abs(ref), 0), axis=1)
        rel_peak = (err.max(axis=1) / np.maximum(peak, 1e-30))
        tol = 1e-5 * np.maximum(np.abs(np.where(fin, ref, 0)), 1e-3)
        inside = (err <= tol) | both_bad
        mismatch_kind = (~fin & ~both_bad).sum()
    print(f"{name:42s} worst err/peak {rel_peak.max():.2e}  median voice {np.median(rel_peak):.2e}  inside per-sample metric {inside.mean() * 100:6.2f} %  "
          f"finite/non-finite disagreements {mismatch_kind}  peak {peak.max():.3g}")


def main():
    V, N = 512, 1024
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (V, N)).astype(f32)
    z = np.zeros(V, f32)
    lp, bp, hp, ap = (f32(1), f32(0), f32(0)), (f32(0), f32(1), f32(0)), (f32(0), f32(0), f32(1)), (f32(1), f32(1), f32(1))
    cases = [
        ("config 3: 200-8000 Hz, res 0-0.9, lowpass", cutoff_from_frequency(200 + 7800 * rng.random(V), 48000.0), (f32(1) - (0.9 * rng.random(V)).astype(f32)), lp),
        ("same, highpass", cutoff_from_frequency(200 + 7800 * rng.random(V), 48000.0), (f32(1) - (0.9 * rng.random(V)).astype(f32)), hp),
        ("res 0.9 everywhere, bandpass", cutoff_from_frequency(200 + 7800 * rng.random(V), 48000.0), np.full(V, f32(1) - f32(0.9)), bp),
        ("res 1.0 (no damping), low cutoffs", cutoff_from_frequency(50 + 500 * rng.random(V), 48000.0), np.zeros(V, f32), lp),
        ("cutoff 1.0, res 0-1 (unstable region)", np.ones(V, f32), (f32(1) - rng.random(V).astype(f32)), ap),
        ("cutoff 0.0", np.zeros(V, f32), np.full(V, f32(0.5)), lp),
        ("tiny cutoff 1e-4", np.full(V, f32(1e-4)), np.full(V, f32(0.3)), lp),
    ]
    for L in (32, 64, 128):
        print(f"--- chunk length {L}")
        for name, c, r, mul in cases:
            ref, rl, rb = run(z.copy(), z.copy(), x, c, r, mul)
            got, gl, gb = chunked(z.copy(), z.copy(), x, c, r, mul, L)
            report(name, ref, got)
    print("--- huge and tiny inputs, chunk 64")
    for scale in (1e-30, 1e-10, 1e10, 1e30):
        c = cutoff_from_frequency(200 + 7800 * rng.random(V), 48000.0); r = (f32(1) - (0.9 * rng.random(V)).astype(f32))
        xs = (x * f32(scale)).astype(f32)
        ref, _, _ = run(z.copy(), z.copy(), xs, c, r, lp)
        got, _, _ = chunked(z.copy(), z.copy(), xs, c, r, lp, 64)
        report(f"input scale {scale:g}", ref, got)


if __name__ == "__main__":
    with np.errstate(over="ignore", invalid="ignore"):
        main()
