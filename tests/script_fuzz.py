"""Random zangscript modules for the parity fuzz of the generated kernels (tests/test_gpu_script_fuzz.py,
tools/fuzz_scripts.py): a seeded generator of script text over every builtin module the HIP backend supports, the
arithmetic operators and functions, `let` names, nested user modules, `cob` params handed down as they are, and `delay`
blocks -- and the driver that paints a generated module over random sub-spans with random per-voice note events, the
fused kernel on the GPU against oracle/zs_interp.py voice by voice, bit for bit.

The generator keeps values finite (no division by a possible zero, sqrt / pow of non-negative arguments only): a NaN's
payload is not part of the parity contract (x86 and gfx950 produce different default NaNs)."""
import numpy as np

F = 96
V = 70          # one full wave + a partial one
SR = 48000.0

CURVES = (".instantaneous", ".linear({d})", ".squared({d})", ".cubed({d})")
FILTER_TYPES = ("bypass", "low_pass", "band_pass", "high_pass", "notch", "all_pass")


class Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.noise = 0

    # ---- small helpers
    def pick(self, xs):
        return xs[int(self.rng.integers(len(xs)))]

    def chance(self, p):
        return self.rng.random() < p

    def lit(self, lo, hi, digits=4):
        x = round(float(self.rng.uniform(lo, hi)), digits)
        return repr(abs(x)) if x >= 0 else "(-%r)" % abs(x)

    def curve(self):
        # stage lengths of 5 .. 150 frames at 48 kHz: stage ends inside the 96-frame buffers, between chunks without one
        return self.pick(CURVES).format(d=round(float(self.rng.uniform(0.0001, 0.003)), 5))

    # ---- constants (literals and the `k` param)
    def const(self, ctx, lo=-2.0, hi=2.0):
        r = self.rng.random()
        if r < 0.6 or not ctx["consts"]:
            return self.lit(lo, hi)
        c = self.pick(ctx["consts"])
        if c == "pitch":
            return "(pitch * 0.001)"
        if r < 0.8:
            return c
        return "(%s * %s)" % (c, self.lit(0.1, 1.5))

    # ---- frequencies: the cob param as it is (the generated kernels follow it), scaled, a literal, or a computed buffer
    def freq(self, ctx, depth):
        r = self.rng.random()
        if r < 0.3 and "freq" in ctx["bufs"]:
            return "freq"
        if r < 0.5 and "freq" in ctx["bufs"]:
            return "(freq * %s)" % self.lit(0.25, 3.0)
        if r < 0.7:
            return self.lit(1.0, 4000.0, 2)
        if depth > 0:
            return "(%s + %s * %s)" % (self.lit(100.0, 900.0, 1), self.lit(5.0, 200.0, 1), self.buf(ctx, depth - 1))
        return self.lit(20.0, 2000.0, 2)

    def module_call(self, ctx, depth):
        names = ["SineOsc", "PulseOsc", "TriSawOsc", "Noise", "Envelope", "Gate", "Cycle"]
        if depth > 0:
            names += ["Filter", "Decimator", "Distortion", "SineOsc"]
        if ctx.get("portamento"):
            names.append("Portamento")
        names.append("Curve")
        names += ctx["helpers"]
        n = self.pick(names)
        if n == "SineOsc":
            ph = self.lit(-1.0, 1.0) if self.chance(0.6) or depth == 0 else "(%s * %s)" % (self.lit(0.05, 0.5), self.buf(ctx, depth - 1))
            return "SineOsc(freq=%s, phase=%s)" % (self.freq(ctx, depth), ph)
        if n in ("PulseOsc", "TriSawOsc"):
            f, c = self.freq(ctx, depth), self.lit(0.0, 1.0, 3)
            if n == "TriSawOsc" and float(c) < 0.15:                    # the sawtooth (brpt == 0: voices.hip.h trisaw_sample_saw) now and then
                c = "0"
            return "%s(freq=%s, color=%s)" % (n, f, c)
        if n == "Noise":
            self.noise += 1
            return "Noise(color=.%s)" % self.pick(("white", "pink"))
        if n == "Envelope":
            return "Envelope(attack=%s, decay=%s, release=%s, sustain_volume=%s, note_on)" % (
                self.curve(), self.curve(), self.curve(), self.pick(("1", self.lit(0.2, 0.9, 3))))
        if n == "Gate":
            return "Gate(note_on)"
        if n == "Cycle":
            return "Cycle(speed=%s)" % self.freq(ctx, depth)
        if n == "Filter":
            cut = self.lit(0.0, 1.0, 3) if self.chance(0.6) else "(%s + %s * %s)" % (self.lit(0.1, 0.5, 3), self.lit(0.05, 0.3, 3), self.buf(ctx, depth - 1))
            res = self.lit(0.0, 1.0, 3) if self.chance(0.7) else "(%s * %s)" % (self.lit(0.1, 0.9, 3), self.buf(ctx, depth - 1))
            return "Filter(input=%s, type=.%s, cutoff=%s, res=%s)" % (self.buf(ctx, depth - 1), self.pick(FILTER_TYPES), cut, res)
        if n == "Decimator":
            return "Decimator(input=%s, fake_sample_rate=%s)" % (self.buf(ctx, depth - 1), self.lit(1500.0, 60000.0, 1))
        if n == "Distortion":
            return "Distortion(input=%s, type=.%s, ingain=%s, outgain=%s, offset=%s)" % (
                self.buf(ctx, depth - 1), self.pick(("overdrive", "clip")), self.lit(0.0, 1.0, 3), self.lit(0.1, 1.0, 3), self.lit(-0.3, 0.3, 3))
        if n == "Portamento":
            return "Portamento(curve=%s, goal=%s, note_on, prev_note_on)" % (self.curve(), self.const(ctx, 0.0, 2.0))
        if n == "Curve":
            t, pts = 0.0, []
            for _ in range(int(self.rng.integers(2, 6))):
                pts.append("%r %r" % (round(t, 5), round(float(self.rng.uniform(0.0, 1.5)), 3)))   # (defcurve takes plain numbers)
                t += float(self.rng.uniform(0.0001, 0.0008))
            return "Curve(function=.%s, curve=defcurve %s end)" % (self.pick(("linear", "smoothstep")), "  ".join(pts))
        # a helper module: (freq: cob, note_on: boolean)
        return "%s(freq=%s, note_on)" % (n, self.pick(("freq", "(freq * %s)" % self.lit(0.5, 2.0), self.lit(50.0, 900.0, 1))) if "freq" in ctx["bufs"] else self.lit(50.0, 900.0, 1))

    # ---- buffer-valued expressions
    def buf(self, ctx, depth):
        r = self.rng.random()
        if depth <= 0 or r < 0.18:
            leaves = list(ctx["bufs"]) + list(ctx["lets"])
            if leaves and self.chance(0.7):
                return self.pick(leaves)
            return self.module_call(ctx, 0)
        if r < 0.45:
            return self.module_call(ctx, depth)
        if r < 0.75:
            op = self.pick(("+", "-", "*"))
            a = self.buf(ctx, depth - 1)
            b = self.buf(ctx, depth - 1) if self.chance(0.5) else self.const(ctx)
            if self.chance(0.3):
                a, b = b, a
            return "(%s %s %s)" % (a, op, b)
        if r < 0.80:
            return "(%s / (2 + cos(%s)))" % (self.buf(ctx, depth - 1), self.buf(ctx, depth - 1))
        if r < 0.84:
            return "(%s / %s)" % (self.buf(ctx, depth - 1), self.lit(0.3, 3.0))
        fn = self.pick(("min", "max", "abs", "sin", "cos", "sqrt", "pow", "neg"))
        a = self.buf(ctx, depth - 1)
        if fn in ("min", "max"):
            return "%s(%s, %s)" % (fn, a, self.const(ctx) if self.chance(0.6) else self.buf(ctx, depth - 1))
        if fn == "sqrt":
            return "sqrt(abs(%s))" % a
        if fn == "pow":
            return "pow(abs(%s) + 0.5, %s)" % (a, self.lit(-1.5, 2.5, 2))
        if fn == "neg":
            return "(-%s)" % a
        return "%s(%s)" % (fn, a)

    def body(self, ctx, depth, indent="    "):
        lines = []
        for i in range(int(self.rng.integers(0, 3))):
            name = "%s%d" % (ctx["prefix"], i)
            lines.append("%s%s = %s" % (indent, name, self.buf(ctx, depth)))
            ctx["lets"].append(name)
        if ctx.get("delay_ok") and self.chance(0.35):
            n = int(self.rng.integers(1, 40))
            inner = dict(ctx, bufs=list(ctx["bufs"]) + ["feedback"], lets=list(ctx["lets"]), delay_ok=False)
            lines.append("%swet = delay %d begin" % (indent, n))
            lines.append("%s    out (feedback * %s + %s)" % (indent, self.lit(0.1, 0.6), self.buf(inner, max(depth - 1, 0))))
            lines.append("%s    feedback (%s + feedback * %s)" % (indent, self.buf(inner, max(depth - 1, 0)), self.lit(0.1, 0.5)))
            lines.append("%send" % indent)
            ctx["lets"].append("wet")
        if ctx.get("track_ok") and self.chance(0.3):
            # a track: note events at times inside the buffers painted (two buffers of 96 / 256 frames at 48 kHz), the body
            # painted sub-span by sub-span by the NoteTracker / Trigger walk; builtin calls inside begin per sub-span
            inner = dict(ctx, consts=list(ctx["consts"]) + ["pitch"], lets=list(ctx["lets"]), track_ok=False, delay_ok=False, prefix=ctx["prefix"] + "k")
            lines.append("%strk = from deftrack" % indent)
            lines.append("%s    pitch: constant," % indent)
            lines.append("%s    note_on: boolean," % indent)
            lines.append("%sbegin" % indent)
            t = 0.0 if self.chance(0.5) else float(self.rng.uniform(0.0, 0.001))
            for _ in range(int(self.rng.integers(1, 6))):
                lines.append("%s    %.6f (pitch=%s, note_on=%s)" % (indent, t, self.lit(100.0, 1500.0, 1), self.pick(("true", "true", "false"))))
                t += float(self.rng.uniform(0.00002, 0.0015))
            lines.append("%send, %s begin" % (indent, self.lit(0.5, 2.0, 3)))
            for l in self.track_body(inner, max(depth - 1, 1), indent + "    "):
                lines.append(l)
            lines.append("%send" % indent)
            ctx["lets"].append("trk")
        for _ in range(int(self.rng.integers(1, 3))):
            lines.append("%sout %s" % (indent, self.buf(ctx, depth)))
        return lines

    def track_body(self, ctx, depth, indent):
        lines = []
        if self.chance(0.5):
            lines.append("%sscale = pitch / 1000" % indent)
            ctx["consts"] = list(ctx["consts"]) + ["scale"]
        r = self.rng.random()
        if r < 0.4:
            osc = self.pick(("SineOsc(freq=pitch, phase=0)", "PulseOsc(freq=pitch, color=0.5)", "TriSawOsc(freq=pitch, color=%s)" % self.lit(0.0, 1.0, 2)))
            lines.append("%sout %s * Envelope(attack=%s, decay=%s, release=%s, sustain_volume=%s, note_on)" % (
                indent, osc, self.curve(), self.curve(), self.curve(), self.lit(0.2, 1.0, 2)))
        else:
            lines.append("%sout %s" % (indent, self.buf(ctx, depth)))
        return lines


def generate(seed):
    """-> (script text, main module name, param kinds in order)."""
    g = Gen(seed)
    text, helpers = [], []
    for h in range(int(g.rng.integers(0, 3))):
        name = "Helper%d" % h
        ctx = {"bufs": ["freq"], "consts": [], "lets": [], "helpers": list(helpers), "prefix": "h%d_" % h}
        text += ["%s = defmodule" % name, "    freq: cob,", "    note_on: boolean,", "begin"] + g.body(ctx, 2) + ["end", ""]
        helpers.append(name)
    ctx = {"bufs": ["freq", "x"], "consts": ["k"], "lets": [], "helpers": helpers, "prefix": "m", "portamento": True, "delay_ok": True, "track_ok": True}
    text += ["Main = defmodule", "    freq: cob,", "    x: waveform,", "    k: constant,", "    note_on: boolean,", "    prev_note_on: boolean,", "begin"]
    text += g.body(ctx, 3) + ["end", ""]
    return "\n".join(text), "Main"


def schedule(seed, F=F):
    """Random paints of one buffer: [(start, end, note_id_changed, params)], spans in order, some empty."""
    rng = np.random.default_rng(seed + 77777)
    cuts = sorted(int(c) for c in rng.integers(0, F + 1, int(rng.integers(1, 4))))
    bounds = [0] + cuts + [F]
    if rng.random() < 0.3:
        bounds = [0, F]
    kind = rng.integers(4)
    if kind == 0:
        freq = np.float32(rng.uniform(30, 3000))                                   # one constant
    elif kind == 1:
        freq = rng.uniform(30, 3000, V).astype(np.float32)                         # a constant per voice
    elif kind == 2:
        freq = rng.uniform(30, 3000, (V, F)).astype(np.float32)                    # an image
    else:
        freq = rng.uniform(30, 3000, V).astype(np.float32)
        freq[rng.integers(V)] = np.float32(rng.choice([4.0e12, -3.0e13, 0.0, -440.0, 1.0e5]))   # one odd voice
    x = rng.uniform(-1.5, 1.5, (V, F)).astype(np.float32)
    k = np.float32(rng.uniform(0.1, 2.0)) if rng.random() < 0.5 else rng.uniform(0.1, 2.0, V).astype(np.float32)
    paints, prev = [], np.zeros(V, bool)
    for a, b in zip(bounds, bounds[1:]):
        on = rng.random(V) < rng.choice([0.0, 0.5, 0.9, 1.0])
        nic = (rng.random(V) < 0.3) & on if rng.random() < 0.7 else bool(rng.random() < 0.3)
        paints.append((a, b, nic, {"sample_rate": SR, "freq": freq, "x": x, "k": k, "note_on": on, "prev_note_on": prev.copy()}))
        prev = on
    return paints


def _per_voice(value, v):
    if isinstance(value, np.ndarray) and value.ndim >= 1 and value.shape[0] == V:
        x = value[v]
        if isinstance(x, np.ndarray):
            return x
        return bool(x) if value.dtype == np.bool_ else np.float32(x)
    return value


def _device_value(value):
    import torch
    from tests.util import to_image
    if isinstance(value, np.ndarray) and value.ndim == 2:
        return to_image(value)
    if isinstance(value, np.ndarray) and value.dtype == np.bool_:
        return torch.from_numpy(value.astype(np.uint8)).cuda()
    if isinstance(value, np.ndarray):
        return torch.from_numpy(value.astype(np.float32)).cuda()
    return value


def run_case(ctx, seed, buffers=2, F=F, ranges=None, text=None, tolerant=False, worst=None, roles=None):
    """One generated module, `buffers` consecutive buffers of random paints; raises AssertionError with the script text on
    a mismatch.  `ranges`: ZH_SCRIPT_RANGES for the case (the library reads it per paint under ZH_ENV_LIVE=1) -- with
    F >= 128 the kernels that allow it are launched as that many frame ranges.  `text`: a given script (module `Main` with
    the generator's params) instead of the generated one; `seed` then only picks the paints.  Returns the script text.
    `tolerant`: paint with ZH_PAINT_TOLERANT and ask, per voice and buffer, for every sample within 1e-5 of the largest of the voice's peak, its inputs' magnitudes and 1
    and the same finite / non-finite pattern instead of bits (kernels without a tolerant sine still answer bit for bit); `worst`
    (a one-element list) collects the largest ratio seen.  `roles`: 1 = every paint through the role-wave form (zs_paint_pc_<name>,
    dispatch row script_pc), 0 = never; the kernels that ran are checked through zh_last_form."""
    import os
    import torch
    from oracle import zangscript as zs
    from oracle import zs_interp
    from tests.util import from_image, to_image
    from zang_amd import script, zang
    text, name = generate(seed) if text is None else (text, "Main")
    from tests import util
    old = os.environ.get("ZH_FORMS")
    rows = {}
    if ranges is not None:
        rows["script_ranges"] = ranges
    if roles is not None:
        rows["script_pc"] = roles
    if rows:
        os.environ["ZH_FORMS"] = util.forms_env(**rows)["ZH_FORMS"]
    from zang_amd import zscript_native as native
    prog = script.ScriptProgram(text, ctx, only=[name], **({"forms": native.FORM_ROLES} if roles else {}))
    has_roles = ("zs_paint_pc_" + name) in prog.hip_source
    try:
        mod = prog.module(name, V, seed)
        voices = zs_interp.make_voices(zs.compile(text, "fuzz"), name, V, seed)
        order = [p[0] for p in mod.params]
        for b in range(buffers):
            base = np.zeros((V, F), np.float32) if b % 2 == 0 else np.random.default_rng(seed + b).uniform(-1, 1, (V, F)).astype(np.float32)
            ref = base.copy()
            img = to_image(base)
            scale = np.ones(V)              # tolerant: the largest magnitude on the voice's signal path -- its inputs too (terms that cancel:
                                            # `out 0.84 - freq` then `out freq - x` leaves 1.4 out of sums rounded at 1,386, seed 209)
            for start, end, nic, params in schedule(seed * 16 + b, F):
                for kk, vv in params.items():
                    if kk in order and kk != "sample_rate" and isinstance(vv, (np.ndarray, float, np.floating)) and getattr(vv, "dtype", np.dtype(np.float32)) != np.bool_:
                        mag = np.abs(np.asarray(vv, np.float64))
                        mag = np.where(np.isfinite(mag), mag, 0.0)
                        scale = np.maximum(scale, mag.reshape(V, -1).max(axis=1) if mag.ndim >= 1 and mag.shape[0] == V else float(mag.max()))
                dev = {kk: _device_value(vv) for kk, vv in params.items() if kk in order}
                nic_dev = torch.from_numpy(nic.astype(np.uint8)).cuda() if isinstance(nic, np.ndarray) else nic
                mod.paint(zang.Span(start, end), [img], None, nic_dev, dev, tolerant=tolerant)
                if roles is not None:
                    ran = ctx.last_form()
                    assert any("zs_paint_pc_" in kname for kname in ran) == bool(roles and has_roles), (roles, has_roles, ran)
                for v in range(V):
                    voices[v].paint(start, end, ref[v], bool(nic[v]) if isinstance(nic, np.ndarray) else nic,
                                    [_per_voice(params[kk], v) for kk in order])
            ctx.sync()
            got = from_image(img)
            if tolerant and "ZS_T" in prog.hip_source:
                a, g = ref.astype(np.float64), got.astype(np.float64)
                with np.errstate(invalid="ignore", over="ignore"):
                    fin = np.isfinite(a)
                    same_pattern = np.array_equal(fin, np.isfinite(g))
                    peak = np.maximum(np.where(fin, np.abs(a), 0.0).max(axis=1), scale)
                    ratio = float((np.where(fin & np.isfinite(g), np.abs(a - g), 0.0).max(axis=1) / peak).max())
                if worst is not None:
                    worst[0] = max(worst[0], ratio)
                if not same_pattern or ratio > 1e-5:
                    raise AssertionError("seed %d buffer %d, ZH_PAINT_TOLERANT: off by %.2e of the voice's scale (finite patterns %s)\n%s"
                                         % (seed, b, ratio, "equal" if same_pattern else "DIFFER", text))
                # the next buffer starts from the tolerant kernel's own state: the oracle's voices are not re-synchronised (a Filter's
                # state and a delay ring carry the difference on, scaled like the signal)
                continue
            if not np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
                bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
                v0, f0 = bad[0]
                raise AssertionError("seed %d buffer %d: %d samples differ (voices %s, frames %s), first at voice %d frame %d: got %r want %r\n%s"
                                     % (seed, b, len(bad), sorted(set(int(q) for q in bad[:, 0]))[:12], sorted(set(int(q) for q in bad[:, 1]))[:40],
                                        v0, f0, got[v0, f0], ref[v0, f0], text))
    finally:
        prog.close()
        if rows:
            if old is None:
                del os.environ["ZH_FORMS"]
            else:
                os.environ["ZH_FORMS"] = old
    return text
