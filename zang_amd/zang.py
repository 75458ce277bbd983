"""Host-side mirror of the `zang` namespace used on the paint path (src/zang.zig:1-41):
Span, the basics.zig buffer ops, ConstantOrBuffer and PaintCurve -- same names and
argument order as the reference, acting on [frame][voice] device images of many voices.
"""
from dataclasses import dataclass

from . import abi
from .runtime import as_buf, as_bool, as_f32, default_context


@dataclass(frozen=True)
class Span:
    """zang.Span (src/zang/basics.zig:3-10)."""
    start: int
    end: int

    @staticmethod
    def init(start, end):
        return Span(start, end)


# ---- ConstantOrBuffer (src/zang/constant_or_buffer.zig:4-15)
def constant(x):
    c = abi.Cob(abi.COB_CONSTANT, 0, as_f32(x), abi.Buf())
    c._keep = x   # ctypes copies the struct by value: keep the tensor behind per_voice alive
    return c


def buffer(buf):
    c = abi.Cob(abi.COB_BUFFER, 0, abi.F32(), as_buf(buf))
    c._keep = buf
    return c


# ---- PaintCurve (src/zang/painter.zig:25-30)
class PaintCurve:
    instantaneous = abi.Curve(abi.CURVE_INSTANTANEOUS, 0, abi.F32())

    # ctypes copies nested structs by value, so the tensor behind a per-voice duration is
    # pinned on the returned struct itself.
    @staticmethod
    def _mk(tag, duration):
        c = abi.Curve(tag, 0, as_f32(duration))
        c._keep = duration
        return c

    @staticmethod
    def linear(duration):
        return PaintCurve._mk(abi.CURVE_LINEAR, duration)

    @staticmethod
    def squared(duration):
        return PaintCurve._mk(abi.CURVE_SQUARED, duration)

    @staticmethod
    def cubed(duration):
        return PaintCurve._mk(abi.CURVE_CUBED, duration)


def _ctx(ctx):
    return ctx or default_context()


# ---- basics.zig:12-78, same names/argument order
def zero(span, dest, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_zero(c.handle, span.start, span.end, as_buf(dest)), "zh_zero")


def set(span, dest, a, ctx=None):  # noqa: A001 (mirrors zang.set)
    c = _ctx(ctx); abi.check(c.lib.zh_set(c.handle, span.start, span.end, as_buf(dest), as_f32(a)), "zh_set")


def copy(span, dest, src, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_copy(c.handle, span.start, span.end, as_buf(dest), as_buf(src)), "zh_copy")


def add(span, dest, a, b, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_add(c.handle, span.start, span.end, as_buf(dest), as_buf(a), as_buf(b)), "zh_add")


def addInto(span, dest, src, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_add_into(c.handle, span.start, span.end, as_buf(dest), as_buf(src)), "zh_add_into")


def addScalar(span, dest, a, b, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_add_scalar(c.handle, span.start, span.end, as_buf(dest), as_buf(a), as_f32(b)), "zh_add_scalar")


def addScalarInto(span, dest, a, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_add_scalar_into(c.handle, span.start, span.end, as_buf(dest), as_f32(a)), "zh_add_scalar_into")


def multiply(span, dest, a, b, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_multiply(c.handle, span.start, span.end, as_buf(dest), as_buf(a), as_buf(b)), "zh_multiply")


def multiplyWith(span, dest, a, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_multiply_with(c.handle, span.start, span.end, as_buf(dest), as_buf(a)), "zh_multiply_with")


def multiplyScalar(span, dest, a, b, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_multiply_scalar(c.handle, span.start, span.end, as_buf(dest), as_buf(a), as_f32(b)), "zh_multiply_scalar")


def multiplyWithScalar(span, dest, a, ctx=None):
    c = _ctx(ctx); abi.check(c.lib.zh_multiply_with_scalar(c.handle, span.start, span.end, as_buf(dest), as_f32(a)), "zh_multiply_with_scalar")


def mixdownVoices(span, dst, src, zero_first=False, sequential=False, ctx=None):
    """dst[f] += sum over voices of src[f][v]: V x zang.addInto onto one mix buffer.
    sequential=True adds the voices in index order in f32, bit-identical to the reference's
    successive `+=` paints (small voice counts)."""
    c = _ctx(ctx)
    flags = (abi.PAINT_ZERO_FIRST if zero_first else abi.PAINT_ADD) | (abi.MIX_SEQUENTIAL if sequential else 0)
    abi.check(c.lib.zh_mixdown_voices(c.handle, span.start, span.end, dst.data_ptr(), as_buf(src), flags), "zh_mixdown_voices")


class AudioFormat:
    """zang.AudioFormat (src/zang/mixdown.zig:3-6)."""
    signed8 = abi.AUDIO_SIGNED8
    signed16_lsb = abi.AUDIO_SIGNED16_LSB


def mixDown(dst, mix_buffer, audio_format, num_channels, channel_index, vol, ctx=None):
    """zang.mixDown (src/zang/mixdown.zig:8-24): dst uint8 CUDA tensor, mix_buffer float32 CUDA tensor [n]."""
    c = _ctx(ctx)
    n = mix_buffer.numel()
    bps = 2 if audio_format == abi.AUDIO_SIGNED16_LSB else 1
    assert dst.numel() == n * bps * num_channels            # mixdown.zig:35,66
    abi.check(c.lib.zh_mix_down(c.handle, dst.data_ptr(), mix_buffer.data_ptr(), n, audio_format, num_channels,
                                channel_index, float(vol)), "zh_mix_down")


# ---- event scheduling (src/zang/notes.zig, src/zang/trigger.zig): re-exported like src/zang.zig does
from .notes import Impulse, Notes, Trigger, ImpulsesAndParamses  # noqa: E402,F401
