"""The builtin packages the compiler starts from (src/zangscript/builtins.zig:153-185): module
names, their Params in declaration order, and the enums those Params use."""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass(frozen=True)
class EnumValue:
    label: str
    payload: str = "none"          # "none" | "f32"


@dataclass(frozen=True)
class BuiltinEnum:
    name: str
    zig_name: str
    values: tuple


@dataclass(frozen=True)
class ParamType:
    """boolean | buffer | constant | constant_or_buffer | curve | one_of(enum)  (parse.zig:35-42)."""
    kind: str
    enum: Optional[BuiltinEnum] = None


BOOLEAN = ParamType("boolean")
BUFFER = ParamType("buffer")
CONSTANT = ParamType("constant")
COB = ParamType("constant_or_buffer")
CURVE = ParamType("curve")


@dataclass(frozen=True)
class ModuleParam:
    name: str
    param_type: ParamType


@dataclass(frozen=True)
class BuiltinModule:
    name: str
    params: tuple
    num_temps: int = 0
    num_outputs: int = 1


@dataclass(frozen=True)
class BuiltinPackage:
    zig_package_name: str
    zig_import_path: str
    builtins: tuple
    enums: tuple


def _enum(name, zig_name, *labels):
    return BuiltinEnum(name, zig_name, tuple(EnumValue(*(l if isinstance(l, tuple) else (l,))) for l in labels))


PAINT_CURVE = _enum("PaintCurve", "zang.PaintCurve", "instantaneous", ("linear", "f32"), ("squared", "f32"), ("cubed", "f32"))
INTERPOLATION = _enum("InterpolationFunction", "mod.Curve.InterpolationFunction", "linear", "smoothstep")
DISTORTION_TYPE = _enum("DistortionType", "mod.Distortion.Type", "overdrive", "clip")
FILTER_TYPE = _enum("FilterType", "mod.Filter.Type", "bypass", "low_pass", "band_pass", "high_pass", "notch", "all_pass")
NOISE_COLOR = _enum("NoiseColor", "mod.Noise.Color", "white", "pink")


def _one_of(e):
    return ParamType("one_of", e)


def _mod(name, *params):
    return BuiltinModule(name, tuple(ModuleParam(n, t) for n, t in params))


zang_builtin_package = BuiltinPackage("zang", "zang", (), (PAINT_CURVE,))

# Params in the modules' declaration order (src/modules/*.zig `pub const Params`); the list and its
# order are builtins.zig:161-176 (Sampler is commented out there too).
modules_builtin_package = BuiltinPackage("mod", "modules", (
    _mod("Curve", ("sample_rate", CONSTANT), ("function", _one_of(INTERPOLATION)), ("curve", CURVE)),
    _mod("Cycle", ("sample_rate", CONSTANT), ("speed", COB)),
    _mod("Decimator", ("sample_rate", CONSTANT), ("input", BUFFER), ("fake_sample_rate", CONSTANT)),
    _mod("Distortion", ("input", BUFFER), ("type", _one_of(DISTORTION_TYPE)), ("ingain", CONSTANT),
         ("outgain", CONSTANT), ("offset", CONSTANT)),
    _mod("Envelope", ("sample_rate", CONSTANT), ("attack", _one_of(PAINT_CURVE)), ("decay", _one_of(PAINT_CURVE)),
         ("release", _one_of(PAINT_CURVE)), ("sustain_volume", CONSTANT), ("note_on", BOOLEAN)),
    _mod("Filter", ("input", BUFFER), ("type", _one_of(FILTER_TYPE)), ("cutoff", COB), ("res", COB)),
    _mod("Gate", ("note_on", BOOLEAN)),
    _mod("Noise", ("color", _one_of(NOISE_COLOR))),
    _mod("Portamento", ("sample_rate", CONSTANT), ("curve", _one_of(PAINT_CURVE)), ("goal", CONSTANT),
         ("note_on", BOOLEAN), ("prev_note_on", BOOLEAN)),
    _mod("PulseOsc", ("sample_rate", CONSTANT), ("freq", COB), ("color", CONSTANT)),
    _mod("SineOsc", ("sample_rate", CONSTANT), ("freq", COB), ("phase", COB)),
    _mod("TriSawOsc", ("sample_rate", CONSTANT), ("freq", COB), ("color", CONSTANT)),
), (INTERPOLATION, DISTORTION_TYPE, FILTER_TYPE, NOISE_COLOR))

DEFAULT_PACKAGES = (zang_builtin_package, modules_builtin_package)
