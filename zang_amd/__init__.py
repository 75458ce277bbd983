"""zang_amd -- zang's module paint() hot path on MI355X (gfx950).

`zang_amd.zang` mirrors the reference's `zang` namespace (Span, basics ops,
ConstantOrBuffer, PaintCurve) and `zang_amd.modules` its `modules` namespace, both over
the C ABI of libzang_hip.so (include/zang_hip.h).  There is no CPU implementation in this
package: importing it without the built library raises.
"""
from . import abi

abi.load()  # fail loudly if libzang_hip.so is missing or incomplete

from . import zang, modules  # noqa: E402
from .runtime import Context, default_context  # noqa: E402

__all__ = ["abi", "zang", "modules", "Context", "default_context"]
