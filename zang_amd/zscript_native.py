"""The zangscript compiler inside libzang_hip.so (csrc/zscript_front.hip, zscript_emit.hip) through its C ABI
(zh_zscript_*): the one front-end of the product (zang_amd.script).  (oracle/zangscript/ holds an independent Python
restatement of the reference's front-end; it is test infrastructure and is never imported from here.)"""
import ctypes as C

from . import abi

# the reference's two builtin packages (src/zangscript/builtins.zig:145-185), as the bits zh_zscript_compile takes
PACKAGE_ZANG, PACKAGE_MODULES = 1, 2
DEFAULT_PACKAGES = PACKAGE_ZANG | PACKAGE_MODULES

# enum labels in declaration order = the values include/zang_hip.h gives them (builtins.zig:147-159; painter.zig:25-30,
# Curve.zig:6-9, Distortion.zig:8-11, Filter.zig:10-17, Noise.zig:11-14)
ENUM_LABELS = {
    "PaintCurve": ["instantaneous", "linear", "squared", "cubed"],
    "InterpolationFunction": ["linear", "smoothstep"],
    "DistortionType": ["overdrive", "clip"],
    "FilterType": ["bypass", "low_pass", "band_pass", "high_pass", "notch", "all_pass"],
    "NoiseColor": ["white", "pink"],
}


class NativeScriptError(Exception):
    """A compile error with the reference's rendering (file:line:col: message, source line, carets)."""


def _package_bits(packages):
    if isinstance(packages, int):
        return packages
    bits = 0
    for p in packages:                                   # objects with the reference's package names ("zang", "mod")
        name = getattr(p, "zig_package_name", p)
        if name == "zang":
            bits |= PACKAGE_ZANG
        elif name == "mod":
            bits |= PACKAGE_MODULES
        else:
            raise ValueError("the native compiler knows the two builtin packages only")
    return bits


FORM_ROLES = 1          # ZH_ZSCRIPT_FORM_ROLES: a role-wave kernel for every module that has more than one role
FORM_ROLES_WORTH = 2    # ZH_ZSCRIPT_FORM_ROLES_WORTH: only where the emitter expects it to pay


class NativeScript:
    def __init__(self, contents, filename="script.txt", packages=DEFAULT_PACKAGES):
        self.lib = abi.load()
        h = C.c_void_p()
        err = C.create_string_buffer(1 << 14)
        rc = self.lib.zh_zscript_compile(contents.encode(), filename.encode(), _package_bits(packages), C.byref(h), err, len(err))
        if rc != 0:
            raise NativeScriptError(err.value.decode(errors="replace"))
        self.handle = h

    def _text(self, fn, *args):
        p = C.c_void_p()
        abi.check(fn(self.handle, *args, C.byref(p)), fn.__name__)
        try:
            return C.string_at(p).decode()
        finally:
            self.lib.zh_zscript_free_text(p)

    def generate_zig(self):
        return self._text(self.lib.zh_zscript_generate_zig)

    def generate_hip(self, only=None, unroll=0, forms=0):
        """-> (text, meta) with meta[name] = {"state_words", "params": [(name, kind, enum name)], "noise_fields"} or {"error"}
        forms: FORM_ROLES = also the role-wave kernels for few voices (zs_paint_pc_<name>, include/zang_hip.h)"""
        text = self._text(self.lib.zh_zscript_generate_hip_forms, None if only is None else ",".join(only).encode(), int(unroll), int(forms))
        meta = {}
        name, err = C.create_string_buffer(256), C.create_string_buffer(1024)
        kind, en = C.create_string_buffer(64), C.create_string_buffer(64)
        for i in range(self.lib.zh_zscript_module_count(self.handle)):
            words, noise, npar = C.c_uint32(), C.c_uint32(), C.c_uint32()
            abi.check(self.lib.zh_zscript_module_info(self.handle, i, name, len(name), C.byref(words), C.byref(noise), C.byref(npar), err, len(err)),
                      "zh_zscript_module_info")
            if err.value:
                meta[name.value.decode()] = {"error": err.value.decode()}
                continue
            params = []
            pname = C.create_string_buffer(256)
            for p in range(npar.value):
                abi.check(self.lib.zh_zscript_module_param(self.handle, i, p, pname, len(pname), kind, len(kind), en, len(en)), "zh_zscript_module_param")
                params.append((pname.value.decode(), kind.value.decode(), en.value.decode() or None))
            nt = C.c_uint32()
            abi.check(self.lib.zh_zscript_module_num_temps(self.handle, i, C.byref(nt)), "zh_zscript_module_num_temps")
            meta[name.value.decode()] = {"state_words": words.value, "params": params, "noise_fields": noise.value, "num_temps": nt.value}
        return text, meta

    def close(self):
        if self.handle:
            self.lib.zh_zscript_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
