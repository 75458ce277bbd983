"""zangscript modules on the GPU: compile a script (the C++ front-end in libzang_hip.so), load its fused kernels
(zh_script_load -> hiprtc) and paint them through the module call shape of the reference
(SineOsc.zig:22-31; generated modules: codegen_zig.zig:558-563)."""
import ctypes as C
import os

import numpy as np
import torch

from . import abi, zscript_native as native
from .runtime import as_bool, as_buf, default_context


class ScriptCompileError(Exception):
    pass


class HipBackendError(Exception):
    """The script compiled, but this module uses something the HIP backend does not generate (the message says what)."""


def compile_hip(hip_source):
    """hiprtc compile only (works without a GPU); returns the gfx950 code object's size."""
    lib = abi.load()
    code, n = C.c_void_p(), C.c_size_t()
    log = C.create_string_buffer(1 << 16)
    rc = lib.zh_script_compile(hip_source.encode(), C.byref(code), C.byref(n), log, len(log))
    if rc != 0:
        raise ScriptCompileError("hiprtc failed (%d):\n%s" % (rc, log.value.decode(errors="replace")))
    lib.zh_script_free_code(code)
    return n.value


class ScriptProgram:
    """One script: front-end result + the loaded hipModule."""

    def __init__(self, text, ctx=None, filename="script.txt", only=None, forms=native.FORM_ROLES_WORTH, hip_patch=None, code_cache=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.text, self.filename = text, filename
        try:                                                    # the C++ compiler in libzang_hip.so (zh_zscript_*)
            compiled = native.NativeScript(text, filename)
        except native.NativeScriptError as e:
            raise ScriptCompileError(str(e))
        # ZH_SCRIPT_UNROLL: frames per unrolled chunk of the generated kernels (an experiment knob; 0 / unset = the emitter's choice by body size)
        # forms: FORM_ROLES_WORTH = modules the emitter expects to gain from it also as a role-wave kernel for few voices (zs_paint_pc_<name>;
        # the library picks per paint); FORM_ROLES = every module (the parity tests force the form)
        forms = int(os.environ.get("ZH_SCRIPT_FORMS", forms))       # (an experiment knob like ZH_SCRIPT_UNROLL)
        self.hip_source, self.meta = compiled.generate_hip(only=only, unroll=int(os.environ.get("ZH_SCRIPT_UNROLL", "0")), forms=forms)
        if hip_patch is not None:                               # experiments (tools/exp/role_probe.py): the generated text, edited
            self.hip_source = hip_patch(self.hip_source)
        h = C.c_void_p()
        log = C.create_string_buffer(1 << 16)
        # code_cache = a directory: the compiled code object is kept there under the hash of the generated text (+ the library's version)
        # and loaded with zh_script_load_code next time -- the reference's compile-once flow; hiprtc takes 1-4 s per module
        self.loaded_from_cache = False
        cache_file = None
        if code_cache is not None and hip_patch is None:
            import hashlib
            key = hashlib.sha256(self.hip_source.encode() + b"\0" + self.lib.zh_version()).hexdigest()[:32]
            cache_file = os.path.join(code_cache, "zs_%s.hsaco" % key)
            if os.path.exists(cache_file):
                blob = open(cache_file, "rb").read()
                if self.lib.zh_script_load_code(self.ctx.handle, blob, len(blob), C.byref(h)) == 0:
                    compiled.close()
                    self.handle, self._modules, self.loaded_from_cache = h, [], True
                    self.ctx._children.add(self)
                    return
        if cache_file is not None:
            code, n = C.c_void_p(), C.c_size_t()
            if self.lib.zh_script_compile(self.hip_source.encode(), C.byref(code), C.byref(n), log, len(log)) == 0:
                blob = C.string_at(code, n.value)
                self.lib.zh_script_free_code(code)
                os.makedirs(code_cache, exist_ok=True)
                tmp = cache_file + ".%d.tmp" % os.getpid()
                open(tmp, "wb").write(blob)
                os.replace(tmp, cache_file)
                if self.lib.zh_script_load_code(self.ctx.handle, blob, len(blob), C.byref(h)) == 0:
                    compiled.close()
                    self.handle, self._modules = h, []
                    self.ctx._children.add(self)
                    return
        rc = self.lib.zh_script_load(self.ctx.handle, self.hip_source.encode(), C.byref(h), log, len(log))
        if rc != 0 and forms and hip_patch is None:
            # the lane kernels alone: a role-wave kernel hiprtc refuses must not take the patch away (none has been seen to)
            self.role_form_error = log.value.decode(errors="replace")
            self.hip_source, self.meta = compiled.generate_hip(only=only, unroll=int(os.environ.get("ZH_SCRIPT_UNROLL", "0")), forms=0)
            rc = self.lib.zh_script_load(self.ctx.handle, self.hip_source.encode(), C.byref(h), log, len(log))
        compiled.close()
        if rc != 0:
            raise ScriptCompileError("zh_script_load failed (%d):\n%s" % (rc, log.value.decode(errors="replace")))
        self.handle = h
        self._modules = []
        self.ctx._children.add(self)

    def module(self, name, n_voices, first_seed=0):
        m = self.meta.get(name)
        if m is None:
            raise KeyError("script exports no module named %r" % name)
        if "error" in m:
            raise HipBackendError("%s: %s" % (name, m["error"]))
        return ScriptModule(self, name, n_voices, first_seed)

    def close(self):
        if self.handle:
            for m in list(self._modules):
                m.close()
            self.lib.zh_script_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_ENUM_LABELS = native.ENUM_LABELS


class ScriptModule:
    """n_voices instances of one exported script module.  num_outputs = 1; the temps the generated Zig
    would need live in registers, so `temps` is accepted and ignored."""
    num_outputs = 1

    def __init__(self, program, name, n_voices, first_seed=0):
        self.program, self.name, self.n = program, name, n_voices
        self.meta = program.meta[name]
        self.params = self.meta["params"]            # [(name, kind, enum name)], sample_rate first
        self.lib = program.lib
        h = C.c_void_p()
        abi.check(self.lib.zh_script_module_create(program.handle, name.encode(), n_voices, self.meta["state_words"],
                                                   first_seed, C.byref(h)), "zh_script_module_create")
        self.handle = h
        program._modules.append(self)

    @property
    def num_temps(self):
        """What the generated Zig struct would ask for (codegen_zig.zig:518); the fused kernel needs none."""
        return self.meta["num_temps"]

    def _param(self, kind, enum, value, keep):
        p = abi.ScriptParam()
        dev = self.program.ctx.device
        if kind == "constant":
            p.kind = abi.SP_CONSTANT
            if torch.is_tensor(value):
                assert value.dtype == torch.float32 and value.numel() == self.n and value.is_contiguous()
                p.pf = value.data_ptr(); keep.append(value)
            else:
                p.f = float(value)
        elif kind == "boolean":
            p.kind = abi.SP_BOOLEAN
            if torch.is_tensor(value):
                assert value.dtype == torch.uint8 and value.numel() == self.n and value.is_contiguous()
                p.pb = value.data_ptr(); keep.append(value)
            else:
                p.u = 1 if value else 0
        elif kind in ("constant_or_buffer", "buffer"):
            p.kind = abi.SP_COB if kind == "constant_or_buffer" else abi.SP_BUFFER
            if isinstance(value, abi.Cob):                      # zang.constant(...) / zang.buffer(...)
                keep.append(value)
                if value.tag == abi.COB_BUFFER:
                    p.is_buffer, p.pf, p.stride = 1, value.buffer.ptr, value.buffer.stride
                else:
                    p.f, p.pf = value.constant.value, value.constant.per_voice
            elif torch.is_tensor(value) and value.dim() == 2:
                b = as_buf(value); keep.append(value)
                p.is_buffer, p.pf, p.stride = 1, b.ptr, b.stride
            elif torch.is_tensor(value):
                assert kind == "constant_or_buffer" and value.dtype == torch.float32 and value.numel() == self.n
                p.pf = value.data_ptr(); keep.append(value)
            else:
                assert kind == "constant_or_buffer", "a waveform param needs a [frames, voices] image"
                p.f = float(value)
        elif kind == "curve":
            p.kind = abi.SP_CURVE
            if torch.is_tensor(value):                          # device array of (value, t) pairs
                p.pf, p.u = value.data_ptr(), value.numel() // 2; keep.append(value)
            else:                                               # [(t, value), ...] like a defcurve block
                nodes = np.array([(v, t) for t, v in value], np.float32).reshape(-1)
                tns = torch.from_numpy(nodes).to(dev)
                p.pf, p.u = tns.data_ptr(), len(value); keep.append(tns)
        else:                                                   # one_of
            p.kind = abi.SP_ENUM
            if isinstance(value, abi.Curve):                    # the library's own zang.PaintCurve.* values
                if enum != "PaintCurve":
                    raise TypeError("a PaintCurve value for a %s param" % enum)
                if value.duration.per_voice:
                    raise ValueError("a script module takes one PaintCurve duration for all voices, not a per-voice array")
                p.u, p.f = value.tag, value.duration.value
            else:                                               # ".label" or (".label", payload)
                label, payload = (value, None) if isinstance(value, str) else value
                p.u = _ENUM_LABELS[enum].index(label.lstrip("."))
                p.f = float(payload) if payload is not None else 0.0
        return p

    def paint(self, span, outputs, temps, note_id_changed, params, zero_first=False, tolerant=False):
        """params: dict by name (sample_rate included), like the reference's Params struct literal.
        tolerant=True: ZH_PAINT_TOLERANT -- the module's sines that reach the output through scaling and adding alone (not another
        oscillator's freq / phase, a distortion, a divisor ...: csrc/zscript_emit.hip) are evaluated in f32; all state that is not a
        Filter's or a delay ring's stays exact."""
        keep = []
        arr = (abi.ScriptParam * abi.SCRIPT_MAX_PARAMS)()
        for i, (name, kind, enum) in enumerate(self.params):
            if name not in params:
                raise KeyError("missing param %r" % name)
            arr[i] = self._param(kind, enum, params[name], keep)
        extra = set(params) - {n for n, _, _ in self.params}
        if extra:
            raise KeyError("module %s has no param(s) %s" % (self.name, sorted(extra)))
        ob = as_buf(outputs[0])
        nic = as_bool(note_id_changed)
        flags = (abi.PAINT_ZERO_FIRST if zero_first else 0) | (abi.PAINT_TOLERANT if tolerant else 0)
        abi.check(self.lib.zh_script_module_paint(self.handle, span.start, span.end, C.byref(ob), nic, arr, len(self.params), flags),
                  "zh_script_module_paint")
        self._keep = (keep, outputs, note_id_changed)

    @property
    def frame_ranges_ok(self):
        """True when the library may launch this module's kernel as frame ranges at small voice counts (no delay ring)."""
        return bool(self.lib.zh_script_module_ranges_ok(self.handle))

    def get_state(self):
        a = np.zeros((self.meta["state_words"], self.n), np.uint32)
        abi.check(self.lib.zh_script_module_get_state(self.handle, a.ctypes.data_as(C.c_void_p)), "zh_script_module_get_state")
        return a

    def set_state(self, a):
        a = np.ascontiguousarray(a, np.uint32)
        assert a.shape == (self.meta["state_words"], self.n)
        abi.check(self.lib.zh_script_module_set_state(self.handle, a.ctypes.data_as(C.c_void_p)), "zh_script_module_set_state")

    def close(self):
        if self.handle:
            self.lib.zh_script_module_destroy(self.handle)
            self.handle = None
            if self in self.program._modules:
                self.program._modules.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
