"""Context and device-image plumbing above the C ABI.

PyTorch is used only for device memory and streams: sample images are float32 CUDA
tensors of shape [frames, voices] (voice contiguous), passed to the C ABI by data_ptr().
"""
import ctypes as C
import weakref

import torch

from . import abi

_default = None


class Context:
    """One zh_ctx bound to a torch device; launches go to torch's current stream."""

    def __init__(self, device=0, adopt_torch_stream=True):
        if not torch.cuda.is_available():
            raise abi.ZangHipError("zang_amd needs a HIP device (torch.cuda.is_available() is False); "
                                   "there is no CPU fallback")
        self.lib = abi.load()
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        abi.check(self.lib.zh_create(C.byref(h), device), "zh_create")
        self.handle = h
        self._children = weakref.WeakSet()     # modules / graphs created on this context
        if adopt_torch_stream:
            self.use_stream(torch.cuda.current_stream(self.device))

    def use_stream(self, stream):
        abi.check(self.lib.zh_set_stream(self.handle, C.c_void_p(stream.cuda_stream)), "zh_set_stream")
        self._stream = stream

    def sync(self):
        abi.check(self.lib.zh_sync(self.handle), "zh_sync")

    def last_form(self):
        """Kernels the last paint (or other entry point) on this context launched, in launch order (zh_last_form)."""
        buf = C.create_string_buffer(512)
        abi.check(self.lib.zh_last_form(self.handle, buf, 512), "zh_last_form")
        return [k for k in buf.value.decode().split(",") if k]

    def forms(self):
        """The library's dispatch table: {name: (default, current, doc)} (zh_form_info, csrc/dispatch.hip)."""
        out = {}
        for i in range(self.lib.zh_form_count()):
            name, doc, d, c = C.c_char_p(), C.c_char_p(), C.c_long(), C.c_long()
            abi.check(self.lib.zh_form_info(i, C.byref(name), C.byref(d), C.byref(c), C.byref(doc)), "zh_form_info")
            out[name.value.decode()] = (d.value, c.value, doc.value.decode())
        return out

    def close(self):
        """Destroy the context; modules and graphs created on it are closed first (their C objects
        hold a pointer to the context)."""
        if self.handle:
            for child in list(self._children):
                child.close()
            self.lib.zh_destroy(self.handle)
            self.handle = None

    def capture(self, fn, coalesce=False):
        """Record everything fn() enqueues on this context into a hipGraph; returns a Graph.  `coalesce`:
        ZH_CAPTURE_COALESCE -- nothing but calls on this context touches its stream inside fn(), and the library may record
        consecutive independent paints as one launch (include/zang_hip.h)."""
        abi.check(self.lib.zh_graph_begin_capture_flags(self.handle, abi.ZH_CAPTURE_COALESCE if coalesce else 0), "zh_graph_begin_capture_flags")
        g = C.c_void_p()
        try:
            fn()
        except BaseException:
            # leave capture mode and drop the partial recording before passing the error on
            if self.lib.zh_graph_end_capture(self.handle, C.byref(g)) == abi.ZH_OK and g:
                self.lib.zh_graph_destroy(g)
            raise
        abi.check(self.lib.zh_graph_end_capture(self.handle, C.byref(g)), "zh_graph_end_capture")
        return Graph(self, g)

    def image(self, frames, voices, fill=None, pad=None):
        """A [frame][voice] sample image (the device form of `voices` reference []f32 slices).
        `pad` = extra voices per row (row stride = voices + pad); None = image_row_pad(voices)."""
        pad = image_row_pad(voices) if pad is None else int(pad)
        if fill is None:
            t = torch.empty((frames, voices + pad), dtype=torch.float32, device=self.device)
        else:
            t = torch.full((frames, voices + pad), float(fill), dtype=torch.float32, device=self.device)
        return t[:, :voices] if pad else t


def image_row_pad(voices):
    """Row padding (in voices) Context.image gives an image by default: zh_buf_alloc's rule (ctx.hip) -- rows that
    are a multiple of 64 KiB get 4 KiB more, so that a lane's consecutive frames do not all map to one HBM bank."""
    return 1024 if voices and (voices * 4) % 65536 == 0 else 0


class Graph:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle
        ctx._children.add(self)

    def launch(self):
        abi.check(self.ctx.lib.zh_graph_launch(self.ctx.handle, self.handle), "zh_graph_launch")

    def info(self):
        """(nodes in the recorded graph, paint calls held back while recording, launches they became)"""
        n, p, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        abi.check(self.ctx.lib.zh_graph_info(self.handle, C.byref(n), C.byref(p), C.byref(l)), "zh_graph_info")
        return n.value, p.value, l.value

    def kernels(self):
        """[(kernel, launches)] a replay of the graph runs, in order of first launch while recording (zh_graph_kernels);
        "k_osc_const4[batch]" = the instantiation that paints several buffers per launch"""
        buf = C.create_string_buffer(2048)
        abi.check(self.ctx.lib.zh_graph_kernels(self.handle, buf, len(buf)), "zh_graph_kernels")
        out = []
        for item in buf.value.decode().split(","):
            if item:
                name, _, n = item.rpartition(" x")
                out.append((name, int(n)))
        return out

    def close(self):
        if self.handle:
            self.ctx.lib.zh_graph_destroy(self.handle)
            self.handle = None


def default_context():
    global _default
    if _default is None:
        _default = Context(torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return _default


def as_buf(t):
    """torch [frames, voices] float32 CUDA tensor (row stride >= voices) -> zh_buf."""
    if isinstance(t, abi.Buf):
        return t
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1)):
        raise ValueError("sample image must be a float32 CUDA tensor [frames, voices] with contiguous voices")
    stride = t.stride(0) if t.shape[0] > 1 else t.shape[1]
    b = abi.Buf(t.data_ptr(), t.shape[1], t.shape[0], max(stride, t.shape[1]), 0)
    b._keep = t
    return b


def as_f32(x):
    """float, or float32 CUDA tensor [n_voices] -> zh_f32."""
    if isinstance(x, abi.F32):
        return x
    if isinstance(x, torch.Tensor):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 1 and x.is_contiguous()):
            raise ValueError("per-voice f32 parameter must be a contiguous float32 CUDA tensor [n_voices]")
        f = abi.F32(0.0, 0, x.data_ptr())
        f._keep = x
        return f
    return abi.F32(float(x), 0, None)


def as_bool(x):
    """bool, or uint8/bool CUDA tensor [n_voices] -> zh_bool."""
    if isinstance(x, abi.Bool):
        return x
    if isinstance(x, torch.Tensor):
        if x.dtype == torch.bool:
            x = x.view(torch.uint8)
        if not (x.is_cuda and x.dtype == torch.uint8 and x.dim() == 1 and x.is_contiguous()):
            raise ValueError("per-voice bool parameter must be a contiguous uint8/bool CUDA tensor [n_voices]")
        b = abi.Bool(0, 0, x.data_ptr())
        b._keep = x
        return b
    return abi.Bool(1 if x else 0, 0, None)
