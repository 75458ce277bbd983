"""GPU parity: basics.zig ops and the voice mixdown vs the oracle (bit-exact)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

OPS = ["zero", "set", "copy", "add", "addInto", "addScalar", "addScalarInto",
       "multiply", "multiplyWith", "multiplyScalar", "multiplyWithScalar"]


def _oracle_op(po, name, s, e, dest, a, b, sc):
    L = po.lib()
    V = dest.shape[0]
    for v in range(V):
        d, av, bv = po.fptr(dest[v]), po.fptr(a[v]), po.fptr(b[v])
        x = float(sc[v])
        {"zero": lambda: L.zo_zero(s, e, d), "set": lambda: L.zo_set(s, e, d, x),
         "copy": lambda: L.zo_copy(s, e, d, av), "add": lambda: L.zo_add(s, e, d, av, bv),
         "addInto": lambda: L.zo_add_into(s, e, d, av), "addScalar": lambda: L.zo_add_scalar(s, e, d, av, x),
         "addScalarInto": lambda: L.zo_add_scalar_into(s, e, d, x), "multiply": lambda: L.zo_multiply(s, e, d, av, bv),
         "multiplyWith": lambda: L.zo_multiply_with(s, e, d, av), "multiplyScalar": lambda: L.zo_multiply_scalar(s, e, d, av, x),
         "multiplyWithScalar": lambda: L.zo_multiply_with_scalar(s, e, d, x)}[name]()


def _gpu_op(ctx, name, span, dest, a, b, sc):
    from zang_amd import zang
    fn = getattr(zang, name)
    if name == "zero": fn(span, dest, ctx=ctx)
    elif name in ("set", "addScalarInto", "multiplyWithScalar"): fn(span, dest, sc, ctx=ctx)
    elif name in ("copy", "addInto", "multiplyWith"): fn(span, dest, a, ctx=ctx)
    elif name in ("add", "multiply"): fn(span, dest, a, b, ctx=ctx)
    else: fn(span, dest, a, sc, ctx=ctx)


@pytest.mark.parametrize("name", OPS)
@pytest.mark.parametrize("V,per_voice,rows", [(256, True, False), (256, False, False), (67, True, False),   # 67: scalar-lane path
                                              (256, True, True), (1000, False, True)])   # rows: the many-voices form (basics_rows_min), forced
def test_basics_bitexact(ctx, oracle, name, V, per_voice, rows, monkeypatch):
    from zang_amd import zang
    if rows:
        util.set_form(monkeypatch, basics_rows_min="1")
    F, s, e = 300, 17, 283
    dest = util.rng_buffers(1, V, F); a = util.rng_buffers(2, V, F); b = util.rng_buffers(3, V, F)
    sc = np.random.default_rng(4).uniform(-2, 2, V).astype(np.float32)
    if not per_voice:
        sc[:] = sc[0]
    ref = dest.copy()
    _oracle_op(oracle, name, s, e, ref, a, b, sc)
    gd, ga, gb = util.to_image(dest), util.to_image(a), util.to_image(b)
    _gpu_op(ctx, name, zang.Span(s, e), gd, ga, gb, util.dev(sc) if per_voice else float(sc[0]))
    ctx.sync()
    assert ctx.last_form() == ["k_elementwise_chunks" if rows else "k_elementwise"]
    util.assert_bitexact(util.from_image(gd), ref, name)


@pytest.mark.parametrize("V", [32768, 32768 + 64])
def test_basics_many_voices_default_form(ctx, oracle, V):
    """From basics_rows_min voices (32,768) the operations take the row-chunk form by default (k_elementwise_chunks): every operation on
    a ragged span against the oracle on a stride of voices plus the edges, and the rows outside the span untouched."""
    import torch
    from zang_amd import zang
    F, s, e = 40, 3, 38
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    dest0 = torch.rand(F, V, device="cuda", generator=g) * 2 - 1
    ga = ctx.image(F, V); ga.copy_(torch.rand(F, V, device="cuda", generator=g) * 2 - 1)
    gb = ctx.image(F, V); gb.copy_(torch.rand(F, V, device="cuda", generator=g) * 2 - 1)
    sc = np.random.default_rng(4).uniform(-2, 2, V).astype(np.float32)
    idx = np.unique(np.concatenate([np.arange(0, V, 509), [0, 1, 255, 256, 257, V - 257, V - 2, V - 1]]))
    tidx = torch.from_numpy(idx).cuda()
    a_h = np.ascontiguousarray(ga[:, tidx].cpu().numpy().T); b_h = np.ascontiguousarray(gb[:, tidx].cpu().numpy().T)
    for name in OPS:
        gd = ctx.image(F, V); gd.copy_(dest0)
        _gpu_op(ctx, name, zang.Span(s, e), gd, ga, gb, util.dev(sc))
        ctx.sync()
        assert ctx.last_form() == ["k_elementwise_chunks"], (name, ctx.last_form())
        ref = np.ascontiguousarray(dest0[:, tidx].cpu().numpy().T)
        _oracle_op(oracle, name, s, e, ref, a_h, b_h, sc[idx])
        util.assert_bitexact(np.ascontiguousarray(gd[:, tidx].cpu().numpy().T), ref, name)
        assert torch.equal(gd[:s], dest0[:s]) and torch.equal(gd[e:], dest0[e:]), name


def test_basics_empty_span_and_errors(ctx):
    from zang_amd import zang, abi
    img = ctx.image(16, 64, fill=3.0)
    zang.zero(zang.Span(5, 5), img, ctx=ctx)
    ctx.sync()
    assert float(img.min()) == 3.0
    with pytest.raises(abi.ZangHipError):
        zang.zero(zang.Span(0, 17), img, ctx=ctx)      # span beyond the image


@pytest.mark.parametrize("V", [64, 1000, 4096, 5000])
def test_mixdown_voices(ctx, V):
    """The GPU sums in a fixed tree order; the reference adds voice after voice in f32.
    Gate against an f64 sum with a sqrt(V)*eps-scaled bound and check
    run-to-run determinism."""
    import torch
    from zang_amd import zang
    F = 256
    src = util.rng_buffers(9, V, F)
    img = util.to_image(src)
    mix = torch.full((F,), 0.5, dtype=torch.float32, device="cuda")
    zang.mixdownVoices(zang.Span(3, F), mix, img, ctx=ctx)
    mix2 = torch.full((F,), 0.5, dtype=torch.float32, device="cuda")
    zang.mixdownVoices(zang.Span(3, F), mix2, img, ctx=ctx)
    ctx.sync()
    got = mix.cpu().numpy()
    assert np.array_equal(got, mix2.cpu().numpy())
    ref = 0.5 + src.astype(np.float64).sum(axis=0)
    ref[:3] = 0.5
    bound = 4 * np.sqrt(V) * np.finfo(np.float32).eps * np.abs(src).astype(np.float64).sum(axis=0).max()
    assert np.abs(got - ref).max() <= bound
    assert np.array_equal(got[:3], np.full(3, 0.5, np.float32))


@pytest.mark.parametrize("V", [100, 16384])
def test_buf_alloc_stride_and_round_trip(ctx, V):
    """zh_buf_alloc pads rows that are a multiple of 64 KiB (stride > voices); uploads, kernels and downloads all
    address a sample as ptr[frame * stride + voice]."""
    import ctypes as C
    from zang_amd import abi
    lib = ctx.lib
    frames = 48
    b = abi.Buf()
    abi.check(lib.zh_buf_alloc(ctx.handle, C.byref(b), V, frames), "zh_buf_alloc")
    try:
        assert b.voices == V and b.frames == frames
        assert b.stride == (V + 1024 if (V * 4) % 65536 == 0 else V)
        rng = np.random.default_rng(V)
        host = rng.standard_normal((V, frames)).astype(np.float32)
        abi.check(lib.zh_buf_upload_voices(ctx.handle, b, host.ctypes.data, frames), "upload")
        abi.check(lib.zh_multiply_with_scalar(ctx.handle, 5, 40, b, abi.F32(2.0, 0, None)), "multiplyWithScalar")
        back = np.zeros_like(host)
        abi.check(lib.zh_buf_download_voices(ctx.handle, back.ctypes.data, b, frames), "download")
        want = host.copy(); want[:, 5:40] *= np.float32(2.0)
        assert np.array_equal(back, want)
        one = np.zeros(frames, np.float32)
        abi.check(lib.zh_buf_download_voice(ctx.handle, one.ctypes.data, b, V - 1, frames), "download_voice")
        assert np.array_equal(one, want[V - 1])
    finally:
        lib.zh_buf_free(ctx.handle, C.byref(b))
