// lanes.hip.h -- lane width for the sequential per-voice kernels: one voice per lane (W = 1) or two
// adjacent voices per lane (W = 2).
//
// Why W = 2 exists: with two voices of one lane held in adjacent VGPRs, every f32 add/sub/mul of the
// pair becomes one v_pk_*_f32 (two IEEE results, same bits as the scalar ops; contraction stays off).
// Why it is not the default: tools/ubench/valu_ops.hip measures, per SIMD at full occupancy, 2.5 cycles
// for a plain 2-operand v_add/v_mul/v_sub_f32 or integer/logic op, but 4.2 cycles for v_pk_*_f32, v_fma,
// v_min/v_max, every v_cmp and every v_cndmask -- so a packed op is barely cheaper than the two plain
// ops it replaces, while compares and selects (no packed form) double.  A NiceInstrument voice pair
// measured 10-15 % SLOWER than two single-voice lanes (composite.hip).  Kept, parity-tested, because
// the generic form costs nothing at W = 1 and documents the experiment.
//
// The per-sample code (dsp.hip.h, envelope.hip.h) is written once, over LaneT<W>::F/U/M, with the
// helpers below for the few things that differ between `bool` and a vector mask.
#pragma once
#include "common.hip.h"

typedef float zf2 __attribute__((ext_vector_type(2)));
typedef uint32_t zu2 __attribute__((ext_vector_type(2)));
typedef int32_t zm2 __attribute__((ext_vector_type(2)));     // comparison result: 0 / -1 per component

template <int W> struct LaneT;
template <> struct LaneT<1> { using F = float; using U = uint32_t; using M = bool; };
template <> struct LaneT<2> { using F = zf2; using U = zu2; using M = zm2; };

#define ZL __device__ __forceinline__

// select
ZL float zsel(bool m, float a, float b) { return m ? a : b; }
ZL uint32_t zsel(bool m, uint32_t a, uint32_t b) { return m ? a : b; }
ZL zf2 zsel(zm2 m, zf2 a, zf2 b) { return m ? a : b; }
ZL zu2 zsel(zm2 m, zu2 a, zu2 b) { return m ? a : b; }
// m ? a : b as a v_cndmask, whatever the optimizer thinks: given a select between two computed values LLVM often
// rebuilds a branch (it sinks an expensive operand into one arm) -- an exec-mask region of ~6 scalar instructions per
// frame where a compare and a select would do.  The empty asm makes both operands opaque values with nothing to sink;
// the select itself stays the compiler's.  (Until round 3 this was the v_cndmask itself as inline asm with
// `"s"(ballot(m))` as its mask operand.  Where m is true at compile time -- a literal TriSawOsc color in a script --
// ballot(true) IS the exec mask, and the compiler may hand the asm the EXEC register itself for an "s" operand; a
// v_cndmask_b32_e64 whose mask operand is EXEC takes the mask bits of lanes 32-63 as zero on gfx950
// (tools/ubench/select_hazard.hip: 32,768 of 32,768 wrong in the upper half-wave, none with a copy of EXEC in an SGPR
// pair; the compiler's own selects use a register class without EXEC).  Found by the random-script fuzz,
// tests/script_fuzz.py seed 1015.  No asm in csrc/ takes a scalar operand any more: test_no_inline_asm_reads_an_sgpr.)
#if defined(__HIP_DEVICE_COMPILE__)
ZL float zsel_hard(bool m, float a, float b) {
    asm("" : "+v"(a), "+v"(b));
    return m ? a : b;
}
#else
ZL float zsel_hard(bool m, float a, float b) { return m ? a : b; }
#endif
// mask logic (masks are `bool` or all-ones/zero vectors)
ZL bool zand(bool a, bool b) { return a && b; }
ZL zm2 zand(zm2 a, zm2 b) { return a & b; }
template <class M> ZL M zand(M a, M b, M c) { return zand(zand(a, b), c); }
ZL bool zor(bool a, bool b) { return a || b; }
ZL zm2 zor(zm2 a, zm2 b) { return a | b; }
ZL bool znot(bool m) { return !m; }
ZL zm2 znot(zm2 m) { return ~m; }
ZL bool zany(bool m) { return m; }
ZL bool zany(zm2 m) { return (m.x | m.y) != 0; }
// any lane of the WAVE (wave-uniform, for a scalar branch)
ZL bool zany_wave(bool m) { return __builtin_amdgcn_ballot_w64(m) != 0; }
ZL bool zany_wave(zm2 m) { return __builtin_amdgcn_ballot_w64((m.x | m.y) != 0) != 0; }
ZL bool zall(bool m) { return m; }
ZL bool zall(zm2 m) { return (m.x & m.y) != 0; }
// splats
template <class T> ZL T zsplat(float v);
template <> ZL float zsplat<float>(float v) { return v; }
template <> ZL zf2 zsplat<zf2>(float v) { return zf2{v, v}; }
template <class T> ZL T zsplatu(uint32_t v);
template <> ZL uint32_t zsplatu<uint32_t>(uint32_t v) { return v; }
template <> ZL zu2 zsplatu<zu2>(uint32_t v) { return zu2{v, v}; }
template <class M> ZL M zmask(bool b);
template <> ZL bool zmask<bool>(bool b) { return b; }
template <> ZL zm2 zmask<zm2>(bool b) { const int32_t x = b ? -1 : 0; return zm2{x, x}; }
ZL zm2 zmask2(bool b0, bool b1) { return zm2{b0 ? -1 : 0, b1 ? -1 : 0}; }
// bit casts
ZL uint32_t zbits_u(float x) { return __builtin_bit_cast(uint32_t, x); }
ZL zu2 zbits_u(zf2 x) { return __builtin_bit_cast(zu2, x); }
ZL float zbits_f(uint32_t u) { return __builtin_bit_cast(float, u); }
ZL zf2 zbits_f(zu2 u) { return __builtin_bit_cast(zf2, u); }

// component access (setup code that runs once per paint call is done per component in scalar form)
ZL float zget(float v, int) { return v; }
ZL float zget(zf2 v, int i) { return i ? v.y : v.x; }
ZL uint32_t zget(uint32_t v, int) { return v; }
ZL uint32_t zget(zu2 v, int i) { return i ? v.y : v.x; }
ZL bool zget(bool v, int) { return v; }
ZL bool zget(zm2 v, int i) { return (i ? v.y : v.x) != 0; }
ZL void zput(float &d, int, float v) { d = v; }
ZL void zput(zf2 &d, int i, float v) { if (i) d.y = v; else d.x = v; }
ZL void zput(uint32_t &d, int, uint32_t v) { d = v; }
ZL void zput(zu2 &d, int i, uint32_t v) { if (i) d.y = v; else d.x = v; }
ZL void zput(bool &d, int, bool v) { d = v; }
ZL void zput(zm2 &d, int i, bool v) { if (i) d.y = v ? -1 : 0; else d.x = v ? -1 : 0; }

// loads / stores of W adjacent voices starting at voice `v` (v even and p 8-byte aligned for W = 2)
template <int W> ZL typename LaneT<W>::F zload_f(const float *p, uint32_t v) {
    if constexpr (W == 1) return p[v];
    else return *reinterpret_cast<const zf2 *>(p + v);
}
template <int W> ZL typename LaneT<W>::U zload_u(const uint32_t *p, uint32_t v) {
    if constexpr (W == 1) return p[v];
    else return *reinterpret_cast<const zu2 *>(p + v);
}
template <int W> ZL void zstore_f(float *p, uint32_t v, typename LaneT<W>::F x) {
    if constexpr (W == 1) p[v] = x;
    else *reinterpret_cast<zf2 *>(p + v) = x;
}
template <int W> ZL void zstore_u(uint32_t *p, uint32_t v, typename LaneT<W>::U x) {
    if constexpr (W == 1) p[v] = x;
    else *reinterpret_cast<zu2 *>(p + v) = x;
}
// ---- image rows through buffer descriptors ---------------------------------------------------------
// A [frame][voice] image is walked row by row with a wave-uniform row pointer.  Through a buffer
// descriptor the whole per-frame address is scalar: the descriptor's base is the first row of the
// current chunk (rebased with two scalar adds per chunk), the row inside the chunk is `soffset` (an
// SGPR) and the lane's voice is one constant byte offset in a VGPR -- a load or store costs no vector
// address arithmetic.  Offsets are 32-bit: a chunk of CH = 8 rows must span < 4 GiB, i.e. stride < 2^27 voices;
// the entry points accept strides up to 2^26 (common.hip.h kMaxRowStride, checked in buf_covers).
#if defined(__HIP_DEVICE_COMPILE__)
ZL zh_rsrc_t zrow_rsrc(const float *base, size_t stride, uint32_t frame) { return make_rsrc(base + (size_t)frame * stride, 0xFFFFFFFFu); }
template <int W> ZL typename LaneT<W>::F zrow_load(zh_rsrc_t r, uint32_t voff, uint32_t soff) {
    if constexpr (W == 1) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    else return __builtin_bit_cast(zf2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// Image rows are written once per paint and not read again by the kernel that writes them: non-temporal stores (aux bit 1 = nt on
// gfx940+).  Against plain stores, alternating on one box (profiles/r05/ab_row_store_nt.txt): the module table at 4,096 voices 2-15 %
// faster on every row that stores through here (Envelope 11.3 -> 9.7 us, Sampler 12.6 -> 10.9, PulseOsc with a frequency image 15.9 ->
// 13.5), 1-4 % at 131,072 voices; config 3 unfused (Noise -> temp -> Filter) 59.9 -> 53.9 us, fused 52.6 -> 51.0.
constexpr int kRowStoreAux = 2;
template <int W> ZL void zrow_store(zh_rsrc_t r, uint32_t voff, uint32_t soff, typename LaneT<W>::F x) {
    if constexpr (W == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), r, voff, soff, kRowStoreAux);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(zu2, x), r, voff, soff, kRowStoreAux);
}
#else   // host pass: kernels are only parsed, never run
ZL zh_rsrc_t zrow_rsrc(const float *, size_t, uint32_t) { return 0; }
template <int W> ZL typename LaneT<W>::F zrow_load(zh_rsrc_t, uint32_t, uint32_t) { return typename LaneT<W>::F{}; }
template <int W> ZL void zrow_store(zh_rsrc_t, uint32_t, uint32_t, typename LaneT<W>::F) {}
#endif

template <int W> ZL typename LaneT<W>::F zget_f32p(const F32P &p, uint32_t v) {
    if constexpr (W == 1) return p.get(v);
    else return p.pv ? *reinterpret_cast<const zf2 *>(p.pv + v) : zf2{p.value, p.value};
}
template <int W> ZL typename LaneT<W>::M zget_boolp(const BoolP &p, uint32_t v) {
    if constexpr (W == 1) return p.get(v);
    else return zmask2(p.get(v), p.get(v + 1));
}
