// filter_tp.hip.h -- the TOLERANT, time-parallel forms of the Filter (src/modules/Filter.zig:74-151) for few voices
// (ZH_PAINT_TOLERANT).  Opt-in: every default form in this library is bit-exact, and so are these for the
// first chunk of a span; north_star asks 1e-5 relative for the Filter, bits only for Gate and Decimator.
//
// With constant cutoff and resonance the 2x-oversampled state-variable step (:135-144) is ONE affine map of the state:
//     (l, b)' = A (l, b) + g(in)           A = A(cut, res), 2 x 2, the same for every frame of the span
// so the state after L frames is A^L (l, b) + e, where e is what the same L inputs leave behind from a ZERO state.  A span is
// cut into C chunks of L frames that run at once, one wave per (64 voices, chunk):
//   pass 1   every chunk runs the reference's f32 recurrence from a zero state over its own inputs -> e_j          (L steps)
//   scan     s_0 = the module's state; s_j = A^L s_{j-1} + e_{j-1}: A from the step run on unit states, A^L by squaring, the
//            products and sums in f64 (C - 1 steps of a 2 x 2 product)
//   pass 2   every chunk runs THE REFERENCE'S OWN recurrence again, from round_f32(s_j), and paints          (L steps)
// The chain a voice waits for is 2 L + C steps instead of C L.  Inside a chunk the arithmetic is the reference's, operation for
// operation; what differs is the chunk's start state: the reference reached it through L more roundings of its own, the scan
// through the roundings of pass 1 -- two f32 evaluations of the same real number, each within a few ulp * sqrt(L) of it.
// Measured against the oracle (tools/exp/filter_tp_error.py on the CPU, tests/test_gpu_tolerant.py on the device): the
// largest error of any sample is <= 2.6e-6 of the voice's peak over the span for every case tried (config 3's parameter range, res 0.9
// and 1.0, cutoff 0 / 1e-4 / 1.0, inputs scaled by 1e-30 .. 1e30) -- inside 1e-5 relative to the signal.  It is NOT inside
// tests/util.py's per-sample metric |err| <= 1e-5 max(|ref|, 1e-3) near zero crossings (1-3 % of the samples): there the metric's
// tolerance is 1e-8 against a signal of order 1, below one ulp of the state the sample was computed from, which no
// re-association of the recurrence can meet.  profiles/r04/NOTES.md 5a states both figures.
//
// Both users run it as TWO kernels over a grid of (256 voices, chunk) workgroups, e_j through a module-owned scratch in HBM:
// k_filter_tp_a / _b   Filter module, input image (read by both passes, the second time from L2).
// k_nf_tp_a / _b       the fused white Noise -> Filter voice (config 3): the noise of chunk j needs the generator's state j L draws
//                      ahead = one application of the jump table T^(jL) (noise_jump.hip.h, 32 KB of LDS; one table per
//                      workgroup, hence the grid's shape); the chunk-start generator states go through the scratch too and pass
//                      B generates the chunk's noise again (30 instructions a sample against 8 bytes of traffic).  The noise
//                      itself is exact.
// A lone wave issues an instruction every ~5 cycles, two waves on a SIMD interleave (2.7 cycles per plain instruction), more add
// nothing (tools/ubench/valu_ops.hip): the launches aim at ~2,048 waves, and from there what a launch costs is its total
// instruction count, so the per-chunk extras (a jump: ~700 instructions; the scan) set the chunk length -- not "as many chunks
// as possible".
#pragma once
#include "common.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "lanes.hip.h"
#include "noise_jump.hip.h"

struct Svf2x2 { double a00, a01, a10, a11; };
// A(cut, res): the step of Filter.zig:135-144 without its inputs (in = 0, no dc offset), on the unit states, in f64
__device__ __forceinline__ Svf2x2 svf_hom(float cutf, float resf) {
    const double c = cutf, r = resf;
    auto hs = [&](double l, double b, double &lo, double &bo) ZH_INLINE_LAMBDA {
        l = l + c * b;                                                // :138
        b = b + c * (-(b * r) - l);                                   // :139
        l = l + c * b;                                                // :142
        const double h = -(b * r) - l;                                // :143
        b = b + c * h;                                                // :144
        lo = l; bo = b;
    };
    Svf2x2 a;
    hs(1.0, 0.0, a.a00, a.a10);
    hs(0.0, 1.0, a.a01, a.a11);
    return a;
}
__device__ __forceinline__ Svf2x2 svf_mul(const Svf2x2 &x, const Svf2x2 &y) {
    return Svf2x2{x.a00 * y.a00 + x.a01 * y.a10, x.a00 * y.a01 + x.a01 * y.a11, x.a10 * y.a00 + x.a11 * y.a10, x.a10 * y.a01 + x.a11 * y.a11};
}
// a^n, n >= 1 wave-uniform
__device__ __forceinline__ Svf2x2 svf_pow(Svf2x2 a, uint32_t n) {
    Svf2x2 r{1.0, 0.0, 0.0, 1.0};
    for (;;) {
        if (n & 1u) r = svf_mul(a, r);
        n >>= 1;
        if (!n) break;
        a = svf_mul(a, a);
    }
    return r;
}
// s_j from s_0 and the zero-state end states e_0 .. e_{j-1} of the (equally long) chunks before it.  fetch(i) -> e_i for ANY
// i < NMAX (slots past j hold something readable and are ignored): all NMAX values are requested before the first is used --
// fetched one by one inside the j-step loop, every global load is a round trip of its own.  A^L is formed in f64 (squaring
// multiplies its error) and rounded once; the j steps are plain f32 products and sums -- as accurate as f64 steps against the
// reference (which is f32 itself; tools/exp/filter_tp_error.py) at a quarter of their issue time.
template <uint32_t NMAX, class Fetch>
__device__ __forceinline__ void svf_scan(float &l, float &b, float cut, float res, uint32_t L, uint32_t j, Fetch fetch) {
    if (j == 0) return;                                               // the first chunk starts from the module's state itself: exact
    float2 e[NMAX];
#pragma unroll
    for (uint32_t i = 0; i < NMAX; i++) e[i] = fetch(i);
    const Svf2x2 md = svf_pow(svf_hom(cut, res), L);
    const float m00 = (float)md.a00, m01 = (float)md.a01, m10 = (float)md.a10, m11 = (float)md.a11;
#pragma unroll
    for (uint32_t i = 0; i < NMAX; i++)
        if (i < j) {                                                  // wave-uniform
            const float nl = (m00 * l + m01 * b) + e[i].x;
            const float nb = (m10 * l + m11 * b) + e[i].y;
            l = nl; b = nb;
        }
}

constexpr uint32_t kTpMaxChunks = 32;    // chunks per launch; a module's scratch holds this many slots whatever a launch uses
// A voice whose clamped cutoff is below this is NOT painted as chunks: its chunk-0 lane walks the span frame by frame from the
// module's state -- the reference's own recurrence, bit for bit.  Near a zero cutoff the output is the filter's dc-offset ramp,
// the reference's f32 accumulation of that nearly constant increment drifts systematically from exact arithmetic, and no chunked
// evaluation follows it (round 5: 1.9e-5 of the peak at cutoff 6e-6; tools/exp/filter_tp_error.py over log-uniform cutoffs: up to
// 1e-4 below 1e-4, under 1e-5 from 1e-3 on).  2^-9 = cutoffFromFrequency(15 Hz at 48 kHz): below the audio band.
constexpr float kTpExactCutBelow = 0.001953125f;

#if !defined(ZH_DEVICE_ONLY)
// chunks for a span of n frames of V voices: enough for ~2,048 waves (two per SIMD), 2..32; 0 = too many voices for the form
static inline uint32_t zh_tp_chunks(uint32_t V, int max_form, uint32_t n) {
    const uint32_t G = (V + 63) / 64;
    const uint32_t tp_max = (uint32_t)zh_form(max_form);              // largest voice count that takes the time-parallel form (dispatch.hip)
    if (V > tp_max || G > 1024) return 0;
    const uint32_t waves = 2048u;                                     // waves a launch should reach
    uint32_t C = (waves + G - 1) / G;
    C = C < 2 ? 2 : (C > kTpMaxChunks ? kTpMaxChunks : C);
    return C > n ? n : C;
}
#endif

// ---------------------------------------------------------------------------------------------------- Filter module
// Two kernels, grid: x = 256-voice groups, y = chunk; block = 256 -- a chunk's waves go wherever there is room (a workgroup of
// all the chunks of 64 voices, exchanging e_j through LDS, kept 64 CUs busy and 192 idle at 4,096 voices: 17 us).  e_j and the
// span's start state go through the module's scratch `e` ([chunks + 1][V]: slot 0 = start state, slot j + 1 = e_j); pass B
// reads its inputs again (from L2).
struct FilterTpArgs {
    float *l, *b;
    float2 *e;
    float4 *m;                   // control-image cutoff / resonance: chunk j's transition matrix (slot j), [chunks][V]
    uint32_t *flag;              // [V]: == serial when a frame of the cutoff IMAGE was below kTpExactCutBelow in this piece (pass A -> pass B)
    uint32_t serial;             // this piece's number (never 0; flags are never cleared: an old number is "not flagged")
    uint32_t V, start, end, L;
    Img out;
    CImg input;
    float l_mul, b_mul, h_mul;
    CobP cutoff, res;
};
// body(f, in, cutoff_i, res_i) for the frames [f0, f1) of one lane, in = input + fcdcoffset (:135): 8-row tiles, the tile's loads
// ahead of its use.  CB / RB: the cutoff / resonance come from control images (clamped per frame, Filter.zig:126-128).
template <bool CB, bool RB, class Body>
__device__ __forceinline__ void tp_input_tiles(const FilterTpArgs &a, uint32_t voff, uint32_t f0, uint32_t f1, float cut_c, float res_c, Body body) {
    const uint32_t irow = (uint32_t)a.input.stride * 4u, crow = (uint32_t)a.cutoff.b.stride * 4u, rrow = (uint32_t)a.res.b.stride * 4u;
    auto clampcut = [](float x) ZH_INLINE_LAMBDA { return zclampf(x, 0.0f, 1.0f); };
    auto clampres = [](float x) ZH_INLINE_LAMBDA { return 1.0f - zclampf(x, 0.0f, 1.0f); };
    uint32_t f = f0;
    for (; f + 8 <= f1; f += 8) {
        const zh_rsrc_t ri = zrow_rsrc(a.input.p, a.input.stride, f);
        float x[8], c[8], r[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            x[q] = zrow_load<1>(ri, voff, q * irow) + kSvfDcOffset;
            if (CB) c[q] = zrow_load<1>(zrow_rsrc(a.cutoff.b.p, a.cutoff.b.stride, f), voff, q * crow);
            if (RB) r[q] = zrow_load<1>(zrow_rsrc(a.res.b.p, a.res.b.stride, f), voff, q * rrow);
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) body(f + q, x[q], CB ? clampcut(c[q]) : cut_c, RB ? clampres(r[q]) : res_c);
    }
    for (; f < f1; f++) {
        const float x = zrow_load<1>(zrow_rsrc(a.input.p, a.input.stride, f), voff, 0) + kSvfDcOffset;
        const float c = CB ? clampcut(zrow_load<1>(zrow_rsrc(a.cutoff.b.p, a.cutoff.b.stride, f), voff, 0)) : cut_c;
        const float r = RB ? clampres(zrow_load<1>(zrow_rsrc(a.res.b.p, a.res.b.stride, f), voff, 0)) : res_c;
        body(f, x, c, r);
    }
}
// the step of Filter.zig:135-144 without its inputs (in = 0, no dc offset): the homogeneous part, in f32 like the step itself
__device__ __forceinline__ void svf_hom_step(float &l, float &b, float cut, float res) {
    l = l + cut * b;                                                  // :138
    b = b + cut * (-(b * res) - l);                                   // :139
    l = l + cut * b;                                                  // :142
    const float h = -(b * res) - l;                                   // :143
    b = b + cut * h;                                                  // :144
}
// Pass A.  With control images the step's matrix changes from frame to frame, so a chunk's transition is the PRODUCT of its
// frames' matrices: the chunk also carries the two unit states through the homogeneous step (two more recurrences) and leaves
// the four numbers they end on -- the columns of that product -- beside e_j.
template <bool CB, bool RB>
__global__ void __launch_bounds__(256) k_filter_tp_a(const FilterTpArgs a) {
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.V) return;
    if (j == 0) a.e[v] = make_float2(a.l[v], a.b[v]);
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end);
    const float cut = CB ? 0.0f : zclampf(a.cutoff.c.get(v), 0.0f, 1.0f);          // Filter.zig:114
    const float res = RB ? 0.0f : 1.0f - zclampf(a.res.c.get(v), 0.0f, 1.0f);      // :118
    float l = 0.0f, b = 0.0f;
    float ul = 1.0f, ub = 0.0f, wl = 0.0f, wb = 1.0f;                               // the unit states (CB || RB)
    float cmin = 1.0f;
    tp_input_tiles<CB, RB>(a, v * 4u, f0, f1, cut, res, [&](uint32_t, float in, float c, float r) ZH_INLINE_LAMBDA {
        svf_core(l, b, in, c, r);
        if (CB) cmin = c < cmin ? c : cmin;
        if (CB || RB) { svf_hom_step(ul, ub, c, r); svf_hom_step(wl, wb, c, r); }
    });
    if (CB && cmin < kTpExactCutBelow) a.flag[v] = a.serial;                         // (pass B walks this voice: kTpExactCutBelow)
    a.e[(size_t)(j + 1) * a.V + v] = make_float2(l, b);
    if (CB || RB) a.m[(size_t)j * a.V + v] = make_float4(ul, wl, ub, wb);            // (m00, m01, m10, m11)
}
template <bool ZF, bool CB, bool RB>
__global__ void __launch_bounds__(256) k_filter_tp_b(const FilterTpArgs a) {
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.V) return;
    const size_t V = a.V;
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end);
    const float cut = CB ? 0.0f : zclampf(a.cutoff.c.get(v), 0.0f, 1.0f);
    const float res = RB ? 0.0f : 1.0f - zclampf(a.res.c.get(v), 0.0f, 1.0f);
    const float2 s0 = a.e[v];
    float l = s0.x, b = s0.y;
    if constexpr (CB || RB) {
        if (j > 0) {                                                  // s_j = M_{j-1} s_{j-1} + e_{j-1}, every operand requested first
            float2 e[kTpMaxChunks - 1];
            float4 m[kTpMaxChunks - 1];
#pragma unroll
            for (uint32_t i = 0; i < kTpMaxChunks - 1; i++) { e[i] = a.e[(size_t)(i + 1) * V + v]; m[i] = a.m[(size_t)i * V + v]; }
#pragma unroll
            for (uint32_t i = 0; i < kTpMaxChunks - 1; i++)
                if (i < j) {                                          // wave-uniform
                    const float nl = (m[i].x * l + m[i].y * b) + e[i].x;
                    const float nb = (m[i].z * l + m[i].w * b) + e[i].y;
                    l = nl; b = nb;
                }
        }
    } else {
        svf_scan<kTpMaxChunks - 1>(l, b, cut, res, a.L, j, [&](uint32_t i) ZH_INLINE_LAMBDA { return a.e[(size_t)(i + 1) * V + v]; });
    }
    const uint32_t voff = v * 4u;
    // a cutoff near zero (constant: this voice's; image: a frame of this piece): not as chunks, see kTpExactCutBelow
    const bool exact_v = CB ? a.flag[v] == a.serial : cut < kTpExactCutBelow;
    auto frame = [&](uint32_t f, float in, float c, float r, bool store) ZH_INLINE_LAMBDA {
        const SvfOut sv = svf_core(l, b, in, c, r);                   // :138-144
        const float val = sv.l * a.l_mul + sv.b * a.b_mul + sv.h * a.h_mul;   // :146
        if (store) {
            const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f);
            const float base = ZF ? 0.0f : zrow_load<1>(ro, voff, 0);
            zrow_store<1>(ro, voff, 0, base + val);
        }
    };
    tp_input_tiles<CB, RB>(a, voff, f0, f1, cut, res, [&](uint32_t f, float in, float c, float r) ZH_INLINE_LAMBDA { frame(f, in, c, r, !exact_v); });
    if (!exact_v && f1 == a.end && f1 > f0) { a.l[v] = l; a.b[v] = b; }
    if (j == 0 && __builtin_amdgcn_ballot_w64(exact_v) != 0) {
        if (exact_v) {                                                // the reference's own walk over the piece, from the module's state
            l = s0.x; b = s0.y;
            tp_input_tiles<CB, RB>(a, voff, a.start, a.end, cut, res, [&](uint32_t f, float in, float c, float r) ZH_INLINE_LAMBDA { frame(f, in, c, r, true); });
            a.l[v] = l; a.b[v] = b;
        }
    }
}

#if !defined(ZH_DEVICE_ONLY)
// floats of scratch per voice: (kTpMaxChunks + 1) float2 of e, then kTpMaxChunks float4 of transition matrices
constexpr size_t kFilterTpFloats = (size_t)(kTpMaxChunks + 1) * 2 + (size_t)kTpMaxChunks * 4 + 1;   // ... and one word of flag (zeroed by whoever allocates)
constexpr size_t kFilterTpFlagAt = (size_t)(kTpMaxChunks + 1) * 2 + (size_t)kTpMaxChunks * 4;         // (floats per voice before the flags)
// Launches the span as pieces of <= 32 chunks.  false = not taken (too many voices, a short span): the caller paints with its
// exact form.  `scratch` = the module's, kFilterTpFloats * V floats.
static inline bool zh_filter_tp_launch(hipStream_t st, float *l, float *b, float *scratch, uint32_t &serial, uint32_t V, Img out, CImg in, uint32_t start, uint32_t end, bool zf,
                                       float l_mul, float b_mul, float h_mul, CobP cut, CobP res) {
    if (end - start < 64) return false;
    const uint32_t C = zh_tp_chunks(V, ZF_FILTER_TP_MAX, end - start);
    if (C < 2) return false;
    const uint32_t piece = 4096;                                      // frames per launch pair: chunks of <= 128 frames
    FilterTpArgs a;
    a.l = l; a.b = b; a.e = reinterpret_cast<float2 *>(scratch); a.m = reinterpret_cast<float4 *>(scratch + (size_t)(kTpMaxChunks + 1) * 2 * V);
    a.flag = reinterpret_cast<uint32_t *>(scratch + kFilterTpFlagAt * V);
    a.V = V; a.out = out; a.input = in; a.l_mul = l_mul; a.b_mul = b_mul; a.h_mul = h_mul; a.cutoff = cut; a.res = res;
    const bool cb = cut.is_buffer != 0, rb = res.is_buffer != 0;
    for (uint32_t s = start; s < end; s += piece) {
        a.start = s; a.end = min(s + piece, end);
        if (++serial == 0) serial = 1;
        a.serial = serial;
        a.L = (a.end - a.start + C - 1) / C;
        const dim3 grid((V + 255) / 256, (a.end - a.start + a.L - 1) / a.L);
#define ZH_FTP(CB_, RB_)                                                                                   \
        do {                                                                                               \
            ZH_LAUNCH((k_filter_tp_a<CB_, RB_>), grid, dim3(256), 0, st, a);                      \
            if (zf) ZH_LAUNCH((k_filter_tp_b<true, CB_, RB_>), grid, dim3(256), 0, st, a);        \
            else ZH_LAUNCH((k_filter_tp_b<false, CB_, RB_>), grid, dim3(256), 0, st, a);          \
        } while (0)
        if (cb && rb) ZH_FTP(true, true);
        else if (cb) ZH_FTP(true, false);
        else if (rb) ZH_FTP(false, true);
        else ZH_FTP(false, false);
#undef ZH_FTP
    }
    return true;
}
#endif

// Eight consecutive white samples as Noise.paint adds them to a zeroed temp (Noise.zig:51).  The common case of Random.float
// (the draw's high word is non-zero) has no test per sample; the tile's minimum high word is tested once, and when it is zero
// (2^-32 per sample) the tile is drawn again with the careful form, which also reports a second draw (2^-41) in `multi`.
__device__ __forceinline__ void noise_tile8(ZXoshiro &r, float (&t)[8], bool &multi) {
    const ZXoshiro r0 = r;
    uint32_t hmin = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = 0.0f + (zrandom_float32_common(r, hmin) * 2.0f - 1.0f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hmin == 0u) != 0, 0)) {
        r = r0;
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = 0.0f + (zrandom_float32_multi(r, multi) * 2.0f - 1.0f);
    }
}
// block -> (chunk j, 256-voice group g), both passes: a one-dimensional grid of ceil(C / 8) * 8 * per blocks in which chunk j of
// every group lands on XCD j % 8 (blocks are dealt round-robin over the eight XCDs): the 32 KiB jump table of a chunk is then read
// into ONE XCD's L2 instead of all eight (8.4 MB of the 36 MB a buffer moved in round 4, pmc_traffic_noise_filter_fused4096_tolerant
// .json), and pass B finds the chunk's generator state, which pass A left in that same L2.  (Pink Noise below: the same.)
template <class A> __device__ __forceinline__ bool nf_tp_block(const A &a, uint32_t &j, uint32_t &g, uint32_t id = blockIdx.x) {
    const uint32_t band = id / (8u * a.per), rr = id % (8u * a.per);
    j = band * 8u + (rr & 7u); g = rr >> 3;
    return j < a.C;
}
// ... and pass B the other way round: every chunk of voice group g on XCD g % 8 (grid of 8 * C * ceil(per / 8) blocks).  Pass B reads
// the e_i of EVERY earlier chunk of its voices: with the groups' chunks on one XCD each 2 KiB block of e is fetched into one L2 once
// and hit by the other chunks (measured with pass A's mapping in pass B too: 13 MB of reads per buffer instead of 5,
// profiles/r05/pmc_traffic_noise_filter_fused4096_tolerant.json history in NOTES.md).
template <class A> __device__ __forceinline__ bool nf_tp_block_b(const A &a, uint32_t &j, uint32_t &g, uint32_t id = blockIdx.x) {
    const uint32_t k = id >> 3;
    j = k % a.C; g = (k / a.C) * 8u + (id & 7u);
    return g < a.per;
}

// ---------------------------------------------------------------------------------------------------- white Noise -> Filter
#if defined(ZH_FILTER_TP_NOISE)          // (composite.hip only: the kernels below are not templates)
constexpr uint32_t kNfTpMaxChunks = kTpMaxChunks;
constexpr uint32_t kNfTpFusedSub = 2;   // chunks per pass-A lane in k_nf_tp_ba
struct NfTpArgs {
    uint64_t *s[4];              // the voices' generator states (Noise.zig:9): read by pass A, written once by pass B
    // Pipelined recording (composite.hip zh_noise_filter_paint, ZH_CAPTURE_COALESCE): pass A of paint n + 1 runs BESIDE pass B of paint n (k_nf_tp_ba).
    // It then starts from the generator state pass A of paint n predicted (`s_in`; equal to what pass B writes unless a multi-draw
    // sample was seen) and leaves its own prediction in `pred`; pass B leaves the filter state for the next paint in the OTHER scratch
    // set's slot 0 (`e_next0`: pass A of the next paint must not wait for it) and, when it had to walk a voice sequentially, flags that
    // voice for the next paint as well (`flag_next`, `serial_next`): that paint's pass A has started from a wrong prediction.
    const uint64_t *s_in[4];     // pass A's start states (== s outside a pipeline)
    uint64_t *pred[4];           // pass A: the generator state after the span's last draw, or null
    float2 *e_next0;             // pass B: the next paint's slot 0, or null
    uint32_t *flag_next;
    uint32_t serial_next;
    uint32_t snapshot;           // pass A: copy (l, b) into slot 0 (the first paint of a pipeline, and every paint outside one)
    float *l, *b;                // filter state
    uint64_t *cs;                // scratch [C][4][V]: generator state at the start of chunk j
    float2 *e;                   // scratch [C + 1][V]: slot 0 = the filter state at span start, slot j + 1 = e_j
    uint32_t *flag;              // scratch [V]: == serial when a multi-draw sample (Random.float, 2^-41 per sample) was seen in THIS paint
    uint32_t serial;             // this paint's number (host counter, never 0; flags are never cleared: an old number is "not flagged",
                                 // and a graph replayed with the number it was recorded with at worst sees a stale flag and walks that
                                 // voice sequentially -- the exact path -- once more)
    const uint4 *tables;         // T^(32 k), k = 1..63
    uint32_t V, start, end, L, C, per;   // per = 256-voice groups
    struct Slots { uint32_t C, per; };   // (what nf_tp_block maps a block id over: pass A of the shared launch has C / SUB slots)
    Img out;
    float l_mul, b_mul, h_mul;
    F32P cutoff, res;
};

// block = 256.  Pass A: jump to the chunk's first draw, keep that state, zero-state response.
// SUB chunks per lane, one after the other (the generator simply runs on; the zero-state response restarts): in the launch shared with pass B
// (k_nf_tp_ba) the chip is full either way, and two chunks per lane halve the table loads and jumps -- ~950 instructions per lane
// against 1,440 for a chunk's 32 frames.
template <uint32_t SUB>
__device__ __forceinline__ void nf_tp_a_run(const NfTpArgs &a, uint32_t bid, uint4 *tbl) {
    uint32_t jp, g;
    {
        NfTpArgs::Slots sl{(a.C + SUB - 1) / SUB, a.per};
        if (!nf_tp_block(sl, jp, g, bid)) return;                     // (block-uniform)
    }
    uint32_t j = jp * SUB;
    if (j > 0) {
        const uint4 *t = a.tables + (size_t)(j * (a.L / 32) - 1) * kNoiseJumpEntries;
        uint4 w[kNoiseJumpEntries / 256];                             // the thread's eight entries requested together, then parked
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) w[q] = t[q * 256 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) tbl[q * 256 + threadIdx.x] = w[q];
        __syncthreads();
    }
    const uint32_t v = g * 256 + threadIdx.x;
    if (v >= a.V) return;
    ZXoshiro r{a.s_in[0][v], a.s_in[1][v], a.s_in[2][v], a.s_in[3][v]};
    if (j > 0) noise_jump_apply(r, tbl);
    const size_t V = a.V;
    if (j == 0 && a.snapshot) a.e[v] = make_float2(a.l[v], a.b[v]);
    const float cut = zclampf(a.cutoff.get(v), 0.0f, 1.0f);           // Filter.zig:114
    const float res = 1.0f - zclampf(a.res.get(v), 0.0f, 1.0f);       // :118
    bool multi = false;
#pragma unroll
    for (uint32_t sub = 0; sub < SUB; sub++, j++) {
        if (j >= a.C) break;                                          // (block-uniform)
        uint64_t *cs = a.cs + (size_t)j * 4 * V + v;
        cs[0] = r.s0; cs[V] = r.s1; cs[2 * V] = r.s2; cs[3 * V] = r.s3;
        const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end), nf = f1 - f0;
        float l = 0.0f, b = 0.0f;
        uint32_t k = 0;
        for (; k + 8 <= nf; k += 8) {
            float t[8];
            noise_tile8(r, t, multi);                                 // zero(temp); temp += noise
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) svf_step(l, b, t[q], cut, res);   // Filter.zig:135-144
        }
        for (; k < nf; k++) {
            const float white = zrandom_float32_multi(r, multi) * 2.0f - 1.0f;
            svf_step(l, b, 0.0f + white, cut, res);
        }
        a.e[(size_t)(j + 1) * V + v] = make_float2(l, b);
        if (a.pred[0] && f1 == a.end && nf > 0) { a.pred[0][v] = r.s0; a.pred[1][v] = r.s1; a.pred[2][v] = r.s2; a.pred[3][v] = r.s3; }
    }
    if (multi) a.flag[v] = a.serial;                                  // (every later chunk of this voice started at the wrong draw)
}
__global__ void __launch_bounds__(256) k_nf_tp_a(const NfTpArgs a) {
    __shared__ uint4 tbl[kNoiseJumpEntries];
    nf_tp_a_run<1>(a, blockIdx.x, tbl);
}

// Pass B, same grid: scan, then the reference's recurrence over the regenerated noise of the chunk.  A flagged voice is painted
// whole by its chunk-0 lane, sequentially from the module's state -- the reference's own walk, bit for bit.
template <bool ZF>
__device__ __forceinline__ void nf_tp_b_run(const NfTpArgs &a, uint32_t bid) {
    uint32_t j, g;
    if (!nf_tp_block_b(a, j, g, bid)) return;
    const uint32_t v = g * 256 + threadIdx.x;
    if (v >= a.V) return;
    const size_t V = a.V;
    const float cut = zclampf(a.cutoff.get(v), 0.0f, 1.0f);
    const float res = 1.0f - zclampf(a.res.get(v), 0.0f, 1.0f);
    const uint32_t voff = v * 4u, orow = (uint32_t)a.out.stride * 4u;
    // walked frame by frame by its chunk-0 lane: a voice that met a multi-draw sample, and a voice whose cutoff is near zero (kTpExactCutBelow)
    const bool flagged = a.flag[v] == a.serial || cut < kTpExactCutBelow;
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end);     // workgroup-uniform: the frame loops stay scalar
    const uint64_t *cs = a.cs + (size_t)j * 4 * V + v;
    ZXoshiro r{cs[0], cs[V], cs[2 * V], cs[3 * V]};
    const float2 s0 = a.e[v];
    float l = s0.x, b = s0.y;
    svf_scan<kNfTpMaxChunks - 1>(l, b, cut, res, a.L, j, [&](uint32_t i) ZH_INLINE_LAMBDA { return a.e[(size_t)(i + 1) * V + v]; });
    auto frame = [&](const zh_rsrc_t &ro, uint32_t k, float temp, float base, bool store) ZH_INLINE_LAMBDA {
        const SvfOut sv = svf_step(l, b, temp, cut, res);            // Filter.zig:135-144
        const float val = sv.l * a.l_mul + sv.b * a.b_mul + sv.h * a.h_mul;   // :146
        if (store) zrow_store<1>(ro, voff, k * orow, base + val);
    };
    // (8 rows per descriptor: 32-bit offsets, common.hip.h kMaxRowStride)
    bool multi = false;                                              // (pass A has reported it already)
    uint32_t c0 = f0;
    for (; c0 + 8 <= f1; c0 += 8) {
        const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, c0);
        float row[8], t[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) row[k] = ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow);     // the `+=` base
        noise_tile8(r, t, multi);                                    // Noise.zig:51: zero(temp); temp += noise
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) frame(ro, k, t[k], row[k], !flagged);
    }
    if (c0 < f1) {
        const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, c0);
        for (uint32_t k = 0; c0 + k < f1; k++) frame(ro, k, 0.0f + (zrandom_float32(r) * 2.0f - 1.0f), ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow), !flagged);
    }
    if (!flagged && f1 == a.end && f1 > f0) {                         // whoever painted the span's last frame leaves the states
        a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
        a.l[v] = l; a.b[v] = b;
        if (a.e_next0) a.e_next0[v] = make_float2(l, b);
    }
    // A flagged voice (one of its draws took Random.float's second draw: every later chunk started at the wrong one) is painted
    // whole by its chunk-0 lane, sequentially from the module's state: the reference's own walk, bit for bit.
    if (j == 0 && __builtin_amdgcn_ballot_w64(flagged) != 0) {
        if (flagged) {
            r = ZXoshiro{a.s[0][v], a.s[1][v], a.s[2][v], a.s[3][v]};
            l = s0.x; b = s0.y;
            for (uint32_t f = a.start; f < a.end; f++) {
                const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f);
                frame(ro, 0, 0.0f + (zrandom_float32(r) * 2.0f - 1.0f), ZF ? 0.0f : zrow_load<1>(ro, voff, 0), true);
            }
            a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
            a.l[v] = l; a.b[v] = b;
            if (a.e_next0) { a.e_next0[v] = make_float2(l, b); a.flag_next[v] = a.serial_next; }
        }
    }
}
template <bool ZF>
__global__ void __launch_bounds__(256) k_nf_tp_b(const NfTpArgs a) { nf_tp_b_run<ZF>(a, blockIdx.x); }
// Pass B of one paint and pass A of the NEXT in one launch (composite.hip zh_noise_filter_paint, pipelined recording): blocks
// [0, a_blocks) are pass A's grid (the longer lanes: dispatched first), the rest pass B's -- neither needs anything the other makes
// (NfTpArgs), so one's launch ramp, table load and jump hide behind the other's frame loops.  a_blocks is a multiple of 8: both keep
// their block -> XCD mapping.
template <bool ZF>
__global__ void __launch_bounds__(256) k_nf_tp_ba(const NfTpArgs b, const NfTpArgs a, uint32_t a_blocks) {
    __shared__ uint4 tbl[kNoiseJumpEntries];
    if (blockIdx.x < a_blocks) nf_tp_a_run<kNfTpFusedSub>(a, blockIdx.x, tbl);
    else nf_tp_b_run<ZF>(b, blockIdx.x - a_blocks);
}
#endif   // ZH_FILTER_TP_NOISE

// ---------------------------------------------------------------------------------------------------- pink Noise
// Paul Kellett's filter (Noise.zig:58-66) is six decoupled one-pole taps b_k' = a_k b_k + c_k w over the white samples, a seventh
// value that is last frame's white sample times a constant, and a left-to-right sum: every tap is its own affine map with a
// CONSTANT a_k, so the time-parallel scheme above applies tap by tap (A^L is the scalar a_k^L), and b[6] needs no scan at all --
// a chunk's start value is the previous chunk's last white sample * 0.115926, which that chunk's pass A leaves behind exactly.
// The white samples themselves (generator jumps, multi-draw voices walked sequentially by their chunk-0 lane) are exact, as in
// the fused Noise -> Filter voice.  (Noise.zig:68: the taps are never written back -- a paint leaves only the generator's state.)
#if defined(ZH_FILTER_TP_PINK)            // (modules.hip only: the kernels below are not templates)
struct PinkTpArgs {
    uint64_t *s[4];              // generator states: read by pass A, written once by pass B
    const float *b0;             // [7][V]: the module's taps at span start (Noise.zig:55 `var b = self.b`)
    float *b_end;                // [7][V]: the taps after this launch's last frame (the next piece of the same span starts from them)
    uint64_t *cs;                // scratch [C][4][V]: generator state at the start of chunk j
    float *e;                    // scratch [C][7][V]: chunk j's zero-state tap values after its last frame, and its b[6] (exact)
    uint32_t *flag;              // scratch [V]: == serial when a multi-draw sample was seen in this paint
    uint32_t serial;
    const uint4 *tables;
    uint32_t V, start, end, L, C, per;   // per = 256-voice groups
    Img out;
};
__device__ __forceinline__ void pink_coeffs(float (&a)[6]) {
    a[0] = 0.99886f; a[1] = 0.99332f; a[2] = 0.96900f; a[3] = 0.86650f; a[4] = 0.55000f; a[5] = -0.7616f;
}

__global__ void __launch_bounds__(256) k_pink_tp_a(const PinkTpArgs a) {
    __shared__ uint4 tbl[kNoiseJumpEntries];
    uint32_t j, g;
    if (!nf_tp_block(a, j, g)) return;                                // chunk j of every group on XCD j % 8
    if (j > 0) {                                                      // block-uniform
        const uint4 *t = a.tables + (size_t)(j * (a.L / 32) - 1) * kNoiseJumpEntries;
        uint4 w[kNoiseJumpEntries / 256];
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) w[q] = t[q * 256 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) tbl[q * 256 + threadIdx.x] = w[q];
        __syncthreads();
    }
    const uint32_t v = g * 256 + threadIdx.x;
    if (v >= a.V) return;
    ZXoshiro r{a.s[0][v], a.s[1][v], a.s[2][v], a.s[3][v]};
    if (j > 0) noise_jump_apply(r, tbl);
    const size_t V = a.V;
    uint64_t *cs = a.cs + (size_t)j * 4 * V + v;
    cs[0] = r.s0; cs[V] = r.s1; cs[2 * V] = r.s2; cs[3 * V] = r.s3;
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end), nf = f1 - f0;
    float b[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool multi = false;
#pragma unroll 8
    for (uint32_t k = 0; k < nf; k++) {
        const float white = zrandom_float32_multi(r, multi) * 2.0f - 1.0f;   // Noise.zig:58
        (void)pink_step(b, white);                                    // :59-66 from zero taps: the chunk's zero-state response
    }
    float *e = a.e + (size_t)j * 7 * V + v;
#pragma unroll
    for (int q = 0; q < 7; q++) e[(size_t)q * V] = b[q];
    if (multi) a.flag[v] = a.serial;
}

template <bool ZF>
__global__ void __launch_bounds__(256) k_pink_tp_b(const PinkTpArgs a) {
    uint32_t j, g;
    if (!nf_tp_block_b(a, j, g)) return;
    const uint32_t v = g * 256 + threadIdx.x;
    if (v >= a.V) return;
    const size_t V = a.V;
    const uint32_t voff = v * 4u, orow = (uint32_t)a.out.stride * 4u;
    const bool flagged = a.flag[v] == a.serial;
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end);     // workgroup-uniform
    const uint64_t *cs = a.cs + (size_t)j * 4 * V + v;
    ZXoshiro r{cs[0], cs[V], cs[2 * V], cs[3 * V]};
    float b[7], b_span[7];
#pragma unroll
    for (int q = 0; q < 7; q++) b[q] = b_span[q] = a.b0[(size_t)q * V + v];
    if (j > 0) {
        // every earlier chunk's end values requested before the first is used (kTpMaxChunks - 1 slots, the rest ignored)
        float e[kTpMaxChunks - 1][6];
#pragma unroll
        for (uint32_t i = 0; i < kTpMaxChunks - 1; i++)
#pragma unroll
            for (int q = 0; q < 6; q++) e[i][q] = a.e[((size_t)i * 7 + q) * V + v];
        float ac[6];
        pink_coeffs(ac);
        float m[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {                                 // a_k^L by squaring in f64, rounded once
            double p = 1.0, x = (double)ac[q];
            for (uint32_t n = a.L;;) { if (n & 1u) p *= x; n >>= 1; if (!n) break; x *= x; }
            m[q] = (float)p;
        }
#pragma unroll
        for (uint32_t i = 0; i < kTpMaxChunks - 1; i++)
            if (i < j) {                                              // wave-uniform
#pragma unroll
                for (int q = 0; q < 6; q++) b[q] = m[q] * b[q] + e[i][q];
            }
        b[6] = a.e[((size_t)(j - 1) * 7 + 6) * V + v];                // the previous chunk's last white * 0.115926: exact
    }
    auto frame = [&](const zh_rsrc_t &ro, uint32_t k, float base, bool store) ZH_INLINE_LAMBDA {
        const float white = zrandom_float32(r) * 2.0f - 1.0f;       // Noise.zig:58
        const float val = pink_step(b, white);                       // :59-66
        if (store) zrow_store<1>(ro, voff, k * orow, base + val);
    };
    uint32_t c0 = f0;
    for (; c0 + 8 <= f1; c0 += 8) {
        const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, c0);
        float base[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) base[k] = ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) frame(ro, k, base[k], !flagged);
    }
    if (c0 < f1) {
        const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, c0);
        for (uint32_t k = 0; c0 + k < f1; k++) frame(ro, k, ZF ? 0.0f : zrow_load<1>(ro, voff, k * orow), !flagged);
    }
    if (!flagged && f1 == a.end && f1 > f0) {                         // Noise.zig:71 (:68: the MODULE's taps stay as they were)
        a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
#pragma unroll
        for (int q = 0; q < 7; q++) a.b_end[(size_t)q * V + v] = b[q];
    }
    if (j == 0 && __builtin_amdgcn_ballot_w64(flagged) != 0) {        // a multi-draw voice: the reference's own walk over the whole span
        if (flagged) {
            r = ZXoshiro{a.s[0][v], a.s[1][v], a.s[2][v], a.s[3][v]};
#pragma unroll
            for (int q = 0; q < 7; q++) b[q] = b_span[q];
            for (uint32_t f = a.start; f < a.end; f++) {
                const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f);
                frame(ro, 0, ZF ? 0.0f : zrow_load<1>(ro, voff, 0), true);
            }
            a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
#pragma unroll
            for (int q = 0; q < 7; q++) a.b_end[(size_t)q * V + v] = b[q];
        }
    }
}
#endif   // ZH_FILTER_TP_PINK

