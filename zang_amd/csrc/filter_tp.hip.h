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
// re-association of the recurrence can meet.  DESIGN.md 5a states both figures.
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
#include "tp_xchg.hip.h"

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

#if !defined(ZH_DEVICE_ONLY)
// chunks for a span of n frames of V voices: enough for ~2,048 waves (two per SIMD), 2..32; 0 = too many voices for the form
static inline uint32_t zh_tp_chunks(uint32_t V, const char *max_env, uint32_t n, uint32_t max_default = 16384u) {
    const uint32_t G = (V + 63) / 64;
    const char *me = zh_env(max_env);                                 // largest voice count that takes the time-parallel form
    const uint32_t tp_max = me ? (uint32_t)strtoul(me, nullptr, 10) : max_default;
    if (V > tp_max || G > 1024) return 0;
    const char *we = zh_env("ZH_TP_WAVES");                           // experiments: waves a launch should reach
    const uint32_t waves = we ? (uint32_t)strtoul(we, nullptr, 10) : 2048u;
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
    tp_input_tiles<CB, RB>(a, v * 4u, f0, f1, cut, res, [&](uint32_t, float in, float c, float r) ZH_INLINE_LAMBDA {
        svf_core(l, b, in, c, r);
        if (CB || RB) { svf_hom_step(ul, ub, c, r); svf_hom_step(wl, wb, c, r); }
    });
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
    tp_input_tiles<CB, RB>(a, voff, f0, f1, cut, res, [&](uint32_t f, float in, float c, float r) ZH_INLINE_LAMBDA {
        const SvfOut sv = svf_core(l, b, in, c, r);                   // :138-144
        const float val = sv.l * a.l_mul + sv.b * a.b_mul + sv.h * a.h_mul;   // :146
        const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f);
        const float base = ZF ? 0.0f : zrow_load<1>(ro, voff, 0);
        zrow_store<1>(ro, voff, 0, base + val);
    });
    if (f1 == a.end && f1 > f0) { a.l[v] = l; a.b[v] = b; }
}

#if !defined(ZH_DEVICE_ONLY)
// floats of scratch per voice: (kTpMaxChunks + 1) float2 of e, then kTpMaxChunks float4 of transition matrices
constexpr size_t kFilterTpFloats = (size_t)(kTpMaxChunks + 1) * 2 + (size_t)kTpMaxChunks * 4;
// Launches the span as pieces of <= 32 chunks.  false = not taken (too many voices, a short span): the caller paints with its
// exact form.  `scratch` = the module's, kFilterTpFloats * V floats.
static inline bool zh_filter_tp_launch(hipStream_t st, float *l, float *b, float *scratch, uint32_t V, Img out, CImg in, uint32_t start, uint32_t end, bool zf,
                                       float l_mul, float b_mul, float h_mul, CobP cut, CobP res) {
    if (end - start < 64) return false;
    const uint32_t C = zh_tp_chunks(V, "ZH_FILTER_TP_MAX", end - start);
    if (C < 2) return false;
    const uint32_t piece = 4096;                                      // frames per launch pair: chunks of <= 128 frames
    FilterTpArgs a;
    a.l = l; a.b = b; a.e = reinterpret_cast<float2 *>(scratch); a.m = reinterpret_cast<float4 *>(scratch + (size_t)(kTpMaxChunks + 1) * 2 * V);
    a.V = V; a.out = out; a.input = in; a.l_mul = l_mul; a.b_mul = b_mul; a.h_mul = h_mul; a.cutoff = cut; a.res = res;
    const bool cb = cut.is_buffer != 0, rb = res.is_buffer != 0;
    for (uint32_t s = start; s < end; s += piece) {
        a.start = s; a.end = min(s + piece, end);
        a.L = (a.end - a.start + C - 1) / C;
        const dim3 grid((V + 255) / 256, (a.end - a.start + a.L - 1) / a.L);
#define ZH_FTP(CB_, RB_)                                                                                   \
        do {                                                                                               \
            hipLaunchKernelGGL((k_filter_tp_a<CB_, RB_>), grid, dim3(256), 0, st, a);                      \
            if (zf) hipLaunchKernelGGL((k_filter_tp_b<true, CB_, RB_>), grid, dim3(256), 0, st, a);        \
            else hipLaunchKernelGGL((k_filter_tp_b<false, CB_, RB_>), grid, dim3(256), 0, st, a);          \
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

// ---------------------------------------------------------------------------------------------------- white Noise -> Filter
#if defined(ZH_FILTER_TP_NOISE)          // (composite.hip only)
constexpr uint32_t kNfTpMaxChunks = kTpMaxChunks;
template <uint32_t I> struct TpIdx { static constexpr uint32_t value = I; };
// ONE launch (round 5; rounds 3-4 ran this as two kernels with the chunk states, the generator states and every e_j going
// through HBM between them, and generated every chunk's noise twice).  Workgroup = (256 voices, chunk j); chunk 0's workgroups
// double as the scanners:
//   chunk j >= 1   jump the generator to the chunk's first draw (one table per workgroup, in LDS), generate the chunk's noise ONCE
//                  into registers, run the zero-state recurrence over it -> e_j, publish e_j (tp_xchg.hip.h); wait for the chunk's
//                  start state from the scanner; run the reference's own recurrence over the SAME registers from it and paint.
//   chunk 0        has no jump and starts from the module's state itself: it paints its frames at once (exact), which also gives
//                  s_1 exactly; then it gathers e_1 .. e_{C-2} of its 64 voices, folds s_{j+1} = M s_j + e_j (M = A^L formed in f64,
//                  rounded once; the steps in f32, as the two-kernel form did), publishes every s_j and raises the wave's flag.
//   multi-draw     a chunk that sees one of Random.float's second draws (2^-41 per sample) says so in its flag; the scanner
//                  collects the lane masks, hands them on with the start states, the voice's chunk lanes then store nothing, and
//                  the scanner's own lane walks the voice from the end of chunk 0 to the end of the span sequentially --
//                  the reference's own walk, bit for bit.
// Chunk j of every voice group runs on XCD j % 8 (block -> (chunk, group) mapping below), so a jump table is fetched into ONE
// XCD's L2 instead of all eight.  Every workgroup of the launch must be resident at once (chunk 0 waits for chunks that are
// dispatched after it): the host sizes the launch by zh_tp1_resident_workgroups.
struct NfTp1Args {
    uint64_t *s[4];              // the voices' generator states (Noise.zig:9)
    float *l, *b;                // filter state
    uint64_t *e;                 // scratch [C][V]: e_j packed {l, b}
    uint64_t *st;                // scratch [C][V]: chunk j's start state, published by the scanner
    uint64_t *eflag;             // scratch [C][W][2]
    uint64_t *sflag;             // scratch [W][2]
    uint32_t *sync;              // {launch number, finished workgroups}
    const uint4 *tables;         // T^(32 k), k = 1..63
    uint32_t V, start, end, C, per, W;   // per = 256-voice groups, W = 64-voice waves
    Img out;
    float l_mul, b_mul, h_mul;
    F32P cutoff, res;
};

// LDS of a workgroup: the jump table (32 KiB), whose space the chunk's noise takes over once every thread has jumped
// ([frame][thread] floats: 32 KiB per 32 frames).  At least 64 KiB are declared whatever NB is, so that a CU (160 KiB) holds
// at most two workgroups and the dispatcher spreads a 512-workgroup launch two per CU (three on some CUs and one on others
// made the slowest chunk 40 % later than the median, profiles/r05/nftp1_trace_4096.txt).  NB >= 3 (96 / 128 KiB of noise)
// keeps the chunk-start generator state instead and draws the chunk's noise a second time.
// Eight consecutive white samples as Noise.paint adds them to a zeroed temp (Noise.zig:51).  The common case of Random.float
// (the draw's high word is non-zero) has no test per sample: a compare and a scalar branch after EVERY sample cost more than the
// sample's arithmetic (tools/exp: the frame loops ran at ~7 cycles per instruction with them).  The tile's minimum high word
// is tested once; when it is zero (2^-32 per sample) the tile is drawn again with the careful form, which also reports a
// second draw (2^-41) in `multi`.
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

template <int NB> struct NfTp1Lds { static constexpr bool kKeep = NB <= 2; static constexpr uint32_t kUint4 = kNoiseJumpEntries * 2u; };

template <bool ZF, int NB>                // chunks of L = 32 * NB frames
__global__ void __launch_bounds__(256, 2) k_nf_tp1(const NfTp1Args a) {
    __shared__ uint4 tbl[NfTp1Lds<NB>::kUint4];
    constexpr uint32_t L = 32u * NB;
    constexpr bool KEEP = NfTp1Lds<NB>::kKeep;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t id = blockIdx.x, band = id / (8u * a.per), rr = id % (8u * a.per);
    const uint32_t j = band * 8u + (rr & 7u), g = rr >> 3;
    if (j >= a.C) return;                                             // (the last band of a launch whose C is not a multiple of 8)
    const uint32_t v = g * 256u + tid;
    const bool live = v < a.V;
    const uint32_t vc = live ? v : a.V - 1u;                          // a dead lane computes on a live voice's data and stores nothing
    const uint32_t w = g * 4u + (tid >> 6);
    const size_t V = a.V;
    const uint32_t tag = tp_launch_tag(a.sync);
    if (j > 0) {                                                      // block-uniform
        const uint4 *t = a.tables + (size_t)(j * NB - 1u) * kNoiseJumpEntries;
        uint4 q8[kNoiseJumpEntries / 256];                            // the thread's eight entries requested together, then parked
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) q8[q] = t[q * 256 + tid];
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) tbl[q * 256 + tid] = q8[q];
        __syncthreads();
    }
    ZXoshiro r{a.s[0][vc], a.s[1][vc], a.s[2][vc], a.s[3][vc]};
    const float cut = zclampf(a.cutoff.get(vc), 0.0f, 1.0f);          // Filter.zig:114
    const float res = 1.0f - zclampf(a.res.get(vc), 0.0f, 1.0f);      // :118
    const uint32_t f0 = min(a.start + j * L, a.end), f1 = min(f0 + L, a.end), nf = f1 - f0;   // workgroup-uniform
    const uint32_t orow = (uint32_t)a.out.stride * 4u, voff = v * 4u;
    const float l_mul = a.l_mul, b_mul = a.b_mul, h_mul = a.h_mul;
    float l, b;
    bool multi = false;
    if (j > 0) {
        noise_jump_apply(r, tbl);
        const ZXoshiro r_chunk = r;                                   // (!KEEP: the chunk's noise is drawn a second time from here)
        if (KEEP) __syncthreads();                                    // every thread is done with the table: its space now holds the noise
        float *wn = reinterpret_cast<float *>(tbl) + tid;             // wn[k * 256]
        l = 0.0f; b = 0.0f;
        uint32_t k = 0;
        for (; k + 8u <= nf; k += 8u) {
            float t[8];
            noise_tile8(r, t, multi);                                 // zero(temp); temp += noise
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                if (KEEP) wn[(k + q) * 256u] = t[q];
                svf_step(l, b, t[q], cut, res);                       // Filter.zig:135-144 from a zero state: e_j
            }
        }
        for (; k < nf; k++) {
            const float white = zrandom_float32_multi(r, multi) * 2.0f - 1.0f;
            const float temp = 0.0f + white;
            if (KEEP) wn[k * 256u] = temp;
            svf_step(l, b, temp, cut, res);
        }
        if (live) tp_store8(a.e + (size_t)j * V + v, tp_pack(l, b));
        tp_drain();
        const uint64_t mm = __builtin_amdgcn_ballot_w64(multi && live);
        if (lane == 0) tp_flag_set(a.eflag + ((size_t)j * a.W + w) * 2u, tag, mm);
        const uint64_t flagged = tp_flag_wait(a.sflag + (size_t)w * 2u, tag);
        tp_unpack(tp_load8(a.st + (size_t)j * V + vc), l, b);
        const bool wr = live && !((flagged >> lane) & 1ull);
        ZXoshiro r2 = r_chunk;
        bool multi2 = false;
        // ALL: every lane of the wave writes (the common case, decided once per wave): no exec-mask region around each store
        auto second_half = [&](auto all_c) ZH_INLINE_LAMBDA {
            constexpr bool ALL = decltype(all_c)::value != 0u;
            auto frame = [&](const zh_rsrc_t &ro, float temp, uint32_t q, float base) ZH_INLINE_LAMBDA {
                const SvfOut sv = svf_step(l, b, temp, cut, res);     // :135-144: the reference's recurrence, from s_j
                const float val = sv.l * l_mul + sv.b * b_mul + sv.h * h_mul;   // :146
                if (ALL || wr) zrow_store<1>(ro, voff, q * orow, base + val);
            };
            uint32_t kk = 0;
            for (; kk + 8u <= nf; kk += 8u) {                         // (8 rows per descriptor: 32-bit offsets, common.hip.h kMaxRowStride)
                const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f0 + kk);
                float base[8], t[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) base[q] = (!ZF && (ALL || wr)) ? zrow_load<1>(ro, voff, q * orow) : 0.0f;
                if (KEEP) {
#pragma unroll
                    for (uint32_t q = 0; q < 8; q++) t[q] = wn[(kk + q) * 256u];
                } else noise_tile8(r2, t, multi2);
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) frame(ro, t[q], q, base[q]);
            }
            if (kk < nf) {
                const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f0 + kk);
                for (uint32_t q = 0; kk + q < nf; q++) {
                    const float temp = KEEP ? wn[(kk + q) * 256u] : 0.0f + (zrandom_float32(r2) * 2.0f - 1.0f);
                    frame(ro, temp, q, (!ZF && (ALL || wr)) ? zrow_load<1>(ro, voff, q * orow) : 0.0f);
                }
            }
        };
        if (__builtin_amdgcn_ballot_w64(!wr) == 0) second_half(TpIdx<1>{}); else second_half(TpIdx<0>{});
        if (wr && f1 == a.end && nf > 0) {                            // whoever painted the span's last frame leaves the states
            a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
            a.l[v] = l; a.b[v] = b;
        }
    } else {
        // ---- chunk 0: its own frames, exactly; then the scan for its 64 voices
        l = a.l[vc]; b = a.b[vc];
        auto first_chunk = [&](auto all_c) ZH_INLINE_LAMBDA {
            constexpr bool ALL = decltype(all_c)::value != 0u;
            auto frame0 = [&](const zh_rsrc_t &ro, float temp, uint32_t q, float base) ZH_INLINE_LAMBDA {
                const SvfOut sv = svf_step(l, b, temp, cut, res);
                const float val = sv.l * l_mul + sv.b * b_mul + sv.h * h_mul;
                if (ALL || live) zrow_store<1>(ro, voff, q * orow, base + val);
            };
            uint32_t k = 0;
            for (; k + 8u <= nf; k += 8u) {
                const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f0 + k);
                float base[8], t[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) base[q] = (!ZF && (ALL || live)) ? zrow_load<1>(ro, voff, q * orow) : 0.0f;
                noise_tile8(r, t, multi);
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) frame0(ro, t[q], q, base[q]);
            }
            if (k < nf) {
                const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f0 + k);
                for (uint32_t q = 0; k + q < nf; q++) {
                    const float temp = 0.0f + (zrandom_float32_multi(r, multi) * 2.0f - 1.0f);
                    frame0(ro, temp, q, (!ZF && (ALL || live)) ? zrow_load<1>(ro, voff, q * orow) : 0.0f);
                }
            }
        };
        if (__builtin_amdgcn_ballot_w64(!live) == 0) first_chunk(TpIdx<1>{}); else first_chunk(TpIdx<0>{});
        const float l1 = l, b1 = b;                                   // s_1, exact
        // M = A^L while the other chunks are still in their first half
        const Svf2x2 md = svf_pow(svf_hom(cut, res), L);
        const float m00 = (float)md.a00, m01 = (float)md.a01, m10 = (float)md.a10, m11 = (float)md.a11;
        // every other chunk's flag for this wave: lane i polls chunk i's
        uint64_t mymask = 0;
        {
            bool ok = !(lane >= 1u && lane < a.C);
            for (;;) {
                if (!ok) ok = tp_flag_try(a.eflag + ((size_t)lane * a.W + w) * 2u, tag, mymask);
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                tp_sleep();
            }
        }
        uint64_t flagged = __builtin_amdgcn_ballot_w64(multi && live);
        {
            const uint32_t mlo = (uint32_t)mymask, mhi = (uint32_t)(mymask >> 32);
            for (uint32_t i = 1; i < a.C; i++)                        // (uniform lane index)
                flagged |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mlo, (int)i) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mhi, (int)i) << 32);
        }
        uint64_t ev[kNfTpMaxChunks];
#pragma unroll
        for (uint32_t i = 1; i < kNfTpMaxChunks; i++) ev[i] = i + 1u < a.C ? tp_load8(a.e + (size_t)i * V + vc) : 0ull;   // e_{C-1} is not needed
        float sl = l1, sb = b1;
#pragma unroll
        for (uint32_t i = 1; i < kNfTpMaxChunks; i++)
            if (i < a.C) {                                            // uniform
                if (live) tp_store8(a.st + (size_t)i * V + v, tp_pack(sl, sb));   // chunk i starts from s_i
                float el, eb;
                tp_unpack(ev[i], el, eb);
                const float nl = (m00 * sl + m01 * sb) + el;          // s_{i+1} = M s_i + e_i
                const float nb = (m10 * sl + m11 * sb) + eb;
                sl = nl; sb = nb;
            }
        tp_drain();
        if (lane == 0) tp_flag_set(a.sflag + (size_t)w * 2u, tag, flagged);
        const bool mine = live && ((flagged >> lane) & 1ull);
        if (flagged != 0) {                                           // a multi-draw voice: the reference's own walk from the end of chunk 0
            if (mine) {
                for (uint32_t f = f1; f < a.end; f++) {
                    const zh_rsrc_t ro = zrow_rsrc(a.out.p, a.out.stride, f);
                    const float base = ZF ? 0.0f : zrow_load<1>(ro, voff, 0);
                    const float white = zrandom_float32(r) * 2.0f - 1.0f;
                    const SvfOut sv = svf_step(l, b, 0.0f + white, cut, res);
                    zrow_store<1>(ro, voff, 0, base + (sv.l * l_mul + sv.b * b_mul + sv.h * h_mul));
                }
                a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
                a.l[v] = l; a.b[v] = b;
            }
        }
        if (live && !mine && f1 == a.end && nf > 0) {                 // (a piece of one chunk: chunk 0 alone paints it, exactly)
            a.s[0][v] = r.s0; a.s[1][v] = r.s1; a.s[2][v] = r.s2; a.s[3][v] = r.s3;
            a.l[v] = l; a.b[v] = b;
        }
    }
    tp_launch_done(a.sync, tag, a.C * a.per);
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
    uint32_t V, start, end, L, C;
    Img out;
};
__device__ __forceinline__ void pink_coeffs(float (&a)[6]) {
    a[0] = 0.99886f; a[1] = 0.99332f; a[2] = 0.96900f; a[3] = 0.86650f; a[4] = 0.55000f; a[5] = -0.7616f;
}

__global__ void __launch_bounds__(256) k_pink_tp_a(const PinkTpArgs a) {
    __shared__ uint4 tbl[kNoiseJumpEntries];
    const uint32_t j = blockIdx.y;
    if (j > 0) {                                                      // block-uniform
        const uint4 *t = a.tables + (size_t)(j * (a.L / 32) - 1) * kNoiseJumpEntries;
        uint4 w[kNoiseJumpEntries / 256];
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) w[q] = t[q * 256 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < kNoiseJumpEntries / 256; q++) tbl[q * 256 + threadIdx.x] = w[q];
        __syncthreads();
    }
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
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
    const uint32_t j = blockIdx.y;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
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

