// script_rt.hip.h -- what a generated zangscript kernel (zang_amd/zangscript/emit_hip.py) is written
// against: the launch record shared with the loader (script.hip), accessors for the script module's
// Params, the [word][voice] state blob, and the frame loop with an accumulating output.
#pragma once
#include "common.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "seq.hip.h"
#include "envelope.hip.h"
#include "voices.hip.h"

// The [word][voice] state blob of a script module.  A kernel launched with gridDim.y > 1 paints the span as gridDim.y
// frame ranges at once (zs_frame_loop below): every range loads the span's start state from `cur`, and only the range
// that ends the span stores -- EVERY state word, into `next` -- and the loader then flips the two blobs on the host
// (script.hip, zh_flipper; the emitter exports the count of stored words, zs_state_words_stored_<name>, and the loader takes
// the range form only when it equals the module's state words).
struct ZsState {
    uint32_t *cur, *next;
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
    __device__ __forceinline__ uint32_t &operator[](size_t i) const { return cur[i]; }     // delay rings: read and written in place
#endif
};

struct ZsLaunch {
    uint32_t V, start, end, flags;
    float *out;
    uint32_t ostride, n_params;
    ZsState state;
    BoolP nic;                             // note_id_changed
    zh_script_param p[ZH_SCRIPT_MAX_PARAMS];
};

// frames per range of a launch with `ranges` = gridDim.y > 1 frame ranges over n frames (the loader picks `ranges` so that
// this reproduces the length it planned with, script.hip)
__host__ __device__ inline uint32_t zs_range_frames(uint32_t n, uint32_t ranges) { return ((n + ranges - 1) / ranges + 7) / 8 * 8; }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
__device__ const float zs_zero_row[1] = {0.0f};

__device__ __forceinline__ float zs_const(const zh_script_param &p, uint32_t v) { return p.pf ? p.pf[v] : p.f; }
__device__ __forceinline__ bool zs_bool(const zh_script_param &p, uint32_t v) { return p.pb ? p.pb[v] != 0 : p.u != 0; }
// the frame-loop input image of a waveform param, or of a cob param (a one-sample dummy row when it is a
// constant): the (wave-uniform) base pointer, the row stride and the lane's byte offset inside a row
__device__ __forceinline__ const float *zs_row(const zh_script_param &p, uint32_t v, size_t &stride, uint32_t &voff) {
    const bool img = p.kind == ZH_SP_BUFFER || p.is_buffer;
    stride = img ? p.stride : 0;
    voff = img ? v * 4u : 0u;
    return img ? p.pf : zs_zero_row;
}
__device__ __forceinline__ float zs_ld_f(const uint32_t *st, uint32_t word, uint32_t V, uint32_t v) { return zu2f(st[(size_t)word * V + v]); }
__device__ __forceinline__ uint32_t zs_ld_u(const uint32_t *st, uint32_t word, uint32_t V, uint32_t v) { return st[(size_t)word * V + v]; }
__device__ __forceinline__ uint64_t zs_ld_u64(const uint32_t *st, uint32_t word, uint32_t V, uint32_t v) {
    return (uint64_t)st[(size_t)word * V + v] | ((uint64_t)st[(size_t)(word + 1) * V + v] << 32);
}
__device__ __forceinline__ void zs_st_f(uint32_t *st, uint32_t word, uint32_t V, uint32_t v, float x) { st[(size_t)word * V + v] = zf2u(x); }
__device__ __forceinline__ void zs_st_u(uint32_t *st, uint32_t word, uint32_t V, uint32_t v, uint32_t x) { st[(size_t)word * V + v] = x; }
__device__ __forceinline__ void zs_st_u64(uint32_t *st, uint32_t word, uint32_t V, uint32_t v, uint64_t x) {
    st[(size_t)word * V + v] = (uint32_t)x;
    st[(size_t)(word + 1) * V + v] = (uint32_t)(x >> 32);
}

// the paint kernels' accessors (the generated text says `L.state`): loads from the start state; stores as described at ZsState
__device__ __forceinline__ float zs_ld_f(const ZsState &s, uint32_t word, uint32_t V, uint32_t v) { return zs_ld_f(s.cur, word, V, v); }
__device__ __forceinline__ uint32_t zs_ld_u(const ZsState &s, uint32_t word, uint32_t V, uint32_t v) { return zs_ld_u(s.cur, word, V, v); }
__device__ __forceinline__ uint64_t zs_ld_u64(const ZsState &s, uint32_t word, uint32_t V, uint32_t v) { return zs_ld_u64(s.cur, word, V, v); }
__device__ __forceinline__ uint32_t *zs_store_target(const ZsState &s) {
    if (gridDim.y > 1) return blockIdx.y + 1 == gridDim.y ? s.next : nullptr;
    return s.cur;
}
__device__ __forceinline__ void zs_st_f(const ZsState &s, uint32_t word, uint32_t V, uint32_t v, float x) { if (uint32_t *p = zs_store_target(s)) zs_st_f(p, word, V, v, x); }
__device__ __forceinline__ void zs_st_u(const ZsState &s, uint32_t word, uint32_t V, uint32_t v, uint32_t x) { if (uint32_t *p = zs_store_target(s)) zs_st_u(p, word, V, v, x); }
__device__ __forceinline__ void zs_st_u64(const ZsState &s, uint32_t word, uint32_t V, uint32_t v, uint64_t x) { if (uint32_t *p = zs_store_target(s)) zs_st_u64(p, word, V, v, x); }

template <bool B> struct zs_tag { static constexpr bool value = B; };   // selects one of a kernel's two frame bodies (zs_quiet)

// std.math.max / min as the generated Zig calls them (codegen_zig.zig:186-187): comparison selects
__device__ __forceinline__ float zs_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float zs_min(float a, float b) { return a < b ? a : b; }

// frame_loop (seq.hip.h) for a body that accumulates into the output sample itself: a script module's
// paint() may `+=` its output several times per frame.  f(frame, x[NIN], o&); `zf` = ZH_PAINT_ZERO_FIRST; `walk` is a flag
// of the caller's that the body reads: true while the frames before a frame range are replayed for their state only.
// `quiet(n)` (wave-uniform) is asked before every chunk of n = CH frames; where it holds the chunk runs `fq`, the body's
// second instance (no rare-path sine branch, no envelope stage end: zscript_emit.hip quiet_terms), else `f`.
template <int CH, int NIN, class F, class Q, class FQ>
__device__ __forceinline__ void zs_frame_loop(float *__restrict__ out, uint32_t v, size_t ostride, const float *const *in,
                                              const size_t *istride, const uint32_t *ivoff, uint32_t start, uint32_t end, bool zf, bool &walk, F &&f,
                                              Q &&quiet, FQ &&fq) {
    constexpr int NI = NIN > 0 ? NIN : 1;
    const uint32_t voff = v * 4u;                                   // rows through buffer descriptors: lanes.hip.h (zrow_*)
    const uint32_t orow = (uint32_t)ostride * 4u;
    if (gridDim.y > 1) {
        // Few voices: this workgroup paints one frame range [f0, f1) of the span.  The state that reaches f0 is the start
        // state after the frames before it, so the body first runs over those frames with its output discarded: what only
        // feeds the output (a sine, a curve, the mix) is dead code there and the compiler drops it; what feeds the state
        // (phase additions, envelope clocks, filter recurrences, generators) stays.  Same operations on the same values as
        // the sequential walk => same state => same bits.
        const uint32_t ch = zs_range_frames(end - start, gridDim.y);
        const uint32_t f0 = min(start + blockIdx.y * ch, end), f1 = min(f0 + ch, end);
        uint32_t r = start;
        walk = true;                                                  // the body's state-only forms (EnvLaneT::frame_s)
        for (; r + CH <= f0; r += CH) {                               // CH frames' input rows requested together, then their bodies
            // (requested a chunk AHEAD instead, like the painting loop below: Pluck 36.7 -> 40.5 us at 4,096 voices -- the copies
            // cost more than the waits; a constant's dummy row, zs_row's stride 0, is not loaded at all)
            float xr[NI][CH];
#pragma unroll
            for (int j = 0; j < NIN; j++) {
                if (istride[j] != 0) {                                // (kernel-uniform)
                    const zh_rsrc_t ri = zrow_rsrc(in[j], istride[j], r);
                    const uint32_t irow = (uint32_t)istride[j] * 4u;
#pragma unroll
                    for (int k = 0; k < CH; k++) xr[j][k] = zrow_load<1>(ri, ivoff[j], k * irow);
                } else {
#pragma unroll
                    for (int k = 0; k < CH; k++) xr[j][k] = 0.0f;     // zs_zero_row
                }
            }
            if (quiet(CH)) {                                            // (the replay too: a quiet chunk's walk is cheaper still)
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    float x[NI];
#pragma unroll
                    for (int j = 0; j < NIN; j++) x[j] = xr[j][k];
                    float o = 0.0f;
                    fq(r + k, x, o);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    float x[NI];
#pragma unroll
                    for (int j = 0; j < NIN; j++) x[j] = xr[j][k];
                    float o = 0.0f;
                    f(r + k, x, o);
                }
            }
        }
        for (; r < f0; r++) {
            float x[NI];
#pragma unroll
            for (int j = 0; j < NIN; j++) x[j] = zrow_load<1>(zrow_rsrc(in[j], istride[j], r), ivoff[j], 0);
            float o = 0.0f;
            f(r, x, o);
        }
        walk = false;
        start = f0; end = f1;
    }
    const uint32_t nfull = (end - start) / CH;
    float oc[CH], xc[NI][CH];
    uint32_t i = start;
    auto load = [&](uint32_t base, float (&o)[CH], float (&x)[NI][CH]) ZH_INLINE_LAMBDA {
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, base);
#pragma unroll
        for (int k = 0; k < CH; k++) o[k] = zf ? 0.0f : zrow_load<1>(ro, voff, k * orow);
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            if (istride[j] != 0) {                                    // (kernel-uniform; a constant's dummy row is not loaded)
                const zh_rsrc_t ri = zrow_rsrc(in[j], istride[j], base);
                const uint32_t irow = (uint32_t)istride[j] * 4u;
#pragma unroll
                for (int k = 0; k < CH; k++) x[j][k] = zrow_load<1>(ri, ivoff[j], k * irow);
            } else {
#pragma unroll
                for (int k = 0; k < CH; k++) x[j][k] = 0.0f;          // zs_zero_row
            }
        }
    };
    if (nfull > 0) load(i, oc, xc);
    for (uint32_t c = 0; c < nfull; c++, i += CH) {
        float on[CH], xn[NI][CH];
        const bool more = c + 1 < nfull;
        if (more) load(i + CH, on, xn);
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        // compute the chunk, then store it (seq.hip.h frame_loop: a store after every frame would pin every load of the
        // body -- delay rings, track tables -- behind the previous frame's store)
        if (quiet(CH)) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float x[NI];
#pragma unroll
                for (int j = 0; j < NIN; j++) x[j] = xc[j][k];
                fq(i + k, x, oc[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float x[NI];
#pragma unroll
                for (int j = 0; j < NIN; j++) x[j] = xc[j][k];
                f(i + k, x, oc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) zrow_store<1>(ro, voff, k * orow, oc[k]);
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                oc[k] = on[k];
#pragma unroll
                for (int j = 0; j < NIN; j++) xc[j][k] = xn[j][k];
            }
        }
    }
    for (; i < end; i++) {
        float x[NI];
#pragma unroll
        for (int j = 0; j < NIN; j++) x[j] = zrow_load<1>(zrow_rsrc(in[j], istride[j], i), ivoff[j], 0);
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        float o = zf ? 0.0f : zrow_load<1>(ro, voff, 0);
        f(i, x, o);
        zrow_store<1>(ro, voff, 0, o);
    }
}
// one body for every chunk
template <int CH, int NIN, class F>
__device__ __forceinline__ void zs_frame_loop(float *__restrict__ out, uint32_t v, size_t ostride, const float *const *in,
                                              const size_t *istride, const uint32_t *ivoff, uint32_t start, uint32_t end, bool zf, bool &walk, F &&f) {
    zs_frame_loop<CH, NIN>(out, v, ostride, in, istride, ivoff, start, end, zf, walk, f, [](int) { return false; }, f);
}
// ---- the role-wave form of a generated kernel (zs_paint_pc_<name>; zscript_emit.hip plan_roles) -------------------------
// Few voices: one workgroup of several waves owns 64 voices, and the frame body's units (builtin modules, arithmetic)
// are dealt to ROLES -- waves that each run their own part of every frame and hand values on through LDS tiles, float4 =
// four frames of a lane side by side ([frame / 4][lane]), CH frames per tile.  Role r works on tile (step - lag(r)); one
// barrier per step; a value handed from role a to role b lives in lag(b) - lag(a) + 1 tile buffers.  The loader roles fetch the
// input rows (and the live output rows of a paint without ZH_PAINT_ZERO_FIRST) one tile ahead, the writer role -- the
// last -- owns `o`: every `+=` into the output in the order the script gives, then the row store.  Lanes past the last voice
// run voice V - 1 again (same loads, same values, same address), so nothing inside a chain is masked.
// Same per-voice operations on the same values in the same order as the lane form => same bits.
struct ZsTileRef { uint32_t off, depth; };     // off: float4 index of buffer 0 in the workgroup's LDS; a buffer = [CH / 4][64] float4

__device__ __forceinline__ float &zs_comp(float4 &q, int k) { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; }
__device__ __forceinline__ float zs_comp(const float4 &q, int k) { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; }

// the pipeline of one role: f / fq(frame, zin[NIN], zout[NOUT], o&) -- fq where quiet(4) holds for the next four frames.
// WR: the writer role -- `o` starts as 0 (zf) or as the LAST input tile's value (the live output row the loader fetched),
// and is stored to the output image after the frame.
// K > 1: the role runs in K waves (`rep` = 0 .. K - 1) that each compute the values of every K-th group of four frames and only
// WALK the others -- the body with its results dropped, of which the compiler keeps what carries state (a phase counter's add,
// an envelope's clock), like the replay of a frame range (zs_frame_loop).  The emitter replicates only roles whose walk is cheap.
template <int CH_, int NIN, int NOUT, int UQ, bool WR, int K, class F, class QT, class FQ>
__device__ __forceinline__ void zs_role_run(float4 *lds, uint32_t lane, uint32_t lag, uint32_t rep, uint32_t steps, uint32_t start, uint32_t n_frames,
                                            const ZsTileRef *tin, const ZsTileRef *tout, float *__restrict__ out, size_t ostride, uint32_t voff, bool zf,
                                            F &&f, QT &&quiet, FQ &&fq) {
    constexpr uint32_t CH = CH_, Q = CH / 4, TILE = Q * 64;
    constexpr int NI = NIN > 0 ? NIN : 1, NO = NOUT > 0 ? NOUT : 1;
    const uint32_t nchunks = (n_frames + CH - 1) / CH;
    const uint32_t orow = (uint32_t)ostride * 4u;
    uint32_t bi[NI], bo[NO];
#pragma unroll
    for (int j = 0; j < NI; j++) bi[j] = 0;
#pragma unroll
    for (int j = 0; j < NO; j++) bo[j] = 0;
#if defined(ZS_PC_TRACE)                                             // tools/exp/role_probe.py: cycles between barriers against cycles in all
    unsigned long long zs_busy = 0, zs_t_begin = __builtin_readcyclecounter(), zs_t0 = zs_t_begin;
#endif
    for (uint32_t c = 0; c < steps; c++) {
#if defined(ZS_PC_TRACE)
        zs_t0 = __builtin_readcyclecounter();
#endif
        if (c >= lag && c - lag < nchunks) {
            const uint32_t d = c - lag, nf = min(CH, n_frames - d * CH), base = start + d * CH;
            const float4 *ti[NI];
            float4 *to[NO];
#pragma unroll
            for (int j = 0; j < NIN; j++) ti[j] = lds + tin[j].off + bi[j] * TILE + lane;
#pragma unroll
            for (int j = 0; j < NOUT; j++) to[j] = lds + tout[j].off + bo[j] * TILE + lane;
            if (nf == CH) {
                float4 a[NI];
                constexpr int NLD = WR ? NIN - 1 : NIN;                 // (the writer's last tile is the live output row: not read when the paint zeroes first)
#pragma unroll
                for (int j = 0; j < NLD; j++) a[j] = ti[j][0];
                if (WR) a[NI - 1] = zf ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : ti[NI - 1][0];
                const bool qtile = quiet((int)CH);                    // the whole tile quiet: no test per group of four frames
#pragma unroll UQ
                for (uint32_t q = 0; q < Q; q++) {
                    float4 an[NI];                                    // the next four frames' values: fetched while these compute
                    if (q + 1 < Q) {
#pragma unroll
                        for (int j = 0; j < NLD; j++) an[j] = ti[j][(q + 1) * 64];
                        if (WR) an[NI - 1] = zf ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : ti[NI - 1][(q + 1) * 64];
                    }
                    float4 b[NO];
                    float o4[4];
                    auto quad = [&](auto &&g, auto keep) ZH_INLINE_LAMBDA {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            float zi[NI], zo[NO];
#pragma unroll
                            for (int j = 0; j < NIN; j++) zi[j] = zs_comp(a[j], k);
                            o4[k] = (WR && !zf) ? zi[NI - 1] : 0.0f;
                            g(base + 4 * q + k, zi, zo, o4[k]);
                            if constexpr (decltype(keep)::value) {
#pragma unroll
                                for (int j = 0; j < NOUT; j++) zs_comp(b[j], k) = zo[j];
                            }
                        }
                    };
                    if (K == 1 || q % K == rep) {
                        if (qtile || quiet(4)) quad(fq, zs_tag<true>{}); else quad(f, zs_tag<true>{});
#pragma unroll
                        for (int j = 0; j < NOUT; j++) to[j][q * 64] = b[j];
                    } else {
                        if (qtile || quiet(4)) quad(fq, zs_tag<false>{}); else quad(f, zs_tag<false>{});
                    }
                    if (WR) {
                        const zh_rsrc_t ro = zrow_rsrc(out, ostride, base + 4 * q);
#pragma unroll
                        for (int k = 0; k < 4; k++) zrow_store<1>(ro, voff, k * orow, o4[k]);
                    }
                    if (q + 1 < Q) {
#pragma unroll
                        for (int j = 0; j < NIN; j++) a[j] = an[j];
                    }
                }
            } else {                                                  // the span's last, partial tile: frame by frame
                for (uint32_t k = 0; k < nf; k++) {
                    float zi[NI], zo[NO];
#pragma unroll
                    for (int j = 0; j < NIN; j++) zi[j] = reinterpret_cast<const float *>(&ti[j][(k >> 2) * 64])[k & 3];
                    float o = (WR && !zf) ? zi[NI - 1] : 0.0f;
                    f(base + k, zi, zo, o);
                    if (K == 1 || rep == 0) {                          // (the partial tile is replica 0's)
#pragma unroll
                        for (int j = 0; j < NOUT; j++) reinterpret_cast<float *>(&to[j][(k >> 2) * 64])[k & 3] = zo[j];
                    }
                    if (WR) zrow_store<1>(zrow_rsrc(out, ostride, base + k), voff, 0, o);
                }
            }
#pragma unroll
            for (int j = 0; j < NIN; j++) bi[j] = bi[j] + 1 == tin[j].depth ? 0 : bi[j] + 1;
#pragma unroll
            for (int j = 0; j < NOUT; j++) bo[j] = bo[j] + 1 == tout[j].depth ? 0 : bo[j] + 1;
        }
#if defined(ZS_PC_TRACE)
        zs_busy += __builtin_readcyclecounter() - zs_t0;
#endif
        __syncthreads();
    }
#if defined(ZS_PC_TRACE)
    if (blockIdx.x == 0 && lane == 0)
        printf("role wave %u (lag %u, rep %u of %d%s): busy %llu of %llu cycles\n", (unsigned)(threadIdx.x >> 6), lag, rep, K, WR ? ", writer" : "", zs_busy,
               (unsigned long long)(__builtin_readcyclecounter() - zs_t_begin));
#endif
}

// A loader role: NR image rows per frame -- input images of the script module's params, or the live output image -- into
// their tiles, one tile ahead of the step that publishes it: source j publishes tile d in step d + lag[j] (one step before its first
// reader) and requests tile d + 1 in that same step.  A row source with stride 0 (a constant's dummy row, a zeroed output) is
// neither loaded nor written: its readers select the constant / start from 0 and discard what the tile holds.
template <int CH_, int NR>
__device__ __forceinline__ void zs_loader_run(float4 *lds, uint32_t lane, uint32_t steps, uint32_t start, uint32_t n_frames, const ZsTileRef *tout,
                                              const float *const *src, const size_t *stride, const uint32_t *voff, const uint32_t *lag) {
    constexpr uint32_t CH = CH_, Q = CH / 4, TILE = Q * 64;
    const uint32_t nchunks = (n_frames + CH - 1) / CH;
    float r[NR][CH];
    uint32_t bo[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) bo[j] = 0;
    auto request = [&](int j, uint32_t d) ZH_INLINE_LAMBDA {
        if (stride[j] == 0) return;                                   // (kernel-uniform)
        const uint32_t nf = min(CH, n_frames - d * CH), base = start + d * CH;
        const zh_rsrc_t ri = zrow_rsrc(src[j], stride[j], base);
        const uint32_t irow = (uint32_t)stride[j] * 4u;
        if (nf == CH) {
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) r[j][k] = zrow_load<1>(ri, voff[j], k * irow);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) r[j][k] = k < nf ? zrow_load<1>(ri, voff[j], k * irow) : 0.0f;
        }
    };
#pragma unroll
    for (int j = 0; j < NR; j++) if (lag[j] == 0 && nchunks > 0) request(j, 0);
    for (uint32_t c = 0; c < steps; c++) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            if (c >= lag[j] && c - lag[j] < nchunks && stride[j] != 0) {
                float4 *t = lds + tout[j].off + bo[j] * TILE + lane;
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) t[q * 64] = make_float4(r[j][4 * q], r[j][4 * q + 1], r[j][4 * q + 2], r[j][4 * q + 3]);
                bo[j] = bo[j] + 1 == tout[j].depth ? 0 : bo[j] + 1;
            }
            if (c + 1 >= lag[j] && c + 1 - lag[j] < nchunks) request(j, c + 1 - lag[j]);
        }
        __syncthreads();
    }
}
#endif
