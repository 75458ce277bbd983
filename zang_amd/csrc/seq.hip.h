// seq.hip.h -- the lane-per-voice sequential frame loop shared by the stateful paint kernels.
//
// One lane owns one voice and walks the span frame by frame (the reference's scalar loop
// with its loop-carried state, e.g. Filter.zig:124-147), state in VGPRs.  Images are
// [frame][voice], so the 64 lanes of a wave touch 256 contiguous bytes per frame.
// Frames are processed in chunks of CH: the loads of chunk k+1 (the `+=` read of the
// output and up to NIN input images) are issued before chunk k is computed, so HBM latency
// hides behind the dependent ALU chain instead of adding to it.
#pragma once
#include "common.hip.h"
#include "lanes.hip.h"

// f(frame, x[NIN], value&) -> painted.  Called for consecutive frames in order; it may
// carry state by reference capture.  painted == false leaves out[frame] untouched (ADD) or
// zero (ZERO_FIRST), like a reference loop that `continue`s or stops early.
// `out` / `in` are the images' (wave-uniform) base pointers and `v` the lane's voice: every access is
// row[v] with a uniform row pointer, so the row address lives in SGPRs (scalar adds per frame) and the
// lane offset is one constant VGPR -- no per-frame 64-bit vector address arithmetic.
// W = voices per lane (lanes.hip.h): with W = 2, `v` is the lane's first voice (even, 8-byte aligned),
// values are zf2 and `painted` is a per-voice mask.
template <int CH, bool ZF, int NIN, int W = 1, class F>
__device__ __forceinline__ void frame_loop(float *__restrict__ out, uint32_t v, size_t ostride,
                                           const float *const *in, const size_t *istride,
                                           uint32_t start, uint32_t end, F &&f) {
    using T = typename LaneT<W>::F;
    constexpr int NI = NIN > 0 ? NIN : 1;
    const uint32_t n = end - start;
    const uint32_t nfull = n / CH;
    const uint32_t voff = v * 4u;                                   // the lane's byte offset inside a row
    const uint32_t orow = (uint32_t)ostride * 4u;                   // bytes per row (lanes.hip.h: < 4 GiB / CH)
    uint32_t irow[NI];
#pragma unroll
    for (int j = 0; j < NIN; j++) irow[j] = (uint32_t)istride[j] * 4u;
    T oc[CH], xc[NI][CH];
    uint32_t i = start;
    if (nfull > 0) {
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (!ZF) oc[k] = zrow_load<W>(ro, voff, k * orow);
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const zh_rsrc_t ri = zrow_rsrc(in[j], istride[j], i);
#pragma unroll
            for (int k = 0; k < CH; k++) xc[j][k] = zrow_load<W>(ri, voff, k * irow[j]);
        }
    }
    for (uint32_t c = 0; c < nfull; c++, i += CH) {
        T on[CH], xn[NI][CH];
        const bool more = c + 1 < nfull;
        if (more) {
            const zh_rsrc_t rn = zrow_rsrc(out, ostride, i + CH);
#pragma unroll
            for (int k = 0; k < CH; k++)
                if (!ZF) on[k] = zrow_load<W>(rn, voff, k * orow);
#pragma unroll
            for (int j = 0; j < NIN; j++) {
                const zh_rsrc_t ri = zrow_rsrc(in[j], istride[j], i + CH);
#pragma unroll
                for (int k = 0; k < CH; k++) xn[j][k] = zrow_load<W>(ri, voff, k * irow[j]);
            }
        }
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        // the chunk's CH frames are computed first and stored afterwards: a store through the descriptor is an
        // opaque memory write to the compiler, and with one after every frame no load of the body (Sampler's PCM
        // gathers, a Curve's nodes) could be issued ahead of the previous frame's store -- every frame paid its
        // full load latency
        T res[CH];
        typename LaneT<W>::M pm[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            T x[NI];
#pragma unroll
            for (int j = 0; j < NIN; j++) x[j] = xc[j][k];
            T val = zsplat<T>(0.0f);
            pm[k] = f(i + k, x, val);
            const T o = ZF ? zsplat<T>(0.0f) : oc[k];
            res[k] = zsel(pm[k], o + val, o);
        }
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (ZF || zany(pm[k])) zrow_store<W>(ro, voff, k * orow, res[k]);
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if (!ZF) oc[k] = on[k];
#pragma unroll
                for (int j = 0; j < NIN; j++) xc[j][k] = xn[j][k];
            }
        }
    }
    for (; i < end; i++) {
        T x[NI];
#pragma unroll
        for (int j = 0; j < NIN; j++) x[j] = zrow_load<W>(zrow_rsrc(in[j], istride[j], i), voff, 0);
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        T val = zsplat<T>(0.0f);
        const auto painted = f(i, x, val);
        T o = ZF ? zsplat<T>(0.0f) : zrow_load<W>(ro, voff, 0);
        o = zsel(painted, o + val, o);
        if (ZF || zany(painted)) zrow_store<W>(ro, voff, 0, o);
    }
}

// frame_loop for a generator (no input images) whose per-frame body has a rare event -- a curve span or an envelope stage
// ending -- that costs a compare, an exec-mask region and a branch in EVERY frame of an unrolled chunk.  `quiet(i)` is
// asked once per chunk of CH frames and must be wave-uniform: true promises that no lane has the event in frames
// [i, i + CH), and the chunk then runs `fast` (the body without the test); otherwise, and for the tail, `slow`.
// Both are f(frame, value&) -> painted, with the same results wherever `quiet` holds.
template <int CH, bool ZF, class Q, class FF, class FS>
__device__ __forceinline__ void frame_loop_gen(float *__restrict__ out, uint32_t v, size_t ostride, uint32_t start, uint32_t end,
                                               Q &&quiet, FF &&fast, FS &&slow) {
    const uint32_t n = end - start;
    const uint32_t nfull = n / CH;
    const uint32_t voff = v * 4u;
    const uint32_t orow = (uint32_t)ostride * 4u;
    float oc[CH];
    uint32_t i = start;
    if (nfull > 0 && !ZF) {
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
#pragma unroll
        for (int k = 0; k < CH; k++) oc[k] = zrow_load<1>(ro, voff, k * orow);
    }
    for (uint32_t c = 0; c < nfull; c++, i += CH) {
        float on[CH];
        const bool more = c + 1 < nfull;
        if (more && !ZF) {
            const zh_rsrc_t rn = zrow_rsrc(out, ostride, i + CH);
#pragma unroll
            for (int k = 0; k < CH; k++) on[k] = zrow_load<1>(rn, voff, k * orow);
        }
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        float res[CH];
        bool pm[CH];
        if (quiet(i)) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float val = 0.0f;
                pm[k] = fast(i + k, val);
                const float o = ZF ? 0.0f : oc[k];
                res[k] = pm[k] ? o + val : o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float val = 0.0f;
                pm[k] = slow(i + k, val);
                const float o = ZF ? 0.0f : oc[k];
                res[k] = pm[k] ? o + val : o;
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (ZF || zany(pm[k])) zrow_store<1>(ro, voff, k * orow, res[k]);
        if (more && !ZF) {
#pragma unroll
            for (int k = 0; k < CH; k++) oc[k] = on[k];
        }
    }
    for (; i < end; i++) {
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        float val = 0.0f;
        const bool painted = slow(i, val);
        float o = ZF ? 0.0f : zrow_load<1>(ro, voff, 0);
        o = painted ? o + val : o;
        if (ZF || zany(painted)) zrow_store<1>(ro, voff, 0, o);
    }
}

// frame_loop_gen with TWO fast bodies: `quiet(i)` returns 0 (the chunk runs `slow`), 1 (`fast1`) or 2 (`fast2`), wave-uniform.
template <int CH, bool ZF, class Q, class F1, class F2, class FS>
__device__ __forceinline__ void frame_loop_gen2(float *__restrict__ out, uint32_t v, size_t ostride, uint32_t start, uint32_t end,
                                                Q &&quiet, F1 &&fast1, F2 &&fast2, FS &&slow) {
    const uint32_t n = end - start;
    const uint32_t nfull = n / CH;
    const uint32_t voff = v * 4u;
    const uint32_t orow = (uint32_t)ostride * 4u;
    float oc[CH];
    uint32_t i = start;
    if (nfull > 0 && !ZF) {
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
#pragma unroll
        for (int k = 0; k < CH; k++) oc[k] = zrow_load<1>(ro, voff, k * orow);
    }
    for (uint32_t c = 0; c < nfull; c++, i += CH) {
        float on[CH];
        const bool more = c + 1 < nfull;
        if (more && !ZF) {
            const zh_rsrc_t rn = zrow_rsrc(out, ostride, i + CH);
#pragma unroll
            for (int k = 0; k < CH; k++) on[k] = zrow_load<1>(rn, voff, k * orow);
        }
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        float res[CH];
        bool pm[CH];
        const int q = quiet(i);
        if (q == 2) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float val = 0.0f;
                pm[k] = fast2(i + k, val);
                const float o = ZF ? 0.0f : oc[k];
                res[k] = pm[k] ? o + val : o;
            }
        } else if (q == 1) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float val = 0.0f;
                pm[k] = fast1(i + k, val);
                const float o = ZF ? 0.0f : oc[k];
                res[k] = pm[k] ? o + val : o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float val = 0.0f;
                pm[k] = slow(i + k, val);
                const float o = ZF ? 0.0f : oc[k];
                res[k] = pm[k] ? o + val : o;
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (ZF || zany(pm[k])) zrow_store<1>(ro, voff, k * orow, res[k]);
        if (more && !ZF) {
#pragma unroll
            for (int k = 0; k < CH; k++) oc[k] = on[k];
        }
    }
    for (; i < end; i++) {
        const zh_rsrc_t ro = zrow_rsrc(out, ostride, i);
        float val = 0.0f;
        const bool painted = slow(i, val);
        float o = ZF ? 0.0f : zrow_load<1>(ro, voff, 0);
        o = painted ? o + val : o;
        if (ZF || zany(painted)) zrow_store<1>(ro, voff, 0, o);
    }
}

// A frame-range kernel's replay over an input image: consume(x) for rows 0 .. count-1 of one voice column, in order.  Two
// batches of 16 rows: one is consumed while the next is in flight (two NAMED arrays, ping-pong by two -- a rotation over an
// array of arrays ended up in scratch memory and was slower than no overlap at all).
template <class C>
__device__ __forceinline__ void replay_rows(const float *__restrict__ fp, size_t stride, uint32_t count, C &&consume) {
    constexpr int B = 16;
    const size_t stb = B * stride;
    uint32_t i = 0;
    if (count >= (uint32_t)B) {
        float xa[B], xb[B];
#pragma unroll
        for (int k = 0; k < B; k++) xa[k] = fp[(size_t)k * stride];
        fp += stb; i = B;
        for (; i + 2 * B <= count; i += 2 * B, fp += 2 * stb) {
#pragma unroll
            for (int k = 0; k < B; k++) xb[k] = fp[(size_t)k * stride];
#pragma unroll
            for (int k = 0; k < B; k++) consume(xa[k]);
#pragma unroll
            for (int k = 0; k < B; k++) xa[k] = (fp + stb)[(size_t)k * stride];
#pragma unroll
            for (int k = 0; k < B; k++) consume(xb[k]);
        }
        if (i + B <= count) {
#pragma unroll
            for (int k = 0; k < B; k++) xb[k] = fp[(size_t)k * stride];
#pragma unroll
            for (int k = 0; k < B; k++) consume(xa[k]);
#pragma unroll
            for (int k = 0; k < B; k++) consume(xb[k]);
            fp += stb; i += B;
        } else {
#pragma unroll
            for (int k = 0; k < B; k++) consume(xa[k]);
        }
    }
    for (; i < count; i++, fp += stride) consume(*fp);
}

// zero the span of one voice column (used when a ZERO_FIRST paint paints nothing)
__device__ __forceinline__ void zero_column(float *__restrict__ out, size_t ostride, uint32_t start, uint32_t end) {
    for (uint32_t i = start; i < end; i++) out[(size_t)i * ostride] = 0.0f;
}
