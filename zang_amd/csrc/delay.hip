// delay.hip -- zang.Delay(n) (src/zang/delay.zig) on the device, and the two modules built on it:
// SimpleDelay and FilteredEchoes (examples/modules.zig:341-461).
//
// The rings of n voices are one HBM image [delay_sample][voice] (voice fastest), so the 64 lanes of
// a wave read/write 256 contiguous bytes per frame like every other image.  The reference moves
// data in chunks of <= delay_samples: read the ring (delay.zig:28-57), do the work, write the ring
// (:62-89); a slot is always read before it is rewritten, so walking the span sample by sample
// (read slot, compute, write slot, advance index modulo delay_samples) gives the same values.
#include "common.hip.h"
#include "zmath.hip.h"
#include "dsp.hip.h"
#include "filter_tp.hip.h"
#include <vector>

struct DelayState {
    float *ring;          // [delay_samples][n]
    uint32_t *index;      // [n]  (delay_buffer_index; equal for all voices unless set_state says otherwise)
    uint32_t n, delay_samples;
};

struct zh_delay { zh_ctx *ctx; DelayState d; };
struct zh_filtered_echoes { zh_ctx *ctx; DelayState d; float *l, *b;
                            float *tp; };   // ZH_PAINT_TOLERANT scratch (kFeTpFloats per voice), allocated by the first tolerant paint outside a capture

// The reference moves data in chunks of <= delay_samples frames: read the ring for the whole chunk, then write it
// (delay.zig:28-89).  Within such a chunk every frame touches a different ring slot, each last written at least
// delay_samples frames ago, so the chunk's loads can all be issued before its stores -- which is what lets them
// overlap: frame by frame (read slot, write slot, next frame) every frame waited out its own load latency
// (SimpleDelay(300), 131,072 voices: 1277 us per 1024 frames).  CH = frames per chunk (8); delays shorter than that,
// and an input image that overlaps the output image (then the per-frame order of reads and writes is observable),
// take the frame-by-frame form (CH = 1).
template <uint32_t CH>
__device__ __forceinline__ uint32_t delay_next(uint32_t idx, uint32_t delay_samples) { return idx + 1 == delay_samples ? 0 : idx + 1; }   // delay.zig:84-87

// SimpleDelay.paint, examples/modules.zig:363-385
template <bool ZF, uint32_t CH>
__global__ void __launch_bounds__(kSeqBlock) k_simple_delay(DelayState d, Img out, CImg input, uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= d.n) return;
    uint32_t idx = d.index[v];
    float *ring = d.ring + v;
    float *o = out.at(start, v);
    const float *in = input.at(start, v);
    uint32_t i = start;
    if constexpr (CH > 1)                                             // CH == 1: only the frame-by-frame loop below, in the reference's order
    for (; i + CH <= end; i += CH, o += CH * out.stride, in += CH * input.stride) {
        float *slot[CH];
        float delayed[CH], x[CH], old[CH];
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) {
            slot[k] = ring + (size_t)idx * d.n;
            idx = delay_next<CH>(idx, d.delay_samples);
            delayed[k] = *slot[k];                                    // readDelayBuffer: out += ring
            x[k] = in[(size_t)k * input.stride];
            old[k] = ZF ? 0.0f : o[(size_t)k * out.stride];
        }
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) {
            store_row(o + (size_t)k * out.stride, old[k] + delayed[k]);
            *slot[k] = x[k];                                          // writeDelayBuffer: ring = input
        }
    }
    for (; i < end; i++, o += out.stride, in += input.stride) {
        float *slot = ring + (size_t)idx * d.n;
        const float delayed = *slot;
        store_row(o, (ZF ? 0.0f : *o) + delayed);
        *slot = *in;
        idx = delay_next<CH>(idx, d.delay_samples);
    }
    d.index[v] = idx;
}

// SimpleDelay has no feedback: a frame's output is the ring slot it reads (the input of delay_samples frames ago) and the ring
// afterwards holds the span's last delay_samples inputs, so the frames of a span are independent.  Few voices (a lone
// wave per 64 voices walks 1,024 frames at ~10 issue slots each): the span as a grid of (64 voices x 32 frames) pieces --
// frame j reads slot (index + j) mod n, which holds the old ring for j < delay_samples and input[j - delay_samples] after
// that (read straight from the input image).  Then the ring is brought up to date from the input image and the indices
// advance -- in kernels of their own, stream-ordered after every piece has read the old ring and the old index.
// WRITE: the span is no longer than the delay, so every slot is touched by exactly one frame and that frame stores its input
// right after reading (no k_delay_store launch).
template <bool ZF, bool WRITE>
__global__ void __launch_bounds__(256) k_delay_frames(DelayState d, Img out, CImg input, uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t piece = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= d.n) return;
    const uint32_t j0 = piece * 32, n = end - start;
    if (j0 >= n) return;
    const uint32_t j1 = min(j0 + 32, n), D = d.delay_samples;
    uint32_t slot = (uint32_t)(((uint64_t)d.index[v] + j0) % D);
    float *o = out.at(start + j0, v);
    const float *in = input.at(start + j0, v);
    uint32_t j = j0;
    for (; j + 8 <= j1; j += 8, o += 8 * (size_t)out.stride, in += 8 * (size_t)input.stride) {   // 8 frames' loads ahead of their stores
        float *rs[8];
        float delayed[8], base[8], x[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            rs[k] = d.ring + (size_t)slot * d.n + v;
            delayed[k] = j + k < D ? *rs[k] : *(in + (size_t)k * input.stride - (size_t)D * input.stride);   // readDelayBuffer (delay.zig:28-57)
            base[k] = ZF ? 0.0f : o[(size_t)k * out.stride];
            if (WRITE) x[k] = in[(size_t)k * input.stride];
            slot = slot + 1 == D ? 0 : slot + 1;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            store_row(o + (size_t)k * out.stride, base[k] + delayed[k]);
            if (WRITE) *rs[k] = x[k];                                 // writeDelayBuffer (:62-89)
        }
    }
    for (; j < j1; j++, o += out.stride, in += input.stride) {
        float *rs = d.ring + (size_t)slot * d.n + v;
        const float delayed = j < D ? *rs : *(in - (size_t)D * input.stride);
        store_row(o, (ZF ? 0.0f : *o) + delayed);
        if (WRITE) *rs = *in;
        slot = slot + 1 == D ? 0 : slot + 1;
    }
}
// the ring after the span: its last min(n, delay_samples) inputs, at the slots the frame walk would have left them in
__global__ void __launch_bounds__(256) k_delay_store(DelayState d, CImg input, uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t piece = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (v >= d.n) return;
    const uint32_t n = end - start, D = d.delay_samples, first = n > D ? n - D : 0;
    const uint32_t j0 = first + piece * 32;
    if (j0 >= n) return;
    const uint32_t j1 = min(j0 + 32, n);
    uint32_t slot = (uint32_t)(((uint64_t)d.index[v] + j0) % D);
    const float *in = input.at(start + j0, v);
    uint32_t j = j0;
    for (; j + 8 <= j1; j += 8, in += 8 * (size_t)input.stride) {
        float x[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) x[k] = in[(size_t)k * input.stride];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            d.ring[(size_t)slot * d.n + v] = x[k];
            slot = slot + 1 == D ? 0 : slot + 1;
        }
    }
    for (; j < j1; j++, in += input.stride) {
        d.ring[(size_t)slot * d.n + v] = *in;
        slot = slot + 1 == D ? 0 : slot + 1;
    }
}
__global__ void __launch_bounds__(256) k_delay_advance(DelayState d, uint32_t frames) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v < d.n) d.index[v] = (uint32_t)(((uint64_t)d.index[v] + frames) % d.delay_samples);
}

// FilteredEchoes.paint, examples/modules.zig:411-460
template <bool ZF, uint32_t CH>
__global__ void __launch_bounds__(kSeqBlock) k_filtered_echoes(DelayState d, float *__restrict__ l_io, float *__restrict__ b_io,
                                                               Img out, CImg input, uint32_t start, uint32_t end,
                                                               F32P feedback_p, F32P cutoff_p) {
    const uint32_t v = blockIdx.x * kSeqBlock + threadIdx.x;
    if (v >= d.n) return;
    const float feedback = feedback_p.get(v);
    const float cut = zclampf(cutoff_p.get(v), 0.0f, 1.0f);           // Filter.zig:114
    const float res = 1.0f - zclampf(0.0f, 0.0f, 1.0f);               // res = constant(0.0) (:441) -> Filter.zig:118
    uint32_t idx = d.index[v];
    float l = l_io[v], b = b_io[v];
    float *ring = d.ring + v;
    float *o = out.at(start, v);
    const float *in = input.at(start, v);
    auto one = [&](float delayed, float x) ZH_INLINE_LAMBDA {
        float t0 = 0.0f + delayed;                                    // zero(temp0); readDelayBuffer (:425-428)
        t0 = t0 * feedback;                                           // multiplyWithScalar (:433)
        t0 = t0 + x;                                                  // addInto (:436)
        const SvfOut s = svf_step(l, b, t0, cut, res);                // Filter.paint low_pass (Filter.zig:135-146)
        return svf_lowpass_into_zero(s.l, s.b);                       // zero(temp1); += (:439)
    };
    uint32_t i = start;
    if constexpr (CH > 1)                                             // CH == 1: only the frame-by-frame loop below, in the reference's order
    for (; i + CH <= end; i += CH, o += CH * out.stride, in += CH * input.stride) {
        float *slot[CH];
        float delayed[CH], x[CH], old[CH], t1[CH];
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) {
            slot[k] = ring + (size_t)idx * d.n;
            idx = delay_next<CH>(idx, d.delay_samples);
            delayed[k] = *slot[k];
            x[k] = in[(size_t)k * input.stride];
            old[k] = ZF ? 0.0f : o[(size_t)k * out.stride];
        }
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) t1[k] = one(delayed[k], x[k]);
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) {
            store_row(o + (size_t)k * out.stride, old[k] + t1[k]);      // addInto(output, temp1) (:448)
            *slot[k] = t1[k];                                         // writeDelayBuffer(temp1) (:452)
        }
    }
    for (; i < end; i++, o += out.stride, in += input.stride) {
        float *slot = ring + (size_t)idx * d.n;
        const float t1 = one(*slot, *in);
        store_row(o, (ZF ? 0.0f : *o) + t1);
        *slot = t1;
        idx = delay_next<CH>(idx, d.delay_samples);
    }
    d.index[v] = idx;
    l_io[v] = l; b_io[v] = b;
}

// FilteredEchoes at a small voice count, delay >= 192 frames: three waves per 64 voices (the form of modules.hip's
// k_filter_pc: float4 LDS tiles, per-role step loops).  Wave 0 requests a tile's ring slots and input rows two tiles ahead and
// forms the filter's input (in = ((0 + delayed) * feedback + x) + fcdcoffset); wave 1 runs the state-variable recurrence alone
// (fetching its next tile from LDS meanwhile) and hands (l, b) on; wave 2 forms temp1, does the `+=` into the output image
// and writes temp1 into the ring, three tiles behind the loader.  A slot is read delay_samples frames after it was written:
// when the loader requests tile c + 2 the writer is certain to have stored tile c - 4, six tiles = 192 frames earlier --
// hence the minimum delay (with it a slot being read is also never one the writer is filling in the same step).  Same
// operations on the same values as k_filtered_echoes => same bits.
template <bool ZF>
__global__ void __launch_bounds__(192) k_filtered_echoes_pc(DelayState d, float *__restrict__ l_io, float *__restrict__ b_io, Img out, CImg input,
                                                            uint32_t start, uint32_t end, F32P feedback_p, F32P cutoff_p) {
    constexpr uint32_t CH = 32, Q = CH / 4;
    __shared__ float4 in_q[2][Q][64], l_q[2][Q][64], b_q[2][Q][64];
    const uint32_t lane = threadIdx.x & 63, role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 loader, 1 filter, 2 writer
    const uint32_t v = blockIdx.x * 64 + lane;
    const bool live = v < d.n;
    const uint32_t vc = live ? v : d.n - 1;
    const uint32_t n = end - start, nchunks = (n + CH - 1) / CH, D = d.delay_samples;
    const float feedback = feedback_p.get(vc);
    const float cut = zclampf(cutoff_p.get(vc), 0.0f, 1.0f);          // Filter.zig:114
    const float res = 1.0f - zclampf(0.0f, 0.0f, 1.0f);               // res = constant(0.0) (:441) -> Filter.zig:118
    const uint32_t idx0 = d.index[vc];
    float *ring = d.ring + vc;
    auto frames = [&](uint32_t c) ZH_INLINE_LAMBDA { return c < nchunks ? min(CH, n - c * CH) : 0u; };
    auto slot_of = [&](uint32_t j) ZH_INLINE_LAMBDA { return (uint32_t)(((uint64_t)idx0 + j) % D); };   // the slot frame j of the span uses
    // frame k of this lane inside a tile of float4 (the scalar path of a partial last tile)
    auto at = [&](float4 (*t)[64], uint32_t k) ZH_INLINE_LAMBDA -> float & { return reinterpret_cast<float *>(&t[k >> 2][lane])[k & 3]; };
    // The voices of a wave normally share one ring index (set_state can make them differ): then a tile whose 32 slots do not
    // wrap is 32 consecutive ring rows, addressed like image rows (one descriptor, scalar row offsets) instead of a 64-bit
    // multiply-add and a wrap test per lane and frame.
    const bool uni = __builtin_amdgcn_ballot_w64(idx0 != (uint32_t)__builtin_amdgcn_readfirstlane((int)idx0)) == 0;
    const uint32_t voff = vc * 4u, rrow = d.n * 4u, irow = (uint32_t)input.stride * 4u, orow = (uint32_t)out.stride * 4u;
    auto rows_ok = [&](uint32_t c) ZH_INLINE_LAMBDA {                   // wave-uniform: tile c's slots are consecutive rows for every lane
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_of(c * CH));
        return uni && s0 + CH <= D;
    };
    // Step c: the loader publishes tile c (requested two steps earlier); the filter wave computes tile c - 2 out of registers
    // while it fetches tile c - 1 from LDS; the writer finishes and stores tile c - 3.  Two steps per iteration where arrays
    // alternate between steps, so that they keep their registers.
    const uint32_t last = nchunks + 2;
    if (role == 0) {
        float da[CH], xa[CH], db[CH], xb[CH];                         // delayed samples and input rows of the tiles c (even / odd), then c + 2
        auto request = [&](uint32_t c, float (&dn)[CH], float (&xn)[CH]) ZH_INLINE_LAMBDA {
            if (frames(c) != CH) return;
            const zh_rsrc_t ri = zrow_rsrc(input.p, input.stride, start + c * CH);
            if (rows_ok(c)) {
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_of(c * CH));
                const zh_rsrc_t rr = zrow_rsrc(d.ring, d.n, s0);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) { dn[k] = zrow_load<1>(rr, voff, k * rrow); xn[k] = zrow_load<1>(ri, voff, k * irow); }
            } else {
                uint32_t sl = slot_of(c * CH);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) {
                    dn[k] = ring[(size_t)sl * d.n];
                    xn[k] = zrow_load<1>(ri, voff, k * irow);
                    sl = sl + 1 == D ? 0 : sl + 1;
                }
            }
        };
        auto in_of = [&](float delayed, float x) ZH_INLINE_LAMBDA {
            float t0 = 0.0f + delayed;                                // zero(temp0); readDelayBuffer (:425-428)
            t0 = t0 * feedback;                                       // multiplyWithScalar (:433)
            t0 = t0 + x;                                              // addInto (:436)
            return t0 + kSvfDcOffset;                                 // Filter.zig:135
        };
        auto publish = [&](uint32_t c, float (&dn)[CH], float (&xn)[CH]) ZH_INLINE_LAMBDA {
            const uint32_t nf = frames(c);
            float4 (*t)[64] = in_q[c & 1];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++)
                    t[q][lane] = make_float4(in_of(dn[4 * q], xn[4 * q]), in_of(dn[4 * q + 1], xn[4 * q + 1]), in_of(dn[4 * q + 2], xn[4 * q + 2]), in_of(dn[4 * q + 3], xn[4 * q + 3]));
            } else {
                // (a partial tile is the last one: the writer has stored every tile up to c - 4 and the slots of this one were
                // written at least 192 frames before its first frame)
                uint32_t sl = slot_of(c * CH);
                const float *ip = input.at(start + c * CH, vc);
                for (uint32_t k = 0; k < nf; k++) {
                    at(t, k) = in_of(ring[(size_t)sl * d.n], ip[(size_t)k * input.stride]);
                    sl = sl + 1 == D ? 0 : sl + 1;
                }
            }
            request(c + 2, dn, xn);
        };
        request(0, da, xa); request(1, db, xb);
        for (uint32_t c = 0; c <= last; c += 2) {
            if (c < nchunks) publish(c, da, xa);
            __syncthreads();
            if (c + 1 <= last) {
                if (c + 1 < nchunks) publish(c + 1, db, xb);
                __syncthreads();
            }
        }
        if (live) d.index[v] = slot_of(n);
    } else if (role == 1) {
        float l = l_io[vc], b = b_io[vc];
        float4 fa[Q], fb[Q];                                          // the tile in hand / the next one
#pragma unroll
        for (uint32_t q = 0; q < Q; q++) fa[q] = fb[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto step = [&](uint32_t c, float4 (&cur)[Q], float4 (&nxt)[Q]) ZH_INLINE_LAMBDA {
            if (c == 0 || c > nchunks + 1) return;
            const float4 (*tn)[64] = in_q[(c - 1) & 1];              // (complete only if that tile is a whole one: otherwise unused)
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) nxt[q] = tn[q][lane];
            if (c == 1) return;
            const uint32_t dd = c - 2, nf = frames(dd);
            float4 (*tl)[64] = l_q[dd & 1], (*tb)[64] = b_q[dd & 1];
            if (nf == CH) {
#pragma unroll
                for (uint32_t q = 0; q < Q; q++) {
                    const SvfOut s0 = svf_core(l, b, cur[q].x, cut, res);
                    const SvfOut s1 = svf_core(l, b, cur[q].y, cut, res);
                    const SvfOut s2 = svf_core(l, b, cur[q].z, cut, res);
                    const SvfOut s3 = svf_core(l, b, cur[q].w, cut, res);
                    tl[q][lane] = make_float4(s0.l, s1.l, s2.l, s3.l);     // (h is not needed: dsp.hip.h svf_lowpass_into_zero)
                    tb[q][lane] = make_float4(s0.b, s1.b, s2.b, s3.b);
                }
            } else {                                                  // (the last tile: the loader has stopped, its buffer stays)
                float4 (*ti)[64] = in_q[dd & 1];
                for (uint32_t k = 0; k < nf; k++) {
                    const SvfOut sv = svf_core(l, b, at(ti, k), cut, res);
                    at(tl, k) = sv.l; at(tb, k) = sv.b;
                }
            }
        };
        for (uint32_t c = 0; c <= last; c += 2) {
            step(c, fa, fb);
            __syncthreads();
            if (c + 1 <= last) {
                step(c + 1, fb, fa);
                __syncthreads();
            }
        }
        if (live) { l_io[v] = l; b_io[v] = b; }
    } else {
        float bn[CH];                                                 // the output rows of the tile after the one in hand
        for (uint32_t c = 0; c <= last; c++) {
            if (c > 2) {
                const uint32_t dd = c - 3, nf = frames(dd);
                float4 (*tl)[64] = l_q[dd & 1], (*tb)[64] = b_q[dd & 1];
                float *op = out.at(start + dd * CH, vc);
                uint32_t sl = slot_of(dd * CH);
                auto one = [&](uint32_t k, float fl, float fb, float base) ZH_INLINE_LAMBDA {
                    const float t1 = svf_lowpass_into_zero(fl, fb);   // zero(temp1); += low-pass (:439, Filter.zig:146)
                    if (live) {
                        op[(size_t)k * out.stride] = base + t1;       // addInto(output, temp1) (:448)
                        ring[(size_t)sl * d.n] = t1;                  // writeDelayBuffer(temp1) (:452)
                    }
                    sl = sl + 1 == D ? 0 : sl + 1;
                };
                if (nf == CH) {
                    float4 xl[Q], xb[Q];
#pragma unroll
                    for (uint32_t q = 0; q < Q; q++) { xl[q] = tl[q][lane]; xb[q] = tb[q][lane]; }
                    if (rows_ok(dd)) {
                        const zh_rsrc_t ro = zrow_rsrc(out.p, out.stride, start + dd * CH);
                        const zh_rsrc_t rr = zrow_rsrc(d.ring, d.n, (uint32_t)__builtin_amdgcn_readfirstlane((int)sl));
                        auto fast = [&](uint32_t k, float fl, float fb) ZH_INLINE_LAMBDA {
                            const float t1 = svf_lowpass_into_zero(fl, fb);         // as in one()
                            if (live) {
                                zrow_store<1>(ro, voff, k * orow, (ZF ? 0.0f : bn[k]) + t1);
                                zrow_store<1>(rr, voff, k * rrow, t1);
                            }
                        };
#pragma unroll
                        for (uint32_t q = 0; q < Q; q++) {
                            fast(4 * q, xl[q].x, xb[q].x); fast(4 * q + 1, xl[q].y, xb[q].y);
                            fast(4 * q + 2, xl[q].z, xb[q].z); fast(4 * q + 3, xl[q].w, xb[q].w);
                        }
                    } else {
#pragma unroll
                        for (uint32_t q = 0; q < Q; q++) {
                            one(4 * q, xl[q].x, xb[q].x, ZF ? 0.0f : bn[4 * q]); one(4 * q + 1, xl[q].y, xb[q].y, ZF ? 0.0f : bn[4 * q + 1]);
                            one(4 * q + 2, xl[q].z, xb[q].z, ZF ? 0.0f : bn[4 * q + 2]); one(4 * q + 3, xl[q].w, xb[q].w, ZF ? 0.0f : bn[4 * q + 3]);
                        }
                    }
                } else {
                    for (uint32_t k = 0; k < nf; k++) one(k, at(tl, k), at(tb, k), ZF ? 0.0f : op[(size_t)k * out.stride]);
                }
            }
            if (!ZF && c >= 2 && frames(c - 2) == CH) {               // the output rows of the tile written at the next step
                const zh_rsrc_t rn = zrow_rsrc(out.p, out.stride, start + (c - 2) * CH);
#pragma unroll
                for (uint32_t k = 0; k < CH; k++) bn[k] = zrow_load<1>(rn, voff, k * orow);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------- FilteredEchoes, tolerant
// ZH_PAINT_TOLERANT at few voices (filter_tp.hip.h).  Over a piece of at most delay_samples frames every frame reads a ring slot
// that was written before the piece began (delay.zig:28-89: a slot is read delay_samples frames after it was written), so the
// filter's input ((0 + ring) * feedback + x, examples/modules.zig:425-436) is known for the whole piece up front and the
// piece is an ordinary constant-parameter Filter paint: chunks at once from a zero state (pass A), the scan of the chunk-start
// states, the reference's own recurrence again from them (pass B), which also does the `+=` and writes temp1 into the ring
// (:448-452).  A span longer than the delay is painted as successive pieces (a launch pair each).  Inside a chunk the
// arithmetic is the reference's; what is tolerant is each chunk's start state, as for the Filter module.
// Scratch per voice: (kTpMaxChunks + 1) float2 -- slot 0 = the filter state at the piece's start, slot j + 1 = e_j -- and the
// ring index at the piece's start (pass B's last chunk moves the module's index while other chunks still need the old one).
constexpr size_t kFeTpFloats = (size_t)(kTpMaxChunks + 1) * 2 + 1;
struct FeTpArgs {
    DelayState d;
    float *l, *b;
    float2 *e;
    uint32_t *idx0;
    uint32_t start, end, L;
    Img out;
    CImg input;
    F32P feedback, cutoff;
};
// body(f, k, delayed, x, base, ro, rr, rows) for the frames [f0, f1) of one lane, 8-frame tiles with the tile's loads ahead of
// its use.  rows: the wave's voices share one ring index and the tile's eight slots do not wrap -> ring rows addressed like
// image rows (descriptor rr, row k); otherwise per-lane slot arithmetic (slot_ptr).
template <bool BASE, class Body>
__device__ __forceinline__ void fe_tp_tiles(const FeTpArgs &a, uint32_t v, uint32_t idx0, uint32_t f0, uint32_t f1, Body body) {
    const uint32_t D = a.d.delay_samples, voff = v * 4u;
    const uint32_t rrow = a.d.n * 4u, irow = (uint32_t)a.input.stride * 4u, orow = (uint32_t)a.out.stride * 4u;
    const bool uni = __builtin_amdgcn_ballot_w64(idx0 != (uint32_t)__builtin_amdgcn_readfirstlane((int)idx0)) == 0;
    auto slot_of = [&](uint32_t f) ZH_INLINE_LAMBDA {                   // f - start < D and idx0 < D: one conditional subtraction
        const uint32_t s = idx0 + (f - a.start);
        return (s >= D || s < idx0) ? s - D : s;
    };
    uint32_t f = f0;
    for (; f + 8 <= f1; f += 8) {
        const zh_rsrc_t ri = zrow_rsrc(a.input.p, a.input.stride, f), ro = zrow_rsrc(a.out.p, a.out.stride, f);
        const uint32_t sl0 = slot_of(f), s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl0);
        const bool rows = uni && s0 + 8 <= D && s0 + 8 > s0;           // wave-uniform
        const zh_rsrc_t rr = zrow_rsrc(a.d.ring, a.d.n, rows ? s0 : 0u);
        float dl[8], x[8], base[8];
        float *sp[8];
        if (rows) {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) { dl[k] = zrow_load<1>(rr, voff, k * rrow); sp[k] = nullptr; }
        } else {
            uint32_t sl = sl0;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                sp[k] = a.d.ring + (size_t)sl * a.d.n + v;
                dl[k] = *sp[k];
                sl = sl + 1 == D ? 0 : sl + 1;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            x[k] = zrow_load<1>(ri, voff, k * irow);
            base[k] = BASE ? zrow_load<1>(ro, voff, k * orow) : 0.0f;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) body(k, dl[k], x[k], base[k], ro, rr, rows, sp[k]);
    }
    for (; f < f1; f++) {
        const zh_rsrc_t ri = zrow_rsrc(a.input.p, a.input.stride, f), ro = zrow_rsrc(a.out.p, a.out.stride, f);
        float *sp = a.d.ring + (size_t)slot_of(f) * a.d.n + v;
        body(0u, *sp, zrow_load<1>(ri, voff, 0), BASE ? zrow_load<1>(ro, voff, 0) : 0.0f, ro, ro, false, sp);
    }
}
__device__ __forceinline__ float fe_filter_input(float delayed, float x, float feedback) {
    float t0 = 0.0f + delayed;                                        // zero(temp0); readDelayBuffer (examples/modules.zig:425-428)
    t0 = t0 * feedback;                                               // multiplyWithScalar (:433)
    return t0 + x;                                                    // addInto (:436)
}
// grid: x = 256-voice groups, y = chunk; block = 256
__global__ void __launch_bounds__(256) k_fe_tp_a(const FeTpArgs a) {
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x, V = a.d.n;
    if (v >= V) return;
    const uint32_t idx0 = a.d.index[v];
    if (j == 0) { a.e[v] = make_float2(a.l[v], a.b[v]); a.idx0[v] = idx0; }
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end);
    const float feedback = a.feedback.get(v);
    const float cut = zclampf(a.cutoff.get(v), 0.0f, 1.0f);           // Filter.zig:114
    const float res = 1.0f - zclampf(0.0f, 0.0f, 1.0f);               // res = constant(0.0) (:441) -> Filter.zig:118
    float l = 0.0f, b = 0.0f;
    fe_tp_tiles<false>(a, v, idx0, f0, f1, [&](uint32_t, float delayed, float x, float, const zh_rsrc_t &, const zh_rsrc_t &, bool, float *) ZH_INLINE_LAMBDA {
        svf_step(l, b, fe_filter_input(delayed, x, feedback), cut, res);
    });
    a.e[(size_t)(j + 1) * V + v] = make_float2(l, b);
}
template <bool ZF>
__global__ void __launch_bounds__(256) k_fe_tp_b(const FeTpArgs a) {
    const uint32_t j = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.d.n) return;
    const size_t V = a.d.n;
    const uint32_t idx0 = a.idx0[v], D = a.d.delay_samples;
    const uint32_t f0 = min(a.start + j * a.L, a.end), f1 = min(f0 + a.L, a.end);
    const float feedback = a.feedback.get(v);
    const float cut = zclampf(a.cutoff.get(v), 0.0f, 1.0f);
    const float res = 1.0f - zclampf(0.0f, 0.0f, 1.0f);
    const float2 s0 = a.e[v];
    float l = s0.x, b = s0.y;
    svf_scan<kTpMaxChunks - 1>(l, b, cut, res, a.L, j, [&](uint32_t i) ZH_INLINE_LAMBDA { return a.e[(size_t)(i + 1) * V + v]; });
    const uint32_t voff = v * 4u, rrow = a.d.n * 4u, orow = (uint32_t)a.out.stride * 4u;
    // a cutoff near zero: not as chunks -- the voice's chunk-0 lane walks the piece frame by frame (filter_tp.hip.h kTpExactCutBelow)
    const bool exact_v = cut < kTpExactCutBelow;
    bool store = !exact_v;
    auto body = [&](uint32_t k, float delayed, float x, float base, const zh_rsrc_t &ro, const zh_rsrc_t &rr, bool rows, float *sp) ZH_INLINE_LAMBDA {
        const SvfOut s = svf_step(l, b, fe_filter_input(delayed, x, feedback), cut, res);   // Filter.paint low_pass (Filter.zig:135-146)
        const float t1 = svf_lowpass_into_zero(s.l, s.b);             // zero(temp1); += (:439)
        if (store) {
            zrow_store<1>(ro, voff, k * orow, base + t1);             // addInto(output, temp1) (:448)
            if (rows) zrow_store<1>(rr, voff, k * rrow, t1);          // writeDelayBuffer(temp1) (:452)
            else *sp = t1;
        }
    };
    fe_tp_tiles<!ZF>(a, v, idx0, f0, f1, body);
    auto leave = [&]() ZH_INLINE_LAMBDA {
        a.l[v] = l; a.b[v] = b;
        const uint32_t s = idx0 + (a.end - a.start);                  // <= delay_samples frames per piece
        a.d.index[v] = (s >= D || s < idx0) ? s - D : s;
    };
    if (!exact_v && f1 == a.end && f1 > f0) leave();
    if (j == 0 && __builtin_amdgcn_ballot_w64(exact_v) != 0) {
        if (exact_v) {                                                // (a piece is <= delay_samples frames: every slot it reads was written before it)
            l = s0.x; b = s0.y;
            store = true;
            fe_tp_tiles<!ZF>(a, v, idx0, a.start, a.end, body);
            leave();
        }
    }
}

// the chunked form needs a delay of at least a chunk and an input image that does not overlap the output image
static bool delay_can_chunk(const DelayState &d, const zh_buf &out, const zh_buf &in) {
    if (d.delay_samples < 8) return false;
    const uintptr_t o0 = (uintptr_t)out.ptr, o1 = o0 + (size_t)out.stride * out.frames * sizeof(float);
    const uintptr_t i0 = (uintptr_t)in.ptr, i1 = i0 + (size_t)in.stride * in.frames * sizeof(float);
    return o1 <= i0 || i1 <= o0;
}

static int delay_alloc(zh_ctx *ctx, DelayState &d, uint32_t n, uint32_t delay_samples) {
    d.ring = nullptr; d.index = nullptr; d.n = n; d.delay_samples = delay_samples;
    int rc = dev_alloc(&d.ring, (size_t)delay_samples * n);
    if (!rc) rc = dev_alloc(&d.index, n);
    if (!rc && n) rc = (int)hipMemsetAsync(d.ring, 0, (size_t)delay_samples * n * 4, ctx->stream);   // delay.zig:12-17
    if (!rc && n) rc = (int)hipMemsetAsync(d.index, 0, (size_t)n * 4, ctx->stream);
    return rc;
}
static void delay_free(DelayState &d) { (void)hipFree(d.ring); (void)hipFree(d.index); }

static int delay_reset(zh_ctx *ctx, DelayState &d) {                  // delay.zig:19-22
    if (!d.n) return ZH_OK;
    ZH_TRY(hipMemsetAsync(d.ring, 0, (size_t)d.delay_samples * d.n * 4, ctx->stream));
    ZH_TRY(hipMemsetAsync(d.index, 0, (size_t)d.n * 4, ctx->stream));
    return ZH_OK;
}

static int delay_get(zh_ctx *ctx, const DelayState &d, float *rings, uint32_t *index) {
    if (!rings || !index) return ZH_ERR_INVALID;
    std::vector<float> img((size_t)d.delay_samples * d.n);
    int rc = zh_download(ctx, img.data(), d.ring, img.size() * 4);
    if (!rc) rc = zh_download(ctx, index, d.index, (size_t)d.n * 4);
    if (rc) return rc;
    for (uint32_t v = 0; v < d.n; v++)
        for (uint32_t k = 0; k < d.delay_samples; k++) rings[(size_t)v * d.delay_samples + k] = img[(size_t)k * d.n + v];
    return ZH_OK;
}
static int delay_set(zh_ctx *ctx, const DelayState &d, const float *rings, const uint32_t *index) {
    if (!rings || !index) return ZH_ERR_INVALID;
    for (uint32_t v = 0; v < d.n; v++) if (index[v] >= d.delay_samples) return ZH_ERR_INVALID;   // "always < delay_samples" (:10)
    std::vector<float> img((size_t)d.delay_samples * d.n);
    for (uint32_t v = 0; v < d.n; v++)
        for (uint32_t k = 0; k < d.delay_samples; k++) img[(size_t)k * d.n + v] = rings[(size_t)v * d.delay_samples + k];
    int rc = zh_upload(ctx, d.ring, img.data(), img.size() * 4);
    if (!rc) rc = zh_upload(ctx, d.index, index, (size_t)d.n * 4);
    return rc;
}

extern "C" {

int zh_delay_create(zh_ctx *ctx, uint32_t n, uint32_t delay_samples, zh_delay **out) { ZH_GUARD(ctx);
    if (!ctx || !out || delay_samples == 0) return ZH_ERR_INVALID;    // Delay(0) never makes progress in the reference
    zh_delay *m = new (std::nothrow) zh_delay();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx;
    int rc = delay_alloc(ctx, m->d, n, delay_samples);
    if (rc) { delay_free(m->d); delete m; return rc; }
    *out = m;
    return ZH_OK;
}
int zh_delay_destroy(zh_delay *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    delay_free(m->d);
    delete m;
    return ZH_OK;
}
int zh_delay_reset(zh_delay *m) { ZH_GUARD(m ? m->ctx : nullptr); return m ? delay_reset(m->ctx, m->d) : ZH_ERR_INVALID; }
int zh_delay_get_state(zh_delay *m, float *rings, uint32_t *index) { ZH_GUARD(m ? m->ctx : nullptr); return m ? delay_get(m->ctx, m->d, rings, index) : ZH_ERR_INVALID; }
int zh_delay_set_state(zh_delay *m, const float *rings, const uint32_t *index) { ZH_GUARD(m ? m->ctx : nullptr); return m ? delay_set(m->ctx, m->d, rings, index) : ZH_ERR_INVALID; }
int zh_delay_paint(zh_delay *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                   zh_bool note_id_changed, const zh_delay_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;                                            // examples/modules.zig:370-371
    if (!m || !outputs || !p || end < start || !buf_covers(outputs[0], m->d.n, end) || !buf_covers(p->input, m->d.n, end)) return ZH_ERR_INVALID;
    if (m->d.n == 0 || end == start) return ZH_OK;
    const bool chunked = delay_can_chunk(m->d, outputs[0], p->input);
    // delay 300, 1,024 frames, walk -> independent frames (three launches): 57 -> 13 us at 4,096 voices, 75 -> 31 at 16,384,
    // 423 -> 271 at 131,072: at every voice count
    const uint32_t frames_max = (uint32_t)zh_form(ZF_DELAY_FRAMES_MAX);
    if (chunked && m->d.n <= frames_max && end - start >= 64) {
        hipStream_t st = m->ctx->stream;
        const uint32_t n = end - start, D = m->d.delay_samples, stored = n < D ? n : D;
        const bool fused = n <= D, zf = flags & ZH_PAINT_ZERO_FIRST;
        const dim3 grid((m->d.n + 63) / 64, ((n + 31) / 32 + 3) / 4);
#define ZH_DF(ZF_, W_) ZH_LAUNCH((k_delay_frames<ZF_, W_>), grid, dim3(256), 0, st, m->d, mk_img(outputs[0]), mk_cimg(p->input), start, end)
        if (zf) { if (fused) ZH_DF(true, true); else ZH_DF(true, false); }
        else { if (fused) ZH_DF(false, true); else ZH_DF(false, false); }
#undef ZH_DF
        if (!fused) ZH_LAUNCH(k_delay_store, dim3((m->d.n + 63) / 64, ((stored + 31) / 32 + 3) / 4), dim3(256), 0, st, m->d, mk_cimg(p->input), start, end);
        ZH_LAUNCH(k_delay_advance, dim3((m->d.n + 255) / 256), dim3(256), 0, st, m->d, n);
        return zh_launch_status();
    }
#define ZH_DL(ZF_, CH_) ZH_LAUNCH((k_simple_delay<ZF_, CH_>), seq_grid(m->d.n), dim3(kSeqBlock), 0, m->ctx->stream, m->d, mk_img(outputs[0]), mk_cimg(p->input), start, end)
    if (flags & ZH_PAINT_ZERO_FIRST) { if (chunked) ZH_DL(true, 8); else ZH_DL(true, 1); }
    else { if (chunked) ZH_DL(false, 8); else ZH_DL(false, 1); }
#undef ZH_DL
    return zh_launch_status();
}

int zh_filtered_echoes_create(zh_ctx *ctx, uint32_t n, uint32_t delay_samples, zh_filtered_echoes **out) { ZH_GUARD(ctx);
    if (!ctx || !out || delay_samples == 0) return ZH_ERR_INVALID;
    zh_filtered_echoes *m = new (std::nothrow) zh_filtered_echoes();
    if (!m) return ZH_ERR_INVALID;
    m->ctx = ctx; m->l = m->b = nullptr; m->tp = nullptr;
    int rc = delay_alloc(ctx, m->d, n, delay_samples);
    if (!rc) rc = dev_alloc(&m->l, n);
    if (!rc) rc = dev_alloc(&m->b, n);
    if (!rc && n) rc = (int)hipMemsetAsync(m->l, 0, (size_t)n * 4, ctx->stream);   // Filter.init()
    if (!rc && n) rc = (int)hipMemsetAsync(m->b, 0, (size_t)n * 4, ctx->stream);
    if (rc) { delay_free(m->d); (void)hipFree(m->l); (void)hipFree(m->b); delete m; return rc; }
    *out = m;
    return ZH_OK;
}
int zh_filtered_echoes_destroy(zh_filtered_echoes *m) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m) return ZH_ERR_INVALID;
    (void)hipStreamSynchronize(m->ctx->stream);
    delay_free(m->d); (void)hipFree(m->l); (void)hipFree(m->b); (void)hipFree(m->tp);
    delete m;
    return ZH_OK;
}
int zh_filtered_echoes_reset(zh_filtered_echoes *m) { ZH_GUARD(m ? m->ctx : nullptr); return m ? delay_reset(m->ctx, m->d) : ZH_ERR_INVALID; }   // :407-409
int zh_filtered_echoes_get_state(zh_filtered_echoes *m, float *rings, uint32_t *index, zh_filter_state *filter) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !filter) return ZH_ERR_INVALID;
    int rc = delay_get(m->ctx, m->d, rings, index);
    if (rc) return rc;
    std::vector<float> l(m->d.n), b(m->d.n);
    rc = zh_download(m->ctx, l.data(), m->l, (size_t)m->d.n * 4);
    if (!rc) rc = zh_download(m->ctx, b.data(), m->b, (size_t)m->d.n * 4);
    if (rc) return rc;
    for (uint32_t v = 0; v < m->d.n; v++) filter[v] = zh_filter_state{l[v], b[v]};
    return ZH_OK;
}
int zh_filtered_echoes_set_state(zh_filtered_echoes *m, const float *rings, const uint32_t *index, const zh_filter_state *filter) { ZH_GUARD(m ? m->ctx : nullptr);
    if (!m || !filter) return ZH_ERR_INVALID;
    int rc = delay_set(m->ctx, m->d, rings, index);
    if (rc) return rc;
    std::vector<float> l(m->d.n), b(m->d.n);
    for (uint32_t v = 0; v < m->d.n; v++) { l[v] = filter[v].l; b[v] = filter[v].b; }
    rc = zh_upload(m->ctx, m->l, l.data(), (size_t)m->d.n * 4);
    if (!rc) rc = zh_upload(m->ctx, m->b, b.data(), (size_t)m->d.n * 4);
    return rc;
}
int zh_filtered_echoes_paint(zh_filtered_echoes *m, uint32_t start, uint32_t end, const zh_buf *outputs, const zh_buf *temps,
                             zh_bool note_id_changed, const zh_filtered_echoes_params *p, uint32_t flags) { ZH_GUARD(m ? m->ctx : nullptr);
    (void)temps; (void)note_id_changed;
    if (!m || !outputs || !p || end < start || !buf_covers(outputs[0], m->d.n, end) || !buf_covers(p->input, m->d.n, end)) return ZH_ERR_INVALID;
    if (m->d.n == 0 || end == start) return ZH_OK;
    const bool chunked = delay_can_chunk(m->d, outputs[0], p->input);
    // ZH_PAINT_TOLERANT, few voices: pieces of <= delay_samples frames, each a time-parallel Filter paint (k_fe_tp_a / _b above).
    // Not for a span of more than three pieces (a delay much shorter than the span): a launch pair per piece, ~10 us each at
    // 4,096 voices -- 1,024 frames over a delay of 300 took 42 us as four pieces against the exact form's 44.
    if ((flags & ZH_PAINT_TOLERANT) && chunked && end - start >= 64) {
        const uint32_t n = end - start, D = m->d.delay_samples, piece = D < 4096u ? D : 4096u;
        const uint32_t C = zh_tp_chunks(m->d.n, ZF_ECHOES_TP_MAX, piece < n ? piece : n);   // (six image streams: level with the exact form at 8,192 voices, HBM-bound behind it from 16,384)
        if (C >= 2 && piece >= 64 && (n + piece - 1) / piece <= 3) {
            if (!m->tp && !m->ctx->capturing && dev_alloc(&m->tp, kFeTpFloats * m->d.n) != ZH_OK) { m->tp = nullptr; (void)hipGetLastError(); }
            if (m->tp) {
                FeTpArgs a;
                a.d = m->d; a.l = m->l; a.b = m->b; a.e = reinterpret_cast<float2 *>(m->tp);
                a.idx0 = reinterpret_cast<uint32_t *>(m->tp + (size_t)(kTpMaxChunks + 1) * 2 * m->d.n);
                a.out = mk_img(outputs[0]); a.input = mk_cimg(p->input); a.feedback = mk_f32(p->feedback_volume); a.cutoff = mk_f32(p->cutoff);
                for (uint32_t s = start; s < end; s += piece) {
                    a.start = s; a.end = end - s > piece ? s + piece : end;
                    const uint32_t len = a.end - a.start, c = C < len ? C : len;
                    a.L = (len + c - 1) / c;
                    const dim3 grid((m->d.n + 255) / 256, (len + a.L - 1) / a.L);
                    ZH_LAUNCH(k_fe_tp_a, grid, dim3(256), 0, m->ctx->stream, a);
                    if (flags & ZH_PAINT_ZERO_FIRST) ZH_LAUNCH(k_fe_tp_b<true>, grid, dim3(256), 0, m->ctx->stream, a);
                    else ZH_LAUNCH(k_fe_tp_b<false>, grid, dim3(256), 0, m->ctx->stream, a);
                }
                return zh_launch_status();
            }
        }
    }
    // delay 300, one wave per 64 voices -> three: 1,024 / 4,096 / 16,384 / 32,768 / 65,536 voices 88 / 91 / 103 / 194 / 263 ->
    // 56 / 57 / 62 / 110 / 215 us; at 131,072 voices the one-wave form is ahead (406 against 461)
    const uint32_t pc_max = (uint32_t)zh_form(ZF_ECHOES_PC_MAX);
    if (chunked && m->d.n <= pc_max && m->d.delay_samples >= 192 && end - start >= 64) {
        if (flags & ZH_PAINT_ZERO_FIRST) ZH_LAUNCH(k_filtered_echoes_pc<true>, dim3((m->d.n + 63) / 64), dim3(192), 0, m->ctx->stream, m->d, m->l, m->b, mk_img(outputs[0]), mk_cimg(p->input), start, end, mk_f32(p->feedback_volume), mk_f32(p->cutoff));
        else ZH_LAUNCH(k_filtered_echoes_pc<false>, dim3((m->d.n + 63) / 64), dim3(192), 0, m->ctx->stream, m->d, m->l, m->b, mk_img(outputs[0]), mk_cimg(p->input), start, end, mk_f32(p->feedback_volume), mk_f32(p->cutoff));
        return zh_launch_status();
    }
#define ZH_FE(ZF_, CH_) ZH_LAUNCH((k_filtered_echoes<ZF_, CH_>), seq_grid(m->d.n), dim3(kSeqBlock), 0, m->ctx->stream, m->d, m->l, m->b, mk_img(outputs[0]), mk_cimg(p->input), start, end, mk_f32(p->feedback_volume), mk_f32(p->cutoff))
    if (flags & ZH_PAINT_ZERO_FIRST) { if (chunked) ZH_FE(true, 8); else ZH_FE(true, 1); }
    else { if (chunked) ZH_FE(false, 8); else ZH_FE(false, 1); }
#undef ZH_FE
    return zh_launch_status();
}

}  // extern "C"
