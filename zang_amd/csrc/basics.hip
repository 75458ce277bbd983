// basics.hip -- zang's buffer primitives (src/zang/basics.zig:12-78) over [frame][voice]
// images, plus the voice mixdown.  Pure HBM streaming: 16 B per lane where the images
// allow it (4 consecutive voices of one frame), scalar lanes otherwise.
#include "common.hip.h"
#include <stdlib.h>

enum EwOp { OP_ZERO, OP_SET, OP_COPY, OP_ADD, OP_ADD_INTO, OP_ADD_SCALAR, OP_ADD_SCALAR_INTO,
            OP_MUL, OP_MUL_WITH, OP_MUL_SCALAR, OP_MUL_WITH_SCALAR };

template <int OP> struct OpTraits {
    static constexpr bool reads_dst = !(OP == OP_ZERO || OP == OP_SET || OP == OP_COPY);
    static constexpr bool uses_a = (OP == OP_COPY || OP == OP_ADD || OP == OP_ADD_INTO || OP == OP_ADD_SCALAR ||
                                    OP == OP_MUL || OP == OP_MUL_WITH || OP == OP_MUL_SCALAR);
    static constexpr bool uses_b = (OP == OP_ADD || OP == OP_MUL);
    static constexpr bool uses_s = (OP == OP_SET || OP == OP_ADD_SCALAR || OP == OP_ADD_SCALAR_INTO ||
                                    OP == OP_MUL_SCALAR || OP == OP_MUL_WITH_SCALAR);
};

// One element of each op, in the reference's evaluation order (no fused multiply-add).
template <int OP> __device__ __forceinline__ float ew_apply(float d, float a, float b, float s) {
    if constexpr (OP == OP_ZERO) return 0.0f;                 // basics.zig:12-14
    else if constexpr (OP == OP_SET) return s;                // :16-18
    else if constexpr (OP == OP_COPY) return a;               // :20-22
    else if constexpr (OP == OP_ADD) return d + (a + b);      // :24-29
    else if constexpr (OP == OP_ADD_INTO) return d + a;       // :31-36
    else if constexpr (OP == OP_ADD_SCALAR) return d + (a + s);   // :38-43
    else if constexpr (OP == OP_ADD_SCALAR_INTO) return d + s;    // :45-50
    else if constexpr (OP == OP_MUL) return d + a * b;        // :52-57
    else if constexpr (OP == OP_MUL_WITH) return d * a;       // :59-64
    else if constexpr (OP == OP_MUL_SCALAR) return d + a * s; // :66-71
    else return d * s;                                        // :73-78
}

// VEC = 4: a thread owns 4 consecutive voices of one frame (float4); VEC = 1: one voice.
// A workgroup is TX x (256 / TX) threads: x = voice quads (TX = 2^tx_log2 <= 256, the smallest power of two that covers the row, so
// that narrow images keep every lane busy), y = frames; blockIdx.y strides over the frames, R rows per step with their loads ahead of
// the arithmetic.  (Round 5: the earlier flat index paid a 64-bit division per element -- ~100 instructions against one 16-byte
// store: zero / set at 131,072 voices 104 -> see profiles/r05/ab_basics_2d.txt.)
template <int OP, int VEC>
__global__ void __launch_bounds__(256) k_elementwise(Img dst, CImg a, CImg b, F32P s, uint32_t start,
                                                     uint32_t nframes, uint32_t nvq /* voices / VEC */, uint32_t tx_log2) {
    using T = OpTraits<OP>;
    constexpr int R = 4;
    const uint32_t tx = threadIdx.x & ((1u << tx_log2) - 1u), ty = threadIdx.x >> tx_log2, TY = 256u >> tx_log2;
    const uint32_t q = (blockIdx.x << tx_log2) + tx;
    if (q >= nvq) return;
    const uint32_t v = q * VEC;
    const uint32_t rstep = gridDim.y * TY;
    if constexpr (VEC == 4) {
        zv4f sv = {s.value, s.value, s.value, s.value};
        if constexpr (T::uses_s) if (s.pv) sv = *reinterpret_cast<const zv4f *>(s.pv + v);
        for (uint32_t r0 = blockIdx.y * TY + ty; r0 < nframes; r0 += rstep * R) {
            zv4f d[R], av[R], bv[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                const uint32_t r = r0 + k * rstep;
                d[k] = av[k] = bv[k] = zv4f{0, 0, 0, 0};
                if (r < nframes) {
                    if constexpr (T::reads_dst) d[k] = *reinterpret_cast<const zv4f *>(dst.at(start + r, v));
                    if constexpr (T::uses_a) av[k] = *reinterpret_cast<const zv4f *>(a.at(start + r, v));
                    if constexpr (T::uses_b) bv[k] = *reinterpret_cast<const zv4f *>(b.at(start + r, v));
                }
            }
#pragma unroll
            for (int k = 0; k < R; k++) {
                const uint32_t r = r0 + k * rstep;
                if (r < nframes) {
                    zv4f o;
                    o.x = ew_apply<OP>(d[k].x, av[k].x, bv[k].x, sv.x);
                    o.y = ew_apply<OP>(d[k].y, av[k].y, bv[k].y, sv.y);
                    o.z = ew_apply<OP>(d[k].z, av[k].z, bv[k].z, sv.z);
                    o.w = ew_apply<OP>(d[k].w, av[k].w, bv[k].w, sv.w);
                    // write-through (sc1): the image drains to HBM while the kernel runs instead of at its end -- 4,096 voices 10-21 % faster
                    // on every operation, 16,384 voices 2-12 %, level at 131,072 (alternating A/B, profiles/r05/ab_basics_sc1.txt)
                    store4_sc1(dst.at(start + r, v), o);
                }
            }
        }
    } else {
        const float sv = T::uses_s ? s.get(v) : 0.0f;
        for (uint32_t r = blockIdx.y * TY + ty; r < nframes; r += rstep) {
            float d = 0, av = 0, bv = 0;
            if constexpr (T::reads_dst) d = *dst.at(start + r, v);
            if constexpr (T::uses_a) av = *a.at(start + r, v);
            if constexpr (T::uses_b) bv = *b.at(start + r, v);
            *dst.at(start + r, v) = ew_apply<OP>(d, av, bv, sv);
        }
    }
}

// Many voices (basics_rows_min, dispatch.hip): the chunked oscillator's launch shape -- a wave owns RC consecutive rows of a 256-voice
// column, a workgroup four consecutive chunks, no loop.  The same bytes in the same 16-byte stores, but spread over the HBM channels the
// way k_osc_const4's are: at 131,072 voices zero 104 -> 87 us, copy 219 -> 178, addInto 322 -> 280, add 428 -> 382; from 32,768 voices
// on never slower, below it slower (4,096 voices: addInto 8.2 -> 10.2 us) -- profiles/r05/ab_basics_rows.txt.
template <int OP, int RC, int SM>
__global__ void __launch_bounds__(256) k_elementwise_chunks(Img dst, CImg a, CImg b, F32P s, uint32_t start, uint32_t nframes, uint32_t nvq) {
    using T = OpTraits<OP>;
    const uint32_t q = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (q >= nvq) return;
    const uint32_t v = q * 4, r0 = chunk * RC;
    zv4f sv = {s.value, s.value, s.value, s.value};
    if constexpr (T::uses_s) if (s.pv) sv = *reinterpret_cast<const zv4f *>(s.pv + v);
    zv4f d[RC], av[RC], bv[RC];
#pragma unroll
    for (int k = 0; k < RC; k++) {
        const uint32_t r = r0 + k;
        d[k] = av[k] = bv[k] = zv4f{0, 0, 0, 0};
        if (r < nframes) {
            if constexpr (T::reads_dst) d[k] = *reinterpret_cast<const zv4f *>(dst.at(start + r, v));
            if constexpr (T::uses_a) av[k] = *reinterpret_cast<const zv4f *>(a.at(start + r, v));
            if constexpr (T::uses_b) bv[k] = *reinterpret_cast<const zv4f *>(b.at(start + r, v));
        }
    }
#pragma unroll
    for (int k = 0; k < RC; k++) {
        const uint32_t r = r0 + k;
        if (r < nframes) {
            zv4f o;
            o.x = ew_apply<OP>(d[k].x, av[k].x, bv[k].x, sv.x);
            o.y = ew_apply<OP>(d[k].y, av[k].y, bv[k].y, sv.y);
            o.z = ew_apply<OP>(d[k].z, av[k].z, bv[k].z, sv.z);
            o.w = ew_apply<OP>(d[k].w, av[k].w, bv[k].w, sv.w);
            if constexpr (SM == ST_NT) __builtin_nontemporal_store(o, reinterpret_cast<zv4f *>(dst.at(start + r, v)));
            else if constexpr (SM == ST_SC1) store4_sc1(dst.at(start + r, v), o);
            else *reinterpret_cast<zv4f *>(dst.at(start + r, v)) = o;
        }
    }
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

template <int OP>
static int launch_ew(zh_ctx *ctx, uint32_t start, uint32_t end, const zh_buf &dest, const zh_buf *a, const zh_buf *b,
                     const zh_f32 *s) {
    using T = OpTraits<OP>;
    if (!ctx || end < start) return ZH_ERR_INVALID;
    if (!buf_covers(dest, dest.voices, end)) return ZH_ERR_INVALID;
    if (T::uses_a && (!a || !buf_covers(*a, dest.voices, end))) return ZH_ERR_INVALID;
    if (T::uses_b && (!b || !buf_covers(*b, dest.voices, end))) return ZH_ERR_INVALID;
    const uint32_t V = dest.voices, nframes = end - start;
    if (V == 0 || nframes == 0) return ZH_OK;
    Img d = mk_img(dest);
    CImg ai = a ? mk_cimg(*a) : CImg{nullptr, 0}, bi = b ? mk_cimg(*b) : CImg{nullptr, 0};
    F32P sp = s ? mk_f32(*s) : F32P{0.0f, nullptr};
    bool vec = (V % 4 == 0) && (dest.stride % 4 == 0) && aligned16(dest.ptr);
    if (T::uses_a) vec = vec && (a->stride % 4 == 0) && aligned16(a->ptr);
    if (T::uses_b) vec = vec && (b->stride % 4 == 0) && aligned16(b->ptr);
    if (T::uses_s && sp.pv) vec = vec && aligned16(sp.pv);
    const uint32_t nvq = vec ? V / 4 : V;
    uint32_t tx_log2 = 0;
    while (tx_log2 < 8 && (1u << tx_log2) < nvq) tx_log2++;
    const uint32_t TX = 1u << tx_log2, TY = 256u / TX;
    const uint32_t gx = (nvq + TX - 1) / TX;
    uint32_t gy = (nframes + TY - 1) / TY;                            // about 16 workgroups per CU; the rows stride over the rest
    const uint32_t gy_max = gx >= 256u * 16u ? 1u : (256u * 16u + gx - 1) / gx;
    if (gy > gy_max) gy = gy_max;
    if (vec && (long)V >= zh_form(ZF_BASICS_ROWS_MIN)) {
        const dim3 g((nvq + 63) / 64, ((nframes + 2) / 3 + 3) / 4);
        ZH_LAUNCH((k_elementwise_chunks<OP, 3, ST_SC1>), g, dim3(256), 0, ctx->stream, d, ai, bi, sp, start, nframes, nvq);
        return zh_launch_status();
    }
    if (vec) ZH_LAUNCH((k_elementwise<OP, 4>), dim3(gx, gy), dim3(256), 0, ctx->stream, d, ai, bi, sp, start, nframes, nvq, tx_log2);
    else ZH_LAUNCH((k_elementwise<OP, 1>), dim3(gx, gy), dim3(256), 0, ctx->stream, d, ai, bi, sp, start, nframes, nvq, tx_log2);
    return zh_launch_status();
}

// ---------------------------------------------------------------------------------- mixdown
// dst[f] += sum_v src[f][v].  Pass 1: a workgroup of 256 lanes owns a tile of 1024 voices
// (float4 per lane) x MIX_FPB frames; per frame the lane adds its 4 voices left to right,
// the wave reduces with a fixed xor-shuffle butterfly, the 4 wave sums meet in LDS and are
// added in wave order.  Pass 2 adds the tile partials in tile order.  The order is fixed,
// so a given (V, layout) always produces the same bits.
constexpr int MIX_TILE = 1024;
constexpr int MIX_FPB = 8;

__device__ __forceinline__ float wave_sum64(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__global__ void __launch_bounds__(256) k_mix_pass1(CImg src, uint32_t V, uint32_t start, uint32_t end,
                                                   float *__restrict__ partials /*[tiles][nframes]*/) {
    __shared__ float wsum[MIX_FPB][4];
    const uint32_t tile = blockIdx.x, f0 = start + blockIdx.y * MIX_FPB;
    const uint32_t v = tile * MIX_TILE + threadIdx.x * 4;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nframes = end - start;
    float acc[MIX_FPB];
#pragma unroll
    for (int k = 0; k < MIX_FPB; k++) {
        const uint32_t f = f0 + k;
        float s = 0.0f;
        if (f < end) {
            if (v + 3 < V) {
                const float4 x = *reinterpret_cast<const float4 *>(src.at(f, v));
                s = ((x.x + x.y) + x.z) + x.w;
            } else {
                for (uint32_t j = 0; j < 4; j++) if (v + j < V) s += *src.at(f, v + j);
            }
        }
        acc[k] = wave_sum64(s);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MIX_FPB; k++) wsum[k][wave] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < MIX_FPB) {
        const uint32_t f = f0 + threadIdx.x;
        if (f < end) {
            const float *w = wsum[threadIdx.x];
            partials[(size_t)tile * nframes + (f - start)] = ((w[0] + w[1]) + w[2]) + w[3];
        }
    }
}

// scalar-lane variant for images whose rows are not 16 B aligned
__global__ void __launch_bounds__(256) k_mix_pass1_scalar(CImg src, uint32_t V, uint32_t start, uint32_t end,
                                                          float *__restrict__ partials) {
    __shared__ float wsum[MIX_FPB][4];
    const uint32_t tile = blockIdx.x, f0 = start + blockIdx.y * MIX_FPB;
    const uint32_t v = tile * MIX_TILE + threadIdx.x * 4;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nframes = end - start;
    float acc[MIX_FPB];
#pragma unroll
    for (int k = 0; k < MIX_FPB; k++) {
        const uint32_t f = f0 + k;
        float s = 0.0f;
        if (f < end) {
            if (v + 3 < V) {
                const float *p = src.at(f, v);
                s = ((p[0] + p[1]) + p[2]) + p[3];
            } else {
                for (uint32_t j = 0; j < 4; j++) if (v + j < V) s += *src.at(f, v + j);
            }
        }
        acc[k] = wave_sum64(s);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MIX_FPB; k++) wsum[k][wave] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < MIX_FPB) {
        const uint32_t f = f0 + threadIdx.x;
        if (f < end) {
            const float *w = wsum[threadIdx.x];
            partials[(size_t)tile * nframes + (f - start)] = ((w[0] + w[1]) + w[2]) + w[3];
        }
    }
}

// Pass 2: dst[f] (+)= sum over tiles of partials[tile][f].  A workgroup owns 64 frames and
// splits the tiles into MIX_SEG contiguous segments (one wave per segment); a lane adds its
// segment's partials in tile order, then lane f of wave 0 adds the MIX_SEG segment sums in
// segment order.  Loads are coalesced (64 consecutive frames per wave) and independent.
constexpr int MIX_SEG = 16;
__global__ void __launch_bounds__(64 * MIX_SEG) k_mix_pass2(const float *__restrict__ partials, uint32_t tiles,
                                                            uint32_t nframes, float *__restrict__ dst, int zero_first) {
    __shared__ float seg_sum[MIX_SEG][64];
    const uint32_t lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const uint32_t f = blockIdx.x * 64 + lane;
    const uint32_t per = (tiles + MIX_SEG - 1) / MIX_SEG;
    const uint32_t t0 = min(seg * per, tiles), t1 = min(t0 + per, tiles);
    float s = 0.0f;
    if (f < nframes) {
        const float *p = partials + (size_t)t0 * nframes + f;
#pragma unroll 8
        for (uint32_t t = t0; t < t1; t++, p += nframes) s += *p;
    }
    seg_sum[seg][lane] = s;
    __syncthreads();
    if (seg == 0 && f < nframes) {
        float tot = seg_sum[0][lane];
#pragma unroll
        for (int k = 1; k < MIX_SEG; k++) tot += seg_sum[k][lane];
        dst[f] = (zero_first ? 0.0f : dst[f]) + tot;
    }
}

// Pass 2 for the fused voice kernels (composite.hip k_nice_mix), whose partials are one row per WAVE of 64 voices -- four
// times the rows of the block form, 16 MB per stereo buffer at 131,072 voices -- laid out [channel][frame / G][row][frame % G], G = 8:
// a workgroup owns G frames of one channel, i.e. ONE contiguous run of rows x 32 bytes; thread (q, fl) adds rows q, q + 128,
// q + 256, ... of frame fl (a wave reads eight consecutive rows = 256 contiguous bytes per step), and the first G threads
// combine the 128 strided sums in q order.  grid = (frames / G, channels, buffers): both channels (and all buffers of a batch)
// in ONE launch (a second pass2 launch was 5 us per buffer).  Fixed order => reproducible bits.
// (grid.z = buffer of a batch: partials[buffer][channel][...], one destination pair per buffer)
constexpr int MIXW_F = kMixGroupFrames, MIXW_SEG = 1024 / MIXW_F;     // 8 frames x 128 row classes: 256 workgroups per stereo buffer of 1024 frames
constexpr int kMixMaxBatch = 16;
struct MixDst { float *l[kMixMaxBatch], *r[kMixMaxBatch]; };
__global__ void __launch_bounds__(MIXW_SEG * MIXW_F) k_mix_pass2_wide(const float *__restrict__ partials, size_t channel_stride, uint32_t rows,
                                                                      uint32_t nframes, const MixDst d, int zero_first) {
    __shared__ float seg_sum[MIXW_SEG][MIXW_F];
    const uint32_t fl = threadIdx.x % MIXW_F, seg = threadIdx.x / MIXW_F;
    const uint32_t f = blockIdx.x * MIXW_F + fl;
    const float *part = partials + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * channel_stride + (size_t)blockIdx.x * rows * MIXW_F;
    float *dst = blockIdx.y ? d.r[blockIdx.z] : d.l[blockIdx.z];
    float s = 0.0f;
    if (f < nframes) {
        const float *p = part + (size_t)seg * MIXW_F + fl;
#pragma unroll 8
        for (uint32_t r = seg; r < rows; r += MIXW_SEG, p += MIXW_SEG * MIXW_F) s += *p;
    }
    seg_sum[seg][fl] = s;
    __syncthreads();
    if (seg == 0 && f < nframes) {
        float tot = seg_sum[0][fl];
#pragma unroll
        for (int k = 1; k < MIXW_SEG; k++) tot += seg_sum[k][fl];
        dst[f] = (zero_first ? 0.0f : dst[f]) + tot;
    }
}

// ZH_MIX_SEQUENTIAL: one lane per frame adds voice 0, 1, 2, ... in f32, the exact order of successive
// `+=` paints onto one buffer in the reference (example_song.zig:340-346).  Each lane walks its
// own image row, so this form is for small voice counts only (17 sub-voices in example_song).
__global__ void __launch_bounds__(64) k_mix_sequential(CImg src, uint32_t V, uint32_t start, uint32_t end,
                                                       float *__restrict__ dst, int zero_first) {
    const uint32_t f = start + blockIdx.x * 64 + threadIdx.x;
    if (f >= end) return;
    float s = zero_first ? 0.0f : dst[f];
    const float *row = src.at(f, 0);
    for (uint32_t v = 0; v < V; v++) s += row[v];
    dst[f] = s;
}

// mixDown (src/zang/mixdown.zig:28-86)
__global__ void __launch_bounds__(256) k_mix_down(uint8_t *__restrict__ dst, const float *__restrict__ mix, uint32_t n,
                                                  int s16, uint32_t num_channels, uint32_t channel_index, float mul) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float value = mix[i] * mul;
    if (s16) {                                                        // :40-56
        int32_t c;
        if (value <= -32767.0f) c = -32767;
        else if (value >= 32766.0f) c = 32766;
        else if (value != value) c = 0;
        else c = (int32_t)value;
        const size_t index = ((size_t)i * num_channels + channel_index) * 2;
        dst[index + 0] = (uint8_t)(c & 0xFF);
        dst[index + 1] = (uint8_t)((c >> 8) & 0xFF);
    } else {                                                          // :71-85
        int32_t c;
        if (value <= -127.0f) c = -127;
        else if (value >= 126.0f) c = 126;
        else if (value != value) c = 0;
        else c = (int32_t)value;
        dst[(size_t)i * num_channels + channel_index] = (uint8_t)(int8_t)c;
    }
}

int zh_mix_reserve(zh_ctx *ctx, size_t floats) { ZH_GUARD(ctx);
    if (ctx->mix_partials_floats >= floats) return ZH_OK;
    // Growing: never while the stream is capturing (an allocation cannot be recorded, and a synchronise would
    // invalidate the capture) -- reserve with an eager call of the same size first.  While a captured graph lives, the old block
    // is retired, not freed: the graph keeps its pointer in its mixdown nodes (freed when the context's last graph is destroyed).
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (ctx->capturing || (hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone))
        return ZH_ERR_UNSUPPORTED;
    float *p = nullptr;
    ZH_TRY(hipMalloc((void **)&p, floats * sizeof(float)));
    if (ctx->mix_partials) {
        if (ctx->graphs_live) ctx->mix_retired.push_back(ctx->mix_partials);   // a live graph may name it in its mixdown nodes
        else (void)hipFree(ctx->mix_partials);                                 // nobody else can (hipFree waits for the work in flight)
    }
    ctx->mix_partials = p;
    ctx->mix_partials_floats = floats;
    return ZH_OK;
}

void zh_mix_pass2_launch_at(zh_ctx *ctx, const float *partials, uint32_t tiles, uint32_t nframes, float *dst, int zero_first) {
    ZH_LAUNCH(k_mix_pass2, dim3((nframes + 63) / 64), dim3(64 * MIX_SEG), 0, ctx->stream, partials, tiles,
                       nframes, dst, zero_first);
}
// channels = 1 or 2 (dst1 unused for 1): partials[channel][row][frame], rows summed in row order
void zh_mix_pass2_wide_batch_launch(zh_ctx *ctx, const float *partials, size_t channel_stride, uint32_t rows, uint32_t nframes,
                                    float *const *dst0, float *const *dst1, uint32_t n_buffers, int channels, int zero_first) {
    MixDst d;
    for (int k = 0; k < kMixMaxBatch; k++) { d.l[k] = (uint32_t)k < n_buffers ? dst0[k] : nullptr; d.r[k] = ((uint32_t)k < n_buffers && dst1) ? dst1[k] : nullptr; }
    ZH_LAUNCH(k_mix_pass2_wide, dim3((nframes + MIXW_F - 1) / MIXW_F, channels, n_buffers), dim3(MIXW_SEG * MIXW_F), 0, ctx->stream, partials,
                       channel_stride, rows, nframes, d, zero_first);
}
void zh_mix_pass2_wide_launch(zh_ctx *ctx, const float *partials, size_t channel_stride, uint32_t rows, uint32_t nframes, float *dst0,
                              float *dst1, int channels, int zero_first) {
    zh_mix_pass2_wide_batch_launch(ctx, partials, channel_stride, rows, nframes, &dst0, dst1 ? &dst1 : nullptr, 1, channels, zero_first);
}
void zh_mix_pass2_launch(zh_ctx *ctx, uint32_t tiles, uint32_t nframes, float *dst, int zero_first) {
    zh_mix_pass2_launch_at(ctx, ctx->mix_partials, tiles, nframes, dst, zero_first);
}

extern "C" {

int zh_zero(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest) { ZH_GUARD(ctx); return launch_ew<OP_ZERO>(ctx, s, e, dest, nullptr, nullptr, nullptr); }
int zh_set(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_f32 a) { ZH_GUARD(ctx); return launch_ew<OP_SET>(ctx, s, e, dest, nullptr, nullptr, &a); }
int zh_copy(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf src) { ZH_GUARD(ctx); return launch_ew<OP_COPY>(ctx, s, e, dest, &src, nullptr, nullptr); }
int zh_add(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf a, zh_buf b) { ZH_GUARD(ctx); return launch_ew<OP_ADD>(ctx, s, e, dest, &a, &b, nullptr); }
int zh_add_into(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf src) { ZH_GUARD(ctx); return launch_ew<OP_ADD_INTO>(ctx, s, e, dest, &src, nullptr, nullptr); }
int zh_add_scalar(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf a, zh_f32 b) { ZH_GUARD(ctx); return launch_ew<OP_ADD_SCALAR>(ctx, s, e, dest, &a, nullptr, &b); }
int zh_add_scalar_into(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_f32 a) { ZH_GUARD(ctx); return launch_ew<OP_ADD_SCALAR_INTO>(ctx, s, e, dest, nullptr, nullptr, &a); }
int zh_multiply(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf a, zh_buf b) { ZH_GUARD(ctx); return launch_ew<OP_MUL>(ctx, s, e, dest, &a, &b, nullptr); }
int zh_multiply_with(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf a) { ZH_GUARD(ctx); return launch_ew<OP_MUL_WITH>(ctx, s, e, dest, &a, nullptr, nullptr); }
int zh_multiply_scalar(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_buf a, zh_f32 b) { ZH_GUARD(ctx); return launch_ew<OP_MUL_SCALAR>(ctx, s, e, dest, &a, nullptr, &b); }
int zh_multiply_with_scalar(zh_ctx *ctx, uint32_t s, uint32_t e, zh_buf dest, zh_f32 a) { ZH_GUARD(ctx); return launch_ew<OP_MUL_WITH_SCALAR>(ctx, s, e, dest, nullptr, nullptr, &a); }

int zh_mixdown_voices(zh_ctx *ctx, uint32_t start, uint32_t end, float *dst, zh_buf src, uint32_t flags) { ZH_GUARD(ctx);
    if (!ctx || !dst || end < start || !buf_covers(src, src.voices, end)) return ZH_ERR_INVALID;
    const uint32_t V = src.voices, nframes = end - start;
    if (nframes == 0) return ZH_OK;
    if (flags & ZH_MIX_SEQUENTIAL) {
        ZH_LAUNCH(k_mix_sequential, dim3((nframes + 63) / 64), dim3(64), 0, ctx->stream, mk_cimg(src), V, start, end,
                           dst, (int)(flags & ZH_PAINT_ZERO_FIRST));
        return zh_launch_status();
    }
    const uint32_t tiles = V == 0 ? 0 : (V + MIX_TILE - 1) / MIX_TILE;
    if (tiles) {
        int rc = zh_mix_reserve(ctx, (size_t)tiles * nframes);
        if (rc) return rc;
        dim3 grid(tiles, (nframes + MIX_FPB - 1) / MIX_FPB);
        if (src.stride % 4 == 0 && aligned16(src.ptr))
            ZH_LAUNCH(k_mix_pass1, grid, dim3(256), 0, ctx->stream, mk_cimg(src), V, start, end, ctx->mix_partials);
        else
            ZH_LAUNCH(k_mix_pass1_scalar, grid, dim3(256), 0, ctx->stream, mk_cimg(src), V, start, end, ctx->mix_partials);
    }
    zh_mix_pass2_launch(ctx, tiles, nframes, dst + start, (int)(flags & ZH_PAINT_ZERO_FIRST));
    return zh_launch_status();
}

int zh_mix_down(zh_ctx *ctx, uint8_t *dst, const float *mix, uint32_t n, uint32_t audio_format, uint32_t num_channels,
                uint32_t channel_index, float vol) { ZH_GUARD(ctx);
    if (!ctx || (n && (!dst || !mix)) || audio_format > ZH_AUDIO_SIGNED16_LSB || num_channels == 0 || channel_index >= num_channels)
        return ZH_ERR_INVALID;
    if (!n) return ZH_OK;
    const int s16 = audio_format == ZH_AUDIO_SIGNED16_LSB;
    const float mul = vol * (s16 ? 32767.0f : 127.0f);                // mixdown.zig:37, :68
    ZH_LAUNCH(k_mix_down, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dst, mix, n, s16, num_channels, channel_index, mul);
    return zh_launch_status();
}

}  // extern "C"
