// noise_jump.hip -- white Noise (Noise.zig:44-53) painted for many frame ranges of a span at once.
//
// With few voices the lane-per-voice walk of k_noise is one wave per 64 voices and the span's 1024 dependent
// xoshiro256++ steps are the whole kernel (68 us at 4,096 voices = 64 waves on 1,024 SIMDs).  Here a workgroup owns 256
// voices x ONE frame range [c * ch, (c + 1) * ch): it jumps the voices' states c * ch draws ahead with the
// nibble-table form of T^(c * ch) (noise_jump.hip.h), then walks only its own ch frames.  A voice whose span contains
// one of Random.float's multi-draw samples (2^-41 per sample) is flagged by the range that sees it, and k_noise_fix
// repaints that voice's whole span with the sequential walk -- so the result is the reference's bit for bit always.
#include "noise_jump.hip.h"
#include "seq.hip.h"
#include <mutex>
#include <vector>
#include <stdlib.h>

// ------------------------------------------------------------------ tables (host)
namespace {
struct HostState { uint64_t s[4]; };
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
inline void transition(HostState &r) {                        // Xoshiro256.next's state update (zmath.hip.h zxoshiro_next)
    const uint64_t t = r.s[1] << 17;
    r.s[2] ^= r.s[0]; r.s[3] ^= r.s[1]; r.s[1] ^= r.s[2]; r.s[0] ^= r.s[3];
    r.s[2] ^= t;
    r.s[3] = rotl64(r.s[3], 45);
}
// [table j-1][half][pos][nib][4 dwords] for T^(32 j): column b of T^k is the state reached from the unit state e_b after k
// transitions (linearity), and an entry is the XOR of the columns of its nibble's set bits
std::vector<uint32_t> build_tables() {
    std::vector<HostState> col(256);
    for (int b = 0; b < 256; b++) { col[b] = HostState{{0, 0, 0, 0}}; col[b].s[b >> 6] = 1ull << (b & 63); }
    std::vector<uint32_t> out((size_t)kNoiseJumpTables * kNoiseJumpEntries * 4);
    for (int j = 0; j < kNoiseJumpTables; j++) {
        for (int b = 0; b < 256; b++)
            for (int k = 0; k < 32; k++) transition(col[b]);
        uint32_t *tb = out.data() + (size_t)j * kNoiseJumpEntries * 4;
        for (int pos = 0; pos < 64; pos++)
            for (int nib = 0; nib < 16; nib++) {
                uint64_t acc[4] = {0, 0, 0, 0};
                for (int bit = 0; bit < 4; bit++)
                    if (nib >> bit & 1)
                        for (int q = 0; q < 4; q++) acc[q] ^= col[pos * 4 + bit].s[q];
                uint32_t *lo = tb + ((size_t)(0 * 64 + pos) * 16 + nib) * 4;     // half 0: s0, s1
                uint32_t *hi = tb + ((size_t)(1 * 64 + pos) * 16 + nib) * 4;     // half 1: s2, s3
                lo[0] = (uint32_t)acc[0]; lo[1] = (uint32_t)(acc[0] >> 32); lo[2] = (uint32_t)acc[1]; lo[3] = (uint32_t)(acc[1] >> 32);
                hi[0] = (uint32_t)acc[2]; hi[1] = (uint32_t)(acc[2] >> 32); hi[2] = (uint32_t)acc[3]; hi[3] = (uint32_t)(acc[3] >> 32);
            }
    }
    return out;
}
std::mutex g_tables_mu;
}  // namespace

const uint4 *zh_noise_jump_tables(zh_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    if (ctx->noise_jump) return (const uint4 *)ctx->noise_jump;
    if (ctx->capturing) return nullptr;
    static const std::vector<uint32_t> host = build_tables();        // the same bits for every context
    void *dev = nullptr;
    if (hipMalloc(&dev, host.size() * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemcpyAsync(dev, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(dev); return nullptr; }
    ctx->noise_jump = dev;
    return (const uint4 *)dev;
}

// ------------------------------------------------------------------ kernels
struct NoiseChunkArgs {
    const uint64_t *s[4];        // the voices' states at span start: read only here
    uint64_t *nx[4];             // states after the span, written by the last range (k_noise_fix moves them into s)
    uint32_t *flag;              // per voice: a multi-draw sample was seen
    const uint4 *tables;
    uint32_t V, start, end, ch, tstep;   // ch = frames per range (a multiple of 32); tstep = ch / 32
    Img out;
};

// grid: x = 256-voice groups, y = frame ranges.  out[f][v] = 0.0f + white  (zero + paint; an ADD paint goes through a scratch image)
__global__ void __launch_bounds__(256) k_noise_white_ranges(const NoiseChunkArgs a) {
    __shared__ uint4 tbl[kNoiseJumpEntries];
    const uint32_t c = blockIdx.y;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (c > 0) {                                                      // block-uniform
        noise_jump_load(tbl, a.tables + (size_t)(c * a.tstep - 1) * kNoiseJumpEntries, threadIdx.x, 256);
        __syncthreads();
    }
    if (v >= a.V) return;
    ZXoshiro r{a.s[0][v], a.s[1][v], a.s[2][v], a.s[3][v]};
    if (c > 0) noise_jump_apply(r, tbl);
    const uint32_t f0 = a.start + c * a.ch;
    const uint32_t f1 = min(f0 + a.ch, a.end);
    bool multi = false;
    const float *const *no_in = nullptr;
    frame_loop<8, true, 0>(a.out.p, v, a.out.stride, no_in, nullptr, f0, f1, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
        val = zrandom_float32_multi(r, multi) * 2.0f - 1.0f;          // Noise.zig:51
        return true;
    });
    if (multi) a.flag[v] = 1u;
    if (f1 == a.end) { a.nx[0][v] = r.s0; a.nx[1][v] = r.s1; a.nx[2][v] = r.s2; a.nx[3][v] = r.s3; }
}

// After the ranges: a clean voice takes the state the last range left; a flagged voice (a draw with >= 41 leading zeros
// somewhere in its span: later ranges started from the wrong draw) is repainted whole by the reference's own walk.
__global__ void __launch_bounds__(64) k_noise_fix(uint64_t *__restrict__ s0, uint64_t *__restrict__ s1, uint64_t *__restrict__ s2,
                                                  uint64_t *__restrict__ s3, const uint64_t *__restrict__ n0, const uint64_t *__restrict__ n1,
                                                  const uint64_t *__restrict__ n2, const uint64_t *__restrict__ n3,
                                                  uint32_t *__restrict__ flag, uint32_t V, Img out, uint32_t start, uint32_t end) {
    const uint32_t v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    if (flag[v] == 0u) { s0[v] = n0[v]; s1[v] = n1[v]; s2[v] = n2[v]; s3[v] = n3[v]; return; }
    flag[v] = 0u;
    ZXoshiro r{s0[v], s1[v], s2[v], s3[v]};
    const float *const *no_in = nullptr;
    frame_loop<8, true, 0>(out.p, v, out.stride, no_in, nullptr, start, end, [&](uint32_t, const float (&)[1], float &val) ZH_INLINE_LAMBDA {
        val = zrandom_float32(r) * 2.0f - 1.0f;
        return true;
    });
    s0[v] = r.s0; s1[v] = r.s1; s2[v] = r.s2; s3[v] = r.s3;
}

// ------------------------------------------------------------------ plan + launch
// Frames per range for V voices over n frames, or 0 = use the sequential kernel.  Measured on MI355X
// (profiles/r02/noise_ranges.txt): enough ranges for about one wave per SIMD (1,024), at least 32 frames each.
uint32_t zh_noise_range_frames(uint32_t V, uint32_t n) {
    const long forced = zh_form(ZF_NOISE_RANGES);                        // -1 = auto, 0 = off, k = k ranges (dispatch.hip)
    if (forced == 0 || V == 0 || n < 128 || n > 2048 || V > 65536) return 0;
    const uint32_t waves = (V + 63) / 64;
    uint32_t want = forced > 0 ? (uint32_t)forced : (V <= 32768 ? 1024u : 2048u) / waves;
    if (want < 2) return 0;
    if (want > 32) want = 32;
    uint32_t ch = ((n + want - 1) / want + 31) / 32 * 32;
    const uint32_t ranges = (n + ch - 1) / ch;
    if (ranges < 2 || (ranges - 1) * (ch / 32) > (uint32_t)kNoiseJumpTables) return 0;
    return ch;
}

// out[start, end) = 0.0f + white noise for every voice, states advanced.  `next` / `flag` are the module's scratch
// (4 x n u64, n u32 zero-initialised).  Returns ZH_ERR_UNSUPPORTED when the tables are not available (first use
// inside a capture): the caller then takes the sequential kernel.
int zh_noise_paint_ranges(zh_ctx *ctx, uint64_t *const s[4], uint64_t *const next[4], uint32_t *flag, uint32_t V, const zh_buf &outb,
                          uint32_t start, uint32_t end, uint32_t ch) {
    const uint4 *tables = zh_noise_jump_tables(ctx);
    if (!tables) return ZH_ERR_UNSUPPORTED;
    NoiseChunkArgs a;
    for (int i = 0; i < 4; i++) { a.s[i] = s[i]; a.nx[i] = next[i]; }
    a.flag = flag; a.tables = tables; a.V = V; a.start = start; a.end = end; a.ch = ch; a.tstep = ch / 32;
    a.out = mk_img(outb);
    const uint32_t ranges = (end - start + ch - 1) / ch;
    ZH_LAUNCH(k_noise_white_ranges, dim3((V + 255) / 256, ranges), dim3(256), 0, ctx->stream, a);
    ZH_LAUNCH(k_noise_fix, dim3((V + 63) / 64), dim3(64), 0, ctx->stream, s[0], s[1], s[2], s[3], next[0], next[1], next[2], next[3],
                       flag, V, a.out, start, end);
    return zh_launch_status();
}

// ------------------------------------------------------------------ host self-test (no device needed)
extern "C" int zh_selftest_noise_jump(uint64_t seed, uint32_t n_states) {
    static const std::vector<uint32_t> host = build_tables();
    uint64_t sm = seed;
    auto next64 = [&]() { sm += 0x9e3779b97f4a7c15ull; uint64_t z = sm; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    int bad = 0;
    for (uint32_t t = 0; t < n_states; t++) {
        HostState r{{next64(), next64(), next64(), next64()}};
        if (t == 0) r = HostState{{1, 0, 0, 0}};
        HostState walk = r;
        for (int j = 0; j < kNoiseJumpTables; j++) {
            for (int k = 0; k < 32; k++) transition(walk);             // 32 (j + 1) transitions from r
            const uint32_t *tb = host.data() + (size_t)j * kNoiseJumpEntries * 4;
            const uint32_t w[8] = {(uint32_t)r.s[0], (uint32_t)(r.s[0] >> 32), (uint32_t)r.s[1], (uint32_t)(r.s[1] >> 32),
                                   (uint32_t)r.s[2], (uint32_t)(r.s[2] >> 32), (uint32_t)r.s[3], (uint32_t)(r.s[3] >> 32)};
            uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int pos = 0; pos < 64; pos++) {
                const uint32_t nib = (w[pos >> 3] >> ((pos & 7) * 4)) & 15u;
                for (int half = 0; half < 2; half++)
                    for (int q = 0; q < 4; q++) acc[half * 4 + q] ^= tb[((size_t)(half * 64 + pos) * 16 + nib) * 4 + q];
            }
            const uint64_t got[4] = {acc[0] | (uint64_t)acc[1] << 32, acc[2] | (uint64_t)acc[3] << 32, acc[4] | (uint64_t)acc[5] << 32, acc[6] | (uint64_t)acc[7] << 32};
            for (int q = 0; q < 4; q++) bad += got[q] != walk.s[q];
        }
    }
    return bad;
}
