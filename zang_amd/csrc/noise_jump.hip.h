// noise_jump.hip.h -- xoshiro256++ jump-ahead, so that one voice's white noise can be generated for many frame
// ranges at once (few voices = few waves: the sequential walk of a 1024-frame span is the whole kernel time then).
//
// The generator's state transition (zxoshiro_next without the output, zmath.hip.h; Zig std Xoshiro256.next) is linear
// over GF(2): state' = T * state for a fixed 256 x 256 bit matrix T, so the state k draws ahead is T^k * state.  For
// k = 32 j (j = 1..63) the library keeps T^k as a nibble table: tbl[half][pos][nib] = the 128-bit half `half` of
// T^k * (the state that has nibble `nib` at nibble position `pos` and zeros elsewhere); T^k * state is the XOR of the 64
// entries picked by the state's 64 nibbles -- 128 LDS reads of 16 bytes and as many XORs, instead of k sequential
// transitions.  The 16 entries of one (half, pos) fill exactly one 256-byte LDS bank row, so the lanes of a
// ds_read_b128 group never conflict whatever their nibbles are.
//
// Noise.zig:51,58 draws ONE u64 per sample through Random.float(f32) -- except when the draw has 41 or more leading
// zeros (probability 2^-41), when float() draws again.  A frame range generated from a jumped state assumes one draw
// per earlier frame; every range reports such an event per voice, and the caller repairs a flagged voice by walking its
// whole span sequentially (noise_jump.hip).
#pragma once
#include "common.hip.h"
#include "zmath.hip.h"

constexpr int kNoiseJumpTables = 63;                 // T^(32 j), j = 1..63: frame ranges may start up to 2016 frames in
constexpr int kNoiseJumpEntries = 2 * 64 * 16;       // uint4 per table (32 KiB)

#if !defined(ZH_DEVICE_ONLY)
// device pointer to the kNoiseJumpTables tables of this context ([table][half][pos][nib] uint4), built and uploaded on
// first use; nullptr on failure.  Not to be called for the first time while the stream is capturing.
const uint4 *zh_noise_jump_tables(zh_ctx *ctx);
#endif

// cooperative copy of one table into LDS by a workgroup of `nthreads` (a divisor of kNoiseJumpEntries); barrier after
__device__ __forceinline__ void noise_jump_load(uint4 *lds, const uint4 *__restrict__ table, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < (uint32_t)kNoiseJumpEntries; i += nthreads) lds[i] = table[i];
}

// r = T^k * r with T^k's table in LDS.  The 128 reads go out in four batches of 32 (16 nibble positions each) before any
// of a batch is consumed -- left to itself the compiler issues a pair of reads and waits for it, 64 exposed LDS
// latencies per jump (measured: ~10,000 cycles per jump against ~3,000 in this form).
__device__ __forceinline__ void noise_jump_apply(ZXoshiro &r, const uint4 *lds) {
    const uint32_t w[8] = {(uint32_t)r.s0, (uint32_t)(r.s0 >> 32), (uint32_t)r.s1, (uint32_t)(r.s1 >> 32),
                           (uint32_t)r.s2, (uint32_t)(r.s2 >> 32), (uint32_t)r.s3, (uint32_t)(r.s3 >> 32)};
    uint4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p0 = 0; p0 < 64; p0 += 16) {
        uint4 x[16], y[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int pos = p0 + q;
            const uint32_t nib = (w[pos >> 3] >> ((pos & 7) * 4)) & 15u;
            x[q] = lds[pos * 16 + nib];
            y[q] = lds[64 * 16 + pos * 16 + nib];
        }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);                           // nothing moves across: every read of the batch is in flight first
#endif
#pragma unroll
        for (int q = 0; q < 16; q++) {
            a.x ^= x[q].x; a.y ^= x[q].y; a.z ^= x[q].z; a.w ^= x[q].w;
            b.x ^= y[q].x; b.y ^= y[q].y; b.z ^= y[q].z; b.w ^= y[q].w;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    r.s0 = (uint64_t)a.x | ((uint64_t)a.y << 32); r.s1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
    r.s2 = (uint64_t)b.x | ((uint64_t)b.y << 32); r.s3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
}

// zrandom_float32 (zmath.hip.h) that also says whether the multi-draw branch ran
__device__ __forceinline__ float zrandom_float32_multi(ZXoshiro &r, bool &multi) {
    const uint64_t rnd = zxoshiro_next(r);
    const uint32_t hi = (uint32_t)(rnd >> 32);
    uint32_t lz;
    asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(hi));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hi == 0u) != 0, 0)) {
        if (hi == 0u) {
            lz = rnd ? (uint32_t)__clzll((long long)rnd) : 64u;
            if (lz >= 41) {
                multi = true;
                uint64_t r2 = zxoshiro_next(r);
                lz = 41 + (r2 ? (uint32_t)__clzll((long long)r2) : 64u);
                if (lz == 41 + 64) lz += (uint32_t)__clz((int)((uint32_t)zxoshiro_next(r) | 0x7FFu));
            }
        }
    }
    return zu2f(((126u - lz) << 23) | ((uint32_t)rnd & 0x7FFFFFu));
}

// The common case of zrandom_float32 without its test: valid whenever the draw's high word is non-zero; `hmin` collects the
// minimum high word so that a caller can test a whole tile at once (and redo it with the careful form when it is 0).
__device__ __forceinline__ float zrandom_float32_common(ZXoshiro &r, uint32_t &hmin) {
    const uint64_t rnd = zxoshiro_next(r);
    const uint32_t hi = (uint32_t)(rnd >> 32);
    hmin = hi < hmin ? hi : hmin;
    uint32_t lz;
    asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(hi));
    return zu2f(((126u - lz) << 23) | ((uint32_t)rnd & 0x7FFFFFu));
}
