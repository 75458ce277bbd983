// nice_mix.hip.h -- NiceInstrument voices + voice mixdown in one kernel (k_nice_mix, k_nice_mix_batch).  Included by composite.hip
// (exact: the bits of the unfused reference composition) and by nice_mix_fma.hip, which is compiled with -ffp-contract=fast and names
// its kernels k_nice_mix_fma / k_nice_mix_batch_fma through ZH_K (the ZH_PAINT_TOLERANT form above nice_tp_max voices).
#pragma once
#include "nice.hip.h"
#ifndef ZH_K
#define ZH_K(name) name
#endif
// Fused chain + voice mixdown.  Every WAVE (64 voices) works on its own: lanes render MIXF frames into the wave's LDS tile
// [frame][lane] (row stride 68 floats: column writes hit 64 consecutive banks, and the 16-byte row reads of eight
// neighbouring lanes start 4 banks apart -- both conflict-free), then every lane sums half a row -- lane
// (f, h) adds voices 32h..32h+31 of frame f left to right -- the two halves of a frame are added (h0 + h1) across the wave
// and lanes 0..MIXF-1 write partials[wave][frame].  One ds_write + one ds_read + one add per lane-frame, instead of a 6-step
// cross-lane butterfly per frame.  Round 3: nothing crosses waves any more -- the round-2 form combined the four waves of a
// workgroup through LDS behind two __syncthreads per 32-frame chunk, and at the config-5 shard size (131,072 voices = two
// waves per SIMD, each at its one-instruction-per-5-cycles issue limit) a wave waiting at a barrier is issue time nobody
// else can use.  The second pass (basics.hip k_mix_pass2_wide) adds the wave partials in wave order, both channels in one
// launch.  Fixed order => reproducible bits.
constexpr int MIXF = 32;
constexpr int MIXS = 68;      // floats per tile row: 16-byte aligned rows, and 68 = 4 mod 32 keeps both access patterns conflict-free

// C = output channels.  C = 1: partials[block][frame] = sum of the block's voices.  C = 2 (stereo,
// examples/example_stereo.zig:92-98: `outputs[c] += voice * pan_c` per voice): the sum phase multiplies each
// voice's sample by that voice's channel gain first -- a lane's 32 voices are the same in every chunk, so their
// 2 x 32 gains sit in registers -- and partials are [channel][block][frame].
// ROLL: the oscillator carries the previous frame's half-period bit as a lane mask (dsp.hip.h pulse_sample_roll).
// every lane needs the gains of the 32 voices it adds up: the workgroup's 256 pairs go through LDS once (two coalesced loads
// per lane instead of 64 scattered ones: 2 us per launch at 131,072 voices) -- the kernel's only workgroup barrier
template <int C, class G2>
__device__ __forceinline__ void nice_mix_gains(G2 &g2, const F32P &gain_l, const F32P &gain_r, bool live, uint32_t v, uint32_t wave, uint32_t rh) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    if constexpr (C == 2) {
        __shared__ f2 gains[512];                                       // (a workgroup is 256 threads, or 512: NW = 8)
        gains[threadIdx.x] = live ? f2{gain_l.get(v), gain_r.get(v)} : f2{0.0f, 0.0f};   // (voices past the last: tile entries are 0.0f)
        __syncthreads();
        const f2 *mine = &gains[wave * 64 + rh * 32];                   // the 32 voices this lane adds up
#pragma unroll
        for (int j = 0; j < 32; j++) g2[j] = mine[j];
    }
}

// One paint's frames of a wave: chunks of MIXF frames into the wave's tile, summed, partial rows written.  `pw` = this wave's
// partial rows of the paint: `pw_paint` = the paint's channel-0 block laid out [frame / G][row][frame % G], G = kMixGroupFrames (a second-pass
// workgroup's G frames of every row are one contiguous run), channel 1 channel_stride floats further; `wrow` = this wave's row.
// WG (round 4): the four waves' row sums of a chunk meet in LDS and ONE row per workgroup goes to HBM -- a quarter of the partial
// rows written and read back -- for one workgroup barrier per 32-frame chunk, in the sum phase only: `wsum`
// = [2][4][C][MIXF], the chunk's parity picks the half (a wave may be a whole chunk ahead of the slowest: it has passed the
// previous barrier, so every wave has finished combining the chunk before that).  `wrow` is then the workgroup's row and
// `rows` the number of workgroups.  Row order inside a workgroup: ((w0 + w1) + w2) + w3 ...  NW = waves per workgroup: 0 = no combine
// (one row per wave), 4, or 8 (512-thread workgroups: an eighth of the rows; ZH_NICE_MIX_WG8_MIN).
template <int C, bool ROLL, int NW, class G2>
__device__ __forceinline__ void nice_mix_frames(NiceLane &n, PulseRoll &roll, const G2 &g2, float (*tile)[MIXS], float *__restrict__ pw_paint,
                                                size_t channel_stride, uint32_t rows, uint32_t wrow, uint32_t start, uint32_t end, uint32_t lane,
                                                uint32_t rf, uint32_t rh, float (*wsum)[NW ? NW : 4][C][MIXF] = nullptr, uint32_t wave = 0, uint32_t first = 0xFFFFFFFFu) {
    constexpr bool WG = NW != 0;
    // `first` (a chunk of a time-parallel paint, k_nice_mix_tp_b): the frames [first, end) of a span that starts at `start` -- the
    // partial rows are indexed from the span's start
    uint32_t parity = 0;
    for (uint32_t f0 = first == 0xFFFFFFFFu ? start : first; f0 < end; f0 += MIXF) {
        // a stage can only end inside a chunk, never begin: a wave with no voice in a timed stage at the chunk's first frame
        // (the 18 sustain buffers of a held note, an idle voice) skips the envelope for the whole chunk
        const bool whole = f0 + MIXF <= end;                            // (uniform)
        if (!__any(n.env.mode == ENV_MODE_TOWARD)) {
            const float e0 = n.env_quiet();
            if (whole) {
#pragma unroll 4
                for (int k = 0; k < MIXF; k++) {
                    const float x = 0.0f + (ROLL ? n.template frame_quiet_roll<true>(e0, roll) : n.template frame_quiet<true>(e0));
                    tile[k][lane] = x;
                }
            } else {
                for (int k = 0; k < MIXF; k++) {
                    float x = 0.0f;
                    if (__builtin_amdgcn_readfirstlane((int)(f0 + k < end))) x = 0.0f + (ROLL ? n.template frame_quiet_roll<true>(e0, roll) : n.template frame_quiet<true>(e0));
                    tile[k][lane] = x;
                }
            }
        } else if (whole && n.env.quiet(MIXF)) {                          // no stage can end in this chunk: the envelope without its stage-end test
            if (__all(n.env.mode == ENV_MODE_TOWARD)) {                   // ... and every voice is inside a stage: nothing to select
#pragma unroll 4
                for (int k = 0; k < MIXF; k++) {
                    const float e0 = n.env.frame_masked_all_toward_quiet();
                    const float x = 0.0f + (ROLL ? n.template frame_quiet_roll<true>(e0, roll) : n.template frame_quiet<true>(e0));
                    tile[k][lane] = x;
                }
            } else {
#pragma unroll 4
                for (int k = 0; k < MIXF; k++) {
                    const float e0 = n.env.frame_masked_quiet();
                    const float x = 0.0f + (ROLL ? n.template frame_quiet_roll<true>(e0, roll) : n.template frame_quiet<true>(e0));
                    tile[k][lane] = x;
                }
            }
        } else if (whole) {
#pragma unroll 4
            for (int k = 0; k < MIXF; k++) {
                const float x = 0.0f + (ROLL ? n.template frame_roll<true>(roll) : n.template frame<true>());   // the voice's own out (zeroed) += env*flt
                tile[k][lane] = x;
            }
        } else {
            for (int k = 0; k < MIXF; k++) {
                float x = 0.0f;
                if (__builtin_amdgcn_readfirstlane((int)(f0 + k < end))) x = 0.0f + (ROLL ? n.template frame_roll<true>(roll) : n.template frame<true>());
                tile[k][lane] = x;
            }
        }
        // the tile is this wave's own: its LDS writes above and reads below execute in program order, no workgroup barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            // the lane's half row: 32 voices = eight 16-byte LDS reads
            const float4 *row4 = reinterpret_cast<const float4 *>(&tile[rf][rh * 32]);
            float row[32];
#pragma unroll
            for (int q = 0; q < 8; q++) { const float4 x = row4[q]; row[4 * q] = x.x; row[4 * q + 1] = x.y; row[4 * q + 2] = x.z; row[4 * q + 3] = x.w; }
            float sl, sr = 0.0f;
            if constexpr (C == 1) {
                sl = row[0];
#pragma unroll
                for (int j = 1; j < 32; j++) sl += row[j];
            } else {
                // zang.multiply: dest += a * b, the product rounded, then added (no fused multiply-add).  (Both channels in
                // packed v_pk_mul_f32 / v_pk_add_f32 -- half the instructions -- measured SLOWER: 124 -> 135 us per buffer at
                // 131,072 voices; the 31 dependent packed adds cost more than the two interleaved scalar chains.)
                sl = row[0] * g2[0].x; sr = row[0] * g2[0].y;
#pragma unroll
                for (int j = 1; j < 32; j++) { sl += row[j] * g2[j].x; sr += row[j] * g2[j].y; }
            }
            // frame rf's two half-row sums meet in lane rf: (voices 0..31) + (voices 32..63)
            const float hl = __shfl_down(sl, 32);
            const float hr = C == 2 ? __shfl_down(sr, 32) : 0.0f;
            if constexpr (!WG) {
                if (rh == 0 && f0 + rf < end) {
                    const uint32_t fr = (f0 - start) + rf;
                    float *pw = pw_paint + ((size_t)(fr / kMixGroupFrames) * rows + wrow) * kMixGroupFrames + (fr % kMixGroupFrames);
                    pw[0] = sl + hl;
                    if constexpr (C == 2) pw[channel_stride] = sr + hr;
                }
            } else {
                if (rh == 0) {
                    wsum[parity][wave][0][rf] = sl + hl;
                    if constexpr (C == 2) wsum[parity][wave][1][rf] = sr + hr;
                }
                __syncthreads();
                // wave w combines the chunk's frames FW w .. FW w + FW - 1 (FW = 8 with four waves, 4 with eight): lanes 0 .. FW-1 the
                // first channel, FW .. 2 FW - 1 the second; FW lanes = 4 FW contiguous bytes of the workgroup's row
                constexpr uint32_t FW = MIXF / (NW ? NW : 4);
                static_assert(kMixGroupFrames % FW == 0, "a wave's frames stay inside one frame group");
                const uint32_t q = lane % FW, c = lane / FW, fw = wave * FW + q;
                if (c < (uint32_t)C && f0 + fw < end) {
                    float t = wsum[parity][0][c][fw] + wsum[parity][1][c][fw];
#pragma unroll
                    for (int w = 2; w < (NW ? NW : 4); w++) t += wsum[parity][w][c][fw];          // ((w0 + w1) + w2) + ...
                    const uint32_t fr = (f0 - start) + fw;
                    pw_paint[c * channel_stride + ((size_t)(fr / kMixGroupFrames) * rows + wrow) * kMixGroupFrames + (fr % kMixGroupFrames)] = t;
                }
                parity ^= 1u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the next chunk rewrites the tile after these reads
        __builtin_amdgcn_wave_barrier();
    }
}

// the silent stand-in of a lane past the last voice (it runs voice V-1's params): contributes +0.0 by construction rather than
// through a select per frame -- a silent oscillator, a filter at rest (finite whatever voice V-1's state is) and an envelope
// that paints nothing: frame_masked() ANDs its value with m_painted, a mode of NONE never leaves NONE inside a paint, and
// 0.0f + (+0.0f * finite) = +0.0f
__device__ __forceinline__ void nice_silence(NiceLane &n) {
    n.k.ifreq = 0u; n.k.brpt = 0u; n.k.gdf2 = 0.0f; n.k.cc121 = 0.0f; n.k.cc212 = 0.0f; n.g = n.ng = 0.0f;   // (as begin() silences a bad frequency)
    n.l = n.b = 0.0f;
    n.env.mode = ENV_MODE_NONE; n.env.m_painted = 0u;
}

template <int C, bool ROLL, int NW = 0>
__global__ void __launch_bounds__(NW == 8 ? 512 : 256) ZH_K(k_nice_mix)(NiceArgs a, uint32_t start, uint32_t end, float *__restrict__ partials,
                                                                  F32P gain_l, F32P gain_r) {
    constexpr bool WG = NW != 0;
    constexpr int NWV = NW ? NW : 4;                                    // waves per workgroup
    __shared__ float tile_all[NWV][MIXF][MIXS];
    __shared__ float wsum[WG ? 2 : 1][NWV][C][MIXF];
    const uint32_t v = blockIdx.x * (NWV * 64) + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*tile)[MIXS] = tile_all[wave];                               // this wave's tile: no other wave touches it
    const uint32_t nframes = end - start;
    const uint32_t wave_global = WG ? blockIdx.x : blockIdx.x * NWV + wave;
    const uint32_t rows = WG ? gridDim.x : gridDim.x * NWV;
    const size_t channel_stride = (size_t)((nframes + kMixGroupFrames - 1) / kMixGroupFrames) * rows * kMixGroupFrames;   // partials[channel][frame / G][wave][frame % G]
    const bool live = v < a.V;
    // lanes past the last voice run voice V-1 again and contribute 0.0f: the frame loop below then needs no per-lane
    // exec-mask region (an s_and_saveexec / branch / restore per frame: ~10 of ~85 instructions)
    NiceLane n;
    nice_load(n, a, live ? v : a.V - 1);
    if (!live) nice_silence(n);
    const uint32_t rf = lane & (MIXF - 1), rh = lane >> 5;              // this lane's row / half in the sum phase
    PulseRoll roll;
    n.roll_begin(roll);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 g2[C == 2 ? 32 : 1];                                             // {left, right} gain of each of the lane's 32 voices
    nice_mix_gains<C>(g2, gain_l, gain_r, live, v, wave, rh);
    nice_mix_frames<C, ROLL, NW>(n, roll, g2, tile, partials, channel_stride, rows, wave_global, start, end, lane, rf, rh, wsum, wave);
    if (live) nice_store(n, a, v);
}

// n_buffers consecutive paints of k_nice_mix in one launch (zh_nice_paint_mix_stereo_batch): the state words stay in
// registers from buffer to buffer -- what a paint stores and the next one loads -- and begin() runs per buffer with that
// buffer's params, exactly as separate launches would run it.  partials[buffer][channel][wave][frame].
constexpr int kNiceMixMaxBatch = 16;
struct NiceBatchArgs {
    NiceArgs a;                                                        // state arrays, V, sample rate; freq / note_on / nic of buffer 0 unused
    F32P freq[kNiceMixMaxBatch];
    BoolP note_on[kNiceMixMaxBatch], nic[kNiceMixMaxBatch];
    uint32_t nb;
};
template <int C, bool ROLL, int NW = 0>
__global__ void __launch_bounds__(NW == 8 ? 512 : 256) ZH_K(k_nice_mix_batch)(const NiceBatchArgs b, uint32_t start, uint32_t end, float *__restrict__ partials,
                                                                        F32P gain_l, F32P gain_r) {
    constexpr bool WG = NW != 0;
    constexpr int NWV = NW ? NW : 4;
    __shared__ float tile_all[NWV][MIXF][MIXS];
    __shared__ float wsum[WG ? 2 : 1][NWV][C][MIXF];
    const NiceArgs &a = b.a;
    const uint32_t v = blockIdx.x * (NWV * 64) + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*tile)[MIXS] = tile_all[wave];
    const uint32_t nframes = end - start;
    const uint32_t wave_global = WG ? blockIdx.x : blockIdx.x * NWV + wave;
    const uint32_t rows = WG ? gridDim.x : gridDim.x * NWV;
    const size_t channel_stride = (size_t)((nframes + kMixGroupFrames - 1) / kMixGroupFrames) * rows * kMixGroupFrames;
    const bool live = v < a.V;
    const uint32_t vc = live ? v : a.V - 1;
    NiceLane n;
    nice_load_state(n, a, vc);
    const float color = a.color[vc];
    const uint32_t rf = lane & (MIXF - 1), rh = lane >> 5;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 g2[C == 2 ? 32 : 1];
    nice_mix_gains<C>(g2, gain_l, gain_r, live, v, wave, rh);
    for (uint32_t k = 0; k < b.nb; k++) {
        n.begin(a.sample_rate, a.srf, a.sr8, b.freq[k].get(vc), color, b.note_on[k].get(vc), b.nic[k].get(vc));
        if (!live) nice_silence(n);
        PulseRoll roll;
        n.roll_begin(roll);
        nice_mix_frames<C, ROLL, NW>(n, roll, g2, tile, partials + (size_t)k * C * channel_stride, channel_stride, rows, wave_global, start, end, lane, rf, rh, wsum, wave);
    }
    if (live) nice_store(n, a, v);
}

// nice_mix_fma.hip: the same two kernels with contraction on (k_nice_mix_fma, k_nice_mix_batch_fma); nw = 0 or 4 (nice_mix_wg)
void zh_nice_mix_fma_launch(int channels, bool roll, int nw, uint32_t blocks, hipStream_t st, const NiceArgs &a, uint32_t start, uint32_t end,
                            float *part, F32P gl, F32P gr);
void zh_nice_mix_batch_fma_launch(bool roll, int nw, uint32_t blocks, hipStream_t st, const NiceBatchArgs &bt, uint32_t start, uint32_t end,
                                  float *part, F32P gl, F32P gr);
