// nice_mix_fma.hip -- the fused NiceInstrument voices + voice mixdown kernels of nice_mix.hip.h compiled with
// -ffp-contract=fast (Makefile: this file only): every `a * b + c` of the voice -- the filter's six updates (Filter.zig:138-144),
// the envelope's `start + tp * (goal - start)` (painter.zig:107-117), the oscillator's polynomials (PulseOsc.zig:102-110), the
// product with the envelope and the gain products of the stereo sum -- becomes one v_fma_f32: 132 instead of 180 instructions
// per four frames of the sustain loop, 405 instead of 513 in the loop with the envelope.  NOT the reference's bits: a fused
// multiply-add rounds once where the reference rounds twice.  Taken only by a paint flagged ZH_PAINT_TOLERANT (include/zang_hip.h:
// within 1e-5 of the voice's peak) above nice_tp_max voices, where the time-parallel forms do not apply and the exact kernel
// is bound by the instructions it issues (DESIGN.md 7, config-5 shard).  The integer state (phase counter, envelope stage) and
// the envelope's clock -- sums only -- stay the reference's bits.
#define ZH_K(name) name##_fma
#include "nice_mix.hip.h"

void zh_nice_mix_fma_launch(int channels, bool roll, int nw, uint32_t blocks, hipStream_t st, const NiceArgs &a, uint32_t start, uint32_t end,
                            float *part, F32P gl, F32P gr) {
    const dim3 g(blocks), b(256);
#define ZH_ONE(C_, ROLL_, NW_) ZH_LAUNCH((k_nice_mix_fma<C_, ROLL_, NW_>), g, b, 0, st, a, start, end, part, gl, gr)
#define ZH_ROLL(C_, NW_) do { if (roll) ZH_ONE(C_, true, NW_); else ZH_ONE(C_, false, NW_); } while (0)
    if (channels == 2) { if (nw) ZH_ROLL(2, 4); else ZH_ROLL(2, 0); }
    else { if (nw) ZH_ROLL(1, 4); else ZH_ROLL(1, 0); }
#undef ZH_ROLL
#undef ZH_ONE
}

void zh_nice_mix_batch_fma_launch(bool roll, int nw, uint32_t blocks, hipStream_t st, const NiceBatchArgs &bt, uint32_t start, uint32_t end,
                                  float *part, F32P gl, F32P gr) {
    const dim3 g(blocks), b(256);
#define ZH_ONE(ROLL_, NW_) ZH_LAUNCH((k_nice_mix_batch_fma<2, ROLL_, NW_>), g, b, 0, st, bt, start, end, part, gl, gr)
    if (nw) { if (roll) ZH_ONE(true, 4); else ZH_ONE(false, 4); }
    else { if (roll) ZH_ONE(true, 0); else ZH_ONE(false, 0); }
#undef ZH_ONE
}
