// envelope.cuh -- Envelope (src/modules/Envelope.zig) over Painter (src/zang/painter.zig) as
// a per-lane state machine, shared by the Envelope kernel and the fused voices.
//
// The reference paints stage after stage (paintToward until finished or the buffer ends, then
// the next stage continues where the last one stopped: Envelope.zig:52-70).  Here every lane
// steps once per frame so that all 64 lanes stay on the same frame (coalesced image rows).
// The per-frame step is kept cheap and uniform: a lane is in one of three modes --
//   TOWARD: one paintToward loop iteration (painter.zig:102-116),
//   FLAT:   paintFlat's constant (painter.zig:53-58),
//   NONE:   nothing is painted (idle, or the assert case below) --
// and everything that takes zero time (state changes, paintToward's entry tests `t >= 1` and
// `instantaneous`, painter.zig:69-80) is resolved in resolve(), which runs exactly where the
// reference evaluates it: at the start of a paint call and whenever a stage finishes (the
// reference then immediately calls the next stage's paintToward, even with no frames left).
#pragma once
#include "common.cuh"

struct CurveP {           // PaintCurve for one voice: shared tag, per-voice duration
    uint32_t tag;
    float duration;
};

enum { ENV_MODE_NONE = 0, ENV_MODE_TOWARD = 1, ENV_MODE_FLAT = 2 };

struct EnvLane {
    // Envelope state (Envelope.zig:23-24; painter.zig:33-36)
    uint32_t state;
    float t, last_value, start;
    // per-paint parameters
    float sample_rate, sustain_volume;
    CurveP attack, decay, release;
    bool note_on;
    // the running stage
    uint32_t mode, cur_tag;
    float cur_step, cur_goal;

    __device__ __forceinline__ void change_state(uint32_t s) {   // Envelope.zig:33-36 + painter.zig:47-50
        state = s;
        start = last_value;
        t = 0.0f;
    }

    // paintToward's entry (painter.zig:69-97) for the stage `on` selects; returns "finished without
    // painting".  Written with selects only: conditional stores to different fields of the lane get
    // sunk by LLVM into one store at a variable offset, which forces the lane out of VGPRs.
    __device__ __forceinline__ bool enter(bool on, uint32_t tag, float duration, float goal) {
        const bool done = t >= 1.0f;                              // :69-71
        const bool inst = on && !done && tag == ZH_CURVE_INSTANTANEOUS;   // :76-80
        t = inst ? 1.0f : t;
        last_value = inst ? goal : last_value;
        const bool fin = done || inst;
        const bool run = on && !fin;
        mode = run ? (uint32_t)ENV_MODE_TOWARD : mode;
        cur_tag = run ? tag : cur_tag;
        cur_goal = run ? goal : cur_goal;
        cur_step = run ? 1.0f / (duration * sample_rate) : cur_step;   // :97
        return on && fin;
    }

    __device__ __forceinline__ void change_state_if(bool c, uint32_t s) {
        state = c ? s : state;
        start = c ? last_value : start;
        t = c ? 0.0f : t;
    }

    // Envelope.zig:52-70 / 85-89 from the current state, up to the next stage that takes time
    __device__ __forceinline__ void resolve() {
        mode = ENV_MODE_NONE;
        const uint32_t after_attack = sustain_volume < 1.0f ? (uint32_t)ZH_ENV_DECAY : (uint32_t)ZH_ENV_SUSTAIN;
        change_state_if(enter(note_on && state == ZH_ENV_ATTACK, attack.tag, attack.duration, 1.0f), after_attack);
        change_state_if(enter(note_on && state == ZH_ENV_DECAY, decay.tag, decay.duration, sustain_volume), ZH_ENV_SUSTAIN);
        mode = (note_on && state == ZH_ENV_SUSTAIN) ? (uint32_t)ENV_MODE_FLAT : mode;
        // note_on && state == release is the assert case of Envelope.zig:45 (note_on without a new note
        // id while releasing): with the assert compiled out nothing matches -> NONE.
        change_state_if(enter(!note_on && state == ZH_ENV_RELEASE, release.tag, release.duration, 0.0f), ZH_ENV_IDLE);
    }

    // Prologue of paintOn / paintOff (Envelope.zig:38-50, 77-84), then the first stage's entry
    __device__ __forceinline__ void begin(bool new_note) {
        change_state_if(note_on && new_note, ZH_ENV_ATTACK);
        change_state_if(note_on && state == ZH_ENV_IDLE, ZH_ENV_ATTACK);
        change_state_if(!note_on && state != ZH_ENV_IDLE && state != ZH_ENV_RELEASE, ZH_ENV_RELEASE);
        resolve();
    }

    // One frame.  Returns whether a value was painted.  Straight-line: the paintToward step
    // (painter.zig:102-116) is computed unconditionally and committed by selects -- per-lane mode
    // branches cost more (exec-mask bookkeeping) than the ~10 VALU ops they would skip; only the
    // rare stage end branches.
    __device__ __forceinline__ bool frame(float &val) {
        const bool toward = mode == ENV_MODE_TOWARD;
        float tn = t + cur_step;
        const bool finished = tn >= 1.0f;
        tn = finished ? 1.0f : tn;
        const float it = 1.0f - tn;
        float tp = tn;
        if (cur_tag == ZH_CURVE_SQUARED) tp = 1.0f - it * it;
        else if (cur_tag == ZH_CURVE_CUBED) tp = 1.0f - it * it * it;
        const float lv = start + tp * (cur_goal - start);         // :114
        t = toward ? tn : t;
        last_value = toward ? lv : last_value;
        val = toward ? lv : sustain_volume;                        // FLAT: Envelope.zig:68-70
        const bool painted = mode != ENV_MODE_NONE;
        if (toward && finished) {                                  // Envelope.zig:53-58, 63-65, 86-88
            const uint32_t after_attack = sustain_volume < 1.0f ? (uint32_t)ZH_ENV_DECAY : (uint32_t)ZH_ENV_SUSTAIN;
            const uint32_t next = state == ZH_ENV_ATTACK ? after_attack
                                  : (state == ZH_ENV_DECAY ? (uint32_t)ZH_ENV_SUSTAIN : (uint32_t)ZH_ENV_IDLE);
            change_state(next);
            resolve();
        }
        return painted;
    }
};

// Envelope.Params (Envelope.zig:6-13) as the kernels see them: tags shared, values per voice.
struct EnvParamsP {
    float sample_rate;
    uint32_t attack_tag, decay_tag, release_tag;
    F32P attack_dur, decay_dur, release_dur, sustain_volume;
    BoolP note_on;
};

__device__ __forceinline__ void env_load(EnvLane &e, const EnvParamsP &p, uint32_t v) {
    e.sample_rate = p.sample_rate;
    e.sustain_volume = p.sustain_volume.get(v);
    e.attack = CurveP{p.attack_tag, p.attack_dur.get(v)};
    e.decay = CurveP{p.decay_tag, p.decay_dur.get(v)};
    e.release = CurveP{p.release_tag, p.release_dur.get(v)};
    e.note_on = p.note_on.get(v);
}

