// envelope.cuh -- Envelope (src/modules/Envelope.zig) over Painter (src/zang/painter.zig) as
// a per-lane, per-frame state machine, shared by the Envelope kernel and the fused voices.
//
// The reference paints stage after stage (paintToward until finished or the buffer ends,
// then the next stage continues where the last one stopped: Envelope.zig:52-70).  Here every
// lane steps its own machine once per frame so that all 64 lanes stay on the same frame
// (coalesced image rows).  Equivalence: paintToward's entry tests (`t >= 1`, instantaneous;
// painter.zig:69-80) cannot become true in the middle of a stage, so evaluating them at
// every frame equals evaluating them once per call; stages that finish WITHOUT painting a
// frame cascade within the frame, exactly like the reference falls from one `if (state ==`
// to the next; and the cascade runs once more after the last frame, because the reference
// still calls paintToward (which may finish instantly) when no frames remain.
#pragma once
#include "common.cuh"

struct CurveP {           // PaintCurve for one voice: shared tag, per-voice duration
    uint32_t tag;
    float duration;
};

struct EnvLane {
    // Envelope state (Envelope.zig:23-24; painter.zig:33-36)
    uint32_t state;
    float t, last_value, start;
    // per-paint parameters
    float sample_rate, sustain_volume;
    CurveP attack, decay, release;
    float step_attack, step_decay, step_release;   // 1 / (duration * sample_rate), painter.zig:97
    bool note_on;

    __device__ __forceinline__ void change_state(uint32_t s) {   // Envelope.zig:33-36 + painter.zig:47-50
        state = s;
        start = last_value;
        t = 0.0f;
    }

    // Prologue of paintOn / paintOff (Envelope.zig:38-50, 77-84)
    __device__ __forceinline__ void begin(bool new_note) {
        step_attack = 1.0f / (attack.duration * sample_rate);
        step_decay = 1.0f / (decay.duration * sample_rate);
        step_release = 1.0f / (release.duration * sample_rate);
        if (note_on) {
            if (new_note) change_state(ZH_ENV_ATTACK);
            // assert(state != release) at :45 is a check only; a voice in `release` that gets
            // note_on without a new note id matches none of the stage tests: paints nothing.
            if (state == ZH_ENV_IDLE) change_state(ZH_ENV_ATTACK);
        } else {
            if (state != ZH_ENV_IDLE && state != ZH_ENV_RELEASE) change_state(ZH_ENV_RELEASE);
        }
    }

    // One paintToward iteration (painter.zig:63-120).  Returns finished; sets painted/val.
    __device__ __forceinline__ bool toward(const CurveP &c, float t_step, float goal, bool have_frame,
                                           bool &painted, float &val) {
        painted = false;
        if (t >= 1.0f) return true;                               // :69-71
        if (c.tag == ZH_CURVE_INSTANTANEOUS) {                    // :76-80
            t = 1.0f;
            last_value = goal;
            return true;
        }
        if (!have_frame) return false;                            // `i < buf.len` fails: not finished
        bool finished = false;
        t += t_step;                                              // :103
        if (t >= 1.0f) { t = 1.0f; finished = true; }
        const float it = 1.0f - t;
        float tp;
        if (c.tag == ZH_CURVE_LINEAR) tp = t;
        else if (c.tag == ZH_CURVE_SQUARED) tp = 1.0f - it * it;
        else tp = 1.0f - it * it * it;
        last_value = start + tp * (goal - start);                 // :114
        val = last_value;
        painted = true;
        return finished;
    }

    // Advance by one frame (have_frame) or run the end-of-span cascade (!have_frame).
    // Returns whether a value was painted for this frame.
    __device__ __forceinline__ bool frame(bool have_frame, float &val) {
        bool painted = false;
        if (note_on) {
            if (state == ZH_ENV_ATTACK) {                         // Envelope.zig:52-60
                if (toward(attack, step_attack, 1.0f, have_frame, painted, val))
                    change_state(sustain_volume < 1.0f ? ZH_ENV_DECAY : ZH_ENV_SUSTAIN);
                if (painted) return true;
            }
            if (state == ZH_ENV_DECAY) {                          // :62-66
                if (toward(decay, step_decay, sustain_volume, have_frame, painted, val))
                    change_state(ZH_ENV_SUSTAIN);
                if (painted) return true;
            }
            if (state == ZH_ENV_SUSTAIN && have_frame) {          // :68-70 paintFlat
                val = sustain_volume;
                return true;
            }
            return false;
        }
        if (state == ZH_ENV_RELEASE) {                            // :85-89
            if (toward(release, step_release, 0.0f, have_frame, painted, val)) change_state(ZH_ENV_IDLE);
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

