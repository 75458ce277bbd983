// envelope.hip.h -- Envelope (src/modules/Envelope.zig) over Painter (src/zang/painter.zig) as
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
#include "common.hip.h"
#include "lanes.hip.h"

// One voice per wave: the running sum acc_{j+1} = acc_j + step over frames [j0, nf) of a 64-frame block,
// computed once per wave (every lane holds the same acc) and handed out through 64 floats of LDS owned by
// the wave: lane j gets acc before (PRE) or after frame j's add.  Two instructions per frame (the add and
// the LDS write) -- the whole inherently sequential part of a phase accumulator or an envelope's clock.
template <bool PRE>
__device__ __forceinline__ float zwalk64(float &acc, float step, uint32_t j0, uint32_t nf, uint32_t lane, float *scratch) {
    float a = acc;
    uint32_t j = j0;
    for (; j + 8 <= nf; j += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (PRE) scratch[j + q] = a;
            a = a + step;
            if (!PRE) scratch[j + q] = a;
        }
    }
    for (; j < nf; j++) {
        if (PRE) scratch[j] = a;
        a = a + step;
        if (!PRE) scratch[j] = a;
    }
    acc = a;
    return scratch[lane & 63];
}

template <int W>
struct CurvePT {          // PaintCurve for one lane: shared tag, per-voice duration
    uint32_t tag;
    typename LaneT<W>::F duration;
};
using CurveP = CurvePT<1>;

enum { ENV_MODE_NONE = 0, ENV_MODE_TOWARD = 1, ENV_MODE_FLAT = 2 };

// W voices per lane (lanes.hip.h).  Everything is selects under masks: conditional stores to
// different fields of the lane get sunk by LLVM into one store at a variable offset, which forces
// the lane out of VGPRs -- and with W = 2 the two voices of a lane are rarely in the same stage.
// FT >= 0: all three curves are known at compile time to have tag FT (the composite instruments use cubed
// throughout, examples/modules.zig:118-125, 238-245), so the per-frame curve needs no tag compares or selects.
template <int W, int FT = -1>
struct EnvLaneT {
    using F = typename LaneT<W>::F;
    using U = typename LaneT<W>::U;
    using M = typename LaneT<W>::M;
    // Envelope state (Envelope.zig:23-24; painter.zig:33-36)
    U state;
    F t, last_value, start;
    // per-paint parameters
    float sample_rate;
    F sustain_volume;
    CurvePT<W> attack, decay, release;
    M note_on;
    // the running stage
    U mode, cur_tag;
    F cur_step, cur_goal;
    // derived, refreshed wherever mode / start / cur_goal change (begin, stage end): cur_goal - start, and an
    // all-ones / zero word per voice for mode != NONE (frame_masked)
    F cur_delta;
    U m_painted;

    static __device__ __forceinline__ U u(uint32_t x) { return zsplatu<U>(x); }
    static __device__ __forceinline__ F f(float x) { return zsplat<F>(x); }

    // Envelope.zig:33-36 + painter.zig:47-50, for the voices in `c`
    __device__ __forceinline__ void change_state_if(M c, U s) {
        state = zsel(c, s, state);
        start = zsel(c, last_value, start);
        t = zsel(c, f(0.0f), t);
    }

    // paintToward's entry (painter.zig:69-97) for the voices in `on`; returns "finished without painting"
    __device__ __forceinline__ M enter(M on, uint32_t tag, F duration, F goal) {
        const M done = t >= f(1.0f);                              // :69-71
        const M inst = zand(on, znot(done), zmask<M>(FT < 0 && tag == ZH_CURVE_INSTANTANEOUS));   // :76-80
        t = zsel(inst, f(1.0f), t);
        last_value = zsel(inst, goal, last_value);
        const M fin = zor(done, inst);
        const M run = zand(on, znot(fin));
        mode = zsel(run, u(ENV_MODE_TOWARD), mode);
        cur_tag = zsel(run, u(tag), cur_tag);
        cur_goal = zsel(run, goal, cur_goal);
        cur_step = zsel(run, f(1.0f) / (duration * sample_rate), cur_step);   // :97
        return zand(on, fin);
    }

    // Envelope.zig:52-70 / 85-89 from the current state, up to the next stage that takes time,
    // for the voices in `m`
    __device__ __forceinline__ void resolve(M m) {
        mode = zsel(m, u(ENV_MODE_NONE), mode);
        const U after_attack = zsel(sustain_volume < f(1.0f), u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN));
        const M on = zand(m, note_on), off = zand(m, znot(note_on));
        change_state_if(enter(zand(on, state == u(ZH_ENV_ATTACK)), attack.tag, attack.duration, f(1.0f)), after_attack);
        change_state_if(enter(zand(on, state == u(ZH_ENV_DECAY)), decay.tag, decay.duration, sustain_volume), u(ZH_ENV_SUSTAIN));
        mode = zsel(zand(on, state == u(ZH_ENV_SUSTAIN)), u(ENV_MODE_FLAT), mode);
        // note_on && state == release is the assert case of Envelope.zig:45 (note_on without a new note
        // id while releasing): with the assert compiled out nothing matches -> NONE.
        change_state_if(enter(zand(off, state == u(ZH_ENV_RELEASE)), release.tag, release.duration, f(0.0f)), u(ZH_ENV_IDLE));
    }

    // Prologue of paintOn / paintOff (Envelope.zig:38-50, 77-84), then the first stage's entry
    __device__ __forceinline__ void begin(M new_note) {
        change_state_if(zand(note_on, new_note), u(ZH_ENV_ATTACK));
        change_state_if(zand(note_on, state == u(ZH_ENV_IDLE)), u(ZH_ENV_ATTACK));
        change_state_if(zand(znot(note_on), state != u(ZH_ENV_IDLE), state != u(ZH_ENV_RELEASE)), u(ZH_ENV_RELEASE));
        resolve(zmask<M>(true));
        refresh_derived();
    }
    __device__ __forceinline__ void refresh_derived() {
        cur_delta = cur_goal - start;
        m_painted = zsel(mode != u(ENV_MODE_NONE), u(0xFFFFFFFFu), u(0u));
    }
    // the curve of the running stage at clock tn (painter.zig:104-112)
    __device__ __forceinline__ F curve(F tn) const {
        const F it = f(1.0f) - tn;
        if constexpr (FT == ZH_CURVE_CUBED) return f(1.0f) - it * it * it;
        else if constexpr (FT == ZH_CURVE_SQUARED) return f(1.0f) - it * it;
        else if constexpr (FT == ZH_CURVE_LINEAR) return tn;
        else return zsel(cur_tag == u(ZH_CURVE_SQUARED), f(1.0f) - it * it,
                         zsel(cur_tag == u(ZH_CURVE_CUBED), f(1.0f) - it * it * it, tn));
    }

    // One frame.  Returns which voices painted a value.  Straight-line: the paintToward step
    // (painter.zig:102-116) is computed unconditionally and committed by selects -- per-lane mode
    // branches cost more (exec-mask bookkeeping) than the ~10 VALU ops they would skip; only the
    // rare stage end branches.
    __device__ __forceinline__ M frame(F &val) {
        const M toward = mode == u(ENV_MODE_TOWARD);
        F tn = t + cur_step;
        const M finished = tn >= f(1.0f);
        tn = zsel(finished, f(1.0f), tn);
        const F tp = curve(tn);
        const F lv = start + tp * cur_delta;                       // :114 (cur_delta == cur_goal - start)
        t = zsel(toward, tn, t);
        last_value = zsel(toward, lv, last_value);
        val = zsel(toward, lv, sustain_volume);                    // FLAT: Envelope.zig:68-70
        const M painted = mode != u(ENV_MODE_NONE);
        const M stage_end = zand(toward, finished);
        if (zany(stage_end)) {                                     // Envelope.zig:53-58, 63-65, 86-88
            const U after_attack = zsel(sustain_volume < f(1.0f), u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN));
            const U next = zsel(state == u(ZH_ENV_ATTACK), after_attack,
                                zsel(state == u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN), u(ZH_ENV_IDLE)));
            change_state_if(stage_end, next);
            resolve(stage_end);
            refresh_derived();
        }
        return painted;
    }
    // frame() with the value discarded, for a replay that walks several stages (a generated script kernel's frame range,
    // script_rt.hip.h): the clock and the stage ends as in frame(); last_value, which frame() rewrites every TOWARD frame, is
    // only brought up to date where the replay reads it -- at a stage end (change_state_if: start = last_value).  Valid where
    // begin() is not called between the replayed frames (begin() reads last_value too) and a frame() follows the walk before
    // the state is stored (it rewrites last_value of every TOWARD voice; the others' is untouched by either form).
    __device__ __forceinline__ void frame_walk() {
        const M toward = mode == u(ENV_MODE_TOWARD);
        F tn = t + cur_step;
        const M finished = tn >= f(1.0f);
        tn = zsel(finished, f(1.0f), tn);
        t = zsel(toward, tn, t);
        const M stage_end = zand(toward, finished);
        if (zany(stage_end)) {
            last_value = zsel(stage_end, start + curve(tn) * cur_delta, last_value);
            const U after_attack = zsel(sustain_volume < f(1.0f), u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN));
            const U next = zsel(state == u(ZH_ENV_ATTACK), after_attack,
                                zsel(state == u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN), u(ZH_ENV_IDLE)));
            change_state_if(stage_end, next);
            resolve(stage_end);
            refresh_derived();
        }
    }
    // frame(), or frame_walk() while `walk` (the flag zs_frame_loop raises over the frames it replays)
    __device__ __forceinline__ M frame_s(F &val, bool walk) {
        if (walk) { frame_walk(); return zmask<M>(false); }
        return frame(val);
    }
    // frame_s() for a generated kernel's two frame bodies: Q = the chunk passed quiet() (and is not a replay)
    // (a replayed quiet chunk steps the clock only: frame_walk() brings last_value up to date at a stage end, and a quiet
    // chunk has none)
    template <bool Q> __device__ __forceinline__ M frame_sq(F &val, bool walk) {
        if constexpr (Q) {
            if (walk) { t = zsel(mode == u(ENV_MODE_TOWARD), t + cur_step, t); return zmask<M>(false); }
            return frame_quiet(val);
        } else {
            return frame_s(val, walk);
        }
    }
    // frame_loop_gen (seq.hip.h): true = no voice ends a stage within the next n frames.  A TOWARD voice's clock after k
    // steps is at most t + k * step + k half-ulps of 1, so t + (n + 1) * step < 0.999 keeps it below 1 with a wide margin
    // (a NaN or infinite step fails the test and takes the exact path).
    __device__ __forceinline__ bool quiet(int n) const {
        const M risky = zand(mode == u(ENV_MODE_TOWARD), znot(t + f((float)(n + 1)) * cur_step < f(0.999f)));
        return !zany_wave(risky);
    }
    // frame() where quiet() holds: the stage cannot finish, so the clamp, the stage-end test and its branch are gone
    __device__ __forceinline__ M frame_quiet(F &val) {
        const M toward = mode == u(ENV_MODE_TOWARD);
        const F tn = t + cur_step;
        const F lv = start + curve(tn) * cur_delta;                // :114
        t = zsel(toward, tn, t);
        last_value = zsel(toward, lv, last_value);
        val = zsel(toward, lv, sustain_volume);
        return mode != u(ENV_MODE_NONE);
    }
    // frame_masked() where quiet() holds
    __device__ __forceinline__ F frame_masked_quiet() {
        const M toward = mode == u(ENV_MODE_TOWARD);
        const F tn = t + cur_step;
        const F lv = start + curve(tn) * cur_delta;                // :114
        t = zsel(toward, tn, t);
        last_value = zsel(toward, lv, last_value);
        const F val = zsel(toward, lv, sustain_volume);
        return zbits_f(zbits_u(f(0.0f) + val) & m_painted);
    }
    // frame_masked_quiet() where, besides, EVERY voice of the wave is in a timed stage (a chord's attack, its release): the
    // mode compare, the three selects and the painted mask go; same operations on the same values for every lane
    __device__ __forceinline__ F frame_masked_all_toward_quiet() {
        const F tn = t + cur_step;
        const F lv = start + curve(tn) * cur_delta;                // :114
        t = tn;
        last_value = lv;
        return f(0.0f) + lv;
    }
    // N calls of frame() / frame_masked() with the values discarded, where quiet(N) holds (a frame-range kernel's replay):
    // the clock is stepped N times, the curve is evaluated once, for the last of them
    template <int N> __device__ __forceinline__ void skip_quiet() {
        const M toward = mode == u(ENV_MODE_TOWARD);
        F tt = t;
#pragma unroll
        for (int k = 0; k < N; k++) tt = tt + cur_step;
        t = zsel(toward, tt, t);
        last_value = zsel(toward, start + curve(tt) * cur_delta, last_value);
    }
    // skip_quiet() without last_value: frame() and frame_quiet() rewrite a TOWARD voice's last_value from the clock before anything
    // reads it (a stage end's `start = last_value` comes after that frame's own update), so a replay that is FOLLOWED by at least
    // one painted frame before the state is stored -- k_envelope_ranges: every range paints a frame of its own -- steps the clock alone
    template <int N> __device__ __forceinline__ void skip_clock() {
        const M toward = mode == u(ENV_MODE_TOWARD);
        F tt = t;
#pragma unroll
        for (int k = 0; k < N; k++) tt = tt + cur_step;
        t = zsel(toward, tt, t);
    }
    // frame() for the callers that want `painted ? 0.0f + value : 0.0f` (the zeroed temp a composite paints the
    // envelope into): the select is an AND with m_painted, and no mode compare is needed for it
    __device__ __forceinline__ F frame_masked() {
        const M toward = mode == u(ENV_MODE_TOWARD);
        F tn = t + cur_step;
        const M finished = tn >= f(1.0f);
        tn = zsel(finished, f(1.0f), tn);
        const F lv = start + curve(tn) * cur_delta;                // :114
        t = zsel(toward, tn, t);
        last_value = zsel(toward, lv, last_value);
        const F val = zsel(toward, lv, sustain_volume);            // FLAT: Envelope.zig:68-70
        const F e0 = zbits_f(zbits_u(f(0.0f) + val) & m_painted);
        const M stage_end = zand(toward, finished);
        if (zany(stage_end)) {                                     // Envelope.zig:53-58, 63-65, 86-88
            const U after_attack = zsel(sustain_volume < f(1.0f), u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN));
            const U next = zsel(state == u(ZH_ENV_ATTACK), after_attack,
                                zsel(state == u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN), u(ZH_ENV_IDLE)));
            change_state_if(stage_end, next);
            resolve(stage_end);
            refresh_derived();
        }
        return e0;
    }

    // ---- one voice per WAVE: the lanes are 64 consecutive frames (W = 1, every field wave-uniform) ----
    // block64() is frame() for frames [0, nf) of such a wave in one go, returning in lane j what frame j
    // adds to the zeroed temp (0 + value when painted, else 0).  Per frame only the time walk
    // t += step is inherently sequential; the curve is a pure function of t and runs once, in all lanes
    // at the same time.  Idle and sustain take no per-frame work at all.  A stage that ends inside the
    // block (first frame whose t reaches 1) is committed exactly as frame() does -- clamp, state change,
    // resolve() -- and the rest of the block runs in the next stage.  `scratch`: 64 floats of LDS owned
    // by this wave.  Same operations on the same values as nf calls of frame() => same bits.
    __device__ __forceinline__ float block64(uint32_t nf, uint32_t lane, float *scratch) {
        static_assert(W == 1, "one voice per wave");
        float mine = 0.0f;
        uint32_t j0 = 0;
        while (j0 < nf) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)mode);
            if (m != ENV_MODE_TOWARD) {                                // nothing can change before the block ends
                const float e = m == ENV_MODE_FLAT ? 0.0f + sustain_volume : 0.0f;   // FLAT: Envelope.zig:68-70
                mine = lane >= j0 ? e : mine;
                break;
            }
            const float tn_raw = zwalk64<false>(t, cur_step, j0, nf, lane, scratch);   // t after frame j's step, unclamped
            const bool in = lane >= j0 && lane < nf;
            const bool fin = in && tn_raw >= 1.0f;
            const uint64_t fm = __builtin_amdgcn_ballot_w64(fin);
            const uint32_t jf = fm ? (uint32_t)__builtin_ctzll(fm) : nf;   // the frame that finishes the stage (none: nf)
            const float tn = fin ? 1.0f : tn_raw;
            const float lv = start + curve(tn) * (cur_goal - start);   // painter.zig:114
            mine = (in && lane <= jf) ? 0.0f + lv : mine;
            const uint32_t last = jf < nf ? jf : nf - 1;               // state as of the stage's last painted frame
            t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tn), (int)last));
            last_value = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lv), (int)last));
            if (jf >= nf) break;
            const U after_attack = zsel(sustain_volume < f(1.0f), u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN));   // Envelope.zig:53-58, 63-65, 86-88
            const U next = zsel(state == u(ZH_ENV_ATTACK), after_attack, zsel(state == u(ZH_ENV_DECAY), u(ZH_ENV_SUSTAIN), u(ZH_ENV_IDLE)));
            change_state_if(zmask<M>(true), next);
            resolve(zmask<M>(true));
            refresh_derived();
            j0 = jf + 1;
        }
        return mine;
    }
};
using EnvLane = EnvLaneT<1>;
using EnvLaneCubed = EnvLaneT<1, ZH_CURVE_CUBED>;      // the composite instruments' envelopes

// Envelope.Params (Envelope.zig:6-13) as the kernels see them: tags shared, values per voice.
struct EnvParamsP {
    float sample_rate;
    uint32_t attack_tag, decay_tag, release_tag;
    F32P attack_dur, decay_dur, release_dur, sustain_volume;
    BoolP note_on;
};

template <int W, int FT>
__device__ __forceinline__ void env_load(EnvLaneT<W, FT> &e, const EnvParamsP &p, uint32_t v) {
    e.sample_rate = p.sample_rate;
    e.sustain_volume = zget_f32p<W>(p.sustain_volume, v);
    e.attack = CurvePT<W>{p.attack_tag, zget_f32p<W>(p.attack_dur, v)};
    e.decay = CurvePT<W>{p.decay_tag, zget_f32p<W>(p.decay_dur, v)};
    e.release = CurvePT<W>{p.release_tag, zget_f32p<W>(p.release_dur, v)};
    e.note_on = zget_boolp<W>(p.note_on, v);
}
