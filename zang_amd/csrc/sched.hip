// sched.hip -- host-side event scheduling: ImpulseQueue, NoteTracker, PolyphonyDispatcher
// (src/zang/notes.zig) and Trigger (src/zang/trigger.zig).  Pure host C++ (no device code);
// it lives in the library because it is the immediate caller of every paint: its output --
// (span, params, note_id_changed) per voice -- is what the paint entry points consume.
// Params are opaque blobs of params_size bytes (the Zig code is generic over NoteParamsType).
//
// Defined where the reference is undefined: a 33rd impulse in one buffer is dropped (the Zig
// arrays hold 32, notes.zig:73-74,142-143; ImpulseQueue.push already drops, :108-111);
// NoteTracker events out of chronological order (assert :177) clamp to frame 0.
#include "common.hip.h"
#include <string.h>
#include <vector>
#include <algorithm>

namespace {
constexpr uint32_t MAXI = ZH_MAX_IMPULSES, MAXP = ZH_MAX_PARAMS_SIZE;
struct Blob { uint8_t b[MAXP]; };
}

struct zh_impulse_queue {
    uint32_t psize;
    zh_impulse impulses[MAXI];
    Blob paramses[MAXI];
    uint64_t length, next_event_id;
    std::vector<uint8_t> packed;   // paramses packed at psize stride for the returned slice
};

struct zh_note_tracker {
    uint32_t psize;
    std::vector<uint8_t> song_params;
    std::vector<float> song_t;
    std::vector<uint64_t> song_note_id;
    uint64_t next_song_event;
    float t;
    zh_impulse impulses[MAXI];
    std::vector<uint8_t> packed;
};

struct SlotState { bool used; uint64_t note_id, event_id; bool note_on; };

struct zh_polyphony_dispatcher {
    uint32_t polyphony, psize, note_on_offset;
    std::vector<SlotState> slots;
    std::vector<zh_impulse> impulses;    // [polyphony][32]
    std::vector<uint8_t> paramses;       // [polyphony][32][psize]
};

struct zh_trigger {
    uint32_t psize;
    bool has_note;                       // "once set, never set back to null" (trigger.zig:39-41)
    uint64_t note_id;
    Blob note_params;
    // Counter (trigger.zig:43-48)
    zh_iap iap;
    uint64_t impulse_index, start, end;
};

namespace {
struct NoteSpan { uint64_t start, end; bool has_note; uint64_t id; const uint8_t *params; };

inline const uint8_t *param_at(const zh_iap &iap, uint32_t psize, uint64_t i) {
    return (const uint8_t *)iap.paramses + i * psize;
}

// trigger.zig:107-137
bool carry_over(const zh_trigger *t, NoteSpan &out) {
    if (!t->has_note) return false;
    if (t->impulse_index < t->iap.len) {
        const uint64_t next_impulse_frame = t->iap.impulses[t->impulse_index].frame;
        if (next_impulse_frame > t->start) {
            out = NoteSpan{t->start, std::min(t->end, next_impulse_frame), true, t->note_id, t->note_params.b};
            return true;
        }
        return false;
    }
    out = NoteSpan{t->start, t->end, true, t->note_id, t->note_params.b};
    return true;
}

// trigger.zig:139-196
NoteSpan get_next_note_span(zh_trigger *t) {
    const uint64_t base = t->impulse_index;
    const uint64_t n = t->iap.len - base;
    for (uint64_t i = 0; i < n; i++) {
        const zh_impulse &impulse = t->iap.impulses[base + i];
        if (impulse.frame >= t->end) break;
        if (impulse.frame > t->start) return NoteSpan{t->start, impulse.frame, false, 0, nullptr};
        t->impulse_index += 1;
        const uint64_t note_end_clipped = (i + 1 < n) ? std::min(t->end, t->iap.impulses[base + i + 1].frame) : t->end;
        if (note_end_clipped <= t->start) continue;
        return NoteSpan{t->start, note_end_clipped, true, impulse.note_id, param_at(t->iap, t->psize, base + i)};
    }
    return NoteSpan{t->start, t->end, false, 0, nullptr};
}

bool psize_ok(uint32_t s) { return s >= 1 && s <= MAXP; }
}  // namespace

extern "C" {

// ------------------------------------------------------------------ ImpulseQueue (notes.zig:72-128)
int zh_impulse_queue_create(uint32_t params_size, zh_impulse_queue **out) {
    if (!out || !psize_ok(params_size)) return ZH_ERR_INVALID;
    zh_impulse_queue *q = new (std::nothrow) zh_impulse_queue();
    if (!q) return ZH_ERR_INVALID;
    q->psize = params_size; q->length = 0; q->next_event_id = 1;
    *out = q;
    return ZH_OK;
}
int zh_impulse_queue_destroy(zh_impulse_queue *q) { if (!q) return ZH_ERR_INVALID; delete q; return ZH_OK; }
int zh_impulse_queue_push(zh_impulse_queue *q, uint64_t impulse_frame, uint64_t note_id, const void *params) {
    if (!q || !params) return ZH_ERR_INVALID;
    if (q->length >= MAXI) return ZH_OK;                                              // :108-111 dropped
    if (q->length > 0 && impulse_frame < q->impulses[q->length - 1].frame) return ZH_OK;   // :112-118 out of order: dropped
    q->impulses[q->length] = zh_impulse{impulse_frame, note_id, q->next_event_id};
    memcpy(q->paramses[q->length].b, params, q->psize);
    q->length += 1;
    q->next_event_id += 1;
    return ZH_OK;
}
int zh_impulse_queue_consume(zh_impulse_queue *q, zh_iap *out) {
    if (!q || !out) return ZH_ERR_INVALID;
    q->packed.resize((size_t)q->length * q->psize);
    for (uint64_t i = 0; i < q->length; i++) memcpy(q->packed.data() + i * q->psize, q->paramses[i].b, q->psize);
    *out = zh_iap{q->impulses, q->packed.data(), q->length};
    q->length = 0;                                                                    // :94
    return ZH_OK;
}

// ------------------------------------------------------------------ NoteTracker (notes.zig:138-207)
int zh_note_tracker_create(uint32_t params_size, uint64_t n, const void *paramses, const float *t, const uint64_t *note_ids,
                           zh_note_tracker **out) {
    if (!out || !psize_ok(params_size) || (n && (!paramses || !t || !note_ids))) return ZH_ERR_INVALID;
    zh_note_tracker *nt = new (std::nothrow) zh_note_tracker();
    if (!nt) return ZH_ERR_INVALID;
    nt->psize = params_size;
    nt->song_params.assign((const uint8_t *)paramses, (const uint8_t *)paramses + n * params_size);
    nt->song_t.assign(t, t + n);
    nt->song_note_id.assign(note_ids, note_ids + n);
    nt->next_song_event = 0;
    nt->t = 0.0f;
    nt->packed.resize((size_t)MAXI * params_size);
    *out = nt;
    return ZH_OK;
}
int zh_note_tracker_destroy(zh_note_tracker *nt) { if (!nt) return ZH_ERR_INVALID; delete nt; return ZH_OK; }
int zh_note_tracker_reset(zh_note_tracker *nt) {                                      // :155-158
    if (!nt) return ZH_ERR_INVALID;
    nt->next_song_event = 0; nt->t = 0.0f;
    return ZH_OK;
}
int zh_note_tracker_consume(zh_note_tracker *nt, float sample_rate, uint64_t span_start, uint64_t span_end, zh_iap *out) {
    if (!nt || !out || span_end < span_start) return ZH_ERR_INVALID;
    uint64_t count = 0;
    const uint64_t out_len = span_end - span_start;
    const float buf_time = (float)out_len / sample_rate;                              // :170
    const float end_t = nt->t + buf_time;                                             // :172
    const uint64_t n = nt->song_t.size();
    while (nt->next_song_event < n) {
        const uint64_t k = nt->next_song_event;
        const float note_t = nt->song_t[k];
        if (!(note_t < end_t)) break;                                                 // :178, :195-197
        const float f = (note_t - nt->t) / buf_time;                                  // :179
        const float pos = f * (float)out_len;
        uint64_t rel = 0;
        if (pos == pos && pos > 0.0f) rel = pos >= 1.8446744073709552e19f ? ~0ull : (uint64_t)pos;
        rel = std::min(rel, out_len - 1);                                             // :180-183
        nt->next_song_event += 1;                                                     // :186
        if (count < MAXI) {
            nt->impulses[count] = zh_impulse{span_start + rel, nt->song_note_id[k], nt->next_song_event};   // :187-191
            memcpy(nt->packed.data() + count * nt->psize, nt->song_params.data() + k * nt->psize, nt->psize);
            count += 1;
        }
    }
    nt->t = end_t;                                                                    // :200
    *out = zh_iap{nt->impulses, nt->packed.data(), count};
    return ZH_OK;
}

// ------------------------------------------------------------------ PolyphonyDispatcher (notes.zig:209-349)
int zh_polyphony_dispatcher_create(uint32_t polyphony, uint32_t params_size, uint32_t note_on_offset, zh_polyphony_dispatcher **out) {
    if (!out || !psize_ok(params_size) || polyphony == 0 || note_on_offset >= params_size) return ZH_ERR_INVALID;
    zh_polyphony_dispatcher *pd = new (std::nothrow) zh_polyphony_dispatcher();
    if (!pd) return ZH_ERR_INVALID;
    pd->polyphony = polyphony; pd->psize = params_size; pd->note_on_offset = note_on_offset;
    pd->slots.assign(polyphony, SlotState{false, 0, 0, false});
    pd->impulses.resize((size_t)polyphony * MAXI);
    pd->paramses.resize((size_t)polyphony * MAXI * params_size);
    *out = pd;
    return ZH_OK;
}
int zh_polyphony_dispatcher_destroy(zh_polyphony_dispatcher *pd) { if (!pd) return ZH_ERR_INVALID; delete pd; return ZH_OK; }
int zh_polyphony_dispatcher_reset(zh_polyphony_dispatcher *pd) {                      // :240-244
    if (!pd) return ZH_ERR_INVALID;
    for (auto &s : pd->slots) s.used = false;
    return ZH_OK;
}

// chooseSlot, notes.zig:246-306; returns -1 for null
static int64_t choose_slot(const zh_polyphony_dispatcher *pd, uint64_t note_id, bool note_on) {
    const uint32_t P = pd->polyphony;
    if (!note_on) {                                                                   // :253-264
        for (uint32_t i = 0; i < P; i++)
            if (pd->slots[i].used && pd->slots[i].note_id == note_id && pd->slots[i].note_on) return i;
        return -1;
    }
    int64_t best = -1;                                                                // :269-293
    for (uint32_t i = 0; i < P; i++) {
        if (pd->slots[i].used) {
            if (!pd->slots[i].note_on) {
                if (best >= 0) { if (pd->slots[i].event_id < pd->slots[best].event_id) best = i; }
                else best = i;
            }
        } else return i;                                                              // empty slot: take it now
    }
    if (best >= 0) return best;
    uint32_t b = 0;                                                                   // :296-305 steal the stalest note-on
    for (uint32_t i = 1; i < P; i++) if (pd->slots[i].event_id < pd->slots[b].event_id) b = i;
    return b;
}

int zh_polyphony_dispatcher_dispatch(zh_polyphony_dispatcher *pd, zh_iap iap, zh_iap *out) {
    if (!pd || !out || (iap.len && (!iap.impulses || !iap.paramses))) return ZH_ERR_INVALID;
    std::vector<uint64_t> counts(pd->polyphony, 0);
    for (uint64_t i = 0; i < iap.len; i++) {                                          // :316-336
        const zh_impulse &impulse = iap.impulses[i];
        const uint8_t *params = param_at(iap, pd->psize, i);
        const bool note_on = params[pd->note_on_offset] != 0;
        const int64_t slot = choose_slot(pd, impulse.note_id, note_on);
        if (slot < 0) continue;
        pd->slots[slot] = SlotState{true, impulse.note_id, impulse.event_id, note_on};
        if (counts[slot] < MAXI) {
            pd->impulses[(size_t)slot * MAXI + counts[slot]] = impulse;
            memcpy(pd->paramses.data() + ((size_t)slot * MAXI + counts[slot]) * pd->psize, params, pd->psize);
            counts[slot] += 1;
        }
    }
    for (uint32_t s = 0; s < pd->polyphony; s++)                                      // :338-346
        out[s] = zh_iap{pd->impulses.data() + (size_t)s * MAXI, pd->paramses.data() + (size_t)s * MAXI * pd->psize, counts[s]};
    return ZH_OK;
}

// ------------------------------------------------------------------ Trigger (trigger.zig:26-198)
int zh_trigger_create(uint32_t params_size, zh_trigger **out) {
    if (!out || !psize_ok(params_size)) return ZH_ERR_INVALID;
    zh_trigger *t = new (std::nothrow) zh_trigger();
    if (!t) return ZH_ERR_INVALID;
    memset(t, 0, sizeof *t);
    t->psize = params_size;
    *out = t;
    return ZH_OK;
}
int zh_trigger_destroy(zh_trigger *t) { if (!t) return ZH_ERR_INVALID; delete t; return ZH_OK; }
int zh_trigger_reset(zh_trigger *t) { if (!t) return ZH_ERR_INVALID; t->has_note = false; return ZH_OK; }   // :62-64
int zh_trigger_counter(zh_trigger *t, uint64_t span_start, uint64_t span_end, zh_iap iap) {                  // :66-78
    if (!t || span_end < span_start || (iap.len && (!iap.impulses || !iap.paramses))) return ZH_ERR_INVALID;
    t->iap = iap; t->impulse_index = 0; t->start = span_start; t->end = span_end;
    return ZH_OK;
}
int zh_trigger_next(zh_trigger *t, zh_paint_span *out) {                                                      // :80-105
    if (!t || !out) return ZH_ERR_INVALID;
    while (t->start < t->end) {
        NoteSpan ns;
        if (!carry_over(t, ns)) ns = get_next_note_span(t);
        t->start = ns.end;
        if (ns.has_note) {
            memset(out, 0, sizeof *out);
            out->start = ns.start; out->end = ns.end;
            memcpy(out->params, ns.params, t->psize);
            out->note_id_changed = t->has_note ? (ns.id != t->note_id ? 1u : 0u) : 1u;   // :96-99
            Blob keep;
            memcpy(keep.b, ns.params, t->psize);       // ns.params may alias note_params (carry-over)
            t->has_note = true; t->note_id = ns.id; t->note_params = keep;               // defer self.note = note (:91)
            return 1;
        }
    }
    return 0;
}


// ------------------------------------------------------------------ Voice(T)'s scheduling half
// examples/example_song.zig:287-350: one NoteTracker -> one PolyphonyDispatcher -> one Trigger per
// sub-voice.  zh_poly_voice_schedule makes the calls of `n_buffers` consecutive Voice(T).paint
// invocations and records, per sub-voice, the (span, params, note_id_changed) tuples whose
// module.paint the reference would run -- laid out like zh_span_table, [span][sub_voice].
struct zh_poly_voice {
    uint32_t polyphony, psize;
    zh_note_tracker *tracker;
    zh_polyphony_dispatcher *dispatcher;
    std::vector<zh_trigger *> triggers;
};

int zh_poly_voice_create(uint32_t polyphony, uint32_t params_size, uint32_t note_on_offset, uint64_t n_events,
                         const void *paramses, const float *t, const uint64_t *note_ids, zh_poly_voice **out) {
    if (!out || polyphony == 0) return ZH_ERR_INVALID;
    *out = nullptr;
    zh_poly_voice *pv = new (std::nothrow) zh_poly_voice();
    if (!pv) return ZH_ERR_INVALID;
    pv->polyphony = polyphony; pv->psize = params_size; pv->tracker = nullptr; pv->dispatcher = nullptr;
    int rc = zh_note_tracker_create(params_size, n_events, paramses, t, note_ids, &pv->tracker);
    if (!rc) rc = zh_polyphony_dispatcher_create(polyphony, params_size, note_on_offset, &pv->dispatcher);
    for (uint32_t i = 0; !rc && i < polyphony; i++) {
        zh_trigger *tr = nullptr;
        rc = zh_trigger_create(params_size, &tr);
        if (!rc) pv->triggers.push_back(tr);
    }
    if (rc) { zh_poly_voice_destroy(pv); return rc; }
    *out = pv;
    return ZH_OK;
}
int zh_poly_voice_destroy(zh_poly_voice *pv) {
    if (!pv) return ZH_ERR_INVALID;
    if (pv->tracker) zh_note_tracker_destroy(pv->tracker);
    if (pv->dispatcher) zh_polyphony_dispatcher_destroy(pv->dispatcher);
    for (zh_trigger *t : pv->triggers) zh_trigger_destroy(t);
    delete pv;
    return ZH_OK;
}
int zh_poly_voice_reset(zh_poly_voice *pv) {                                          // example_song.zig:318-324
    if (!pv) return ZH_ERR_INVALID;
    zh_note_tracker_reset(pv->tracker);
    zh_polyphony_dispatcher_reset(pv->dispatcher);
    for (zh_trigger *t : pv->triggers) zh_trigger_reset(t);
    return ZH_OK;
}
int zh_poly_voice_schedule(zh_poly_voice *pv, float sample_rate, const uint32_t *frames, uint32_t n_buffers, uint32_t max_spans,
                           uint32_t *counts, uint32_t *start, uint32_t *end, void *params, uint8_t *note_id_changed) {
    if (!pv || (n_buffers && !frames) || !counts || (max_spans && (!start || !end || !params || !note_id_changed))) return ZH_ERR_INVALID;
    const uint32_t P = pv->polyphony;
    for (uint32_t v = 0; v < P; v++) counts[v] = 0;
    std::vector<zh_iap> poly(P);
    uint64_t base = 0;
    for (uint32_t b = 0; b < n_buffers; b++) {
        zh_iap iap;
        int rc = zh_note_tracker_consume(pv->tracker, sample_rate, 0, frames[b], &iap);       // :333
        if (!rc) rc = zh_polyphony_dispatcher_dispatch(pv->dispatcher, iap, poly.data());     // :335
        if (rc) return rc;
        for (uint32_t v = 0; v < P; v++) {                                                    // :337-347
            rc = zh_trigger_counter(pv->triggers[v], 0, frames[b], poly[v]);
            if (rc) return rc;
            zh_paint_span ps;
            while ((rc = zh_trigger_next(pv->triggers[v], &ps)) == 1) {
                const uint32_t k = counts[v];
                if (k >= max_spans) return ZH_ERR_INVALID;
                const size_t idx = (size_t)k * P + v;
                start[idx] = (uint32_t)(base + ps.start);
                end[idx] = (uint32_t)(base + ps.end);
                memcpy((uint8_t *)params + idx * pv->psize, ps.params, pv->psize);
                note_id_changed[idx] = (uint8_t)ps.note_id_changed;
                counts[v] = k + 1;
            }
            if (rc < 0) return rc;
        }
        base += frames[b];
    }
    return ZH_OK;
}

}  // extern "C"
