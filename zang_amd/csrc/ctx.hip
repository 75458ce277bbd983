// ctx.hip -- context, memory, image upload/download, events (C ABI: include/zang_hip.h).
#include "common.hip.h"
#include <string.h>
#include <vector>
#include <string>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

// ---- flippers: modules with host-flipped double-buffered state (osc.hip) -------------------------------
static std::mutex g_flip_mu;
static std::unordered_map<uint64_t, zh_flipper *> g_flippers;
static uint64_t g_flip_next = 1;

void zh_flipper_register(zh_flipper *f) {
    std::lock_guard<std::mutex> lk(g_flip_mu);
    f->id = g_flip_next++;
    g_flippers[f->id] = f;
}
void zh_flipper_unregister(zh_flipper *f) {
    std::lock_guard<std::mutex> lk(g_flip_mu);
    g_flippers.erase(f->id);
}
// A paint of a flipper module during a capture, whether it will flip or paint in place on cnt[cur]: the graph bakes in
// cnt[cur as of now], so a replay must find the live state there (zh_graph_launch copies it over when later flipping
// paints -- eager range-form paints, other graphs -- have moved it to the other buffer).
void zh_flipper_used(zh_flipper *f) {
    zh_ctx *c = f->ctx;
    if (!c->capturing) return;
    for (const zh_flip_use &u : c->capture_log)
        if (u.id == f->id) return;
    c->capture_log.push_back(zh_flip_use{f->id, f, f->cur, 0u});
}
void zh_flipper_painted(zh_flipper *f) {
    zh_ctx *c = f->ctx;
    if (!c->capturing) return;
    for (zh_flip_use &u : c->capture_log)
        if (u.id == f->id) { u.flips++; return; }
    c->capture_log.push_back(zh_flip_use{f->id, f, f->cur, 1u});
}

// ---- the epoch of a ZH_CAPTURE_COALESCE capture (common.hip.h) ------------------------------------------
void zh_epoch_flush_batch(zh_ctx *c, bool last) {
    zh_co_batch &b = c->co;
    if (!b.active) return;
    b.active = false;
    const uint32_t n = (uint32_t)b.imgs.size();
    uint32_t flips = 0;
    for (const zh_flip_use &u : c->capture_log)
        if ((const void *)u.f == b.owner) flips = u.flips;
    if (b.flips && last && n >= 2 && ((flips + 1u) & 1u)) {              // one launch would leave an odd number of flips: two halves instead
        const uint32_t h = n / 2;
        c->co_launches += 2;
        b.launch(c->stream, b.imgs.data(), h);
        b.launch(c->stream, b.imgs.data() + h, n - h);
    } else {
        c->co_launches++;
        b.launch(c->stream, b.imgs.data(), n);
    }
    b.imgs.clear();
    b.items.reset();
    b.launch = nullptr;
}
void zh_epoch_barrier(zh_ctx *c) {
    if (!c->epoch_open) return;
    c->epoch_open = false;                                    // (first: the launches below must not re-enter)
    zh_epoch_flush_batch(c, true);
}

thread_local zh_ctx *zh_tls_ctx = nullptr;

// ZH_STORE_MODE: -1 = not set (the launcher's own choice), else 0..3
int zh_store_mode_env() {
    static int mode = -2;
    if (mode == -2) {
        const char *e = getenv("ZH_STORE_MODE");
        mode = e ? atoi(e) : -1;
        if (mode < -1 || mode > 3) mode = -1;
    }
    return mode;
}
int zh_store_mode() { const int m = zh_store_mode_env(); return m < 0 ? ST_SC1 : m; }

extern "C" {

const char *zh_version(void) { return "zang_hip 0.1 (gfx950)"; }

const char *zh_error_string(int err) {
    switch (err) {
    case ZH_OK: return "ok";
    case ZH_ERR_INVALID: return "invalid argument";
    case ZH_ERR_UNSUPPORTED: return "unsupported";
    case ZH_ERR_NO_DEVICE: return "no HIP device";
    case ZH_ERR_COMM: return "RCCL unavailable or rendezvous failed (zh_comm_last_error)";
    default:
        if (err <= ZH_ERR_RCCL_BASE && err > ZH_ERR_RCCL_BASE - 64) return "an RCCL call failed (zh_comm_last_error)";
        return err > 0 ? hipGetErrorString((hipError_t)err) : "unknown error";
    }
}

int zh_create(zh_ctx **out, int device) {
    if (!out) return ZH_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZH_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return ZH_ERR_INVALID;
    ZhDeviceGuard guard(device);                  // the caller's current device is restored on return
    zh_ctx *c = new (std::nothrow) zh_ctx();
    if (!c) return ZH_ERR_INVALID;
    c->device = device;
    c->own_stream = true;
    c->mix_partials = nullptr;
    c->mix_partials_floats = 0;
    c->capturing = false;
    c->capture_serial = 0;
    c->noise_jump = nullptr;
    c->capture_flags = 0; c->epoch_open = false; c->co_paints = c->co_launches = 0; c->form_fresh = false;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return (int)e; }
    *out = c;
    return ZH_OK;
}

int zh_destroy(zh_ctx *ctx) { ZH_GUARD(ctx);
    if (!ctx) return ZH_ERR_INVALID;
    hipStreamSynchronize(ctx->stream);
    // graphs that outlive their context (the documented order is graphs first; a host written against rounds 1-3 may not keep it):
    // they forget the context, and zh_graph_destroy then frees the graph alone (it used to dereference the freed context)
    for (zh_graph *g : ctx->graphs) g->ctx = nullptr;
    ctx->graphs.clear();
    if (ctx->mix_partials) hipFree(ctx->mix_partials);
    for (float *p : ctx->mix_retired) hipFree(p);
    if (ctx->noise_jump) hipFree(ctx->noise_jump);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return ZH_OK;
}

int zh_set_stream(zh_ctx *ctx, void *hip_stream) { ZH_GUARD(ctx);
    if (!ctx) return ZH_ERR_INVALID;
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    // NULL is HIP's default (null) stream: work is then ordered with everything else the
    // process enqueues there (e.g. PyTorch's default-stream copies and allocator reuse).
    ctx->stream = (hipStream_t)hip_stream;
    return ZH_OK;
}

void *zh_get_stream(zh_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int zh_sync(zh_ctx *ctx) { ZH_GUARD(ctx);
    if (!ctx) return ZH_ERR_INVALID;
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    return ZH_OK;
}

int zh_malloc(zh_ctx *ctx, void **dev_ptr, size_t bytes) { ZH_GUARD(ctx);
    if (!ctx || !dev_ptr) return ZH_ERR_INVALID;
    *dev_ptr = nullptr;
    if (bytes == 0) return ZH_OK;
    ZH_TRY(hipMalloc(dev_ptr, bytes));
    return ZH_OK;
}

int zh_free(zh_ctx *ctx, void *dev_ptr) { ZH_GUARD(ctx);
    if (!ctx) return ZH_ERR_INVALID;
    if (!dev_ptr) return ZH_OK;
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    ZH_TRY(hipFree(dev_ptr));
    return ZH_OK;
}

int zh_upload(zh_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) { ZH_GUARD(ctx);
    if (!ctx || (bytes && (!dev_dst || !host_src))) return ZH_ERR_INVALID;
    if (!bytes) return ZH_OK;
    ZH_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    return ZH_OK;
}

int zh_download(zh_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) { ZH_GUARD(ctx);
    if (!ctx || (bytes && (!host_dst || !dev_src))) return ZH_ERR_INVALID;
    if (!bytes) return ZH_OK;
    ZH_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    return ZH_OK;
}

int zh_buf_alloc(zh_ctx *ctx, zh_buf *out, uint32_t voices, uint32_t frames) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    memset(out, 0, sizeof *out);
    // A lane-per-voice kernel's consecutive stores are one row apart.  When a row is a multiple of 64 KiB they all
    // land on the same HBM channel and bank; padding per row spreads them.  4 KiB (1 KiB until round 5): with the chunked
    // oscillator's three-frame chunks no power-of-two voice count from 16,384 to 524,288 falls below 0.83 of the HBM peak
    // (1 KiB: 0.63 at 32,768 voices, 0.72 at 131,072; profiles/r05/osc_large_voice_counts.txt).
    const uint32_t stride = (voices != 0 && ((size_t)voices * sizeof(float)) % 65536 == 0) ? voices + kRowPadVoices : voices;
    void *p = nullptr;
    int rc = zh_malloc(ctx, &p, (size_t)stride * frames * sizeof(float));
    if (rc) return rc;
    out->ptr = (float *)p;
    out->voices = voices;
    out->frames = frames;
    out->stride = stride;
    return ZH_OK;
}

int zh_buf_free(zh_ctx *ctx, zh_buf *buf) { ZH_GUARD(ctx);
    if (!ctx || !buf) return ZH_ERR_INVALID;
    int rc = zh_free(ctx, buf->ptr);
    memset(buf, 0, sizeof *buf);
    return rc;
}

// Host [voice][frame] <-> device [frame][voice].  These are test/plumbing paths (PCIe
// bound); the transpose is done on the host into a staging vector.
int zh_buf_upload_voices(zh_ctx *ctx, zh_buf dst, const float *host, uint32_t frames) { ZH_GUARD(ctx);
    if (!ctx || !dst.ptr || !host || frames > dst.frames) return ZH_ERR_INVALID;
    std::vector<float> stage((size_t)frames * dst.voices);
    for (uint32_t v = 0; v < dst.voices; v++)
        for (uint32_t f = 0; f < frames; f++) stage[(size_t)f * dst.voices + v] = host[(size_t)v * frames + f];
    ZH_TRY(hipMemcpy2DAsync(dst.ptr, (size_t)dst.stride * 4, stage.data(), (size_t)dst.voices * 4,
                            (size_t)dst.voices * 4, frames, hipMemcpyHostToDevice, ctx->stream));
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    return ZH_OK;
}

int zh_buf_download_voices(zh_ctx *ctx, float *host, zh_buf src, uint32_t frames) { ZH_GUARD(ctx);
    if (!ctx || !src.ptr || !host || frames > src.frames) return ZH_ERR_INVALID;
    std::vector<float> stage((size_t)frames * src.voices);
    ZH_TRY(hipMemcpy2DAsync(stage.data(), (size_t)src.voices * 4, src.ptr, (size_t)src.stride * 4,
                            (size_t)src.voices * 4, frames, hipMemcpyDeviceToHost, ctx->stream));
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t v = 0; v < src.voices; v++)
        for (uint32_t f = 0; f < frames; f++) host[(size_t)v * frames + f] = stage[(size_t)f * src.voices + v];
    return ZH_OK;
}

int zh_buf_upload_voice(zh_ctx *ctx, zh_buf dst, uint32_t voice, const float *host, uint32_t frames) { ZH_GUARD(ctx);
    if (!ctx || !dst.ptr || !host || frames > dst.frames || voice >= dst.voices) return ZH_ERR_INVALID;
    ZH_TRY(hipMemcpy2DAsync(dst.ptr + voice, (size_t)dst.stride * 4, host, 4, 4, frames, hipMemcpyHostToDevice, ctx->stream));
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    return ZH_OK;
}

int zh_buf_download_voice(zh_ctx *ctx, float *host, zh_buf src, uint32_t voice, uint32_t frames) { ZH_GUARD(ctx);
    if (!ctx || !src.ptr || !host || frames > src.frames || voice >= src.voices) return ZH_ERR_INVALID;
    ZH_TRY(hipMemcpy2DAsync(host, 4, src.ptr + voice, (size_t)src.stride * 4, 4, frames, hipMemcpyDeviceToHost, ctx->stream));
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    return ZH_OK;
}

int zh_graph_begin_capture_flags(zh_ctx *ctx, uint32_t flags) { ZH_GUARD(ctx);
    if (!ctx || ctx->capturing || (flags & ~(uint32_t)ZH_CAPTURE_COALESCE)) return ZH_ERR_INVALID;
    ZH_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    ctx->capture_serial++;
    ctx->capture_flags = flags;
    ctx->capture_log.clear();
    ctx->epoch_open = false;
    ctx->co = zh_co_batch{};
    ctx->co_paints = ctx->co_launches = 0;
    ctx->capture_kernels.clear();
    ctx->deferred_error = 0;
    return ZH_OK;
}
int zh_graph_begin_capture(zh_ctx *ctx) { return zh_graph_begin_capture_flags(ctx, 0); }

int zh_graph_end_capture(zh_ctx *ctx, zh_graph **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    *out = nullptr;
    hipGraph_t g = nullptr;
    ctx->capturing = false;                                   // (ZH_GUARD above launched what was held back and published the counters)
    ctx->capture_flags = 0;
    std::vector<zh_flip_use> log;
    log.swap(ctx->capture_log);
    // The paints of the capture were recorded, not run, but each flipped its module's host-side buffer index:
    // put the indices back where the capture found them -- the module's state is still in that buffer.
    {
        std::lock_guard<std::mutex> lk(g_flip_mu);
        for (const zh_flip_use &u : log) {
            auto it = g_flippers.find(u.id);
            if (it != g_flippers.end()) it->second->cur = u.first_cur;
        }
    }
    ZH_TRY(hipStreamEndCapture(ctx->stream, &g));
    if (ctx->deferred_error) {                                // a held-back paint failed when it was finally launched: no graph
        const int rc = ctx->deferred_error;
        ctx->deferred_error = 0;
        ctx->capture_kernels.clear();
        if (g) hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc;
    }
    zh_graph *zg = new (std::nothrow) zh_graph();
    if (!zg) { hipGraphDestroy(g); return ZH_ERR_INVALID; }
    zg->graph = g;
    zg->ctx = ctx;
    hipError_t e = hipGraphInstantiate(&zg->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(g); delete zg; return (int)e; }
    ctx->graphs_live++;
    ctx->graphs.push_back(zg);
    zg->flips.swap(log);
    zg->co_paints = ctx->co_paints; zg->co_launches = ctx->co_launches;
    zg->kernels.swap(ctx->capture_kernels);
    size_t nn = 0;
    if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess) zg->nodes = (uint32_t)nn;
    (void)hipGetLastError();
    *out = zg;
    return ZH_OK;
}

int zh_graph_launch(zh_ctx *ctx, zh_graph *graph) { ZH_GUARD(ctx);
    if (!ctx || !graph || ctx->capturing || graph->ctx != ctx) return ZH_ERR_INVALID;     // (graph->ctx is null once its context was destroyed)
    // A replay reads each chunked oscillator's phase counters from the buffer the capture started on.  Paints since
    // then (eager ones, or another graph with an odd number of them) may have left the live state in the other
    // buffer: copy it over first (n * 4 bytes, enqueued ahead of the replay), then account for the replay's flips.
    {
        std::lock_guard<std::mutex> lk(g_flip_mu);
        for (const zh_flip_use &u : graph->flips)
            if (g_flippers.find(u.id) == g_flippers.end()) return ZH_ERR_INVALID;     // module destroyed since the capture
        for (const zh_flip_use &u : graph->flips) {
            zh_flipper *f = u.f;
            if (f->cur != u.first_cur) {
                ZH_TRY(hipMemcpyAsync(f->cnt[u.first_cur], f->cnt[f->cur], (size_t)f->n * f->words * 4, hipMemcpyDeviceToDevice, ctx->stream));
                f->cur = u.first_cur;
            }
        }
    }
    ZH_TRY(hipGraphLaunch(graph->exec, ctx->stream));
    for (const zh_flip_use &u : graph->flips) u.f->cur = u.first_cur ^ (int)(u.flips & 1u);
    return ZH_OK;
}

int zh_graph_info(const zh_graph *graph, uint32_t *nodes, uint32_t *paints_held, uint32_t *launches_of_held) {
    if (!graph) return ZH_ERR_INVALID;
    if (nodes) *nodes = graph->nodes;
    if (paints_held) *paints_held = graph->co_paints;
    if (launches_of_held) *launches_of_held = graph->co_launches;
    return ZH_OK;
}

int zh_graph_kernels(const zh_graph *graph, char *out, size_t n) {
    if (!graph || !out || n == 0) return ZH_ERR_INVALID;
    std::string s;
    for (const auto &kv : graph->kernels) {
        if (!s.empty()) s += ',';
        s += kv.first + " x" + std::to_string(kv.second);
    }
    const size_t k = s.size() < n - 1 ? s.size() : n - 1;
    memcpy(out, s.data(), k);
    out[k] = 0;
    return ZH_OK;
}

int zh_graph_destroy(zh_graph *graph) {
    if (!graph) return ZH_ERR_INVALID;
    zh_ctx *ctx = graph->ctx;
    ZH_GUARD(ctx);
    hipGraphExecDestroy(graph->exec);
    hipGraphDestroy(graph->graph);
    delete graph;
    // the last graph is gone: nothing can name a retired scratch block any more (a host that went from single paints
    // to batches kept every outgrown block until zh_destroy).  hipFree waits for the work in flight.
    if (ctx) {
        for (size_t i = 0; i < ctx->graphs.size(); i++)
            if (ctx->graphs[i] == graph) { ctx->graphs.erase(ctx->graphs.begin() + (long)i); break; }
    }
    if (ctx && ctx->graphs_live && --ctx->graphs_live == 0) {
        for (float *p : ctx->mix_retired) (void)hipFree(p);
        ctx->mix_retired.clear();
    }
    return ZH_OK;
}

int zh_event_create(zh_ctx *ctx, zh_event **out) { ZH_GUARD(ctx);
    if (!ctx || !out) return ZH_ERR_INVALID;
    zh_event *e = new (std::nothrow) zh_event();
    if (!e) return ZH_ERR_INVALID;
    hipError_t rc = hipEventCreate(&e->ev);
    if (rc != hipSuccess) { delete e; return (int)rc; }
    *out = e;
    return ZH_OK;
}

int zh_event_destroy(zh_event *ev) {
    if (!ev) return ZH_ERR_INVALID;
    hipEventDestroy(ev->ev);
    delete ev;
    return ZH_OK;
}

int zh_event_record(zh_ctx *ctx, zh_event *ev) { ZH_GUARD(ctx);
    if (!ctx || !ev) return ZH_ERR_INVALID;
    ZH_TRY(hipEventRecord(ev->ev, ctx->stream));
    return ZH_OK;
}

int zh_event_elapsed_ms(zh_event *start, zh_event *stop, float *ms) {
    if (!start || !stop || !ms) return ZH_ERR_INVALID;
    ZH_TRY(hipEventSynchronize(stop->ev));
    ZH_TRY(hipEventElapsedTime(ms, start->ev, stop->ev));
    return ZH_OK;
}

}  // extern "C"
