// comm.hip -- the collective form of the multi-GPU path's one exchange step, behind the C ABI:
// every GPU's partial mix [buffers][channels][frames] is summed with an RCCL reduce / all-reduce over xGMI, enqueued
// on the context's stream right behind the mixdown kernels that produced it.  A host in any language (the Zig host of
// INTEGRATION.md, tests/cpp) creates the communicator from a 128-byte id that rank 0 makes and hands to the other
// processes over whatever host channel it has -- no Python, no torch.distributed.
// librccl is opened with dlopen at first use (like hiprtc in script.hip) so that the library itself does not depend
// on it: a single-GPU host never loads it.
#include "common.hip.h"
#include <chrono>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>

namespace {

// the handful of declarations of <rccl/rccl.h> this file uses (RCCL 2.x ABI)
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId { char internal[ZH_COMM_ID_BYTES]; };
enum { kNcclSuccess = 0, kNcclInProgress = 7, kNcclSum = 0, kNcclFloat32 = 7 };
// ncclConfig_t as RCCL 2.14 first published it: {size, magic, version, blocking}.  ncclCommInitRankConfig reads `size` bytes
// and takes every later field's default, so this prefix is accepted by every RCCL from 2.14 on -- the process may hold
// torch's bundled 2.26 or the ROCm installation's 2.27, whose full structs differ.
struct NcclConfigPrefix { size_t size; unsigned int magic; unsigned int version; int blocking; };

struct Rccl {
    void *lib = nullptr;
    std::string path, why;
    int (*get_version)(int *) = nullptr;
    int (*get_unique_id)(ncclUniqueId *) = nullptr;
    int (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*comm_init_rank_config)(ncclComm_t *, int, ncclUniqueId, int, void *) = nullptr;
    int (*comm_get_async_error)(ncclComm_t, int *) = nullptr;
    int (*comm_finalize)(ncclComm_t) = nullptr;
    int (*comm_destroy)(ncclComm_t) = nullptr;
    int (*comm_abort)(ncclComm_t) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
    bool ok = false;
};

thread_local std::string g_last_error;

// the directory the process's HIP runtime was loaded from: its sibling librccl is the one built against it (a Python
// host has torch's bundled pair loaded, a C++ / Zig host the ROCm installation's)
std::string hip_runtime_dir() {
    Dl_info info;
    if (!dladdr((void *)&hipGetDeviceCount, &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "" : p.substr(0, k);
}

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        auto open = [&](const std::string &name, int extra) {
            if (x.lib || name.empty()) return;
            x.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL | extra);
            if (x.lib) x.path = name;
        };
        const char *forced = getenv("ZH_RCCL_LIB");
        if (forced && forced[0]) {
            open(forced, 0);
        } else {
            open("librccl.so.1", RTLD_NOLOAD);          // whatever copy the process already has (one RCCL per process)
            open("librccl.so", RTLD_NOLOAD);
            const std::string dir = hip_runtime_dir();
            if (!dir.empty()) { open(dir + "/librccl.so.1", 0); open(dir + "/librccl.so", 0); }
            open("librccl.so.1", 0);
            open("librccl.so", 0);
            open("/opt/rocm/lib/librccl.so.1", 0);
        }
        if (!x.lib) {
            const char *e = dlerror();
            x.why = std::string("librccl could not be opened") + (e ? std::string(": ") + e : "");
            return x;
        }
#define ZH_SYM(field, sym) *(void **)(&x.field) = dlsym(x.lib, sym)
        ZH_SYM(get_version, "ncclGetVersion");
        ZH_SYM(get_unique_id, "ncclGetUniqueId");
        ZH_SYM(comm_init_rank, "ncclCommInitRank");
        ZH_SYM(comm_init_rank_config, "ncclCommInitRankConfig");
        ZH_SYM(comm_get_async_error, "ncclCommGetAsyncError");
        ZH_SYM(comm_finalize, "ncclCommFinalize");
        ZH_SYM(comm_destroy, "ncclCommDestroy");
        ZH_SYM(comm_abort, "ncclCommAbort");
        ZH_SYM(all_reduce, "ncclAllReduce");
        ZH_SYM(reduce, "ncclReduce");
        ZH_SYM(error_string, "ncclGetErrorString");
#undef ZH_SYM
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce && x.reduce;
        if (!x.ok) x.why = "librccl (" + x.path + ") lacks a required symbol";
        return x;
    }();
    return r;
}

int rccl_fail(const char *what, int rc) {
    Rccl &r = rccl();
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s (ncclResult %d)", what, r.error_string ? r.error_string(rc) : "error", rc);
    g_last_error = buf;
    return ZH_ERR_RCCL_BASE - rc;
}

}   // namespace

struct zh_comm {
    zh_ctx *ctx;
    ncclComm_t comm;
    uint32_t world, rank;
    bool nonblocking;            // created through ncclCommInitRankConfig(blocking = 0): calls may answer ncclInProgress
};

namespace {
// seconds a rendezvous (and any later call that answers "in progress") may take: zh_comm_set_timeout, else ZH_COMM_TIMEOUT_S,
// else 180.  <= 0 = no limit.
double g_timeout_s = -1.0;
bool g_timeout_set = false;
double comm_timeout() {
    if (g_timeout_set) return g_timeout_s;
    const char *te = getenv("ZH_COMM_TIMEOUT_S");
    return te ? atof(te) : 180.0;
}
// Wait until the communicator's asynchronous state leaves ncclInProgress.  Returns the state, or kNcclInProgress on a timeout.
int comm_wait(ncclComm_t cm, double limit) {
    Rccl &r = rccl();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
        int state = kNcclSuccess;
        const int rc = r.comm_get_async_error(cm, &state);
        if (rc != kNcclSuccess) return rc;
        if (state != kNcclInProgress) return state;
        if (limit > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return kNcclInProgress;
        if (spin < 200) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(spin < 2000 ? 50 : 1000));
    }
}
}   // namespace

extern "C" {

int zh_comm_available(void) {
    Rccl &r = rccl();
    if (!r.ok) g_last_error = r.why;
    return r.ok ? 1 : 0;
}

const char *zh_comm_library(void) { return rccl().path.c_str(); }

int zh_comm_version(void) {
    Rccl &r = rccl();
    int v = 0;
    if (!r.ok || !r.get_version || r.get_version(&v) != kNcclSuccess) return 0;
    return v;
}

const char *zh_comm_last_error(void) { return g_last_error.c_str(); }

int zh_comm_unique_id(uint8_t *id128) {
    if (!id128) return ZH_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) { g_last_error = r.why; return ZH_ERR_COMM; }
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    const int rc = r.get_unique_id(&id);
    if (rc != kNcclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, ZH_COMM_ID_BYTES);
    return ZH_OK;
}

int zh_comm_create(zh_ctx *ctx, uint32_t world, uint32_t rank, const uint8_t *id128, zh_comm **out) { ZH_GUARD(ctx);
    if (out) *out = nullptr;
    if (!ctx || !out || !id128 || world == 0 || rank >= world) return ZH_ERR_INVALID;
    if (ctx->capturing) return ZH_ERR_UNSUPPORTED;
    Rccl &r = rccl();
    if (!r.ok) { g_last_error = r.why; return ZH_ERR_COMM; }
    zh_comm *c = new (std::nothrow) zh_comm();
    if (!c) return ZH_ERR_INVALID;
    c->ctx = ctx; c->comm = nullptr; c->world = world; c->rank = rank;
    ncclUniqueId id;
    memcpy(id.internal, id128, ZH_COMM_ID_BYTES);
    // The rendezvous is bounded by default: the communicator is created NON-BLOCKING
    // (ncclCommInitRankConfig, blocking = 0), the calling thread polls ncclCommGetAsyncError until the bootstrap has met every
    // rank, and after comm_timeout() seconds (default 180) gives up with ncclCommAbort -- which RCCL supports on a communicator
    // that is still initialising -- and ZH_ERR_COMM.  A rank that never arrives (it failed earlier, or returned early from this
    // very function: include/zang_hip.h "ALL OR NONE") no longer leaves the others waiting for ever, and no helper thread stays
    // behind (round 4's opt-in form detached one inside ncclCommInitRank).  A librccl without the three entry
    // points (before 2.14) falls back to the blocking ncclCommInitRank.
    const double limit = comm_timeout();
    if (!r.comm_init_rank_config || !r.comm_get_async_error || !r.comm_abort) {
        const int rc = r.comm_init_rank(&c->comm, (int)world, id, (int)rank);
        if (rc != kNcclSuccess) { delete c; return rccl_fail("ncclCommInitRank", rc); }
        c->nonblocking = false;
        *out = c;
        return ZH_OK;
    }
    NcclConfigPrefix cfg{sizeof(NcclConfigPrefix), 0xcafebeefu, 21400u, 0};
    int rc = r.comm_init_rank_config(&c->comm, (int)world, id, (int)rank, &cfg);
    if (rc != kNcclSuccess && rc != kNcclInProgress) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    c->nonblocking = true;
    rc = c->comm ? comm_wait(c->comm, limit) : rc;
    if (rc == kNcclInProgress) {
        if (c->comm) (void)r.comm_abort(c->comm);
        delete c;
        char msg[240];
        snprintf(msg, sizeof msg, "zh_comm_create: rank %u of %u saw no rendezvous within %.0f s (zh_comm_set_timeout / ZH_COMM_TIMEOUT_S): has every rank "
                 "called it with the same id?  The communicator was aborted.", rank, world, limit);
        g_last_error = msg;
        return ZH_ERR_COMM;
    }
    if (rc != kNcclSuccess) {
        if (c->comm) (void)r.comm_abort(c->comm);
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return ZH_OK;
}

int zh_comm_set_timeout(double seconds) {
    g_timeout_s = seconds; g_timeout_set = true;
    return ZH_OK;
}

// Asynchronous errors of a live communicator (a peer that died, a failed transport): ZH_OK while there is none -- an
// operation still in progress is not an error -- else ZH_ERR_RCCL_BASE - state with the text in zh_comm_last_error().
int zh_comm_check(zh_comm *comm) {
    if (!comm) return ZH_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.comm_get_async_error) return ZH_OK;
    int state = kNcclSuccess;
    const int rc = r.comm_get_async_error(comm->comm, &state);
    if (rc != kNcclSuccess) return rccl_fail("ncclCommGetAsyncError", rc);
    if (state != kNcclSuccess && state != kNcclInProgress) return rccl_fail("asynchronous RCCL error", state);
    return ZH_OK;
}

// End a communicator whose peers may be gone: ncclCommAbort (no collective hand-shake), nothing synchronised.
int zh_comm_abort(zh_comm *comm) {
    if (!comm) return ZH_OK;
    ZH_GUARD(comm->ctx);
    Rccl &r = rccl();
    int rc = kNcclSuccess;
    if (r.ok && comm->comm) rc = r.comm_abort ? r.comm_abort(comm->comm) : r.comm_destroy(comm->comm);
    delete comm;
    return rc == kNcclSuccess ? ZH_OK : rccl_fail("ncclCommAbort", rc);
}

int zh_comm_destroy(zh_comm *comm) {
    if (!comm) return ZH_OK;
    ZH_GUARD(comm->ctx);
    Rccl &r = rccl();
    int rc = kNcclSuccess;
    if (r.ok && comm->comm) {
        (void)hipStreamSynchronize(comm->ctx->stream);      // nothing of ours is still in flight on the communicator
        if (comm->nonblocking && r.comm_finalize) {           // the documented order for a non-blocking communicator
            rc = r.comm_finalize(comm->comm);
            if (rc == kNcclInProgress || rc == kNcclSuccess) rc = comm_wait(comm->comm, 30.0);
            if (rc == kNcclInProgress) rc = kNcclSuccess;     // (destroy below ends it either way)
        }
        const int rd = r.comm_destroy(comm->comm);
        if (rc == kNcclSuccess && rd != kNcclInProgress) rc = rd;
    }
    delete comm;
    return rc == kNcclSuccess ? ZH_OK : rccl_fail("ncclCommDestroy", rc);
}

int zh_comm_world(const zh_comm *comm) { return comm ? (int)comm->world : ZH_ERR_INVALID; }
int zh_comm_rank(const zh_comm *comm) { return comm ? (int)comm->rank : ZH_ERR_INVALID; }

int zh_allreduce_mix(zh_comm *comm, float *mix, size_t n) {
    if (!comm || (!mix && n)) return ZH_ERR_INVALID;
    ZH_GUARD(comm->ctx);
    if (n == 0) return ZH_OK;
    int rc = rccl().all_reduce(mix, mix, n, kNcclFloat32, kNcclSum, comm->comm, comm->ctx->stream);
    if (rc == kNcclInProgress && comm->nonblocking) rc = comm_wait(comm->comm, comm_timeout());   // the enqueue completes in the background
    return rc == kNcclSuccess ? ZH_OK : rccl_fail("ncclAllReduce", rc);
}

int zh_reduce_mix(zh_comm *comm, float *mix, size_t n, uint32_t root) {
    if (!comm || (!mix && n) || root >= comm->world) return ZH_ERR_INVALID;
    ZH_GUARD(comm->ctx);
    if (n == 0) return ZH_OK;
    int rc = rccl().reduce(mix, mix, n, kNcclFloat32, kNcclSum, (int)root, comm->comm, comm->ctx->stream);
    if (rc == kNcclInProgress && comm->nonblocking) rc = comm_wait(comm->comm, comm_timeout());
    return rc == kNcclSuccess ? ZH_OK : rccl_fail("ncclReduce", rc);
}

}  // extern "C"
