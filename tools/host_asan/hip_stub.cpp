// hip_stub.cpp -- a HIP runtime that is not one: just enough of the API for the HOST side of libzang_hip.so to run on a box
// without a GPU, under AddressSanitizer + UBSan (tools/host_asan.sh).  Memory is malloc'ed and tracked (a hipFree of a pointer
// this stub never returned, or twice, aborts), streams know whether they are capturing, a captured "graph" counts its kernel
// nodes, kernels are never run.  What is exercised is the library's own bookkeeping: capture logs, held-back batches, flips,
// scratch growth and retirement, graph / module / context lifetimes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <set>

namespace {
struct FakeGraph { unsigned nodes = 0; };
struct FakeExec { unsigned nodes = 0; };
struct FakeStream { bool capturing = false; FakeGraph *g = nullptr; };
struct FakeEvent { int x = 0; };
std::map<void *, size_t> g_blocks;
std::set<FakeStream *> g_streams;
std::set<FakeGraph *> g_graphs;
std::set<FakeExec *> g_execs;
unsigned long g_launches = 0, g_captured = 0;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallCfg g_cfg[8];
thread_local int g_cfg_n = 0;
void die(const char *what) { fprintf(stderr, "hip_stub: %s\n", what); abort(); }
FakeStream *S(hipStream_t s) {
    FakeStream *fs = reinterpret_cast<FakeStream *>(s);
    if (fs && !g_streams.count(fs)) die("a stream that does not exist (destroyed?)");
    return fs;
}
bool known(const void *p, size_t n) {                       // inside one live block
    auto it = g_blocks.upper_bound(const_cast<void *>(p));
    if (it == g_blocks.begin()) return false;
    --it;
    return (const char *)p + n <= (const char *)it->first + it->second;
}
}   // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "hip_stub error"; }
hipError_t hipMalloc(void **p, size_t n) {
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0x5a, n);                                     // (device memory is not zeroed)
    g_blocks[*p] = n;
    return hipSuccess;
}
hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    if (!g_blocks.erase(p)) die("hipFree of a pointer that is not a live allocation (double free?)");
    free(p);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { FakeStream *fs = new FakeStream(); g_streams.insert(fs); *s = reinterpret_cast<hipStream_t>(fs); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { FakeStream *fs = S(s); if (!fs) return hipErrorInvalidValue; if (fs->capturing) die("a capturing stream destroyed"); g_streams.erase(fs); delete fs; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { FakeStream *fs = S(s); if (fs && fs->capturing) die("hipStreamSynchronize on a capturing stream (it would invalidate the capture)"); return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
    FakeStream *fs = S(s);
    if (!fs || fs->capturing) return hipErrorInvalidValue;
    fs->capturing = true; fs->g = new FakeGraph(); g_graphs.insert(fs->g);
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g) {
    FakeStream *fs = S(s);
    if (!fs || !fs->capturing) return hipErrorInvalidValue;
    fs->capturing = false; *g = reinterpret_cast<hipGraph_t>(fs->g); fs->g = nullptr;
    return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st) { FakeStream *fs = S(s); *st = fs && fs->capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipGraphGetNodes(hipGraph_t g, hipGraphNode_t *, size_t *n) { FakeGraph *fg = reinterpret_cast<FakeGraph *>(g); if (!g_graphs.count(fg)) die("hipGraphGetNodes: dead graph"); *n = fg->nodes; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, hipGraphNode_t *, char *, size_t) {
    FakeGraph *fg = reinterpret_cast<FakeGraph *>(g);
    if (!g_graphs.count(fg)) die("hipGraphInstantiate: dead graph");
    FakeExec *fe = new FakeExec(); fe->nodes = fg->nodes; g_execs.insert(fe); *e = reinterpret_cast<hipGraphExec_t>(fe);
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) { if (!g_execs.count(reinterpret_cast<FakeExec *>(e))) die("hipGraphLaunch: dead executable graph"); FakeStream *fs = S(s); if (fs && fs->capturing) die("hipGraphLaunch into a capturing stream"); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { FakeGraph *fg = reinterpret_cast<FakeGraph *>(g); if (!g_graphs.erase(fg)) die("hipGraphDestroy: not a live graph"); delete fg; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { FakeExec *fe = reinterpret_cast<FakeExec *>(e); if (!g_execs.erase(fe)) die("hipGraphExecDestroy: not a live executable graph"); delete fe; return hipSuccess; }
static hipError_t copy_like(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    FakeStream *fs = S(s);
    const bool dev_dst = kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice, dev_src = kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice;
    if (n && dev_dst && !known(dst, n)) die("a copy into device memory that is not (any more) allocated, or past its end");
    if (n && dev_src && !known(src, n)) die("a copy from device memory that is not (any more) allocated, or past its end");
    if (fs && fs->capturing) { fs->g->nodes++; return hipSuccess; }   // recorded, not run
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) { return copy_like(dst, src, n, kind, s); }
hipError_t hipMemcpyDtoH(void *dst, hipDeviceptr_t src, size_t n) { return copy_like(dst, src, n, hipMemcpyDeviceToHost, nullptr); }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s) {
    for (size_t r = 0; r < height; r++) { hipError_t e = copy_like((char *)dst + r * dpitch, (const char *)src + r * spitch, width, kind, s); if (e) return e; }
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s) {
    FakeStream *fs = S(s);
    if (n && !known(dst, n)) die("hipMemsetAsync outside a live allocation");
    if (fs && fs->capturing) { fs->g->nodes++; return hipSuccess; }
    memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new FakeEvent()); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<FakeEvent *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t s) { (void)S(s); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
// kernel launches: counted, recorded into a capturing stream's graph, never run
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s) { if (g_cfg_n >= 8) die("launch configuration stack"); g_cfg[g_cfg_n++] = CallCfg{grid, block, shmem, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *s) { if (!g_cfg_n) die("launch configuration stack empty"); const CallCfg &c = g_cfg[--g_cfg_n]; *grid = c.grid; *block = c.block; *shmem = c.shmem; *s = c.stream; return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t, hipStream_t s) {
    if (!grid.x || !grid.y || !grid.z || !block.x || block.x * block.y * block.z > 1024 || grid.y > 65535 || grid.z > 65535) die("a kernel launch with an impossible grid or block");
    FakeStream *fs = S(s);
    if (fs && fs->capturing) { fs->g->nodes++; g_captured++; } else g_launches++;
    return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipUnregisterFatBinary(void **) {}
// script.hip / xchg.hip: not part of this exercise
hipError_t hipModuleLoadData(hipModule_t *, const void *) { return hipErrorNotSupported; }
hipError_t hipModuleUnload(hipModule_t) { return hipErrorNotSupported; }
hipError_t hipModuleGetFunction(hipFunction_t *, hipModule_t, const char *) { return hipErrorNotSupported; }
hipError_t hipModuleGetGlobal(hipDeviceptr_t *, size_t *, hipModule_t, const char *) { return hipErrorNotSupported; }
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void **, void **) { return hipErrorNotSupported; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *, void *) { return hipErrorNotSupported; }
hipError_t hipIpcOpenMemHandle(void **, hipIpcMemHandle_t, unsigned) { return hipErrorNotSupported; }
hipError_t hipIpcCloseMemHandle(void *) { return hipErrorNotSupported; }
// what the harness asks at the end
void hip_stub_report(unsigned long *launches, unsigned long *captured, unsigned long *live_blocks, unsigned long *live_graphs, unsigned long *live_streams) {
    *launches = g_launches; *captured = g_captured; *live_blocks = g_blocks.size(); *live_graphs = g_graphs.size() + g_execs.size(); *live_streams = g_streams.size();
}
}   // extern "C"
