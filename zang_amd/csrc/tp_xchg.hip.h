// tp_xchg.hip.h -- hand-offs BETWEEN workgroups of one launch, for the single-launch time-parallel forms (filter_tp.hip.h).
//
// gfx950: the eight XCDs' L2s are not coherent with each other for ordinary accesses and a CU's L1 is never refreshed by
// another CU's stores, so data that one workgroup hands to another inside a launch travels as agent-scope accesses
// (`sc1`: write-through stores, L1-bypassing loads; MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup
// visibility").  The forms used here, all from that section's table of valid hand-offs:
//   payload   8-byte words ({l, b} of a filter state), every one stored by an agent-scope relaxed atomic store and loaded by
//             an agent-scope relaxed atomic load (global_store / global_load ... sc1);
//   flag      per (chunk, 64-voice wave): after its payload stores the storing wave waits `s_waitcnt vmcnt(0)`, then ONE lane
//             stores two 8-byte granules {tag, 32 bits of a lane mask} the same way.  A granule is written by one store, so its
//             tag and its data arrive together; the consumer polls both granules (sc1 loads) until both carry the launch's tag,
//             and only then loads payload;
//   tag       a launch's number, read from the module's scratch at kernel entry (`sync[0]` + 1) and advanced by the LAST workgroup
//             to finish (a counter of finished workgroups, `sync[1]`): flags left by an earlier launch -- or an earlier REPLAY of the
//             same recorded launch, which carries the same kernel arguments -- never match.
// Waiting on a workgroup that has not been dispatched yet is only safe when every workgroup of the launch is resident at once:
// the host sizes a launch by the occupancy the runtime reports (zh_tp1_resident_workgroups) and never launches more.
#pragma once
#include "common.hip.h"

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void tp_store8(void *p, uint64_t x) {
    __hip_atomic_store(reinterpret_cast<uint64_t *>(p), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t tp_load8(const void *p) {
    return __hip_atomic_load(reinterpret_cast<const uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every vector-memory operation of this wave has completed (inline asm: the compiler's own wait can be dropped by a pass that
// proves the scoreboard empty, MI355X_MICROARCH.md "Compiler hazard")
__device__ __forceinline__ void tp_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tp_sleep() { __builtin_amdgcn_s_sleep(2); }
#else
__device__ inline void tp_store8(void *, uint64_t) {}
__device__ inline uint64_t tp_load8(const void *) { return 0; }
__device__ inline void tp_drain() {}
__device__ inline void tp_sleep() {}
#endif

__device__ __forceinline__ uint64_t tp_pack(float l, float b) {
    return (uint64_t)__builtin_bit_cast(uint32_t, l) | ((uint64_t)__builtin_bit_cast(uint32_t, b) << 32);
}
__device__ __forceinline__ void tp_unpack(uint64_t x, float &l, float &b) {
    l = __builtin_bit_cast(float, (uint32_t)x);
    b = __builtin_bit_cast(float, (uint32_t)(x >> 32));
}
// a flag = two granules {tag, mask[31:0]}, {tag, mask[63:32]}
__device__ __forceinline__ void tp_flag_set(uint64_t *f, uint32_t tag, uint64_t mask) {
    tp_store8(f, (uint64_t)tag | ((mask & 0xffffffffull) << 32));
    tp_store8(f + 1, (uint64_t)tag | ((mask >> 32) << 32));
}
__device__ __forceinline__ bool tp_flag_try(const uint64_t *f, uint32_t tag, uint64_t &mask) {
    const uint64_t a = tp_load8(f), b = tp_load8(f + 1);
    mask = (a >> 32) | ((b >> 32) << 32);
    return (uint32_t)a == tag && (uint32_t)b == tag;
}
// One wave waits for a flag: lane 0 polls, every lane gets the mask.
__device__ __forceinline__ uint64_t tp_flag_wait(const uint64_t *f, uint32_t tag) {
    uint32_t lo = 0, hi = 0;
    for (;;) {
        uint32_t ok = 0;
        if ((threadIdx.x & 63u) == 0) {
            uint64_t m;
            ok = tp_flag_try(f, tag, m) ? 1u : 0u;
            lo = (uint32_t)m; hi = (uint32_t)(m >> 32);
        }
        if (__builtin_amdgcn_readfirstlane(ok)) break;
        tp_sleep();
    }
    lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// The launch is over for this workgroup: the last one to say so advances the launch number.  Call with every thread of the
// workgroup (it is a barrier); `total` = workgroups of the launch that call it.
__device__ __forceinline__ void tp_launch_done(uint32_t *sync, uint32_t tag, uint32_t total) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == total - 1u) {
            __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync[0], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ uint32_t tp_launch_tag(const uint32_t *sync) {
    return __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
}

#if !defined(ZH_DEVICE_ONLY)
// Workgroups of `kernel` (256 threads, its static LDS) that are resident at once on this device, with one per CU held back
// where the runtime reports three or more (the report can be one too many: MI355X_MICROARCH.md "Residency and cooperative
// launch"); 0 on failure.
template <class K> static inline uint32_t zh_tp1_resident_workgroups(K kernel, int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (per_cu >= 3) per_cu -= 1;
    if (per_cu > 8) per_cu = 8;
    return per_cu > 0 && cus > 0 ? (uint32_t)per_cu * (uint32_t)cus : 0u;
}
#endif
