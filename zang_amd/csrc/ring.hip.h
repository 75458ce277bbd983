// ring.hip.h -- LDS ring counters for kernels whose waves form a pipeline (k_noise_filter_ring, k_pink_pipe).
//
// Ring counters live in LDS and only LDS data is handed over: a wave's LDS instructions are performed in issue order (a
// tile's ds_writes before the counter's ds_write; a slot's ds_reads, whose data the wave has consumed, before the counter
// that frees the slot), and a reader's accesses to a tile are control-dependent on the counter it polled.  The asm
// statements are compiler barriers.
#pragma once
#include "common.hip.h"

// (bit 31 of a counter word is a flag the publisher may set beside the count -- k_noise_filter_ring marks tiles that hold a
// multi-draw sample -- and is not part of the comparison; `seen_out` receives the word that ended the wait)
constexpr uint32_t kRingFlag = 0x80000000u;
__device__ __forceinline__ bool ring_reached(uint32_t seen, uint32_t want) { return (int32_t)((seen & ~kRingFlag) - want) >= 0; }
__device__ __forceinline__ bool ring_wait_ge(const uint32_t *counter, uint32_t want, uint32_t *seen_out = nullptr) {
    for (uint32_t it = 0; it < (1u << 22); it++) {
        const uint32_t seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(counter, __ATOMIC_RELAXED));
        if (ring_reached(seen, want)) { asm volatile("" ::: "memory"); if (seen_out) *seen_out = seen; return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__device__ __forceinline__ void ring_publish(uint32_t *counter, uint32_t value, uint32_t lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this wave's LDS reads have returned, its writes are queued in order
    if (lane == 0) __atomic_store_n(counter, value, __ATOMIC_RELAXED);
    asm volatile("" ::: "memory");
}

// Publication of DATA WRITES: the counter store is queued behind this wave's tile writes and a wave's LDS instructions are
// performed in issue order, so no drain (s_waitcnt) is needed -- a reader that sees the new counter issues its tile reads
// after it, behind those writes.  (ring_publish, which frees a slot after READS, keeps its wait: the data must have reached
// this wave's registers before another wave may overwrite the slot... the reads themselves execute in order too, but the
// wait there is cheap and off the critical wave.)
__device__ __forceinline__ void ring_publish_writes(uint32_t *counter, uint32_t value, uint32_t lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __atomic_store_n(counter, value, __ATOMIC_RELAXED);
    asm volatile("" ::: "memory");
}
