// ring.hip.h -- LDS ring counters for kernels whose waves form a pipeline (k_noise_filter_ring, k_pink_pipe).
//
// Ring counters live in LDS and only LDS data is handed over: a wave's LDS instructions are performed in issue order (a
// tile's ds_writes before the counter's ds_write; a slot's ds_reads, whose data the wave has consumed, before the counter
// that frees the slot), and a reader's accesses to a tile are control-dependent on the counter it polled.  The asm
// statements are compiler barriers.
#pragma once
#include "common.hip.h"

__device__ __forceinline__ bool ring_wait_ge(const uint32_t *counter, uint32_t want) {
    for (uint32_t it = 0; it < (1u << 22); it++) {
        const uint32_t seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(counter, __ATOMIC_RELAXED));
        if ((int32_t)(seen - want) >= 0) { asm volatile("" ::: "memory"); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__device__ __forceinline__ void ring_publish(uint32_t *counter, uint32_t value, uint32_t lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this wave's LDS reads have returned, its writes are queued in order
    if (lane == 0) __atomic_store_n(counter, value, __ATOMIC_RELAXED);
    asm volatile("" ::: "memory");
}
