// xchg.hip -- the one exchange step of the multi-GPU path: every GPU's partial mix
// [buffers][channels][frames] is summed into the root's.  Two forms exist: an RCCL all-reduce issued by the host
// binding (torch.distributed / zang_amd/sharding.py), and the direct form here -- the root owns one slot per
// rank in a block of its own HBM, shares it with the other processes through a HIP IPC handle, each rank's
// mixdown kernels store their partials STRAIGHT into their slot (4 KiB per buffer over xGMI, no staging copy, no
// collective), and the root adds the slots in rank order: a fixed order, so the mix is reproducible bit for bit.
// Cross-process ordering is the host's job (each rank synchronises its stream, then signals; sharding.py).
#include "common.hip.h"
#include <string.h>

static_assert(sizeof(hipIpcMemHandle_t) == 64, "zh_ipc_* pass a 64-byte handle");

// dst[i] (+)= ((slot0[i] + slot1[i]) + slot2[i]) + ...   16 B per lane where alignment allows
template <int VEC>
__global__ void __launch_bounds__(256) k_sum_slots(float *__restrict__ dst, const float *__restrict__ slots, uint32_t n_slots,
                                                   size_t slot_stride, size_t n, int zero_first) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= n) return;
    if constexpr (VEC == 4) {
        float4 s = *reinterpret_cast<const float4 *>(slots + i);
        for (uint32_t r = 1; r < n_slots; r++) {
            const float4 x = *reinterpret_cast<const float4 *>(slots + (size_t)r * slot_stride + i);
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        if (!zero_first) {
            const float4 d = *reinterpret_cast<const float4 *>(dst + i);
            s.x = d.x + s.x; s.y = d.y + s.y; s.z = d.z + s.z; s.w = d.w + s.w;
        }
        *reinterpret_cast<float4 *>(dst + i) = s;
    } else {
        float s = slots[i];
        for (uint32_t r = 1; r < n_slots; r++) s += slots[(size_t)r * slot_stride + i];
        dst[i] = zero_first ? s : dst[i] + s;
    }
}

extern "C" {

int zh_ipc_alloc(zh_ctx *ctx, size_t bytes, void **dev_ptr, uint8_t *handle64) { ZH_GUARD(ctx);
    if (!ctx || !dev_ptr || !handle64 || bytes == 0) return ZH_ERR_INVALID;
    *dev_ptr = nullptr;
    void *p = nullptr;
    // fine-grained: stores arriving from another GPU over xGMI are visible to this GPU's next kernel without relying
    // on how its L2 treats lines of its own HBM; plain device memory if the runtime refuses
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        ZH_TRY(hipMalloc(&p, bytes));
    }
    hipError_t e = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { hipFree(p); return (int)e; }
    memcpy(handle64, &h, 64);
    *dev_ptr = p;
    return ZH_OK;
}

int zh_ipc_open(zh_ctx *ctx, const uint8_t *handle64, void **dev_ptr) { ZH_GUARD(ctx);
    if (!ctx || !handle64 || !dev_ptr) return ZH_ERR_INVALID;
    *dev_ptr = nullptr;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    ZH_TRY(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return ZH_OK;
}

int zh_ipc_close(zh_ctx *ctx, void *dev_ptr) { ZH_GUARD(ctx);
    if (!ctx) return ZH_ERR_INVALID;
    if (!dev_ptr) return ZH_OK;
    ZH_TRY(hipStreamSynchronize(ctx->stream));
    ZH_TRY(hipIpcCloseMemHandle(dev_ptr));
    return ZH_OK;
}

int zh_sum_slots(zh_ctx *ctx, float *dst, const float *slots, uint32_t n_slots, size_t slot_stride_floats, size_t n,
                 uint32_t flags) { ZH_GUARD(ctx);
    if (!ctx || !dst || !slots || n_slots == 0 || slot_stride_floats < n) return ZH_ERR_INVALID;
    if (n == 0) return ZH_OK;
    const int zf = (int)(flags & ZH_PAINT_ZERO_FIRST);
    const bool vec = n % 4 == 0 && slot_stride_floats % 4 == 0 && (((uintptr_t)dst | (uintptr_t)slots) & 15u) == 0;
    if (vec) ZH_LAUNCH(k_sum_slots<4>, dim3((uint32_t)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, dst, slots, n_slots, slot_stride_floats, n, zf);
    else ZH_LAUNCH(k_sum_slots<1>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst, slots, n_slots, slot_stride_floats, n, zf);
    return zh_launch_status();
}

}  // extern "C"
