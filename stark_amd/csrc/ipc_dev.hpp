// ipc_dev.hpp — device side of the IPC-window exchange (dist.hpp: IpcCollective): granule stores and bounded granule polls.
//
// A granule is one naturally aligned 8-byte word {tag (high 32 bits), 32 data bits} written by ONE system-scope (write-through, sc0 sc1)
// store and read by system-scope loads: the data is its own flag, so an exchange needs neither a fence nor a separate flag store, and a
// granule can never be seen half-written (MI355X_MICROARCH.md "Granule"; cdna_hip_programming.md Guideline 16, form R2). A double travels as
// two granules (low word first). Tags are the exchange's sequence number, never 0; windows start zeroed, and a slot is reused only by the
// exchange two sequence numbers later, which every rank issues after it has consumed the one in between (dist.hip: "slot reuse").
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mistark {

using ipc_gu64 = __attribute__((address_space(1))) unsigned long long;

__device__ __forceinline__ void granule_store(unsigned long long* g, uint32_t tag, uint32_t v)
{
    __hip_atomic_store((ipc_gu64*)g, ((unsigned long long)tag << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* g)
{
    return __hip_atomic_load((ipc_gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void granule_store_f64(unsigned long long* g, uint32_t tag, double v)
{
    granule_store(g, tag, (uint32_t)__double2loint(v));
    granule_store(g + 1, tag, (uint32_t)__double2hiint(v));
}
// Polls the two granules of a double until both carry `tag`. Gives up after `budget` ticks of the device's constant clock counted from
// t0: sets *err (system scope, the host looks at it at its synchronisation points) and returns NaN, so that a peer that died or never
// launched ends this kernel with an error instead of hanging the GPU.
__device__ __forceinline__ double granule_wait_f64(const unsigned long long* g, uint32_t tag, unsigned int* err, unsigned long long t0, unsigned long long budget)
{
    for (unsigned spins = 0;; spins++) {
        const unsigned long long a = granule_load(g), b = granule_load(g + 1);
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) return __hiloint2double((int)(uint32_t)b, (int)(uint32_t)a);
        if ((spins & 63u) == 63u && wall_clock64() - t0 > budget) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return __longlong_as_double(0x7ff8000000000000ll);
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

}  // namespace mistark
