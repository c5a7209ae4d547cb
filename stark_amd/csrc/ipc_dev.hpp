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
// Both granules of a double in ONE 16-byte write-through store (global_store_dwordx4 sc0 sc1; 16-byte aligned: a double's granule pair
// always is): half the fabric writes of two 8-byte stores. Each 8-byte half is a complete granule, so a reader that sees the halves at
// different times still never sees a torn one.
typedef unsigned int ipc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule_store_f64(unsigned long long* g, uint32_t tag, double v)
{
    ipc_u32x4 q;
    q.x = (uint32_t)__double2loint(v);
    q.y = tag;
    q.z = (uint32_t)__double2hiint(v);
    q.w = tag;
    // (an asm store is not in the compiler's wait-count bookkeeping; the trailing s_nop keeps the data registers intact until the store has
    // read them: cdna_hip_programming.md 5.7)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(g), "v"(q) : "memory");
}
// Polls the two granules of a double until both carry `tag`. Gives up after `budget` ticks of the device's constant clock counted from
// t0: sets *err (system scope, the host looks at it at its synchronisation points) and returns NaN, so that a peer that died or never
// launched ends this kernel with an error instead of hanging the GPU.
// `code` says which wait it was (the host's error message carries it).
__device__ __forceinline__ double granule_wait_f64(const unsigned long long* g, uint32_t tag, unsigned int* err, unsigned long long t0, unsigned long long budget, unsigned int code = 1u)
{
    for (unsigned spins = 0;; spins++) {
        const unsigned long long a = granule_load(g), b = granule_load(g + 1);
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) return __hiloint2double((int)(uint32_t)b, (int)(uint32_t)a);
        if ((spins & 63u) == 63u && wall_clock64() - t0 > budget) {
            unsigned int expected = 0u;  // (the FIRST wait that gave up names the cause; the ones behind it time out in its wake)
            __hip_atomic_compare_exchange_strong(err, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return __longlong_as_double(0x7ff8000000000000ll);
        }
        // (tight at first: an exchange between running kernels completes within microseconds; a rank waiting for a peer that is busy with
        // something else for milliseconds should not flood the memory system with polls)
        if (spins < 256u) __builtin_amdgcn_s_sleep(1);
        else __builtin_amdgcn_s_sleep(64);
    }
}

}  // namespace mistark
