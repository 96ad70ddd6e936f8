// handoff.h — band-to-band hand-off INSIDE one launch (the single-launch fills of psa_trace.hip / psa_trace_packed.hip).
//
// Form (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", valid forms / sc1 table):
//   producer wave:  sc1 (write-through) stores of the boundary-row entries  ->  s_waitcnt vmcnt(0)  ->  sc1 store of its
//                   progress word (each wave signals only for its own stores)
//   consumer wave:  sc1-load poll of that word (bypasses its CU's L1)  ->  sc1 loads of the entries
// No agent-scope fence on either side: no L2 write-back, no L1/L2 invalidate.  All accesses are global_* (the pointers are
// cast to the global address space: the compiler would otherwise emit flat_*, which the guide excludes).
// Deadlock freedom is the callers' business (bands are handed out as tickets in dependency order); a poll that does not
// end sets a status bit and gives up, so that every wave terminates; the word is read by the finishing kernel of every such
// launch (results are poisoned, never silently wrong) and by the host entry points, which run the call again through the
// launch-per-diagonal form of the same kernels.  Test hook shared by all single-launch families: engine option
// "test_abandon" = t makes the wave (workgroup) that draws ticket t-1 do its work but never publish it, so that its
// dependants really run into the bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsta {

typedef __attribute__((address_space(1))) unsigned long long GlobalU64;
typedef __attribute__((address_space(1))) int GlobalI32;
typedef __attribute__((address_space(1))) unsigned GlobalU32;

__device__ __forceinline__ int2 load_sc1_int2(const int2* p)
{
    const unsigned long long v = __hip_atomic_load((const GlobalU64*)reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
}

__device__ __forceinline__ int load_sc1_int(const int* p)
{
    return __hip_atomic_load((const GlobalI32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_sc1_int(int* p, int v)
{
    __hip_atomic_store((GlobalI32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_sc1_int2(int2* p, int x, int y)
{
    __hip_atomic_store((GlobalU64*)reinterpret_cast<unsigned long long*>(p), ((unsigned long long)(unsigned)y << 32) | (unsigned)x,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until *word >= need (word = progress of the band above, in steps).  The poll backs off with the distance still
// to go: a band parked far behind the one above (pipeline ramp-up) looks once per ~3.4 us per 128 steps of distance instead
// of hammering the fabric (thousands of waves polling every 100 cycles cost the working waves more than the launches this
// form removes: measured, 64 x 50 kbp pairs 120 -> 88 ms); close to the goal it looks every ~130 cycles.
__device__ __forceinline__ void wait_progress(const int* word, int need, unsigned* status, int lane)
{
    int spins = 0;
    for (;;) {
        const int cur = __builtin_amdgcn_readfirstlane(__hip_atomic_load((const GlobalI32*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (cur >= need) break;
        const int far = (need - cur) >> 7;                      // 64-step blocks take >= 12 us: 3.4 us of sleep per 128 steps is safe
        // once any wave has given up, nobody waits any more and the launch drains quickly.  How often a waiting wave looks at that ONE
        // word matters: thousands of waves poll it, and looking before every long sleep cost the 128 x 50 kbp band pipeline 30 %
        // (46 -> 61 ms), looking every 16 short spins cost config #2 35 %.  Every 8th long sleep (<= 0.9 ms between looks instead of
        // the 28 ms ADVICE round 3 objected to) and every 256th short one cost nothing measurable.
        if (((far > 0 && (spins & 7) == 7) || (spins & 255) == 255) &&
            __hip_atomic_load((const GlobalU32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (far > 0) { for (int i = 0; i < min(far, 32); ++i) __builtin_amdgcn_s_sleep(127); }
        else __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 17)) { if (lane == 0) atomicOr(status, 1u); break; }      // far: ~14 s; near: ~0.1 s
    }
    // compiler-only barrier: the sc1 loads of the entries that follow (other addresses, relaxed) must not be hoisted above
    // the final poll.  The hardware ordering comes from the s_waitcnt on the poll's result, which the loop exit depends on.
    asm volatile("" ::: "memory");
}

// one look at the word, no waiting (wave-uniform answer)
__device__ __forceinline__ bool progress_reached(const int* word, int need)
{
    const bool ok = __builtin_amdgcn_readfirstlane(__hip_atomic_load((const GlobalI32*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= need;
    asm volatile("" ::: "memory");                              // (as at the end of wait_progress)
    return ok;
}

// this wave's boundary-row stores have left; publish how far it is
__device__ __forceinline__ void publish_progress(int* word, int value, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store((GlobalI32*)word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tsta
