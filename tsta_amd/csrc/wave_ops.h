// wave_ops.h — wave64 cross-lane primitives for gfx950 (CDNA4).  Device code only.
//
// The DP kernels pass neighbour cells between lanes with DPP whole-wave shifts (GFX9-family
// `wave_shr:1` / `wave_shl:1`), which stay in the VALU (no LDS round trip as `__shfl_up`'s
// ds_bpermute_b32 would take).  Lanes without a source lane keep `old` (bound_ctrl off), which is how a
// boundary value is injected at lane 0 (shr) or lane 63 (shl) in the same instruction.
#pragma once
#include <hip/hip_runtime.h>

namespace tsta {

constexpr int kWave = 64;

// lane l <- src[l-1] for l >= 1; lane 0 <- old[0]
__device__ __forceinline__ int wave_shr1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
// lane l <- src[l+1] for l <= 62; lane 63 <- old[63]
__device__ __forceinline__ int wave_shl1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Opaque copy: stops the compiler from hoisting per-lane 64-bit addresses derived from `x` out of a loop and
// keeping them live in VGPR pairs across the hot loop (they cost occupancy; recomputing them is 2-3 VALU ops
// per 64-step block).
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }   // -> v_max3_i32

// Orders this wave's own global stores before its later loads of the same bytes by OTHER lanes of the
// same wave (band hand-off through the in-place boundary row).  Same wave => same CU/L1, in-order vector
// memory pipe; the waitcnt drains the stores, the fence stops compiler reordering.
__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace tsta
