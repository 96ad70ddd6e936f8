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

// The whole per-step lane hand-off of the skewed DP sweep in ONE asm block (8 VALU + 3 SALU issue slots):
//   cch  <- lane l-1's cch            (lane 0  <- ach[0])      column character moves down one lane
//   hu   <- lane l-1's hbot           (lane 0  <- th[0])       H of the row above
//   eu   <- lane l-1's ebot           (lane 0  <- te[0])       Ebar of the row above
//   sh   <- lane l+1's sh             (lane 63 <- hbot[63])    bottom-row capture (shift register)
//   se   <- lane l+1's se             (lane 63 <- ebot[63])
//   ach, th, te rotate left by one lane (next column's character / top-row entry into lane 0)
// `v_cndmask_b32_dpp d, shifted, inject, vcc ... bound_ctrl:0` selects the injected value in the ONE lane named by VCC
// and the DPP-shifted value elsewhere, so no register copy for an `old` operand and none of the s_nop pairs the
// compiler puts between such a copy and its v_mov_dpp.  Hazards: a DPP source needs 2 wait states after a VALU write, and
// the compiler's hazard recognizer does not look into the block, so the block itself guarantees them for anything the
// caller may have written in its last instruction: s_mov + s_nop come before the first DPP read (cch), every other
// DPP source is read at least two instructions into the block.
__device__ __forceinline__ void lane_exchange(unsigned& cch, unsigned& hu, unsigned& eu, unsigned& sh, unsigned& se,
                                              unsigned& ach, unsigned& th, unsigned& te, unsigned hbot, unsigned ebot)
{
    // hu / eu are produced IN PLACE in the registers holding hbot / ebot (their last readers come first)
    hu = hbot;
    eu = ebot;
    asm volatile(
        "s_mov_b64 vcc, 1\n\t"
        "s_nop 0\n\t"      /* 2 wait states between a VALU write of cch just before the block and its DPP read */
        "v_cndmask_b32_dpp %0, %0, %5, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_lshl_b64 vcc, 1, 63\n\t"
        "v_cndmask_b32_dpp %3, %3, %1, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_cndmask_b32_dpp %4, %4, %2, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_mov_b64 vcc, 1\n\t"
        "v_cndmask_b32_dpp %1, %1, %6, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %2, %7, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(cch), "+v"(hu), "+v"(eu), "+v"(sh), "+v"(se), "+v"(ach), "+v"(th), "+v"(te)
        :
        : "vcc", "scc");
}

// The same hand-off for kernels whose waves run ALONE on their SIMD (the single-pair trace fills of psa_trace.hip: one wave per
// band, the chain of dependent steps is the run time): such a wave pays ~5 cycles for every instruction it issues, scalar
// ones included, so the block is reordered to need one VCC switch instead of two and no s_nop (10 instead of 12 issue slots).
// Contract of the caller: sh / se are written by nothing but this block and the initialisation of a tile.
__device__ __forceinline__ void lane_exchange_chain(unsigned& cch, unsigned& hu, unsigned& eu, unsigned& sh, unsigned& se,
                                              unsigned& ach, unsigned& th, unsigned& te, unsigned hbot, unsigned ebot)
{
    // hu / eu are produced IN PLACE in the registers holding hbot / ebot (their last readers come first)
    hu = hbot;
    eu = ebot;
    // Ten instructions (a lone wave pays ~5 cycles for EVERY instruction it issues, scalar ones included): the two captures
    // first, under the lane-63 mask — they read hbot / ebot before hu / eu overwrite them in place — then everything that
    // injects at lane 0 under ONE mask.  No s_nop: the DPP sources sh / se were last written a step ago (at a tile's first
    // step a stale read would only shift in garbage that is shifted out again before anything is stored), cch a step ago,
    // hbot / ebot at least four instructions before their DPP reads.
    asm volatile(
        "s_lshl_b64 vcc, 1, 63\n\t"
        "v_cndmask_b32_dpp %3, %3, %1, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_cndmask_b32_dpp %4, %4, %2, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_mov_b64 vcc, 1\n\t"
        "v_cndmask_b32_dpp %0, %0, %5, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %1, %6, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %2, %7, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(cch), "+v"(hu), "+v"(eu), "+v"(sh), "+v"(se), "+v"(ach), "+v"(th), "+v"(te)
        :
        : "vcc", "scc");
}

// lane_exchange for a column character that travels as TWO registers (the per-pair byte tables of the v_perm path)
__device__ __forceinline__ void lane_exchange2(unsigned& cch, unsigned& cch2, unsigned& hu, unsigned& eu, unsigned& sh, unsigned& se,
                                               unsigned& ach, unsigned& ach2, unsigned& th, unsigned& te, unsigned hbot, unsigned ebot)
{
    hu = hbot;
    eu = ebot;
    asm volatile(
        "s_mov_b64 vcc, 1\n\t"
        "s_nop 0\n\t"      /* 2 wait states between a VALU write of cch just before the block and its DPP read */
        "v_cndmask_b32_dpp %0, %0, %5, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %8, %8, %9, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_lshl_b64 vcc, 1, 63\n\t"
        "v_cndmask_b32_dpp %3, %3, %1, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_cndmask_b32_dpp %4, %4, %2, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_mov_b64 vcc, 1\n\t"
        "v_cndmask_b32_dpp %1, %1, %6, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %2, %7, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(cch), "+v"(hu), "+v"(eu), "+v"(sh), "+v"(se), "+v"(ach), "+v"(th), "+v"(te), "+v"(cch2), "+v"(ach2)
        :
        : "vcc", "scc");
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
