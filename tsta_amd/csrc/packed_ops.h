// packed_ops.h — two 16-bit lanes per 32-bit register: the arithmetic the packed kernels (psa_score_packed.hip,
// psa_trace_packed.hip) are built from.  Scores are stored biased and non-negative (v = H - base + kBias), so plain
// 32-bit v_add/v_sub act on both halves without carry/borrow crossing; max/min are v_pk_*_u16.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.h"

namespace tsta {

constexpr int kBias = 0x8000;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pkmin(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pksub(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pkadd(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}
// Three-input packed maximum of gfx950 ON BIT PATTERNS: v_pk_maximum3_f16 compares halves as IEEE binary16, and for patterns in
// [0x0400, 0x7bff] (positive normal numbers: no zero, denormal, infinity or NaN) that order IS the unsigned integer order, so there it
// is an exact u16 max3 in one instruction (tools/cell_microbench.hip checks all 30 720^2 pairs with a third operand: 0 mismatches;
// it issues like v_pk_max_u16, 4.06 cycles).  Callers guarantee the range (psa_score_packed.hip: bias 0x4000 + a range check per
// 64-step block).  Plain asm, not volatile: the scheduler may move it like any other VALU instruction.
constexpr unsigned kF16Lo = 0x0400u, kF16Hi = 0x7bffu;
__device__ __forceinline__ unsigned pkmax3f(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pack2(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ unsigned ushr1(unsigned old, unsigned src) { return (unsigned)wave_shr1((int)old, (int)src); }
__device__ __forceinline__ unsigned ushl1(unsigned old, unsigned src) { return (unsigned)wave_shl1((int)old, (int)src); }

constexpr int kNegInfP = -(1 << 30);

__device__ __forceinline__ int hcol0p(int i, int E, int O) { return i == 0 ? 0 : O + i * E; }   // H(i, column 0), psa.c:227-231

// per-pair int32 bookkeeping of a couple: base = anchor of the H' frame (scores are stored as H' - base + kBias),
// anc = anchor of the running max (H frame), mx = running maximum so far (true value)
struct PairAcc { int baseA, baseB, mxA, mxB, ancA, ancB; };

// fold the packed running max of the last block (0 = nothing seen) into the int32 maxima
template <int BIAS = kBias>
__device__ __forceinline__ void fold_mx(unsigned& mx, PairAcc& acc)
{
    if ((mx & 0xffffu) != 0u) acc.mxA = max(acc.mxA, (int)(mx & 0xffffu) + acc.ancA - BIAS);
    if ((mx >> 16) != 0u)     acc.mxB = max(acc.mxB, (int)(mx >> 16) + acc.ancB - BIAS);
    mx = 0u;
}

// constants pinned in VGPRs (an SGPR source does not change the 4-cycle issue of a mixed stream, but keeps the
// compiler from re-materialising the constant with v_mov before every use)
__device__ __forceinline__ unsigned in_vgpr(unsigned x) { asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(x)); return x; }

}  // namespace tsta
