// cell_microbench.hip — issue cost of the packed kernel's steady-state cell (elided + byte-permute form: 8 VALU per
// cell pair) in several instruction orders / class mixes, at 1..3 waves per SIMD.  Diagnostic tool, not product.
//   hipcc --offload-arch=gfx950 -O3 -o cell_microbench tools/cell_microbench.hip && ./cell_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// registers: %0 eu (vertical chain), %1 d (diagonal), %2..%5 fb of 4 rows, %6..%9 hl of 4 rows, %10 tt, %11 f, %12 dg/h, %13 g
// constants: %14 pf, %15 po, %16 table A, %17 table B, %18..%21 selectors
#define KERNEL(NAME, BODY, NINSTR)                                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters, int seed) {    \
        unsigned eu = threadIdx.x + seed + 40000, d = eu + 3, fb0 = eu + 5, fb1 = eu + 6, fb2 = eu + 7, fb3 = eu + 8, \
                 hl0 = eu + 9, hl1 = eu + 10, hl2 = eu + 11, hl3 = eu + 12, tt = 0, f = 0, h = 0, g = 0;                 \
        unsigned pf = 6 + seed, po = 4 + seed, ta = 0x07070007, tb = 0x07000707, s0 = 0x0c040c00, s1 = 0x0c050c01,       \
                 s2 = 0x0c060c02, s3 = 0x0c070c03;                                                   \
        asm volatile("" : "+v"(pf), "+v"(po), "+v"(ta), "+v"(tb), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();  \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(BODY : "+v"(eu), "+v"(d), "+v"(fb0), "+v"(fb1), "+v"(fb2), "+v"(fb3), "+v"(hl0), "+v"(hl1),     \
                         "+v"(hl2), "+v"(hl3), "+v"(tt), "+v"(f), "+v"(h), "+v"(g)                                      \
                         : "v"(pf), "v"(po), "v"(ta), "v"(tb), "v"(s0), "v"(s1), "v"(s2), "v"(s3));                      \
        }                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();  \
        if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; if (blockIdx.x == 0 && threadIdx.x == 0) out[1 << 20] = q1 - q0; } \
        if (eu + d + fb0 + fb1 + fb2 + fb3 + hl0 + hl1 + hl2 + hl3 + tt + f + h + g == 0x7fffffff) out[0] = 0;          \
    }

// one row as the compiler emits it (fb in FB, hl in HL, selector SEL)
#define ROW(FB, HL, SEL, SUBOP)                          \
    SUBOP " %11, " FB ", %14\n"                          \
    "v_perm_b32 %10, %17, %16, " SEL "\n"                \
    SUBOP " %12, %1, %10\n"                              \
    "v_pk_max_u16 %12, %12, %0\n"                        \
    "v_pk_max_u16 %12, %12, %11\n"                       \
    SUBOP " %13, %12, %15\n"                             \
    "v_mov_b32 %1, " HL "\n"                             \
    "v_pk_max_u16 " FB ", %11, %13\n"                    \
    "v_pk_max_u16 %0, %0, %13\n"
// (the v_mov stands for the register renaming the unrolled real loop does for free; counted separately below)
#define ROWN(FB, HL, SEL, SUBOP)                         \
    SUBOP " %11, " FB ", %14\n"                          \
    "v_perm_b32 %10, %17, %16, " SEL "\n"                \
    SUBOP " %12, " HL ", %10\n"                          \
    "v_pk_max_u16 %12, %12, %0\n"                        \
    "v_pk_max_u16 " HL ", %12, %11\n"                    \
    SUBOP " %13, " HL ", %15\n"                          \
    "v_pk_max_u16 " FB ", %11, %13\n"                    \
    "v_pk_max_u16 %0, %0, %13\n"
KERNEL(k_rows, ROWN("%2", "%6", "%18", "v_sub_u32") ROWN("%3", "%7", "%19", "v_sub_u32") ROWN("%4", "%8", "%20", "v_sub_u32") ROWN("%5", "%9", "%21", "v_sub_u32"), 32)
KERNEL(k_rows_pk, ROWN("%2", "%6", "%18", "v_pk_sub_u16") ROWN("%3", "%7", "%19", "v_pk_sub_u16") ROWN("%4", "%8", "%20", "v_pk_sub_u16") ROWN("%5", "%9", "%21", "v_pk_sub_u16"), 32)
// only the five slow ops of a row
#define ROWSLOW(FB, HL, SEL)                             \
    "v_perm_b32 %10, %17, %16, " SEL "\n"                \
    "v_pk_max_u16 %12, " HL ", %0\n"                     \
    "v_pk_max_u16 " HL ", %12, " FB "\n"                 \
    "v_pk_max_u16 " FB ", " FB ", %10\n"                 \
    "v_pk_max_u16 %0, %0, " HL "\n"
KERNEL(k_rows_slow, ROWSLOW("%2", "%6", "%18") ROWSLOW("%3", "%7", "%19") ROWSLOW("%4", "%8", "%20") ROWSLOW("%5", "%9", "%21"), 20)
// only the three fast ops of a row
#define ROWFAST(FB, HL)                                  \
    "v_sub_u32 %11, " FB ", %14\n"                       \
    "v_sub_u32 %12, " HL ", %10\n"                       \
    "v_sub_u32 %13, " HL ", %15\n"
KERNEL(k_rows_fast, ROWFAST("%2", "%6") ROWFAST("%3", "%7") ROWFAST("%4", "%8") ROWFAST("%5", "%9"), 12)
// clustered: the independent fast/lookup ops of 4 rows first, then the dependent chains
#define PRE(FB, SEL, T) "v_sub_u32 " FB ", " FB ", %14\n" "v_perm_b32 " T ", %17, %16, " SEL "\n"
KERNEL(k_clustered,
    "v_sub_u32 %2, %2, %14\nv_sub_u32 %3, %3, %14\nv_sub_u32 %4, %4, %14\nv_sub_u32 %5, %5, %14\n"
    "v_perm_b32 %10, %17, %16, %18\nv_perm_b32 %11, %17, %16, %19\nv_perm_b32 %12, %17, %16, %20\nv_perm_b32 %13, %17, %16, %21\n"
    "v_sub_u32 %6, %6, %10\nv_sub_u32 %7, %7, %11\nv_sub_u32 %8, %8, %12\nv_sub_u32 %9, %9, %13\n"
    "v_pk_max_u16 %6, %6, %0\nv_pk_max_u16 %6, %6, %2\nv_sub_u32 %10, %6, %15\nv_pk_max_u16 %2, %2, %10\nv_pk_max_u16 %0, %0, %10\n"
    "v_pk_max_u16 %7, %7, %0\nv_pk_max_u16 %7, %7, %3\nv_sub_u32 %10, %7, %15\nv_pk_max_u16 %3, %3, %10\nv_pk_max_u16 %0, %0, %10\n"
    "v_pk_max_u16 %8, %8, %0\nv_pk_max_u16 %8, %8, %4\nv_sub_u32 %10, %8, %15\nv_pk_max_u16 %4, %4, %10\nv_pk_max_u16 %0, %0, %10\n"
    "v_pk_max_u16 %9, %9, %0\nv_pk_max_u16 %9, %9, %5\nv_sub_u32 %10, %9, %15\nv_pk_max_u16 %5, %5, %10\nv_pk_max_u16 %0, %0, %10\n", 32)
// plain streams for reference
#define R8(x) x x x x x x x x
KERNEL(k_perm, R8("v_perm_b32 %10, %17, %16, %18\nv_perm_b32 %11, %17, %16, %19\nv_perm_b32 %12, %17, %16, %20\nv_perm_b32 %13, %17, %16, %21\n"), 32)
KERNEL(k_pkmax, R8("v_pk_max_u16 %2, %2, %14\nv_pk_max_u16 %3, %3, %14\nv_pk_max_u16 %4, %4, %14\nv_pk_max_u16 %5, %5, %14\n"), 32)
KERNEL(k_sub, R8("v_sub_u32 %2, %2, %14\nv_sub_u32 %3, %3, %14\nv_sub_u32 %4, %4, %14\nv_sub_u32 %5, %5, %14\n"), 32)
// alternating fast / slow, independent
KERNEL(k_alt, R8("v_sub_u32 %2, %2, %14\nv_pk_max_u16 %3, %3, %14\nv_sub_u32 %4, %4, %14\nv_pk_max_u16 %5, %5, %14\n"), 32)
// 32-bit max instead of packed (is VOP3P the slow part, or max itself?)
KERNEL(k_max32, R8("v_max_u32 %2, %2, %14\nv_max_u32 %3, %3, %14\nv_max_u32 %4, %4, %14\nv_max_u32 %5, %5, %14\n"), 32)
KERNEL(k_alt32, R8("v_sub_u32 %2, %2, %14\nv_max_u32 %3, %3, %14\nv_sub_u32 %4, %4, %14\nv_max_u32 %5, %5, %14\n"), 32)

// round 4: the three-input packed maximum of gfx950 (v_pk_maximum3_f16) on BIT PATTERNS: for patterns in [0x0400, 0x7bff]
// (positive normal halves: no NaN / inf / denormal / zero) the IEEE order is the unsigned integer order, so it is an exact
// u16 max3 there.  Does it issue like v_pk_max_u16?
KERNEL(k_pkmax3, R8("v_pk_maximum3_f16 %2, %2, %14, %15\nv_pk_maximum3_f16 %3, %3, %14, %15\nv_pk_maximum3_f16 %4, %4, %14, %15\nv_pk_maximum3_f16 %5, %5, %14, %15\n"), 32)
KERNEL(k_pkmaxf, R8("v_pk_max_f16 %2, %2, %14\nv_pk_max_f16 %3, %3, %14\nv_pk_max_f16 %4, %4, %14\nv_pk_max_f16 %5, %5, %14\n"), 32)
// the elided row with it: 7 instead of 8 instructions
#define ROW3(FB, HL, SEL)                                \
    "v_sub_u32 %11, " FB ", %14\n"                       \
    "v_perm_b32 %10, %17, %16, " SEL "\n"                \
    "v_sub_u32 %12, " HL ", %10\n"                       \
    "v_pk_maximum3_f16 " HL ", %12, %0, %11\n"           \
    "v_sub_u32 %13, " HL ", %15\n"                       \
    "v_pk_max_u16 " FB ", %11, %13\n"                    \
    "v_pk_max_u16 %0, %0, %13\n"
KERNEL(k_rows3, ROW3("%2", "%6", "%18") ROW3("%3", "%7", "%19") ROW3("%4", "%8", "%20") ROW3("%5", "%9", "%21"), 28)
// ... and the free-free frame's row (no f-sub): 6
#define ROW3FF(FB, HL, SEL)                              \
    "v_perm_b32 %10, %17, %16, " SEL "\n"                \
    "v_add_u32 %12, " HL ", %10\n"                       \
    "v_pk_maximum3_f16 " HL ", %12, %0, " FB "\n"        \
    "v_sub_u32 %13, " HL ", %15\n"                       \
    "v_pk_max_u16 " FB ", " FB ", %13\n"                 \
    "v_pk_max_u16 %0, %0, %13\n"
KERNEL(k_rows3ff, ROW3FF("%2", "%6", "%18") ROW3FF("%3", "%7", "%19") ROW3FF("%4", "%8", "%20") ROW3FF("%5", "%9", "%21"), 24)

// exactness of the bit-pattern trick: every pair of halves (a, b) on a grid over [0x0400, 0x7bff] and a third operand
__global__ void k_max3_exact(unsigned* bad, unsigned lo, unsigned hi)
{
    const unsigned a = lo + blockIdx.x;                        // one block per value of a
    unsigned n = 0;
    for (unsigned b = lo + threadIdx.x; b <= hi; b += blockDim.x) {
        const unsigned c = lo + ((a * 2654435761u + b * 40503u) % (hi - lo + 1));
        const unsigned pa = a | (b << 16), pb = b | (c << 16), pc = c | (a << 16);
        unsigned r;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(pa), "v"(pb), "v"(pc));
        const unsigned m = a > b ? (a > c ? a : c) : (b > c ? b : c);
        if ((r & 0xffffu) != m || (r >> 16) != m) ++n;
    }
    if (n) atomicAdd(bad, n);
}

typedef void (*kern_t)(unsigned long long*, int, int);
struct Test { const char* name; kern_t k; int instr; };

int main()
{
    Test tests[] = {{"4 rows, compiler order (32)", k_rows, 32}, {"4 rows, subs as v_pk_sub_u16 (32)", k_rows_pk, 32},
                    {"4 rows, slow ops only (20)", k_rows_slow, 20}, {"4 rows, fast ops only (12)", k_rows_fast, 12},
                    {"4 rows, clustered (32)", k_clustered, 32}, {"v_perm_b32", k_perm, 32}, {"v_pk_max_u16", k_pkmax, 32},
                    {"v_sub_u32", k_sub, 32}, {"alternating sub / pk_max", k_alt, 32}, {"v_max_u32", k_max32, 32},
                    {"alternating sub / max_u32", k_alt32, 32}, {"v_pk_maximum3_f16", k_pkmax3, 32}, {"v_pk_max_f16", k_pkmaxf, 32},
                    {"4 rows with pk_maximum3 (28)", k_rows3, 28}, {"4 rows, free-free, pk_maximum3 (24)", k_rows3ff, 24}};
    {   // exactness first
        unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
        const unsigned lo = 0x0400, hi = 0x7bff;
        hipLaunchKernelGGL(k_max3_exact, dim3(hi - lo + 1), dim3(256), 0, 0, bad, lo, hi);
        unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("v_pk_maximum3_f16 as u16 max3 on patterns [0x%04x, 0x%04x]: %u mismatches over %u x %u triples\n", lo, hi, h, hi - lo + 1, hi - lo + 1);
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_max3_exact, dim3(0x7fff - 0x0001 + 1), dim3(256), 0, 0, bad, 0x0001u, 0x7fffu);
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("   (outside that range, patterns [0x0001, 0x7fff]: %u mismatches — denormals / NaN patterns)\n", h);
        hipFree(bad);
    }
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    unsigned long long* d; hipMalloc(&d, sizeof(unsigned long long) * ((1 << 20) + 8));
    const int iters = 100000, rounds = 6;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-40s", "cycles per BODY per SIMD @ waves/SIMD:");
    int ws[] = {1, 2, 3, 4};
    for (int w : ws) printf("%14d", w);
    printf("   (cycles per instruction at 2 waves)\n");
    for (auto& t : tests) {
        printf("%-40s", t.name);
        double at2 = 0;
        for (int w : ws) {
            // residency limited to w blocks (= w waves per SIMD) per CU through dynamic LDS; `rounds` blocks per slot so
            // that placement imbalance averages out; wall time by events, clock from wave 0's two counters
            const int blocks = ncu * w * rounds;
            const size_t lds = (size_t)(160 * 1024 / w) - 2048;
            hipFuncSetAttribute((const void*)t.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            double best = 1e30, mhz = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), lds, 0, d, iters, 1);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long q[2]; hipMemcpy(&q[0], d, 8, hipMemcpyDeviceToHost); hipMemcpy(&q[1], d + (1 << 20), 8, hipMemcpyDeviceToHost);
                const double f = (double)q[0] / (double)q[1] * 100.0;         // MHz
                const double cyc = (double)ms * 1e-3 * f * 1e6 / ((double)iters * rounds * w);
                if (cyc < best) { best = cyc; mhz = f; }
            }
            if (w == 2) at2 = best / t.instr;
            printf("%9.1f@%4.0f", best, mhz);
        }
        printf("   %.2f\n", at2);
    }
    hipFree(d);
    return 0;
}
