// lone_wave_microbench.hip — what ONE wave pays per instruction when it is alone on its SIMD (the single-pair trace solvers, the POA
// dataflow solver, K4e and both walkers are such waves: their chain of dependent steps is the run time).  Diagnostic tool, not product.
//   hipcc --offload-arch=gfx950 -O3 -o lone_wave_microbench tools/lone_wave_microbench.hip && ./lone_wave_microbench
// Every kernel runs `iters` times a body of 32 instructions in ONE wave of ONE workgroup (nothing else on the device); cycles are
// s_memtime ticks (reported against the wall clock in the first line, so the unit can be read off).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP4(x) x x x x
#define KERNEL(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(64) void NAME(unsigned long long* out, int iters, int seed) {      \
        int r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 7, r4 = r0 - 9,     \
            r5 = r0 * 5, r6 = r0 | 3, r7 = r0 & 0x3f, c0 = seed + 11, c1 = seed - 4;                \
        int s0 = seed + 11, s1 = seed - 4;                                                          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < iters; ++i) {                                                           \
            asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),        \
                         "+v"(r6), "+v"(r7), "+s"(s1) : "v"(c0), "v"(c1), "s"(s0) : "vcc", "scc", "s90", "s91"); \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                     \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + s1 == 0x7fffffff) out[1] = 0;                   \
    }
// operands: %0..%7 VGPRs (read-write), %8 SGPR (read-write), %9 %10 constant VGPRs, %11 constant SGPR

#define I4(op, a, b, c, d) op(a) op(b) op(c) op(d)
// ---- plain VALU, destination reused every 4 instructions
#define ADDV(d) "v_add_u32 " d ", " d ", %9\n"
KERNEL(k_add_v, REP8(I4(ADDV, "%0", "%1", "%2", "%3")))
#define ADDS(d) "v_add_u32 " d ", %11, " d "\n"
KERNEL(k_add_s, REP8(I4(ADDS, "%0", "%1", "%2", "%3")))
#define ADDI(d) "v_add_u32 " d ", 7, " d "\n"
KERNEL(k_add_inl, REP8(I4(ADDI, "%0", "%1", "%2", "%3")))
#define ADDL(d) "v_add_u32 " d ", 0x12345, " d "\n"
KERNEL(k_add_lit, REP8(I4(ADDL, "%0", "%1", "%2", "%3")))
#define ADDVV(d, e) "v_add_u32 " d ", " d ", " e "\n"
KERNEL(k_add_vv, REP8(ADDVV("%0", "%4") ADDVV("%1", "%5") ADDVV("%2", "%6") ADDVV("%3", "%7")))
#define MAXV(d) "v_max_i32 " d ", " d ", %9\n"
KERNEL(k_max_v, REP8(I4(MAXV, "%0", "%1", "%2", "%3")))
#define MAXS(d) "v_max_i32 " d ", %11, " d "\n"
KERNEL(k_max_s, REP8(I4(MAXS, "%0", "%1", "%2", "%3")))
#define MAXI_(d) "v_max_i32 " d ", 7, " d "\n"
KERNEL(k_max_inl, REP8(I4(MAXI_, "%0", "%1", "%2", "%3")))
// dependency distance
KERNEL(k_max_v_d1, REP8(REP4(MAXV("%0"))))
KERNEL(k_max_s_d1, REP8(REP4(MAXS("%0"))))
KERNEL(k_max_v_d2, REP8(MAXV("%0") MAXV("%1") MAXV("%0") MAXV("%1")))
KERNEL(k_max_v_d8, REP4(MAXV("%0") MAXV("%1") MAXV("%2") MAXV("%3") MAXV("%4") MAXV("%5") MAXV("%6") MAXV("%7")))
KERNEL(k_max_s_d8, REP4(MAXS("%0") MAXS("%1") MAXS("%2") MAXS("%3") MAXS("%4") MAXS("%5") MAXS("%6") MAXS("%7")))
// three operands
#define MAX3V(d) "v_max3_i32 " d ", " d ", %9, %10\n"
KERNEL(k_max3_v, REP8(I4(MAX3V, "%0", "%1", "%2", "%3")))
#define MAX3S(d) "v_max3_i32 " d ", " d ", %11, %10\n"
KERNEL(k_max3_s, REP8(I4(MAX3S, "%0", "%1", "%2", "%3")))
#define MAX3SS(d) "v_max3_i32 " d ", " d ", %11, 7\n"
KERNEL(k_max3_ss, REP8(I4(MAX3SS, "%0", "%1", "%2", "%3")))
// VOP3 encoding of a two-operand op, all VGPR
#define MAXE64(d) "v_max_i32_e64 " d ", " d ", %9\n"
KERNEL(k_max_e64, REP8(I4(MAXE64, "%0", "%1", "%2", "%3")))
#define BFE(d) "v_bfe_i32 " d ", %9, " d ", 8\n"
KERNEL(k_bfe, REP8(I4(BFE, "%0", "%1", "%2", "%3")))
#define LSHLADD(d) "v_lshl_add_u32 " d ", %9, 3, " d "\n"
KERNEL(k_lshladd, REP8(I4(LSHLADD, "%0", "%1", "%2", "%3")))
#define PKMAX(d) "v_pk_max_u16 " d ", " d ", %9\n"
KERNEL(k_pkmax_v, REP8(I4(PKMAX, "%0", "%1", "%2", "%3")))
#define PKMAXS(d) "v_pk_max_u16 " d ", " d ", %11\n"
KERNEL(k_pkmax_s, REP8(I4(PKMAXS, "%0", "%1", "%2", "%3")))
// moves
#define MOVV(d, s) "v_mov_b32 " d ", " s "\n"
KERNEL(k_mov_v, REP8(MOVV("%0", "%4") MOVV("%1", "%5") MOVV("%2", "%6") MOVV("%3", "%7")))
#define MOVS(d) "v_mov_b32 " d ", %11\n"
KERNEL(k_mov_s, REP8(I4(MOVS, "%0", "%1", "%2", "%3")))
// DPP
#define DPPW(d, s) "v_mov_b32_dpp " d ", " s " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpp_wave, REP8(DPPW("%0", "%4") DPPW("%1", "%5") DPPW("%2", "%6") DPPW("%3", "%7")))
#define DPPR(d, s) "v_mov_b32_dpp " d ", " s " row_shr:1 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpp_row, REP8(DPPR("%0", "%4") DPPR("%1", "%5") DPPR("%2", "%6") DPPR("%3", "%7")))
#define DPPC(d, s) "v_cndmask_b32_dpp " d ", " d ", " s ", vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
KERNEL(k_dpp_cnd, REP8(DPPC("%0", "%4") DPPC("%1", "%5") DPPC("%2", "%6") DPPC("%3", "%7")))
#define DPPADD(d, s) "v_add_u32_dpp " d ", " s ", " d " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpp_add, REP8(DPPADD("%0", "%4") DPPADD("%1", "%5") DPPADD("%2", "%6") DPPADD("%3", "%7")))
// compares and selects
#define CMPV(d) "v_cmp_lt_i32 vcc, " d ", %9\n"
KERNEL(k_cmp, REP8(I4(CMPV, "%0", "%1", "%2", "%3")))
#define CMPS(d) "v_cmp_lt_i32 s[90:91], " d ", %9\n"
KERNEL(k_cmp_e64, REP8(I4(CMPS, "%0", "%1", "%2", "%3")))
#define SEL(d) "v_cndmask_b32 " d ", " d ", %9, vcc\n"
KERNEL(k_sel, REP8(I4(SEL, "%0", "%1", "%2", "%3")))
// scalar
#define SADD "s_add_u32 %8, %8, %11\n"
KERNEL(k_salu, REP8(SADD SADD SADD SADD))
#define SNOP "s_nop 0\n"
KERNEL(k_snop, REP8(SNOP SNOP SNOP SNOP))
// mixes (32 instructions each)
KERNEL(k_mix_v_salu, REP8(MAXV("%0") SADD MAXV("%1") SADD))
KERNEL(k_mix_v_snop, REP8(MAXV("%0") SNOP MAXV("%1") SNOP))
KERNEL(k_mix_v_s, REP8(MAXV("%0") MAXS("%1") MAXV("%2") MAXS("%3")))
KERNEL(k_mix_v_cmp, REP8(MAXV("%0") CMPV("%1") MAXV("%2") CMPV("%3")))
KERNEL(k_mix_v_dpp, REP8(MAXV("%0") DPPW("%1", "%5") MAXV("%2") DPPW("%3", "%7")))
KERNEL(k_mix_add_max, REP8(ADDV("%0") MAXV("%1") ADDV("%2") MAXV("%3")))
KERNEL(k_mix_adds_max, REP8(ADDS("%0") MAXV("%1") ADDS("%2") MAXV("%3")))
// the solver's real dependent chain: add(SGPR) max3 add(SGPR) max | ... on one register pair
#define CHAIN "v_add_u32 %0, %11, %0\nv_max3_i32 %0, %0, %9, %10\nv_add_u32 %1, %11, %0\nv_max_i32 %0, %0, %1\n"
KERNEL(k_chain, REP8(CHAIN))
// the same chain with an independent instruction after every dependent one
#define CHAIN_I "v_add_u32 %0, %11, %0\nv_max_i32 %4, %4, %9\nv_max3_i32 %0, %0, %9, %10\nv_max_i32 %5, %5, %9\nv_add_u32 %1, %11, %0\nv_max_i32 %6, %6, %9\nv_max_i32 %0, %0, %1\nv_max_i32 %7, %7, %9\n"
KERNEL(k_chain_i, REP4(CHAIN_I))
// LDS round trip on the chain (address from the result)
#define LDSRT "ds_read_b32 %0, %4\ns_waitcnt lgkmcnt(0)\nv_and_b32 %4, 0xfc, %0\nv_add_u32 %1, %11, %1\n"
KERNEL(k_lds_rt, REP8(LDSRT))
#define LDSW "ds_write_b32 %4, %0\nv_add_u32 %1, %11, %1\nv_add_u32 %2, %11, %2\nv_add_u32 %3, %11, %3\n"
KERNEL(k_lds_w, REP8(LDSW))
#define RFL "v_readfirstlane_b32 s90, %0\ns_add_u32 %8, %8, s90\nv_add_u32 %1, %11, %1\nv_add_u32 %2, %11, %2\n"
KERNEL(k_readfirstlane, REP8(RFL))

typedef void (*kern_t)(unsigned long long*, int, int);
struct Test { const char* name; kern_t k; };

int main()
{
    Test tests[] = {
        {"v_add_u32 v,v,vconst (reuse 4)", k_add_v}, {"v_add_u32 v,SGPR,v", k_add_s}, {"v_add_u32 v,inline,v", k_add_inl}, {"v_add_u32 v,literal,v", k_add_lit},
        {"v_add_u32 v,v,v2 (no shared reg)", k_add_vv},
        {"v_max_i32 v,v,vconst", k_max_v}, {"v_max_i32 v,SGPR,v", k_max_s}, {"v_max_i32 v,inline,v", k_max_inl},
        {"v_max_i32 vconst, distance 1", k_max_v_d1}, {"v_max_i32 SGPR, distance 1", k_max_s_d1}, {"v_max_i32 vconst, distance 2", k_max_v_d2},
        {"v_max_i32 vconst, distance 8", k_max_v_d8}, {"v_max_i32 SGPR, distance 8", k_max_s_d8},
        {"v_max3_i32 v,v,v", k_max3_v}, {"v_max3_i32 v,SGPR,v", k_max3_s}, {"v_max3_i32 v,SGPR,inline", k_max3_ss},
        {"v_max_i32_e64 v,v,v", k_max_e64}, {"v_bfe_i32 vconst,v,8", k_bfe}, {"v_lshl_add_u32 vconst,3,v", k_lshladd},
        {"v_pk_max_u16 v,v,vconst", k_pkmax_v}, {"v_pk_max_u16 v,v,SGPR", k_pkmax_s},
        {"v_mov_b32 v,v", k_mov_v}, {"v_mov_b32 v,SGPR", k_mov_s},
        {"v_mov_b32_dpp wave_shr", k_dpp_wave}, {"v_mov_b32_dpp row_shr", k_dpp_row}, {"v_cndmask_b32_dpp wave_shr", k_dpp_cnd}, {"v_add_u32_dpp wave_shr", k_dpp_add},
        {"v_cmp -> vcc", k_cmp}, {"v_cmp -> s[90:91] (e64)", k_cmp_e64}, {"v_cndmask_b32 v,v,vconst,vcc", k_sel},
        {"s_add_u32", k_salu}, {"s_nop 0", k_snop},
        {"mix: v_max(v) / s_add", k_mix_v_salu}, {"mix: v_max(v) / s_nop", k_mix_v_snop}, {"mix: v_max(v) / v_max(SGPR)", k_mix_v_s},
        {"mix: v_max(v) / v_cmp", k_mix_v_cmp}, {"mix: v_max(v) / dpp", k_mix_v_dpp}, {"mix: v_add(v) / v_max(v)", k_mix_add_max}, {"mix: v_add(SGPR) / v_max(v)", k_mix_adds_max},
        {"solver chain add max3 add max", k_chain}, {"same + 1 independent each", k_chain_i},
        {"ds_read round trip + 2 valu", k_lds_rt}, {"ds_write + 3 v_add(SGPR)", k_lds_w}, {"readfirstlane + s_add + 2 v_add", k_readfirstlane}};
    unsigned long long* d; hipMalloc(&d, 64);
    const int iters = 20000;
    bool first = true;
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, d, 100, 1);
        hipDeviceSynchronize();
        const auto w0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, d, iters, 1);
        hipDeviceSynchronize();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        if (first) { printf("s_memtime ticks per second against the wall clock of one launch: %.3g (ticks %llu, wall %.3f ms incl. launch)\n", (double)h[0] / wall, h[0], wall * 1e3); first = false; }
        printf("%-40s %7.2f ticks per instruction\n", t.name, (double)h[0] / ((double)iters * 32));
    }
    hipFree(d);
    return 0;
}
