// valu_microbench.hip — measures VALU issue cost (cycles per wave-instruction per SIMD) on gfx950 for the
// instruction kinds the DP kernels use, at 1..8 waves per SIMD.  Diagnostic tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o valu_microbench tools/valu_microbench.hip && ./valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define KERNEL(NAME, BODY)                                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters, int seed) {    \
        int r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 7, r4 = r0 - 9,    \
            r5 = r0 * 5, r6 = r0 | 3, r7 = r0 & 0x3f, c0 = seed + 11, c1 = seed - 4;               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),       \
                         "+v"(r6), "+v"(r7) : "v"(c0), "v"(c1) : "vcc");                            \
        }                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;           \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x7fffffff) out[0] = 0;                       \
    }

// 32 instructions per body, 8 independent chains
#define ADD(d) "v_add_u32 " d ", " d ", %8\n"
KERNEL(k_add, REP8(ADD("%0") ADD("%1") ADD("%2") ADD("%3")) )
#define MAXI(d) "v_max_i32 " d ", " d ", %8\n"
KERNEL(k_max, REP8(MAXI("%0") MAXI("%1") MAXI("%2") MAXI("%3")))
#define MAX3(d) "v_max3_i32 " d ", " d ", %8, %9\n"
KERNEL(k_max3, REP8(MAX3("%0") MAX3("%1") MAX3("%2") MAX3("%3")))
#define CMPSEL(d) "v_cmp_eq_u32 vcc, " d ", %8\nv_cndmask_b32 " d ", %8, %9, vcc\n"
KERNEL(k_cmpsel, REP8(CMPSEL("%0") CMPSEL("%1")))
#define DPP(d, s) "v_mov_b32_dpp " d ", " s " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpp, REP8(DPP("%0", "%1") DPP("%2", "%3") DPP("%4", "%5") DPP("%6", "%7")))
#define DPPROW(d, s) "v_mov_b32_dpp " d ", " s " row_shr:1 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpprow, REP8(DPPROW("%0", "%1") DPPROW("%2", "%3") DPPROW("%4", "%5") DPPROW("%6", "%7")))
#define PKADD(d) "v_pk_add_i16 " d ", " d ", %8\n"
KERNEL(k_pkadd, REP8(PKADD("%0") PKADD("%1") PKADD("%2") PKADD("%3")))
#define PKMAX(d) "v_pk_max_i16 " d ", " d ", %8\n"
KERNEL(k_pkmax, REP8(PKMAX("%0") PKMAX("%1") PKMAX("%2") PKMAX("%3")))
#define PKMINU(d) "v_pk_min_u16 " d ", " d ", %8\n"
KERNEL(k_pkminu, REP8(PKMINU("%0") PKMINU("%1") PKMINU("%2") PKMINU("%3")))
#define PKMAD(d) "v_pk_mad_i16 " d ", " d ", %8, %9\n"
KERNEL(k_pkmad, REP8(PKMAD("%0") PKMAD("%1") PKMAD("%2") PKMAD("%3")))
#define XOR(d) "v_xor_b32 " d ", " d ", %8\n"
KERNEL(k_xor, REP8(XOR("%0") XOR("%1") XOR("%2") XOR("%3")))
// dependent chain of adds on one register (latency)
#define DEP "v_add_u32 %0, %0, %8\n"
KERNEL(k_dep, REP8(DEP DEP DEP DEP))
// the K1 cell mix: add add cmp cndmask add max3 add max max max  (10 instr) x3 + 2 dpp = 32
#define CELL(h, f, e, d) "v_add_u32 " e ", " e ", %8\nv_add_u32 " f ", " f ", %8\nv_cmp_eq_u32 vcc, " d ", %9\n" \
    "v_cndmask_b32 " d ", %8, %9, vcc\nv_add_u32 " d ", " d ", " h "\nv_max3_i32 " h ", " d ", " e ", " f "\n"     \
    "v_add_u32 " d ", " h ", %9\nv_max_i32 " e ", " e ", " d "\nv_max_i32 " f ", " f ", " d "\nv_max_i32 %7, %7, " h "\n"
KERNEL(k_cell, CELL("%0", "%1", "%2", "%3") CELL("%0", "%4", "%2", "%3") CELL("%0", "%5", "%2", "%3") DPP("%6", "%0") DPP("%3", "%2"))

#define SUB(d) "v_sub_u32 " d ", " d ", %8\n"
KERNEL(k_sub, REP8(SUB("%0") SUB("%1") SUB("%2") SUB("%3")))
#define SUBREV(d) "v_subrev_u32 " d ", %8, " d "\n"
KERNEL(k_subrev, REP8(SUBREV("%0") SUBREV("%1") SUBREV("%2") SUBREV("%3")))
#define AND(d) "v_and_b32 " d ", " d ", %8\n"
KERNEL(k_and, REP8(AND("%0") AND("%1") AND("%2") AND("%3")))
#define LSHL(d) "v_lshlrev_b32 " d ", 1, " d "\n"
KERNEL(k_lshl, REP8(LSHL("%0") LSHL("%1") LSHL("%2") LSHL("%3")))
#define MOV(d, s) "v_mov_b32 " d ", " s "\n"
KERNEL(k_mov, REP8(MOV("%0", "%1") MOV("%2", "%3") MOV("%4", "%5") MOV("%6", "%7")))
#define CNDM(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n"
KERNEL(k_cndmask, REP8(CNDM("%0") CNDM("%1") CNDM("%2") CNDM("%3")))
#define CMPO(d) "v_cmp_eq_u32 vcc, " d ", %8\n"
KERNEL(k_cmp, REP8(CMPO("%0") CMPO("%1") CMPO("%2") CMPO("%3")))
#define MINU(d) "v_min_u32 " d ", " d ", %8\n"
KERNEL(k_minu, REP8(MINU("%0") MINU("%1") MINU("%2") MINU("%3")))
#define ADD3(d) "v_add3_u32 " d ", " d ", %8, %9\n"
KERNEL(k_add3, REP8(ADD3("%0") ADD3("%1") ADD3("%2") ADD3("%3")))
#define LSHLADD(d) "v_lshl_add_u32 " d ", " d ", 1, %8\n"
KERNEL(k_lshladd, REP8(LSHLADD("%0") LSHLADD("%1") LSHLADD("%2") LSHLADD("%3")))
#define PKMAXU(d) "v_pk_max_u16 " d ", " d ", %8\n"
KERNEL(k_pkmaxu, REP8(PKMAXU("%0") PKMAXU("%1") PKMAXU("%2") PKMAXU("%3")))
#define PKSUB(d) "v_pk_sub_u16 " d ", " d ", %8\n"
KERNEL(k_pksub, REP8(PKSUB("%0") PKSUB("%1") PKSUB("%2") PKSUB("%3")))
#define MAXU16(d) "v_max_u16 " d ", " d ", %8\n"
KERNEL(k_maxu16, REP8(MAXU16("%0") MAXU16("%1") MAXU16("%2") MAXU16("%3")))
// projected packed cell: sub sub xor pkmin sub pkmax pkmax sub pkmax pkmax add pkmax (12) x2 + 8 dpp... here 2 cells + 8 pad adds
#define PCELL(h, f, e, d) "v_sub_u32 " e ", " e ", %8\nv_sub_u32 " f ", " f ", %8\nv_xor_b32 " d ", " d ", %9\n" \
    "v_pk_min_u16 " d ", " d ", %8\nv_sub_u32 " d ", " h ", " d "\nv_pk_max_u16 " d ", " d ", " e "\nv_pk_max_u16 " h ", " d ", " f "\n" \
    "v_sub_u32 " d ", " h ", %9\nv_pk_max_u16 " e ", " e ", " d "\nv_pk_max_u16 " f ", " f ", " d "\nv_add_u32 " d ", " h ", %8\nv_pk_max_u16 %7, %7, " h "\n"
KERNEL(k_pcell, PCELL("%0", "%1", "%2", "%3") PCELL("%0", "%4", "%2", "%3") DPP("%6", "%0") DPP("%3", "%2") DPP("%5", "%0") DPP("%3", "%2") DPP("%6", "%0") DPP("%3", "%2") DPP("%5", "%0") DPP("%3", "%2"))

#define MINU16(d) "v_min_u16 " d ", " d ", %8\n"
KERNEL(k_minu16, REP8(MINU16("%0") MINU16("%1") MINU16("%2") MINU16("%3")))
#define MAXI16(d) "v_max_i16 " d ", " d ", %8\n"
KERNEL(k_maxi16, REP8(MAXI16("%0") MAXI16("%1") MAXI16("%2") MAXI16("%3")))
#define MAX3U16(d) "v_max3_u16 " d ", " d ", %8, %9\n"
KERNEL(k_max3u16, REP8(MAX3U16("%0") MAX3U16("%1") MAX3U16("%2") MAX3U16("%3")))
// unpacked 16-bit cell: sub sub xor min sub max max sub max max add max (12), x2 + 8 dpp = 32
#define UCELL(h, f, e, d) "v_sub_u32 " e ", " e ", %8\nv_sub_u32 " f ", " f ", %8\nv_xor_b32 " d ", " d ", %9\n" \
    "v_min_u16 " d ", " d ", %8\nv_sub_u32 " d ", " h ", " d "\nv_max_u16 " d ", " d ", " e "\nv_max_u16 " h ", " d ", " f "\n" \
    "v_sub_u32 " d ", " h ", %9\nv_max_u16 " e ", " e ", " d "\nv_max_u16 " f ", " f ", " d "\nv_add_u32 " d ", " h ", %8\nv_max_u16 %7, %7, " h "\n"
KERNEL(k_ucell, UCELL("%0", "%1", "%2", "%3") UCELL("%0", "%4", "%2", "%3") DPP("%6", "%0") DPP("%3", "%2") DPP("%5", "%0") DPP("%3", "%2") DPP("%6", "%0") DPP("%3", "%2") DPP("%5", "%0") DPP("%3", "%2"))
// the same cells without the DPP overhead (pure cell streams): 32 = 2 cells + 8 extra fast/slow ops dropped -> use 24 instr
KERNEL(k_ucell_nodpp, UCELL("%0", "%1", "%2", "%3") UCELL("%0", "%4", "%2", "%3"))
KERNEL(k_pcell_nodpp, PCELL("%0", "%1", "%2", "%3") PCELL("%0", "%4", "%2", "%3"))
#define CELLND(h, f, e, d) CELL(h, f, e, d)
KERNEL(k_cell_nodpp, CELLND("%0", "%1", "%2", "%3") CELLND("%0", "%4", "%2", "%3"))

#define KERNEL_S(NAME, BODY)                                                                       \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters, int seed) {    \
        int r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 7, r4 = r0 - 9,    \
            r5 = r0 * 5, r6 = r0 | 3, r7 = r0 & 0x3f; int c0 = seed + 11, c1 = seed - 4;           \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),       \
                         "+v"(r6), "+v"(r7) : "s"(c0), "s"(c1) : "vcc");                            \
        }                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;           \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x7fffffff) out[0] = 0;                       \
    }
KERNEL_S(k_max_s, REP8(MAXI("%0") MAXI("%1") MAXI("%2") MAXI("%3")))
KERNEL_S(k_pkmaxu_s, REP8(PKMAXU("%0") PKMAXU("%1") PKMAXU("%2") PKMAXU("%3")))
KERNEL_S(k_sub_s, REP8(SUB("%0") SUB("%1") SUB("%2") SUB("%3")))
// long dependent packed chain like the real kernel's eu chain, VGPR operands: max max sub max sub ...
#define CH(e, f) "v_pk_max_u16 " e ", " e ", " f "\nv_pk_max_u16 " e ", " e ", %9\nv_sub_u32 " f ", " e ", %8\nv_pk_max_u16 " e ", " e ", " f "\n"
KERNEL(k_chain, REP8(CH("%0", "%1")))
// two interleaved independent chains
KERNEL(k_chain2, CH("%0", "%1") CH("%2", "%3") CH("%0", "%1") CH("%2", "%3") CH("%0", "%1") CH("%2", "%3") CH("%0", "%1") CH("%2", "%3"))

typedef void (*kern_t)(unsigned long long*, int, int);
struct Test { const char* name; kern_t k; int instr; };

int main()
{
    Test tests[] = {{"v_add_u32", k_add, 32}, {"v_max_i32", k_max, 32}, {"v_max3_i32", k_max3, 32},
                    {"v_cmp+v_cndmask", k_cmpsel, 32}, {"v_mov_dpp wave_shr", k_dpp, 32}, {"v_mov_dpp row_shr", k_dpprow, 32},
                    {"v_pk_add_i16", k_pkadd, 32}, {"v_pk_max_i16", k_pkmax, 32}, {"v_pk_min_u16", k_pkminu, 32},
                    {"v_pk_mad_i16", k_pkmad, 32}, {"v_xor_b32", k_xor, 32}, {"dependent v_add chain", k_dep, 32},
                    {"K1 cell mix (3 cells + 2 dpp)", k_cell, 32},
                    {"v_sub_u32", k_sub, 32}, {"v_subrev_u32", k_subrev, 32}, {"v_and_b32", k_and, 32}, {"v_lshlrev_b32", k_lshl, 32},
                    {"v_mov_b32", k_mov, 32}, {"v_cndmask_b32", k_cndmask, 32}, {"v_cmp_eq_u32", k_cmp, 32}, {"v_min_u32", k_minu, 32},
                    {"v_add3_u32", k_add3, 32}, {"v_lshl_add_u32", k_lshladd, 32}, {"v_pk_max_u16", k_pkmaxu, 32},
                    {"v_pk_sub_u16", k_pksub, 32}, {"v_max_u16", k_maxu16, 32},
                    {"packed cell mix (2x12 + 8 dpp)", k_pcell, 32},
                    {"v_min_u16", k_minu16, 32}, {"v_max_i16", k_maxi16, 32}, {"v_max3_u16", k_max3u16, 32},
                    {"u16 cell mix (2x12 + 8 dpp)", k_ucell, 32}, {"u16 cells only (2x12)", k_ucell_nodpp, 24},
                    {"packed cells only (2x12)", k_pcell_nodpp, 24}, {"int32 cells only (2x10)", k_cell_nodpp, 20},
                    {"v_max_i32 SGPR operand", k_max_s, 32}, {"v_pk_max_u16 SGPR operand", k_pkmaxu_s, 32}, {"v_sub_u32 SGPR operand", k_sub_s, 32},
                    {"dependent pk chain (1 chain)", k_chain, 32}, {"dependent pk chain (2 interleaved)", k_chain2, 32}};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    unsigned long long* d; hipMalloc(&d, sizeof(unsigned long long) * ncu * 8 * 4);
    const int iters = 20000;
    printf("%-34s", "cycles per wave-instr per SIMD @ waves/SIMD:");
    int ws[] = {1, 2, 3, 4, 6, 8};
    for (int w : ws) printf("%8d", w);
    printf("\n");
    for (auto& t : tests) {
        printf("%-34s          ", t.name);
        for (int w : ws) {
            const int blocks = ncu * w;
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 100, 1);          // warm-up
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, iters, 1);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks * 4);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            // SIMD throughput from the LAST wave to finish: waves of one SIMD are served oldest-first, so the median
            // wave duration would overstate the overlap
            const double worst = (double)h.back();
            printf("%8.2f", worst / ((double)iters * t.instr) / w);
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
