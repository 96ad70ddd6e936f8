// psa_trace.hip — K2/K3 (trace fill + traceback).  Placeholder until the kernels land.
#include "../../include/tsta_engine.h"
#include "engine.h"
extern "C" int tsta_psa_align(tsta_engine* eng, const tsta_params* p, const uint8_t* s1, int64_t l1,
                              const uint8_t* s2, int64_t l2, int32_t* maxscore, char** row_a, char** row_b,
                              int64_t* out_len)
{
    (void)p; (void)s1; (void)l1; (void)s2; (void)l2; (void)maxscore; (void)row_a; (void)row_b; (void)out_len;
    if (eng) reinterpret_cast<tsta::Engine*>(eng)->set_error("tsta_psa_align: not implemented yet");
    return TSTA_ERR_UNSUPPORTED;
}
