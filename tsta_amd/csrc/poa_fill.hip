// poa_fill.hip — K4 (sequence-vs-DAG fill).  Placeholder until the kernel lands.
#include "../../include/tsta_engine.h"
#include "engine.h"
extern "C" int tsta_poa_fill(tsta_engine* eng, const tsta_params* p, const tsta_dag_csr* dag, const uint8_t* seq,
                             int32_t n, int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore,
                             int32_t* best_sink)
{
    (void)p; (void)dag; (void)seq; (void)n; (void)source; (void)esource; (void)fsource; (void)lastscore; (void)best_sink;
    if (eng) reinterpret_cast<tsta::Engine*>(eng)->set_error("tsta_poa_fill: not implemented yet");
    return TSTA_ERR_UNSUPPORTED;
}
