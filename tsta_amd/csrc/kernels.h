// kernels.h — host<->kernel launch interface inside libtsta_hip.so (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsta {

struct PsaBatchArgs {
    const uint8_t* seqs;
    const int64_t* off1;
    const int32_t* len1;
    const int64_t* off2;
    const int32_t* len2;
    int32_t*       out;
    int2*          rows;        // per-wave boundary row workspace: [slots][row_stride] of {H, Ebar}
    int64_t        row_stride;  // >= longest column sequence in the batch
    unsigned int*  counter;     // ticket counter, zeroed before every launch
    int64_t        n_pairs;
    int32_t        M, X, E, O;
};

void launch_psa_score_batch(int R, const PsaBatchArgs& args, int blocks, hipStream_t stream);
int  psa_score_blocks_per_cu(int R);
void launch_dpp_selftest(int* d_out, hipStream_t stream);

}  // namespace tsta
