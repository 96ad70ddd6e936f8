// poa_types.h — arguments of the POA fill kernel (poa_fill.hip: tile wavefront), the best-sink reduction and the walker.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct tsta_params;
struct tsta_dag_csr;

namespace tsta {

struct Engine;

constexpr int kNeg = -1000000000;
// columns per lane KC = 4 (256-column tiles) or 2 (128-column tiles: more tiles in flight for short sequences)
constexpr int kRB = 32;                  // graph nodes (rows) per tile; <= 64 (per-row side data lives one row per lane)

struct PoaArgs {
    int V, n;
    const int* pred_off; const int* pred; const uint8_t* base; const uint8_t* is_sink;
    const uint8_t* seq;                  // n bytes
    const int* slot;                     // [V]: row slot of node r (slots are recycled once no later tile needs the row)
    const int* pred_slot;                // parallel to pred[]: slot of that predecessor
    const int* ptab;                     // [V][4]: the first four predecessors of every node, -1 padded (the walker's window staging)
    int* Hrow;                           // [nslots][pitch]: Hrow[slot][c+3] = H(r,c), c = 0..n   (slot 0 = virtual init)
    int* Erow;                           // [nslots][pitch]: Erow[slot][c]   = Eagg(r,c+1), c = 0..n-1
    int64_t pitch;                       // ints per row, multiple of 4
    int* carry;                          // [V][nchunks+1]: Fbar entering chunk q of row r
    uint8_t* cflag;                      // [V][nchunks+1]: "F at the first column of chunk q is an open"
    int nchunks, nrb, d, q_lo;           // launch d runs the tiles (q, rb = d - q)
    int kc;                              // columns per lane of this run (2 or 4)
    int8_t* source; int8_t* esource; int8_t* fsource; int64_t tpitch;   // [V][tpitch] device layout (tpitch mult. of 8)
    int* lastscore;                      // [V]
    int* best;                           // [0] = best sink; [4..7] walker results; [8] = largest spread seen between two predecessor rows of a node
    int M, X, E, O;
    int links;                           // 1: fsource bytes carry, above the F code's 3 bits, the distance (1..31 rows, 0 = look it up) to the
                                         // predecessor the cell's own move goes to: the on-device walk then needs no predecessor list (K4d only;
                                         // never set when the arrays go back to the caller)
};


// poa_dataflow.hip: the single-launch fill (one workgroup per 128-column chunk, solver + emitter waves)
struct PoaDfArgs {
    PoaArgs A;
    unsigned* ticket;                    // [1]: roles in start order (0: column 0, t: chunk t-1)
    unsigned* status;
    int* prog;                           // [nchunks+1]: blocks of 32 nodes completed by the column-0 solver ([0]) and by chunk q ([q+1])
    int* hand;                           // [V][nchunks+1][4]: what chunk q takes over from the chunk on its left for row r: {F carry
                                         // entering the chunk, "its first F is an open", H(r, column left of the chunk), unused} (sc1 hand-off)
    int test_abandon;                    // test hook (handoff.h): the workgroup that draws ticket test_abandon-1 never publishes its blocks; 0 = off
};
int launch_poa_dataflow(Engine* e, const PoaArgs& A, int* d_work, size_t work_ints, int* d_hand, hipStream_t s);

// poa_ref8.hip (K4e): the fill in the reference's own int8 arithmetic and AVX512 blocking (inputs on which its output depends on them)
struct PoaRef8Args {
    PoaArgs A;                           // DAG, sequence, trace arrays, lastscore, scoring
    int W, Wq, nblk, maxp;               // columns per byte lane (L = 64 W per block), dwords per lane and row = ceil(W / 4), blocks, largest in-degree
    int fmaxtag, lmaxtag, maxtag, nrowblk;   // control()'s schedule constants, in blocks (c-t-simd.c:987-1001)
    int* rows;                           // [V][nblk][2][Wq][64]: per row and block the differences / E differences, 4 columns per dword
    int* simple;                         // [V][nblk + 1]: running sums (simple_sorce)
    int* first_col;                      // [V]: frist_col_sorce
    int* f0;                             // [V][maxp]: row->f0, the F carries into the row's next block
    unsigned* ticket; unsigned* status; int* prog;   // blocks as tickets; prog[k] = rows block k has finished
    int test_abandon, k_fixed;
};
size_t poa_ref8_lds_bytes(int maxp, int Wq);
int launch_poa_ref8(PoaRef8Args P, bool single, hipStream_t s);           // -> number of launches (-1: the LDS size was refused)

void launch_poa_init(const PoaArgs& A, hipStream_t s);      // poa_fill.hip: virtual init row into slot 0
void launch_poa_best(const PoaArgs& A, hipStream_t s);      // poa_fill.hip: best sink

}  // namespace tsta
