/*
 * tsta_engine.h — C-ABI of the MI355X-native alignment engine (libtsta_hip.so).
 *
 * Drop-in boundary for TSTA's SIMD hot path.  The reference has no library API: its hot path is reached
 * through three `main()`s and a `void block_alignment(void* pack)` thread-pool task that communicates
 * through process globals (psa/psa.c:113-137,365-433; msa/c-t-simd.c:26-34,602-657).  Each entry point
 * below replaces one such region of the reference; the citation says which.  Plain C, plain pointers and
 * sizes; no HIP, C++ or torch types.  Every function returns a tsta_status and never calls exit() (the reference
 * exit()s / crashes on errors, seqio.c:38,47,180; c-t-simd.c:154).
 *
 * STATUS CONVENTION: rc < 0 is a failure (outputs invalid), rc == 0 success, rc > 0 success WITH A WARNING (outputs valid;
 * today only TSTA_WARN_REFERENCE_UNDEFINED: from every tsta_psa_* entry point when the SCORING lies outside the domain in which
 * the reference's int8 encoding is exact (tsta_psa_reference_exact), and from tsta_poa_fill / tsta_poa_align where the
 * reference-arithmetic path cannot run).  Test with TSTA_FAILED(rc), never with rc != TSTA_OK.
 *
 * Threading: one engine per host thread / per GPU.  An engine owns its device workspaces.
 */
#ifndef TSTA_ENGINE_H
#define TSTA_ENGINE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum tsta_status {
    /* > 0: the call SUCCEEDED and its outputs are valid, but the REFERENCE's own output on this input is not a function
     * of the input and the scoring alone: it depends on -W / the SIMD width of the build (tsta_psa_reference_exact documents
     * the PSA case, tsta_poa_fill / tsta_poa_align the POA cases).  tsta_last_warning() says which one was seen. */
    TSTA_WARN_REFERENCE_UNDEFINED = 1,
    TSTA_OK = 0,
    TSTA_ERR_INVALID_ARG = -1,
    TSTA_ERR_NO_DEVICE = -2,      /* no HIP device / extension cannot run: there is NO CPU fallback */
    TSTA_ERR_HIP = -3,            /* a HIP runtime call failed; see tsta_last_error() */
    TSTA_ERR_OOM = -4,
    TSTA_ERR_UNSUPPORTED = -5,    /* input outside the supported domain (documented per function) */
    TSTA_ERR_INTERNAL = -6
} tsta_status;

#define TSTA_FAILED(rc)    ((rc) < 0)     /* failure: outputs invalid */
#define TSTA_SUCCEEDED(rc) ((rc) >= 0)    /* success, possibly with a TSTA_WARN_* (rc > 0) */

/* Scoring parameters = the reference's globals M, X, E, O (psa.c:114-117, msa.c:22-25).
 * A gap of length k costs O + k*E. */
typedef struct tsta_params {
    int32_t M;   /* -M match,      default  2 */
    int32_t X;   /* -X mismatch,   default -5 */
    int32_t E;   /* -E gap extend, default -2 */
    int32_t O;   /* -O gap open,   default -4 */
} tsta_params;

typedef struct tsta_engine tsta_engine;

const char* tsta_strerror(int status);
const char* tsta_last_error(const tsta_engine* eng);       /* text of the last failing HIP call */
const char* tsta_last_warning(const tsta_engine* eng);     /* text behind the last TSTA_WARN_* status ("" if none) */
int  tsta_device_count(void);                              /* <0: tsta_status */
int  tsta_engine_create(int device, tsta_engine** out);
void tsta_engine_destroy(tsta_engine* eng);
void tsta_free(void* p);                                   /* frees host buffers returned by this library */
/* An engine keeps its device workspaces between calls and only grows them (a 64-pair batch of 50 kbp alignments holds
 * ~165 GB of trace).  This hands them back to the device; the next call allocates what it needs.  Waits for the
 * engine's stream.  Do not call while a *_dev call of this engine is still in flight on another stream. */
int  tsta_engine_release_workspaces(tsta_engine* eng);

/* Tuning knobs (tests / experiments; results never depend on them).  Keys:
 *   "psa_packed"      -1 auto (default) | 0 never use the packed 16-bit kernel | 8 / 16 / 32 force it with that many rows per lane
 *   "psa_bands"        -1 auto (default) | 0 never | 8 / 16: run the bands of every couple concurrently (band pipeline, K1b)
 *                      with that many rows per lane; auto uses it for batches too small to fill the chip with K1p
 *   "psa_elide"        1 (default) | 0: packed kernel keeps the running max in every block, also where a bound shows
 *                      the block cannot raise the pair's maximum (exact either way; see psa_score_packed.hip)
 *   "psa_ffree"        1 (default) | 0: never use the frame with both gap extensions free (scorings with X >= -2|E|)
 *   "psa_perm"         1 (default) | 0: never use the byte-permute mismatch lookup for pure-ACGT pairs
 *   "psa_f16max"       1 (default) | 0: the packed score kernel keeps its biased 16-bit scores inside the bit patterns of positive normal
 *                      binary16 numbers where a band's plausible spread allows, and takes the three-way maximum of a cell as ONE
 *                      v_pk_maximum3_f16 in every 64-step block whose operands provably stay there (exact: for those patterns the
 *                      IEEE order is the integer order); a couple whose scores leave the 16-bit window is scored by the int32 kernel.
 *                      1 = where the host's estimate of a band's spread fits that range | 0 = round-3 behaviour (two v_pk_max_u16) |
 *                      2 = whatever the estimate says | 3 / 4 (tests): as 2 without the fast step / with every couple to the int32 kernel
 *   "psa_overlap"      1 (default) | 0: the host-pointer score forms (tsta_psa_score_batch[_multi]) cut a batch of at least two balanced
 *                      launches (4 couples per SIMD each: 8 192 pairs on 256 CUs) into such sub-batches and upload sub-batch j + 1 on a
 *                      second stream while j computes (only the first sub-batch's upload is exposed); 0 = one upload, then the kernels
 *   "psa_rows"         0 auto | 4 / 8 / 16 rows per lane of the int32 score kernel
 *   "trace_packed"     1 (default) | 0: tsta_psa_align_batch never couples two pairs per wave (packed 16-bit trace fill)
 *   "trace_streams"    0 auto (default: up to 4 overlapping launch sequences in tsta_psa_align_batch) | 1 one stream
 *   "trace_rows"       0 auto | 1 / 2 / 4  rows per lane of the trace-fill kernel (1: tsta_psa_align only)
 *   "trace_tile"       0 auto | multiple of 64: steps per tile of the trace-fill pipeline
 *   "trace_persistent" 1 (default): bands that fill alone (one pair per wave) run as ONE launch, band to band hand-off through
 *                      sc1 stores / polls without fences (psa_trace.hip) | 0: one launch per tile diagonal
 *   "poa_cols"         0 auto | 2 / 4      columns per lane of the POA fill kernel
 *   "align_ws_gb"      0 auto (default: tsta_psa_align_batch works in groups of <= 16 GB of trace on an engine's first call,
 *                      64 GB on the second, everything after that: device memory costs ~35 ms per GB to allocate) | cap in GB
 *   "trace_lds"        1 (default; 2: also with more workgroups than compute units): single-pair trace fill hands the bottom row of three of every four bands to the next band
 *                      through LDS rings inside a workgroup (psa_trace_quad_kernel) | 0: every hand-off through memory
 *   "trace_split"      1 (default): single-pair trace fill with >= 16 bands runs the bands as SOLVERS (the sweep without the trace
 *                      bytes: only what the next step needs is on the chain of dependent steps) and computes the trace bytes in
 *                      independent EMITTER tiles that restart from checkpoints the solvers leave (psa_trace_split_kernel) | 0: never |
 *                      2: at any size (tests)
 *   "walk_table"       1 (default): the traceback walk of ONE pair with >= 192 bands of 128 rows crosses its bands through a
 *                      table computed for all entry columns of a corridor at once (psa_trace.hip, K3t; a band whose entry is not in
 *                      the table is walked step by step as before: the ops are the same either way) | 0: never | 2: from one band up (tests)
 *   "walk_corridor"    512 (default) | multiple of 256: width in columns of that corridor, centred on the straight line between
 *                      the corners of the matrix (the table's cost grows with it: 0.3 / 0.55 / 1.5 ms at 256 / 512 / 1024 on a
 *                      100 kbp pair)
 *   "poa_dataflow"     1 (default): POA fill as one launch, one workgroup per 128-column chunk with a solver wave, five emitter
 *                      waves and a fetcher wave (poa_dataflow.hip; sequences without 'N').  Roles are tickets drawn in start
 *                      order, so a workgroup only waits for one that is already running; should a launch abandon its (bounded)
 *                      waits all the same, the call is run again through the other form | 0: always one launch per tile
 *                      anti-diagonal | 2: as 1, and every call is treated as abandoned once (tests of that second run)
 *   "poa_ref_arith"    1 (default): a fill on which the reference's output is a property of its int8 arithmetic and blocking (see
 *                      tsta_poa_fill: a literal 'N', a sequence spanning more blocks than the graph, long padding at large -W,
 *                      predecessor rows beyond its trackers) runs in that arithmetic, in the blocking of the reference's AVX512
 *                      build at -W = "poa_ref_block" / 64 (poa_ref8.hip), and returns the reference's bytes; needs that W in
 *                      1..32 and (largest in-degree) x (2 ceil(W/4) + 3) x 256 B of LDS <= 152 KB | 0: never (plain DP + warning) |
 *                      2: every fill (tests).  "poa_n_blocked" is accepted as an older name of the same option
 *   "test_abandon"     0 (default) | t >= 1: in every single-launch form the holder of ticket t-1 computes its work but never
 *                      publishes it, so that its dependants run into their bounded waits (tests of the second run; see
 *                      tsta_psa_score_batch_dev_status)
 *   "poa_ref_block"    L = block * W of the reference build whose behaviour TSTA_WARN_REFERENCE_UNDEFINED is judged against
 *                      (default 640 = AVX512 x -W 10; the TSTA_msa drop-in passes 64 x its -W); multiple of 16, 16 .. 1048576
 * Returns TSTA_ERR_INVALID_ARG for an unknown key or value. */
int  tsta_engine_set_option(tsta_engine* eng, const char* key, int64_t value);

/* ------------------------------------------------------------------------------------------------
 * PSA, score only  ==  TSTA_psa_notrace's fill: blockmatrix_init + the block-anti-diagonal wavefront of
 * block_alignment/row() + the `ms` reduction (psa.c:219-232, 234-433, 593-639) for MANY independent pairs.
 * out[k] = the value the reference prints as `maxsorce` (psa.c:666) = max(-100, max_{i,j>=1} H(i,j)).
 * The longer sequence of a pair becomes the column sequence (psa.c:194-199).
 *
 * seqs: concatenated sequence bytes (any byte values; compared for equality, case-sensitive, psa.c:253).
 * off1/len1, off2/len2: start offset and length of the -1 / -2 sequence of each pair.
 * Exactness domain of the engine: |H| must fit int32 (always true for lengths < 2^24 and |params| <= 127).
 *
 * WHAT IS COMPUTED, AND WHERE IT IS WHAT THE REFERENCE PRINTS.  Every tsta_psa_* entry point computes the plain integer
 * affine-gap DP.  The reference computes the same DP in int8 DIFFERENCES with wrap-around adds and
 * MIN = -100 as minus infinity (psa.c:110,219-232,244-246,285-327); inside the scoring domain below its output is that DP for
 * every input, independent of ISA, -T and -W (88 + 200 golden cases from the compiled reference, three ISAs, two -W).  Outside
 * it the reference's int8 arithmetic wraps: its printed maxsorce and its alignment then depend on the SIMD width of the build and
 * on -W (e.g. -E -30: 126379 at -W 10, 155324 at -W 3, where the DP gives 518) — "reference-undefined".  There the engine still
 * returns the plain DP, and says so: the call returns TSTA_WARN_REFERENCE_UNDEFINED (> 0, outputs valid) and
 * tsta_last_warning() names the first condition that failed.  With g = O + E, S = max(M, 0), s = min(M, X) the domain is
 *     E <= 0, O <= 0, X <= 0,   E >= -28,   2g + E >= -128,   S - g <= 127,   s >= -128,   max(s, 2g) + g >= -128,
 *     2g >= -100  or  s + O >= -100
 * (each is one statement of row() staying inside int8, or MIN really acting as minus infinity; engine.hip: psa_reference_exact;
 * X <= 0 because the reference's maximum also runs over its padded columns, psa.c:333-357).  Conservative by construction; scanned
 * against the compiled reference on tests/golden/psa_domain_golden.json: no in-domain case differs from the DP, every scanned
 * out-of-domain case does.  One more reference-undefined input is NOT detectable: the reference leaves the padding of its column
 * sequence uninitialised (psa.c:206-210), so a padded column whose garbage byte equals a base can raise its maxsorce.
 * ------------------------------------------------------------------------------------------------ */

/* 1: the scoring is inside the domain above (the reference prints what the engine computes), 0: outside; < 0: tsta_status.
 * why (may be NULL) receives the text tsta_last_warning() would carry ("" inside the domain).  Pure host arithmetic. */
int tsta_psa_reference_exact(const tsta_params* p, char* why, size_t why_bytes);

/* Host-pointer form (copies in, runs, copies out, synchronises). */
int tsta_psa_score_batch(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                         int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                         const int64_t* off2, const int32_t* len2, int32_t* out);

/* Device-pointer form: every pointer is a device pointer on the engine's device; asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the engine's own stream).  max_len = upper bound on every
 * sequence length in the batch (sizes the per-wave boundary-row workspace). */
int tsta_psa_score_batch_dev(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* d_seqs,
                             const int64_t* d_off1, const int32_t* d_len1, const int64_t* d_off2,
                             const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream);

/* Single-launch forms and what happens when one gives up.  Several fills run as ONE launch whose waves hand rows to each
 * other through memory inside the launch (K1b band pipeline of small score batches, the trace fills, the POA dataflow fill).
 * Work is handed out as tickets in dependency order, so a wave only ever waits for one that is already running; the waits are
 * bounded all the same, and a wave whose wait does not end (a chip shared with other kernels) sets a status word and the
 * launch drains.  EVERY entry point then behaves the same way: the host-pointer forms (tsta_psa_score_batch[_2bit|_multi],
 * tsta_psa_align[_batch], tsta_poa_fill, tsta_poa_align) notice it and run the call again, in the same process, through the
 * launch-per-diagonal form of the same kernels (tsta_engine_stat "single_launch_reruns" counts these); the asynchronous
 * device-pointer forms cannot, so there every score of the call is poisoned — out[k] = INT32_MIN, a value no alignment has —
 * and tsta_psa_score_batch_dev_status() (synchronises `stream`) returns TSTA_ERR_INTERNAL.  No path returns success with a
 * wrong number.  Option "test_abandon" forces the event in tests. */
int tsta_psa_score_batch_dev_status(tsta_engine* eng, void* stream);
/* Counters of an engine: "single_launch_reruns"; "last_trace_form" = how the last tsta_psa_align[_batch] group was filled (0 one
 * launch per tile diagonal, 1 single launch with tiles through memory, 2 LDS-linked quads, 3 / 4 solver bands + emitter tiles
 * with the generic / pure-ACGT step, 5 coupled pairs in dataflow form); "last_sub_batches" = how many sub-batches the last
 * host-pointer score call was cut into (option "psa_overlap"); "psa_redo_pairs" = pairs of the last score call whose scores left
 * the packed kernel's 16-bit window and were scored by the int32 kernel instead (option "psa_f16max"; synchronises the device);
 * "last_walk_jumps" = bands the last single-pair
 * walk crossed by table look-up; "poa_last_spread" = largest distance between two predecessor rows of one node the last POA fill saw, "poa_last_ref_w" = the -W
 * at which the last POA fill ran in the reference's own arithmetic (0: the plain kernels served it).
 * TSTA_ERR_INVALID_ARG for an unknown key. */
int tsta_engine_stat(const tsta_engine* eng, const char* key, int64_t* value);

/* 2-bit packed input for pure-ACGT batches: a quarter of the host memory, PCIe traffic and input
 * HBM of the byte form.  Every sequence starts on a byte boundary of `packed`; 4 bases per byte, base i of a sequence in
 * bits 2(i%4)..2(i%4)+1 of its byte i/4; codes A 0, C 1, T 2, G 3 (= (ASCII >> 1) & 3).  boff1/boff2 = BYTE offset of
 * each sequence's first packed byte, len1/len2 = lengths in BASES.  The engine unpacks on the device (one pass) and
 * runs the same kernels; results are those of the byte form on the corresponding ASCII sequences.
 * tsta_pack_2bit packs n ASCII bases into (n+3)/4 bytes; TSTA_ERR_UNSUPPORTED if a byte is not one of A C G T. */
int tsta_pack_2bit(const uint8_t* src, int64_t n, uint8_t* dst);
int tsta_psa_score_batch_2bit(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* packed,
                              int64_t packed_bytes, const int64_t* boff1, const int32_t* len1,
                              const int64_t* boff2, const int32_t* len2, int32_t* out);
int tsta_psa_score_batch_dev_2bit(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* d_packed,
                                  const int64_t* d_boff1, const int32_t* d_len1, const int64_t* d_boff2,
                                  const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PSA with traceback  ==  TSTA_psa: fill with back/eback/fback (psa.c:294-327,358-362) + trace()
 * (psa.c:435-518) for one pair.  Returns the printed maxsorce and the two gapped rows exactly as the
 * reference writes them after ">1\n" and "\n>2\n" (psa.c:512-516): row_a belongs to the LONGER input.
 * row_a/row_b are malloc'd, NUL-terminated, length *out_len; free with tsta_free().
 * The packed trace matrix (1 byte per DP cell) lives in HBM only.
 * ------------------------------------------------------------------------------------------------ */
int tsta_psa_align(tsta_engine* eng, const tsta_params* p, const uint8_t* s1, int64_t l1,
                   const uint8_t* s2, int64_t l2, int32_t* maxscore, char** row_a, char** row_b,
                   int64_t* out_len);

/* Many independent pairs with traceback: what a loop of TSTA_psa processes would produce (the reference has no batch
 * mode: one process per pair, psa.c:539-668), with the pairs' fills in flight together on the device.  Inputs as for
 * tsta_psa_score_batch (host pointers).  out[k].row_a / row_b are malloc'd (free them with tsta_alignments_free or
 * each with tsta_free); row_a belongs to the longer input of pair k.  Pairs are processed in groups whose trace
 * matrices (1 byte per DP cell) fit the device together; a single pair too large for it returns TSTA_ERR_OOM. */
typedef struct tsta_alignment {
    int32_t maxscore; int32_t reserved;
    int64_t len;                    /* length of both gapped rows */
    char*   row_a;
    char*   row_b;
} tsta_alignment;

int  tsta_psa_align_batch(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                          int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                          const int64_t* off2, const int32_t* len2, tsta_alignment* out);
void tsta_alignments_free(tsta_alignment* out, int64_t n_pairs);

/* ------------------------------------------------------------------------------------------------
 * POA fill  ==  one control() wavefront of TSTA_msa without node_fuse (c-t-simd.c:979-1045 driving
 * block_alignment :602-657 / block_line_alignment :165-599): aligns `seq` (length n, 'N'-free) to a DAG
 * given in topological order (p->sort[], topo.c:243-260) as CSR predecessor lists (order of poa::pre[]).
 * Outputs (host pointers, any may be NULL):
 *   source/esource/fsource : int8 [n_nodes][n] row-major, the reference's trace codes
 *                            (c-t-simd.c:485-506, 530-538, 521-522/564-565) in plain (un-striped) layout
 *   lastscore              : int32 [n_nodes], H(r, n)  (the reference keeps it for sinks: c-t-simd.c:577-578)
 *   best_sink              : sink with the highest lastscore, ties -> lowest index (c-t-simd.c:737-750)
 * Domain: every node has < 42 predecessors (poa.h:109 / trace code space).
 *
 * What is computed is the plain int32 partial-order DP that the reference's saturating int8 machinery encodes (SURVEY
 * appendix A.2) — except in the situations where the reference's own output is NOT that DP but a property of its encoding
 * (and therefore of -W / the SIMD width of the build):
 *   (1) `seq` contains a literal 'N': the reference forces H(r,c) := H(r,c-1) in such columns (trace codes by equality with that
 *       value, c-t-simd.c:461,481,485-506) in its second pass only, its speculative first pass (c-t-simd.c:357-404) does not
 *       (measured: 17 of 40 N-containing inputs gave 2-5 different outputs over -W 3/5/10/16/30);
 *   (2) the sequence spans more L-blocks than the graph, ceil(n/L) > ceil(n_nodes/L): the first-column branch at
 *       c-t-simd.c:194 lacks the guard of :263 (L = option "poa_ref_block");
 *   (3) two predecessor rows of one node lie further apart than the reference's int8 offsets hold exactly: an offset enters a block
 *       clipped at 100 (c-t-simd.c:203-222), and although such a predecessor is cut off from the maximum (v > 60), its clipped
 *       offset still feeds the F / E flags of the cell; measured on the device (largest distance between two predecessor rows of a
 *       node), limit 100 minus what one row can add (max(|M|,|X|) + |O| + |E|) — a 90-node jump edge in a 7 x 300 bp family at
 *       30 % divergence (spread 179) changes 19 fsource bytes at the default scoring: tests/golden/msa_spread_golden.json;
 *   (4) the last block is padded and W max(|E|, M) + |O| >= 60: one byte lane's W padding columns ('N', readseq) shift a row beyond
 *       the trackers' cutoff between the two passes (measured: -W 32 at the default scoring; the bound is empirical).
 *   (5) the SCORING lies outside M >= 1, X < M, E <= -1, O <= 0, W max(|E|, M) + |O| + |X| < 64 (W = "poa_ref_block" / 64): W columns
 *       of matches / gap extensions, a gap open and a mismatch no longer stay inside the reference's saturating int8 trackers and
 *       their cutoff at 60; an empirical rule (2 168 scanned points, smallest left-hand side with a difference: 71;
 *       tests/golden/msa_domain_golden.json keeps 178 of them with the compiled reference's output) — tsta_poa_reference_exact.
 * There the fill runs — (1), (2), (4), (5) at once, (3) as a second run after the plain fill has seen the distance — in the reference's
 * own arithmetic and in the blocking of its AVX512 build at -W = "poa_ref_block" / 64 (option "poa_ref_arith"), and the result is
 * what that build prints, byte for byte (all 200 outputs of tests/golden/msa_n_golden.json, all of msa_w_golden.json): status
 * TSTA_OK.  Only where that path cannot run ("poa_ref_arith" = 0, "poa_ref_block" not 64 x 1..32, in-degree x W beyond the LDS, its
 * 2 B per cell of rows beyond the free device memory, a scoring parameter outside int8)
 * the call returns the plain DP ('N' rule applied consistently) with TSTA_WARN_REFERENCE_UNDEFINED and says which case it was.
 * ------------------------------------------------------------------------------------------------ */
typedef struct tsta_dag_csr {
    int32_t        n_nodes;
    const int32_t* pred_off;    /* n_nodes + 1 */
    const int32_t* pred;        /* pred_off[n_nodes] entries, each < its node's index */
    const uint8_t* base;        /* n_nodes */
    const uint8_t* is_sink;     /* n_nodes (out == 0) */
} tsta_dag_csr;

int tsta_poa_fill(tsta_engine* eng, const tsta_params* p, const tsta_dag_csr* dag, const uint8_t* seq,
                  int32_t n, int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore,
                  int32_t* best_sink);

/* 1: for this scoring the reference's POA fill at block length ref_block (= 64 x its -W; option "poa_ref_block") is the plain DP on
 * every input outside cases (1)-(4) above, i.e. the fast kernels serve the call; 0: case (5), the call runs in the reference's
 * arithmetic; < 0: tsta_status.  why (may be NULL) receives the reason ("" for 1).  Pure host arithmetic. */
int tsta_poa_reference_exact(const tsta_params* p, int32_t ref_block, char* why, size_t why_bytes);

/* ------------------------------------------------------------------------------------------------
 * POA fill + traceback walk  ==  control() up to and including the WALK of node_fuse (c-t-simd.c:737-957), i.e.
 * everything of node_fuse that reads the trace arrays; the graph surgery itself stays with the caller.  The
 * 3 x n_nodes x n trace bytes never leave HBM.  One step per visited cell, in visiting order (from the best sink
 * and the last character backwards):
 *   node  sort index of the graph node,  col  0-based sequence position,
 *   code  the EFFECTIVE source code of the cell (after the forced-continuation overwrites, :767-769, :776-782):
 *         code/42 = 3 F (insert the character), 0 E (skip the node), 1 match, 2 mismatch; code%42 = predecessor index
 *   peek  for match/mismatch cells: 1 iff the diagonal predecessor cell is itself a match (:790, :827, :873)
 * end_col = sequence position at which the walk stopped (-1: sequence consumed; >= 0: that prefix becomes new nodes,
 * :959-972).  steps is malloc'd: free with tsta_free().  lastscore may be NULL.
 * ------------------------------------------------------------------------------------------------ */
typedef struct tsta_poa_step { int32_t node, col; int8_t code, peek; int16_t pad; } tsta_poa_step;

int tsta_poa_align(tsta_engine* eng, const tsta_params* p, const tsta_dag_csr* dag, const uint8_t* seq,
                   int32_t n, tsta_poa_step** steps, int64_t* n_steps, int32_t* end_col, int32_t* lastscore);

/* ------------------------------------------------------------------------------------------------
 * Several GPUs of one node  ==  the reference's ONE driver loop over the blocks of ONE pair (psa.c:619-639) turned
 * into the sharding of a BATCH of independent pairs (north_star / SURVEY 8e): contiguous ranges of pairs per device,
 * balanced by DP cells; one host thread + one engine per device; no data-path collective.  The only exchange is the
 * final gather of the int32 scores (4 B per pair):
 *   "gather" = 0 (default)  every device copies its shard straight into out[lo..hi) (single process: the shortest way)
 *   "gather" = 1            RCCL all-gather over xGMI (ncclCommInitAll + one grouped ncclAllGather of int32, librccl is
 *                           dlopen'ed on first use): every device ends up with all n_pairs scores, device 0's copy goes to out
 *   "gather" = 2            as 1, and every device's gathered copy is read back and compared (tests)
 * devices = NULL -> devices 0..n_dev-1; n_dev = 0 -> every device of the node.  A tsta_multi is used by one host thread at a time.
 * Per-engine tuning knobs: tsta_engine_set_option(tsta_multi_engine(m, i), ...).
 * ------------------------------------------------------------------------------------------------ */
typedef struct tsta_multi tsta_multi;
int  tsta_multi_create(const int* devices, int n_dev, tsta_multi** out);
void tsta_multi_destroy(tsta_multi* m);
int  tsta_multi_device_count(const tsta_multi* m);
tsta_engine* tsta_multi_engine(tsta_multi* m, int i);      /* engine of the i-th device (NULL if out of range) */
const char*  tsta_multi_last_error(const tsta_multi* m);
int  tsta_multi_set_option(tsta_multi* m, const char* key, int64_t value);   /* "gather": 0 | 1 | 2 */

/* bounds[0..n_shards]: shard s = pairs [bounds[s], bounds[s+1]); contiguous, cut where the running sum of
 * len1[k]*len2[k] (+1 per pair, so that empty pairs spread too) crosses s/n_shards of the total.  Equal-size pairs
 * give shards that differ by at most one pair.  Pure host arithmetic (no device needed). */
int  tsta_shard_bounds(int64_t n_pairs, const int32_t* len1, const int32_t* len2, int n_shards, int64_t* bounds);

/* tsta_psa_score_batch / tsta_psa_align_batch over all devices of `m` (same arguments, same results).  Each device
 * receives only the byte range of `seqs` its shard refers to. */
int  tsta_psa_score_batch_multi(tsta_multi* m, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                                int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                                const int64_t* off2, const int32_t* len2, int32_t* out);
int  tsta_psa_align_batch_multi(tsta_multi* m, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                                int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                                const int64_t* off2, const int32_t* len2, tsta_alignment* out);

/* ------------------------------------------------------------------------------------------------
 * Identification.  tsta_build_info(): "libtsta_hip <16 hex digits> gfx950" where the digits are the first 64 bits
 * of the SHA-256 over the library's sources (the .hip and .h files of tsta_amd/csrc in name order, then this header) as they
 * were when THIS binary was compiled; tsta_amd/lib/BUILD_INFO holds the same line.  A stale shipped .so shows as a
 * mismatch with the sources next to it (tests/test_abi.py).  tsta_device_uuid(): 32 hex digits + NUL into out[33].
 * ------------------------------------------------------------------------------------------------ */
const char* tsta_build_info(void);
int  tsta_device_uuid(int device, char* out33);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): duration of the kernels of the last *_dev / host call, measured with HIP
 * events on the stream they ran on, and the number of kernel launches it took.
 * ------------------------------------------------------------------------------------------------ */
int tsta_last_kernel_ms(tsta_engine* eng, float* ms, int32_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* TSTA_ENGINE_H */
