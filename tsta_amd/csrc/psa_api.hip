// psa_api.hip — C-ABI entry points for the PSA kernels (score-only batch).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "kernels.h"
#include "trace_types.h"

using tsta::Engine;

namespace tsta {

// Rows per lane (band height = 64*R).  Larger R amortises the per-step lane hand-off over more cells but
// leaves lanes idle when the row sequence is shorter than a band.
static int psa_rows_per_lane(int32_t max_len, int forced)
{
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    return max_len >= 4096 ? 16 : (max_len >= 1024 ? 8 : 4);
}

// Packed-int16 kernel eligibility (psa_score_packed.hip): returns rows per lane (16 or 8), or 0 when the scoring
// does not fit: the kernel needs E,O <= 0 <= M, 0 <= M-X <= 255 and every value in flight within the 16-bit
// window around the per-pair base: (64R+256)*U + |O+E| + |O| + 200*M < 28000, U = max(M,X,0)+|O|+|E| bounding the
// score difference of adjacent cells.
static int psa_packed_rows_per_lane(const tsta_params& p, int64_t couples, int mode, int32_t max_len)   // mode: -1 auto, 0 never, 8/16/32 forced
{
    if (mode == 0 || couples < 1) return 0;
    if (p.E > 0 || p.O > 0 || p.M < 0 || p.M < p.X || p.M - p.X > 255) return 0;
    if (p.E < -4096 || p.O < -4096 || p.M > 4096 || p.X < -4096) return 0;
    auto fits = [&](int R) { return psa_packed_window(p.M, p.X, p.E, p.O, R) < 28000; };
    if (mode == 32) return fits(32) ? 32 : (fits(16) ? 16 : (fits(8) ? 8 : 0));
    if (mode == 16) return fits(16) ? 16 : (fits(8) ? 8 : 0);
    if (mode == 8) return fits(8) ? 8 : 0;
    // auto: the band height 64R that costs least for sequences of this length: padded rows x instructions per cell
    // (5.5 per cell + ~19 per step of 2R cells for the lane hand-off); taller bands also halve the boundary-row traffic
    int best = 0; double best_cost = 0;
    for (int R : {32, 16, 8}) {
        if (!fits(R)) continue;
        const double band = 64.0 * R;
        const double cost = std::ceil(std::max(1, (int)max_len) / band) * band * (5.5 + 9.5 / R);
        if (!best || cost < best_cost) { best = R; best_cost = cost; }
    }
    return best;
}

// Should the packed kernel run its F3 instantiation (kernels.h: PsaPackedArgs::f3)?  Exactness never depends on the answer — the
// kernel checks every block's range itself and hands a couple that leaves the 16-bit window to the int32 kernel —, only speed does:
// where the scores of a band plausibly spread beyond what the binary16-pattern range (30 719 wide) holds, blocks would fall back to
// the two-instruction maximum or couples to the int32 kernel.  Estimate: (64 R rows + 256 columns) at the long-range rate M + |E| per
// step, twice the gap open, plus the block margins the kernel uses.  option "psa_f16max": 0 never, 1 auto, 2 always (whatever the
// estimate says), 3 / 4 (tests) always, and without the fast step / with every couple handed to the int32 kernel.
static int psa_f16max_mode(const tsta_params& p, int R, int opt)
{
    if (opt == 0) return 0;
    if (opt >= 2) return opt - 1;
    const int64_t ae = -(int64_t)p.E, ao = -(int64_t)p.O, M = p.M, mx = (int64_t)p.M - p.X;
    const int64_t spread = (64LL * R + 256) * (M + ae) + 2 * ao;
    // the kernel's per-block margins (psa_score_packed.hip): 64 steps of drift, the next re-anchoring, a change of frame
    const int64_t between = 64 * (ao + ae + M) + 64 * (ae + M) + 32 * (ae * R + ae + M) + ae * R + 64 * (ae + M);
    const int64_t margin = std::max(66 * (mx + ao + 2 * ae + M), 66 * (2 * ae + M) + 130 * M) + between;
    return spread + margin < 15000 ? 1 : 0;
}

// K1b (band pipeline over the couples of a batch, psa_trace_packed.hip: psa_score_bands_kernel) pays when K1p could
// not give every SIMD two waves: fewer couples than 2x the SIMDs, sequences of at least two bands.  Returns rows per
// lane (16 or 8) or 0; `rows_bytes` = the boundary-row workspace it needs (every band of every pair keeps its own row).
static int psa_bands_rows_per_lane(const tsta_params& p, int64_t couples, int32_t max_len, int n_cu, int mode, size_t* rows_bytes)
{
    if (mode == 0 || couples < 1) return 0;
    if (p.E > 0 || p.O > 0 || p.M < 0 || p.M < p.X || p.M - p.X > 255) return 0;
    if (p.E < -4096 || p.O < -4096 || p.M > 4096 || p.X < -4096) return 0;
    const int64_t n_simd = (int64_t)n_cu * 4;
    if (mode < 0 && couples >= 2 * n_simd) return 0;
    const int64_t row_stride = ((int64_t)max_len + 63) / 64 * 64 + 64;
    auto bytes = [&](int R) { return (size_t)couples * 2 * (size_t)(((int64_t)max_len + 64 * R - 1) / (64 * R) + 1) * (size_t)row_stride * sizeof(int2); };
    auto ok = [&](int R) { return psa_packed_window(p.M, p.X, p.E, p.O, R) < 28000 && bytes(R) <= ((size_t)48 << 30); };
    int R = 0;
    if (mode == 16 || mode == 8) R = ok(mode) ? mode : 0;
    else if (max_len >= 2 * 1024 && ok(16) && couples * (((int64_t)max_len + 1023) / 1024) >= 2 * n_simd) R = 16;   // still >= 2 waves per SIMD
    else if (max_len >= 2 * 512 && ok(8)) R = 8;
    else if (max_len >= 2 * 1024 && ok(16)) R = 16;
    if (R && rows_bytes) *rows_bytes = bytes(R);
    return R;
}

// flags[k] = 1 iff both sequences of pair k consist of the bytes 'A', 'C', 'G', 'T' only: such pairs take the
// byte-permute form of the mismatch term in the packed kernel (psa_score_packed.hip: mismatch_term).  One block per
// pair (grid-stride); reads every sequence byte once, ~0.1 % of the fill's time.
__global__ __launch_bounds__(256) void psa_alphabet_flags_kernel(const uint8_t* __restrict__ seqs, const int64_t* off1, const int32_t* len1,
                                                                 const int64_t* off2, const int32_t* len2, int64_t n_pairs, uint8_t* flags)
{
    __shared__ int bad;
    for (int64_t k = blockIdx.x; k < n_pairs; k += gridDim.x) {
        if (threadIdx.x == 0) bad = 0;
        __syncthreads();
        int mine = 0;
        for (int which = 0; which < 2; ++which) {
            const uint8_t* p = seqs + (which ? off2[k] : off1[k]);
            const int64_t n = which ? len2[k] : len1[k];
            // head up to 4-byte alignment, aligned words, tail
            const int64_t head = min<int64_t>(n, (int64_t)((4 - ((uintptr_t)p & 3)) & 3));
            for (int64_t i = threadIdx.x; i < head; i += blockDim.x) { const uint8_t c = p[i]; mine |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T'); }
            const uint32_t* w = reinterpret_cast<const uint32_t*>(p + head);
            const int64_t nw = (n - head) / 4;
            for (int64_t i = threadIdx.x; i < nw; i += blockDim.x) {
                const uint32_t x = w[i];
                // per byte: x == 'A' (0x41) | 'C' (0x43) | 'G' (0x47) | 'T' (0x54)
                uint32_t ok = 0;
                for (int b = 0; b < 4; ++b) { const uint32_t c = (x >> (8 * b)) & 0xffu; ok += (c == 'A' || c == 'C' || c == 'G' || c == 'T'); }
                mine |= ok != 4;
            }
            for (int64_t i = head + nw * 4 + threadIdx.x; i < n; i += blockDim.x) { const uint8_t c = p[i]; mine |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T'); }
        }
        if (mine) bad = 1;
        __syncthreads();
        if (threadIdx.x == 0) flags[k] = bad ? 0 : 1;
        __syncthreads();
    }
}

// 2-bit packed input (tsta_psa_score_batch*_2bit): sequence q (q = 2k: -1 sequence of pair k, 2k+1: its -2 sequence)
// starts at byte boff[q] of `packed`, 4 bases per byte, base i in bits 2(i%4)..2(i%4)+1 of byte i/4, codes
// A 0, C 1, T 2, G 3 (= (ASCII >> 1) & 3).  Unpacked to ASCII at stride-sized slots of `out`; also writes the byte
// offsets the batch kernels index with.  One block per sequence (grid-stride).
__global__ __launch_bounds__(256) void psa_unpack_2bit_kernel(const uint8_t* __restrict__ packed, const int64_t* boff1, const int32_t* len1,
                                                              const int64_t* boff2, const int32_t* len2, int64_t n_pairs, int64_t stride,
                                                              uint8_t* __restrict__ out, int64_t* off1, int64_t* off2)
{
    for (int64_t q = blockIdx.x; q < 2 * n_pairs; q += gridDim.x) {
        const int64_t k = q >> 1;
        const uint8_t* src = packed + ((q & 1) ? boff2[k] : boff1[k]);
        const int n = (q & 1) ? len2[k] : len1[k];
        uint8_t* dst = out + q * stride;
        if (threadIdx.x == 0) { if (q & 1) off2[k] = q * stride; else off1[k] = q * stride; }
        // one packed byte -> one aligned 32-bit word of ASCII (stride is a multiple of 16); the tail word may run past
        // n inside the slot, which is padding
        uint32_t* dw = reinterpret_cast<uint32_t*>(dst);
        for (int i = threadIdx.x; i < (n + 3) / 4; i += blockDim.x) {
            const uint32_t b = src[i];
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= (uint32_t)((0x47544341u >> (8 * ((b >> (2 * j)) & 3u))) & 0xffu) << (8 * j);   // "ACTG"
            dw[i] = w;
        }
    }
}

// geometry key of a pair: (columns n = longer sequence) << 32 | (rows m = shorter)
__global__ void psa_geometry_keys_kernel(const int32_t* len1, const int32_t* len2, int64_t n_pairs, uint64_t* keys, uint32_t* idx)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const uint32_t a = (uint32_t)len1[k], b = (uint32_t)len2[k];
    keys[k] = ((uint64_t)(a > b ? a : b) << 32) | (a > b ? b : a);
    idx[k] = (uint32_t)k;
}

// Orders the pairs of a device-resident batch by geometry (radix sort of 64-bit keys, rocPRIM via hipCUB) so that
// pairs of equal / similar size become neighbours and can share a wave in the packed kernel.  Returns the device
// permutation (position -> pair index) in e->scratch, or nullptr on failure (callers then use the identity order).
static const uint32_t* psa_order_by_geometry(Engine* e, const int32_t* d_len1, const int32_t* d_len2, int64_t n_pairs, hipStream_t s)
{
    const size_t n = (size_t)n_pairs;
    size_t temp_bytes = 0;
    uint64_t* kdummy = nullptr; uint32_t* vdummy = nullptr;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, kdummy, kdummy, vdummy, vdummy, (int)n_pairs, 0, 64, s) != hipSuccess) return nullptr;
    const size_t kb = (n * 8 + 255) / 256 * 256, vb = (n * 4 + 255) / 256 * 256;
    if (e->reserve(e->scratch, 2 * kb + 2 * vb + temp_bytes + 256, "hipMalloc(geometry order)") != TSTA_OK) return nullptr;
    char* base = reinterpret_cast<char*>(e->scratch.p);
    uint64_t* k_in = reinterpret_cast<uint64_t*>(base);
    uint64_t* k_out = reinterpret_cast<uint64_t*>(base + kb);
    uint32_t* v_in = reinterpret_cast<uint32_t*>(base + 2 * kb);
    uint32_t* v_out = reinterpret_cast<uint32_t*>(base + 2 * kb + vb);
    void* temp = base + 2 * kb + 2 * vb;
    hipLaunchKernelGGL(psa_geometry_keys_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, d_len1, d_len2, n_pairs, k_in, v_in);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (int)n_pairs, 0, 64, s) != hipSuccess) return nullptr;
    return v_out;
}

}  // namespace tsta

namespace tsta {

int psa_score_batch_dev_impl(Engine* e, const tsta_params* p, int64_t n_pairs, const uint8_t* d_seqs,
                             const int64_t* d_off1, const int32_t* d_len1, const int64_t* d_off2,
                             const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream)
{
    if (!e || !p || n_pairs < 0 || max_len < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!d_seqs || !d_off1 || !d_len1 || !d_off2 || !d_len2 || !d_out) return TSTA_ERR_INVALID_ARG;
    if (n_pairs > 0x7fffffffLL) return TSTA_ERR_UNSUPPORTED;
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : e->stream;
    e->psa_status = nullptr;

    // ---- packed-int16 kernel (K1p) for couples of equal-geometry pairs, when the scoring fits 16-bit lanes
    // one wave nearly saturates a SIMD, so packing two pairs per wave only pays once there are more
    // pairs than SIMDs; below that the int32 kernel keeps every SIMD busy with one pair each
    // ---- K1b first: a batch too small to fill the chip with one wave per couple runs the bands of every couple
    // concurrently (band pipeline, one launch per tile diagonal)
    size_t bands_rows = 0;
    const int Rb = (e->opt_psa_packed >= 0 && e->opt_psa_bands <= 0) ? 0 : tsta::psa_bands_rows_per_lane(*p, n_pairs / 2, max_len, e->n_cu, e->opt_psa_bands, &bands_rows);
    const bool enough = (n_pairs > (int64_t)e->n_cu * 4 || e->opt_psa_packed > 0) && !Rb;
    const int Rp = enough ? tsta::psa_packed_rows_per_lane(*p, n_pairs / 2, e->opt_psa_packed, max_len) : 0;
    const int64_t couples = (Rp || Rb) ? n_pairs / 2 : 0;
    int64_t pblocks = 0;
    if (couples && Rp) {
        pblocks = std::min<int64_t>((couples + 3) / 4, (int64_t)e->n_cu * tsta::psa_packed_blocks_per_cu(Rp));
        if (pblocks < 1) pblocks = 1;
    }
    // ---- int32 kernel (K1) for everything else
    const int R = tsta::psa_rows_per_lane(max_len, e->opt_psa_rows);
    const int per_cu = tsta::psa_score_blocks_per_cu(R);
    int64_t blocks = std::min<int64_t>((n_pairs + 3) / 4, (int64_t)e->n_cu * per_cu);
    if (blocks < 1) blocks = 1;
    const int64_t row_stride = ((int64_t)max_len + 63) / 64 * 64 + 64;
    const size_t need32 = (size_t)blocks * 4 * (size_t)row_stride * sizeof(int2);
    const size_t need16 = Rb ? bands_rows : (size_t)pblocks * 4 * 2 * (size_t)row_stride * sizeof(int2);
    if ((rc = e->reserve(e->rows, need32 + need16, "hipMalloc(boundary rows)")) != TSTA_OK) return rc;

    tsta::PsaBatchArgs a;
    a.seqs = d_seqs; a.off1 = d_off1; a.len1 = d_len1; a.off2 = d_off2; a.len2 = d_len2; a.out = d_out;
    a.rows = reinterpret_cast<int2*>(e->rows.p); a.row_stride = row_stride;
    a.counter = reinterpret_cast<unsigned int*>(e->counter.p);
    a.n_pairs = n_pairs; a.M = p->M; a.X = p->X; a.E = p->E; a.O = p->O;
    a.packed_couples = couples;
    a.order = couples ? tsta::psa_order_by_geometry(e, d_len1, d_len2, n_pairs, s) : nullptr;
    a.redo = nullptr;
    static const bool want_clock = getenv("TSTA_PSA_CLOCK") != nullptr;      // diagnostics: in-kernel shader clock
    a.clock_out = want_clock ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(e->counter.p) + 64) : nullptr;

    if ((rc = e->check(hipMemsetAsync(a.counter, 0, 2 * sizeof(unsigned int), s), "hipMemsetAsync")) != TSTA_OK) return rc;
    const uint8_t* acgt = nullptr;
    if (couples && (rc = e->reserve(e->pair_flags, 2 * (size_t)n_pairs, "hipMalloc(pair flags)")) != TSTA_OK) return rc;   // [alphabet flags][redo flags]
    if (couples && e->opt_psa_perm != 0 && p->M - p->X <= 255)
        acgt = reinterpret_cast<const uint8_t*>(e->pair_flags.p);
    // F3: scores around 0x4000 so that a steady block's three-way maximum can be one v_pk_maximum3_f16 (psa_score_packed.hip); a couple
    // whose scores leave the 16-bit window on the way is handed to the int32 kernel through these flags
    const int f3 = couples ? tsta::psa_f16max_mode(*p, Rp ? Rp : Rb, e->opt_psa_f16max) : 0;
    e->last_redo = nullptr; e->last_redo_n = 0;
    if (f3) {
        a.redo = reinterpret_cast<uint8_t*>(e->pair_flags.p) + n_pairs;
        e->last_redo = a.redo; e->last_redo_n = n_pairs;
        if ((rc = e->check(hipMemsetAsync(a.redo, 0, (size_t)n_pairs, s), "hipMemsetAsync(redo flags)")) != TSTA_OK) return rc;
    }
    e->timing_begin(s);
    int launches = 0;
    if (couples && Rb) {
        tsta::PsaBandsArgs ba;
        ba.b = a;
        ba.b.rows = a.rows + need32 / sizeof(int2);
        ba.b.clock_out = nullptr;
        ba.W_max = (int)(((int64_t)max_len + 64 * Rb - 1) / (64 * Rb));
        ba.rows_per_pair = (int64_t)(ba.W_max + 1) * row_stride;
        const size_t state_bytes = (size_t)couples * ba.W_max * tsta::packed_state_ints(Rb) * 64 * sizeof(int);
        if ((rc = e->reserve(e->scratch2, state_bytes, "hipMalloc(band state)")) != TSTA_OK) return rc;
        ba.state = reinterpret_cast<int*>(e->scratch2.p);
        const uint32_t ae = (uint32_t)(-p->E), ao = (uint32_t)(-p->O), am = (uint32_t)p->M, amx = (uint32_t)(p->M - p->X);
        ba.pe = ae | (ae << 16); ba.po = ao | (ao << 16); ba.pm = am | (am << 16); ba.pmx = amx | (amx << 16);
        ba.acgt = acgt;
        ba.f3 = f3;
        ba.ffree = acgt && e->opt_psa_ffree != 0 && tsta::psa_ffree_ok(p->M, p->X, p->E, p->O, Rb);
        ba.elide = e->opt_psa_elide != 0 && tsta::psa_packed_window(p->M, p->X, p->E, p->O, Rb) + (int64_t)ae * (32 * (Rb + 1) + 192) < 28000;
        if (acgt) {
            hipLaunchKernelGGL(tsta::psa_alphabet_flags_kernel, dim3((unsigned)std::min<int64_t>(n_pairs, 65536)), dim3(256), 0, s,
                               d_seqs, d_off1, d_len1, d_off2, d_len2, n_pairs, reinterpret_cast<uint8_t*>(e->pair_flags.p));
            ++launches;
        }
        // tile length: 128 steps (round 4, with the cheaper F3 step the per-tile work — lane state in and out, waits, ticket — weighs more:
        // config #2 at 64 / 128 / 192 / 256 steps 4.10 / 3.84 / 3.87 / 3.89 ms, 128 x 50 kbp at 128 / 256 / 512 40.7 / 41.2 / 44.6 ms;
        // tests/tools/config2_tiles.py)
        ba.T = (e->opt_trace_tile >= 64 && e->opt_trace_tile % 64 == 0) ? e->opt_trace_tile : 128;
        const int64_t S_total = ((int64_t)max_len + 64 + 63) / 64 * 64;
        const int64_t D = (S_total + (int64_t)(ba.W_max - 1) * (ba.T + 64) + ba.T - 1) / ba.T;
        const int64_t bblocks = (couples * ba.W_max + 3) / 4;
        if (bblocks > 0x7fffffffLL) return TSTA_ERR_UNSUPPORTED;
        ba.ticket = nullptr; ba.progress = nullptr; ba.status = nullptr; ba.tile_first = ba.tile_wlo = ba.tile_nw = nullptr; ba.n_d = 0; ba.n_tiles = 0;
        ba.test_abandon = e->opt_test_abandon;
        bool dataflow = e->opt_trace_persistent != 0 && !e->no_single_launch && couples * ba.W_max <= (1 << 22) && D <= (1 << 20);
        std::vector<int>& tab = e->host_tab;
        if (dataflow) {
            if (e->ev_tab) (void)hipEventSynchronize(e->ev_tab);          // the previous call's upload of this vector has finished
            else if (hipEventCreateWithFlags(&e->ev_tab, hipEventDisableTiming) != hipSuccess) { e->ev_tab = nullptr; dataflow = false; }
        }
        if (dataflow) {
            // ONE launch: the tiles of the D launches as tickets, d-major (psa_trace_packed.hip, psa_score_bands_kernel<.., true>)
            tab.assign((size_t)(3 * D + 1), 0);
            int64_t total = 0;
            for (int64_t d = 0; d < D; ++d) {
                // bands with work in launch d: S_w(d) > 0 and S_w(d-1) < S_total  (as psa_trace.hip: active())
                int64_t w_hi = ((d + 1) * ba.T + (ba.T + 64) - 1) / (ba.T + 64);
                const int64_t w_lo = (d * ba.T - S_total < 0) ? 0 : (d * ba.T - S_total) / (ba.T + 64);
                w_hi = std::min<int64_t>(ba.W_max, w_hi);
                const int64_t nw = std::max<int64_t>(0, w_hi - w_lo);
                tab[(size_t)d] = (int)total; tab[(size_t)(D + 1 + d)] = (int)w_lo; tab[(size_t)(2 * D + 1 + d)] = (int)std::max<int64_t>(nw, 1);
                total += couples * nw;
                if (total > 0x7fffffff) { dataflow = false; break; }
            }
            tab[(size_t)D] = (int)total;
            const size_t sync_ints = 64 + (size_t)couples * ba.W_max, tab_ints = tab.size();
            if (dataflow && e->reserve(e->unpacked2, (sync_ints + tab_ints) * sizeof(int) + 256, "hipMalloc(band tickets)") == TSTA_OK) {
                int* d_sync = reinterpret_cast<int*>(e->unpacked2.p);
                if ((rc = e->check(hipMemsetAsync(d_sync, 0, sync_ints * sizeof(int), s), "hipMemsetAsync(band tickets)")) ||
                    (rc = e->check(hipMemcpyAsync(d_sync + sync_ints, tab.data(), tab_ints * sizeof(int), hipMemcpyHostToDevice, s), "H2D tile table")))
                    return rc;
                (void)hipEventRecord(e->ev_tab, s);             // `tab` is the engine's: it is not touched again before this event has passed
                ba.ticket = reinterpret_cast<unsigned*>(d_sync); ba.status = reinterpret_cast<unsigned*>(d_sync + 1); ba.progress = d_sync + 64;
                ba.tile_first = d_sync + sync_ints; ba.tile_wlo = ba.tile_first + D + 1; ba.tile_nw = ba.tile_wlo + D;
                ba.n_d = (int)D; ba.n_tiles = (int)total;
            } else dataflow = false;
        }
        if (dataflow) {
            ba.d = 0;
            tsta::launch_psa_score_bands_persistent(Rb, ba, e->n_cu, s);
            ++launches;
            e->psa_status = ba.status;                          // read by the finish kernel (poison) and by the host forms (second run)
        } else
        for (int64_t d = 0; d < D; ++d) {
            ba.d = (int)d;
            tsta::launch_psa_score_bands(Rb, ba, (int)bblocks, s);
            ++launches;
        }
        tsta::launch_psa_score_bands_finish(Rb, ba, s);
        ++launches;
    } else if (couples) {
        tsta::PsaPackedArgs pa;
        pa.b = a;
        pa.b.rows = a.rows + need32 / sizeof(int2);
        pa.b.counter = a.counter + 1;
        pa.b.clock_out = nullptr;
        pa.wave_log = nullptr;
        static const bool want_log = want_clock && atoi(getenv("TSTA_PSA_CLOCK")) == 2;
        if (want_clock && e->reserve(e->scratch2, 256 + (want_log ? (size_t)couples * 32 : 0), "diag") == TSTA_OK) {
            pa.b.clock_out = reinterpret_cast<unsigned long long*>(e->scratch2.p);
            if (want_log) { pa.wave_log = pa.b.clock_out + 32; (void)hipMemsetAsync(pa.wave_log, 0, (size_t)couples * 32, s); }
        }
        const uint32_t ae = (uint32_t)(-p->E), ao = (uint32_t)(-p->O), am = (uint32_t)p->M, amx = (uint32_t)(p->M - p->X);
        pa.pe = ae | (ae << 16); pa.po = ao | (ao << 16); pa.pm = am | (am << 16); pa.pmx = amx | (amx << 16);
        // the frame of the elided blocks spreads the rows of a band by |E| each: needs that much more of the 16-bit window
        pa.elide = e->opt_psa_elide != 0 && tsta::psa_packed_window(p->M, p->X, p->E, p->O, Rp) + (int64_t)ae * (32 * (Rp + 1) + 192) < 28000;
        pa.acgt = acgt;
        pa.f3 = f3;
        pa.ffree = acgt && e->opt_psa_ffree != 0 && tsta::psa_ffree_ok(p->M, p->X, p->E, p->O, Rp);
        if (acgt) {
            hipLaunchKernelGGL(tsta::psa_alphabet_flags_kernel, dim3((unsigned)std::min<int64_t>(n_pairs, 65536)), dim3(256), 0, s,
                               d_seqs, d_off1, d_len1, d_off2, d_len2, n_pairs, reinterpret_cast<uint8_t*>(e->pair_flags.p));
            ++launches;
        }
        tsta::launch_psa_score_packed(Rp, pa, (int)pblocks, s);
        ++launches;
    }
    tsta::launch_psa_score_batch(R, a, (int)blocks, s);
    ++launches;
    e->n_launches = launches;
    e->timing_end(s);
    if (want_clock) {
        unsigned long long h[4];
        if (hipMemcpyAsync(h, a.clock_out, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess &&
            hipStreamSynchronize(s) == hipSuccess && h[3] > h[1])
            fprintf(stderr, "[tsta] psa_score wave0: %.1f ms, shader clock %.0f MHz (R=%d, %lld blocks)\n",
                    (double)(h[3] - h[1]) / 1e5, (double)(h[2] - h[0]) / (double)(h[3] - h[1]) * 100.0, R, (long long)blocks);
        if (couples && Rp && atoi(getenv("TSTA_PSA_CLOCK")) == 2 && e->scratch2.p) {
            // per couple: when and where it ran -> TSTA_PSA_CLOCK_FILE (text: ticket begin_us end_us xcc se cu simd wave slot)
            std::vector<unsigned long long> w((size_t)couples * 4);
            if (e->copy_sync(w.data(), reinterpret_cast<unsigned long long*>(e->scratch2.p) + 32, w.size() * 8, hipMemcpyDeviceToHost, "D2H wave log") == TSTA_OK) {
                const char* fn = getenv("TSTA_PSA_CLOCK_FILE");
                if (FILE* f = fopen(fn ? fn : "psa_wave_log.txt", "w")) {
                    unsigned long long t0 = ~0ull;
                    for (int64_t c = 0; c < couples; ++c) if (w[4 * c + 1]) t0 = std::min(t0, w[4 * c]);
                    for (int64_t c = 0; c < couples; ++c) {
                        if (!w[4 * c + 1]) continue;
                        const unsigned hw = (unsigned)w[4 * c + 2], xcc = (unsigned)(w[4 * c + 2] >> 32) & 15u;
                        fprintf(f, "%lld %.1f %.1f %u %u %u %u %u %llu\n", (long long)c, (double)(w[4 * c] - t0) / 100.0, (double)(w[4 * c + 1] - t0) / 100.0,
                                xcc, (hw >> 13) & 7u, (hw >> 8) & 15u, (hw >> 4) & 3u, hw & 15u, w[4 * c + 3]);
                    }
                    fclose(f);
                }
            }
        }
        if (couples && e->scratch2.p && e->copy_sync(h, e->scratch2.p, sizeof(h), hipMemcpyDeviceToHost, "D2H clock") == TSTA_OK && h[3] > h[1])
            fprintf(stderr, "[tsta] psa_score_packed wave0: %.1f ms, shader clock %.0f MHz (R=%d, %lld blocks, %d blocks/CU by API)\n",
                    (double)(h[3] - h[1]) / 1e5, (double)(h[2] - h[0]) / (double)(h[3] - h[1]) * 100.0, Rp, (long long)pblocks,
                    tsta::psa_packed_blocks_per_cu(Rp));
    }
    return e->check(hipGetLastError(), "psa_score_batch launch");
}

}  // namespace tsta

extern "C" {

// Every PSA entry point ends a successful call with psa_reference_warning(): TSTA_OK, or TSTA_WARN_REFERENCE_UNDEFINED when the
// scoring lies outside the domain in which the reference's int8 encoding computes this DP (tsta_psa_reference_exact).
int tsta_psa_score_batch_dev(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* d_seqs,
                             const int64_t* d_off1, const int32_t* d_len1, const int64_t* d_off2,
                             const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream)
{
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = tsta::psa_score_batch_dev_impl(e, p, n_pairs, d_seqs, d_off1, d_len1, d_off2, d_len2, d_out, max_len, stream);
    return rc == TSTA_OK ? tsta::psa_reference_warning(e, p) : rc;
}

int tsta_psa_score_batch(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                         int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                         const int64_t* off2, const int32_t* len2, int32_t* out)
{
    if (!out && n_pairs > 0) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = tsta::psa_score_batch_host(e, p, n_pairs, seqs, seqs_bytes, off1, len1, off2, len2, out, 0, nullptr);
    return rc == TSTA_OK ? tsta::psa_reference_warning(e, p) : rc;
}

}  // extern "C"

namespace tsta {

// After a *_dev call enqueued on the engine's stream by a host-pointer form: if its single-launch band pipeline (K1b)
// abandoned a wait — the device scores are then poison, psa_score_bands_finish_kernel — the same call is enqueued again with
// the single-launch forms off (launch-per-diagonal pipeline: the kernel boundary is the only synchronisation), as the trace
// and POA entry points do (psa_trace.hip: run_group, poa_fill.hip: poa_with_fallback).  Synchronises the stream.
template <class F> static int psa_score_rerun_if_abandoned(Engine* e, F again)
{
    if (!e->psa_status) return TSTA_OK;
    unsigned st = 0;
    int rc = e->copy_sync(&st, e->psa_status, sizeof(st), hipMemcpyDeviceToHost, "D2H band pipeline status");
    if (rc != TSTA_OK || st == 0u) return rc;
    ++e->single_launch_reruns;
    e->no_single_launch = true;
    rc = again();
    e->no_single_launch = false;
    return rc;
}

// the copy stream and the events of the overlapped upload (created on first use)
static int psa_overlap_streams(Engine* e)
{
    if (!e->side[0] && hipStreamCreateWithFlags(&e->side[0], hipStreamNonBlocking) != hipSuccess) { e->side[0] = nullptr; (void)hipGetLastError(); return TSTA_ERR_HIP; }
    if (!e->ev_fork && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) { e->ev_fork = nullptr; (void)hipGetLastError(); return TSTA_ERR_HIP; }
    for (hipEvent_t& ev : e->ev_up)
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; (void)hipGetLastError(); return TSTA_ERR_HIP; }
    return TSTA_OK;
}

// tsta_psa_score_batch proper.  out (host) may be NULL when d_scores is given: the scores then stay on the device, in a
// buffer of at least max(n_pairs, dev_count) int32 (tsta_psa_score_batch_multi gathers them with RCCL from there), and
// the stream is synchronised either way.
int psa_score_batch_host(Engine* e, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                         int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                         const int64_t* off2, const int32_t* len2, int32_t* out, int64_t dev_count, int32_t** d_scores)
{
    tsta_engine* eng = reinterpret_cast<tsta_engine*>(e);
    if (!eng || !p || n_pairs < 0 || seqs_bytes < 0) return TSTA_ERR_INVALID_ARG;
    if (d_scores) {
        *d_scores = nullptr;
        int rc0 = e->check(hipSetDevice(e->device), "hipSetDevice");
        if (rc0 != TSTA_OK) return rc0;
        if ((rc0 = e->reserve(e->out_scores, (size_t)std::max<int64_t>(std::max(n_pairs, dev_count), 1) * sizeof(int32_t), "hipMalloc(scores)")) != TSTA_OK) return rc0;
        *d_scores = reinterpret_cast<int32_t*>(e->out_scores.p);
    }
    if (n_pairs == 0) return TSTA_OK;
    if (!seqs || !off1 || !len1 || !off2 || !len2 || (!out && !d_scores)) return TSTA_ERR_INVALID_ARG;
    int32_t max_len = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
        if (len1[k] < 0 || len2[k] < 0 || off1[k] < 0 || off2[k] < 0 ||
            off1[k] + len1[k] > seqs_bytes || off2[k] + len2[k] > seqs_bytes) {
            e->set_error("pair %lld: sequence range outside seqs[0..%lld)", (long long)k, (long long)seqs_bytes);
            return TSTA_ERR_INVALID_ARG;
        }
        max_len = std::max(max_len, std::max(len1[k], len2[k]));
    }
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    const size_t nb = (size_t)n_pairs;
    // A handful of long pairs: the batch kernels give each pair one wave, which leaves the chip idle; run every such
    // pair through the pipelined single-pair fill (the bands of one pair spread over many waves) instead.
    if (n_pairs <= 64 && (n_pairs == 1 || !tsta::psa_bands_rows_per_lane(*p, 1, 4096, e->n_cu, e->opt_psa_bands, nullptr))) {
        bool all_big = true;
        for (int64_t k = 0; k < n_pairs; ++k) all_big = all_big && (int64_t)len1[k] * len2[k] >= 200000000LL;
        if (all_big) {
            std::vector<int32_t> tmp(out ? 0 : nb);
            int32_t* res = out ? out : tmp.data();
            for (int64_t k = 0; k < n_pairs; ++k) {
                const bool swap = len2[k] > len1[k];                                   // psa.c:194-199
                const uint8_t* a = seqs + (swap ? off2[k] : off1[k]); const int n = swap ? len2[k] : len1[k];
                const uint8_t* b = seqs + (swap ? off1[k] : off2[k]); const int m = swap ? len1[k] : len2[k];
                if ((rc = tsta::psa_score_single_pair(e, p, a, n, b, m, &res[k])) != TSTA_OK) return rc;
            }
            if (d_scores) return e->copy_sync(*d_scores, res, nb * sizeof(int32_t), hipMemcpyHostToDevice, "H2D scores");
            return TSTA_OK;
        }
    }
    e->last_sub_batches = 1;
    std::vector<int64_t> h_off(2 * nb);
    std::vector<int32_t> h_len(2 * nb);
    for (size_t k = 0; k < nb; ++k) { h_off[k] = off1[k]; h_off[nb + k] = off2[k]; h_len[k] = len1[k]; h_len[nb + k] = len2[k]; }
    const size_t meta_bytes = nb * (2 * sizeof(int64_t) + 2 * sizeof(int32_t));
    if ((rc = e->reserve(e->in_seqs, (size_t)seqs_bytes + 16, "hipMalloc(seqs)")) != TSTA_OK) return rc;
    if ((rc = e->reserve(e->in_meta, meta_bytes, "hipMalloc(meta)")) != TSTA_OK) return rc;
    if (!d_scores && (rc = e->reserve(e->out_scores, nb * sizeof(int32_t), "hipMalloc(scores)")) != TSTA_OK) return rc;
    char* m = reinterpret_cast<char*>(e->in_meta.p);
    int64_t* d_off1 = reinterpret_cast<int64_t*>(m);
    int64_t* d_off2 = d_off1 + nb;
    int32_t* d_len1 = reinterpret_cast<int32_t*>(d_off2 + nb);
    int32_t* d_len2 = d_len1 + nb;
    hipStream_t s = e->stream;

    // ---- upload / compute overlap (">= 2 streams per device"; what the reference's readseq, psa.c:179-217, does before
    // its fill starts).  A large batch is cut into sub-batches — contiguous ranges of pairs —: sub-batch j + 1 travels on the copy
    // stream while the kernels of sub-batch j run on the compute stream, so only the FIRST sub-batch's upload is exposed.
    // The size of a sub-batch is not free (measured, tests/tools/wave_log.py + e2e_timing.py, 50 kbp pairs): a SIMD serves its two
    // resident waves oldest-first — the older wave takes a couple in 716 ms (a lone wave: 637), the younger one in 2 189 ms while
    // the older does three — so a launch is balanced when every SIMD gets 4 couples (3 + 1; 8 192 pairs on 256 CUs: 2 189 ms =
    // 4 x 547), and a launch of 2 couples per SIMD takes 1 214 ms instead of 1 094: cutting the 8 192-pair shard in two costs
    // 10 %, twenty times the 11 ms of upload it would hide.  Hence: sub-batches of 4 couples per SIMD, and a batch below two of
    // them goes up in one piece (exposed upload of an 8 192 x 50 kbp shard: 0.5 %).
    {
        const int64_t round_pairs = (int64_t)e->n_cu * 4 * 4 * 2;                       // 256 CUs x 4 SIMDs x 4 couples x 2 pairs
        const int64_t n_sub = e->opt_psa_overlap ? std::min<int64_t>(n_pairs / round_pairs, 8) : 1;
        struct Sub { int64_t lo, hi, b0, b1; };
        std::vector<Sub> sub;
        int64_t span = 0;
        for (int64_t j = 0; j < n_sub && n_sub >= 2; ++j) {
            Sub u; u.lo = n_pairs * j / n_sub; u.hi = n_pairs * (j + 1) / n_sub; u.b0 = seqs_bytes; u.b1 = 0;
            for (int64_t k = u.lo; k < u.hi; ++k) {
                u.b0 = std::min(u.b0, std::min(off1[k], off2[k]));
                u.b1 = std::max(u.b1, std::max(off1[k] + len1[k], off2[k] + len2[k]));
            }
            if (u.b1 < u.b0) { u.b0 = 0; u.b1 = 0; }
            span += u.b1 - u.b0;
            sub.push_back(u);
        }
        // (sequences shared between pairs — TSTA_psa_batch -A — make the byte ranges overlap: then one upload is cheaper)
        if (!sub.empty() && span <= seqs_bytes + seqs_bytes / 4 && psa_overlap_streams(e) == TSTA_OK) {
            hipStream_t cs = e->side[0];
            if ((rc = e->check(hipMemcpyAsync(d_off1, h_off.data(), nb * 16, hipMemcpyHostToDevice, s), "H2D offsets")) ||
                (rc = e->check(hipMemcpyAsync(d_len1, h_len.data(), nb * 8, hipMemcpyHostToDevice, s), "H2D lengths")) ||
                (rc = e->check(hipEventRecord(e->ev_fork, s), "hipEventRecord(fork)")) ||
                (rc = e->check(hipStreamWaitEvent(cs, e->ev_fork, 0), "hipStreamWaitEvent(fork)")))      // (the copy stream must not run ahead of
                return rc;                                                                                 //  an earlier call still reading in_seqs)
            uint8_t* d_seqs = reinterpret_cast<uint8_t*>(e->in_seqs.p);
            e->last_sub_batches = (int)sub.size();
            e->timing_hold = false;
            for (size_t j = 0; j < sub.size() && rc == TSTA_OK; ++j) {
                const Sub& u = sub[j];
                // pageable memory: the runtime stages it chunk by chunk and the call returns when the last chunk is on its way,
                // i.e. the host thread is here while the kernels of sub-batch j - 1 run
                if (u.b1 > u.b0) rc = e->check(hipMemcpyAsync(d_seqs + u.b0, seqs + u.b0, (size_t)(u.b1 - u.b0), hipMemcpyHostToDevice, cs), "H2D seqs (sub-batch)");
                if (rc == TSTA_OK) rc = e->check(hipEventRecord(e->ev_up[j], cs), "hipEventRecord(upload)");
                if (rc == TSTA_OK) rc = e->check(hipStreamWaitEvent(s, e->ev_up[j], 0), "hipStreamWaitEvent(upload)");
                if (rc == TSTA_OK) rc = psa_score_batch_dev_impl(e, p, u.hi - u.lo, d_seqs, d_off1 + u.lo, d_len1 + u.lo, d_off2 + u.lo, d_len2 + u.lo,
                                                                 reinterpret_cast<int32_t*>(e->out_scores.p) + u.lo, max_len, s);
                e->timing_hold = true;                           // tsta_last_kernel_ms: from the first sub-batch's first kernel to the last one's end
            }
            e->timing_hold = false;
            if (rc != TSTA_OK) { (void)hipStreamSynchronize(cs); (void)hipStreamSynchronize(s); return rc; }
            // (sub-batches of whole rounds never take the small-batch band pipeline: no single launch to run again)
            if (out && (rc = e->check(hipMemcpyAsync(out, e->out_scores.p, nb * 4, hipMemcpyDeviceToHost, s), "D2H scores"))) return rc;
            return e->check(hipStreamSynchronize(s), "psa_score_batch sync");
        }
    }
    if ((rc = e->check(hipMemcpyAsync(e->in_seqs.p, seqs, (size_t)seqs_bytes, hipMemcpyHostToDevice, s), "H2D seqs")) ||
        (rc = e->check(hipMemcpyAsync(d_off1, h_off.data(), nb * 16, hipMemcpyHostToDevice, s), "H2D offsets")) ||
        (rc = e->check(hipMemcpyAsync(d_len1, h_len.data(), nb * 8, hipMemcpyHostToDevice, s), "H2D lengths")))
        return rc;
    rc = psa_score_batch_dev_impl(e, p, n_pairs, reinterpret_cast<const uint8_t*>(e->in_seqs.p), d_off1, d_len1,
                                  d_off2, d_len2, reinterpret_cast<int32_t*>(e->out_scores.p), max_len, s);
    if (rc != TSTA_OK) return rc;
    if ((rc = psa_score_rerun_if_abandoned(e, [&] {
            return psa_score_batch_dev_impl(e, p, n_pairs, reinterpret_cast<const uint8_t*>(e->in_seqs.p), d_off1, d_len1,
                                            d_off2, d_len2, reinterpret_cast<int32_t*>(e->out_scores.p), max_len, s); })) != TSTA_OK)
        return rc;
    if (out && (rc = e->check(hipMemcpyAsync(out, e->out_scores.p, nb * 4, hipMemcpyDeviceToHost, s), "D2H scores"))) return rc;
    return e->check(hipStreamSynchronize(s), "psa_score_batch sync");
}

}  // namespace tsta

extern "C" {


int tsta_pack_2bit(const uint8_t* src, int64_t n, uint8_t* dst)
{
    if (n < 0 || (n && (!src || !dst))) return TSTA_ERR_INVALID_ARG;
    for (int64_t i = 0; i < n; i += 4) {
        uint8_t b = 0;
        for (int j = 0; j < 4 && i + j < n; ++j) {
            const uint8_t c = src[i + j];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return TSTA_ERR_UNSUPPORTED;
            b |= (uint8_t)(((c >> 1) & 3u) << (2 * j));
        }
        dst[i / 4] = b;
    }
    return TSTA_OK;
}

static int psa_score_batch_dev_2bit_impl(Engine* e, const tsta_params* p, int64_t n_pairs, const uint8_t* d_packed,
                                        const int64_t* d_boff1, const int32_t* d_len1, const int64_t* d_boff2,
                                        const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream)
{
    if (!e || !p || n_pairs < 0 || max_len < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!d_packed || !d_boff1 || !d_len1 || !d_boff2 || !d_len2 || !d_out) return TSTA_ERR_INVALID_ARG;
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : e->stream;
    const int64_t stride = ((int64_t)max_len + 15) / 16 * 16 + 16;
    const size_t nb = (size_t)n_pairs;
    if ((rc = e->reserve(e->unpacked, 2 * nb * (size_t)stride + 2 * nb * sizeof(int64_t) + 64, "hipMalloc(unpacked sequences)")) != TSTA_OK) return rc;
    uint8_t* d_seqs = reinterpret_cast<uint8_t*>(e->unpacked.p);
    int64_t* d_off1 = reinterpret_cast<int64_t*>(d_seqs + 2 * nb * (size_t)stride);
    int64_t* d_off2 = d_off1 + nb;
    hipLaunchKernelGGL(tsta::psa_unpack_2bit_kernel, dim3((unsigned)std::min<int64_t>(2 * n_pairs, 65536)), dim3(256), 0, s,
                       d_packed, d_boff1, d_len1, d_boff2, d_len2, n_pairs, stride, d_seqs, d_off1, d_off2);
    if ((rc = e->check(hipGetLastError(), "unpack launch")) != TSTA_OK) return rc;
    return tsta::psa_score_batch_dev_impl(e, p, n_pairs, d_seqs, d_off1, d_len1, d_off2, d_len2, d_out, max_len, stream);
}

int tsta_psa_score_batch_dev_2bit(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* d_packed,
                                  const int64_t* d_boff1, const int32_t* d_len1, const int64_t* d_boff2,
                                  const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream)
{
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = psa_score_batch_dev_2bit_impl(e, p, n_pairs, d_packed, d_boff1, d_len1, d_boff2, d_len2, d_out, max_len, stream);
    return rc == TSTA_OK ? tsta::psa_reference_warning(e, p) : rc;
}

int tsta_psa_score_batch_2bit(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* packed,
                              int64_t packed_bytes, const int64_t* boff1, const int32_t* len1,
                              const int64_t* boff2, const int32_t* len2, int32_t* out)
{
    if (!eng || !p || n_pairs < 0 || packed_bytes < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!packed || !boff1 || !len1 || !boff2 || !len2 || !out) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int32_t max_len = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
        if (len1[k] < 0 || len2[k] < 0 || boff1[k] < 0 || boff2[k] < 0 ||
            boff1[k] + ((int64_t)len1[k] + 3) / 4 > packed_bytes || boff2[k] + ((int64_t)len2[k] + 3) / 4 > packed_bytes) {
            e->set_error("pair %lld: packed sequence range outside packed[0..%lld)", (long long)k, (long long)packed_bytes);
            return TSTA_ERR_INVALID_ARG;
        }
        max_len = std::max(max_len, std::max(len1[k], len2[k]));
    }
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    const size_t nb = (size_t)n_pairs;
    std::vector<int64_t> h_off(2 * nb);
    std::vector<int32_t> h_len(2 * nb);
    for (size_t k = 0; k < nb; ++k) { h_off[k] = boff1[k]; h_off[nb + k] = boff2[k]; h_len[k] = len1[k]; h_len[nb + k] = len2[k]; }
    const size_t meta_bytes = nb * (2 * sizeof(int64_t) + 2 * sizeof(int32_t));
    if ((rc = e->reserve(e->in_seqs, (size_t)packed_bytes + 16, "hipMalloc(packed seqs)")) != TSTA_OK) return rc;
    if ((rc = e->reserve(e->in_meta, meta_bytes, "hipMalloc(meta)")) != TSTA_OK) return rc;
    if ((rc = e->reserve(e->out_scores, nb * sizeof(int32_t), "hipMalloc(scores)")) != TSTA_OK) return rc;
    char* m = reinterpret_cast<char*>(e->in_meta.p);
    int64_t* d_off1 = reinterpret_cast<int64_t*>(m);
    int64_t* d_off2 = d_off1 + nb;
    int32_t* d_len1 = reinterpret_cast<int32_t*>(d_off2 + nb);
    int32_t* d_len2 = d_len1 + nb;
    hipStream_t s = e->stream;
    if ((rc = e->check(hipMemcpyAsync(e->in_seqs.p, packed, (size_t)packed_bytes, hipMemcpyHostToDevice, s), "H2D packed seqs")) ||
        (rc = e->check(hipMemcpyAsync(d_off1, h_off.data(), nb * 16, hipMemcpyHostToDevice, s), "H2D offsets")) ||
        (rc = e->check(hipMemcpyAsync(d_len1, h_len.data(), nb * 8, hipMemcpyHostToDevice, s), "H2D lengths")))
        return rc;
    rc = psa_score_batch_dev_2bit_impl(e, p, n_pairs, reinterpret_cast<const uint8_t*>(e->in_seqs.p), d_off1, d_len1,
                                       d_off2, d_len2, reinterpret_cast<int32_t*>(e->out_scores.p), max_len, s);
    if (rc != TSTA_OK) return rc;
    if ((rc = tsta::psa_score_rerun_if_abandoned(e, [&] {
            return psa_score_batch_dev_2bit_impl(e, p, n_pairs, reinterpret_cast<const uint8_t*>(e->in_seqs.p), d_off1, d_len1,
                                                 d_off2, d_len2, reinterpret_cast<int32_t*>(e->out_scores.p), max_len, s); })) != TSTA_OK)
        return rc;
    if ((rc = e->check(hipMemcpyAsync(out, e->out_scores.p, nb * 4, hipMemcpyDeviceToHost, s), "D2H scores"))) return rc;
    if ((rc = e->check(hipStreamSynchronize(s), "psa_score_batch_2bit sync"))) return rc;
    return tsta::psa_reference_warning(e, p);
}

}  // extern "C"
