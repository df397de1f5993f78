// gfx950 (MI355X / CDNA4) kernels of libsealfm.so and the C-ABI query entry points.
//
// All work here is HBM-latency/bandwidth bound integer work: dependent 128-byte
// gathers into a 16-ary wavelet matrix (one 128-B line per rank probe, four symbol
// bits per probe), 5-byte gathers into the suffix array, binary searches over
// doc boundaries.  No MFMA.
// Wave size is 64 throughout.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fmi_internal.h"

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fmi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FMI_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#include "fmi_device.h"

// sdsl wt_int::rank(i, c) as the reference reaches it, incl. i == size()+1
// (quirk Q1: occ(c) + q1[c]); i beyond that is undefined in the reference and
// is clamped to the same value here.
__device__ __forceinline__ uint64_t rank_like_sdsl(const FmiDev &ix, uint64_t c, uint64_t i, uint64_t *probes)
{
    if (i > ix.n) return (as_const(ix.C)[c + 1] - as_const(ix.C)[c]) + as_const(ix.q1)[c];
    if (i == 0) return 0;
    return wm_rank_sym(ix, c, i, probes);
}

// sdsl backward_search(csa, l, r, c, l_res, r_res) on the inclusive [l, r]
__device__ __forceinline__ void bs_step(const FmiDev &ix, uint64_t c, uint64_t l, uint64_t r,
                                        uint64_t &l_res, uint64_t &r_res, uint64_t *probes)
{
    const bool absent = (c > ix.max_sym) || (as_const(ix.C)[c + 1] == as_const(ix.C)[c]);
    if (absent && c > 0) { l_res = 1; r_res = 0; return; }
    const uint64_t cb = as_const(ix.C)[c];
    uint64_t rl, rr;
    const uint64_t j = r + 1;
    // the first step of every search starts from [0, size()] (index.py:106-107): rank(0) = 0 and the
    // rank one past the end is the quirk value -- a table look-up, no probe of the wavelet matrix
    if (l == 0 && j > ix.n) { l_res = cb; r_res = cb + rank_like_sdsl(ix, c, j, probes) - 1; return; }
    wm_rank_sym_pair(ix, c, l < ix.n ? l : ix.n, j < ix.n ? j : ix.n, rl, rr, probes);
    if (l > ix.n) rl = rank_like_sdsl(ix, c, l, probes);       // beyond size(): the quirk value, no probe
    if (j > ix.n) rr = rank_like_sdsl(ix, c, j, probes);
    l_res = cb + rl;
    r_res = cb + rr - 1;
}

// ---------------------------------------------------------------------------
// K1: backward_search_step for n independent triples (fm_index.cpp:67-76)
// ---------------------------------------------------------------------------
__global__ void k_bs_step(FmiDev ix, uint64_t n, const uint64_t *sym, const uint64_t *lo, const uint64_t *hi,
                          uint64_t *lo_out, uint64_t *hi_out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t l, r;
    bs_step(ix, sym[i], lo[i], hi[i], l, r, nullptr);
    lo_out[i] = l; hi_out[i] = r;
}

// K5: get_range / backward_search_multi over CSR sequences (fm_index.cpp:55-65,
// seal/index.py:102-111): start at (0, size()), one step per token, return (l, r+1)
template <typename OffT, typename TokT>
__global__ void k_get_range(FmiDev ix, uint64_t n_seq, const OffT *offsets, const TokT *tokens, int64_t shift,
                            uint64_t *lo_out, uint64_t *hi_out)
{
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seq) return;
    uint64_t l = 0, r = ix.n;
    for (uint64_t t = (uint64_t)offsets[s]; t < (uint64_t)offsets[s + 1]; t++) {
        uint64_t c = (uint64_t)((int64_t)tokens[t] + shift);
        bs_step(ix, c, l, r, l, r, nullptr);
    }
    lo_out[s] = l; hi_out[s] = r + 1;
}

// ---------------------------------------------------------------------------
// K2: interval -> distinct symbols (+counts)   (sdsl interval_symbols as used by
// fm_index.cpp:78-109).
//
// Work item = (row, top digit d1): ONE wavefront finds child d1 of the row's root
// interval itself (a single-digit rank at both ends -- every wave of a row repeats
// that probe, an L2/MALL hit for all but the first) and then expands the whole
// sub-tree below it.  No queue, no second launch, no inter-wave traffic.  Items are
// laid out d1-major: the waves resident at any moment work below the same few top
// digits, i.e. inside the same slices of every deeper level of the wavelet matrix
// (a level is sorted by the digits above it), which keeps the address-translation
// working set a fraction of the 11.5 GB structure (profiles/r1_gather_calib.txt:
// random 128-byte requests run 3.6x slower once they spread over more than ~3 GiB).
//
// Inside the wave the frontier lives in LDS, one array per relative level (a level-j
// array never holds more than min(16^j, 1024) nodes: it is only refilled, by at most
// 32 parents, when it and every deeper level are empty).  Each iteration pops up to
// 32 nodes of the deepest non-empty level; a node is served by a PAIR of lanes, one
// per interval end: each lane loads one 128-byte block and computes the sixteen
// ranks of its end (32 + 16 live registers instead of 64 + 32 for both ends in one
// lane: < 128 VGPRs, four waves per SIMD), the partner's ranks arrive by DPP
// (quad_perm), and each lane of the pair compacts eight of the sixteen children with
// ballot + mbcnt.  In mask mode the leaves of a sub-tree are one contiguous symbol
// range owned by this wave alone: they are collected in an LDS bitmap (one byte store
// per leaf-level half node) and flushed with plain coalesced stores -- no global
// atomics except on the (at most two) words a wave shares with its neighbours.
// ---------------------------------------------------------------------------
struct ExpandItem {
    uint64_t lo, hi;
    uint32_t row, pad;
};

enum { EMIT_BITS = 0, EMIT_DENSE = 1 };

struct EmitTarget {
    uint32_t *bits;       // EMIT_BITS : [rows][words_per_row]
    uint64_t words_per_row;
    int64_t shift;        // token = symbol - shift
    uint64_t vocab;
    uint64_t *dense;      // EMIT_DENSE: [rows][dense_stride] counts by symbol
    uint64_t dense_stride;
};

// LDS traffic between the lanes of ONE wave: DS operations of a wave execute in
// order, so a wavefront-scope fence (no cache maintenance) plus a scheduling
// barrier is all that is needed.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

static constexpr uint32_t PROBE_SLOTS = 256;         // counter slots (measurement mode only), one 64-byte line each
static constexpr int EXP_LVL_CAP = 1024;
static constexpr uint32_t EXP_PAIRS = 32;            // nodes per wave iteration (one lane pair each)
// relative level j occupies [lvl_off(j), lvl_off(j) + min(16^j, 1024))
__host__ __device__ constexpr int lvl_cap(int j) { return j < 3 ? (1 << (4 * j)) : EXP_LVL_CAP; }
__host__ __device__ constexpr int lvl_off(int j) { return j <= 3 ? ((1 << (4 * j)) - 1) / 15 : 273 + (j - 3) * EXP_LVL_CAP; }
// LDS slots a wave needs to expand a sub-tree spanning `nlev` stored levels
__host__ __device__ constexpr int exp_slots(int nlev) { return nlev <= 0 ? 1 : lvl_off(nlev - 1) + lvl_cap(nlev - 1); }

// measurement mode (fmi_dev_enable_probe_count): distinct blocks loaded, nodes of the binary model of
// SURVEY.md 8(d), wave iterations, nodes expanded
struct ExpCounters { uint32_t probes, model, iters, nodes; };      // per lane, per launch

// lane <-> lane^1
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// frontier slot: positions < 2^40 (8 high bits each) + the node's symbol prefix (<= 16 bits)
__device__ __forceinline__ uint4 pack_node(uint64_t lo, uint64_t hi, uint32_t prefix)
{
    return make_uint4((uint32_t)lo, (uint32_t)hi, (uint32_t)(lo >> 32) | ((uint32_t)(hi >> 32) << 8) | (prefix << 16), 0u);
}

// the same node in the BINARY level-per-bit model of SURVEY.md 8(d): itself plus its non-empty halves,
// quarters and pairs of children; the phantom high bits of the top digit (level 0) have no level
__device__ __forceinline__ uint32_t model_nodes(uint32_t em, uint32_t k, uint32_t pad_bits)
{
    const uint32_t g2 = (em | (em >> 1)) & 0x5555u, g4 = (g2 | (g2 >> 2)) & 0x1111u, g8 = (g4 | (g4 >> 4)) & 0x0101u;
    const uint32_t skip = k == 0 ? pad_bits : 0;
    return (skip < 1 ? 1u : 0u) + (skip < 2 ? (uint32_t)__popc(g8) : 0u) + (skip < 3 ? (uint32_t)__popc(g4) : 0u) + (uint32_t)__popc(g2);
}

// rank of the lane among the lanes of a ballot
__device__ __forceinline__ uint32_t lane_rank_in(uint64_t bal)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
}

__device__ __forceinline__ void flush_counters(uint64_t *probe_counter, const ExpCounters &c)
{
    // one 64-byte line per slot: thousands of waves adding to ONE address cost tens of microseconds
    unsigned long long *slot = (unsigned long long *)probe_counter + (size_t)(blockIdx.x & (PROBE_SLOTS - 1)) * 8;
    if (c.probes) atomicAdd(slot, (unsigned long long)c.probes);
    if (c.model) atomicAdd(slot + 3, (unsigned long long)c.model);
    if ((threadIdx.x & 63) == 0 && c.iters) {
        atomicAdd(slot + 1, (unsigned long long)c.iters);
        atomicAdd(slot + 2, (unsigned long long)c.nodes);
    }
}

// the sixteen existing-children flags of the ROOT node of a row (measurement mode only: the binary
// model needs the whole mask, the expansion itself only ever asks for one digit per wave); a rolled
// loop of single-digit ranks over the same one or two blocks, so that it costs no registers
__device__ __forceinline__ uint32_t root_children_mask(const FmiDev &ix, uint64_t lo, uint64_t hi)
{
    uint32_t em = 0;
#pragma unroll 1
    for (uint32_t d = 0; d < 16; d++) em |= (uint32_t)(wm_step(ix, 0, hi, d) > wm_step(ix, 0, lo, d)) << d;
    return em;
}

// Expansion of the sub-tree below ONE node (level `root` >= 1 ... D-1, interval [rlo, rhi), symbol
// prefix rprefix) by one wavefront.  s_node: exp_slots(D - root) frontier slots; s_cnt: one counter per
// relative level; EMIT_BITS: s_bits8 = the wave's leaf bitmap (16^(D-root) bits, zeroed by the caller):
// bit i = symbol (rprefix << 4 (D-root)) + i.  `counting` (wave-uniform) switches the measurement
// bookkeeping on.
template <int MODE, bool SB>
__device__ __forceinline__ void expand_subtree(const FmiDev &ix, uint4 *s_node, uint32_t *s_cnt, uint8_t *s_bits8,
                                               const uint32_t row, const uint32_t root, const uint64_t rlo, const uint64_t rhi,
                                               const uint32_t rprefix, const EmitTarget &tgt, const bool counting, ExpCounters &ctr)
{
    const uint32_t lane = threadIdx.x & 63, end = lane & 1, pair = lane >> 1;
    const uint32_t D = ix.dlevels;
    const uint32_t pad_bits = FMI_DIGIT_BITS * D - ix.levels;   // phantom high bits of the top digit (0..3)
    const uint32_t rel_mask = (1u << (FMI_DIGIT_BITS * (D - 1 - root))) - 1;   // prefix bits of a leaf-level node below the root
    if (lane < FMI_MAX_DLEVELS) s_cnt[lane] = lane == 0 ? 1u : 0u;
    if (lane == 0) s_node[0] = pack_node(rlo, rhi, rprefix);
    wave_sync();
    int deepest = 0;   // relative level of the deepest non-empty array (wave uniform)
    while (deepest >= 0) {
        // LDS hands the counter back in a VGPR; it is the same in every lane, and the whole loop
        // (level, offsets, the per-level dbase[] loads) stays scalar only if the compiler knows
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(s_cnt[deepest]);
        if (cnt == 0) { deepest--; continue; }
        const uint32_t m = cnt < EXP_PAIRS ? cnt : EXP_PAIRS;
        const uint32_t base = lvl_off(deepest) + (cnt - m);
        const uint32_t k = root + deepest;      // absolute level of the popped nodes
        const bool act = pair < m;
        uint4 nd = make_uint4(0u, 0u, 0u, 0u);
        if (act) nd = s_node[base + pair];      // both lanes of the pair read the same slot (LDS broadcast)
        wave_sync();
        if (lane == 0) s_cnt[deepest] = cnt - m;
        const uint64_t lo = (uint64_t)nd.x | ((uint64_t)(nd.z & 0xff) << 32);
        const uint64_t hi = (uint64_t)nd.y | ((uint64_t)((nd.z >> 8) & 0xff) << 32);
        const uint32_t prefix = nd.z >> 16;
        const uint64_t p = end ? hi : lo;                              // my end of the interval
        const uint64_t blk = p >> FMI_BLOCK_SHIFT, oblk = (end ? lo : hi) >> FMI_BLOCK_SHIFT;
        uint32_t r[16];
#pragma unroll
        for (uint32_t d = 0; d < 16; d++) r[d] = 0;
        if (act) {
            HBlock b;
            wm_load_block(ix, k, blk, b);        // when both ends share a block the pair asks for the same line once
            wm_block_ranks(b, (uint32_t)p & (FMI_BLOCK_BITS - 1), r);
        }
        // superblocked index: the rows of the two ends
        const uint64_t *rowl = nullptr, *rowh = nullptr;
        if constexpr (SB) {
            const uint64_t mine = ((uint64_t)k * ix.nsb + (blk >> ix.sb_shift)) * FMI_ARITY;
            const uint64_t other = ((uint64_t)k * ix.nsb + (oblk >> ix.sb_shift)) * FMI_ARITY;
            rowl = ix.sbase + (end ? other : mine);
            rowh = ix.sbase + (end ? mine : other);
        }
        const bool leaf = (k + 1 == D);
        uint32_t hm = 0;                         // which of MY eight children (digits s + 8 * end) exist
        uint32_t added = 0;
        const uint32_t dst = leaf ? 0u : lvl_off(deepest + 1) + __builtin_amdgcn_readfirstlane(s_cnt[deepest + 1]);
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) {
            // lane `end` = 0 holds rank_lo[] and takes digit s, lane 1 holds rank_hi[] and takes digit s + 8
            const uint32_t x = dpp_xor1(r[s]), y = dpp_xor1(r[s + 8]);
            const uint32_t cl = end ? y : r[s], ch = end ? r[s + 8] : x;
            const uint32_t dm = s + 8 * end;
            uint64_t clo, chi;
            if constexpr (SB) {
                clo = (act ? rowl[dm] : 0) + cl; chi = (act ? rowh[dm] : 0) + ch;
            } else {
                const uint64_t bs = end ? ix.dbase[k][s + 8] : ix.dbase[k][s];
                clo = bs + cl; chi = bs + ch;
            }
            const bool ex = act && chi > clo;
            hm |= (uint32_t)ex << s;
            if (leaf) {
                if (MODE == EMIT_DENSE && ex) tgt.dense[(uint64_t)row * tgt.dense_stride + ((prefix << 4) | dm)] = chi - clo;
            } else {
                const uint64_t bal = __ballot(ex);
                if (ex) s_node[dst + added + lane_rank_in(bal)] = pack_node(clo, chi, (prefix << 4) | dm);
                added += (uint32_t)__popcll(bal);
            }
        }
        if (counting) {
            const uint32_t other_hm = dpp_xor1(hm);
            if (act && end == 0) {
                ctr.model += model_nodes(hm | (other_hm << 8), k, pad_bits);
                ctr.probes += oblk != blk ? 2 : 1;
            }
            ctr.iters++; ctr.nodes += m;
        }
        if (leaf) {
            if (MODE == EMIT_BITS) {
                if (prefix == 0 && end == 0) hm &= ~1u;      // symbol 0 is the sentinel, never a token
                if (hm) s_bits8[((prefix & rel_mask) << 1) + end] = (uint8_t)hm;
            }
        } else {
            wave_sync();
            if (lane == 0) s_cnt[deepest + 1] += added;
            wave_sync();
            if (added) deepest++;
        }
    }
    wave_sync();
}

// One level of the sub-trees of the W items of a workgroup (k_constrain, W waves per workgroup, dlevels <= 4),
// served by all its waves together: the level's nodes -- of all W items, each slot carries its item in .w -- are
// one array in LDS; chunk c (32 nodes, one lane pair each) goes to wave c % W.  Above the leaf level the children
// are appended to the next level's array (space reserved with one LDS atomic per chunk: pass 1 finds which
// children exist, pass 2 writes them); at the leaf level a node's two 8-bit existence masks go into the LDS
// bitmap of its item (distinct bytes for distinct nodes: no conflicts between waves).  The caller puts a
// workgroup barrier between levels.  Against one self-contained wave per item this (a) balances the waves -- an
// item with 256 leaf-level nodes next to items with a handful costs every wave one or two iterations instead of
// one wave eight --, (b) packs the lane pairs across items, and (c) runs the upper levels, where an item has 1
// and <= 16 nodes, in 1/8 and <= 1/2 of the wave iterations: with four waves per SIMD in lockstep those
// iterations are VALU time, not latency.
template <bool SB, int W>
__device__ __forceinline__ void wg_level(const FmiDev &ix, const uint4 *s_in, const uint32_t total, uint4 *s_out, uint32_t *s_cnt_out,
                                         const uint32_t k, uint4 *s_bitmaps, const uint32_t bm_slots, const bool counting, ExpCounters &ctr,
                                         const uint32_t my_rank, const uint32_t n_live)
{
    const uint32_t lane = threadIdx.x & 63, end = lane & 1, pair = lane >> 1;
    const uint32_t D = ix.dlevels;
    const bool leaf = (k + 1 == D);
    const uint32_t pad_bits = FMI_DIGIT_BITS * D - ix.levels;
    const uint32_t rel_mask = (1u << (FMI_DIGIT_BITS * (D - 2))) - 1;      // sub-tree roots are level-1 nodes
    // chunk c goes to the c-th of the waves that are still in the workgroup (waves whose item is empty have left)
    for (uint32_t c0 = my_rank * EXP_PAIRS; c0 < total; c0 += n_live * EXP_PAIRS) {
        const uint32_t g = c0 + pair;
        const bool act = g < total;
        uint4 nd = make_uint4(0u, 0u, 0u, 0u);
        if (act) nd = s_in[g];               // both lanes of the pair read the same slot (LDS broadcast)
        const uint64_t lo = (uint64_t)nd.x | ((uint64_t)(nd.z & 0xff) << 32);
        const uint64_t hi = (uint64_t)nd.y | ((uint64_t)((nd.z >> 8) & 0xff) << 32);
        const uint32_t prefix = nd.z >> 16, item = nd.w;
        const uint64_t p = end ? hi : lo;                              // my end of the interval
        const uint64_t blk = p >> FMI_BLOCK_SHIFT, oblk = (end ? lo : hi) >> FMI_BLOCK_SHIFT;
        uint32_t r[16];
#pragma unroll
        for (uint32_t d = 0; d < 16; d++) r[d] = 0;
        if (act) {
            HBlock b;
            wm_load_block(ix, k, blk, b);        // when both ends share a block the pair asks for the same line once
            wm_block_ranks(b, (uint32_t)p & (FMI_BLOCK_BITS - 1), r);
        }
        const uint64_t *rowl = nullptr, *rowh = nullptr;   // superblocked index: the rows of the two ends (!act: block 0's, a valid address)
        if constexpr (SB) {
            const uint64_t mine = ((uint64_t)k * ix.nsb + (blk >> ix.sb_shift)) * FMI_ARITY;
            const uint64_t other = ((uint64_t)k * ix.nsb + (oblk >> ix.sb_shift)) * FMI_ARITY;
            rowl = ix.sbase + (end ? other : mine);
            rowh = ix.sbase + (end ? mine : other);
        }
        // pass 1: which of MY eight children (digits s + 8 * end) exist
        uint32_t hm = 0, nchild = 0;
        uint64_t bal[8];
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) {
            // lane `end` = 0 holds rank_lo[] and takes digit s, lane 1 holds rank_hi[] and takes digit s + 8
            const uint32_t x = dpp_xor1(r[s]), y = dpp_xor1(r[s + 8]);
            const uint32_t cl = end ? y : r[s], ch = end ? r[s + 8] : x;
            bool ex;
            if constexpr (SB) {
                const uint32_t dm = s + 8 * end;
                ex = act && (rowh[dm] + ch) > (rowl[dm] + cl);
            } else {
                ex = act && ch > cl;        // both ends add the same dbase[k][digit]
            }
            hm |= (uint32_t)ex << s;
            if (!leaf) { bal[s] = __ballot(ex); nchild += (uint32_t)__popcll(bal[s]); }
        }
        if (counting) {
            const uint32_t other_hm = dpp_xor1(hm);
            if (act && end == 0) {
                ctr.model += model_nodes(hm | (other_hm << 8), k, pad_bits);
                ctr.probes += oblk != blk ? 2 : 1;
            }
            const uint32_t left = total - c0;
            ctr.iters++; ctr.nodes += left < EXP_PAIRS ? left : EXP_PAIRS;
        }
        if (leaf) {
            if (prefix == 0 && end == 0) hm &= ~1u;          // symbol 0 is the sentinel, never a token
            if (hm) reinterpret_cast<uint8_t *>(s_bitmaps + (size_t)item * bm_slots)[((prefix & rel_mask) << 1) + end] = (uint8_t)hm;
        } else if (nchild) {
            // pass 2: reserve, then write the children (positions recomputed: cheaper than sixteen live registers)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(s_cnt_out, nchild);
            uint32_t run = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (uint32_t s = 0; s < 8; s++) {
                const uint32_t x = dpp_xor1(r[s]), y = dpp_xor1(r[s + 8]);
                const uint32_t cl = end ? y : r[s], ch = end ? r[s + 8] : x;
                const uint32_t dm = s + 8 * end;
                uint64_t clo, chi;
                if constexpr (SB) {
                    clo = rowl[dm] + cl; chi = rowh[dm] + ch;
                } else {
                    const uint64_t bs = end ? ix.dbase[k][s + 8] : ix.dbase[k][s];
                    clo = bs + cl; chi = bs + ch;
                }
                if ((hm >> s) & 1u) {
                    uint4 c = pack_node(clo, chi, (prefix << 4) | dm);
                    c.w = item;
                    s_out[run + lane_rank_in(bal[s])] = c;
                }
                run += (uint32_t)__popcll(bal[s]);
            }
        }
    }
}

// 32 bits [o, o + 32) of an LDS bit array of nw words, zeros outside it
__device__ __forceinline__ uint32_t lds_bits32(const uint32_t *s_w, int32_t nw, int64_t o)
{
    const int64_t i = o >> 5;                    // arithmetic shift: floor
    const uint32_t sh = (uint32_t)o & 31;
    const uint32_t w0 = (i >= 0 && i < nw) ? s_w[i] : 0u;
    const uint32_t w1 = (i + 1 >= 0 && i + 1 < nw) ? s_w[i + 1] : 0u;
    return sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
}

// The wave's leaf bitmap (symbols [sym0, sym0 + nsym)) -> the row's token bitmap (token = symbol - shift,
// clipped to [0, vocab)).  Words that lie entirely inside the wave's own token range are written with plain
// stores (the bitmap was zero and nobody else owns a bit of them), the at most two shared words with atomicOr.
__device__ __forceinline__ void flush_leaf_bits(const EmitTarget &t, uint32_t row, const uint32_t *s_w, uint32_t sym0, uint32_t nsym)
{
    const uint32_t lane = threadIdx.x & 63;
    const int64_t tlo = (int64_t)sym0 - t.shift;
    int64_t a = tlo, b = tlo + (int64_t)nsym;
    if (a < 0) a = 0;
    if (b > (int64_t)t.vocab) b = (int64_t)t.vocab;
    if (a >= b) return;
    const int32_t nw = (int32_t)((nsym + 31) >> 5);
    uint32_t *rowp = t.bits + (uint64_t)row * t.words_per_row;
    const int64_t W1 = (b - 1) >> 5;
    for (int64_t W = (a >> 5) + lane; W <= W1; W += 64) {
        uint32_t v = lds_bits32(s_w, nw, 32 * W - tlo);
        uint32_t keep = ~0u;
        if (32 * W < a) keep &= ~0u << (uint32_t)(a - 32 * W);
        if (32 * W + 32 > b) keep &= (1u << (uint32_t)(b - 32 * W)) - 1;
        v &= keep;
        if (!v) continue;
        if (keep == ~0u) rowp[W] = v; else atomicOr(&rowp[W], v);
    }
}

// child `d` of the root interval [lo, hi) (level 0): two single-digit ranks, loads issued back to back
__device__ __forceinline__ void root_child(const FmiDev &ix, uint64_t lo, uint64_t hi, uint32_t d, uint64_t &clo, uint64_t &chi)
{
    const uint64_t a = wm_step(ix, 0, lo, d), b = wm_step(ix, 0, hi, d);
    clo = a; chi = b;
}

// API mode (distinct / distinct_count / distinct_count_multi): dense per-row counts.  One wave per
// (interval, top digit), d1-major; dynamic LDS = exp_slots(D - 1) frontier slots + counters.
template <bool SB>
__global__ __launch_bounds__(64) void k_expand_dense(FmiDev ix, const ExpandItem *items, uint32_t rows, EmitTarget tgt, uint64_t *probe_counter)
{
    extern __shared__ uint4 s_dyn[];
    const uint32_t D = ix.dlevels;
    uint4 *s_node = s_dyn;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_node + exp_slots((int)D - 1));
    const uint32_t d1 = blockIdx.x / rows, i = blockIdx.x - d1 * rows;
    const ExpandItem it = items[i];
    if (it.hi <= it.lo) return;
    const bool counting = probe_counter != nullptr;
    ExpCounters ctr{0, 0, 0, 0};
    uint64_t clo, chi;
    root_child(ix, it.lo, it.hi, d1, clo, chi);
    if (counting && d1 == 0 && (threadIdx.x & 63) == 0) {
        ctr.probes += (it.lo >> FMI_BLOCK_SHIFT) != (it.hi >> FMI_BLOCK_SHIFT) ? 2 : 1;
        ctr.model += model_nodes(root_children_mask(ix, it.lo, it.hi), 0, FMI_DIGIT_BITS * D - ix.levels);
    }
    if (chi > clo) {
        if (D == 1) { if ((threadIdx.x & 63) == 0) tgt.dense[(uint64_t)it.row * tgt.dense_stride + d1] = chi - clo; }
        else expand_subtree<EMIT_DENSE, SB>(ix, s_node, s_cnt, nullptr, it.row, 1, clo, chi, d1, tgt, counting, ctr);
    }
    if (counting) flush_counters(probe_counter, ctr);
}
// dense per-row symbol counts -> CSR, ascending symbols.  One workgroup per row.
__global__ __launch_bounds__(256) void k_dense_count(const uint64_t *dense, uint64_t stride, uint64_t nsym, uint64_t *row_k)
{
    __shared__ uint32_t s_part[256];
    const uint64_t *d = dense + (uint64_t)blockIdx.x * stride;
    uint32_t c = 0;
    for (uint64_t s = threadIdx.x; s < nsym; s += 256) c += d[s] != 0;
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) row_k[blockIdx.x] = s_part[0];
}

__global__ __launch_bounds__(256) void k_dense_compact(const uint64_t *dense, uint64_t stride, uint64_t nsym,
                                                       const uint64_t *offsets, uint64_t *syms, uint64_t *cnts)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const uint64_t *d = dense + (uint64_t)blockIdx.x * stride;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const uint64_t out0 = offsets[blockIdx.x];
    for (uint64_t s0 = 0; s0 < nsym; s0 += 256) {
        const uint64_t s = s0 + threadIdx.x;
        const uint64_t v = s < nsym ? d[s] : 0;
        const uint64_t b = __ballot(v != 0);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = s_base;
        for (uint32_t w = 0; w < wv; w++) before += s_wave[w];
        if (v != 0) {
            const uint64_t o = out0 + before + (uint32_t)__popcll(b & ((1ull << lane) - 1));
            syms[o] = s;
            if (cnts) cnts[o] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// a9: IndexBasedLogitsProcessor.__call__, cur_len >= 2 (seal/beam_search.py:79-140)
// as ONE launch.  One wave per (row, top digit d1), d1-major (see K2).  Every wave of
// a row repeats the row's own few dependent probes -- range of the prefix (one
// backward-search step from the parent's kept range, or the full search), class of
// the row (lines 87-105 and the branch order of 111-131), child d1 of the root node --
// with wave-uniform addresses read through the constant address space: scalar loads
// and scalar ALU (one request each, L2/MALL hits for all but the first wave), no VALU
// issue slots.  Then the sub-tree below (row, d1) is expanded into the item's LDS
// bitmap, which its wave stores: by the eight waves of the workgroup together, level
// by level, for vocabularies of up to four digit levels (wg_level), by the wave alone
// otherwise (expand_subtree).  Only the d1 = 0 wave writes the row's side effects
// (kept range, measurement counters).
// ---------------------------------------------------------------------------
static constexpr int MAX_FORCE = 8;
struct ForceFrom { int64_t tok[MAX_FORCE]; uint32_t n; };
// Row groups: one call may serve the rows of several decodes that run in lockstep (the searcher's body and title
// decodes as ONE loop of batch * beams * 2 rows): group g = rows [grp_first[g], grp_first[g + 1]) with its own
// end-of-sequence token and forced prefix (reference retrieval.py:70-83 vs 162-176).  One group = the classic call.
static constexpr int MAX_ROW_GROUPS = 3;

// Row-first calls (k_constrain_rows, then k_constrain): what ONE wave per row worked out for its row -- range, class, and the
// sixteen children of its root node -- for the (row, top digit) waves of the second launch to pick up with a single load
struct RowPre { uint64_t lo, hi; int64_t single; uint32_t expand, child_mask; };

struct ConstrainArgs {
    uint32_t rows, ndig0;          // grid = rows * ndig0 waves
    uint64_t cur_len;
    const int64_t *ids;            // [rows][ids_stride], the first cur_len of each row used
    uint64_t ids_stride;
    int64_t shift, pad_id;
    uint32_t grp_first[MAX_ROW_GROUPS];   // first row of group g (grp_first[0] = 0; unused groups: 0xffffffff)
    int64_t grp_eos[MAX_ROW_GROUPS];
    ForceFrom grp_ff[MAX_ROW_GROUPS];
    int64_t grp_stop[MAX_ROW_GROUPS];     // stop_at_count of the group's decode (reference retrieval.py:70-83: the body decode's only; titles / codes 0)
    int always_allow_eos;
    uint64_t vocab, words_per_row;
    uint32_t *bits;                // [rows][words_per_row], zero on entry
    uint32_t *clear;               // words [0, clear_words) of the OTHER bitmap buffer are zeroed for the next call
    uint64_t clear_words;
    const uint64_t *st_in;         // incremental prefix state of the previous step, or null
    const int64_t *parent;
    uint64_t *st_out;
    uint64_t *probe_counter;
    uint64_t *tstamp;              // tools only: 8 realtime stamps (100 MHz) per wave, or null
    uint32_t groups;               // W > 1: row groups per top digit (ceil(rows / W); grid = groups * ndig0 workgroups)
    int leave_early;               // the waves of empty items leave before the level loops (see k_constrain)
    RowPre *pre_rows;              // row-first calls: [rows], written by k_constrain_rows, read by k_constrain (else null)
    uint64_t *pre_child;           // [rows][16][2]: child d of the row's root node, [lo, hi) on level 1
};

// a special token (pad / eos) of the row: into the LDS bitmap of the wave that owns its symbol; tokens
// whose symbol lies above every wave's range are set by the d1 = 0 wave directly
__device__ __forceinline__ void set_special(const FmiDev &ix, const ConstrainArgs &a, uint32_t *s_bits, uint32_t row, uint32_t d1,
                                            uint32_t sub_bits, int64_t tok)
{
    if (tok < 0 || (uint64_t)tok >= a.vocab) return;
    const int64_t sym = tok + a.shift;
    const bool owned = sym >= 0 && (uint64_t)(sym >> sub_bits) < a.ndig0;
    if (owned) {
        if ((uint32_t)(sym >> sub_bits) == d1) {
            const uint32_t o = (uint32_t)sym & ((1u << sub_bits) - 1);
            s_bits[o >> 5] |= 1u << (o & 31);
        }
    } else if (d1 == 0) {
        atomicOr(&a.bits[(uint64_t)row * a.words_per_row + ((uint64_t)tok >> 5)], 1u << (tok & 31));
    }
}

// one allowed token of row r straight into the row's bitmap in global memory
__device__ __forceinline__ void set_bit_global(const ConstrainArgs &a, uint32_t r, int64_t tok)
{
    if (tok < 0 || (uint64_t)tok >= a.vocab) return;
    atomicOr(&a.bits[(uint64_t)r * a.words_per_row + ((uint64_t)tok >> 5)], 1u << (tok & 31));
}

// W = waves per workgroup.  W = 1: one self-contained wave per item (any depth; dynamic LDS: exp_slots(D - 1)
// frontier slots + 8 counters + the item's leaf bitmap).  W > 1 (host: 2 <= dlevels <= 4): the W waves of a
// workgroup take W consecutive rows of one top digit; each finds its row's range and root child, then the
// workgroup expands the W sub-trees level by level together (wg_level).  Dynamic LDS of that shape, in 16-byte
// slots: [2: node counters per level] [W x bm_slots: leaf bitmaps, 16^(D-1) bits each]
//        [level j = 0 .. D-2: W * 16^j node slots]
__host__ __device__ constexpr uint32_t constrain_bm_slots(uint32_t D) { return (((1u << (FMI_DIGIT_BITS * (D - 1))) + 31) / 32 + 3) / 4; }
__host__ __device__ constexpr uint32_t constrain_lvl_off(uint32_t W, uint32_t j) { return W * (((1u << (FMI_DIGIT_BITS * j)) - 1) / 15); }
__host__ __device__ constexpr uint32_t constrain_lds_slots(uint32_t D, uint32_t W)
{
    return W > 1 ? 2 + W * constrain_bm_slots(D) + constrain_lvl_off(W, D - 1)
                 : (uint32_t)exp_slots((int)D - 1) + 2 + constrain_bm_slots(D);
}

// k_constrain<.., W > 1> keeps the per-level node counters in s_cnt[0 .. dlevels - 2] and the mask of the waves that stay in
// s_cnt[7]; waves of empty items END before the workgroup's level barriers (gfx9 s_barrier waits for the waves that have not
// terminated: CDNA ISA "S_BARRIER ... waves that have ended are not counted"; leave_early = 0 keeps them, and a GPU test runs both)
static_assert(FMI_MAX_DLEVELS < 7, "s_cnt[7] holds the live-wave mask: the level counters must end below it");
static constexpr int CONSTRAIN_WG = 8;       // 39 KB of LDS per workgroup at BART's depth, two workgroups per CU
// Chained calls (the decode loop's calls from the 3rd token on: the rows' chains are done, most items are empty, a few hundred are wide):
// workgroups of FOUR waves.  Measured on the bench workload with 2 / 4 / 8 / 16 waves (profiles/r6_constrain_ab_workgroup_width.txt, us per
// batch): 313 / 317 / 329 / 335 -- the level barriers of a workgroup wait for its slowest dependent access, and fewer waves wait less;
// packing the live items densely into 8-wave workgroups (a list written by k_beam_advance) lost for the same reason
// (profiles/r6_constrain_ab_item_list_lost.txt: 365 vs 346).  4 rather than 2: a wide row still shares its leaf level with three helpers.
static constexpr int CONSTRAIN_WG_CHAINED = 4;

// the row's group (wave-uniform: scalar compares on kernel arguments)
__device__ __forceinline__ uint32_t row_group(const ConstrainArgs &a, uint32_t r)
{
    uint32_t grp = 0;
#pragma unroll
    for (int g = 1; g < MAX_ROW_GROUPS; g++) grp += r >= a.grp_first[g] ? 1u : 0u;
    return grp;
}

// The row's own dependent chain (beam_search.py:87-131): range of its prefix -- one backward-search step from the range kept
// for its parent row, or the full search --, count of the prefix without its last token, and the class: `single` >= 0 = the
// only token allowed (eos below stop_at_count, pad for a finished row), else `expand` the range [lo, hi).  Wave-uniform
// addresses read through the constant address space: scalar loads, scalar ALU.
__device__ __forceinline__ void row_range_and_class(const FmiDev &ix, const ConstrainArgs &a, const uint32_t r, const bool valid, const bool write_state,
                                                    uint64_t &lo, uint64_t &hi, int64_t &single, bool &expand, int64_t &eos_id,
                                                    uint64_t &probes, uint32_t &model)
{
    uint64_t count = 0;
    bool dead = true;
    const uint32_t grp = row_group(a, r);
    eos_id = a.grp_eos[grp];
    const ForceFrom &ff = a.grp_ff[grp];
    lo = hi = 0;
    if (valid) {
        // ids / parent / the kept ranges were written by earlier launches, not by this one: constant here
        const cptr<int64_t> sent = as_const(a.ids) + (uint64_t)r * a.ids_stride;
        const int64_t last = sent[a.cur_len - 1];
        dead = last == eos_id || last == a.pad_id;
        // A finished row (last token eos / pad) needs no range for ITS mask (beam_search.py:89-92: low = high = count = 0), but the row
        // that continues it does: when a query has fewer than 2K finite candidates the beam fills up with not-allowed tokens (quirk Q4)
        // and such a row's next prefix runs THROUGH the eos / pad -- the reference searches it from scratch (an eos inside a prefix
        // matches across a document end, a pad matches nothing).  So the range is advanced and kept for every row (round 5: a finished
        // row used to leave its slot of the kept ranges untouched, and its continuation read a stale range; found by holding the
        // bitmaps the timed path applies, not recomputed ones, to the oracle).  Only live rows pay for a search when nothing is kept.
        if (!dead || a.st_out) {
            // get_range(force_decoding_from + sent[1:]) and get_count(... sent[1:-1])
            uint64_t l = 0, rr = ix.n, cnt = 0;
            if (a.st_in) {
                // incremental: the row extends row parent[r] of the previous step, whose inclusive range
                // [l, rr] after the same prefix was kept -- one backward-search step instead of len
                const uint64_t pr = (uint64_t)as_const(a.parent)[r];
                l = as_const(a.st_in)[2 * pr]; rr = as_const(a.st_in)[2 * pr + 1];
                cnt = (rr + 1) - l;
                bs_step(ix, (uint64_t)(last + a.shift), l, rr, l, rr, &probes);
                model += ix.levels * (uint32_t)(ff.n + (a.cur_len - 1));    // the reference re-searches the whole prefix
            } else {
                const uint64_t total = ff.n + (a.cur_len - 1);
                for (uint64_t t = 0; t < total; t++) {
                    if (t + 1 == total) cnt = (rr + 1) - l;
                    const int64_t tok = t < ff.n ? ff.tok[t] : sent[1 + (t - ff.n)];
                    bs_step(ix, (uint64_t)(tok + a.shift), l, rr, l, rr, &probes);
                    model += ix.levels;
                }
                if (total == 0) cnt = (rr + 1) - l;
            }
            if (write_state && a.st_out) { a.st_out[2 * r] = l; a.st_out[2 * r + 1] = rr; }
            if (!dead) { lo = l; hi = rr + 1; count = cnt; }
        }
    }
    single = -1;
    expand = false;
    if (!valid) {}
    else if (a.grp_stop[grp] > 0 && (int64_t)count <= a.grp_stop[grp]) single = eos_id;
    else if (dead) single = a.pad_id;
    else { expand = true; if (hi > ix.n) hi = ix.n; }
}

// Row-first call, first launch: ONE wave per row runs the row's chain (13 x fewer waves than when every (row, top digit) wave
// of k_constrain repeats it) and splits the root node over all sixteen digits with one single-digit rank per lane.
__global__ __launch_bounds__(256) void k_constrain_rows(FmiDev ix, ConstrainArgs a)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= a.rows) return;
    uint64_t lo, hi, probes = 0;
    uint32_t model = 0;
    int64_t single, eos_id;
    bool expand;
    row_range_and_class(ix, a, r, true, lane == 0, lo, hi, single, expand, eos_id, probes, model);
    const bool split = expand && hi > lo;
    const uint32_t e = lane & 1, d = lane >> 1;
    uint64_t q = 0;
    if (split && lane < 32) q = wm_step(ix, 0, e ? hi : lo, d);
    const uint64_t qo = (uint64_t)dpp_xor1((uint32_t)q) | ((uint64_t)dpp_xor1((uint32_t)(q >> 32)) << 32);
    const uint64_t bal = __ballot(lane < 32 && e == 0 && qo > q);
    if (lane < 32) a.pre_child[((uint64_t)r * FMI_ARITY + d) * 2 + e] = q;
    uint32_t em = 0;
#pragma unroll
    for (uint32_t x = 0; x < 16; x++) em |= (uint32_t)((bal >> (2 * x)) & 1ull) << x;
    if (lane == 0) {
        RowPre p;
        p.lo = lo; p.hi = hi; p.single = single; p.expand = expand ? 1u : 0u; p.child_mask = em;
        a.pre_rows[r] = p;
    }
    if (a.probe_counter) {
        ExpCounters ctr{0, 0, 0, 0};
        if (lane == 0) {
            ctr.probes = (uint32_t)probes + (split ? ((lo >> FMI_BLOCK_SHIFT) != (hi >> FMI_BLOCK_SHIFT) ? 2u : 1u) : 0u);
            ctr.model = model + (split ? model_nodes(em, 0, FMI_DIGIT_BITS * ix.dlevels - ix.levels) : 0u);
        }
        flush_counters(a.probe_counter, ctr);
    }
}

template <bool SB, int W>
__global__ __launch_bounds__(64 * W, W > 1 ? 4 : 1) void k_constrain(FmiDev ix, ConstrainArgs a)
{
    extern __shared__ uint4 s_dyn[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform, and the compiler knows
    const uint32_t D = ix.dlevels;
    const uint32_t sub_bits = FMI_DIGIT_BITS * (D - 1);        // symbol bits below the top digit
    const uint32_t nsym = 1u << sub_bits;                       // symbols of one item's sub-tree
    const uint32_t nw = (nsym + 31) >> 5;
    const uint32_t bm_slots = constrain_bm_slots(D);
    // W == 1: frontier, counters, bitmap of the wave;  W > 1: see above
    uint4 *s_node = s_dyn;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(W > 1 ? s_dyn : s_dyn + exp_slots((int)D - 1));
    uint4 *s_bitmaps = W > 1 ? s_dyn + 2 : s_dyn + exp_slots((int)D - 1) + 2;
    uint4 *s_lvl = s_bitmaps + W * bm_slots;                    // W > 1 only
    uint32_t *s_bits = reinterpret_cast<uint32_t *>(s_bitmaps + (size_t)wave * bm_slots);
    uint32_t d1, r;
    if constexpr (W == 1) {
        // rows are padded to a multiple of 8 in the grid: workgroup i runs on XCD i % 8, so every wave of a row lands on
        // the same XCD and the row's own probes (prefix step, root) miss its L2 once instead of once per XCD
        const uint32_t rows8 = (a.rows + 7) & ~7u;
        d1 = blockIdx.x / rows8; r = blockIdx.x - d1 * rows8;
    } else {
        // workgroups in row-group-major order -- the top digits of one group of W rows follow each other in the grid (round 6; top-digit-major
        // before): a batch's calls 344 -> 322 us on one box (profiles/r6_constrain_ab_item_order.txt); striding a workgroup's rows over the
        // call instead of taking W consecutive ones changed nothing
        const uint32_t g = blockIdx.x / a.ndig0;
        d1 = blockIdx.x - g * a.ndig0; r = g * W + wave;
    }
    const uint32_t slot = blockIdx.x * W + wave;               // of the debug stamps
    const bool writer = d1 == 0;
    const bool counting = a.probe_counter != nullptr;
    ExpCounters ctr{0, 0, 0, 0};
#define STAMP(i) do { if (a.tstamp && lane == 0) a.tstamp[(uint64_t)slot * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    STAMP(0);

    // housekeeping for the next call: this workgroup's share of the other bitmap buffer
    if (a.clear) {
        const uint64_t per = (a.clear_words + gridDim.x - 1) / gridDim.x;
        const uint64_t w0 = (uint64_t)blockIdx.x * per;
        for (uint64_t w = w0 + threadIdx.x; w < w0 + per && w < a.clear_words; w += 64 * W) a.clear[w] = 0u;
    }
    const bool valid = r < a.rows;      // padding wave: its share of the clearing (and of the workgroup's levels) is all it does
    if (W == 1 && !valid) return;
    if constexpr (W > 1) {
        // A row-first / chained call knows whether this (row, top digit) item has any work before it touches anything: the row's record
        // says what the row needs (a special token, an expansion, nothing: ROW_DONE) and which of its root node's children exist.  Most
        // items of a decode step are empty -- all of them once every row is in list mode -- and such a wave ends HERE, before the LDS
        // initialisation and the workgroup's first barrier (round 5: an all-empty 600-row call cost 21 us of workgroups that zeroed
        // their bitmaps and met at a barrier only to leave).  Measurement modes keep every wave.
        if (a.pre_rows != nullptr && a.leave_early && a.probe_counter == nullptr && a.tstamp == nullptr) {
            bool work = false;
            if (valid) {
                const cptr<RowPre> p = as_const(a.pre_rows) + r;
                const int64_t sg = p->single;
                // (a special token is set by the wave that owns its symbol, or by the top digit 0 wave when no wave does: set_special)
                auto sets = [&](int64_t tok) {
                    const int64_t sym = tok + a.shift;
                    const bool owned = sym >= 0 && (uint64_t)(sym >> sub_bits) < a.ndig0;
                    return tok >= 0 && (uint64_t)tok < a.vocab && (owned ? (uint32_t)(sym >> sub_bits) == d1 : d1 == 0);
                };
                work = (sg >= 0 && sets(sg)) || (a.always_allow_eos && sg != -2 && sets(a.grp_eos[row_group(a, r)]));
                if (p->expand != 0 && p->hi > p->lo) work = work || ((p->child_mask >> d1) & 1u) != 0;
            }
            if (!__builtin_amdgcn_readfirstlane((int)work)) return;
        }
    }
    for (uint32_t w = lane; w < nw; w += 64) s_bits[w] = 0u;
    if (W > 1 && lane < 8) s_cnt[lane] = 0u;       // (by every wave that stays: the one that used to do it may have left)

    // ---- the row: prefix range, class (identical in every wave of the row) ----
    uint64_t lo = 0, hi = 0, probes = 0;
    uint32_t model = 0;      // in nodes of the binary model: one backward-search step = `levels` nodes (2 L probes)
    int64_t single = -1, eos_id = 0;
    bool expand = false;
    const bool pre = W > 1 && a.pre_rows != nullptr;      // row-first call: k_constrain_rows has done the rows
    if (pre) {
        eos_id = a.grp_eos[row_group(a, r)];
        if (valid) {
            const cptr<RowPre> p = as_const(a.pre_rows) + r;
            lo = p->lo; hi = p->hi; single = p->single; expand = p->expand != 0;
        }
    } else {
        row_range_and_class(ix, a, r, valid, writer && lane == 0, lo, hi, single, expand, eos_id, probes, model);
    }
    STAMP(1);
    if constexpr (W > 1) __syncthreads(); else wave_sync();            // bitmaps and counters zeroed
    // ---- child d1 of the row's root node, then its sub-tree ----
    bool live = false;                  // this wave's item has a sub-tree
    if (expand && hi > lo) {
        uint64_t clo, chi;
        if (pre) {
            const cptr<uint64_t> c = as_const(a.pre_child) + ((uint64_t)r * FMI_ARITY + d1) * 2;
            clo = c[0]; chi = c[1];
        } else {
            root_child(ix, lo, hi, d1, clo, chi);
        }
        live = chi > clo;
        if (a.tstamp && lane == 0) a.tstamp[(uint64_t)slot * 8 + 2] = chi > clo ? __builtin_amdgcn_s_memrealtime() : 0;
        if (counting && writer && lane == 0 && !pre) {
            probes += (lo >> FMI_BLOCK_SHIFT) != (hi >> FMI_BLOCK_SHIFT) ? 2 : 1;
            model += model_nodes(root_children_mask(ix, lo, hi), 0, FMI_DIGIT_BITS * D - ix.levels);
        }
        if (chi > clo) {
            if (D == 1) { if (lane == 0 && d1 != 0) s_bits[0] |= 1u; }
            else if constexpr (W > 1) {
                if (D == 2) {
                    // level 1 is the leaf level: the workgroup serves the W root children together
                    if (lane == 0) {
                        uint4 c = pack_node(clo, chi, d1);
                        c.w = wave;
                        s_lvl[atomicAdd(&s_cnt[0], 1u)] = c;
                    }
                } else {
                    // level 1 = ONE node per item: lane (d2, end) of the first 32 takes one single-digit rank -- a
                    // sixth of the instructions of a wave iteration that computes sixteen ranks per lane for one
                    // node -- and no barrier in front of it; the children go to the workgroup's level-2 array
                    const uint32_t e = lane & 1, d2 = lane >> 1;
                    uint64_t q = 0;
                    if (lane < 32) q = wm_step(ix, 1, e ? chi : clo, d2);
                    const uint64_t qo = (uint64_t)dpp_xor1((uint32_t)q) | ((uint64_t)dpp_xor1((uint32_t)(q >> 32)) << 32);
                    const uint64_t c_lo = e ? qo : q, c_hi = e ? q : qo;
                    const bool ex = lane < 32 && e == 0 && c_hi > c_lo;
                    const uint64_t bal = __ballot(ex);
                    const uint32_t nch = (uint32_t)__popcll(bal);
                    if (nch) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_cnt[1], nch);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (ex) {
                            uint4 c = pack_node(c_lo, c_hi, (d1 << 4) | d2);
                            c.w = wave;
                            s_lvl[constrain_lvl_off(W, 1) + base + lane_rank_in(bal)] = c;
                        }
                    }
                    if (counting && lane == 0) {
                        uint32_t em = 0;
#pragma unroll
                        for (uint32_t d = 0; d < 16; d++) em |= (uint32_t)((bal >> (2 * d)) & 1ull) << d;
                        ctr.model += model_nodes(em, 1, FMI_DIGIT_BITS * D - ix.levels);
                        ctr.probes += (clo >> FMI_BLOCK_SHIFT) != (chi >> FMI_BLOCK_SHIFT) ? 2 : 1;
                        ctr.iters++; ctr.nodes++;
                    }
                }
            } else {
                expand_subtree<EMIT_BITS, SB>(ix, s_node, s_cnt, reinterpret_cast<uint8_t *>(s_bits), r, 1, clo, chi, d1,
                                              EmitTarget{}, counting, ctr);
            }
        }
    }
    if constexpr (W > 1) {
        // A wave whose item is empty -- most of them once the prefixes are a few tokens long: a row's symbols then sit below
        // one or two of the top digits -- has nothing to expand, nothing to store (the bitmap is zero) and leaves now, unless
        // its row needs a special token set: its registers go back to the CU at once, so that the other workgroups of the
        // launch become resident (at two workgroups of eight 126-register waves per CU a 600-row call otherwise runs in two
        // rounds).  The barriers below count the waves that are left (s_barrier waits on the surviving waves only).
        // Measurement modes keep every wave (the counters are flushed at the end).
        // (single == ROW_DONE: k_beam_advance has put the row's tokens into the bitmap already)
        const bool stays = live || (valid && (single >= 0 || (a.always_allow_eos && single != -2))) || counting;
        // (keeping the empty waves as helpers where a workgroup has much to share measured the same on the bench workload: 39.0 vs 39.4 us
        //  per call, option leave_early = 0; on 600 narrow rows leaving is what lets the second launch of a row-first call finish in 12 us)
        if (a.leave_early && !__builtin_amdgcn_readfirstlane((int)stays)) {     // (a scalar condition: the whole wave branches to its end)
            STAMP(3); STAMP(4);
            return;
        }
        if (lane == 0) atomicOr(&s_cnt[7], 1u << wave);
        uint32_t my_rank = 0, n_live = 1;
        // levels 2 .. D-1 (dlevels 2: level 1) of the W sub-trees, the workgroup's remaining waves together
        for (uint32_t j = D == 2 ? 0 : 1; j + 1 < D; j++) {
            __syncthreads();        // level j complete (the first one: by every wave on its own)
            if (j == D - 2) STAMP(5);
            const uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[7]);
            my_rank = (uint32_t)__popc(mask & ((1u << wave) - 1)); n_live = (uint32_t)__popc(mask);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[j]);
            wg_level<SB, W>(ix, s_lvl + constrain_lvl_off(W, j), total, s_lvl + constrain_lvl_off(W, j + 1), &s_cnt[j + 1], 1 + j,
                            s_bitmaps, bm_slots, counting, ctr, my_rank, n_live);
        }
        __syncthreads();            // the leaf bits other waves found for my item
    }
    wave_sync();
    STAMP(3);
    if (valid) {
        // pad / eos of the row's class: after the expansion, whose byte stores would overwrite them
        if (lane == 0) {
            if (single >= 0) set_special(ix, a, s_bits, r, d1, sub_bits, single);
            if (a.always_allow_eos && single != -2) set_special(ix, a, s_bits, r, d1, sub_bits, eos_id);
        }
        wave_sync();
        EmitTarget tgt{};
        tgt.bits = a.bits; tgt.words_per_row = a.words_per_row; tgt.shift = a.shift; tgt.vocab = a.vocab;
        flush_leaf_bits(tgt, r, s_bits, d1 << sub_bits, nsym);
    }
    STAMP(4);
#undef STAMP
    if (counting) {
        if (writer && lane == 0) { ctr.probes += (uint32_t)probes; ctr.model += model; }
        flush_counters(a.probe_counter, ctr);
    }
}
// out = allowed ? in : -inf     (scores + mask with mask in {0, -inf}, beam_search.py:64,140)
__global__ __launch_bounds__(256) void k_apply_bits(const float *in, float *out, const uint32_t *bits, uint64_t rows,
                                                    uint64_t vocab, uint64_t words_per_row)
{
    const uint64_t row = blockIdx.y;
    const float ninf = -__builtin_huge_valf();
    const float *src = in + row * vocab;
    float *dst = out + row * vocab;
    const uint32_t *b = bits + row * words_per_row;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < vocab; v += (uint64_t)gridDim.x * blockDim.x)
        dst[v] = ((b[v >> 5] >> (v & 31)) & 1) ? src[v] : ninf;
}

// ---------------------------------------------------------------------------
// Fused constrained top-2K of one decode step (reference beam_search.py:244-310):
//   next_token_scores = log_softmax(logits); InfNanRemove; unconstrained = + beam_score;
//   constrained = unconstrained where the token is allowed, else -inf; top-2K of the constrained
//   scores over the K*V candidates of a query; carry the UNCONSTRAINED score of the picks.
// Nothing of shape [rows, vocab] is written: k_row_pick streams each row's logits (statistics, then the
// best allowed tokens of the row under the bitmap of k_constrain), k_query_merge merges the K per-row
// lists of a query.  Ties go to the lower flat index.
// ---------------------------------------------------------------------------
static constexpr int TOPK_MAX = 64;          // 2 * num_beams <= 64

__device__ __forceinline__ float logp_processed(float x, float mx, float lsum)
{
    float lp = (x - mx) - lsum;                                  // log_softmax
    if (lp != lp) lp = 0.0f;                                     // InfNanRemoveLogitsProcessor (HF 4.13): nan -> 0
    if (lp == __builtin_huge_valf()) lp = 3.402823466e+38f;      // +inf -> finfo.max
    return lp;
}

// order-preserving map float -> uint32 (ascending)
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One workgroup of 512 threads per row: log-softmax statistics of the row and its `want` (<= 64) best
// ALLOWED tokens by processed log-prob (descending, ties to the lower token id) -> row_tok / row_lp
// [rows, want]; row_cnt = how many exist.  The row (200 KB at BART's vocabulary) is streamed from L2 with
// coalesced loads, a few registers per thread, so that all rows of a decode step are resident at once:
//   sweep 1      max and log(sum(exp(x - max))) in one pass (running maximum), and every thread's best allowed
//                logit on the side                                                           (every row)
//   narrow rows  (<= 1024 allowed tokens): the allowed tokens are gathered by walking the bitmap and ranked
//                in LDS -- nothing else is read;
//   wide rows    a lower bound of the want-th best allowed log-prob from the threads' best logits (log-softmax
//                is monotone in the logit): sweep 2 collects the allowed tokens that reach it -- a few dozen on
//                any real distribution -- and they are ranked in LDS.  No histogram, no atomics beyond the
//                list counter.  Only if more than PICK_CAP keys pass (mass ties) the exact radix select runs:
//                four 8-bit histogram passes pin the want-th largest key, a bisection on the token id finds the
//                lowest tokens among its ties.
static constexpr int PICK_BLOCK = 512;
static constexpr int PICK_WAVES = PICK_BLOCK / 64;
static constexpr int PICK_CAP = 1024;        // candidate list in LDS; also the widest "narrow" row
static constexpr int TOPK_NARROW = PICK_CAP;
enum { PICK_NO_PREFILTER = 1 };              // test switch: wide rows go straight to the radix select

__device__ __forceinline__ bool bm_bit(const uint32_t *s_bm, uint32_t tok) { return (s_bm[tok >> 5] >> (tok & 31)) & 1u; }

// f(token, logit) for every token of a row, PICK_BLOCK threads.  The loads are unconditional and wide:
// a scalar head up to the first 16-byte boundary (rows are vocab floats apart, vocab odd for BART), an
// aligned float4 body -- 16 bytes per lane per load, several loads in flight per thread: one CU has to
// pull a 200 KB row out of L2 / MALL in a few microseconds, which is a matter of bytes in flight --, a
// scalar tail.  Whatever f tests (bitmap bits) is applied to values that are already on their way.
template <typename F>
__device__ __forceinline__ void row_sweep(const float *x, uint32_t vocab, F &&f)
{
    const uint32_t tid = threadIdx.x;
    uint32_t head = (4u - (uint32_t)((reinterpret_cast<uintptr_t>(x) >> 2) & 3u)) & 3u;
    if (head > vocab) head = vocab;
    if (tid < head) f(tid, x[tid]);
    const float4 *x4 = reinterpret_cast<const float4 *>(x + head);
    const uint32_t n4 = (vocab - head) >> 2;
#pragma unroll 8
    for (uint32_t i = tid; i < n4; i += PICK_BLOCK) {
        const float4 v = x4[i];
        const uint32_t t = head + 4 * i;
        f(t, v.x); f(t + 1, v.y); f(t + 2, v.z); f(t + 3, v.w);
    }
    const uint32_t t = head + 4 * n4 + tid;
    if (t < vocab) f(t, x[t]);
}

// r-th largest of the 64 values of a wave (r >= 1, duplicates count separately), -inf if there are fewer
__device__ __forceinline__ float wave_rth_largest(float v, uint32_t r)
{
    const uint32_t lane = threadIdx.x & 63;
    float m = v;
    for (uint32_t k = 0; k < r; k++) {
        m = v;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const uint64_t at = __ballot(v == m);
        if (at && lane == (uint32_t)__builtin_ctzll(at)) v = -__builtin_huge_valf();     // take one holder of the maximum out
    }
    return m;
}

// A lower bound of the want-th largest value among the values the threads of the workgroup have seen, from
// the threads' own maxima: every wave contributes its ceil(want / waves)-th largest thread maximum, the
// smallest of those is returned -- at least `want` thread maxima, hence at least `want` values, are >= it.
// (-inf when some wave has too few threads with a value: then everything passes.)
__device__ __forceinline__ float block_lower_bound(float tmax, uint32_t want, float *s_w /* PICK_WAVES floats */)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float mine = wave_rth_largest(tmax, (want + PICK_WAVES - 1) / PICK_WAVES);
    __syncthreads();
    if (lane == 0) s_w[wv] = mine;
    __syncthreads();
    float b = s_w[0];
    for (int i = 1; i < PICK_WAVES; i++) b = fminf(b, s_w[i]);
    return b;
}

// the same walk with the aligned body handed over four logits at a time
template <typename F1, typename F4>
__device__ __forceinline__ void row_sweep_chunks(const float *x, uint32_t vocab, F1 &&f1, F4 &&f4)
{
    const uint32_t tid = threadIdx.x;
    uint32_t head = (4u - (uint32_t)((reinterpret_cast<uintptr_t>(x) >> 2) & 3u)) & 3u;
    if (head > vocab) head = vocab;
    if (tid < head) f1(tid, x[tid]);
    const float4 *x4 = reinterpret_cast<const float4 *>(x + head);
    const uint32_t n4 = (vocab - head) >> 2;
#pragma unroll 8
    for (uint32_t i = tid; i < n4; i += PICK_BLOCK) f4(head + 4 * i, x4[i]);
    const uint32_t t = head + 4 * n4 + tid;
    if (t < vocab) f1(t, x[t]);
}

// the n_list candidates in s_ctok / s_cval -> their places under (value descending, token ascending); the first k_sel leave
__device__ __forceinline__ void pick_rank_and_store(const int32_t *s_ctok, const float *s_cval, uint32_t n_list, uint32_t k_sel,
                                                    uint32_t row, uint32_t want, int32_t *row_tok, float *row_lp)
{
    for (uint32_t i = threadIdx.x; i < n_list; i += PICK_BLOCK) {
        const float v = s_cval[i]; const int32_t t = s_ctok[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_list; j++) {
            const float ov = s_cval[j]; const int32_t ot = s_ctok[j];
            rank += (ov > v) || (ov == v && ot < t);
        }
        if (rank < k_sel) { row_tok[(uint64_t)row * want + rank] = t; row_lp[(uint64_t)row * want + rank] = v; }
    }
}

// n candidates in LDS (n <= PICK_CAP) -> the k_sel best written out.  Ranking is quadratic, so a long list is
// first cut down with the same lower bound (over the list entries, two per thread) to a second, short list.
static constexpr uint32_t PICK_DIRECT = 128;     // lists up to this length are ranked as they are
static constexpr uint32_t PICK_SHORT = 256;      // capacity of the short list
__device__ __forceinline__ void pick_finish(const int32_t *s_ctok, const float *s_cval, int32_t *s_tok2, float *s_val2, float *s_w,
                                            uint32_t *s_n2, uint32_t n, uint32_t k_sel, uint32_t row, uint32_t want,
                                            int32_t *row_tok, float *row_lp)
{
    const uint32_t tid = threadIdx.x;
    if (n > PICK_DIRECT && k_sel == want) {
        const float ninf = -__builtin_huge_valf();
        const float a = tid < n ? s_cval[tid] : ninf, b = tid + PICK_BLOCK < n ? s_cval[tid + PICK_BLOCK] : ninf;
        const float bound = block_lower_bound(fmaxf(a, b), want, s_w);
        if (tid == 0) *s_n2 = 0;
        __syncthreads();
        if (tid < n && a >= bound) { const uint32_t o = atomicAdd(s_n2, 1u); if (o < PICK_SHORT) { s_tok2[o] = s_ctok[tid]; s_val2[o] = a; } }
        if (tid + PICK_BLOCK < n && b >= bound) { const uint32_t o = atomicAdd(s_n2, 1u); if (o < PICK_SHORT) { s_tok2[o] = s_ctok[tid + PICK_BLOCK]; s_val2[o] = b; } }
        __syncthreads();
        const uint32_t n2 = *s_n2;
        if (n2 <= PICK_SHORT) { pick_rank_and_store(s_tok2, s_val2, n2, k_sel, row, want, row_tok, row_lp); return; }
    }
    pick_rank_and_store(s_ctok, s_cval, n, k_sel, row, want, row_tok, row_lp);
}

__global__ __launch_bounds__(PICK_BLOCK) void k_row_pick(const float *logits, const uint32_t *bits, uint64_t words_per_row,
                                                         uint32_t row_broadcast_bits, uint64_t vocab64, uint32_t want,
                                                         float *row_max, float *row_lsum, int32_t *row_tok, float *row_lp,
                                                         uint32_t *row_cnt, uint32_t narrow_max, uint32_t flags)
{
    extern __shared__ uint32_t s_bm[];               // the row's bitmap, words_per_row words + one zero word
    __shared__ int32_t s_ctok[PICK_CAP];
    __shared__ float s_cval[PICK_CAP];
    __shared__ int32_t s_tok2[PICK_SHORT];
    __shared__ float s_val2[PICK_SHORT];
    __shared__ uint32_t s_hist[256];
    __shared__ float s_a[PICK_WAVES], s_b[PICK_WAVES], s_w[PICK_WAVES];
    __shared__ uint32_t s_scan[PICK_WAVES];
    __shared__ float s_ls;
    __shared__ uint32_t s_n, s_n2, s_prefix, s_remaining, s_tie_rank;
    const uint32_t row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t vocab = (uint32_t)vocab64, wpr = (uint32_t)words_per_row;
    const float *x = logits + (uint64_t)row * vocab64;
    const float ninf = -__builtin_huge_valf();
    // ---- the row's bitmap -> LDS (tokens >= vocab do not exist); allowed tokens per thread, scanned ----
    uint32_t my_allowed = 0;
    {
        const uint32_t *b = bits + (row_broadcast_bits ? 0 : (uint64_t)row * words_per_row);
        for (uint32_t w = tid; w < wpr; w += PICK_BLOCK) {
            uint32_t word = 32 * w < vocab ? b[w] : 0u;
            if (32 * w + 32 > vocab && 32 * w < vocab) word &= (1u << (vocab - 32 * w)) - 1;
            s_bm[w] = word;
            my_allowed += (uint32_t)__popc(word);
        }
    }
    if (tid == 0) { s_n = 0; s_prefix = 0; s_bm[wpr] = 0u; }
    uint32_t incl = my_allowed;                      // inclusive scan over the workgroup (slots of the narrow-row gather)
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += v; }
    if (lane == 63) s_scan[wv] = incl;
    __syncthreads();
    uint32_t before = incl - my_allowed, total = 0;
    for (uint32_t w = 0; w < (uint32_t)PICK_WAVES; w++) { if (w < wv) before += s_scan[w]; total += s_scan[w]; }
    // ---- sweep 1: max and sum(exp(x - max)) in ONE pass (running maximum, the sum rescaled when it moves: once
    // per float4 at most, rarely after the first few).  Five instructions per logit: the pass has to stay
    // memory-bound (300 rows x 50 265 logits per decode step).  The running maximum starts at -FLT_MAX, so that
    // a -inf logit adds exp(-inf) = 0 without a special case; NaN / +inf logits make the sum NaN, which is what
    // marks the row (as log_softmax would).  Wide rows also track the thread's best ALLOWED logit (log-softmax
    // is monotone in the logit); narrow rows do not need it.
    const bool wide = total > (narrow_max > want ? narrow_max : want);
    float mx = -3.402823466e+38f, sum = 0.f, tbest = ninf;
    auto one = [&](uint32_t v, float a) {
        if (a > mx) { sum *= __expf(mx - a); mx = a; }
        sum += __expf(a - mx);
        if (wide) tbest = bm_bit(s_bm, v) ? fmaxf(tbest, a) : tbest;
    };
    row_sweep_chunks(x, vocab, one,
        [&](uint32_t v, float4 q) {
            const float m4 = fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w));
            if (m4 > mx) { sum *= __expf(mx - m4); mx = m4; }
            sum += (__expf(q.x - mx) + __expf(q.y - mx)) + (__expf(q.z - mx) + __expf(q.w - mx));
            if (wide) {
                // the four bitmap bits of tokens v .. v+3 (they may straddle two words; s_bm has a spare zero word)
                const uint32_t w0 = s_bm[v >> 5], w1 = s_bm[(v >> 5) + 1];
                const uint32_t b4 = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (v & 31));
                tbest = (b4 & 1) ? fmaxf(tbest, q.x) : tbest;
                tbest = (b4 & 2) ? fmaxf(tbest, q.y) : tbest;
                tbest = (b4 & 4) ? fmaxf(tbest, q.z) : tbest;
                tbest = (b4 & 8) ? fmaxf(tbest, q.w) : tbest;
            }
        });
    // (max, sum) pairs combine as  M = max(m1, m2),  S = s1 exp(m1 - M) + s2 exp(m2 - M)
    auto combine = [&](float &m1, float &s1, float m2, float s2) {
        const float M = fmaxf(m1, m2);
        s1 = s1 * __expf(m1 - M) + s2 * __expf(m2 - M);
        m1 = M;
    };
    for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(mx, o), s2 = __shfl_xor(sum, o); combine(mx, sum, m2, s2); }
    if (lane == 0) { s_a[wv] = mx; s_b[wv] = sum; }
    __syncthreads();
    mx = s_a[0]; sum = s_b[0];
    for (int i = 1; i < PICK_WAVES; i++) combine(mx, sum, s_a[i], s_b[i]);
    const bool row_nan = sum != sum;             // a NaN or +inf logit
    if (tid == 0) {
        const float qnan = __builtin_nanf("");
        const float l = row_nan ? qnan : logf(sum);
        row_max[row] = row_nan ? qnan : mx;
        row_lsum[row] = l;
        s_ls = l;
    }
    __syncthreads();
    const float ls = s_ls;
    if (row_nan) mx = __builtin_nanf("");
    const uint32_t k_sel = total < want ? total : want;
    if (k_sel == 0) { if (tid == 0) row_cnt[row] = 0; return; }
    // ---- narrow row: walk the bitmap into the list (slots from the scan, no atomics), then pick ----
    if (total <= (narrow_max > want ? narrow_max : want)) {
        uint32_t o = before;
        for (uint32_t w = tid; w < wpr; w += PICK_BLOCK) {
            uint32_t word = s_bm[w];
            while (word) {
                const uint32_t tok = 32 * w + (uint32_t)__builtin_ctz(word);
                word &= word - 1;
                s_ctok[o] = (int32_t)tok; s_cval[o] = logp_processed(x[tok], mx, ls);
                o++;
            }
        }
        __syncthreads();
        pick_finish(s_ctok, s_cval, s_tok2, s_val2, s_w, &s_n2, total, k_sel, row, want, row_tok, row_lp);
        if (tid == 0) row_cnt[row] = k_sel;
        return;
    }
    // ---- wide row: sweep 3 collects what reaches the lower bound of the thread maxima ----
    if (!(flags & PICK_NO_PREFILTER)) {
        const float xb = block_lower_bound(tbest, want, s_w);
        const float lpb = xb == ninf ? ninf : logp_processed(xb, mx, ls);      // at least `want` allowed tokens have lp >= lpb
        // a logit below xb can still round to lp == lpb; anything below xb - margin cannot (the margin is far above
        // the two roundings of (x - max) - lsum), so one compare on the raw logit rejects almost everything
        const float xcut = (xb == ninf || lpb != lpb) ? ninf : xb - (1e-3f + (fabsf(xb) + fabsf(mx) + fabsf(ls)) * 1e-6f);
        auto take = [&](uint32_t v, float a) {
            const float lp = logp_processed(a, mx, ls);
            if (lp >= lpb) {
                const uint32_t o = atomicAdd(&s_n, 1u);
                if (o < (uint32_t)PICK_CAP) { s_ctok[o] = (int32_t)v; s_cval[o] = lp; }
            }
        };
        row_sweep_chunks(x, vocab,
            [&](uint32_t v, float a) { if (bm_bit(s_bm, v) && !(a < xcut)) take(v, a); },
            [&](uint32_t v, float4 q) {
                const uint32_t w0 = s_bm[v >> 5], w1 = s_bm[(v >> 5) + 1];
                const uint32_t b4 = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (v & 31)) & 15u;
                if (!b4) return;
                if ((b4 & 1) && !(q.x < xcut)) take(v, q.x);
                if ((b4 & 2) && !(q.y < xcut)) take(v + 1, q.y);
                if ((b4 & 4) && !(q.z < xcut)) take(v + 2, q.z);
                if ((b4 & 8) && !(q.w < xcut)) take(v + 3, q.w);
            });
        __syncthreads();
        const uint32_t n = s_n;
        if (n <= (uint32_t)PICK_CAP) {
            pick_finish(s_ctok, s_cval, s_tok2, s_val2, s_w, &s_n2, n, k_sel, row, want, row_tok, row_lp);
            if (tid == 0) row_cnt[row] = k_sel;
            return;
        }
        __syncthreads();
        if (tid == 0) s_n = 0;
    }
    // ---- exact radix select (mass ties): 4 x 8 bits pin the want-th largest key T ----
    if (tid == 0) s_remaining = want;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t pmask = pass == 0 ? 0u : (~0u << (shift + 8));
        row_sweep(x, vocab, [&](uint32_t v, float a) {
            const uint32_t key = float_key(logp_processed(a, mx, ls));
            if (bm_bit(s_bm, v) && (key & pmask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255], 1u);
        });
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = s_remaining, bin = 255;
            for (;; bin--) {
                const uint32_t c = s_hist[bin];
                if (c >= rem || bin == 0) break;
                rem -= c;
            }
            s_remaining = rem;                 // rank of the target inside the chosen bin
            s_prefix = prefix | (bin << shift);
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix;
    const uint32_t need_eq = s_remaining;       // ties with T still needed: the need_eq LOWEST tokens among them
    // the token below which exactly need_eq ties lie, by bisection on the token id (the count of ties below m is
    // monotone in m; every probe is one sweep -- this path only runs on mass ties)
    uint32_t lo_t = 0, hi_t = vocab;            // smallest m with count(ties < m) >= need_eq
    while (lo_t < hi_t) {
        const uint32_t mid = lo_t + ((hi_t - lo_t) >> 1);
        if (tid == 0) s_tie_rank = 0;
        __syncthreads();
        uint32_t c = 0;
        row_sweep(x, vocab, [&](uint32_t v, float a) {
            c += (v < mid && bm_bit(s_bm, v) && float_key(logp_processed(a, mx, ls)) == T) ? 1u : 0u;
        });
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
        if (lane == 0 && c) atomicAdd(&s_tie_rank, c);
        __syncthreads();
        const uint32_t below = s_tie_rank;
        __syncthreads();
        if (below >= need_eq) hi_t = mid; else lo_t = mid + 1;
    }
    // lo_t = one past the need_eq-th lowest tie.  Collect keys > T and ties with token < lo_t: exactly k_sel entries
    row_sweep(x, vocab, [&](uint32_t v, float a) {
        const float lp = logp_processed(a, mx, ls);
        const uint32_t key = float_key(lp);
        if (bm_bit(s_bm, v) && (key > T || (key == T && v < lo_t))) {
            const uint32_t o = atomicAdd(&s_n, 1u);
            if (o < (uint32_t)PICK_CAP) { s_ctok[o] = (int32_t)v; s_cval[o] = lp; }
        }
    });
    __syncthreads();
    pick_rank_and_store(s_ctok, s_cval, k_sel, k_sel, row, want, row_tok, row_lp);
    if (tid == 0) row_cnt[row] = k_sel;
}

// One workgroup per query: merge the K per-row lists.  The lists (with the beam scores added) are staged in
// LDS; every candidate finds its place in the merged order by itself -- its own position in its list plus, for
// every other list, the number of entries that beat it (a binary search: the lists are sorted) -- and the
// first `want` write themselves out.  If a query has fewer than `want` finite candidates the rest is filled
// with not-allowed tokens (constrained score -inf, as torch.topk would).
static constexpr int MERGE_BLOCK = 256;
__global__ __launch_bounds__(MERGE_BLOCK) void k_query_merge(const float *logits, const uint32_t *bits, uint64_t words_per_row,
                                                             uint32_t row_broadcast_bits, uint64_t vocab, uint32_t beams, uint32_t want,
                                                             const float *beam_scores, const float *row_max, const float *row_lsum,
                                                             const int32_t *row_tok, const float *row_lp, const uint32_t *row_cnt,
                                                             int64_t *top_idx, float *top_con, float *top_unc)
{
    __shared__ float s_val[32 * TOPK_MAX];
    __shared__ int32_t s_tok[32 * TOPK_MAX];
    __shared__ uint32_t s_cnt[32];
    __shared__ uint32_t s_fill[MERGE_BLOCK / 64];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n_all = beams * want;
    const float qnan = __builtin_nanf("");          // padding: compares false with everything, so it is never `before` anything
    if (tid < beams) s_cnt[tid] = row_cnt[q * beams + tid];
    for (uint32_t i = tid; i < n_all; i += MERGE_BLOCK) {
        const uint32_t bm = i / want, j = i - bm * want, r = q * beams + bm;
        // every load is issued before any is needed; entries past the end of a list become NaN: they beat
        // nothing (not even a -inf candidate), so the searches below need no list lengths
        const uint32_t c = row_cnt[r];
        const float lp = row_lp[(uint64_t)r * want + j], bs = beam_scores[r];
        const int32_t tk = row_tok[(uint64_t)r * want + j];
        s_val[i] = j < c ? lp + bs : qnan;
        s_tok[i] = j < c ? tk : 0x7fffffff;
    }
    __syncthreads();
    uint32_t n_cand = 0;
    for (uint32_t bm = 0; bm < beams; bm++) n_cand += s_cnt[bm];
    for (uint32_t i = tid; i < n_all; i += MERGE_BLOCK) {
        const uint32_t bm = i / want, j = i - bm * want;
        if (j >= s_cnt[bm]) continue;
        const float v = s_val[i]; const int32_t t = s_tok[i];
        uint32_t rank = j;                                   // its own list is strictly ordered
#pragma unroll 4
        for (uint32_t ob = 0; ob < beams; ob++) {
            // entries of list ob that come before (v, bm, t): value descending, then the lower flat index; a
            // branch-free lower bound over the padded list, so that the searches of several lists overlap
            const float *lv = s_val + ob * want;
            const int32_t *lt = s_tok + ob * want;
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t step = 64; step; step >>= 1) {
                const uint32_t probe = pos + step - 1;
                const bool in = pos + step <= want;
                const float ov = lv[in ? probe : 0];
                const int32_t ot = lt[in ? probe : 0];
                const bool before = ov > v || (ov == v && (ob < bm || (ob == bm && ot < t)));
                pos += (in && before) ? step : 0;
            }
            rank += ob == bm ? 0 : pos;
        }
        if (rank < want) {
            top_idx[(uint64_t)q * want + rank] = (int64_t)bm * (int64_t)vocab + t;
            top_con[(uint64_t)q * want + rank] = v;
            top_unc[(uint64_t)q * want + rank] = v;          // allowed token: constrained == unconstrained
        }
    }
    // fillers: lowest flat indices that are NOT allowed (beam 0 first); constrained -inf, real unconstrained score.
    // MERGE_BLOCK tokens are examined at a time (ballot + prefix count keeps the token order)
    if (n_cand < want) {
        const uint32_t lane = tid & 63, wv = tid >> 6;
        uint32_t out = n_cand;
        for (uint32_t beam = 0; beam < beams && out < want; beam++) {
            const uint32_t r = q * beams + beam;
            const uint32_t *b = bits + (row_broadcast_bits ? 0 : (uint64_t)r * words_per_row);
            for (uint64_t base = 0; base < vocab && out < want; base += MERGE_BLOCK) {
                const uint64_t tok = base + tid;
                const bool na = tok < vocab && !((b[tok >> 5] >> (tok & 31)) & 1);
                const uint64_t bal = __ballot(na);
                __syncthreads();
                if (lane == 0) s_fill[wv] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t mine = (uint32_t)__popcll(bal & ((1ull << lane) - 1)), chunk = 0;
                for (uint32_t w = 0; w < MERGE_BLOCK / 64; w++) { if (w < wv) mine += s_fill[w]; chunk += s_fill[w]; }
                if (na && out + mine < want) {
                    const float lp = logp_processed(logits[(uint64_t)r * vocab + tok], row_max[r], row_lsum[r]);
                    top_idx[(uint64_t)q * want + out + mine] = (int64_t)beam * (int64_t)vocab + (int64_t)tok;
                    top_con[(uint64_t)q * want + out + mine] = -__builtin_huge_valf();
                    top_unc[(uint64_t)q * want + out + mine] = lp + beam_scores[r];
                }
                out += chunk;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K3/K4: locate + doc binning
// ---------------------------------------------------------------------------
__global__ void k_locate(FmiDev ix, uint64_t n, const uint64_t *rows, uint64_t *pos_out, uint64_t *doc_out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t row = rows[i];
    if (row >= ix.n) { pos_out[i] = ~0ull; if (doc_out) doc_out[i] = ~0ull; return; }
    const uint64_t pos = sa_at(ix, row);
    pos_out[i] = pos;
    if (doc_out) doc_out[i] = ix.n_begin ? doc_of(ix, pos) : ~0ull;
}

// ranges form: output element e belongs to range j = upper_bound(out_offsets, e)-1,
// row = lo[j] + (e - out_offsets[j])
__global__ void k_locate_ranges(FmiDev ix, uint64_t n_ranges, const uint64_t *lo, const uint64_t *out_offsets,
                                uint64_t total, uint64_t *pos_out, uint64_t *doc_out)
{
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint64_t a = 0, b = n_ranges;   // last j with out_offsets[j] <= e
    while (b - a > 1) { uint64_t mid = (a + b) >> 1; if (out_offsets[mid] <= e) a = mid; else b = mid; }
    const uint64_t row = lo[a] + (e - out_offsets[a]);
    if (row >= ix.n) { pos_out[e] = ~0ull; if (doc_out) doc_out[e] = ~0ull; return; }
    const uint64_t pos = sa_at(ix, row);
    pos_out[e] = pos;
    if (doc_out) doc_out[e] = ix.n_begin ? doc_of(ix, pos) : ~0ull;
}

// ---------------------------------------------------------------------------
// K6: extract_text / get_doc: the text itself is resident, so
// T[end-1] ... T[begin] (fm_index.cpp:169-184) is a reversed contiguous read.
// ---------------------------------------------------------------------------
__global__ void k_extract(FmiDev ix, uint64_t begin, uint64_t end, uint64_t *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (begin + i >= end) return;
    out[i] = text_at(ix, end - 1 - i);
}

__global__ void k_get_docs(FmiDev ix, const uint64_t *docs, const uint64_t *out_offsets, int64_t shift, int64_t *out)
{
    const uint64_t d = docs[blockIdx.x];
    const uint64_t b = ix.doc_begin[d], e = ix.doc_begin[d + 1];
    int64_t *o = out + out_offsets[blockIdx.x];
    for (uint64_t i = threadIdx.x; b + i < e; i += blockDim.x) o[i] = (int64_t)text_at(ix, e - 1 - i) - shift;
}

// ---------------------------------------------------------------------------
// host side of the query entry points
// ---------------------------------------------------------------------------
static int need_device(fmi *h)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    if (h->device < 0) {
        fmi_set_error("index is not resident on a GPU: libsealfm has no CPU query path (call fmi_to_device on a gfx950 box)");
        return FMI_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(h->device));
    return FMI_OK;
}

static int service_stream(fmi *h, hipStream_t *out)
{
    if (!h->service_stream) {
        hipStream_t s;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        h->service_stream = (void *)s;
    }
    *out = (hipStream_t)h->service_stream;
    return FMI_OK;
}

// synchronous copies on the service stream (pageable host memory: staged by the runtime)
#define COPY_H2D(dst, src, bytes) do { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, sst)); } while (0)
#define COPY_D2H(dst, src, bytes) do { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, sst)); HIPCHK(hipStreamSynchronize(sst)); } while (0)

static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { fmi_set_error("hipMalloc(%zu) failed", bytes); return FMI_ERR_HIP; } return FMI_OK; }
    template <class T> T *as() { return (T *)p; }
};

extern "C" int fmi_dev_enable_probe_count(fmi_t *h, int enable)
{
    int rc = need_device(h); if (rc) return rc;
    if (enable && !h->d_probe_counter) {
        HIPCHK(hipMalloc((void **)&h->d_probe_counter, PROBE_SLOTS * 64));
    }
    if (h->d_probe_counter) HIPCHK(hipMemset(h->d_probe_counter, 0, PROBE_SLOTS * 64));
    h->probe_count_enabled = enable;
    return FMI_OK;
}

static int read_probe_slots(fmi *h, uint64_t out3[4], bool reset)
{
    if (!h->d_probe_counter) { fmi_set_error("probe counter not enabled"); return FMI_ERR_STATE; }
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint64_t> slots(PROBE_SLOTS * 8);
    HIPCHK(hipMemcpy(slots.data(), h->d_probe_counter, PROBE_SLOTS * 64, hipMemcpyDeviceToHost));
    for (int e = 0; e < 4; e++) out3[e] = h->probe_accum[e];      // what the per-call log drained since the last read
    for (uint32_t i = 0; i < PROBE_SLOTS; i++)
        for (int e = 0; e < 4; e++) out3[e] += slots[i * 8 + e];
    if (reset) {
        HIPCHK(hipMemset(h->d_probe_counter, 0, PROBE_SLOTS * 64));
        for (int e = 0; e < 4; e++) h->probe_accum[e] = 0;
    }
    return FMI_OK;
}

// the per-call log: one record per constraint call since it was switched on.  Timing mode (fmi_dev_enable_timing): the record names
// the call's event pair; counting mode (fmi_dev_enable_probe_count): the stream is drained after the call and the blocks its launches
// loaded are read back (a measurement pass: it serialises the calls)
static void call_log_begin(fmi *h, uint32_t kind, uint64_t cur_len, uint64_t rows, bool timed)
{
    if (!h->call_log_enabled) return;
    fmi::CallRec r{(uint32_t)cur_len, (uint32_t)rows, kind, timed ? (int64_t)h->ev_used : -1, 0};
    h->call_log.push_back(r);
}

static int call_log_end(fmi *h, hipStream_t st)
{
    if (!h->call_log_enabled || !h->probe_count_enabled || !h->d_probe_counter || h->call_log.empty()) return FMI_OK;
    HIPCHK(hipStreamSynchronize(st));
    uint64_t slots[PROBE_SLOTS * 8];
    HIPCHK(hipMemcpy(slots, h->d_probe_counter, PROBE_SLOTS * 64, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d_probe_counter, 0, PROBE_SLOTS * 64));
    uint64_t v[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < PROBE_SLOTS; i++)
        for (int e = 0; e < 4; e++) v[e] += slots[i * 8 + e];
    for (int e = 0; e < 4; e++) h->probe_accum[e] += v[e];
    h->call_log.back().blocks = v[0];
    return FMI_OK;
}

extern "C" int fmi_dev_call_log(fmi_t *h, int enable)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    h->call_log_enabled = enable;
    h->call_log.clear();
    return FMI_OK;
}

extern "C" int fmi_dev_read_call_log(fmi_t *h, uint64_t cap, uint32_t *cur_len, uint32_t *rows, uint32_t *kind, float *us, uint64_t *blocks,
                                     uint64_t *n_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (!n_out) { fmi_set_error("null n_out"); return FMI_ERR_ARG; }
    HIPCHK(hipDeviceSynchronize());
    const uint64_t n = std::min<uint64_t>(cap, h->call_log.size());
    for (uint64_t i = 0; i < n; i++) {
        const fmi::CallRec &r = h->call_log[i];
        if (cur_len) cur_len[i] = r.cur_len;
        if (rows) rows[i] = r.rows;
        if (kind) kind[i] = r.kind;
        if (blocks) blocks[i] = r.blocks;
        if (us) {
            float ms = -1.0f;
            if (r.ev >= 0 && (uint64_t)r.ev < h->ev_start.size() && (uint64_t)r.ev < h->ev_used)
                HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)h->ev_start[r.ev], (hipEvent_t)h->ev_stop[r.ev]));
            us[i] = ms < 0 ? -1.0f : ms * 1000.0f;
        }
    }
    *n_out = h->call_log.size();
    h->call_log.clear();
    return FMI_OK;
}

extern "C" int fmi_dev_read_probe_count(fmi_t *h, uint64_t *out)
{
    int rc = need_device(h); if (rc) return rc;
    if (!out) { fmi_set_error("null out"); return FMI_ERR_ARG; }
    uint64_t v[4];
    rc = read_probe_slots(h, v, true); if (rc) return rc;
    *out = v[0];
    return FMI_OK;
}

// diagnostics of the same counting mode: {sectors, wave iterations, nodes expanded, nodes of the binary
// 16-level model} since the last fmi_dev_read_probe_count (nodes / (64 * iterations) = lane utilisation
// of k_expand; 2 * out[3] = the level-probes of SURVEY.md 8(d) for the same work); does not reset.
extern "C" int fmi_dev_read_expand_stats(fmi_t *h, uint64_t *out3)
{
    int rc = need_device(h); if (rc) return rc;
    if (!out3) { fmi_set_error("null out"); return FMI_ERR_ARG; }
    return read_probe_slots(h, out3, false);
}

extern "C" int fmi_dev_bs_step(fmi_t *h, void *stream, uint64_t n, const uint64_t *d_sym, const uint64_t *d_lo,
                               const uint64_t *d_hi, uint64_t *d_lo_out, uint64_t *d_hi_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (n == 0) return FMI_OK;
    hipLaunchKernelGGL(k_bs_step, dim3(blocks_for(n, 64)), dim3(64), 0, (hipStream_t)stream, h->dev, n, d_sym, d_lo, d_hi, d_lo_out, d_hi_out);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

extern "C" int fmi_dev_get_range(fmi_t *h, void *stream, uint64_t n_seq, const int64_t *d_offsets,
                                 const int64_t *d_tokens, int64_t shift, uint64_t *d_lo_out, uint64_t *d_hi_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (n_seq == 0) return FMI_OK;
    hipLaunchKernelGGL((k_get_range<int64_t, int64_t>), dim3(blocks_for(n_seq, 64)), dim3(64), 0, (hipStream_t)stream,
                       h->dev, n_seq, d_offsets, d_tokens, shift, d_lo_out, d_hi_out);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

// workspace of the fmi_dev_* constraint calls: two allowed-token bitmaps (vocab <= 2^17 -> 4096 words/row;
// a call fills one and clears what the previous call left in the other, so no memset launch sits in front
// of a decode step's constraint) and two buffers of (lo, inclusive hi) per row, the incremental constraint
// state of consecutive decode steps
static constexpr uint64_t WS_BITS_WORDS = (1ull << FMI_MAX_LEVELS) / 32;
static inline uint32_t *ws_bits(fmi *h, int which) { return (uint32_t *)h->ws + (uint64_t)which * h->ws_rows * WS_BITS_WORDS; }
static inline uint64_t *ws_state(fmi *h, int which) { return (uint64_t *)(ws_bits(h, 2)) + (uint64_t)which * 2 * h->ws_rows; }
// row-first calls: per row a RowPre (32 B) and the sixteen children of its root node (256 B)
static inline RowPre *ws_pre_rows(fmi *h) { return (RowPre *)ws_state(h, 2); }
static inline uint64_t *ws_pre_child(fmi *h) { return (uint64_t *)(ws_pre_rows(h) + h->ws_rows); }
// list mode of the chained steps (k_beam_advance), two buffers each like the kept ranges: lengths, text positions, BWT symbols
static constexpr uint64_t WS_LIST_MAX = 64;        // = LIST_MAX of k_beam_advance (static_assert there)
static inline uint64_t *ws_list_pos(fmi *h, int which) { return (uint64_t *)h->ws_list + (uint64_t)which * h->ws_rows * WS_LIST_MAX; }
static inline uint32_t *ws_list_sym(fmi *h, int which) { return (uint32_t *)ws_list_pos(h, 2) + (uint64_t)which * h->ws_rows * WS_LIST_MAX; }
static inline uint32_t *ws_list_len(fmi *h, int which) { return ws_list_sym(h, 2) + (uint64_t)which * h->ws_rows; }
extern "C" int fmi_dev_reserve(fmi_t *h, uint64_t max_rows)
{
    int rc = need_device(h); if (rc) return rc;
    if (max_rows <= h->ws_rows) return FMI_OK;
    if (h->ws) { HIPCHK(hipFree(h->ws)); h->ws = nullptr; h->ws_rows = 0; }
    if (h->ws_list) { HIPCHK(hipFree(h->ws_list)); h->ws_list = nullptr; }
    const uint64_t bytes = max_rows * (2 * WS_BITS_WORDS * 4 + 32 + sizeof(RowPre) + FMI_ARITY * 16) + 256;
    HIPCHK(hipMalloc(&h->ws, bytes));
    HIPCHK(hipMalloc(&h->ws_list, max_rows * (2 * WS_LIST_MAX * (8 + 4) + 2 * 4) + 256));
    HIPCHK(hipMemset(h->ws_list, 0xFF, max_rows * (2 * WS_LIST_MAX * (8 + 4) + 2 * 4) + 256));
    h->ws_bytes = bytes; h->ws_rows = max_rows;
    HIPCHK(hipMemset(h->ws, 0, max_rows * 2 * WS_BITS_WORDS * 4));
    HIPCHK(hipDeviceSynchronize());     // the memset runs on the null stream; the callers' streams are non-blocking
    h->ws_seq = 0; h->ws_dirty[0] = h->ws_dirty[1] = 0;
    h->state_tag = 0; h->chain_tag = 0; h->state_base = 0; h->bits_prefilled = 0;
    return FMI_OK;
}

static constexpr size_t MAX_TIMED_LAUNCHES = 8192;

extern "C" int fmi_dev_enable_timing(fmi_t *h, int enable)
{
    int rc = need_device(h); if (rc) return rc;
    if (enable && h->ev_start.empty()) {
        h->ev_start.resize(MAX_TIMED_LAUNCHES); h->ev_stop.resize(MAX_TIMED_LAUNCHES);
        for (size_t i = 0; i < MAX_TIMED_LAUNCHES; i++) {
            HIPCHK(hipEventCreate((hipEvent_t *)&h->ev_start[i]));
            HIPCHK(hipEventCreate((hipEvent_t *)&h->ev_stop[i]));
        }
    }
    h->timing_enabled = enable;
    h->ev_used = 0;
    return FMI_OK;
}

extern "C" int fmi_dev_read_timing(fmi_t *h, uint64_t *launches_out, double *total_ms_out)
{
    int rc = need_device(h); if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    double total = 0;
    const uint64_t m = std::min<uint64_t>(h->ev_used, MAX_TIMED_LAUNCHES);
    for (uint64_t i = 0; i < m; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)h->ev_start[i], (hipEvent_t)h->ev_stop[i]));
        total += ms;
    }
    if (launches_out) *launches_out = m;
    if (total_ms_out) *total_ms_out = total;
    h->ev_used = 0;
    return FMI_OK;
}

extern "C" const void *fmi_dev_array(const fmi_t *h, const char *name, uint64_t *n_out, uint32_t *elem_out)
{
    if (!h || !name || h->device < 0) return nullptr;
    std::string s(name);
    auto ret = [&](const void *p, uint64_t n, uint32_t e) { if (n_out) *n_out = n; if (elem_out) *elem_out = e; return p; };
    const FmiDev &d = h->dev;
    if (s == "sa_lo") return ret(d.sa_lo, d.n, 4);
    if (s == "sa_hi") return ret(d.sa_hi, d.sa_hi ? d.n : 0, 1);
    if (s == "text") return ret(d.text, d.n, d.sym_bytes);
    if (s == "wm") return ret(d.wm, (uint64_t)d.dlevels * d.nblk * FMI_BLOCK_WORDS, 8);
    if (s == "C") return ret(d.C, d.max_sym + 2, 8);
    if (s == "leaf") return ret(d.leaf, d.max_sym + 1, 8);
    if (s == "q1") return ret(d.q1, d.max_sym + 1, 1);
    if (s == "doc_begin") return ret(d.doc_begin, d.n_begin, 8);
    return nullptr;
}

static size_t expand_lds_bytes(uint32_t dlevels, bool with_bits)
{
    const size_t nsym = (size_t)1 << (FMI_DIGIT_BITS * (dlevels - 1));
    return (size_t)exp_slots((int)dlevels - 1) * 16 + 8 * 4 + (with_bits ? ((nsym + 31) / 32) * 4 : 0);
}

// top digits that occur: one wave per (row, top digit)
static uint32_t top_digits(const fmi *h) { return (uint32_t)(h->max_sym >> (FMI_DIGIT_BITS * (h->dlevels - 1))) + 1; }

// dense per-row symbol counts of `rows` intervals (items[0..rows)), one launch
static int launch_expand_dense(fmi *h, hipStream_t st, const ExpandItem *items, uint64_t rows, const EmitTarget &tgt)
{
    uint64_t *pc = h->probe_count_enabled ? h->d_probe_counter : nullptr;
    const uint64_t grid = rows * top_digits(h);
    if (grid > 0x7fffffffull) { fmi_set_error("too many intervals in one call"); return FMI_ERR_CAPACITY; }
    auto kern = h->dev.nsb > 1 ? k_expand_dense<true> : k_expand_dense<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), expand_lds_bytes(h->dlevels, false), st, h->dev, items, (uint32_t)rows, tgt, pc);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

// One constraint call = ONE launch of k_constrain.  `d_bits` = null: the workspace bitmaps, alternating
// between calls (the kernel clears the other one); otherwise the caller's buffer, cleared here first.
// host view of the row groups of one call (ConstrainArgs::grp_*)
struct RowGroups {
    uint32_t n = 1;
    uint64_t rows[MAX_ROW_GROUPS] = {0, 0, 0};
    int64_t eos[MAX_ROW_GROUPS] = {0, 0, 0};
    const int64_t *force[MAX_ROW_GROUPS] = {nullptr, nullptr, nullptr};
    uint64_t n_force[MAX_ROW_GROUPS] = {0, 0, 0};
    int64_t stop[MAX_ROW_GROUPS] = {-1, -1, -1};      // per-group stop_at_count; -1: the call's
    static RowGroups one(uint64_t rows, int64_t eos_id, const int64_t *force_from, uint64_t n_force)
    {
        RowGroups g; g.rows[0] = rows; g.eos[0] = eos_id; g.force[0] = force_from; g.n_force[0] = n_force; return g;
    }
};

// ---------------------------------------------------------------------------
// The first constrained step of a decode (cur_len == 2: every row's prefix is the forced prefix P of its decode + ONE token) from
// per-token tables (FmiPrefixTable, fmi_internal.h).  The expansion of such a row starts from [C[c], C[c + 1]) -- a static property of
// the index -- and is the widest of the decode (10^3..10^4 distinct continuations): in k_constrain its leaf level, 95 % of the bytes,
// streams at the chip's random-request rate, but only after a ramp of dependent upper levels (~10 us) and with a tail of workgroups that
// finish alone (~12 us of a 70 us call).  With the leaf-level nodes of every token tabulated once per index (a few hundred MB at NQ
// size, built in milliseconds by the level-by-level expansion below), the call is ONE flat list -- the rows' node lists back to back,
// found through a prefix sum over the rows in LDS -- cut evenly over the grid: every lane pair loads the one or two blocks of its node,
// takes the sixteen child-exists bits and ORs them into the row's token bitmap.  No ramp, no tail.
// ---------------------------------------------------------------------------
struct PtNode { uint64_t lo, hi; uint32_t prefix, owner; };

// level-0 nodes: the interval of P + [t] for every token t (the same backward-search steps as row_range_and_class, quirk Q1 included)
__global__ void k_pt_roots(FmiDev ix, ForceFrom ff, int64_t shift, uint64_t vocab, uint64_t *root, uint32_t *flag)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= vocab) return;
    uint64_t l = 0, rr = ix.n;
    for (uint32_t j = 0; j <= ff.n; j++) {
        const int64_t tok = j < ff.n ? ff.tok[j] : (int64_t)t;
        bs_step(ix, (uint64_t)(tok + shift), l, rr, l, rr, nullptr);
    }
    uint64_t lo = l, hi = rr + 1;
    if (hi > ix.n) hi = ix.n;
    root[2 * t] = l; root[2 * t + 1] = rr;
    flag[t] = hi > lo ? 1u : 0u;
}

__global__ void k_pt_root_nodes(const uint64_t *root, const uint32_t *flag, const uint64_t *offs, uint64_t vocab, uint64_t n, PtNode *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= vocab || !flag[t]) return;
    PtNode nd;
    nd.lo = root[2 * t]; nd.hi = min(root[2 * t + 1] + 1, n); nd.prefix = 0; nd.owner = (uint32_t)t;
    out[offs[t]] = nd;
}

// children of every node of level k: count, then (after an exclusive scan) write; one thread per node, sixteen digits each
__global__ void k_pt_children(FmiDev ix, uint32_t k, const PtNode *nodes, uint64_t n, const uint64_t *offs, uint32_t *cnt, PtNode *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PtNode nd = nodes[i];
    uint64_t at = out ? offs[i] : 0;
    uint32_t c = 0;
    for (uint32_t d = 0; d < FMI_ARITY; d++) {
        const uint64_t clo = wm_step(ix, k, nd.lo, d), chi = wm_step(ix, k, nd.hi, d);
        if (chi > clo) {
            c++;
            if (out) { PtNode ch; ch.lo = clo; ch.hi = chi; ch.prefix = (nd.prefix << FMI_DIGIT_BITS) | d; ch.owner = nd.owner; out[at++] = ch; }
        }
    }
    if (!out) cnt[i] = c;
}

__global__ void k_pt_pack(const PtNode *nodes, uint64_t n, uint4 *packed, uint32_t *owner)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    packed[i] = pack_node(nodes[i].lo, nodes[i].hi, nodes[i].prefix);
    owner[i] = nodes[i].owner;
}

// off[t] = first node whose owner is >= t (the nodes are grouped by owner, ascending)
__global__ void k_pt_offsets(const uint32_t *owner, uint64_t n, uint64_t vocab, uint64_t *off)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > vocab) return;
    uint64_t a = 0, b = n;
    while (a < b) { const uint64_t mid = (a + b) >> 1; if (owner[mid] < t) a = mid + 1; else b = mid; }
    off[t] = a;
}

static constexpr uint64_t PT_MAX_NODES = 1ull << 28;      // 4 GiB of packed nodes: beyond that the generic path serves the prefix

static int build_prefix_table_impl(fmi *h, FmiPrefixTable &T);

// A table is an optimisation: the generic expansion serves every prefix without one and needs no extra memory.  So whatever goes wrong
// while building -- an allocation on an index sized to HBM, a scan -- leaves "no table" (T.ok = false, partial buffers freed, the HIP
// error state cleared), never a failed constraint call.
static int build_prefix_table(fmi *h, FmiPrefixTable &T)
{
    const int rc = build_prefix_table_impl(h, T);
    if (rc != FMI_OK || !T.ok) {
        (void)hipDeviceSynchronize();
        if (T.d_root) (void)hipFree(T.d_root);
        if (T.d_nodes) (void)hipFree(T.d_nodes);
        if (T.d_off) (void)hipFree(T.d_off);
        T.d_root = nullptr; T.d_nodes = nullptr; T.d_off = nullptr; T.n_nodes = 0; T.ok = false;
        (void)hipGetLastError();
    }
    return FMI_OK;
}

static int build_prefix_table_impl(fmi *h, FmiPrefixTable &T)
{
    // (runs once per index and forced prefix, on the null stream, synchronously: allocation + a few passes; milliseconds at NQ size)
    const uint64_t V = T.vocab;
    ForceFrom ff{};
    ff.n = (uint32_t)T.force.size();
    for (uint32_t i = 0; i < ff.n; i++) ff.tok[i] = T.force[i];
    DevBuf flag, offs, tmp;
    int rc;
    HIPCHK(hipMalloc((void **)&T.d_root, V * 16));
    if (h->opt.pt_inject_failure) { fmi_set_error("prefix table: injected failure"); return FMI_ERR_HIP; }
    if ((rc = flag.alloc((V + 1) * 4)) || (rc = offs.alloc((V + 1) * 8))) return rc;
    HIPCHK(hipMemsetAsync(flag.p, 0, (V + 1) * 4, 0));             // (the scan runs over V + 1 flags: the last one stays 0)
    hipLaunchKernelGGL(k_pt_roots, dim3(blocks_for(V, 256)), dim3(256), 0, 0, h->dev, ff, T.shift, V, T.d_root, flag.as<uint32_t>());
    size_t tb = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tb, flag.as<uint32_t>(), offs.as<uint64_t>(), (uint64_t)0, V + 1, rocprim::plus<uint64_t>(), (hipStream_t)0));
    if ((rc = tmp.alloc(tb + 256))) return rc;
    HIPCHK(rocprim::exclusive_scan(tmp.p, tb, flag.as<uint32_t>(), offs.as<uint64_t>(), (uint64_t)0, V + 1, rocprim::plus<uint64_t>(), (hipStream_t)0));
    uint64_t n = 0;
    HIPCHK(hipMemcpy(&n, offs.as<uint64_t>() + V, 8, hipMemcpyDeviceToHost));
    PtNode *cur = nullptr, *nxt = nullptr;
    if (hipMalloc((void **)&cur, std::max<uint64_t>(n, 1) * sizeof(PtNode)) != hipSuccess) { fmi_set_error("prefix table: hipMalloc failed"); return FMI_ERR_HIP; }
    hipLaunchKernelGGL(k_pt_root_nodes, dim3(blocks_for(V, 256)), dim3(256), 0, 0, (const uint64_t *)T.d_root, flag.as<uint32_t>(), offs.as<uint64_t>(), V, h->n, cur);
    auto fail = [&](int code) { if (cur) (void)hipFree(cur); if (nxt) (void)hipFree(nxt); cur = nxt = nullptr; return code; };
    for (uint32_t k = 0; k + 1 < h->dlevels && n; k++) {
        DevBuf cnt, off2, tmp2;
        if ((rc = cnt.alloc((n + 1) * 4)) || (rc = off2.alloc((n + 1) * 8))) return fail(rc);
        if (hipMemsetAsync(cnt.p, 0, (n + 1) * 4, 0) != hipSuccess) return fail(FMI_ERR_HIP);
        hipLaunchKernelGGL(k_pt_children, dim3(blocks_for(n, 256)), dim3(256), 0, 0, h->dev, k, (const PtNode *)cur, n, (const uint64_t *)nullptr, cnt.as<uint32_t>(), (PtNode *)nullptr);
        size_t t2 = 0;
        if (rocprim::exclusive_scan(nullptr, t2, cnt.as<uint32_t>(), off2.as<uint64_t>(), (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), (hipStream_t)0) != hipSuccess) return fail(FMI_ERR_HIP);
        if ((rc = tmp2.alloc(t2 + 256))) return fail(rc);
        if (rocprim::exclusive_scan(tmp2.p, t2, cnt.as<uint32_t>(), off2.as<uint64_t>(), (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), (hipStream_t)0) != hipSuccess) return fail(FMI_ERR_HIP);
        uint64_t m = 0;
        if (hipMemcpy(&m, off2.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(FMI_ERR_HIP);
        if (m > PT_MAX_NODES) { (void)fail(0); T.ok = false; return FMI_OK; }        // too large for a table: not an error
        if (hipMalloc((void **)&nxt, std::max<uint64_t>(m, 1) * sizeof(PtNode)) != hipSuccess) { (void)fail(0); (void)hipGetLastError(); T.ok = false; return FMI_OK; }
        hipLaunchKernelGGL(k_pt_children, dim3(blocks_for(n, 256)), dim3(256), 0, 0, h->dev, k, (const PtNode *)cur, n, (const uint64_t *)off2.as<uint64_t>(), (uint32_t *)nullptr, nxt);
        if (hipDeviceSynchronize() != hipSuccess) return fail(FMI_ERR_HIP);
        (void)hipFree(cur); cur = nxt; nxt = nullptr; n = m;
    }
    DevBuf owner;
    if ((rc = owner.alloc(std::max<uint64_t>(n, 1) * 4))) return fail(rc);
    if (hipMalloc(&T.d_nodes, std::max<uint64_t>(n, 1) * 16) != hipSuccess || hipMalloc((void **)&T.d_off, (V + 1) * 8) != hipSuccess) { (void)hipGetLastError(); (void)fail(0); T.ok = false; return FMI_OK; }
    if (n) hipLaunchKernelGGL(k_pt_pack, dim3(blocks_for(n, 256)), dim3(256), 0, 0, (const PtNode *)cur, n, (uint4 *)T.d_nodes, owner.as<uint32_t>());
    hipLaunchKernelGGL(k_pt_offsets, dim3(blocks_for(V + 1, 256)), dim3(256), 0, 0, (const uint32_t *)owner.as<uint32_t>(), n, V, T.d_off);
    if (hipDeviceSynchronize() != hipSuccess) return fail(FMI_ERR_HIP);
    (void)fail(0);
    T.n_nodes = n;
    T.ok = true;
    return FMI_OK;
}

// the table of forced prefix `force` for this (shift, vocab), built on first use; nullptr: none (too large / switched off)
static const FmiPrefixTable *prefix_table_for(fmi *h, const int64_t *force, uint64_t n_force, int64_t shift, uint64_t vocab, int *rc_out)
{
    *rc_out = FMI_OK;
    for (const FmiPrefixTable &t : h->prefix_tables)
        if (t.shift == shift && t.vocab == vocab && t.force.size() == n_force && std::equal(t.force.begin(), t.force.end(), force)) return t.ok ? &t : nullptr;
    if (h->prefix_tables.size() >= 8) return nullptr;          // a handful of decodes per searcher: anything beyond takes the generic path
    h->prefix_tables.emplace_back();
    FmiPrefixTable &T = h->prefix_tables.back();
    T.force.assign(force, force + n_force); T.shift = shift; T.vocab = vocab;
    *rc_out = build_prefix_table(h, T);
    if (*rc_out != FMI_OK) { T.ok = false; return nullptr; }
    return T.ok ? &T : nullptr;
}

struct TableArgs {
    const uint64_t *off[MAX_ROW_GROUPS];
    const uint64_t *root[MAX_ROW_GROUPS];
    const uint4 *nodes[MAX_ROW_GROUPS];
    uint8_t *sym;                  // [rows][sym_row_words * 4]: bit s of a row = symbol s follows its prefix; zero on entry
    uint64_t sym_row_words;
};

// entry g of a per-group kernel argument for a per-lane g: selects between scalar registers (indexing the array with a vector would
// be a load from the kernel-argument segment, and one more dependent access in front of what it addresses)
template <class T>
__device__ __forceinline__ T of_group(const T (&v)[MAX_ROW_GROUPS], uint32_t g)
{
    static_assert(MAX_ROW_GROUPS == 3, "of_group selects between three entries");
    return g == 0 ? v[0] : (g == 1 ? v[1] : v[2]);
}

static constexpr uint32_t TABLE_WG = 256;
static constexpr uint32_t TABLE_GRID = 1024;      // workgroups of the flat leaf-level pass: the 4 x 4 waves a CU holds at ~100 registers, ONE round
__host__ __device__ constexpr size_t table_lds_bytes(uint32_t rows) { return (size_t)(rows + 1) * 4 + (size_t)rows * 8 + 264 * 4; }

template <bool SB>
__global__ __launch_bounds__(TABLE_WG) void k_constrain_table(FmiDev ix, ConstrainArgs a, TableArgs tb)
{
    extern __shared__ uint64_t s_dyn64[];
    uint64_t *s_base = s_dyn64;                                           // [rows]   first table node of the row
    uint32_t *s_start = reinterpret_cast<uint32_t *>(s_base + a.rows);   // [rows+1] first flat node index of the row
    uint32_t *s_part = s_start + a.rows + 1;                              // [256 + 8] scan scratch
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t D = ix.dlevels, kl = D - 1;                            // the leaf level
    const bool counting = a.probe_counter != nullptr;
    ExpCounters ctr{0, 0, 0, 0};
    if (a.clear) {      // housekeeping for the next call: this workgroup's share of the other bitmap buffer
        const uint64_t per = (a.clear_words + gridDim.x - 1) / gridDim.x;
        const uint64_t w0 = (uint64_t)blockIdx.x * per;
        for (uint64_t w = w0 + tid; w < w0 + per && w < a.clear_words; w += TABLE_WG) a.clear[w] = 0u;
    }
    // ---- every workgroup: the rows' node counts (two dependent loads: last token, its table offsets) and their prefix sum ----
    // (rows strided over the threads, four per thread in flight: the loads of a batch are independent of one another)
    for (uint32_t r0 = 0; r0 < a.rows; r0 += 4 * TABLE_WG) {
        int64_t tok[4];
        uint64_t base[4], next[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * TABLE_WG + tid;
            tok[u] = r < a.rows ? a.ids[(uint64_t)r * a.ids_stride + (a.cur_len - 1)] : a.pad_id;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * TABLE_WG + tid;
            const uint32_t grp = row_group(a, r < a.rows ? r : 0);
            const bool dead = tok[u] == of_group(a.grp_eos, grp) || tok[u] == a.pad_id;
            const bool in_table = !dead && tok[u] >= 0 && (uint64_t)tok[u] < a.vocab;
            base[u] = in_table ? of_group(tb.off, grp)[tok[u]] : 0;
            next[u] = in_table ? of_group(tb.off, grp)[tok[u] + 1] : 0;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * TABLE_WG + tid;
            if (r >= a.rows) continue;
            s_base[r] = base[u];
            s_start[r] = (uint32_t)(next[u] - base[u]);
            if (blockIdx.x == 0) {
                // the row's side effects, once: its kept range for the next step's one-step advance, the special tokens of its class
                const uint32_t grp = row_group(a, r);
                const bool dead = tok[u] == of_group(a.grp_eos, grp) || tok[u] == a.pad_id;
                if (a.st_out) {                        // (inclusive range; a token outside the vocabulary: the empty one; finished rows too: row_range_and_class)
                    const bool in_table = tok[u] >= 0 && (uint64_t)tok[u] < a.vocab;
                    a.st_out[2 * (uint64_t)r] = in_table ? of_group(tb.root, grp)[2 * tok[u]] : 1;
                    a.st_out[2 * (uint64_t)r + 1] = in_table ? of_group(tb.root, grp)[2 * tok[u] + 1] : 0;
                }
                if (dead) set_bit_global(a, r, a.pad_id);
                if (a.always_allow_eos) set_bit_global(a, r, of_group(a.grp_eos, grp));
            }
        }
    }
    __syncthreads();
    // thread t sums its per_thread consecutive rows; then the scan over the threads
    const uint32_t per_thread = (a.rows + TABLE_WG - 1) / TABLE_WG;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per_thread; j++) {
        const uint32_t r = tid * per_thread + j;
        if (r < a.rows) mine += s_start[r];
    }
    s_part[tid] = mine;
    __syncthreads();
    if (wave == 0) {                       // exclusive scan of the 256 partial sums by one wave: four per lane
        uint32_t v[4], sum = 0;
        for (int j = 0; j < 4; j++) { v[j] = s_part[4 * lane + j]; sum += v[j]; }
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += y; }
        uint32_t run = incl - sum;
        for (int j = 0; j < 4; j++) { s_part[4 * lane + j] = run; run += v[j]; }
        if (lane == 63) s_part[256] = incl;
    }
    __syncthreads();
    const uint32_t T = s_part[256];
    {
        uint32_t run = s_part[tid];
        for (uint32_t j = 0; j < per_thread; j++) {
            const uint32_t r = tid * per_thread + j;
            if (r >= a.rows) break;
            const uint32_t c = s_start[r];
            s_start[r] = run;
            run += c;
        }
        if (tid == 0) s_start[a.rows] = T;
    }
    __syncthreads();
    // ---- the flat node list, cut evenly: workgroup w takes [w * per, (w + 1) * per), 32 nodes (lane pairs) per wave iteration ----
    const uint32_t per = ((T + gridDim.x - 1) / gridDim.x + EXP_PAIRS - 1) / EXP_PAIRS * EXP_PAIRS;
    const uint32_t lo_i = blockIdx.x * per, hi_i = min(T, lo_i + per);
    const uint32_t end = lane & 1, pair = lane >> 1;
    const uint32_t pad_bits = FMI_DIGIT_BITS * D - ix.levels;
    // (a wave's iteration = a row lookup in LDS, the node's table slot, then its block: two dependent global loads.  The lookup and the
    //  slot of the NEXT iteration are issued behind this iteration's block load, so that a wave waits for one latency per 32 nodes, not two)
    const uint32_t step = (TABLE_WG / 64) * EXP_PAIRS;
    auto fetch = [&](uint32_t at, uint32_t &r, uint4 &nd) -> bool {
        const uint32_t i = at + pair;
        r = 0;
        nd = make_uint4(0u, 0u, 0u, 0u);
        if (i >= hi_i) return false;
        uint32_t x = 0, y = a.rows;           // last row with s_start[row] <= i
        while (y - x > 1) { const uint32_t mid = (x + y) >> 1; if (s_start[mid] <= i) x = mid; else y = mid; }
        r = x;
        nd = of_group(tb.nodes, row_group(a, r))[s_base[r] + (i - s_start[r])];
        return true;
    };
    uint32_t c0 = lo_i + wave * EXP_PAIRS;
    uint32_t r_next = 0;
    uint4 nd_next = make_uint4(0u, 0u, 0u, 0u);
    bool act_next = c0 < hi_i && fetch(c0, r_next, nd_next);
    for (; c0 < hi_i; c0 += step) {
        const uint32_t r = r_next;
        const uint4 nd = nd_next;
        const bool act = act_next;
        const uint64_t nlo = (uint64_t)nd.x | ((uint64_t)(nd.z & 0xff) << 32);
        const uint64_t nhi = (uint64_t)nd.y | ((uint64_t)((nd.z >> 8) & 0xff) << 32);
        const uint32_t prefix = nd.z >> 16;
        const uint64_t p = end ? nhi : nlo;
        const uint64_t blk = p >> FMI_BLOCK_SHIFT, oblk = (end ? nlo : nhi) >> FMI_BLOCK_SHIFT;
        HBlock b;
        if (act) wm_load_block(ix, kl, blk, b);
        act_next = c0 + step < hi_i && fetch(c0 + step, r_next, nd_next);
        uint32_t rk[16];
#pragma unroll
        for (uint32_t d = 0; d < 16; d++) rk[d] = 0;
        if (act) wm_block_ranks(b, (uint32_t)p & (FMI_BLOCK_BITS - 1), rk);
        // superblocked index: the ends' rows of counts matter only when the ends lie in different superblocks (2^20 positions: rare at the
        // leaf level) -- then digit d has (rowh[d] - rowl[d]) occurrences between the two superblock starts, which is all an existence
        // test needs of them (saturated to 32 bits: the in-superblock counters are smaller).  Sixteen loads for the lanes that need them.
        uint32_t between[8];
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) between[s] = 0u;
        if constexpr (SB) {
            const uint64_t sb_mine = blk >> ix.sb_shift, sb_other = oblk >> ix.sb_shift;
            if (act && sb_mine != sb_other) {
                const uint64_t *rowl = ix.sbase + ((uint64_t)kl * ix.nsb + (end ? sb_other : sb_mine)) * FMI_ARITY + 8 * end;
                const uint64_t *rowh = ix.sbase + ((uint64_t)kl * ix.nsb + (end ? sb_mine : sb_other)) * FMI_ARITY + 8 * end;
#pragma unroll
                for (uint32_t s = 0; s < 8; s++) {
                    const uint64_t d = rowh[s] - rowl[s];
                    between[s] = d > 0xffffffffull ? 0xffffffffu : (uint32_t)d;
                }
            }
        }
        uint32_t hm = 0;                     // which of MY eight children (digits s + 8 * end) exist
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) {
            // lane `end` = 0 holds rank_lo[] and takes digit s, lane 1 holds rank_hi[] and takes digit s + 8
            const uint32_t x = dpp_xor1(rk[s]), y = dpp_xor1(rk[s + 8]);
            const uint32_t cl = end ? y : rk[s], ch = end ? rk[s + 8] : x;
            hm |= (uint32_t)(act && (uint64_t)between[s] + ch > cl) << s;
        }
        if (counting) {
            const uint32_t other_hm = dpp_xor1(hm);
            if (act && end == 0) { ctr.model += model_nodes(hm | (other_hm << 8), kl, pad_bits); ctr.probes += oblk != blk ? 2 : 1; }
            if (lane == 0) ctr.probes += 4;                  // the 32 table slots of the iteration: four 128-byte lines
            const uint32_t left = hi_i - c0;
            ctr.iters++; ctr.nodes += left < EXP_PAIRS ? left : EXP_PAIRS;
        }
        if (act) {
            // my eight child bits = byte (2 * prefix + end) of the row's SYMBOL bitmap: one owner per byte, plain stores (8-bit masks
            // ORed into the token bitmap with atomics -- tokens = symbols - shift straddle bytes -- ran at half the speed: 5 M atomics)
            if (prefix == 0 && end == 0) hm &= ~1u;          // symbol 0 is the sentinel, never a token
            if (hm) tb.sym[(uint64_t)r * (tb.sym_row_words * 4) + ((uint64_t)prefix << 1) + end] = (uint8_t)hm;
        }
    }
    if (counting) flush_counters(a.probe_counter, ctr);
}

// second launch of a table call: the rows' symbol bitmaps -> token bitmaps (token = symbol - shift, clipped to the vocabulary; ORed:
// k_constrain_table has set the special tokens), and the OTHER symbol buffer -- the previous table call's -- zeroed for the next one
__global__ __launch_bounds__(256) void k_table_bits(ConstrainArgs a, const uint32_t *sym, uint64_t sym_row_words, uint32_t *other, uint32_t other_rows)
{
    const uint32_t r = blockIdx.y;
    const uint64_t W = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < other_rows && W < sym_row_words) other[(uint64_t)r * sym_row_words + W] = 0u;
    if (r >= a.rows || W >= a.words_per_row) return;
    const int64_t s0 = (int64_t)(32 * W) + a.shift;          // the symbol of the word's first token
    const int64_t i = s0 >> 5;                               // (arithmetic shift: floor)
    const uint32_t sh = (uint32_t)s0 & 31;
    const uint32_t *row = sym + (uint64_t)r * sym_row_words;
    const uint32_t w0 = (i >= 0 && (uint64_t)i < sym_row_words) ? row[i] : 0u;
    const uint32_t w1 = (i + 1 >= 0 && (uint64_t)(i + 1) < sym_row_words) ? row[i + 1] : 0u;
    uint32_t v = sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
    if (32 * W + 32 > a.vocab) v &= (1u << (uint32_t)(a.vocab - 32 * W)) - 1;
    if (v) a.bits[(uint64_t)r * a.words_per_row + W] |= v;
}

// ---------------------------------------------------------------------------
// k_beam_advance -- what the beam loop does between two model steps, as ONE launch (reference beam_search.py:309-332 + the
// keep-history scorer's process(), 658-685; seal_amd/beam_search.py ran it as ~15 torch launches per step: //, %, index, argsort,
// gathers, cat, the ancestry-table permutation), AND the rows' chains of the NEXT constraint call.
// One workgroup per query, one wave per new beam:
//   * the 2K ranked candidates of the query (k_query_merge's output) are read by every wave (lane = candidate); the first K that are not
//     the decode's eos continue, in rank order (stable; the same order torch.argsort(stable=True) of the eos flags gives);
//   * history: candidate c -> its hypothesis (prefix of its source row + its token) and score into the decode's packed history;
//   * new beam j (wave j): ids row = its source row + its token (in place: all loads of a query's rows come before its stores),
//     beam score, parent row, the decoder's next input token, and column j of the decoder's ancestry table = its source's column;
//   * chain (`chain`): the new row's prefix range = ONE backward-search step with its token from the range kept for its source row, its
//     class (beam_search.py:87-131), and the split of its root node over the sixteen top digits -- exactly what k_constrain_rows computes in
//     front of a row-first call, but here, a model step ahead of the call that needs it, from values this kernel holds in registers.
//     The next call's k_constrain starts from the results (RowPre / pre_child) with a single load per wave.
// ---------------------------------------------------------------------------
static constexpr uint32_t ADV_MAX_WAVES = 16;
static constexpr uint32_t ADV_ITERS = 2;          // new beams per wave (two explicit register sets in the kernel): num_beams <= 32, the top-2K kernels' own limit

struct AdvanceArgs {
    uint32_t batch, beams;
    uint64_t cur_len;                 // tokens per row before the advance
    int64_t *ids; uint64_t ids_stride;
    uint64_t vocab; int64_t shift, pad_id;
    const int64_t *top_idx; const float *top_unc;          // [batch][2 * beams]
    float *beam_scores; int64_t *beam_idx; int64_t *tokens_out;   // [batch * beams]
    int32_t *anc; uint64_t anc_rows; uint32_t anc_T;       // the decoder's ancestry table [anc_T][anc_rows], or null
    int64_t *hist_tok[MAX_ROW_GROUPS]; float *hist_sc[MAX_ROW_GROUPS];
    uint32_t hist_H[MAX_ROW_GROUPS], hist_L[MAX_ROW_GROUPS], hist_off;
    uint32_t grp_first_q[MAX_ROW_GROUPS];                  // first query of group g (unused groups: 0xffffffff)
    int64_t grp_eos[MAX_ROW_GROUPS], grp_stop[MAX_ROW_GROUPS];
    uint32_t grp_nff[MAX_ROW_GROUPS];                      // tokens of the group's forced prefix (measurement mode: the binary model's count)
    int chain;
    int phase;                        // 0: everything; 1: the beam bookkeeping only; 2: the chains only (timing passes: two launches, so that events bracket the chains)
    int always_allow_eos;
    const uint64_t *st_in; uint64_t st_base; uint64_t *st_out;
    RowPre *pre_rows; uint64_t *pre_child;
    // list mode (below): per row the text positions / BWT symbols of its <= LIST_MAX suffix-array rows; *_in: the previous step's
    const uint32_t *lm_in; const uint64_t *lp_in; const uint32_t *ls_in;
    uint32_t *lm_out; uint64_t *lp_out; uint32_t *ls_out;
    uint32_t *bits_next; uint64_t words_per_row;       // the bitmap the NEXT constraint call will fill (zero now): list rows' tokens go straight in
    uint64_t *probe_counter;
};

// List mode.  Once the interval of a row's prefix holds at most LIST_MAX suffix-array rows, the row stops walking the wavelet matrix: it
// keeps the TEXT POSITIONS of those rows, P[k] = SA[lo + k], and their BWT symbols S[k] = text[P[k] - 1] (both arrays are resident: this is
// what 288 GB buy).  Extending the prefix by token c is then arithmetic on the list -- the rows that survive are those with S[k] == c, in the
// same order, and their new positions are P[k] - 1, because SA[LF(i)] = SA[i] - 1 and LF keeps the order of equal symbols -- plus ONE
// dependent access, the gather of the survivors' new symbols text[P[k] - 2], which are at the same time the row's allowed tokens
// (fm_index.cpp:91-109: distinct symbols of BWT[lo, hi)).  Against the wavelet-matrix route (4 dependent probes for the backward-search
// step, then root + 3 levels of the expansion in a second launch: ~9 dependent accesses) that is 2 (the parent's list, the gather), in the
// launch that advances the beams.  Same sets, same counts: both are functions of the same suffix-array rows.  Lists only ever shrink, so
// a row in list mode stays there.  lm[row] = its length, or LIST_RANGE_MODE while the row still carries an interval.
// entries of a row's list per lane of its wave.  Measured on the bench workload (profiles/r5_list_length_ab.txt): 1 (lists of <= 64 rows)
// adds 10 us of chains to a step's k_beam_advance and leaves k_constrain 16 us at the 8th / 9th token; 16 (<= 1024 rows: k_constrain is idle
// from the 8th token on, 9 -> 7 us) makes the chains themselves 25 - 55 us -- a thousand gathers and as many atomics on a handful of words by
// ONE wave per row.  The code is written for any value; 1 is what runs.
static constexpr uint32_t LIST_PER_LANE = 1;
static constexpr uint32_t LIST_MAX = 64 * LIST_PER_LANE;
static constexpr uint32_t LIST_RANGE_MODE = 0xffffffffu;
static_assert(LIST_MAX == WS_LIST_MAX, "the workspace holds LIST_MAX entries per row");
static constexpr int64_t ROW_DONE = -2;            // RowPre::single: the row's bits are in the bitmap already, k_constrain has nothing to do for it

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// nodes the binary 16-level model visits to emit k distinct symbols: N(k) = sum over the levels of min(2^level, k),
// the survey's own formula (measurement mode only; k = the bits the wave has just set in the row's bitmap)
__device__ __forceinline__ uint32_t model_nodes_of_k(uint32_t k, uint32_t levels)
{
    uint32_t nodes = 0;
    for (uint32_t l = 0; l < levels; l++) nodes += (l < 31 && (1u << l) < k) ? (1u << l) : k;
    return nodes;
}



__global__ __launch_bounds__(64 * ADV_MAX_WAVES) void k_beam_advance(FmiDev ix, AdvanceArgs a)
{
    const uint32_t q = blockIdx.x, lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const uint32_t K = a.beams, W2 = 2 * K, t = (uint32_t)a.cur_len;
    uint32_t grp = 0;
#pragma unroll
    for (int g = 1; g < MAX_ROW_GROUPS; g++) grp += q >= a.grp_first_q[g] ? 1u : 0u;
    const int64_t eos = of_group(a.grp_eos, grp);
    const uint32_t q_in_grp = q - of_group(a.grp_first_q, grp);
    // ---- the query's 2K ranked candidates, in every wave (lane = candidate) ----
    const bool valid = lane < W2 && a.phase != 2;            // (a chains-only launch reads what the bookkeeping launch left instead)
    const uint64_t flat = valid ? (uint64_t)a.top_idx[(uint64_t)q * W2 + lane] : 0ull;
    const float csc = valid ? a.top_unc[(uint64_t)q * W2 + lane] : 0.f;
    const uint32_t cbeam = (uint32_t)(flat / a.vocab);                                  // next_indices = flat // V (beam_search.py:309)
    const uint32_t ctok = (uint32_t)(flat - (uint64_t)cbeam * a.vocab);                // next_tokens = flat % V
    const bool keep = valid && (int64_t)ctok != eos;
    const uint64_t vb = __ballot(valid), kb = __ballot(keep);
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    // position in the stable order "not eos first" (beam_search.py:670-685: the first K non-eos candidates, in rank order, continue)
    const uint32_t rank = keep ? (uint32_t)__popcll(kb & below) : (uint32_t)__popcll(kb) + (uint32_t)__popcll(vb & ~kb & below);
    // ---- history of the step: candidate c = (its source row's tokens, its token) with its summed log-prob (beam_search.py:658-668) ----
    int64_t *htok = of_group(a.hist_tok, grp);
    if (htok && a.phase != 2) {
        float *hsc = of_group(a.hist_sc, grp);
        const uint32_t H = of_group(a.hist_H, grp), L = of_group(a.hist_L, grp);
        for (uint32_t c = wave; c < W2; c += nw) {
            const uint32_t sb = rl_u32(cbeam, c), tk = rl_u32(ctok, c);
            const float sc = __uint_as_float(rl_u32(__float_as_uint(csc), c));
            const uint64_t src = (uint64_t)q * K + sb;
            const uint64_t slot = (uint64_t)q_in_grp * H + a.hist_off + c;
            int64_t *dst = htok + slot * L;
            if (lane < t && lane < L) dst[lane] = a.ids[src * a.ids_stride + lane];
            if (lane == 0) { if (t < L) dst[t] = (int64_t)tk; hsc[slot] = sc; }
        }
    }
    // ---- the new beams: everything that is read from the query's old rows, then a barrier, then the stores ----
    // (wave w takes new beams w and w + nw: two explicit sets of registers, no indexed arrays)
    struct NewBeam { uint32_t par, tok; float sc; int64_t id; int32_t anc; };
    auto pick = [&](uint32_t j, NewBeam &b) {
        b.par = 0; b.tok = 0; b.sc = 0.f; b.id = 0; b.anc = 0;
        if (j >= K) return;
        const uint64_t m = __ballot(valid && rank == j);
        const uint32_t c = m ? (uint32_t)__ffsll((unsigned long long)m) - 1u : 0u;
        b.par = rl_u32(cbeam, c); b.tok = rl_u32(ctok, c);
        b.sc = __uint_as_float(rl_u32(__float_as_uint(csc), c));
        const uint64_t src = (uint64_t)q * K + b.par;
        if (lane < t) b.id = a.ids[src * a.ids_stride + lane];
        if (a.anc && lane < a.anc_T) b.anc = a.anc[(uint64_t)lane * a.anc_rows + src];
    };
    auto put = [&](uint32_t j, const NewBeam &b) {
        if (j >= K) return;
        const uint64_t r = (uint64_t)q * K + j;
        if (lane < t) a.ids[r * a.ids_stride + lane] = b.id;
        if (a.anc && lane < a.anc_T) a.anc[(uint64_t)lane * a.anc_rows + r] = b.anc;
        if (lane == 0) {
            a.ids[r * a.ids_stride + t] = (int64_t)b.tok;
            a.beam_scores[r] = b.sc;
            a.beam_idx[r] = (int64_t)((uint64_t)q * K + b.par);
            if (a.tokens_out) a.tokens_out[r] = (int64_t)b.tok;
        }
    };
    NewBeam b0, b1;
    if (a.phase != 2) {
        pick(wave, b0);
        pick(wave + nw, b1);
        __syncthreads();
        put(wave, b0);
        put(wave + nw, b1);
    } else {
        // chains only: the new rows' sources and tokens as the bookkeeping launch left them
        auto reload = [&](uint32_t j, NewBeam &b) {
            b.par = 0; b.tok = 0; b.sc = 0.f; b.id = 0; b.anc = 0;
            if (j >= K) return;
            const uint64_t r = (uint64_t)q * K + j;
            b.par = (uint32_t)((uint64_t)as_const(a.beam_idx)[r] - (uint64_t)q * K);
            b.tok = (uint32_t)as_const(a.ids)[r * a.ids_stride + t];
        };
        reload(wave, b0);
        reload(wave + nw, b1);
    }
    if (!a.chain || a.phase == 1) return;
    // ---- the chains of the next constraint call ----
    const int64_t stop = of_group(a.grp_stop, grp);
    ExpCounters ctr{0, 0, 0, 0};
    auto set_next_bit = [&](uint64_t r, int64_t tk) {
        if (tk >= 0 && (uint64_t)tk < a.vocab) atomicOr(&a.bits_next[r * a.words_per_row + ((uint64_t)tk >> 5)], 1u << (tk & 31));
    };
    auto chain = [&](uint32_t j, const NewBeam &b) {
        if (j >= K) return;
        const uint64_t r = (uint64_t)q * K + j, prow = a.st_base + (uint64_t)q * K + b.par;
        const int64_t tok = (int64_t)b.tok;
        const uint64_t sym = (uint64_t)(tok + a.shift);
        const bool dead = tok == eos || tok == a.pad_id;
        // (what the previous launches left: constant here, scalar loads where the address is wave-uniform)
        const uint32_t pm = as_const(a.lm_in)[prow];
        uint64_t count = 0, probes = 0;
        uint32_t m = LIST_RANGE_MODE;            // this row's list length, if it has (or gets) one
        uint64_t lo = 0, hi = 0;
        // lane k holds entries k, k + 64, ... of the list: positions, then the symbols in front of them (0: no entry)
        uint64_t pp[LIST_PER_LANE];
        uint32_t sv[LIST_PER_LANE];
#pragma unroll
        for (uint32_t i = 0; i < LIST_PER_LANE; i++) { pp[i] = 0; sv[i] = 0; }
        uint32_t slot[LIST_PER_LANE];
        uint64_t sel = 0;                        // bit i: entry i of this lane belongs to the new list
        if (pm != LIST_RANGE_MODE) {
            // ---- list mode: filter the source row's list by the token, step the survivors one position back ----
            count = pm;
            uint32_t ps[LIST_PER_LANE];
#pragma unroll
            for (uint32_t i = 0; i < LIST_PER_LANE; i++) {          // every load of the list is issued before any is looked at
                ps[i] = 0;
                if (64 * i < pm) {                                  // (uniform: whole chunks beyond the list are skipped)
                    const uint32_t idx = lane + 64 * i;
                    if (idx < pm) { pp[i] = a.lp_in[prow * LIST_MAX + idx]; ps[i] = a.ls_in[prow * LIST_MAX + idx]; }
                }
            }
            uint32_t base = 0;
#pragma unroll
            for (uint32_t i = 0; i < LIST_PER_LANE; i++) {
                slot[i] = 0;
                if (64 * i < pm) {
                    const bool match = lane + 64 * i < pm && (uint64_t)ps[i] == sym;
                    const uint64_t bal = __ballot(match);
                    slot[i] = base + lane_rank_in(bal);
                    base += (uint32_t)__popcll(bal);
                    sel |= (uint64_t)match << i;
                }
            }
            m = base;
#pragma unroll
            for (uint32_t i = 0; i < LIST_PER_LANE; i++) {          // ONE dependent access: the survivors' new symbols, gathered together
                // (cyclic, as the BWT is: the symbol in front of position 0 is text[n-1] -- the sentinel in every index this library builds,
                //  which no search symbol equals; the range -> list conversion below reads it the same way)
                if ((sel >> i) & 1) { pp[i] = pp[i] ? pp[i] - 1 : ix.n - 1; sv[i] = (uint32_t)text_at(ix, pp[i] ? pp[i] - 1 : ix.n - 1); }
            }
            if (a.probe_counter && lane == 0) probes += ((uint64_t)pm * 8 + 127) / 128 + ((uint64_t)pm * 4 + 127) / 128 + m;
        } else {
            // ---- range mode: one backward-search step from the source row's kept interval ----
            uint64_t l = as_const(a.st_in)[2 * prow], rr = as_const(a.st_in)[2 * prow + 1];
            count = (rr + 1) - l;
            bs_step(ix, sym, l, rr, l, rr, &probes);
            if (lane == 0) { a.st_out[2 * r] = l; a.st_out[2 * r + 1] = rr; }
            lo = l; hi = rr + 1;
            if (hi > ix.n) hi = ix.n;
            const uint64_t cnt = hi > lo ? hi - lo : 0;
            if (cnt <= LIST_MAX && rr + 1 <= ix.n) {        // (an interval that reaches past the last row -- quirk Q1 -- keeps its exact count as an interval)
                // the interval has become small: from here on the row carries its suffix-array rows' text positions
                m = (uint32_t)cnt;
#pragma unroll
                for (uint32_t i = 0; i < LIST_PER_LANE; i++) {
                    slot[i] = lane + 64 * i;
                    if (64 * i < m && lane + 64 * i < m) { pp[i] = sa_at(ix, lo + lane + 64 * i); sel |= 1ull << i; }
                }
#pragma unroll
                for (uint32_t i = 0; i < LIST_PER_LANE; i++)
                    if ((sel >> i) & 1) sv[i] = pp[i] ? (uint32_t)text_at(ix, pp[i] - 1) : (uint32_t)text_at(ix, ix.n - 1);
                if (a.probe_counter && lane == 0) probes += ((uint64_t)m * 4 + 127) / 128 + m;
            }
        }
        if (m != LIST_RANGE_MODE) {
#pragma unroll
            for (uint32_t i = 0; i < LIST_PER_LANE; i++)
                if ((sel >> i) & 1) { a.lp_out[r * LIST_MAX + slot[i]] = pp[i]; a.ls_out[r * LIST_MAX + slot[i]] = sv[i]; }
        }
        if (lane == 0) a.lm_out[r] = m;
        // ---- class (beam_search.py:87-131): a finished row counts 0 ----
        const uint64_t class_count = dead ? 0 : count;
        int64_t single = -1;
        bool expand = false;
        if (stop > 0 && (int64_t)class_count <= stop) single = eos;
        else if (dead) single = a.pad_id;
        else expand = true;
        RowPre pr;
        pr.lo = lo; pr.hi = hi; pr.single = single; pr.expand = 0u; pr.child_mask = 0u;
        uint32_t model = 0;
        if (a.probe_counter && !dead) model = ix.levels * (of_group(a.grp_nff, grp) + (uint32_t)a.cur_len);      // the reference re-searches the whole prefix
        if (!expand || m != LIST_RANGE_MODE) {
            // nothing is left for k_constrain: the row's tokens go into the next call's bitmap here
            if (expand) {
#pragma unroll
                for (uint32_t i = 0; i < LIST_PER_LANE; i++)
                    if (((sel >> i) & 1) && sv[i] != 0) set_next_bit(r, (int64_t)sv[i] - a.shift);
                if (a.probe_counter) {
                    // (measurement mode: the distinct tokens of the row = the bits this wave has just set in the row's words)
                    __threadfence();
                    uint32_t k = 0;
                    for (uint64_t w = lane; w < a.words_per_row; w += 64) k += (uint32_t)__popc(__hip_atomic_load(&a.bits_next[r * a.words_per_row + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    for (uint32_t o = 32; o; o >>= 1) k += __shfl_xor(k, o);
                    model += model_nodes_of_k(k, ix.levels);
                }
            } else if (lane == 0) {
                set_next_bit(r, single);
            }
            if (a.always_allow_eos && lane == 0) set_next_bit(r, eos);
            pr.single = ROW_DONE;
        } else {
            // a wide row: the root node split over the sixteen top digits, for the (row, top digit) waves of k_constrain
            const bool split = hi > lo;
            const uint32_t e = lane & 1, d = lane >> 1;
            uint64_t qv = 0;
            if (split && lane < 32) qv = wm_step(ix, 0, e ? hi : lo, d);
            const uint64_t qo = (uint64_t)dpp_xor1((uint32_t)qv) | ((uint64_t)dpp_xor1((uint32_t)(qv >> 32)) << 32);
            const uint64_t bal = __ballot(lane < 32 && e == 0 && qo > qv);
            if (lane < 32) a.pre_child[(r * FMI_ARITY + d) * 2 + e] = qv;
            uint32_t em = 0;
#pragma unroll
            for (uint32_t x = 0; x < 16; x++) em |= (uint32_t)((bal >> (2 * x)) & 1ull) << x;
            pr.expand = 1u; pr.child_mask = em;
            if (a.probe_counter && lane == 0) {
                probes += split ? ((lo >> FMI_BLOCK_SHIFT) != (hi >> FMI_BLOCK_SHIFT) ? 2u : 1u) : 0u;
                model += split ? model_nodes(em, 0, FMI_DIGIT_BITS * ix.dlevels - ix.levels) : 0u;
            }
        }
        if (lane == 0) {
            a.pre_rows[r] = pr;
            if (a.probe_counter) { ctr.probes += (uint32_t)probes; ctr.model += model; }
        }
    };
    chain(wave, b0);
    chain(wave + nw, b1);
    if (a.probe_counter) flush_counters(a.probe_counter, ctr);
}

// the two symbol bitmaps of the table calls (they alternate: each call's second launch zeroes the other one), grown on demand
static int reserve_sym_bits(fmi *h, uint64_t rows, hipStream_t st)
{
    const uint64_t words = ((uint64_t)top_digits(h) << (FMI_DIGIT_BITS * (h->dlevels - 1))) / 32 + 1;
    if (h->sym_bits && rows <= h->sym_rows && words == h->sym_row_words) return FMI_OK;
    const uint64_t cap = std::max<uint64_t>(rows, h->ws_rows);
    if (cap * words * 8 > (1ull << 30)) return FMI_ERR_CAPACITY;       // (the caller takes the generic path)
    HIPCHK(hipStreamSynchronize(st));
    if (h->sym_bits) { HIPCHK(hipFree(h->sym_bits)); h->sym_bits = nullptr; }
    HIPCHK(hipMalloc(&h->sym_bits, cap * words * 8));
    HIPCHK(hipMemsetAsync(h->sym_bits, 0, cap * words * 8, st));
    h->sym_rows = cap; h->sym_row_words = words; h->sym_dirty_rows[0] = h->sym_dirty_rows[1] = 0;
    return FMI_OK;
}

static int allowed_bits_impl(fmi *h, hipStream_t st, uint64_t rows, uint64_t cur_len, const int64_t *d_ids,
                             uint32_t *d_bits, uint64_t vocab, int64_t shift, int64_t pad_id, const RowGroups &rg,
                             int64_t stop_at_count, int always_allow_eos,
                             uint64_t state_tag = 0, const int64_t *d_parent = nullptr, const uint32_t **bits_out = nullptr,
                             uint64_t ids_stride = 0, bool allow_chain = false, uint64_t dropped_rows = 0)
{
    if (cur_len < 2) { fmi_set_error("cur_len must be >= 2 (cur_len == 1 is the constant occurring_distinct mask, beam_search.py:73-77)"); return FMI_ERR_ARG; }
    if (rg.n < 1 || rg.n > MAX_ROW_GROUPS) { fmi_set_error("1..%d row groups per call", MAX_ROW_GROUPS); return FMI_ERR_UNSUPPORTED; }
    if (rows > h->ws_rows) { int rc = fmi_dev_reserve(h, rows); if (rc) return rc; }
    const uint64_t wpr = (vocab + 31) / 32;
    ConstrainArgs a{};
    a.rows = (uint32_t)rows; a.ndig0 = top_digits(h);
    if ((rows + 8) * a.ndig0 > 0x7fffffffull) { fmi_set_error("too many rows in one call"); return FMI_ERR_CAPACITY; }
    a.cur_len = cur_len; a.ids = d_ids; a.ids_stride = ids_stride ? ids_stride : cur_len; a.shift = shift; a.pad_id = pad_id;
    uint64_t first = 0;
    for (uint32_t g = 0; g < MAX_ROW_GROUPS; g++) {
        a.grp_first[g] = g < rg.n ? (uint32_t)first : 0xffffffffu;
        if (g >= rg.n) continue;
        if (rg.n_force[g] > MAX_FORCE) { fmi_set_error("force_decoding_from longer than %d", MAX_FORCE); return FMI_ERR_UNSUPPORTED; }
        a.grp_eos[g] = rg.eos[g];
        a.grp_stop[g] = rg.stop[g] >= 0 ? rg.stop[g] : stop_at_count;
        a.grp_ff[g].n = (uint32_t)rg.n_force[g];
        for (uint64_t i = 0; i < rg.n_force[g]; i++) a.grp_ff[g].tok[i] = rg.force[g][i];
        first += rg.rows[g];
    }
    if (first != rows) { fmi_set_error("the row groups hold %llu rows, the call %llu", (unsigned long long)first, (unsigned long long)rows); return FMI_ERR_ARG; }
    a.always_allow_eos = always_allow_eos; a.vocab = vocab; a.words_per_row = wpr;
    a.probe_counter = h->probe_count_enabled ? h->d_probe_counter : nullptr;
    // waves per workgroup: CONSTRAIN_WG waves that share their leaf-level nodes when the whole sub-tree of an item fits
    // the deferred expansion (dlevels 2..4: every real vocabulary below 2^16 symbols), else self-contained waves
    // (FmiOptions: the environment was read when the handle was created; "constrain_waves" 1 = the self-contained waves)
    const unsigned W = (h->opt.constrain_waves != 1 && h->dlevels >= 2 && h->dlevels <= 4) ? (unsigned)CONSTRAIN_WG : 1u;
    if (W > 1) a.groups = (uint32_t)((rows + W - 1) / W);
    const unsigned grid = W > 1 ? a.groups * a.ndig0 : (unsigned)(((rows + 7) & ~7ull) * a.ndig0);
    a.tstamp = h->dbg_tstamp && h->dbg_tstamp_cap >= (uint64_t)grid * W * 8 ? h->dbg_tstamp : nullptr;
    if (d_bits) {
        HIPCHK(hipMemsetAsync(d_bits, 0, rows * wpr * 4, st));
        a.bits = d_bits;
    } else {
        const int cur = (int)(h->ws_seq & 1), nxt = cur ^ 1;
        h->ws_seq++;
        a.bits = ws_bits(h, cur);
        // the other buffer still holds the previous call's bitmap: this launch clears it, unless that would
        // be more than a few stores per lane (a much larger previous call) -- then a memset does
        if (h->ws_dirty[nxt] > (uint64_t)grid * W * 64 * 8) HIPCHK(hipMemsetAsync(ws_bits(h, nxt), 0, h->ws_dirty[nxt] * 4, st));
        else if (h->ws_dirty[nxt]) { a.clear = ws_bits(h, nxt); a.clear_words = h->ws_dirty[nxt]; }
        h->ws_dirty[nxt] = 0;
        h->ws_dirty[cur] = rows * wpr;
    }
    if (bits_out) *bits_out = a.bits;
    h->last_bits = a.bits; h->last_bits_rows = rows; h->last_bits_wpr = wpr;
    // Chained call (fmi_dev_beam_step): the previous step's k_beam_advance has run every row's chain already -- kept range of its parent
    // -> one backward-search step with the token it was given -> class -> root node split -- and left the results where the
    // (row, top digit) waves of k_constrain pick them up.  This call is then ONE launch with no dependent access in front of the sub-trees,
    // and reads neither ids nor parents.  `dropped_rows`: rows that left the front of the loop since that step (their decode ended).
    const bool chained = allow_chain && state_tag && !d_bits && h->chain_tag == state_tag && h->chain_len == cur_len &&
                         rows + dropped_rows + h->state_base == h->chain_rows && h->opt.constrain_waves != 1 && h->dlevels >= 2 && h->dlevels <= 4 &&
                         !(h->dbg_tstamp);
    if (chained) {
        h->state_base += dropped_rows;
        h->state_len = cur_len; h->state_rows = rows;
        a.pre_rows = ws_pre_rows(h) + h->state_base; a.pre_child = ws_pre_child(h) + h->state_base * FMI_ARITY * 2;
        // the bitmap holds the list rows' tokens already (k_beam_advance wrote them under the numbering of the step before: rows that have
        // left since sit in front of this call's)
        const int ccur = (int)((h->ws_seq - 1) & 1);
        a.bits = ws_bits(h, ccur) + h->state_base * wpr;
        h->ws_dirty[ccur] = (h->state_base + rows) * wpr;
        h->bits_prefilled = 0;
        if (bits_out) *bits_out = a.bits;
        h->last_bits = a.bits;
        a.groups = (uint32_t)((rows + CONSTRAIN_WG_CHAINED - 1) / CONSTRAIN_WG_CHAINED);
        a.leave_early = (int)h->opt.leave_early;
        const unsigned cgrid = a.groups * a.ndig0;
        const bool ctimed = h->timing_enabled && h->ev_used < MAX_TIMED_LAUNCHES;
        if (ctimed) HIPCHK(hipEventRecord((hipEvent_t)h->ev_start[h->ev_used], st));
        void (*ck)(FmiDev, ConstrainArgs) = h->dev.nsb > 1 ? k_constrain<true, CONSTRAIN_WG_CHAINED> : k_constrain<false, CONSTRAIN_WG_CHAINED>;
        hipLaunchKernelGGL(ck, dim3(cgrid), dim3(64 * CONSTRAIN_WG_CHAINED), (size_t)constrain_lds_slots(h->dlevels, CONSTRAIN_WG_CHAINED) * 16, st, h->dev, a);
        HIPCHK(hipGetLastError());
        call_log_begin(h, FMI_CALL_CHAINED, cur_len, rows, ctimed);
        if (ctimed) { HIPCHK(hipEventRecord((hipEvent_t)h->ev_stop[h->ev_used], st)); h->ev_used++; }
        return call_log_end(h, st);
    }
    h->chain_tag = 0;
    h->state_base = 0;
    if (!d_bits && h->bits_prefilled) {
        // chains were run for a call that never came (the loop ended or changed its form): their tokens must not leak into this one
        HIPCHK(hipMemsetAsync(ws_bits(h, (int)((h->ws_seq - 1) & 1)), 0, std::min<uint64_t>(h->chain_rows, h->ws_rows) * wpr * 4, st));
    }
    if (!d_bits) h->bits_prefilled = 0;
    // (a call that keeps ranges starts a lineage of rows in range mode: no lists yet)
    if (state_tag && h->ws_list) HIPCHK(hipMemsetAsync(ws_list_len(h, h->state_flip ^ 1), 0xFF, rows * 4, st));
    // incremental prefix state: valid when the caller vouches (tag + parent rows) that this call extends,
    // by exactly one token, the rows of the previous call with the same tag
    // (fewer rows than the previous call: a loop over several decodes in lockstep dropped the finished ones; the parents
    //  still name rows of the previous call, whose kept ranges are all there)
    const bool inc = state_tag && d_parent && h->state_tag == state_tag && rows <= h->state_rows && h->state_len + 1 == cur_len;
    a.st_in = inc ? ws_state(h, h->state_flip) : nullptr;
    a.parent = d_parent;
    a.st_out = state_tag ? ws_state(h, h->state_flip ^ 1) : nullptr;
    if (state_tag) { h->state_tag = state_tag; h->state_rows = rows; h->state_len = cur_len; h->state_flip ^= 1; }
    else h->state_tag = 0;
    const bool timed = h->timing_enabled && h->ev_used < MAX_TIMED_LAUNCHES;
    if (timed) HIPCHK(hipEventRecord((hipEvent_t)h->ev_start[h->ev_used], st));
    const size_t lds = (size_t)constrain_lds_slots(h->dlevels, W) * 16;
    const bool sb = h->dev.nsb > 1;
    // The first constrained step of a decode (every prefix = its decode's forced prefix + one token) from the per-token node tables:
    // one flat, evenly cut leaf-level pass (k_constrain_table) instead of the expansion from the root.  Tables are built on first use
    // (once per index and forced prefix, synchronously: a few ms); a prefix whose table would be too large takes the generic path.
    // (a packed node carries 16 bits of symbol prefix: leaf-level nodes of up to five digit levels)
    if (h->opt.prefix_tables && cur_len == 2 && !inc && !a.tstamp && h->dlevels <= 5 && table_lds_bytes((uint32_t)rows) <= 64 * 1024) {
        TableArgs tb{};
        bool all = true;
        for (uint32_t g = 0; g < rg.n; g++) {
            if (a.grp_stop[g] > 0) { all = false; break; }      // (the count of the prefix without its last token: not tabulated)
            int trc = FMI_OK;
            const FmiPrefixTable *T = prefix_table_for(h, rg.force[g], rg.n_force[g], shift, vocab, &trc);
            if (trc) return trc;
            if (!T) { all = false; break; }
            tb.off[g] = T->d_off; tb.root[g] = T->d_root; tb.nodes[g] = (const uint4 *)T->d_nodes;
        }
        if (all && reserve_sym_bits(h, rows, st) != FMI_OK) { all = false; (void)hipGetLastError(); }
        if (all) {
            const int cur = h->sym_flip, oth = cur ^ 1;
            h->sym_flip = oth;
            uint32_t *buf[2] = {(uint32_t *)h->sym_bits, (uint32_t *)h->sym_bits + h->sym_rows * h->sym_row_words};
            tb.sym = (uint8_t *)buf[cur]; tb.sym_row_words = h->sym_row_words;
            void (*kt)(FmiDev, ConstrainArgs, TableArgs) = sb ? k_constrain_table<true> : k_constrain_table<false>;
            const unsigned tgrid = h->opt.table_grid > 0 ? (unsigned)std::min<int64_t>(h->opt.table_grid, 1 << 16) : TABLE_GRID;
            hipLaunchKernelGGL(kt, dim3(tgrid), dim3(TABLE_WG), table_lds_bytes((uint32_t)rows), st, h->dev, a, tb);
            const uint64_t oth_rows = h->sym_dirty_rows[oth];
            const uint64_t per_row = std::max<uint64_t>(wpr, h->sym_row_words);
            hipLaunchKernelGGL(k_table_bits, dim3((unsigned)blocks_for(per_row, 256), (unsigned)std::max<uint64_t>(rows, oth_rows)), dim3(256), 0, st,
                               a, (const uint32_t *)buf[cur], h->sym_row_words, buf[oth], (uint32_t)oth_rows);
            h->sym_dirty_rows[cur] = rows; h->sym_dirty_rows[oth] = 0;
            HIPCHK(hipGetLastError());
            call_log_begin(h, FMI_CALL_TABLE, cur_len, rows, timed);
            if (timed) { HIPCHK(hipEventRecord((hipEvent_t)h->ev_stop[h->ev_used], st)); h->ev_used++; }
            return call_log_end(h, st);
        }
    }
    // Row-first: once the prefixes are a few tokens long, nine of ten (row, top digit) items are empty and a call is the chain of
    // dependent accesses of a row (parent -> kept range -> one backward-search step -> root child), which every one of the 13 waves
    // of a row would repeat: ONE wave per row runs it first (k_constrain_rows), the item waves pick the result up with a single load
    // and the empty ones are gone a microsecond into the second launch.  Wide rows (the first constrained steps of a decode) gain
    // nothing from the extra launch: the host picks by prefix length (option row_first = 0 / 1 forces either; same results).
    uint64_t longest = 0;
    for (uint32_t g = 0; g < rg.n; g++) longest = std::max<uint64_t>(longest, rg.n_force[g] + (cur_len - 1));
    a.leave_early = (int)h->opt.leave_early;
    // measured (profiles/r3_rowfirst_ab.txt): wins from 3-token prefixes on at 300 rows (29.6 -> 28.1, 26.3 -> 24.7, 18.9 -> 17.5 us at 3 / 4 / 6
    // tokens; 2 tokens 44.5 -> 45.3), from 2 tokens on at 600 rows (81 -> 79, 52 -> 45, 40 -> 33, 33 -> 21 us at 2 / 3 / 4 / 6); a call of
    // single-token prefixes loses 2.7 us to the extra launch (62.9 -> 65.6).  (A third form -- the rows append their non-empty items to
    // ONE list that a fixed grid walks -- measured slower everywhere, 22.1 vs 17.5 us on the narrowest call: 600 returning atomics on one
    // counter cost more than the empty waves they save.)
    const uint64_t rf_from = h->opt.row_first_from >= 0 ? (uint64_t)h->opt.row_first_from : (rows >= 512 ? 2 : 3);
    const bool row_first = W > 1 && (h->opt.row_first >= 0 ? h->opt.row_first != 0 : longest >= rf_from);
    if (row_first) {
        a.pre_rows = ws_pre_rows(h); a.pre_child = ws_pre_child(h);
        hipLaunchKernelGGL(k_constrain_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, h->dev, a);
    }
    void (*kern)(FmiDev, ConstrainArgs) = W > 1 ? (sb ? k_constrain<true, CONSTRAIN_WG> : k_constrain<false, CONSTRAIN_WG>)
                                                : (sb ? k_constrain<true, 1> : k_constrain<false, 1>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * W), lds, st, h->dev, a);
    HIPCHK(hipGetLastError());
    call_log_begin(h, row_first ? FMI_CALL_ROW_FIRST : FMI_CALL_GENERIC, cur_len, rows, timed);
    if (timed) { HIPCHK(hipEventRecord((hipEvent_t)h->ev_stop[h->ev_used], st)); h->ev_used++; }
    return call_log_end(h, st);
}

extern "C" int fmi_dev_allowed_bits(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len, const int64_t *d_input_ids,
                                    uint32_t *d_bits, uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                                    const int64_t *force_from, uint64_t n_force, int64_t stop_at_count, int always_allow_eos)
{
    int rc = need_device(h); if (rc) return rc;
    if (rows == 0) return FMI_OK;
    return allowed_bits_impl(h, (hipStream_t)stream, rows, cur_len, d_input_ids, d_bits, vocab, shift, pad_id,
                             RowGroups::one(rows, eos_id, force_from, n_force), stop_at_count, always_allow_eos);
}

extern "C" int fmi_dev_allowed_bits_step(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len, const int64_t *d_input_ids,
                                         uint32_t *d_bits, uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                                         const int64_t *force_from, uint64_t n_force, int64_t stop_at_count, int always_allow_eos,
                                         uint64_t state_tag, const int64_t *d_parent_rows, const uint32_t **d_bits_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (rows == 0) return FMI_OK;
    if ((vocab + 31) / 32 > WS_BITS_WORDS) { fmi_set_error("vocab %llu too large", (unsigned long long)vocab); return FMI_ERR_UNSUPPORTED; }
    return allowed_bits_impl(h, (hipStream_t)stream, rows, cur_len, d_input_ids, d_bits, vocab, shift, pad_id,
                             RowGroups::one(rows, eos_id, force_from, n_force), stop_at_count, always_allow_eos, state_tag, d_parent_rows, d_bits_out);
}

extern "C" int fmi_dev_debug_timestamps(fmi_t *h, uint64_t *d_buf, uint64_t n_words)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    h->dbg_tstamp = d_buf; h->dbg_tstamp_cap = d_buf ? n_words : 0;
    return FMI_OK;
}

extern "C" int fmi_dev_debug_marks(fmi_t *h, uint32_t *marks)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    h->dbg_marks = marks;
    return FMI_OK;
}

extern "C" int fmi_dev_mark(void *stream, uint32_t *word, uint32_t value)
{
    if (!word) { fmi_set_error("null argument"); return FMI_ERR_ARG; }
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, (hipStream_t)stream, word, value);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

extern "C" int fmi_dev_constrain_scores(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len, const int64_t *d_input_ids,
                                        const float *d_in, float *d_out, uint64_t vocab, int64_t shift, int64_t pad_id,
                                        int64_t eos_id, const int64_t *force_from, uint64_t n_force, int64_t stop_at_count,
                                        int always_allow_eos)
{
    int rc = need_device(h); if (rc) return rc;
    if (rows == 0) return FMI_OK;
    const uint64_t wpr = (vocab + 31) / 32;
    if (wpr > WS_BITS_WORDS) { fmi_set_error("vocab %llu too large", (unsigned long long)vocab); return FMI_ERR_UNSUPPORTED; }
    if (rows > h->ws_rows) { rc = fmi_dev_reserve(h, rows); if (rc) return rc; }
    const uint32_t *bits = nullptr;
    rc = allowed_bits_impl(h, (hipStream_t)stream, rows, cur_len, d_input_ids, nullptr, vocab, shift, pad_id,
                           RowGroups::one(rows, eos_id, force_from, n_force), stop_at_count, always_allow_eos, 0, nullptr, &bits);
    if (rc) return rc;
    hipLaunchKernelGGL(k_apply_bits, dim3(blocks_for(vocab, 256 * 4), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       d_in, d_out, bits, rows, vocab, wpr);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

extern "C" int fmi_dev_constrained_topk(fmi_t *h, void *stream, uint64_t batch, uint64_t beams, uint64_t cur_len,
                                        const int64_t *d_input_ids, const float *d_logits, const float *d_beam_scores,
                                        uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id, const int64_t *force_from,
                                        uint64_t n_force, int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                                        void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc)
{
    return fmi_dev_constrained_topk_step(h, stream, batch, beams, cur_len, d_input_ids, d_logits, d_beam_scores, vocab, shift, pad_id,
                                         eos_id, force_from, n_force, stop_at_count, always_allow_eos, d_first_bits, d_scratch,
                                         scratch_bytes, d_top_idx, d_top_con, d_top_unc, 0, nullptr);
}

extern "C" int fmi_dev_constrained_topk_step(fmi_t *h, void *stream, uint64_t batch, uint64_t beams, uint64_t cur_len,
                                             const int64_t *d_input_ids, const float *d_logits, const float *d_beam_scores,
                                             uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id, const int64_t *force_from,
                                             uint64_t n_force, int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                                             void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc,
                                             uint64_t state_tag, const int64_t *d_parent_rows)
{
    return fmi_dev_constrained_topk_groups(h, stream, 1, &batch, &eos_id, force_from, &n_force, beams, cur_len, d_input_ids, d_logits,
                                           d_beam_scores, vocab, shift, pad_id, stop_at_count, always_allow_eos, d_first_bits, d_scratch,
                                           scratch_bytes, d_top_idx, d_top_con, d_top_unc, state_tag, d_parent_rows, nullptr);
}

static int topk_groups_impl(fmi_t *h, void *stream, uint64_t n_groups, const uint64_t *group_batch,
                            const int64_t *group_eos, const int64_t *group_force, const uint64_t *group_n_force,
                            uint64_t beams, uint64_t cur_len, const int64_t *d_input_ids, const float *d_logits,
                            const float *d_beam_scores, uint64_t vocab, int64_t shift, int64_t pad_id,
                            int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                            void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc,
                            uint64_t state_tag, const int64_t *d_parent_rows, const int64_t *group_stop_at_count,
                            uint64_t ids_stride, bool allow_chain, uint64_t dropped_rows, uint64_t force_slots);

extern "C" int fmi_dev_constrained_topk_groups(fmi_t *h, void *stream, uint64_t n_groups, const uint64_t *group_batch,
                                               const int64_t *group_eos, const int64_t *group_force, const uint64_t *group_n_force,
                                               uint64_t beams, uint64_t cur_len, const int64_t *d_input_ids, const float *d_logits,
                                               const float *d_beam_scores, uint64_t vocab, int64_t shift, int64_t pad_id,
                                               int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                                               void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc,
                                               uint64_t state_tag, const int64_t *d_parent_rows, const int64_t *group_stop_at_count)
{
    // one group: the caller's forced-prefix array as it is (fmi_dev_constrained_topk_step); several: MAX_FORCE slots per group
    return topk_groups_impl(h, stream, n_groups, group_batch, group_eos, group_force, group_n_force, beams, cur_len, d_input_ids, d_logits,
                            d_beam_scores, vocab, shift, pad_id, stop_at_count, always_allow_eos, d_first_bits, d_scratch, scratch_bytes,
                            d_top_idx, d_top_con, d_top_unc, state_tag, d_parent_rows, group_stop_at_count, 0, false, 0, n_groups == 1 ? 0 : MAX_FORCE);
}

static int topk_groups_impl(fmi_t *h, void *stream, uint64_t n_groups, const uint64_t *group_batch,
                            const int64_t *group_eos, const int64_t *group_force, const uint64_t *group_n_force,
                            uint64_t beams, uint64_t cur_len, const int64_t *d_input_ids, const float *d_logits,
                            const float *d_beam_scores, uint64_t vocab, int64_t shift, int64_t pad_id,
                            int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                            void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc,
                            uint64_t state_tag, const int64_t *d_parent_rows, const int64_t *group_stop_at_count,
                            uint64_t ids_stride, bool allow_chain, uint64_t dropped_rows, uint64_t force_slots)
{
    int rc = need_device(h); if (rc) return rc;
    if (n_groups < 1 || n_groups > MAX_ROW_GROUPS || !group_batch || !group_eos || !group_n_force) {
        fmi_set_error("fmi_dev_constrained_topk_groups: 1..%d groups with batch / eos / n_force arrays", MAX_ROW_GROUPS); return FMI_ERR_ARG;
    }
    uint64_t batch = 0;
    RowGroups rg;
    rg.n = (uint32_t)n_groups;
    for (uint64_t g = 0; g < n_groups; g++) {
        batch += group_batch[g];
        rg.rows[g] = group_batch[g] * beams; rg.eos[g] = group_eos[g]; rg.n_force[g] = group_n_force[g];
        if (group_stop_at_count) rg.stop[g] = group_stop_at_count[g] < 0 ? 0 : group_stop_at_count[g];
        rg.force[g] = group_force ? group_force + g * force_slots : nullptr;
        if (group_n_force[g] && !group_force) { fmi_set_error("group %llu: n_force without tokens", (unsigned long long)g); return FMI_ERR_ARG; }
    }
    const uint64_t rows = batch * beams, want = 2 * beams;
    if (rows == 0) return FMI_OK;
    if (want > TOPK_MAX || beams > 64) { fmi_set_error("num_beams %llu: at most %d", (unsigned long long)beams, TOPK_MAX / 2); return FMI_ERR_UNSUPPORTED; }
    const uint64_t wpr = (vocab + 31) / 32;
    if (wpr > WS_BITS_WORDS) { fmi_set_error("vocab %llu too large", (unsigned long long)vocab); return FMI_ERR_UNSUPPORTED; }
    // scratch: row_max[rows] row_lsum[rows] row_lp[rows*want] (f32) | row_tok[rows*want] (i32) | row_cnt[rows] (u32)
    const uint64_t need = rows * 4 * (2 + 2 * want + 1);
    if (!d_scratch || scratch_bytes < need) { fmi_set_error("scratch too small: need %llu bytes", (unsigned long long)need); return FMI_ERR_CAPACITY; }
    float *row_max = (float *)d_scratch, *row_lsum = row_max + rows, *row_lp = row_lsum + rows;
    int32_t *row_tok = (int32_t *)(row_lp + rows * want);
    uint32_t *row_cnt = (uint32_t *)(row_tok + rows * want);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t *bits;
    uint32_t broadcast = 0;
    if (cur_len < 2) {
        if (!d_first_bits) { fmi_set_error("cur_len == 1 needs the occurring_distinct bitmap"); return FMI_ERR_ARG; }
        bits = d_first_bits; broadcast = 1;      // the constant first-step mask (beam_search.py:73-77), the same for every group
        h->state_tag = 0; h->chain_tag = 0; h->state_base = 0;
        h->last_bits = nullptr; h->last_bits_rows = 0;
    } else {
        if (rows > h->ws_rows) { rc = fmi_dev_reserve(h, rows); if (rc) return rc; }
        rc = allowed_bits_impl(h, st, rows, cur_len, d_input_ids, nullptr, vocab, shift, pad_id, rg,
                               stop_at_count, always_allow_eos, state_tag, d_parent_rows, &bits, ids_stride, allow_chain, dropped_rows);
        if (rc) return rc;
    }
    // (FmiOptions) tests: topk_narrow 0 sends every row down the wide-row path; topk_legacy: wide rows skip the thread-maxima bound
    const uint32_t narrow_max = h->opt.topk_narrow >= 0 ? std::min<uint32_t>((uint32_t)h->opt.topk_narrow, TOPK_NARROW) : TOPK_NARROW;
    const uint32_t pick_flags = h->opt.topk_legacy ? (uint32_t)PICK_NO_PREFILTER : 0u;
    hipLaunchKernelGGL(k_row_pick, dim3((unsigned)rows), dim3(PICK_BLOCK), (wpr + 1) * 4, st, d_logits, bits, wpr, broadcast, vocab,
                       (uint32_t)want, row_max, row_lsum, row_tok, row_lp, row_cnt, narrow_max, pick_flags);
    hipLaunchKernelGGL(k_query_merge, dim3((unsigned)batch), dim3(MERGE_BLOCK), 0, st, d_logits, bits, wpr, broadcast, vocab, (uint32_t)beams,
                       (uint32_t)want, d_beam_scores, row_max, row_lsum, row_tok, row_lp, row_cnt, d_top_idx, d_top_con, d_top_unc);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

// One decode step of the beam loop behind the model's forward (sealfm.h: fmi_beam_step_t): constraint + log-softmax + top-2K + merge
// (fmi_dev_constrained_topk_groups), then k_beam_advance.
extern "C" int fmi_dev_beam_step(fmi_t *h, void *stream, const fmi_beam_step_t *s)
{
    int rc = need_device(h); if (rc) return rc;
    if (!s || s->struct_bytes != sizeof(fmi_beam_step_t)) { fmi_set_error("fmi_dev_beam_step: struct_bytes must be sizeof(fmi_beam_step_t)"); return FMI_ERR_ARG; }
    if (s->n_groups < 1 || s->n_groups > MAX_ROW_GROUPS) { fmi_set_error("fmi_dev_beam_step: 1..%d groups", MAX_ROW_GROUPS); return FMI_ERR_ARG; }
    if (s->beams < 1 || s->beams > ADV_MAX_WAVES * ADV_ITERS || 2 * s->beams > TOPK_MAX) { fmi_set_error("fmi_dev_beam_step: num_beams %llu: at most %d", (unsigned long long)s->beams, TOPK_MAX / 2); return FMI_ERR_UNSUPPORTED; }
    if (s->cur_len < 1 || s->cur_len >= 63 || s->ids_stride < s->cur_len + 1) { fmi_set_error("fmi_dev_beam_step: cur_len %llu needs an ids row of cur_len + 1 <= 63 slots", (unsigned long long)s->cur_len); return FMI_ERR_ARG; }
    if (!s->d_ids || !s->d_logits || !s->d_beam_scores || !s->d_top_idx || !s->d_top_con || !s->d_top_unc || !s->d_beam_idx) { fmi_set_error("fmi_dev_beam_step: null buffer"); return FMI_ERR_ARG; }
    if (s->d_anc && s->anc_positions > 64) { fmi_set_error("fmi_dev_beam_step: at most 64 decoder positions in the ancestry table"); return FMI_ERR_UNSUPPORTED; }
    uint64_t batch = 0;
    for (uint64_t g = 0; g < s->n_groups; g++) batch += s->group_batch[g];
    if (batch == 0) return FMI_OK;
    hipStream_t st = (hipStream_t)stream;
    rc = topk_groups_impl(h, stream, s->n_groups, s->group_batch, s->group_eos, &s->group_force[0][0], s->group_n_force, s->beams, s->cur_len, s->d_ids,
                          s->d_logits, s->d_beam_scores, s->vocab, s->shift, s->pad_id, 0, s->always_allow_eos, s->d_first_bits, s->d_scratch, s->scratch_bytes,
                          s->d_top_idx, s->d_top_con, s->d_top_unc, s->state_tag, nullptr, s->group_stop, s->ids_stride, true, s->dropped_rows, MAX_FORCE);
    if (rc) return rc;
    const uint64_t rows = batch * s->beams;
    AdvanceArgs a{};
    a.batch = (uint32_t)batch; a.beams = (uint32_t)s->beams; a.cur_len = s->cur_len; a.ids = s->d_ids; a.ids_stride = s->ids_stride;
    a.vocab = s->vocab; a.shift = s->shift; a.pad_id = s->pad_id;
    a.top_idx = s->d_top_idx; a.top_unc = s->d_top_unc; a.beam_scores = s->d_beam_scores; a.beam_idx = s->d_beam_idx; a.tokens_out = s->d_tokens_out;
    a.anc = s->d_anc; a.anc_rows = s->anc_rows; a.anc_T = (uint32_t)s->anc_positions;
    uint64_t first_q = 0;
    for (uint32_t g = 0; g < MAX_ROW_GROUPS; g++) {
        a.grp_first_q[g] = g < s->n_groups ? (uint32_t)first_q : 0xffffffffu;
        if (g >= s->n_groups) continue;
        a.grp_eos[g] = s->group_eos[g]; a.grp_stop[g] = s->group_stop[g] < 0 ? 0 : s->group_stop[g]; a.grp_nff[g] = (uint32_t)s->group_n_force[g];
        a.hist_tok[g] = s->d_hist_tok[g]; a.hist_sc[g] = s->d_hist_sc[g]; a.hist_H[g] = (uint32_t)s->hist_H[g]; a.hist_L[g] = (uint32_t)s->hist_L[g];
        if (a.hist_tok[g] && (!a.hist_sc[g] || s->hist_off + 2 * s->beams > s->hist_H[g])) { fmi_set_error("fmi_dev_beam_step: history of group %u too small", g); return FMI_ERR_ARG; }
        first_q += s->group_batch[g];
    }
    a.hist_off = (uint32_t)s->hist_off;
    // the chains of the next call: only behind a constraint call of this decode whose kept ranges are in the workspace (cur_len >= 2)
    const bool chain = s->chain_next && s->state_tag && s->cur_len >= 2 && h->state_tag == s->state_tag && h->state_len == s->cur_len &&
                       h->state_rows == rows && h->opt.chain_steps && h->opt.constrain_waves != 1 && h->dlevels >= 2 && h->dlevels <= 4;
    if (chain) {
        a.chain = 1;
        a.always_allow_eos = s->always_allow_eos;
        const int in = h->state_flip, out = h->state_flip ^ 1;
        a.st_in = ws_state(h, in); a.st_base = h->state_base; a.st_out = ws_state(h, out);
        a.lm_in = ws_list_len(h, in); a.lp_in = ws_list_pos(h, in); a.ls_in = ws_list_sym(h, in);
        a.lm_out = ws_list_len(h, out); a.lp_out = ws_list_pos(h, out); a.ls_out = ws_list_sym(h, out);
        a.pre_rows = ws_pre_rows(h); a.pre_child = ws_pre_child(h);
        // the bitmap the NEXT call fills: this step's call cleared it (allowed_bits_impl: `clear`), so it is zero now
        a.bits_next = ws_bits(h, (int)(h->ws_seq & 1)); a.words_per_row = (s->vocab + 31) / 32;
        a.probe_counter = h->probe_count_enabled ? h->d_probe_counter : nullptr;
        h->state_flip ^= 1; h->state_base = 0; h->state_len = s->cur_len + 1;
        h->chain_tag = s->state_tag; h->chain_len = s->cur_len + 1; h->chain_rows = rows;
        h->bits_prefilled = 1;
    } else {
        h->chain_tag = 0;
    }
    const unsigned nw = (unsigned)std::min<uint64_t>(s->beams, ADV_MAX_WAVES);
    // The product runs ONE launch.  Measurement passes (fmi_dev_enable_timing / fmi_dev_call_log) run the bookkeeping and the chains as two
    // launches of the same kernel, so that an event pair / a counter read-out brackets exactly the index work (the chains, the list
    // steps): that costs the chains a launch of their own -- the figure they are charged with is an upper bound of what they cost the product.
    const bool apart = chain && (h->timing_enabled || h->call_log_enabled) && h->opt.advance_apart;
    for (int phase = apart ? 1 : 0; phase <= (apart ? 2 : 0); phase++) {
        a.phase = phase;
        const bool timed = h->timing_enabled && h->ev_used < MAX_TIMED_LAUNCHES;
        if (timed) HIPCHK(hipEventRecord((hipEvent_t)h->ev_start[h->ev_used], st));
        hipLaunchKernelGGL(k_beam_advance, dim3((unsigned)batch), dim3(64 * nw), 0, st, h->dev, a);
        HIPCHK(hipGetLastError());
        // (a record of the per-call log of its own: cur_len = the call it prepares; bench.py adds the chains to that call)
        call_log_begin(h, (chain && phase != 1) ? FMI_CALL_ADVANCE_CHAIN : FMI_CALL_ADVANCE, s->cur_len + 1, rows, timed);
        if (timed) { HIPCHK(hipEventRecord((hipEvent_t)h->ev_stop[h->ev_used], st)); h->ev_used++; }
        rc = call_log_end(h, st);
        if (rc) return rc;
    }
    return FMI_OK;
}

extern "C" const uint32_t *fmi_dev_last_constraint_bits(fmi_t *h, uint64_t *rows_out, uint64_t *words_per_row_out)
{
    if (!h) return nullptr;
    if (rows_out) *rows_out = h->last_bits_rows;
    if (words_per_row_out) *words_per_row_out = h->last_bits_wpr;
    return h->last_bits;
}

extern "C" int fmi_dev_locate(fmi_t *h, void *stream, uint64_t n, const uint64_t *d_rows, uint64_t *d_pos_out, uint64_t *d_doc_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (n == 0) return FMI_OK;
    if (!h->dev.sa_lo) { fmi_set_error("rank/select-only index: no suffix array resident"); return FMI_ERR_STATE; }
    hipLaunchKernelGGL(k_locate, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, h->dev, n, d_rows, d_pos_out, d_doc_out);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

extern "C" int fmi_dev_locate_ranges(fmi_t *h, void *stream, uint64_t n_ranges, const uint64_t *d_lo, const uint64_t *d_hi,
                                     uint64_t max_per_range, const uint64_t *d_out_offsets, uint64_t total,
                                     uint64_t *d_pos_out, uint64_t *d_doc_out)
{
    (void)d_hi; (void)max_per_range;   // already folded into d_out_offsets by the caller
    int rc = need_device(h); if (rc) return rc;
    if (total == 0 || n_ranges == 0) return FMI_OK;
    if (!h->dev.sa_lo) { fmi_set_error("rank/select-only index: no suffix array resident"); return FMI_ERR_STATE; }
    hipLaunchKernelGGL(k_locate_ranges, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, h->dev, n_ranges,
                       d_lo, d_out_offsets, total, d_pos_out, d_doc_out);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

extern "C" int fmi_dev_get_docs(fmi_t *h, void *stream, uint64_t n_docs, const uint64_t *d_docs, const uint64_t *d_out_offsets,
                                int64_t shift, int64_t *d_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (n_docs == 0) return FMI_OK;
    if (!h->dev.doc_begin || !h->dev.text) { fmi_set_error("doc beginnings / text not resident"); return FMI_ERR_STATE; }
    hipLaunchKernelGGL(k_get_docs, dim3((unsigned)n_docs), dim3(64), 0, (hipStream_t)stream, h->dev, d_docs, d_out_offsets, shift, d_out);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

// ---- host-buffer wrappers (what each SWIG method call becomes) -------------

extern "C" int fmi_backward_search_step(fmi_t *h, uint64_t symbol, uint64_t low, uint64_t high, uint64_t out[2])
{
    int rc = need_device(h); if (rc) return rc;
    DevBuf b; if ((rc = b.alloc(5 * 8))) return rc;
    uint64_t in[3] = {symbol, low, high};
    uint64_t *d = b.as<uint64_t>();
    HIPCHK(hipMemcpy(d, in, 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bs_step, dim3(1), dim3(64), 0, 0, h->dev, (uint64_t)1, d, d + 1, d + 2, d + 3, d + 4);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, d + 3, 16, hipMemcpyDeviceToHost));
    return FMI_OK;
}

extern "C" int fmi_backward_search_multi_batch(fmi_t *h, uint64_t n_seq, const uint64_t *offsets, const uint64_t *symbols,
                                               uint64_t *lo_out, uint64_t *hi_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (n_seq == 0) return FMI_OK;
    hipStream_t sst; if ((rc = service_stream(h, &sst))) return rc;
    const uint64_t ntok = offsets[n_seq];
    DevBuf off, tok, res;
    if ((rc = off.alloc((n_seq + 1) * 8)) || (rc = tok.alloc(ntok * 8)) || (rc = res.alloc(n_seq * 16))) return rc;
    COPY_H2D(off.p, offsets, (n_seq + 1) * 8);
    if (ntok) COPY_H2D(tok.p, symbols, ntok * 8);
    hipLaunchKernelGGL((k_get_range<uint64_t, uint64_t>), dim3(blocks_for(n_seq, 64)), dim3(64), 0, sst, h->dev, n_seq,
                       off.as<uint64_t>(), tok.as<uint64_t>(), (int64_t)0, res.as<uint64_t>(), res.as<uint64_t>() + n_seq);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(lo_out, res.p, n_seq * 8, hipMemcpyDeviceToHost, sst));
    COPY_D2H(hi_out, res.as<uint64_t>() + n_seq, n_seq * 8);
    return FMI_OK;
}

extern "C" int fmi_backward_search_multi(fmi_t *h, const uint64_t *query, uint64_t len, uint64_t out[2])
{
    uint64_t offs[2] = {0, len};
    return fmi_backward_search_multi_batch(h, 1, offs, query, &out[0], &out[1]);
}

extern "C" int fmi_distinct_count_multi(fmi_t *h, uint64_t n, const uint64_t *lows, const uint64_t *highs,
                                        uint64_t *offsets_out, uint64_t *syms_out, uint64_t *cnts_out, uint64_t cap)
{
    int rc = need_device(h); if (rc) return rc;
    offsets_out[0] = 0;
    if (n == 0) return FMI_OK;
    const uint64_t nsym = h->max_sym + 1;
    // bounded dense scratch: process the intervals in chunks of rows
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (512ull << 20) / (nsym * 8)));
    DevBuf dense, items, rowk, offs, osym, ocnt;
    if ((rc = dense.alloc(chunk * nsym * 8)) || (rc = items.alloc(chunk * sizeof(ExpandItem))) ||
        (rc = rowk.alloc(chunk * 8)) || (rc = offs.alloc((chunk + 1) * 8))) return rc;
    std::vector<ExpandItem> hitems(chunk);
    std::vector<uint64_t> hk(chunk), hoff(chunk + 1);
    uint64_t written = 0;
    bool overflow = false;
    for (uint64_t c0 = 0; c0 < n; c0 += chunk) {
        const uint64_t m = std::min(chunk, n - c0);
        for (uint64_t i = 0; i < m; i++) {
            uint64_t lo = lows[c0 + i], hi = highs[c0 + i];
            if (hi > h->n) hi = h->n;               // rows past the end do not exist (reference: undefined)
            if (lo >= hi) { lo = hi = 0; }          // low == high -> empty (fm_index.cpp:81,99)
            hitems[i] = ExpandItem{lo, hi, (uint32_t)i, 0};
        }
        HIPCHK(hipMemcpy(items.p, hitems.data(), m * sizeof(ExpandItem), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(dense.p, 0, m * nsym * 8));
        EmitTarget tgt{}; tgt.dense = dense.as<uint64_t>(); tgt.dense_stride = nsym;
        rc = launch_expand_dense(h, 0, items.as<ExpandItem>(), m, tgt);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dense_count, dim3((unsigned)m), dim3(256), 0, 0, dense.as<uint64_t>(), nsym, nsym, rowk.as<uint64_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(hk.data(), rowk.p, m * 8, hipMemcpyDeviceToHost));
        hoff[0] = 0;
        for (uint64_t i = 0; i < m; i++) { hoff[i + 1] = hoff[i] + hk[i]; offsets_out[c0 + i + 1] = written + hoff[i + 1]; }
        const uint64_t tot = hoff[m];
        if (written + tot > cap || overflow || !syms_out) { overflow = true; written += tot; continue; }
        if (tot) {
            DevBuf ds, dc;
            if ((rc = ds.alloc(tot * 8)) || (rc = dc.alloc(tot * 8))) return rc;
            HIPCHK(hipMemcpy(offs.p, hoff.data(), (m + 1) * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_dense_compact, dim3((unsigned)m), dim3(256), 0, 0, dense.as<uint64_t>(), nsym, nsym,
                               offs.as<uint64_t>(), ds.as<uint64_t>(), cnts_out ? dc.as<uint64_t>() : (uint64_t *)nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy(syms_out + written, ds.p, tot * 8, hipMemcpyDeviceToHost));
            if (cnts_out) HIPCHK(hipMemcpy(cnts_out + written, dc.p, tot * 8, hipMemcpyDeviceToHost));
        }
        written += tot;
    }
    if (overflow) { fmi_set_error("output capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)written); return FMI_ERR_CAPACITY; }
    return FMI_OK;
}

extern "C" int fmi_locate(fmi_t *h, uint64_t n, const uint64_t *rows, uint64_t *pos_out, uint64_t *doc_out)
{
    int rc = need_device(h); if (rc) return rc;
    if (n == 0) return FMI_OK;
    if (doc_out && !h->dev.doc_begin) { fmi_set_error("doc beginnings not set"); return FMI_ERR_STATE; }
    if (!h->dev.sa_lo) { fmi_set_error("rank/select-only index: no suffix array resident"); return FMI_ERR_STATE; }
    hipStream_t sst; if ((rc = service_stream(h, &sst))) return rc;
    DevBuf b; if ((rc = b.alloc(n * 24))) return rc;
    uint64_t *d = b.as<uint64_t>();
    COPY_H2D(d, rows, n * 8);
    hipLaunchKernelGGL(k_locate, dim3(blocks_for(n, 256)), dim3(256), 0, sst, h->dev, n, d, d + n, doc_out ? d + 2 * n : (uint64_t *)nullptr);
    HIPCHK(hipGetLastError());
    if (doc_out) HIPCHK(hipMemcpyAsync(doc_out, d + 2 * n, n * 8, hipMemcpyDeviceToHost, sst));
    COPY_D2H(pos_out, d + n, n * 8);
    return FMI_OK;
}

extern "C" int fmi_extract_text(fmi_t *h, uint64_t begin, uint64_t end, uint64_t *out)
{
    int rc = need_device(h); if (rc) return rc;
    if (end <= begin) return FMI_OK;
    if (end > h->n) { fmi_set_error("extract_text: end %llu > size %llu", (unsigned long long)end, (unsigned long long)h->n); return FMI_ERR_ARG; }
    if (!h->dev.text) { fmi_set_error("rank/select-only index: no text resident"); return FMI_ERR_STATE; }
    const uint64_t m = end - begin;
    DevBuf b; if ((rc = b.alloc(m * 8))) return rc;
    hipLaunchKernelGGL(k_extract, dim3(blocks_for(m, 256)), dim3(256), 0, 0, h->dev, begin, end, b.as<uint64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, b.p, m * 8, hipMemcpyDeviceToHost));
    return FMI_OK;
}

// GPU construction lives in fmi_build_gpu.hip
