// gfx950 (MI355X / CDNA4) kernels of libsealfm.so and the C-ABI query entry points.
//
// All work here is HBM-latency/bandwidth bound integer work: dependent 128-byte
// gathers into a 16-ary wavelet matrix (one 128-B line per rank probe, four symbol
// bits per probe), 5-byte gathers into the suffix array, binary searches over
// doc boundaries.  No MFMA.
// Wave size is 64 throughout.
#include <hip/hip_runtime.h>

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
    if (i > ix.n) return (ix.C[c + 1] - ix.C[c]) + ix.q1[c];
    if (i == 0) return 0;
    return wm_rank_sym(ix, c, i, probes);
}

// sdsl backward_search(csa, l, r, c, l_res, r_res) on the inclusive [l, r]
__device__ __forceinline__ void bs_step(const FmiDev &ix, uint64_t c, uint64_t l, uint64_t r,
                                        uint64_t &l_res, uint64_t &r_res, uint64_t *probes)
{
    const bool absent = (c > ix.max_sym) || (ix.C[c + 1] == ix.C[c]);
    if (absent && c > 0) { l_res = 1; r_res = 0; return; }
    const uint64_t cb = ix.C[c];
    uint64_t rl, rr;
    const uint64_t j = r + 1;
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
// fm_index.cpp:78-109).  One wavefront per work item (a hex-wavelet-matrix node
// [lo, hi) at some level with its symbol prefix).  The wave keeps its frontier
// in LDS as one small array per relative level (a level-j array can never hold
// more than min(16^j, 1024) nodes: it is only refilled, by at most 64 parents,
// when it and every deeper level are empty), pops up to 64 nodes of the deepest
// non-empty level, loads the one or two 128-byte blocks of each node in
// parallel lanes (one when both ends of the interval share a block) and
// compacts the up to sixteen surviving children per node with ballot + popcount.
// ---------------------------------------------------------------------------
struct ExpandItem {
    uint64_t lo, hi;
    uint32_t row, level, prefix, pad;
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
static constexpr int EXP_WAVES = 4;                 // waves per workgroup
static constexpr int EXP_LVL_CAP = 1024;
static constexpr uint32_t EXP_SPLIT_LEVEL = 1;      // phase 1 hands the sub-trees over to phase 2 at this level (<= 16 per row)
static constexpr unsigned EXP_P2_BLOCKS = 1024;     // phase-2 grid cap = 4 workgroups per CU; waves loop over the queue
// relative level j occupies [lvl_off(j), lvl_off(j) + min(16^j, 1024))
__host__ __device__ constexpr int lvl_cap(int j) { return j < 3 ? (1 << (4 * j)) : EXP_LVL_CAP; }
__host__ __device__ constexpr int lvl_off(int j) { return j <= 3 ? ((1 << (4 * j)) - 1) / 15 : 273 + (j - 3) * EXP_LVL_CAP; }
// LDS slots a wave needs to expand a sub-tree spanning `nlev` stored levels
__host__ __device__ constexpr int exp_slots(int nlev) { return nlev <= 0 ? 1 : lvl_off(nlev - 1) + lvl_cap(nlev - 1); }

template <int MODE>
__device__ __forceinline__ void emit_leaf(const EmitTarget &t, uint32_t row, uint32_t sym, uint64_t count)
{
    if (MODE == EMIT_BITS) {
        int64_t tok = (int64_t)sym - t.shift;
        if (sym > 0 && tok >= 0 && (uint64_t)tok < t.vocab)
            atomicOr(&t.bits[(uint64_t)row * t.words_per_row + ((uint64_t)tok >> 5)], 1u << (tok & 31));
    } else {
        t.dense[(uint64_t)row * t.dense_stride + sym] = count;
    }
}

// The (up to) sixteen leaves below one last-level node are sixteen consecutive symbols, i.e. sixteen
// consecutive bits of the row's token bitmap: one or two atomicOr per node instead of one per symbol
// (the bitmap atomics, not the rank probes, were the largest cost of the leaf level before).
__device__ __forceinline__ void emit_leaf_group_bits(const EmitTarget &t, uint32_t row, uint32_t prefix, uint32_t em)
{
    uint32_t mask = em;
    if (prefix == 0) mask &= ~1u;                       // symbol 0 is the sentinel, never a token
    int64_t t0 = (int64_t)((uint64_t)prefix << FMI_DIGIT_BITS) - t.shift;   // token of child 0
    if (t0 < 0) { mask = (-t0 >= 16) ? 0u : (mask >> (uint32_t)(-t0)); t0 = 0; }
    const int64_t room = (int64_t)t.vocab - t0;         // tokens t0 .. vocab-1 exist
    if (room <= 0) mask = 0; else if (room < 16) mask &= (1u << (uint32_t)room) - 1;
    if (!mask) return;
    const uint64_t big = (uint64_t)mask << ((uint32_t)t0 & 31);
    uint32_t *w = &t.bits[(uint64_t)row * t.words_per_row + ((uint64_t)t0 >> 5)];
    if ((uint32_t)big) atomicOr(w, (uint32_t)big);
    if ((uint32_t)(big >> 32)) atomicOr(w + 1, (uint32_t)(big >> 32));
}

// The sixteen children of a node [lo, hi) on level k: child d is [kid_lo(d), kid_hi(d)) on level
// k+1 = superblock row + in-superblock rank.  SB = false: single-superblock index (n < 2^32), the row
// is dbase[k][] in scalar registers and only the 32-bit ranks live in VGPRs; SB = true: the rows of
// the two ends come from the sbase table.
template <bool SB> struct Kids;
template <> struct Kids<false> { uint32_t rl[16], rh[16]; };
template <> struct Kids<true> { uint32_t rl[16], rh[16]; uint64_t bl[16], bh[16]; };

__device__ __forceinline__ uint64_t kid_lo(const FmiDev &ix, uint32_t k, const Kids<false> &c, uint32_t d) { return ix.dbase[k][d] + c.rl[d]; }
__device__ __forceinline__ uint64_t kid_hi(const FmiDev &ix, uint32_t k, const Kids<false> &c, uint32_t d) { return ix.dbase[k][d] + c.rh[d]; }
__device__ __forceinline__ uint64_t kid_lo(const FmiDev &, uint32_t, const Kids<true> &c, uint32_t d) { return c.bl[d] + c.rl[d]; }
__device__ __forceinline__ uint64_t kid_hi(const FmiDev &, uint32_t, const Kids<true> &c, uint32_t d) { return c.bh[d] + c.rh[d]; }

// returns the mask of children that exist
template <bool SB>
__device__ __forceinline__ uint32_t node_children(const FmiDev &ix, uint32_t k, uint64_t lo, uint64_t hi, Kids<SB> &c, uint64_t &probes)
{
    const uint64_t blo = lo >> FMI_BLOCK_SHIFT, bhi = hi >> FMI_BLOCK_SHIFT;
    uint32_t em = 0;
    {
        HBlock a;
        wm_load_block(ix, k, blo, a);
        if (bhi != blo) {
            HBlock b;
            wm_load_block(ix, k, bhi, b);
            wm_block_ranks(a, (uint32_t)lo & (FMI_BLOCK_BITS - 1), c.rl);
            wm_block_ranks(b, (uint32_t)hi & (FMI_BLOCK_BITS - 1), c.rh);
        } else {
            wm_block_ranks(a, (uint32_t)lo & (FMI_BLOCK_BITS - 1), c.rl);
            wm_block_ranks(a, (uint32_t)hi & (FMI_BLOCK_BITS - 1), c.rh);
        }
    }
    if constexpr (SB) {
        const uint64_t slo = blo >> ix.sb_shift, shi = bhi >> ix.sb_shift;
        const uint64_t *rowl = ix.sbase + ((uint64_t)k * ix.nsb + slo) * FMI_ARITY;
        const uint64_t *rowh = ix.sbase + ((uint64_t)k * ix.nsb + shi) * FMI_ARITY;
#pragma unroll
        for (uint32_t d = 0; d < 16; d += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rowl + d);
            c.bl[d] = v.x; c.bl[d + 1] = v.y;
        }
        if (shi != slo) {
#pragma unroll
            for (uint32_t d = 0; d < 16; d += 2) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rowh + d);
                c.bh[d] = v.x; c.bh[d + 1] = v.y;
            }
        } else {
#pragma unroll
            for (uint32_t d = 0; d < 16; d++) c.bh[d] = c.bl[d];
        }
#pragma unroll
        for (uint32_t d = 0; d < 16; d++) em |= (uint32_t)(c.bh[d] + c.rh[d] > c.bl[d] + c.rl[d]) << d;
    } else {
#pragma unroll
        for (uint32_t d = 0; d < 16; d++) em |= (uint32_t)(c.rh[d] > c.rl[d]) << d;
    }
    probes += bhi != blo ? 2 : 1;
    return em;
}

// the same node in the BINARY level-per-bit model of SURVEY.md 8(d): itself plus its non-empty halves,
// quarters and pairs of children; the phantom high bits of the top digit (level 0) have no level
__device__ __forceinline__ uint32_t model_nodes(uint32_t em, uint32_t k, uint32_t pad_bits)
{
    const uint32_t g2 = (em | (em >> 1)) & 0x5555u, g4 = (g2 | (g2 >> 2)) & 0x1111u, g8 = (g4 | (g4 >> 4)) & 0x0101u;
    const uint32_t skip = k == 0 ? pad_bits : 0;
    return (skip < 1 ? 1u : 0u) + (skip < 2 ? (uint32_t)__popc(g8) : 0u) + (skip < 3 ? (uint32_t)__popc(g4) : 0u) + (uint32_t)__popc(g2);
}

// wave-wide compaction of the existing children, one digit at a time: the slot of child d of this
// lane is (children of smaller digits in the wave) + (rank of the lane among the lanes having child d)
__device__ __forceinline__ uint32_t lane_rank_in(uint64_t bal)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
}

// append the children of this wave's nodes (level k, wave-uniform) to the phase-2 queue; all lanes call
template <bool SB>
__device__ __forceinline__ void hand_over(const FmiDev &ix, uint32_t k, uint32_t row, uint32_t prefix, uint32_t em,
                                          const Kids<SB> &c, ExpandItem *out_items, uint32_t *out_count, uint32_t out_cap)
{
    uint32_t total = 0;
#pragma unroll
    for (uint32_t d = 0; d < 16; d++) total += (uint32_t)__popcll(__ballot((em >> d) & 1));
    uint32_t obase = 0;
    if ((threadIdx.x & 63) == 0 && total) obase = atomicAdd(out_count, total);
    obase = __shfl(obase, 0);
#pragma unroll
    for (uint32_t d = 0; d < 16; d++) {
        const uint64_t bal = __ballot((em >> d) & 1);
        if ((em >> d) & 1) {
            const uint32_t o = obase + lane_rank_in(bal);
            if (o < out_cap) out_items[o] = ExpandItem{kid_lo(ix, k, c, d), kid_hi(ix, k, c, d), row, k + 1, (prefix << 4) | d, 0};
        }
        obase += (uint32_t)__popcll(bal);
    }
}

__device__ __forceinline__ void flush_counters(uint64_t *probe_counter, uint64_t probes, uint32_t model, uint32_t iters, uint32_t nodes)
{
    // one 64-byte line per slot: thousands of waves adding to ONE address cost tens of microseconds
    unsigned long long *slot = (unsigned long long *)probe_counter + (size_t)(blockIdx.x & (PROBE_SLOTS - 1)) * 8;
    if (probes) atomicAdd(slot, (unsigned long long)probes);
    if (model) atomicAdd(slot + 3, (unsigned long long)model);
    if ((threadIdx.x & 63) == 0 && iters) {
        atomicAdd(slot + 1, (unsigned long long)iters);
        atomicAdd(slot + 2, (unsigned long long)nodes);
    }
}

// Work items are hex-wavelet-matrix nodes.  Nodes whose children would sit on
// `stop_level` are not expanded further but appended to `out_items` (phase 1 ->
// phase 2 hand-over, so that a wide interval is spread over up to 16^stop_level
// wavefronts instead of one); stop_level >= dlevels disables the hand-over.
// Dynamic LDS per wave: 3 x slots words (lo32, hi32, packed high bits + prefix)
// + FMI_MAX_DLEVELS counters.
template <int MODE, bool SB>
__device__ __forceinline__ void expand_body(const FmiDev &ix, const ExpandItem *items, const uint32_t *n_items_ptr,
                                            uint32_t n_items_static, const EmitTarget &tgt, uint32_t slots,
                                            uint32_t stop_level, ExpandItem *out_items, uint32_t *out_count,
                                            uint32_t out_cap, uint64_t *probe_counter)
{
    extern __shared__ uint32_t s_mem[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t *s_lo = s_mem + (size_t)wv * (3 * slots + FMI_MAX_DLEVELS);
    uint32_t *s_hi = s_lo + slots;
    uint32_t *s_mx = s_hi + slots;     // lo[39:32] | hi[39:32]<<8 | prefix<<16 (prefix <= 16 bits)
    uint32_t *s_cnt = s_mx + slots;
    const uint32_t n_items = n_items_ptr ? min(*n_items_ptr, n_items_static) : n_items_static;
    const uint32_t D = ix.dlevels;
    const uint32_t pad_bits = FMI_DIGIT_BITS * D - ix.levels;   // phantom high bits of the top digit (0..3)
    uint64_t probes = 0;
    uint32_t iters = 0, nodes = 0, model = 0;

    for (uint32_t item = blockIdx.x * EXP_WAVES + wv; item < n_items; item += gridDim.x * EXP_WAVES) {
        const ExpandItem it = items[item];
        if (it.hi <= it.lo) continue;
        const uint32_t row = it.row;
        // wave-uniform by construction; say so, or every dbase[k][d] becomes a per-lane global load
        // in the middle of the fan-out instead of a scalar load
        const uint32_t root = __builtin_amdgcn_readfirstlane(it.level);
        // a root sitting below the last level is already a leaf
        if (root >= D) { if (lane == 0) emit_leaf<MODE>(tgt, row, it.prefix, it.hi - it.lo); continue; }
        if (lane < FMI_MAX_DLEVELS) s_cnt[lane] = 0;
        if (lane == 0) {
            s_lo[0] = (uint32_t)it.lo; s_hi[0] = (uint32_t)it.hi;
            // positions < 2^40: 8 high bits each
            s_mx[0] = (uint32_t)(it.lo >> 32) | ((uint32_t)(it.hi >> 32) << 8) | (it.prefix << 16);
            s_cnt[0] = 1;
        }
        wave_sync();
        int deepest = 0;   // relative level of the deepest non-empty array (wave uniform)
        while (deepest >= 0) {
            // LDS hands the counter back in a VGPR; it is the same in every lane, and the whole loop
            // (level, offsets, the per-level dbase[] loads) stays scalar only if the compiler knows
            const uint32_t cnt = __builtin_amdgcn_readfirstlane(s_cnt[deepest]);
            if (cnt == 0) { deepest--; continue; }
            const uint32_t m = cnt < 64 ? cnt : 64;
            const uint32_t base = lvl_off(deepest) + (cnt - m);
            const uint32_t k = root + deepest;      // absolute level of the popped nodes
            const bool act = lane < m;
            uint64_t lo = 0, hi = 0; uint32_t prefix = 0;
            if (act) {
                const uint32_t mx = s_mx[base + lane];
                lo = (uint64_t)s_lo[base + lane] | ((uint64_t)(mx & 0xff) << 32);
                hi = (uint64_t)s_hi[base + lane] | ((uint64_t)((mx >> 8) & 0xff) << 32);
                prefix = mx >> 16;
            }
            wave_sync();
            if (lane == 0) s_cnt[deepest] = cnt - m;
            Kids<SB> kids;
            uint32_t em = 0;                        // children that exist
            if (act) {
                em = node_children<SB>(ix, k, lo, hi, kids, probes);
                model += model_nodes(em, k, pad_bits);
            }
            iters++; nodes += m;
            if (k + 1 == D) {
                if (MODE == EMIT_BITS) {
                    if (em) emit_leaf_group_bits(tgt, row, prefix, em);
                } else {
#pragma unroll
                    for (uint32_t d = 0; d < 16; d++)
                        if ((em >> d) & 1) emit_leaf<MODE>(tgt, row, (prefix << 4) | d, kid_hi(ix, k, kids, d) - kid_lo(ix, k, kids, d));
                }
            } else if (k + 1 == stop_level) {
                hand_over<SB>(ix, k, row, prefix, em, kids, out_items, out_count, out_cap);
            } else {
                uint32_t dst = lvl_off(deepest + 1) + __builtin_amdgcn_readfirstlane(s_cnt[deepest + 1]), added = 0;
#pragma unroll
                for (uint32_t d = 0; d < 16; d++) {
                    const uint64_t bal = __ballot((em >> d) & 1);
                    if ((em >> d) & 1) {
                        const uint32_t o = dst + added + lane_rank_in(bal);
                        const uint64_t clo = kid_lo(ix, k, kids, d), chi = kid_hi(ix, k, kids, d);
                        s_lo[o] = (uint32_t)clo; s_hi[o] = (uint32_t)chi;
                        s_mx[o] = (uint32_t)(clo >> 32) | ((uint32_t)(chi >> 32) << 8) | (((prefix << 4) | d) << 16);
                    }
                    added += (uint32_t)__popcll(bal);
                }
                wave_sync();
                if (lane == 0) s_cnt[deepest + 1] += added;
                wave_sync();
                if (added) deepest++;
            }
        }
        wave_sync();
    }
    if (probe_counter) flush_counters(probe_counter, probes, model, iters, nodes);
}

// Single-superblock indexes (n < 2^32, the NQ case).  226 VGPRs = two waves per SIMD; forcing three
// (amdgpu_waves_per_eu(3,3), 168 VGPRs + 84 B of scratch) measured slower: 120 vs 97 us per wide call.
template <int MODE>
__global__ __launch_bounds__(EXP_WAVES * 64)
void k_expand(FmiDev ix, const ExpandItem *items, const uint32_t *n_items_ptr, uint32_t n_items_static, EmitTarget tgt, uint32_t slots,
              uint32_t stop_level, ExpandItem *out_items, uint32_t *out_count, uint32_t out_cap, uint64_t *probe_counter)
{
    expand_body<MODE, false>(ix, items, n_items_ptr, n_items_static, tgt, slots, stop_level, out_items, out_count, out_cap, probe_counter);
}

// Superblocked indexes (n >= 2^32): the two superblock rows add 64 VGPRs
template <int MODE>
__global__ __launch_bounds__(EXP_WAVES * 64)
void k_expand_sb(FmiDev ix, const ExpandItem *items, const uint32_t *n_items_ptr, uint32_t n_items_static, EmitTarget tgt, uint32_t slots,
                 uint32_t stop_level, ExpandItem *out_items, uint32_t *out_count, uint32_t out_cap, uint64_t *probe_counter)
{
    expand_body<MODE, true>(ix, items, n_items_ptr, n_items_static, tgt, slots, stop_level, out_items, out_count, out_cap, probe_counter);
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
// ---------------------------------------------------------------------------
static constexpr int MAX_FORCE = 8;
struct ForceFrom { int64_t tok[MAX_FORCE]; uint32_t n; };

// one thread per (batch, beam) row: ranges of the prefix, the row's class, and
// the work item for the expansion.  Lines 87-105 and the branch order of 111-131.
// With a queue (`out_items`) the same lane also expands the root node of its row
// (level 0) and hands the level-1 children straight to phase 2: the narrow steps of
// a decode are launch-latency bound, and this saves them a kernel.  `zero_next`
// (two words) is cleared for the NEXT call, whose counters alternate with this one's.
template <bool SB>
__global__ __launch_bounds__(64) void k_prefix_ranges(FmiDev ix, uint64_t rows, uint64_t cur_len, const int64_t *ids, int64_t shift,
                                int64_t pad_id, int64_t eos_id, ForceFrom ff, int64_t stop_at_count,
                                int always_allow_eos, uint64_t vocab, uint64_t words_per_row,
                                uint32_t *bits, ExpandItem *items, ExpandItem *out_items, uint32_t *out_count,
                                uint32_t out_cap, uint32_t *zero_next, uint64_t *probe_counter,
                                const uint64_t *st_in, const int64_t *parent, uint64_t *st_out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < rows;
    if (zero_next && r == 0) { zero_next[0] = 0; zero_next[1] = 0; }
    ExpandItem it{0, 0, (uint32_t)r, 0, 0, 0};
    uint64_t probes = 0;
    uint32_t model = 0;      // in nodes of the binary model: one backward-search step = `levels` nodes (2 L probes)
    if (valid) {
        const int64_t *sent = ids + r * cur_len;
        const int64_t last = sent[cur_len - 1];
        uint64_t lo = 0, hi = 0, count = 0;
        if (!(last == eos_id || last == pad_id)) {
            // get_range(force_decoding_from + sent[1:]) and get_count(... sent[1:-1])
            uint64_t l = 0, rr = ix.n;
            if (st_in) {
                // incremental: the row extends row parent[r] of the previous step, whose inclusive range
                // [l, rr] after the same prefix was kept -- one backward-search step instead of len
                const uint64_t pr = (uint64_t)parent[r];
                l = st_in[2 * pr]; rr = st_in[2 * pr + 1];
                count = (rr + 1) - l;
                bs_step(ix, (uint64_t)(last + shift), l, rr, l, rr, &probes);
                model += ix.levels * (uint32_t)(ff.n + (cur_len - 1));    // the reference re-searches the whole prefix
            } else {
                const uint64_t total = ff.n + (cur_len - 1);
                for (uint64_t t = 0; t < total; t++) {
                    if (t + 1 == total) count = (rr + 1) - l;
                    const int64_t tok = t < ff.n ? ff.tok[t] : sent[1 + (t - ff.n)];
                    bs_step(ix, (uint64_t)(tok + shift), l, rr, l, rr, &probes);
                    model += ix.levels;
                }
                if (total == 0) count = (rr + 1) - l;
            }
            if (st_out) { st_out[2 * r] = l; st_out[2 * r + 1] = rr; }
            lo = l; hi = rr + 1;
        }
        uint32_t *myrow = bits + r * words_per_row;
        int64_t single = -1;
        if (stop_at_count > 0 && (int64_t)count <= stop_at_count) single = eos_id;
        else if (last == eos_id || last == pad_id) single = pad_id;
        else { it.lo = lo; it.hi = hi > ix.n ? ix.n : hi; }
        if (!out_items) items[r] = it;
        if (single >= 0 && (uint64_t)single < vocab) atomicOr(&myrow[single >> 5], 1u << (single & 31));
        if (always_allow_eos && eos_id >= 0 && (uint64_t)eos_id < vocab) atomicOr(&myrow[eos_id >> 5], 1u << (eos_id & 31));
    }
    if (!out_items) return;
    // root expansion, the level-0 step of k_expand
    Kids<SB> kids;
    uint32_t em = 0;
    const bool act = valid && it.hi > it.lo;
    if (act) {
        em = node_children<SB>(ix, 0, it.lo, it.hi, kids, probes);
        model += model_nodes(em, 0, FMI_DIGIT_BITS * ix.dlevels - ix.levels);
    }
    hand_over<SB>(ix, 0, (uint32_t)r, 0, em, kids, out_items, out_count, out_cap);
    if (probe_counter) flush_counters(probe_counter, probes, model, __ballot(act) ? 1u : 0u, (uint32_t)__popcll(__ballot(act)));
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
// Nothing of shape [rows, vocab] is written: k_row_lse reads the logits once, k_row_topk reads
// only the allowed tokens of each row (bitmap from k_prefix_ranges + k_expand), k_query_merge
// merges the K per-row lists of a query.  Ties go to the lower flat index.
// ---------------------------------------------------------------------------
static constexpr int TOPK_MAX = 64;          // 2 * num_beams <= 64

__device__ __forceinline__ float logp_processed(float x, float mx, float lsum)
{
    float lp = (x - mx) - lsum;                                  // log_softmax
    if (lp != lp) lp = 0.0f;                                     // InfNanRemoveLogitsProcessor (HF 4.13): nan -> 0
    if (lp == __builtin_huge_valf()) lp = 3.402823466e+38f;      // +inf -> finfo.max
    return lp;
}

// one workgroup (1024 threads) per row: max and log(sum(exp(x - max)))
static constexpr int ROW_BLOCK = 1024;
static constexpr int TOPK_NARROW = 1024;     // rows with at most this many allowed tokens are selected in LDS (k_row_topk)
__global__ __launch_bounds__(ROW_BLOCK) void k_row_lse(const float *logits, uint64_t vocab, float *row_max, float *row_lsum)
{
    __shared__ float s_a[ROW_BLOCK / 64], s_b[ROW_BLOCK / 64];
    const float *x = logits + (uint64_t)blockIdx.x * vocab;
    float mx = -__builtin_huge_valf();
    bool nan = false;
    for (uint64_t v = threadIdx.x; v < vocab; v += ROW_BLOCK) { const float a = x[v]; nan |= (a != a); mx = fmaxf(mx, a); }
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o)); }
    const uint64_t any_nan = __ballot(nan);
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = mx; s_b[threadIdx.x >> 6] = any_nan ? 1.f : 0.f; }
    __syncthreads();
    mx = s_a[0];
    float nn = 0.f;
    for (int i = 0; i < ROW_BLOCK / 64; i++) { mx = fmaxf(mx, s_a[i]); nn += s_b[i]; }
    const bool row_nan = nn > 0.f;
    __syncthreads();
    float sum = 0.f;
    for (uint64_t v = threadIdx.x; v < vocab; v += ROW_BLOCK) sum += expf(x[v] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63) == 0) s_a[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < ROW_BLOCK / 64; i++) tot += s_a[i];
        const float qnan = __builtin_nanf("");
        row_max[blockIdx.x] = row_nan ? qnan : mx;
        row_lsum[blockIdx.x] = row_nan ? qnan : logf(tot);
    }
}

// order-preserving map float -> uint32 (ascending)
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one workgroup per row: the `want` (<= 64) best allowed tokens of the row by processed log-prob
// (descending, ties to the lower token id) -> row_tok / row_lp [rows, want]; row_cnt = how many exist.
// Exact radix select: 4 coalesced passes (8 bits each) over the row's allowed tokens histogram the
// keys that match the prefix found so far and pin the want-th largest key T; one more pass collects
// the keys > T plus as many keys == T as still needed (lowest tokens first); <= 64 survivors are
// ordered by counting ranks.  Lanes read consecutive tokens, the bitmap word is a broadcast.
__global__ __launch_bounds__(ROW_BLOCK) void k_row_topk(const float *logits, const uint32_t *bits, uint64_t words_per_row,
                                                  uint32_t row_broadcast_bits, uint64_t vocab, const float *row_max,
                                                  const float *row_lsum, uint32_t want, int32_t *row_tok, float *row_lp,
                                                  uint32_t *row_cnt, uint32_t narrow_max)
{
    __shared__ int32_t s_gtok[TOPK_NARROW];
    __shared__ float s_gval[TOPK_NARROW];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_prefix, s_remaining, s_n_gt, s_n_eq, s_total;
    __shared__ int32_t s_ctok[TOPK_MAX];
    __shared__ float s_cval[TOPK_MAX];
    __shared__ uint32_t s_wave[ROW_BLOCK / 64];
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    const float *x = logits + (uint64_t)row * vocab;
    const uint32_t *b = bits + (row_broadcast_bits ? 0 : (uint64_t)row * words_per_row);
    const float mx = row_max[row], ls = row_lsum[row];
    if (tid == 0) { s_prefix = 0; s_remaining = want; s_n_gt = 0; s_n_eq = 0; s_total = 0; }
    // number of allowed tokens
    {
        uint32_t c = 0;
        for (uint64_t w = tid; w < words_per_row; w += ROW_BLOCK) {
            uint32_t word = b[w];
            if ((w + 1) * 32 > vocab) { const uint32_t keep = (uint32_t)(vocab - w * 32); word &= keep >= 32 ? ~0u : ((1u << keep) - 1); }
            c += (uint32_t)__popc(word);
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
        __syncthreads();
        if ((tid & 63) == 0 && c) atomicAdd(&s_total, c);
        __syncthreads();
    }
    const uint32_t total = s_total;
    const uint32_t k_sel = total < want ? total : want;        // how many we will output
    if (k_sel == 0) { if (tid == 0) row_cnt[row] = 0; return; }
    if (total <= narrow_max) {
        // narrow row (most decode steps after the first few): one pass over the bitmap words gathers the
        // allowed tokens into LDS, every candidate then finds its own rank under the output order
        // (value descending, ties to the lower token id) -- no radix passes, no per-chunk barriers
        for (uint64_t w = tid; w < words_per_row; w += ROW_BLOCK) {
            uint32_t word = b[w];
            if ((w + 1) * 32 > vocab) { const uint32_t keep = (uint32_t)(vocab - w * 32); word &= keep >= 32 ? ~0u : ((1u << keep) - 1); }
            while (word) {
                const uint32_t tok = (uint32_t)(w * 32) + (uint32_t)__builtin_ctz(word);
                word &= word - 1;
                const uint32_t o = atomicAdd(&s_n_gt, 1u);
                s_gtok[o] = (int32_t)tok; s_gval[o] = logp_processed(x[tok], mx, ls);
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < total; i += ROW_BLOCK) {
            const float v = s_gval[i]; const int32_t t = s_gtok[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < total; j++) {
                const float ov = s_gval[j]; const int32_t ot = s_gtok[j];
                rank += (ov > v) || (ov == v && ot < t);
            }
            if (rank < k_sel) { row_tok[(uint64_t)row * want + rank] = t; row_lp[(uint64_t)row * want + rank] = v; }
        }
        if (tid == 0) row_cnt[row] = k_sel;
        return;
    }
    uint32_t T = 0;
    if (total > want) {
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            const uint32_t pmask = pass == 0 ? 0u : (~0u << (shift + 8));
            for (uint64_t tok = tid; tok < vocab; tok += ROW_BLOCK) {
                if (!((b[tok >> 5] >> (tok & 31)) & 1)) continue;
                const uint32_t key = float_key(logp_processed(x[tok], mx, ls));
                if ((key & pmask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255], 1u);
            }
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
        T = s_prefix;
    }
    const uint32_t need_eq_max = total > want ? s_remaining : 0;     // ties with T still needed
    __syncthreads();
    // collect: everything (total <= want) or keys > T and the first need_eq_max keys == T
    for (uint64_t base = 0; base < vocab; base += ROW_BLOCK) {
        const uint64_t tok = base + tid;
        bool ok = tok < vocab && ((b[tok >> 5] >> (tok & 31)) & 1);
        float lp = 0.f; uint32_t key = 0;
        if (ok) { lp = logp_processed(x[tok], mx, ls); key = float_key(lp); }
        const bool gt = ok && (total <= want || key > T);
        const bool eq = ok && total > want && key == T;
        if (gt) { const uint32_t o = atomicAdd(&s_n_gt, 1u); if (o < TOPK_MAX) { s_ctok[o] = (int32_t)tok; s_cval[o] = lp; } }
        // ties in token order: workgroup prefix count per ROW_BLOCK-token chunk
        const uint64_t be = __ballot(eq);
        if ((tid & 63) == 0) s_wave[tid >> 6] = (uint32_t)__popcll(be);
        __syncthreads();
        uint32_t chunk_eq = 0;
        for (int i = 0; i < ROW_BLOCK / 64; i++) chunk_eq += s_wave[i];
        if (chunk_eq) {
            uint32_t before = s_n_eq;
            for (uint32_t w = 0; w < (tid >> 6); w++) before += s_wave[w];
            const uint32_t my = before + (uint32_t)__popcll(be & ((1ull << (tid & 63)) - 1));
            __syncthreads();
            if (eq && my < need_eq_max) {
                const uint32_t o = (k_sel - need_eq_max) + my;       // ties fill the tail slots
                s_ctok[o] = (int32_t)tok; s_cval[o] = lp;
            }
            if (tid == 0) s_n_eq += chunk_eq;
        }
        __syncthreads();
    }
    // order the k_sel survivors by (value desc, token asc)
    if (tid < k_sel) {
        const float v = s_cval[tid]; const int32_t t = s_ctok[tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < k_sel; j++) {
            const float ov = s_cval[j]; const int32_t ot = s_ctok[j];
            rank += (ov > v) || (ov == v && ot < t);
        }
        row_tok[(uint64_t)row * want + rank] = t;
        row_lp[(uint64_t)row * want + rank] = v;
    }
    if (tid == 0) row_cnt[row] = k_sel;
}

// one wavefront per query: merge the K per-row lists; fill up with not-allowed tokens (constrained
// score -inf, as torch.topk would when a query has fewer than `want` finite candidates)
__global__ __launch_bounds__(64) void k_query_merge(const float *logits, const uint32_t *bits, uint64_t words_per_row,
                                                    uint32_t row_broadcast_bits, uint64_t vocab, uint32_t beams, uint32_t want,
                                                    const float *beam_scores, const float *row_max, const float *row_lsum,
                                                    const int32_t *row_tok, const float *row_lp, const uint32_t *row_cnt,
                                                    int64_t *top_idx, float *top_con, float *top_unc)
{
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    // lane l < beams walks row q*beams + l
    const uint32_t row = q * beams + (lane < beams ? lane : 0);
    uint32_t head = 0;
    const uint32_t cnt = lane < beams ? row_cnt[row] : 0;
    const float bs = beam_scores[row];
    uint32_t out = 0;
    for (; out < want; out++) {
        float v = -__builtin_huge_valf(); int64_t idx = -1;
        if (lane < beams && head < cnt) {
            v = row_lp[(uint64_t)row * want + head] + bs;
            idx = (int64_t)lane * (int64_t)vocab + row_tok[(uint64_t)row * want + head];
        }
        float bv = v; int64_t bi = idx;
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(bv, o); const int64_t oi = __shfl_down(bi, o);
            if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        bv = __shfl(bv, 0); bi = __shfl(bi, 0);
        if (bi < 0) break;
        if (idx == bi) head++;
        if (lane == 0) {
            top_idx[(uint64_t)q * want + out] = bi;
            top_con[(uint64_t)q * want + out] = bv;
            top_unc[(uint64_t)q * want + out] = bv;      // allowed token: constrained == unconstrained
        }
    }
    // fillers: lowest flat indices that are NOT allowed (beam 0 first); constrained -inf, real unconstrained score
    if (lane == 0) {
        uint32_t beam = 0; uint64_t tok = 0;
        while (out < want && beam < beams) {
            const uint32_t r = q * beams + beam;
            const uint32_t *b = bits + (row_broadcast_bits ? 0 : (uint64_t)r * words_per_row);
            if (tok >= vocab) { beam++; tok = 0; continue; }
            if (!((b[tok >> 5] >> (tok & 31)) & 1)) {
                const float lp = logp_processed(logits[(uint64_t)r * vocab + tok], row_max[r], row_lsum[r]);
                top_idx[(uint64_t)q * want + out] = (int64_t)beam * (int64_t)vocab + (int64_t)tok;
                top_con[(uint64_t)q * want + out] = -__builtin_huge_valf();
                top_unc[(uint64_t)q * want + out] = lp + beam_scores[r];
                out++;
            }
            tok++;
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
    out3[0] = out3[1] = out3[2] = out3[3] = 0;
    for (uint32_t i = 0; i < PROBE_SLOTS; i++)
        for (int e = 0; e < 4; e++) out3[e] += slots[i * 8 + e];
    if (reset) HIPCHK(hipMemset(h->d_probe_counter, 0, PROBE_SLOTS * 64));
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

// workspace = expansion items + allowed-token bitmap (vocab <= 2^17 -> 4096 words/row)
static constexpr uint64_t WS_BITS_WORDS = (1ull << FMI_MAX_LEVELS) / 32;
static constexpr uint32_t EXP_SPLIT_MAX = 2;                                   // hex levels
static constexpr uint64_t WS_QUEUE_PER_ROW = 1ull << (4 * EXP_SPLIT_MAX);     // room for any split level up to EXP_SPLIT_MAX
// layout: items[rows] | queue[rows * WS_QUEUE_PER_ROW] | bits[rows * WS_BITS_WORDS] | counter
static inline ExpandItem *ws_items(fmi *h) { return (ExpandItem *)h->ws; }
static inline ExpandItem *ws_queue(fmi *h) { return ws_items(h) + h->ws_rows; }
static inline uint32_t *ws_bits(fmi *h) { return (uint32_t *)(ws_queue(h) + h->ws_rows * WS_QUEUE_PER_ROW); }
static inline uint32_t *ws_qcount(fmi *h) { return ws_bits(h) + h->ws_rows * WS_BITS_WORDS; }
// two buffers of (lo, inclusive hi) per row: the incremental constraint state of consecutive decode steps
static inline uint64_t *ws_state(fmi *h, int which) { return (uint64_t *)(ws_qcount(h) + 64) + (uint64_t)which * 2 * h->ws_rows; }
extern "C" int fmi_dev_reserve(fmi_t *h, uint64_t max_rows)
{
    int rc = need_device(h); if (rc) return rc;
    if (max_rows <= h->ws_rows) return FMI_OK;
    if (h->ws) { HIPCHK(hipFree(h->ws)); h->ws = nullptr; h->ws_rows = 0; }
    const uint64_t bytes = max_rows * (sizeof(ExpandItem) * (1 + WS_QUEUE_PER_ROW) + WS_BITS_WORDS * 4 + 32) + 256;
    HIPCHK(hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes; h->ws_rows = max_rows;
    HIPCHK(hipMemset(ws_qcount(h), 0, 64));   // [0..1] / [2..3]: alternating queue counters of the fused path, [4]: generic path
    h->ws_seq = 0;
    h->state_tag = 0;
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

static unsigned expand_grid(uint64_t n_items)
{
    uint64_t g = (n_items + EXP_WAVES - 1) / EXP_WAVES;
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(g, 256ull * 16));
}

static size_t expand_lds_bytes(int nlev) { return (size_t)EXP_WAVES * (3 * (size_t)exp_slots(nlev) + FMI_MAX_DLEVELS) * 4; }

// Expansion of `rows` root intervals (items[0..rows), level 0) in two launches:
//   phase 1: one wave per row walks levels [0, split) and appends the surviving
//            level-`split` nodes (<= 16^split per row) to the queue behind the roots;
//   phase 2: one wave per queued node finishes its sub-tree.
// The queue (rows << 4*split entries) and its counter live in the workspace.
static uint32_t expand_split(const fmi *h, uint64_t rows, const void *queue, uint64_t qcap)
{
    const uint32_t Q = h->dlevels;
    static const char *e_split = getenv("SEALFM_SPLIT");   // tuning knob
    const uint32_t want = e_split ? std::min<uint32_t>((uint32_t)atoi(e_split), EXP_SPLIT_MAX) : EXP_SPLIT_LEVEL;
    return (want > 0 && Q > want + 1 && queue && qcap >= (rows << (4 * want))) ? want : Q;   // shallow trees: single phase
}

template <int MODE>
static int launch_phase2(fmi *h, hipStream_t st, uint32_t split, const ExpandItem *queue, const uint32_t *qcount, uint64_t qcap,
                         const EmitTarget &tgt)
{
    const uint32_t Q = h->dlevels;
    uint64_t *pc = h->probe_count_enabled ? h->d_probe_counter : nullptr;
    const int nlev2 = (int)(Q - split);
    static const char *e_p2 = getenv("SEALFM_P2_BLOCKS");
    const unsigned p2_blocks = e_p2 ? (unsigned)atoi(e_p2) : EXP_P2_BLOCKS;
    auto kern = h->dev.nsb > 1 ? k_expand_sb<MODE> : k_expand<MODE>;
    hipLaunchKernelGGL(kern, dim3(std::min(expand_grid(qcap), p2_blocks)), dim3(EXP_WAVES * 64), expand_lds_bytes(nlev2), st, h->dev,
                       queue, qcount, (uint32_t)qcap, tgt, (uint32_t)exp_slots(nlev2), Q, (ExpandItem *)nullptr, (uint32_t *)nullptr, 0u, pc);
    HIPCHK(hipGetLastError());
    return FMI_OK;
}

template <int MODE>
static int launch_expand(fmi *h, hipStream_t st, ExpandItem *items, uint64_t rows, ExpandItem *queue, uint32_t *qcount,
                         uint64_t qcap, const EmitTarget &tgt)
{
    const uint32_t Q = h->dlevels;
    uint64_t *pc = h->probe_count_enabled ? h->d_probe_counter : nullptr;
    const uint32_t split = expand_split(h, rows, queue, qcap);
    if (split < Q) HIPCHK(hipMemsetAsync(qcount, 0, 4, st));
    const int nlev1 = (int)split;
    auto kern = h->dev.nsb > 1 ? k_expand_sb<MODE> : k_expand<MODE>;
    hipLaunchKernelGGL(kern, dim3(expand_grid(rows)), dim3(EXP_WAVES * 64), expand_lds_bytes(nlev1), st, h->dev,
                       (const ExpandItem *)items, (const uint32_t *)nullptr, (uint32_t)rows, tgt, (uint32_t)exp_slots(nlev1),
                       split, queue, qcount, (uint32_t)qcap, pc);
    HIPCHK(hipGetLastError());
    if (split < Q) return launch_phase2<MODE>(h, st, split, queue, qcount, qcap, tgt);
    return FMI_OK;
}

static int allowed_bits_impl(fmi *h, hipStream_t st, uint64_t rows, uint64_t cur_len, const int64_t *d_ids,
                             uint32_t *d_bits, uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                             const int64_t *force_from, uint64_t n_force, int64_t stop_at_count, int always_allow_eos,
                             uint64_t state_tag = 0, const int64_t *d_parent = nullptr)
{
    if (cur_len < 2) { fmi_set_error("cur_len must be >= 2 (cur_len == 1 is the constant occurring_distinct mask, beam_search.py:73-77)"); return FMI_ERR_ARG; }
    if (n_force > MAX_FORCE) { fmi_set_error("force_decoding_from longer than %d", MAX_FORCE); return FMI_ERR_UNSUPPORTED; }
    if (rows > h->ws_rows) { int rc = fmi_dev_reserve(h, rows); if (rc) return rc; }
    const uint64_t wpr = (vocab + 31) / 32;
    ForceFrom ff{}; ff.n = (uint32_t)n_force;
    for (uint64_t i = 0; i < n_force; i++) ff.tok[i] = force_from[i];
    ExpandItem *items = ws_items(h);
    const uint64_t qcap = h->ws_rows * WS_QUEUE_PER_ROW;
    uint64_t *pc = h->probe_count_enabled ? h->d_probe_counter : nullptr;
    EmitTarget tgt{}; tgt.bits = d_bits; tgt.words_per_row = wpr; tgt.shift = shift; tgt.vocab = vocab;
    const bool timed = h->timing_enabled && h->ev_used < MAX_TIMED_LAUNCHES;
    HIPCHK(hipMemsetAsync(d_bits, 0, rows * wpr * 4, st));
    // the usual case (root hand-over at level 1): the prefix kernel expands the roots itself and the
    // queue counters alternate between calls, each call clearing the other pair -- two launches per
    // constraint step (prefix + phase 2), no memset, no phase-1 kernel.  The timed region covers both,
    // i.e. the backward searches of the prefix are inside it.
    if (expand_split(h, rows, ws_queue(h), qcap) == 1) {
        uint32_t *cur = ws_qcount(h) + 2 * (h->ws_seq & 1), *nxt = ws_qcount(h) + 2 * ((h->ws_seq + 1) & 1);
        h->ws_seq++;
        // incremental prefix state: valid when the caller vouches (tag + parent rows) that this call extends,
        // by exactly one token, the rows of the previous call with the same tag
        const bool inc = state_tag && d_parent && h->state_tag == state_tag && h->state_rows == rows && h->state_len + 1 == cur_len;
        const uint64_t *st_in = inc ? ws_state(h, h->state_flip) : nullptr;
        uint64_t *st_out = state_tag ? ws_state(h, h->state_flip ^ 1) : nullptr;
        if (state_tag) { h->state_tag = state_tag; h->state_rows = rows; h->state_len = cur_len; h->state_flip ^= 1; }
        else h->state_tag = 0;
        if (timed) HIPCHK(hipEventRecord((hipEvent_t)h->ev_start[h->ev_used], st));
        auto pk = h->dev.nsb > 1 ? k_prefix_ranges<true> : k_prefix_ranges<false>;
        hipLaunchKernelGGL(pk, dim3(blocks_for(rows, 64)), dim3(64), 0, st, h->dev, rows, cur_len, d_ids, shift,
                           pad_id, eos_id, ff, stop_at_count, always_allow_eos, vocab, wpr, d_bits, items, ws_queue(h), cur,
                           (uint32_t)qcap, nxt, pc, st_in, d_parent, st_out);
        int rc = launch_phase2<EMIT_BITS>(h, st, 1, ws_queue(h), cur, qcap, tgt);
        if (rc) return rc;
        if (timed) { HIPCHK(hipEventRecord((hipEvent_t)h->ev_stop[h->ev_used], st)); h->ev_used++; }
        return FMI_OK;
    }
    hipLaunchKernelGGL(k_prefix_ranges<false>, dim3(blocks_for(rows, 64)), dim3(64), 0, st, h->dev, rows, cur_len, d_ids, shift,
                       pad_id, eos_id, ff, stop_at_count, always_allow_eos, vocab, wpr, d_bits, items, (ExpandItem *)nullptr,
                       (uint32_t *)nullptr, 0u, (uint32_t *)nullptr, (uint64_t *)nullptr, (const uint64_t *)nullptr,
                       (const int64_t *)nullptr, (uint64_t *)nullptr);
    h->state_tag = 0;
    if (timed) HIPCHK(hipEventRecord((hipEvent_t)h->ev_start[h->ev_used], st));
    int rc = launch_expand<EMIT_BITS>(h, st, items, rows, ws_queue(h), ws_qcount(h) + 4, qcap, tgt);
    if (rc) return rc;
    if (timed) { HIPCHK(hipEventRecord((hipEvent_t)h->ev_stop[h->ev_used], st)); h->ev_used++; }
    return FMI_OK;
}

extern "C" int fmi_dev_allowed_bits(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len, const int64_t *d_input_ids,
                                    uint32_t *d_bits, uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                                    const int64_t *force_from, uint64_t n_force, int64_t stop_at_count, int always_allow_eos)
{
    int rc = need_device(h); if (rc) return rc;
    if (rows == 0) return FMI_OK;
    return allowed_bits_impl(h, (hipStream_t)stream, rows, cur_len, d_input_ids, d_bits, vocab, shift, pad_id, eos_id,
                             force_from, n_force, stop_at_count, always_allow_eos);
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
    // bitmap lives behind the items in the workspace
    uint32_t *bits = ws_bits(h);
    rc = allowed_bits_impl(h, (hipStream_t)stream, rows, cur_len, d_input_ids, bits, vocab, shift, pad_id, eos_id,
                           force_from, n_force, stop_at_count, always_allow_eos);
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
    int rc = need_device(h); if (rc) return rc;
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
        bits = d_first_bits; broadcast = 1;      // the constant first-step mask (beam_search.py:73-77)
        h->state_tag = 0;
    } else {
        if (rows > h->ws_rows) { rc = fmi_dev_reserve(h, rows); if (rc) return rc; }
        rc = allowed_bits_impl(h, st, rows, cur_len, d_input_ids, ws_bits(h), vocab, shift, pad_id, eos_id, force_from, n_force,
                               stop_at_count, always_allow_eos, state_tag, d_parent_rows);
        if (rc) return rc;
        bits = ws_bits(h);
    }
    hipLaunchKernelGGL(k_row_lse, dim3((unsigned)rows), dim3(ROW_BLOCK), 0, st, d_logits, vocab, row_max, row_lsum);
    const char *e_narrow = getenv("SEALFM_TOPK_NARROW");      // tests: 0 forces the radix-select path on every row
    const uint32_t narrow_max = e_narrow ? std::min<uint32_t>((uint32_t)atoi(e_narrow), TOPK_NARROW) : TOPK_NARROW;
    hipLaunchKernelGGL(k_row_topk, dim3((unsigned)rows), dim3(ROW_BLOCK), 0, st, d_logits, bits, wpr, broadcast, vocab, row_max, row_lsum,
                       (uint32_t)want, row_tok, row_lp, row_cnt, narrow_max);
    hipLaunchKernelGGL(k_query_merge, dim3((unsigned)batch), dim3(64), 0, st, d_logits, bits, wpr, broadcast, vocab, (uint32_t)beams,
                       (uint32_t)want, d_beam_scores, row_max, row_lsum, row_tok, row_lp, row_cnt, d_top_idx, d_top_con, d_top_unc);
    HIPCHK(hipGetLastError());
    return FMI_OK;
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
            hitems[i] = ExpandItem{lo, hi, (uint32_t)i, 0, 0, 0};
        }
        HIPCHK(hipMemcpy(items.p, hitems.data(), m * sizeof(ExpandItem), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(dense.p, 0, m * nsym * 8));
        EmitTarget tgt{}; tgt.dense = dense.as<uint64_t>(); tgt.dense_stride = nsym;
        {
            DevBuf q, qc;
            const uint64_t qcap = m * WS_QUEUE_PER_ROW;
            if ((rc = q.alloc(qcap * sizeof(ExpandItem))) || (rc = qc.alloc(8))) return rc;
            rc = launch_expand<EMIT_DENSE>(h, 0, items.as<ExpandItem>(), m, q.as<ExpandItem>(), qc.as<uint32_t>(), qcap, tgt);
            if (rc) return rc;
            HIPCHK(hipDeviceSynchronize());   // q / qc are freed at scope exit
        }
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
