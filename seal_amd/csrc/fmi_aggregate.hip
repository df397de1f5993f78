// Evidence aggregation of seal/keys.py:311-497 on the GPU (gfx950), for a chunk of queries at once, from the located
// rows to the ranked, fully scored documents -- nothing but the top-k leaves the device.
//
// The reference walks every rare key (descending score), every matching row of it, and keeps python sets/dicts:
// `covered_points` (text positions claimed by earlier occurrences), per-document key lists, a repetition discount,
// a stable ranking, then for the best `n_docs_complete_score` documents a trie match over the document text, a heap
// of occurrences and a greedy non-overlapping cover.  All of it is order-sensitive float64 bookkeeping; here it is
// restructured so that the ORDER is carried by sort keys and every sum runs in the reference's order:
//
//  first stage (keys.py:311-367)
//   k_agg_locate        row -> text position for every occurrence i (i = the reference's processing order: keys by
//                       descending score, rows ascending); the suffix array is resident
//   radix sort by (query, position); k_occ_prepare: in that order, the occurrence's key and its DOCUMENT (sampled
//                       position -> document table + boundaries: still two random sectors per row -- a query's positions
//                       are 10^4 symbols apart --, but in this order the documents come out GROUPED, see below);
//                       k_mis: an occurrence is "new" iff no EARLIER-processed new occurrence overlaps
//                       its window [pos - len, pos) -- the greedy maximal independent set of the interval graph in
//                       priority order, resolved cluster by cluster (clusters = connected runs of overlapping
//                       windows, independent of each other) with a parallel fixed point: a vertex is decided once
//                       all its higher-priority neighbours are
//   (no second sort: the document of a position is monotone in the position, so the order by (query, position) already
//                       groups the occurrences by (query, document); rounds 2-5 sorted them once more, 0.34 ms)
//   k_entries           one wave per document: its occurrences put back into PROCESSING order (the order their keys arrive
//                       in the reference; a handful per document), first touch, best key, the keys that count once per document
//                       (keys.py:343-350), the repetition discount in key order with the covered token set as an
//                       LDS bitmap over query-local token ids (keys.py:352-364), float64 in the reference's order
//   three stable radix sorts (first touch, rank key, query) = sorted(first_stage.items(), key=...) (keys.py:366)
//  full scoring (keys.py:377-497)
//   k_full_score        one wave per (query, ranked document): document text read from the resident text, every
//                       occurrence of every key through a hash-table trie, candidates ordered like the reference's
//                       heap by a rank sort in LDS, greedy cover with lane-parallel bit tests, unigram fill in
//                       first-free-occurrence order, float64 sums in acceptance order
//   k_rank_docs         stable ranking by descending score (keys.py:496), one workgroup per query
//   k_full_score<REC>   the same kernel over the top `keep` documents only, recording what the caller gets back
//
// The host routines fmi_first_stage / fmi_full_score (fmi_evidence.cpp) are the bit-exact checkers of this file.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>

#include "fmi_agg.h"
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

namespace {

// device view of the packed plan
struct AggView {
    uint32_t nq, n_keys, n_rare, max_key_len;
    uint32_t pos_bits, doc_bits;       // widths of the position / document fields of the two sort keys: this index's, not the format's maximum
    uint64_t total, vocab;
    const uint32_t *q_key_off, *q_rare_off, *rare_key, *key_len, *key_q, *key_rank, *kset_off, *kset_ids, *q_tok_off, *tok_list, *q_trie_off;
    const uint64_t *rare_occ_off, *key_lo, *uni_flat;
    const double *key_score, *uni_score;
    const uint4 *trie;
};

AggView make_view(const FmiAggHeader &H, const uint8_t *d)
{
    AggView v;
    v.nq = (uint32_t)H.nq; v.n_keys = (uint32_t)H.n_keys; v.n_rare = (uint32_t)H.n_rare; v.max_key_len = (uint32_t)H.max_key_len;
    v.total = H.total_occ; v.vocab = H.vocab;
    auto u32 = [&](uint64_t o) { return (const uint32_t *)(d + o); };
    v.q_key_off = u32(H.o_q_key_off); v.q_rare_off = u32(H.o_q_rare_off); v.rare_key = u32(H.o_rare_key); v.key_len = u32(H.o_key_len);
    v.key_q = u32(H.o_key_q); v.key_rank = u32(H.o_key_rank); v.kset_off = u32(H.o_kset_off); v.kset_ids = u32(H.o_kset_ids);
    v.q_tok_off = u32(H.o_q_tok_off); v.tok_list = u32(H.o_tok_list); v.q_trie_off = u32(H.o_q_trie_off);
    v.rare_occ_off = (const uint64_t *)(d + H.o_rare_occ_off); v.key_lo = (const uint64_t *)(d + H.o_key_lo);
    v.uni_flat = (const uint64_t *)(d + H.o_uni_flat);
    v.key_score = (const double *)(d + H.o_key_score); v.uni_score = (const double *)(d + H.o_uni_score);
    v.trie = (const uint4 *)(d + H.o_trie);
    return v;
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }
__device__ __forceinline__ double rfld(double v) { return __longlong_as_double((long long)rfl64((uint64_t)__double_as_longlong(v))); }
__device__ __forceinline__ uint64_t lanes_below(uint32_t lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// order-preserving map double -> uint64 (-0.0 == +0.0, as float comparison has it)
__device__ __forceinline__ uint64_t f64_order_key(double x)
{
    const uint64_t u = (x == 0.0) ? 0ull : (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | (1ull << 63));
}

// ---------------------------------------------------------------------------
// first stage
// ---------------------------------------------------------------------------
// last rare key r with rare_occ_off[r] <= i: the key occurrence i belongs to (the offsets: a few thousand words, cache-resident)
__device__ __forceinline__ uint32_t rare_of_occurrence(const AggView &v, uint64_t i)
{
    uint32_t a = 0, b = v.n_rare;
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (v.rare_occ_off[mid] <= i) a = mid; else b = mid; }
    return a;
}

// occurrence i -> suffix-array row -> text position, as the sort key (query, position).  The rows of a key are consecutive: the suffix
// array is read in runs, 4 B per row, and nothing else is touched here (the document comes later, in position order: k_occ_prepare).
__global__ __launch_bounds__(256) void k_agg_locate(FmiDev ix, AggView v, uint64_t *key_pos, uint32_t *val)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.total) return;
    const uint32_t a = rare_of_occurrence(v, i);
    const uint32_t k = v.rare_key[a];
    const uint64_t row = v.key_lo[k] + (i - v.rare_occ_off[a]);
    const uint64_t pos = sa_at(ix, row);
    key_pos[i] = ((uint64_t)v.key_q[k] << v.pos_bits) + pos + 256;      // window [pos - len, pos) never goes below 0 after the offset
    val[i] = (uint32_t)i;
}

// in the order of the sort by (query, position): the occurrence's rare key, its window length, and its document (two random sectors per
// row: the sampled position -> document table and a boundary; neighbours in this order are ~10^4 positions apart at NQ size)
__global__ __launch_bounds__(256) void k_occ_prepare(FmiDev ix, AggView v, const uint64_t *E, const uint32_t *sorted_val, uint32_t *occ_s, uint32_t *doc_s,
                                                     uint16_t *M, uint8_t *state)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.total) return;
    // (the search again -- 13 steps through a cache-resident array -- rather than a gather of what k_agg_locate found: one random sector per row less,
    //  measured 294 against 316 us for this kernel + k_mis)
    const uint32_t a = rare_of_occurrence(v, sorted_val[j]);
    occ_s[j] = a;
    M[j] = (uint16_t)v.key_len[v.rare_key[a]];
    state[j] = 0;
    doc_s[j] = (uint32_t)doc_of(ix, (E[j] & ((1ull << v.pos_bits) - 1)) - 256);
}

enum : uint8_t { ST_UNKNOWN = 0, ST_NEW = 1, ST_OLD = 2 };
static constexpr uint32_t MIS_CHUNK = 2048;

// first cluster boundary at or after `from`: an index j such that no window starting at or after position j of the
// sorted order reaches back before the end E[j-1] (windows are sorted by their END; total counts as a boundary)
__device__ uint32_t mis_first_boundary(const uint64_t *E, const uint16_t *M, uint32_t total, uint32_t maxlen, uint32_t from, uint32_t *s_min)
{
    for (uint32_t base = from;; base += blockDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) *s_min = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t j = base + threadIdx.x;
        bool is_b = false;
        if (j >= total) is_b = (j == total);
        else if (j == 0) is_b = true;
        else {
            const uint64_t lim = E[j - 1];
            is_b = true;
            for (uint32_t x = j; x < total && E[x] < lim + maxlen; x++)
                if (E[x] - M[x] < lim) { is_b = false; break; }
        }
        if (is_b) atomicMin(s_min, j);
        __syncthreads();
        const uint32_t found = *s_min;
        if (found != 0xFFFFFFFFu) return found;
    }
}

// E: sorted (query, end position) keys; M: window length; PRI: processing order of the occurrence (lower = earlier)
__global__ __launch_bounds__(256) void k_mis(const uint64_t *E, const uint16_t *M, const uint32_t *PRI, uint8_t *state, uint8_t *newflag,
                                             uint32_t total, uint32_t maxlen, uint32_t *err)
{
    __shared__ uint32_t s_min, s_pending;
    const uint32_t c0 = blockIdx.x * MIS_CHUNK;
    const uint32_t c1 = min(total, c0 + MIS_CHUNK);
    const uint32_t a = mis_first_boundary(E, M, total, maxlen, c0, &s_min);
    if (a >= c1) return;                      // no cluster starts in this chunk (uniform across the workgroup)
    const uint32_t b = (c1 == total) ? total : mis_first_boundary(E, M, total, maxlen, c1, &s_min);
    // every round decides at least the earliest undecided occurrence of the range: b - a rounds always suffice
    for (uint32_t round = 0;; round++) {
        if (round > b - a + 1) { if (threadIdx.x == 0) atomicOr(err, 1u); break; }
        __syncthreads();
        if (threadIdx.x == 0) s_pending = 0;
        __syncthreads();
        for (uint32_t j = a + threadIdx.x; j < b; j += blockDim.x) {
            if (state[j] != ST_UNKNOWN) continue;
            const uint32_t m = M[j];
            if (m == 0) { state[j] = ST_NEW; continue; }      // an empty window claims and needs nothing
            const uint64_t Ej = E[j], Sj = Ej - m;
            const uint32_t pj = PRI[j];
            bool blocked = false, pending = false;
            for (uint32_t x = j; x-- > a;) {                   // ends in (Sj, Ej]: every one of them overlaps
                if (E[x] <= Sj) break;
                if (M[x] == 0 || PRI[x] > pj) continue;
                const uint8_t st = state[x];
                if (st == ST_NEW) { blocked = true; break; }
                if (st == ST_UNKNOWN) pending = true;
            }
            if (!blocked)
                for (uint32_t x = j + 1; x < b && E[x] < Ej + maxlen; x++) {   // later ends: overlap iff they start before Ej
                    if (M[x] == 0 || PRI[x] > pj || E[x] - M[x] >= Ej) continue;
                    const uint8_t st = state[x];
                    if (st == ST_NEW) { blocked = true; break; }
                    if (st == ST_UNKNOWN) pending = true;
                }
            if (blocked) state[j] = ST_OLD;
            else if (!pending) state[j] = ST_NEW;
            else s_pending = 1;
        }
        __syncthreads();
        if (!s_pending) break;
    }
    for (uint32_t j = a + threadIdx.x; j < b; j += blockDim.x) newflag[j] = state[j] == ST_NEW;      // (in the sorted order: k_entries reads it there)
}

// entry boundaries: where (query, document) changes in the order by (query, position)
__global__ __launch_bounds__(256) void k_heads(const uint64_t *E, const uint32_t *doc_s, uint32_t pos_bits, uint32_t *head, uint64_t total)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    head[j] = (j == 0 || doc_s[j] != doc_s[j - 1] || (E[j] >> pos_bits) != (E[j - 1] >> pos_bits)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_entry_starts(const uint32_t *head, const uint32_t *eid, uint32_t *estart, uint32_t *n_entries, uint64_t total)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    if (head[j]) estart[eid[j]] = (uint32_t)j;
    if (j == total - 1) { const uint32_t ne = eid[j] + head[j]; estart[ne] = (uint32_t)total; *n_entries = ne; }   // eid = heads BEFORE j
}

// Document entries (persistent grid): the keys that count for the document and their discounted scores.  A wave takes 64
// consecutive entries at a time.  Most documents are touched by ONE located row (a few million entries per batch, two
// thirds of them singles): such an entry is one LANE's work -- the key counts iff its occurrence is new, no discount
// applies to a single key (keys.py:352: only from the second key on) -- and is done by all 64 lanes at once; the entries
// with several occurrences are then walked by the whole wave, one after the other, as before.
// The occurrences [s, t) of one (query, document) entry arrive in POSITION order (the sort's); the reference meets them in processing order
// (keys by descending score, rows ascending: the occurrence number i), and everything below depends on that order (which key touches the
// document first, which occurrence of a key counts).  So the wave first puts the entry's (i, rare key, new flag) triples in ascending i, into
// scratch arrays at the same offsets: every member's rank = how many members have a smaller i, counted strip by strip (a handful of members
// in nearly every entry: one strip, as many readlanes as members).  Rounds 2-5 got this order from a second, stable radix sort over all rows.
__device__ __forceinline__ void entry_in_processing_order(const uint32_t *ID, const uint32_t *occ_s, const uint8_t *new_s, uint32_t s, uint32_t t,
                                                          uint32_t *ID2, uint32_t *occ2, uint8_t *new2)
{
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = s; base < t; base += 64) {
        const uint32_t j = base + lane;
        const bool valid = j < t;
        const uint32_t mine = valid ? ID[j] : 0xFFFFFFFFu;
        uint32_t rank = 0;
        for (uint32_t ob = s; ob < t; ob += 64) {
            const uint32_t oj = ob + lane;
            const uint32_t other = oj < t ? ID[oj] : 0xFFFFFFFFu;
            const uint32_t n = min(64u, t - ob);
            for (uint32_t m = 0; m < n; m++) rank += rfl((uint32_t)__shfl((int)other, (int)m)) < mine ? 1u : 0u;
        }
        if (valid) { ID2[s + rank] = mine; occ2[s + rank] = occ_s[j]; new2[s + rank] = new_s[j]; }
    }
    __threadfence_block();
    wave_sync();
}

__device__ __forceinline__ void entry_by_wave(const AggView &v, const uint64_t *E, const uint32_t *doc_s, const uint32_t *ID, const uint32_t *occ_rk,
                                              const uint8_t *newflag, int allow_overlaps, double beta, double single_key, uint32_t cover_words, uint32_t *cover,
                                              uint32_t e, uint32_t s, uint32_t t, uint32_t *ckey, double *cscore, uint32_t *ent_nkeys,
                                              uint64_t *ent_rank, uint32_t *ent_first, uint32_t *ent_q, uint32_t *ent_doc, double *ent_score, uint32_t *ent_best)
{
    // (ID / occ_rk / newflag: the entry's members in processing order -- entry_in_processing_order's output --, indexed by slot)
    const uint32_t lane = threadIdx.x & 63;
    // ---- keys that count: first occurrence of each key that is new (or any, with allow_overlaps) ----
    uint32_t nL = 0, carry = 0xFFFFFFFFu;          // carry = rare key of the last counting occurrence so far
    for (uint32_t base = s; base < t; base += 64) {
        const uint32_t j = base + lane;
        const bool valid = j < t;
        const uint32_t r = valid ? occ_rk[j] : 0xFFFFFFFEu;
        const bool c = valid && (allow_overlaps || newflag[j]);
        uint32_t prev = (uint32_t)__shfl_up((int)r, 1);
        if (lane == 0) prev = (base > s) ? occ_rk[base - 1] : 0xFFFFFFFFu;
        const bool runstart = valid && (r != prev);
        const uint64_t cm = __ballot(c), rs = __ballot(runstart);
        const uint64_t at_or_below = rs & (lanes_below(lane) | (1ull << lane));
        bool sel;
        if (at_or_below) {
            const uint32_t r0 = 63 - (uint32_t)__builtin_clzll(at_or_below);
            sel = c && !(cm & lanes_below(lane) & ~lanes_below(r0));
        } else {
            sel = c && !(cm & lanes_below(lane)) && (r != carry);      // run continued from the previous strip
        }
        const uint64_t sm = __ballot(sel);
        if (sel) ckey[s + nL + (uint32_t)__popcll(sm & lanes_below(lane))] = r;
        nL += (uint32_t)__popcll(sm);
        if (cm) carry = (uint32_t)__shfl((int)r, 63 - (int)__builtin_clzll(cm));
    }
    __threadfence_block();
    wave_sync();
    // ---- repetition discount in key order (keys.py:352-364) ----
    double current = 0.0;
    if (nL >= 2) {
        for (uint32_t w = lane; w < cover_words; w += 64) cover[w] = 0;
        wave_sync();
    }
    bool cover_any = false;
    for (uint32_t x = 0; x < nL; x++) {
        const uint32_t k = v.rare_key[rfl(ckey[s + x])];
        const double sco = v.key_score[k];
        const uint32_t o0 = v.kset_off[k], nset = v.kset_off[k + 1] - o0;
        double nsco = sco;
        if (nL >= 2) {
            if (cover_any) {
                uint32_t covered = 0;
                for (uint32_t u = 0; u < nset; u += 64) {
                    const uint32_t id = (u + lane < nset) ? v.kset_ids[o0 + u + lane] : 0xFFFFFFFFu;
                    const bool hit = id != 0xFFFFFFFFu && ((cover[id >> 5] >> (id & 31)) & 1);
                    covered += (uint32_t)__popcll(__ballot(hit));
                }
                const double coeff = (1.0 - beta) + ((beta * (double)(nset - covered)) / (double)nset);
                nsco = coeff * sco;
            }
            for (uint32_t u = 0; u < nset; u += 64)
                if (u + lane < nset) { const uint32_t id = v.kset_ids[o0 + u + lane]; atomicOr(&cover[id >> 5], 1u << (id & 31)); }
            wave_sync();
            cover_any = cover_any || nset > 0;
        }
        current += nsco;
        if (lane == 0) cscore[s + x] = nsco;
    }
    if (lane == 0) {
        const uint32_t i0 = ID[s], r0 = occ_rk[s];
        const double best = v.key_score[v.rare_key[r0]];     // keys arrive by descending score: the first one to touch the document
        const double rk = (1.0 - single_key) * (-current) + single_key * (-best);
        ent_nkeys[e] = nL; ent_score[e] = current; ent_best[e] = r0; ent_first[e] = i0;
        ent_q[e] = (uint32_t)(E[s] >> v.pos_bits); ent_doc[e] = doc_s[s]; ent_rank[e] = f64_order_key(rk);
    }
}

__global__ __launch_bounds__(256) void k_entries(AggView v, const uint64_t *E, const uint32_t *ID, const uint32_t *doc_s, const uint32_t *estart,
                                                 const uint32_t *n_entries_p, const uint32_t *occ_s, const uint8_t *new_s, uint32_t *ID2, uint32_t *occ2,
                                                 uint8_t *new2, int allow_overlaps, double beta, double single_key,
                                                 uint32_t cover_words, uint32_t *ckey, double *cscore, uint32_t *ent_nkeys, uint64_t *ent_rank,
                                                 uint32_t *ent_first, uint32_t *ent_q, uint32_t *ent_doc, double *ent_score, uint32_t *ent_best)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *cover = lds + wave * cover_words;
    const uint32_t ne = *n_entries_p;
    for (uint32_t e0 = (blockIdx.x * 4 + wave) * 64; e0 < ne; e0 += gridDim.x * 4 * 64) {
        const uint32_t e = e0 + lane;
        const bool have = e < ne;
        const uint32_t s = have ? estart[e] : 0, t = have ? estart[e + 1] : 0;
        if (have && t - s == 1) {
            // one occurrence: the same values the wave path computes for it (0.0 + x == x; a single key is never discounted)
            const uint32_t i0 = ID[s], r0 = occ_s[s];
            const bool c = allow_overlaps || new_s[s];
            const double best = v.key_score[v.rare_key[r0]];
            double current = 0.0;
            if (c) { ckey[s] = r0; cscore[s] = best; current += best; }
            const double rk = (1.0 - single_key) * (-current) + single_key * (-best);
            ent_nkeys[e] = c ? 1u : 0u; ent_score[e] = current; ent_best[e] = r0; ent_first[e] = i0;
            ent_q[e] = (uint32_t)(E[s] >> v.pos_bits); ent_doc[e] = doc_s[s]; ent_rank[e] = f64_order_key(rk);
        }
        uint64_t multi = __ballot(have && t - s != 1);
        while (multi) {
            const uint32_t l = (uint32_t)__builtin_ctzll(multi);
            multi &= multi - 1;
            const uint32_t ms = rfl((uint32_t)__shfl((int)s, (int)l)), mt = rfl((uint32_t)__shfl((int)t, (int)l));
            entry_in_processing_order(ID, occ_s, new_s, ms, mt, ID2, occ2, new2);
            entry_by_wave(v, E, doc_s, ID2, occ2, new2, allow_overlaps, beta, single_key, cover_words, cover, e0 + l, ms, mt, ckey, cscore, ent_nkeys,
                          ent_rank, ent_first, ent_q, ent_doc, ent_score, ent_best);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_gather(const T *src, const uint32_t *idx, T *dst, uint64_t n)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) dst[x] = src[idx[x]];
}

// The ranking of the first stage, sorted(first_stage.items(), key=...)[:n_docs_complete_score] (keys.py:366-375): per query the n_top
// entries with the smallest (rank key, first touch) -- Python's sort is stable and the dict is in first-touch order, and a first touch (the
// index of an occurrence) is unique, so the composite 96-bit key K = rank key << 32 | first touch is a total order.  Rounds 2-5 ran three full
// stable radix sorts over every (query, document) entry slot of the batch (first touch, rank key, query: 1.1 ms for 5.7 M slots) to read
// 1 500 per query off the front; this is a SELECTION, one workgroup per query over its entries (contiguous: they were built from the
// (query, document) sort):
//   1. min / max of the rank keys: the bits all entries share are skipped (scores of one query share sign and most of the exponent);
//   2. MSD radix select from the first differing bit, 8 bits a pass: histogram of the digit among the entries that match the bits decided so
//      far, the bin that holds the n_top-th, descend -- it ends when a bin boundary coincides with n_top, at the latest when all 96 bits are
//      decided.  Once the candidates of the boundary bin fit the LDS (SEL_CACHE entries) they are copied there and the remaining passes never
//      touch memory again: a typical query streams its entries four times (min/max, one histogram, the copy, the final collection);
//   3. the entries at or below the threshold are collected and ordered by counting (each one's rank = how many of them are smaller).
// Streaming loops keep SEL_U loads per thread in flight (one workgroup has to pull ~2 MB per pass on its own); histograms are per wave, and a
// wave whose 64 entries share the digit adds once (tie groups: thousands of documents with the same score).
static constexpr uint32_t SEL_WG = 1024, SEL_WAVES = SEL_WG / 64, SEL_U = 8, SEL_CACHE_MAX = 4096;
__host__ __device__ constexpr size_t select_lds_bytes(uint32_t n_top, uint32_t cache)
{
    return (size_t)n_top * 16 + (size_t)cache * 12 + SEL_WAVES * 256 * 4 + 256 * 4 + 2 * SEL_WAVES * 8 + 64;
}
// candidates the LDS can hold beside the n_top selected entries (160 KB per workgroup)
__host__ constexpr uint32_t select_cache_for(uint32_t n_top)
{
    const size_t fixed = select_lds_bytes(n_top, 0), room = 160 * 1024 > fixed ? 160 * 1024 - fixed : 0;
    return (uint32_t)(room / 12 < SEL_CACHE_MAX ? (room / 12) & ~(size_t)7 : SEL_CACHE_MAX);
}
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 key96(uint64_t hi, uint32_t lo) { return ((u128)hi << 32) | lo; }
// bits [pos, pos + width) of a 96-bit key, counted from its most significant bit
__device__ __forceinline__ uint32_t key96_bits(u128 k, uint32_t pos, uint32_t width) { return (uint32_t)(k >> (96 - pos - width)) & ((1u << width) - 1); }
__device__ __forceinline__ bool key96_same_prefix(u128 k, u128 p, uint32_t pos) { return pos == 0 || (k >> (96 - pos)) == (p >> (96 - pos)); }

__device__ __forceinline__ void select_by_one_workgroup(uint64_t *sel_lds, const uint32_t *n_entries_p, const uint32_t *ent_q, const uint64_t *ent_rank,
                                                        const uint32_t *ent_first, const uint32_t *ent_doc, const double *ent_score, uint32_t n_top,
                                                        uint32_t SEL_CACHE, uint32_t *top_doc, uint32_t *top_ent, uint32_t *top_cnt, uint32_t *fs_doc,
                                                        double *fs_score, uint32_t *fs_cnt)
{
    uint64_t *s_hi = sel_lds;                                             // [n_top] the selected entries' rank keys
    uint64_t *c_hi = s_hi + n_top;                                        // [SEL_CACHE] the cached candidates' rank keys
    uint64_t *s_red = c_hi + SEL_CACHE;                                   // [2 * SEL_WAVES] min / max per wave
    uint32_t *s_lo = reinterpret_cast<uint32_t *>(s_red + 2 * SEL_WAVES); // [n_top] ... first touches
    uint32_t *s_e = s_lo + n_top;                                         // [n_top] ... entry numbers
    uint32_t *c_lo = s_e + n_top;                                         // [SEL_CACHE]
    uint32_t *s_hist = c_lo + SEL_CACHE;                                  // [SEL_WAVES][256]
    uint32_t *s_cnt = s_hist + SEL_WAVES * 256;                           // [256] bins summed over the waves
    uint32_t *s_misc = s_cnt + 256;                                       // [0..1] segment, [2] bin, [3] below, [4] done, [5] collected, [6] bin count, [7] cached
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ne = *n_entries_p;
    if (tid < 2) {
        const uint32_t want_q = q + tid;                                  // first entry with ent_q >= want_q
        uint32_t a = 0, b = ne;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (ent_q[mid] < want_q) a = mid + 1; else b = mid; }
        s_misc[tid] = a;
    }
    __syncthreads();
    const uint32_t e0 = s_misc[0], e1 = s_misc[1];
    const uint32_t want = min(e1 - e0, n_top);
    if (tid == 0) { top_cnt[q] = want; fs_cnt[q] = want; }
    if (want == 0) return;
    // ---- the threshold T: the want-th smallest key (all ones when every entry is taken) ----
    u128 T = ~(u128)0;
    if (want < e1 - e0) {
        // 1. the bits every rank key of the query shares (min / max over the segment; equal keys throughout: the first touches decide)
        uint64_t mn = ~0ull, mx = 0;
        for (uint32_t base = e0; base < e1; base += SEL_WG * SEL_U) {
            uint64_t v[SEL_U];
#pragma unroll
            for (uint32_t u = 0; u < SEL_U; u++) { const uint32_t e = base + u * SEL_WG + tid; v[u] = e < e1 ? ent_rank[e] : ent_rank[e0]; }
#pragma unroll
            for (uint32_t u = 0; u < SEL_U; u++) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
        }
        for (int o = 32; o; o >>= 1) {
            const uint64_t a = ((uint64_t)(uint32_t)__shfl_xor((int)(mn >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)mn, o);
            const uint64_t b = ((uint64_t)(uint32_t)__shfl_xor((int)(mx >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)mx, o);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        if (lane == 0) { s_red[wave] = mn; s_red[SEL_WAVES + wave] = mx; }
        __syncthreads();
        mn = s_red[0]; mx = s_red[SEL_WAVES];
        for (uint32_t w = 1; w < SEL_WAVES; w++) { mn = s_red[w] < mn ? s_red[w] : mn; mx = s_red[SEL_WAVES + w] > mx ? s_red[SEL_WAVES + w] : mx; }
        uint32_t pos = mn == mx ? 64u : (uint32_t)__builtin_clzll(mn ^ mx);       // bits decided so far
        u128 P = pos ? ((key96(mn, 0) >> (96 - pos)) << (96 - pos)) : (u128)0;      // ... and their values
        uint32_t need = want, cached = 0;          // cached > 0: the candidates (same first `pos` bits as P) are c_hi / c_lo [0, cached)
        if (tid == 0) s_misc[7] = 0;
        for (;;) {
            const uint32_t width = min(8u, 96u - pos);
            const bool use_lo = pos + width > 64;
            for (uint32_t i = tid; i < SEL_WAVES * 256; i += SEL_WG) s_hist[i] = 0;
            __syncthreads();
            auto tally = [&](bool cand, uint32_t d) {
                const uint64_t cm = __ballot(cand);
                if (!cm) return;
                const uint32_t d0 = (uint32_t)__shfl((int)d, (int)__builtin_ctzll(cm));
                if (__ballot(cand && d == d0) == cm) { if (lane == 0) s_hist[wave * 256 + d0] += (uint32_t)__popcll(cm); }      // one digit in the whole wave
                else if (cand) atomicAdd(&s_hist[wave * 256 + d], 1u);
            };
            if (cached) {
                for (uint32_t base = 0; base < cached; base += SEL_WG) {
                    const uint32_t i = base + tid;
                    const u128 k = i < cached ? key96(c_hi[i], c_lo[i]) : (u128)0;
                    tally(i < cached && key96_same_prefix(k, P, pos), key96_bits(k, pos, width));
                }
            } else {
                for (uint32_t base = e0; base < e1; base += SEL_WG * SEL_U) {
                    uint64_t hi[SEL_U];
                    uint32_t lo[SEL_U];
#pragma unroll
                    for (uint32_t u = 0; u < SEL_U; u++) {
                        const uint32_t e = base + u * SEL_WG + tid;
                        hi[u] = e < e1 ? ent_rank[e] : 0ull;
                        lo[u] = (use_lo && e < e1) ? ent_first[e] : 0u;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < SEL_U; u++) {
                        const uint32_t e = base + u * SEL_WG + tid;
                        const u128 k = key96(hi[u], lo[u]);
                        tally(e < e1 && key96_same_prefix(k, P, pos < 64 ? pos : 64) && (pos <= 64 || key96_same_prefix(k, P, pos)), key96_bits(k, pos, width));
                    }
                }
            }
            __syncthreads();
            if (tid < 256) { uint32_t c = 0; for (uint32_t w = 0; w < SEL_WAVES; w++) c += s_hist[w * 256 + tid]; s_cnt[tid] = c; }
            __syncthreads();
            if (wave == 0) {                      // the bin that holds the need-th candidate: scan of 256 counts, four per lane
                uint32_t c[4], sum = 0;
                for (int j = 0; j < 4; j++) { c[j] = s_cnt[4 * lane + j]; sum += c[j]; }
                uint32_t incl = sum;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += y; }
                uint32_t run = incl - sum;
                for (int j = 0; j < 4; j++) {
                    if (run < need && need <= run + c[j]) { s_misc[2] = 4 * lane + j; s_misc[3] = run; s_misc[4] = (run + c[j] == need) ? 1u : 0u; s_misc[6] = c[j]; }
                    run += c[j];
                }
            }
            __syncthreads();
            const uint32_t bin = s_misc[2], below = s_misc[3], done = s_misc[4], in_bin = s_misc[6];
            P |= (u128)bin << (96 - pos - width);
            pos += width;
            if (done || pos == 96) { T = pos < 96 ? (P | (((u128)1 << (96 - pos)) - 1)) : P; break; }
            need -= below;
            if (!cached && in_bin <= SEL_CACHE) {
                // 2b. the boundary bin's candidates into the LDS: every later pass reads them there
                __syncthreads();
                for (uint32_t base = e0; base < e1; base += SEL_WG * SEL_U) {
                    uint64_t hi[SEL_U];
#pragma unroll
                    for (uint32_t u = 0; u < SEL_U; u++) { const uint32_t e = base + u * SEL_WG + tid; hi[u] = e < e1 ? ent_rank[e] : 0ull; }
#pragma unroll
                    for (uint32_t u = 0; u < SEL_U; u++) {
                        const uint32_t e = base + u * SEL_WG + tid;
                        if (e >= e1 || !key96_same_prefix(key96(hi[u], 0), P, pos < 64 ? pos : 64)) continue;
                        const uint32_t lo = ent_first[e];
                        if (pos > 64 && !key96_same_prefix(key96(hi[u], lo), P, pos)) continue;
                        const uint32_t slot = atomicAdd(&s_misc[7], 1u);
                        if (slot < SEL_CACHE) { c_hi[slot] = hi[u]; c_lo[slot] = lo; }
                    }
                }
                __syncthreads();
                cached = min(s_misc[7], SEL_CACHE);
            }
        }
    }
    // ---- 3. collect, then order by counting ----
    __syncthreads();
    if (tid == 0) s_misc[5] = 0;
    __syncthreads();
    const uint64_t thi = (uint64_t)(T >> 32);
    const uint32_t tlo = (uint32_t)T;
    for (uint32_t base = e0; base < e1; base += SEL_WG * SEL_U) {
        uint64_t hi[SEL_U];
#pragma unroll
        for (uint32_t u = 0; u < SEL_U; u++) { const uint32_t e = base + u * SEL_WG + tid; hi[u] = e < e1 ? ent_rank[e] : ~0ull; }
#pragma unroll
        for (uint32_t u = 0; u < SEL_U; u++) {
            const uint32_t e = base + u * SEL_WG + tid;
            if (e >= e1 || hi[u] > thi) continue;
            const uint32_t lo = ent_first[e];
            if (hi[u] == thi && lo > tlo) continue;
            const uint32_t slot = atomicAdd(&s_misc[5], 1u);
            if (slot < n_top) { s_hi[slot] = hi[u]; s_lo[slot] = lo; s_e[slot] = e; }
        }
    }
    __syncthreads();
    const uint32_t got = min(s_misc[5], n_top);        // (== want: the composite keys are distinct)
    for (uint32_t x = tid; x < got; x += SEL_WG) {
        const uint64_t hi = s_hi[x];
        const uint32_t lo = s_lo[x];
        uint32_t r = 0;
        for (uint32_t y = 0; y < got; y++) { const uint64_t h2 = s_hi[y]; r += (h2 < hi || (h2 == hi && s_lo[y] < lo)) ? 1u : 0u; }
        const uint32_t e = s_e[x];
        top_doc[(uint64_t)q * n_top + r] = ent_doc[e];
        top_ent[(uint64_t)q * n_top + r] = e;
        fs_doc[(uint64_t)q * n_top + r] = ent_doc[e];
        fs_score[(uint64_t)q * n_top + r] = ent_score[e];
    }
}

__global__ __launch_bounds__(SEL_WG) void k_select_top(const uint32_t *n_entries_p, const uint32_t *ent_q, const uint64_t *ent_rank, const uint32_t *ent_first,
                                                       const uint32_t *ent_doc, const double *ent_score, uint32_t n_top, uint32_t cache,
                                                       uint32_t *top_doc, uint32_t *top_ent, uint32_t *top_cnt, uint32_t *fs_doc, double *fs_score,
                                                       uint32_t *fs_cnt)
{
    extern __shared__ uint64_t sel_lds[];
    select_by_one_workgroup(sel_lds, n_entries_p, ent_q, ent_rank, ent_first, ent_doc, ent_score, n_top, cache, top_doc, top_ent, top_cnt, fs_doc, fs_score, fs_cnt);
}

// ---- the same selection with the streaming spread over the chip ------------------------------------------------------------------------------
// One workgroup pulls ~15 GB/s on its own (64 KB in flight against ~3 us of latency): the single-workgroup form above spends ~130 us per pass over a
// query's ~200 k entries, 0.9 ms for a typical query -- no better than the sorts it replaces.  So the passes over the ENTRIES run on SEL_G
// workgroups per query, one launch per pass, through per-query records in global memory:
//   k_sel_minmax    segment bounds of the queries (once), min / max of the rank keys (atomicMin / atomicMax)
//   k_sel_hist x 2  histogram of the next 11 bits behind the bits decided so far (every workgroup re-derives the state of its query from the
//                   min / max and the histograms of the passes before: a scan of 2 048 counters; nothing is handed from launch to launch but
//                   counters)
//   k_sel_compact   entries below the boundary bin -> the query's `sure` list (fewer than n_top by construction), entries inside it -> its
//                   `maybe` list (a few, or one tie group)
//   k_sel_final     one workgroup per query: the n_top-th among the `maybe` entries by the remaining bits (in the LDS), sure + chosen ordered by
//                   counting, written out.  A `maybe` list that overflows its buffer (a tie group of more than SEL_MAYBE_CAP documents after
//                   22 more bits) sends the query through the single-workgroup routine instead.
static constexpr uint32_t SEL_G = 12, SEL_GT = 256, SEL_BITS = 11, SEL_BINS = 1u << SEL_BITS, SEL_PASSES = 2, SEL_MAYBE_CAP = 16384;
struct SelWork {
    uint64_t *mm;        // [2][nq] min (memset to ones), max (zero) of the rank keys
    uint32_t *hist;      // [SEL_PASSES][nq][SEL_BINS]
    uint32_t *cnt;       // [nq][4]: sure, maybe, -, -
    uint32_t *qstart;    // [nq + 1]
    uint32_t *sure;      // [nq][n_top] entry numbers
    uint32_t *maybe;     // [nq][SEL_MAYBE_CAP]
    uint32_t n_top, nq;
};
struct SelState { u128 P; uint32_t pos, need, done, in_bin; };

// the state of query q's selection after `passes` histogram passes; by the whole workgroup (s_scr: 8 words of LDS)
__device__ __forceinline__ SelState sel_state(const SelWork &w, uint32_t q, uint32_t passes, uint32_t count, uint32_t want, uint32_t *s_scr)
{
    SelState st;
    st.P = 0; st.pos = 0; st.need = want; st.done = 0; st.in_bin = count;
    if (want >= count) { st.done = 1; return st; }                 // every entry is taken: the threshold is all ones
    const uint64_t mn = w.mm[q], mx = w.mm[w.nq + q];
    st.pos = mn == mx ? 64u : (uint32_t)__builtin_clzll(mn ^ mx);
    st.P = st.pos ? ((key96(mn, 0) >> (96 - st.pos)) << (96 - st.pos)) : (u128)0;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t k = 0; k < passes && !st.done && st.pos < 96; k++) {
        const uint32_t width = min(SEL_BITS, 96u - st.pos), bins = 1u << width;
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t *h = w.hist + ((uint64_t)k * w.nq + q) * SEL_BINS;
            const uint32_t per = (bins + 63) / 64, b0 = lane * per;
            uint32_t sum = 0;
            for (uint32_t j = 0; j < per; j++) sum += b0 + j < bins ? h[b0 + j] : 0u;
            uint32_t incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += y; }
            uint32_t run = incl - sum;
            if (run < st.need && st.need <= incl) {
                for (uint32_t j = 0; j < per; j++) {
                    const uint32_t c = b0 + j < bins ? h[b0 + j] : 0u;
                    if (run < st.need && st.need <= run + c) { s_scr[0] = b0 + j; s_scr[1] = run; s_scr[2] = (run + c == st.need) ? 1u : 0u; s_scr[3] = c; }
                    run += c;
                }
            }
        }
        __syncthreads();
        st.P |= (u128)s_scr[0] << (96 - st.pos - width);
        st.pos += width;
        st.done = s_scr[2] | (st.pos == 96 ? 1u : 0u);
        st.in_bin = s_scr[3];
        if (!st.done) st.need -= s_scr[1];
    }
    __syncthreads();
    return st;
}

// this workgroup's share [a, b) of query q's entries
__device__ __forceinline__ void sel_share(const SelWork &w, uint32_t &q, uint32_t &a, uint32_t &b, uint32_t &count)
{
    q = blockIdx.x / SEL_G;
    const uint32_t g = blockIdx.x - q * SEL_G, e0 = w.qstart[q], e1 = w.qstart[q + 1];
    count = e1 - e0;
    const uint32_t per = ((count + SEL_G - 1) / SEL_G + 63) & ~63u;
    a = min(e1, e0 + g * per); b = min(e1, a + per);
}

__global__ __launch_bounds__(SEL_GT) void k_sel_minmax(const uint32_t *n_entries_p, const uint32_t *ent_q, const uint64_t *ent_rank, SelWork w)
{
    __shared__ uint32_t s_seg[2];
    const uint32_t q = blockIdx.x / SEL_G, g = blockIdx.x - q * SEL_G, tid = threadIdx.x, lane = tid & 63;
    const uint32_t ne = *n_entries_p;
    if (tid < 2) {
        const uint32_t want_q = q + tid;
        uint32_t a = 0, b = ne;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (ent_q[mid] < want_q) a = mid + 1; else b = mid; }
        s_seg[tid] = a;
        if (g == 0 && (tid == 0 || q + 1 == w.nq)) w.qstart[q + tid] = a;
    }
    __syncthreads();
    const uint32_t e0 = s_seg[0], e1 = s_seg[1], count = e1 - e0;
    if (count <= w.n_top) return;                              // every entry is taken: no threshold to look for
    const uint32_t per = ((count + SEL_G - 1) / SEL_G + 63) & ~63u;
    const uint32_t a = min(e1, e0 + g * per), b = min(e1, a + per);
    if (a >= b) return;
    uint64_t mn = ~0ull, mx = 0;
    for (uint32_t base = a; base < b; base += SEL_GT * SEL_U) {
        uint64_t v[SEL_U];
#pragma unroll
        for (uint32_t u = 0; u < SEL_U; u++) { const uint32_t e = base + u * SEL_GT + tid; v[u] = e < b ? ent_rank[e] : ent_rank[a]; }
#pragma unroll
        for (uint32_t u = 0; u < SEL_U; u++) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
    }
    for (int o = 32; o; o >>= 1) {
        const uint64_t x = ((uint64_t)(uint32_t)__shfl_xor((int)(mn >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)mn, o);
        const uint64_t y = ((uint64_t)(uint32_t)__shfl_xor((int)(mx >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)mx, o);
        mn = x < mn ? x : mn; mx = y > mx ? y : mx;
    }
    if (lane == 0) { atomicMin((unsigned long long *)&w.mm[q], (unsigned long long)mn); atomicMax((unsigned long long *)&w.mm[w.nq + q], (unsigned long long)mx); }
}

__global__ __launch_bounds__(SEL_GT) void k_sel_hist(uint32_t pass, const uint64_t *ent_rank, const uint32_t *ent_first, SelWork w)
{
    __shared__ uint32_t s_hist[SEL_BINS];
    __shared__ uint32_t s_scr[8];
    uint32_t q, a, b, count;
    sel_share(w, q, a, b, count);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const SelState st = sel_state(w, q, pass, count, min(count, w.n_top), s_scr);
    if (st.done || a >= b) return;
    const uint32_t pos = st.pos, width = min(SEL_BITS, 96u - pos);
    const bool use_lo = pos + width > 64;
    for (uint32_t i = tid; i < SEL_BINS; i += SEL_GT) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t base = a; base < b; base += SEL_GT * SEL_U) {
        uint64_t hi[SEL_U];
        uint32_t lo[SEL_U];
#pragma unroll
        for (uint32_t u = 0; u < SEL_U; u++) {
            const uint32_t e = base + u * SEL_GT + tid;
            hi[u] = e < b ? ent_rank[e] : 0ull;
            lo[u] = (use_lo && e < b) ? ent_first[e] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < SEL_U; u++) {
            const uint32_t e = base + u * SEL_GT + tid;
            const u128 k = key96(hi[u], lo[u]);
            const bool cand = e < b && key96_same_prefix(k, st.P, pos);
            const uint32_t d = key96_bits(k, pos, width);
            const uint64_t cm = __ballot(cand);
            if (cm) {
                const uint32_t d0 = (uint32_t)__shfl((int)d, (int)__builtin_ctzll(cm));
                if (__ballot(cand && d == d0) == cm) { if (lane == 0) atomicAdd(&s_hist[d0], (uint32_t)__popcll(cm)); }
                else if (cand) atomicAdd(&s_hist[d], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *h = w.hist + ((uint64_t)pass * w.nq + q) * SEL_BINS;
    for (uint32_t i = tid; i < SEL_BINS; i += SEL_GT) { const uint32_t c = s_hist[i]; if (c) atomicAdd(&h[i], c); }
}

__global__ __launch_bounds__(SEL_GT) void k_sel_compact(const uint64_t *ent_rank, const uint32_t *ent_first, SelWork w)
{
    __shared__ uint32_t s_scr[8];
    __shared__ uint32_t s_n[2], s_base[2];
    uint32_t q, a, b, count;
    sel_share(w, q, a, b, count);
    const uint32_t tid = threadIdx.x;
    const SelState st = sel_state(w, q, SEL_PASSES, count, min(count, w.n_top), s_scr);
    if (a >= b) return;
    const uint32_t pos = st.pos;
    const bool use_lo = pos > 64;
    // two sweeps over the share: count, reserve with ONE atomic per list, write (the entries are re-read from the L2)
    for (uint32_t sweep = 0; sweep < 2; sweep++) {
        if (tid < 2) s_n[tid] = 0;
        __syncthreads();
        for (uint32_t base = a; base < b; base += SEL_GT * SEL_U) {
            uint64_t hi[SEL_U];
            uint32_t lo[SEL_U];
#pragma unroll
            for (uint32_t u = 0; u < SEL_U; u++) {
                const uint32_t e = base + u * SEL_GT + tid;
                hi[u] = e < b ? ent_rank[e] : 0ull;
                lo[u] = (use_lo && e < b) ? ent_first[e] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < SEL_U; u++) {
                const uint32_t e = base + u * SEL_GT + tid;
                if (e >= b) continue;
                const u128 kp = pos ? key96(hi[u], lo[u]) >> (96 - pos) : (u128)0, pp = pos ? st.P >> (96 - pos) : (u128)0;
                const bool sure = kp < pp || (st.done && kp == pp), maybe = !st.done && kp == pp;
                if (!sure && !maybe) continue;
                const uint32_t slot = atomicAdd(&s_n[sure ? 0 : 1], 1u);
                if (sweep == 1) {
                    const uint32_t at = s_base[sure ? 0 : 1] + slot;
                    if (sure) { if (at < w.n_top) w.sure[(uint64_t)q * w.n_top + at] = e; }
                    else if (at < SEL_MAYBE_CAP) w.maybe[(uint64_t)q * SEL_MAYBE_CAP + at] = e;
                }
            }
        }
        __syncthreads();
        if (sweep == 0 && tid < 2) s_base[tid] = s_n[tid] ? atomicAdd(&w.cnt[4 * q + tid], s_n[tid]) : 0u;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SEL_WG) void k_sel_final(const uint32_t *n_entries_p, const uint32_t *ent_q, const uint64_t *ent_rank, const uint32_t *ent_first,
                                                      const uint32_t *ent_doc, const double *ent_score, SelWork w, uint32_t cache, uint32_t p2,
                                                      uint32_t *top_doc, uint32_t *top_ent, uint32_t *top_cnt, uint32_t *fs_doc, double *fs_score,
                                                      uint32_t *fs_cnt)
{
    extern __shared__ uint64_t sel_lds[];
    const uint32_t n_top = w.n_top;                                       // (p2 = the power of two at or above it: the sort's array length)
    uint64_t *s_hi = sel_lds;                                             // [p2] selected: rank keys
    uint64_t *c_hi = s_hi + p2;                                           // [cache] the `maybe` entries' rank keys
    uint32_t *s_lo = reinterpret_cast<uint32_t *>(c_hi + cache);          // [p2]
    uint32_t *s_e = s_lo + p2;                                            // [p2]
    uint32_t *c_lo = s_e + p2;                                            // [cache]
    uint32_t *c_e = c_lo + cache;                                         // [cache]
    uint32_t *s_hist = c_e + cache;                                       // [SEL_WAVES][256]
    uint32_t *s_cnt = s_hist + SEL_WAVES * 256;                           // [256]
    uint32_t *s_misc = s_cnt + 256;                                       // [0..7] scratch of sel_state, [8] bin, [9] below, [10] done, [11] taken
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t e0 = w.qstart[q], e1 = w.qstart[q + 1], count = e1 - e0, want = min(count, n_top);
    const uint32_t n_sure = w.cnt[4 * q], n_maybe = w.cnt[4 * q + 1];
    if (want == 0) { if (tid == 0) { top_cnt[q] = 0; fs_cnt[q] = 0; } return; }
    SelState st = sel_state(w, q, SEL_PASSES, count, want, s_misc);
    if (!st.done && (n_maybe > SEL_MAYBE_CAP || n_maybe > cache)) {
        // a tie group that does not fit: this query through the single-workgroup routine, over its entries
        __syncthreads();
        select_by_one_workgroup(sel_lds, n_entries_p, ent_q, ent_rank, ent_first, ent_doc, ent_score, n_top, 0u, top_doc, top_ent, top_cnt, fs_doc, fs_score, fs_cnt);
        return;
    }
    if (tid == 0) { top_cnt[q] = want; fs_cnt[q] = want; }
    // the sure entries, and the maybe entries with their keys
    for (uint32_t i = tid; i < n_sure && i < n_top; i += SEL_WG) {
        const uint32_t e = w.sure[(uint64_t)q * n_top + i];
        s_e[i] = e; s_hi[i] = ent_rank[e]; s_lo[i] = ent_first[e];
    }
    uint32_t taken = min(n_sure, n_top);
    if (!st.done) {
        for (uint32_t i = tid; i < n_maybe; i += SEL_WG) {
            const uint32_t e = w.maybe[(uint64_t)q * SEL_MAYBE_CAP + i];
            c_e[i] = e; c_hi[i] = ent_rank[e]; c_lo[i] = ent_first[e];
        }
        __syncthreads();
        // the need-th smallest of the maybe entries by the bits behind st.pos, 8 at a time, in the LDS
        u128 P = st.P;
        uint32_t pos = st.pos, need = st.need;
        u128 T = ~(u128)0;
        for (;;) {
            const uint32_t width = min(8u, 96u - pos);
            for (uint32_t i = tid; i < SEL_WAVES * 256; i += SEL_WG) s_hist[i] = 0;
            __syncthreads();
            for (uint32_t base = 0; base < n_maybe; base += SEL_WG) {
                const uint32_t i = base + tid;
                const u128 k = i < n_maybe ? key96(c_hi[i], c_lo[i]) : (u128)0;
                if (i < n_maybe && key96_same_prefix(k, P, pos)) atomicAdd(&s_hist[wave * 256 + key96_bits(k, pos, width)], 1u);
            }
            __syncthreads();
            if (tid < 256) { uint32_t c = 0; for (uint32_t x = 0; x < SEL_WAVES; x++) c += s_hist[x * 256 + tid]; s_cnt[tid] = c; }
            __syncthreads();
            if (wave == 0) {
                uint32_t c[4], sum = 0;
                for (int j = 0; j < 4; j++) { c[j] = s_cnt[4 * lane + j]; sum += c[j]; }
                uint32_t incl = sum;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += y; }
                uint32_t run = incl - sum;
                for (int j = 0; j < 4; j++) {
                    if (run < need && need <= run + c[j]) { s_misc[8] = 4 * lane + j; s_misc[9] = run; s_misc[10] = (run + c[j] == need) ? 1u : 0u; }
                    run += c[j];
                }
            }
            __syncthreads();
            P |= (u128)s_misc[8] << (96 - pos - width);
            pos += width;
            if (s_misc[10] || pos == 96) { T = pos < 96 ? (P | (((u128)1 << (96 - pos)) - 1)) : P; break; }
            need -= s_misc[9];
        }
        if (tid == 0) s_misc[11] = taken;
        __syncthreads();
        for (uint32_t base = 0; base < n_maybe; base += SEL_WG) {
            const uint32_t i = base + tid;
            if (i < n_maybe && key96(c_hi[i], c_lo[i]) <= T) {
                const uint32_t slot = atomicAdd(&s_misc[11], 1u);
                if (slot < n_top) { s_hi[slot] = c_hi[i]; s_lo[slot] = c_lo[i]; s_e[slot] = c_e[i]; }
            }
        }
        __syncthreads();
        taken = min(s_misc[11], n_top);
    }
    __syncthreads();
    // order: bitonic sort of the p2 slots by (rank key, first touch) (counting every entry's rank against all others is 2 M key compares on
    // ONE compute unit: 180 us for 1 500 entries; this is 66 passes over 2 048 slots)
    for (uint32_t x = taken + tid; x < p2; x += SEL_WG) { s_hi[x] = ~0ull; s_lo[x] = ~0u; s_e[x] = 0; }
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < p2; i += SEL_WG) {
                const uint32_t o = i ^ j;
                if (o > i) {
                    const uint64_t h1 = s_hi[i], h2 = s_hi[o];
                    const uint32_t l1 = s_lo[i], l2 = s_lo[o];
                    const bool gt = h1 > h2 || (h1 == h2 && l1 > l2);
                    if (gt == ((i & k) == 0)) {
                        s_hi[i] = h2; s_hi[o] = h1; s_lo[i] = l2; s_lo[o] = l1;
                        const uint32_t t = s_e[i]; s_e[i] = s_e[o]; s_e[o] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t r = tid; r < taken; r += SEL_WG) {
        const uint32_t e = s_e[r];
        top_doc[(uint64_t)q * n_top + r] = ent_doc[e];
        top_ent[(uint64_t)q * n_top + r] = e;
        fs_doc[(uint64_t)q * n_top + r] = ent_doc[e];
        fs_score[(uint64_t)q * n_top + r] = ent_score[e];
    }
}

// entry slots beyond the number of documents are padding: they sort behind everything and belong to "query nq"
__global__ __launch_bounds__(256) void k_pad_entries(const uint32_t *n_entries_p, uint32_t *ent_first, uint64_t *ent_rank, uint32_t *ent_q,
                                                     uint32_t nq, uint32_t *val, uint64_t n)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    if (x >= *n_entries_p) { ent_first[x] = 0xFFFFFFFFu; ent_rank[x] = ~0ull; ent_q[x] = nq; }
    val[x] = (uint32_t)x;
}

// after the three stable sorts the entries are ordered by (query, rank key, first touch); padding entries carry
// query id nq.  top documents of every query, cut to n_top.
__global__ __launch_bounds__(256) void k_top_docs(const uint32_t *sorted_q, const uint32_t *sorted_ent, const uint32_t *ent_doc, uint64_t n, uint32_t nq,
                                                  uint32_t n_top, const double *ent_score, uint32_t *top_doc, uint32_t *top_ent, uint32_t *top_cnt,
                                                  uint32_t *fs_doc, double *fs_score, uint32_t *fs_cnt)
{
    const uint32_t q = blockIdx.x;
    __shared__ uint32_t seg[2];
    if (threadIdx.x < 2) {
        const uint32_t want = q + threadIdx.x;      // first index with sorted_q >= want
        uint64_t a = 0, b = n;
        while (a < b) { const uint64_t mid = (a + b) >> 1; if (sorted_q[mid] < want) a = mid + 1; else b = mid; }
        seg[threadIdx.x] = (uint32_t)a;
    }
    __syncthreads();
    const uint32_t cnt = min(seg[1] - seg[0], n_top);
    if (threadIdx.x == 0) { top_cnt[q] = cnt; fs_cnt[q] = cnt; }
    for (uint32_t r = threadIdx.x; r < cnt; r += blockDim.x) {
        const uint32_t e = sorted_ent[seg[0] + r];
        top_doc[(uint64_t)q * n_top + r] = ent_doc[e];
        top_ent[(uint64_t)q * n_top + r] = e;
        fs_doc[(uint64_t)q * n_top + r] = ent_doc[e];
        fs_score[(uint64_t)q * n_top + r] = ent_score[e];
    }
}

// ---------------------------------------------------------------------------
// full scoring
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_unigrams(AggView v, uint64_t n_uni, double *type_dense)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n_uni) type_dense[v.uni_flat[x]] = v.uni_score[x];
}

__global__ __launch_bounds__(256) void k_scatter_local_ids(AggView v, uint64_t n_tok, uint32_t *tok2local)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_tok) return;
    uint32_t a = 0, b = v.nq;              // query of local token x
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (v.q_tok_off[mid] <= x) a = mid; else b = mid; }
    const uint32_t tok = v.tok_list[x];
    if (tok < v.vocab) tok2local[(uint64_t)a * v.vocab + tok] = (uint32_t)(x - v.q_tok_off[a]);
}

struct ScoreParams {
    int allow_overlaps, single_key_add_unigrams, unigrams_ignore_free_places;
    double beta, single_key;
    int64_t shift;
    uint32_t n_top, per_q, t_cap, cover_words, cand_cap, pick_cap;
};

struct ScoreOut {                // what the recording pass writes (device pointers into the caller's output buffer)
    uint32_t *n_out, *flags, *cursor;
    uint64_t *rec_doc;
    double *rec_score, *rec_best_score;
    int32_t *rec_best_key;
    uint32_t *rec_T, *rec_npicks, *rec_pick_off, *rec_tok_off;
    int32_t *pick_id; double *pick_score; int32_t *tokens;       // pools
    uint32_t *fs_doc; double *fs_score;                          // first-stage ranking [nq][n_top] (keys.py:366), top_cnt entries each
    uint32_t *fs_cnt;
    int32_t *stage_id; double *stage_score;                      // per-wave staging [waves][pick_cap]
};

struct OverflowPool { uint64_t *sk; uint32_t *key; uint32_t *cursor; uint32_t cap; const uint32_t *err; };

enum : uint32_t { AGG_FLAG_FALLBACK = 1u };      // the host must recompute this query with the checker routines

__device__ __forceinline__ bool trie_lookup(const AggView &v, uint32_t tbase, uint32_t tmask, uint32_t node, uint32_t tok, uint32_t &child, uint32_t &key)
{
    uint32_t i = fmi_agg_trie_hash(node, tok) & tmask;
    for (uint32_t n = 0; n <= tmask; n++, i = (i + 1) & tmask) {      // the table is at most half full
        const uint4 s = v.trie[tbase + i];
        if (s.x == FMI_AGG_TRIE_EMPTY) return false;
        if (s.x == node && s.y == tok) { child = s.z; key = s.w; return true; }
    }
    return false;
}

// LDS per wave: tokens [t_cap] | free [t_cap/32] | cover [cover_words] | cand_sk 2 x u64 [cand_cap] | cand_key 2 x u32 [cand_cap] | counter
__host__ __device__ inline size_t score_lds_words(uint32_t t_cap, uint32_t cover_words, uint32_t cand_cap)
{
    return (size_t)t_cap + (t_cap + 31) / 32 + cover_words + 4 * (size_t)cand_cap + 2 * (size_t)cand_cap + 4;
}

template <bool RECORD>
__global__ __launch_bounds__(256) void k_full_score(FmiDev ix, AggView v, ScoreParams p, const uint32_t *top_doc, const uint32_t *top_cnt,
                                                    const uint32_t *order, const double *type_dense, const uint32_t *tok2local, double *scores,
                                                    ScoreOut out, OverflowPool pool)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * 4 + wave;
    const uint32_t q = w / p.per_q, x = w % p.per_q;
    if (q >= v.nq) return;
    const uint32_t cnt = top_cnt[q];
    if (RECORD && x == 0 && lane == 0) {
        out.n_out[q] = min(cnt, p.per_q);
        if (*pool.err) atomicOr(&out.flags[q], AGG_FLAG_FALLBACK);          // an internal invariant of the first stage broke
    }
    if (x >= min(cnt, p.per_q)) return;
    const uint32_t r = RECORD ? order[(uint64_t)q * p.n_top + x] : x;
    const uint32_t d = top_doc[(uint64_t)q * p.n_top + r];
    // ---- LDS carve-up (per wave) ----
    const size_t per_wave = score_lds_words(p.t_cap, p.cover_words, p.cand_cap);
    uint32_t *base = lds + wave * ((per_wave + 1) & ~(size_t)1);
    uint64_t *l_sk0 = (uint64_t *)base;                            // 8-byte aligned parts first
    uint64_t *l_sk1 = l_sk0 + p.cand_cap;
    uint32_t *l_key0 = (uint32_t *)(l_sk1 + p.cand_cap);
    uint32_t *l_key1 = l_key0 + p.cand_cap;
    uint32_t *tok = l_key1 + p.cand_cap;
    uint32_t *freew = tok + p.t_cap;
    uint32_t *cover = freew + (p.t_cap + 31) / 32;
    uint32_t *l_count = cover + p.cover_words;
    // ---- document tokens: [2] + get_doc(doc)[:-1]  (keys.py:388; the text holds the documents reversed, + shift) ----
    const uint64_t db = ix.doc_begin[d], de = ix.doc_begin[d + 1];
    const uint32_t T = (uint32_t)(de - db);
    const uint32_t rec = RECORD ? (q * p.per_q + x) : 0;
    if (T > p.t_cap) {              // cannot happen when t_cap = the longest document of the index
        if (lane == 0) { atomicOr(&out.flags[q], AGG_FLAG_FALLBACK); if (!RECORD) scores[(uint64_t)q * p.n_top + r] = 0.0; }
        return;
    }
    for (uint32_t i = lane; i < T; i += 64) tok[i] = (i == 0) ? 2u : (uint32_t)((int64_t)text_at(ix, de - i) - p.shift);
    const uint32_t fwords = (T + 31) / 32;
    for (uint32_t i = lane; i < fwords; i += 64) freew[i] = (32 * i + 32 <= T) ? ~0u : ((1u << (T & 31)) - 1);
    for (uint32_t i = lane; i < p.cover_words; i += 64) cover[i] = 0;
    if (lane == 0) *l_count = 0;
    wave_sync();
    // ---- every occurrence of every key in the document (keys.py:396-420) ----
    const uint32_t tbase = v.q_trie_off[q], tmask = v.q_trie_off[q + 1] - tbase - 1;
    uint64_t *csk = l_sk0, *ssk = l_sk1;
    uint32_t *ckey = l_key0, *skey = l_key1;
    uint32_t cap = p.cand_cap;
    uint32_t N;
    for (int attempt = 0;; attempt++) {
        for (uint32_t sb = 0; sb < T; sb += 64) {
            const uint32_t s = sb + lane;
            if (s < T) {
                uint32_t node = 0;
                for (uint32_t e = s; e < T; e++) {
                    uint32_t child, key;
                    if (!trie_lookup(v, tbase, tmask, node, tok[e], child, key)) break;
                    node = child;
                    if (key != FMI_AGG_TRIE_EMPTY) {
                        const uint32_t at = atomicAdd(l_count, 1u);
                        if (at < cap) { csk[at] = ((uint64_t)v.key_rank[key] << 32) | s; ckey[at] = key; }
                    }
                }
            }
        }
        wave_sync();
        N = rfl(*l_count);
        if (N <= cap || attempt == 1) break;
        // more occurrences than the LDS list holds: take room from the global pool and enumerate again
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(pool.cursor, 2 * N);
        at = rfl(at);
        if ((uint64_t)at + 2 * (uint64_t)N > pool.cap) {
            if (lane == 0) { atomicOr(&out.flags[q], AGG_FLAG_FALLBACK); if (!RECORD) scores[(uint64_t)q * p.n_top + r] = 0.0; }
            return;
        }
        csk = pool.sk + at; ssk = csk + N; ckey = pool.key + at; skey = ckey + N; cap = N;
        if (lane == 0) *l_count = 0;
        wave_sync();
    }
    // ---- the reference's heap order: (-score, key tokens, start) = (key rank, start); rank sort ----
    for (uint32_t c = lane; c < N; c += 64) {
        const uint64_t my = csk[c];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < N; y++) rank += csk[y] < my;
        ssk[rank] = my;
        skey[rank] = ckey[c];
    }
    // ---- best single key: first registered among the largest score (keys.py:424-441) ----
    double best_score = 0.0;
    int32_t best_key = -1;
    if (RECORD || p.single_key != 0.0) {
        uint64_t bh = ~0ull, bl = ~0ull;
        uint32_t bk = 0xFFFFFFFFu;
        for (uint32_t c = lane; c < N; c += 64) {
            const uint32_t k = ckey[c], len = v.key_len[k];
            const uint32_t e = (uint32_t)csk[c] + len - 1;
            const uint64_t hi = f64_order_key(-v.key_score[k]);
            // registration order at one end position: odd lengths ascending, then even lengths descending
            const uint64_t lo = ((uint64_t)e << 32) | ((len & 1) ? (uint64_t)len : ((1ull << 31) | (uint64_t)(0x7FFFFFFFu - len)));
            if (hi < bh || (hi == bh && lo < bl)) { bh = hi; bl = lo; bk = k; }
        }
        for (int off = 32; off; off >>= 1) {
            const uint64_t oh = ((uint64_t)(uint32_t)__shfl_xor((int)(bh >> 32), off) << 32) | (uint32_t)__shfl_xor((int)bh, off);
            const uint64_t ol = ((uint64_t)(uint32_t)__shfl_xor((int)(bl >> 32), off) << 32) | (uint32_t)__shfl_xor((int)bl, off);
            const uint32_t ok = (uint32_t)__shfl_xor((int)bk, off);
            if (oh < bh || (oh == bh && ol < bl)) { bh = oh; bl = ol; bk = ok; }
        }
        if (bk != 0xFFFFFFFFu) { best_key = (int32_t)bk; best_score = v.key_score[bk]; }
    }
    wave_sync();
    // ---- greedy cover in heap order (keys.py:443-472) ----
    int32_t *st_id = RECORD ? out.stage_id + (uint64_t)rec * p.pick_cap : nullptr;
    double *st_sc = RECORD ? out.stage_score + (uint64_t)rec * p.pick_cap : nullptr;
    uint32_t npicks = 0, prev = 0xFFFFFFFFu;
    double multi = 0.0, last_s = 0.0;
    bool cover_any = false;
    for (uint32_t c = 0; c < N; c++) {
        const uint64_t sk = rfl64(ssk[c]);
        const uint32_t k = rfl(skey[c]);
        const uint32_t i = (uint32_t)sk, j = i + v.key_len[k];
        double new_s;
        const uint32_t o0 = v.kset_off[k], nset = v.kset_off[k + 1] - o0;
        if (k == prev) new_s = last_s;
        else if (nset == 0) new_s = 0.0;
        else {
            const double s = v.key_score[k];
            if (!cover_any) new_s = s;
            else {
                uint32_t covered = 0;
                for (uint32_t u = 0; u < nset; u += 64) {
                    const uint32_t id = (u + lane < nset) ? v.kset_ids[o0 + u + lane] : 0xFFFFFFFFu;
                    const bool hit = id != 0xFFFFFFFFu && ((cover[id >> 5] >> (id & 31)) & 1);
                    covered += (uint32_t)__popcll(__ballot(hit));
                }
                const double coeff = (1.0 - p.beta) + ((p.beta * (double)(nset - covered)) / (double)nset);
                new_s = coeff * s;
            }
        }
        if (new_s <= 0.0) continue;
        const uint32_t w0 = i >> 5, w1 = (j - 1) >> 5;
        if (!p.allow_overlaps) {
            bool bad = false;
            for (uint32_t wi = w0 + lane; wi <= w1; wi += 64) {
                const uint32_t lo = max(i, 32 * wi) - 32 * wi, hi = min(j, 32 * wi + 32) - 32 * wi;       // bits [lo, hi) of word wi
                const uint32_t m = (hi - lo == 32) ? ~0u : (((1u << (hi - lo)) - 1) << lo);
                bad = bad || ((freew[wi] & m) != m);
            }
            if (__ballot(bad)) continue;
        }
        if (k != prev) {
            prev = k;
            for (uint32_t u = 0; u < nset; u += 64)
                if (u + lane < nset) { const uint32_t id = v.kset_ids[o0 + u + lane]; atomicOr(&cover[id >> 5], 1u << (id & 31)); }
            cover_any = cover_any || nset > 0;
            multi += new_s;
            last_s = new_s;
            if (RECORD && lane == 0 && npicks < p.pick_cap) { st_id[npicks] = (int32_t)k; st_sc[npicks] = new_s; }
            npicks++;
        }
        for (uint32_t wi = w0 + lane; wi <= w1; wi += 64) {
            const uint32_t lo = max(i, 32 * wi) - 32 * wi, hi = min(j, 32 * wi + 32) - 32 * wi;
            const uint32_t m = (hi - lo == 32) ? ~0u : (((1u << (hi - lo)) - 1) << lo);
            freew[wi] &= ~m;
        }
        wave_sync();
    }
    // ---- free positions: every distinct token once, in order of its first free occurrence (keys.py:474-487) ----
    if (p.unigrams_ignore_free_places) {
        for (uint32_t i = lane; i < fwords; i += 64) freew[i] = (32 * i + 32 <= T) ? ~0u : ((1u << (T & 31)) - 1);
        wave_sync();
    }
    double uni = 0.0;
    const double *ts = type_dense + (uint64_t)q * v.vocab;
    const uint32_t *t2l = tok2local + (uint64_t)q * v.vocab;
    for (uint32_t pb = 0; pb < T; pb += 64) {
        const uint32_t pos = pb + lane;
        bool cand = false;
        double s = 0.0;
        uint32_t t = 0;
        if (pos < T && ((freew[pos >> 5] >> (pos & 31)) & 1)) {
            t = tok[pos];
            if (t < v.vocab) s = ts[t];
            cand = s > 0.0;
            for (uint32_t y = 0; cand && y < pos; y++)
                if (tok[y] == t && ((freew[y >> 5] >> (y & 31)) & 1)) cand = false;      // not its first free occurrence
        }
        uint64_t cm = __ballot(cand);
        while (cm) {
            const int l = __builtin_ctzll(cm);
            cm &= cm - 1;
            const uint32_t tl = (uint32_t)__shfl((int)t, l);
            double sl = __longlong_as_double(((long long)(uint32_t)__shfl((int)((uint64_t)__double_as_longlong(s) >> 32), l) << 32) |
                                             (uint32_t)__shfl((int)(uint32_t)__double_as_longlong(s), l));
            if (cover_any) {
                const uint32_t id = t2l[tl];
                const bool in_cover = id != 0xFFFFFFFFu && ((cover[id >> 5] >> (id & 31)) & 1);
                const double coeff = (1.0 - p.beta) + ((p.beta * (double)(in_cover ? 0 : 1)) / (double)1);
                sl = coeff * sl;
            }
            if (sl != 0.0) {
                uni += sl;
                if (RECORD && lane == 0 && npicks < p.pick_cap) { st_id[npicks] = -(int32_t)(tl + 1); st_sc[npicks] = sl; }
                npicks++;
            }
        }
    }
    double single = best_score;
    if (p.single_key_add_unigrams) single += uni;
    multi += uni;
    const double score = (1.0 - p.single_key) * multi + p.single_key * single;
    if (!RECORD) {
        if (lane == 0) scores[(uint64_t)q * p.n_top + r] = score;
        return;
    }
    // ---- record: fixed part + picks and tokens in compact pools ----
    const uint32_t np = min(npicks, p.pick_cap);
    uint32_t po = 0, to = 0;
    if (lane == 0) { po = atomicAdd(&out.cursor[0], np); to = atomicAdd(&out.cursor[1], T); }
    po = rfl(po); to = rfl(to);
    __threadfence_block();
    wave_sync();
    for (uint32_t i = lane; i < np; i += 64) { out.pick_id[po + i] = st_id[i]; out.pick_score[po + i] = st_sc[i]; }
    for (uint32_t i = lane; i < T; i += 64) out.tokens[to + i] = (int32_t)tok[i];
    if (lane == 0) {
        if (npicks > p.pick_cap) atomicOr(&out.flags[q], AGG_FLAG_FALLBACK);
        out.rec_doc[rec] = d; out.rec_score[rec] = score; out.rec_best_key[rec] = best_key; out.rec_best_score[rec] = best_score;
        out.rec_T[rec] = T; out.rec_npicks[rec] = np; out.rec_pick_off[rec] = po; out.rec_tok_off[rec] = to;
    }
}

// sorted(results.items(), key=lambda x: -x[1][0]) (keys.py:496): stable, one workgroup per query: bitonic sort of (key, position) pairs over the
// p2 slots of the LDS (p2 = the power of two at or above n_top; the position breaks ties, which is what stability means here).  Round 5
// counted every document's rank against all others: 2 M compares on one compute unit, 92 us.
__global__ __launch_bounds__(1024) void k_rank_docs(const double *scores, const uint32_t *top_cnt, uint32_t n_top, uint32_t p2, uint32_t *order)
{
    extern __shared__ uint64_t keys[];
    uint32_t *idx = reinterpret_cast<uint32_t *>(keys + p2);
    const uint32_t q = blockIdx.x, cnt = top_cnt[q];
    for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) { keys[i] = i < cnt ? f64_order_key(-scores[(uint64_t)q * n_top + i]) : ~0ull; idx[i] = i; }
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) {
                const uint32_t o = i ^ j;
                if (o > i) {
                    const uint64_t k1 = keys[i], k2 = keys[o];
                    const uint32_t i1 = idx[i], i2 = idx[o];
                    const bool gt = k1 > k2 || (k1 == k2 && i1 > i2);
                    if (gt == ((i & k) == 0)) { keys[i] = k2; keys[o] = k1; idx[i] = i2; idx[o] = i1; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t r = threadIdx.x; r < cnt; r += blockDim.x) order[(uint64_t)q * n_top + r] = idx[r];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Carver {
    uint8_t *base;
    uint64_t off = 0;
    explicit Carver(void *p) : base((uint8_t *)p) {}
    template <class T> T *take(uint64_t n)
    {
        off = (off + 255) & ~(uint64_t)255;
        T *r = base ? (T *)(base + off) : nullptr;
        off += std::max<uint64_t>(n, 1) * sizeof(T);
        return r;
    }
};

struct Work {          // workspace layout; base == nullptr: sizes only
    uint32_t *occ_rk, *doc, *v0, *v1, *head, *eid, *estart, *n_entries, *ckey, *ent_nkeys, *ent_first, *ent_q, *ent_doc, *ent_best, *tmp32;
    uint64_t *k0, *k1, *ent_rank;
    uint16_t *M;
    uint8_t *state, *newflag;
    double *cscore, *ent_score, *type_dense, *scores;
    uint32_t *tok2local, *top_doc, *top_ent, *top_cnt, *order, *pool_key, *pool_cursor;
    uint64_t *pool_sk;
    int32_t *stage_id;
    double *stage_score;
    uint64_t *sel_mm;                                  // k_sel_*: [2][nq] min / max, then (contiguous, zeroed together) the histograms and counters
    uint32_t *sel_hist, *sel_cnt, *sel_qstart, *sel_sure, *sel_maybe;
    uint64_t sel_zero_bytes;
    void *rp_tmp;
    uint64_t rp_bytes, pool_cap, bytes;
};

uint64_t rocprim_temp_bytes(uint64_t n)
{
    size_t a = 0, b = 0, c = 0;
    rocprim::double_buffer<uint64_t> k64(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> k32(nullptr, nullptr), v32(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, a, k64, v32, n, 0u, 64u, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, b, k32, v32, n, 0u, 32u, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, c, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return std::max(a, std::max(b, c)) + 256;
}

Work carve(void *base, const FmiAggHeader &H, uint32_t n_top, uint32_t keep, uint32_t pick_cap)
{
    Work w{};
    Carver c(base);
    const uint64_t N = std::max<uint64_t>(H.total_occ, 1), nq = H.nq;
    w.occ_rk = c.take<uint32_t>(N); w.doc = c.take<uint32_t>(N);
    w.k0 = c.take<uint64_t>(N); w.k1 = c.take<uint64_t>(N); w.v0 = c.take<uint32_t>(N); w.v1 = c.take<uint32_t>(N);
    w.M = c.take<uint16_t>(N); w.state = c.take<uint8_t>(N); w.newflag = c.take<uint8_t>(N);
    w.head = c.take<uint32_t>(N); w.eid = c.take<uint32_t>(N); w.estart = c.take<uint32_t>(N + 1); w.n_entries = c.take<uint32_t>(4);
    w.ckey = c.take<uint32_t>(N); w.cscore = c.take<double>(N);
    w.ent_nkeys = c.take<uint32_t>(N); w.ent_rank = c.take<uint64_t>(N); w.ent_first = c.take<uint32_t>(N); w.ent_q = c.take<uint32_t>(N);
    w.ent_doc = c.take<uint32_t>(N); w.ent_score = c.take<double>(N); w.ent_best = c.take<uint32_t>(N); w.tmp32 = c.take<uint32_t>(N);
    w.type_dense = c.take<double>(nq * H.vocab); w.tok2local = c.take<uint32_t>(nq * H.vocab);
    w.top_doc = c.take<uint32_t>(nq * n_top); w.top_ent = c.take<uint32_t>(nq * n_top); w.top_cnt = c.take<uint32_t>(nq);
    w.scores = c.take<double>(nq * n_top); w.order = c.take<uint32_t>(nq * n_top);
    w.pool_cap = 1ull << 23;                                   // 8 M candidate slots (96 MB) for documents that overflow the LDS list
    w.pool_sk = c.take<uint64_t>(w.pool_cap); w.pool_key = c.take<uint32_t>(w.pool_cap); w.pool_cursor = c.take<uint32_t>(8);
    w.stage_id = c.take<int32_t>(nq * (uint64_t)keep * pick_cap); w.stage_score = c.take<double>(nq * (uint64_t)keep * pick_cap);
    w.sel_mm = c.take<uint64_t>(2 * nq);
    w.sel_hist = c.take<uint32_t>((uint64_t)SEL_PASSES * nq * SEL_BINS); w.sel_cnt = c.take<uint32_t>(nq * 4);
    w.sel_zero_bytes = base ? (uint64_t)((uint8_t *)(w.sel_cnt + nq * 4) - (uint8_t *)(w.sel_mm + nq)) : 0;       // [max | histograms | counters]
    w.sel_qstart = c.take<uint32_t>(nq + 1); w.sel_sure = c.take<uint32_t>(nq * n_top); w.sel_maybe = c.take<uint32_t>(nq * (uint64_t)SEL_MAYBE_CAP);
    w.rp_bytes = rocprim_temp_bytes(N);
    w.rp_tmp = c.take<uint8_t>(w.rp_bytes);
    w.bytes = c.off + 256;
    return w;
}

// byte offsets of the output arrays, as fmi_dev_aggregate_sizes reports them (include/sealfm.h FMI_AGG_OUT_*)
struct OutLayout { ScoreOut o; uint64_t off[20]; uint64_t fixed_bytes, bytes; };

OutLayout carve_out(void *base, uint64_t nq, uint32_t n_top, uint32_t keep, uint32_t pick_cap, uint64_t t_cap)
{
    OutLayout L{};
    Carver c(base);
    const uint64_t R = nq * keep;
    auto at = [&](int slot) { L.off[slot] = (c.off + 255) & ~(uint64_t)255; };
    at(0); L.o.n_out = c.take<uint32_t>(nq);
    at(1); L.o.flags = c.take<uint32_t>(nq);
    at(2); L.o.cursor = c.take<uint32_t>(4);
    at(3); L.o.rec_doc = c.take<uint64_t>(R);
    at(4); L.o.rec_score = c.take<double>(R);
    at(5); L.o.rec_best_score = c.take<double>(R);
    at(6); L.o.rec_best_key = c.take<int32_t>(R);
    at(7); L.o.rec_T = c.take<uint32_t>(R);
    at(8); L.o.rec_npicks = c.take<uint32_t>(R);
    at(9); L.o.rec_pick_off = c.take<uint32_t>(R);
    at(10); L.o.rec_tok_off = c.take<uint32_t>(R);
    at(11); L.o.fs_cnt = c.take<uint32_t>(nq);
    L.fixed_bytes = (c.off + 255) & ~(uint64_t)255;
    at(12); L.o.pick_id = c.take<int32_t>(R * pick_cap);
    at(13); L.o.pick_score = c.take<double>(R * pick_cap);
    at(14); L.o.tokens = c.take<int32_t>(R * t_cap);
    at(15); L.o.fs_doc = c.take<uint32_t>(nq * n_top);
    at(16); L.o.fs_score = c.take<double>(nq * n_top);
    L.off[17] = L.fixed_bytes;
    L.bytes = c.off + 256;
    L.off[18] = L.bytes;
    return L;
}

inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

uint32_t pick_cap_for(const fmi *h, const FmiAggHeader &H, int allow_overlaps)
{
    const uint64_t T = std::max<uint64_t>(h->max_doc_len, 1);
    return (uint32_t)(T + (allow_overlaps ? std::max<uint64_t>(T, H.max_q_keys) : T));
}

}  // namespace

struct fmi_agg_plan;
extern "C" const void *fmi_agg_plan_blob(const fmi_agg_plan *p, uint64_t *bytes_out);

static constexpr uint32_t AGG_MAX_DOC_LEN = 8192;      // LDS budget of the scoring kernel (4 waves per workgroup)
static constexpr uint32_t AGG_MAX_TOP = 8192;
static constexpr uint32_t AGG_CAND_CAP = 128;          // occurrences of keys per document held in LDS before the pool is used

static int agg_check(fmi *h, const FmiAggHeader &H, uint64_t n_top, uint64_t keep)
{
    if (H.magic != FMI_AGG_MAGIC) { fmi_set_error("fmi_dev_aggregate: not a plan blob"); return FMI_ERR_ARG; }
    if (h->device < 0) { fmi_set_error("index is not resident on a GPU"); return FMI_ERR_NO_DEVICE; }
    if (!h->dev.sa_lo || !h->dev.text || !h->dev.doc_begin) { fmi_set_error("fmi_dev_aggregate needs the suffix array, the text and the document boundaries resident"); return FMI_ERR_STATE; }
    if (h->max_doc_len == 0 || h->max_doc_len > AGG_MAX_DOC_LEN) { fmi_set_error("fmi_dev_aggregate: longest document %llu tokens (1..%u supported)", (unsigned long long)h->max_doc_len, AGG_MAX_DOC_LEN); return FMI_ERR_ARG; }
    if (n_top == 0 || n_top > AGG_MAX_TOP || keep == 0 || keep > n_top) { fmi_set_error("fmi_dev_aggregate: 1 <= keep <= n_top <= %u", AGG_MAX_TOP); return FMI_ERR_ARG; }
    if (h->n >= (1ull << 40) || h->dev.n_begin >= (1ull << 32)) { fmi_set_error("fmi_dev_aggregate: index too large for the sort keys"); return FMI_ERR_ARG; }
    return FMI_OK;
}

extern "C" int fmi_dev_aggregate_sizes(fmi_t *h, const fmi_agg_plan *plan, uint64_t n_top, uint64_t keep, int allow_overlaps,
                                       uint64_t *ws_bytes, uint64_t *out_layout /* [20] */)
{
    if (!h || !plan) { fmi_set_error("null argument"); return FMI_ERR_ARG; }
    const FmiAggHeader &H = *(const FmiAggHeader *)fmi_agg_plan_blob(plan, nullptr);
    int rc = agg_check(h, H, n_top, keep); if (rc) return rc;
    const uint32_t pc = pick_cap_for(h, H, allow_overlaps);
    const Work w = carve(nullptr, H, (uint32_t)n_top, (uint32_t)keep, pc);
    const OutLayout L = carve_out(nullptr, H.nq, (uint32_t)n_top, (uint32_t)keep, pc, h->max_doc_len);
    if (ws_bytes) *ws_bytes = w.bytes;
    if (out_layout) memcpy(out_layout, L.off, sizeof(L.off));
    return FMI_OK;
}

static unsigned bits_for(uint64_t max_value) { unsigned b = 1; while (b < 64 && (max_value >> b)) b++; return b; }

template <class K>
static int sort_pairs(Work &w, K *&ka, K *&kb, uint32_t *&va, uint32_t *&vb, uint64_t n, unsigned bits, hipStream_t st)
{
    rocprim::double_buffer<K> dk(ka, kb);
    rocprim::double_buffer<uint32_t> dv(va, vb);
    size_t tb = w.rp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(w.rp_tmp, tb, dk, dv, n, 0u, bits, st));
    ka = dk.current(); kb = dk.alternate(); va = dv.current(); vb = dv.alternate();
    return FMI_OK;
}

extern "C" int fmi_dev_aggregate(fmi_t *h, void *stream, const fmi_agg_plan *plan, const void *d_plan_blob, uint64_t n_top, uint64_t keep,
                                 int allow_overlaps, double beta, double single_key, int single_key_add_unigrams,
                                 int unigrams_ignore_free_places, int64_t shift, void *d_ws, uint64_t ws_bytes, void *d_out, uint64_t out_bytes)
{
    if (!h || !plan || !d_plan_blob || !d_ws || !d_out) { fmi_set_error("null argument"); return FMI_ERR_ARG; }
    const FmiAggHeader &H = *(const FmiAggHeader *)fmi_agg_plan_blob(plan, nullptr);
    int rc = agg_check(h, H, n_top, keep); if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const uint32_t pc = pick_cap_for(h, H, allow_overlaps);
    Work w = carve(d_ws, H, (uint32_t)n_top, (uint32_t)keep, pc);
    OutLayout L = carve_out(d_out, H.nq, (uint32_t)n_top, (uint32_t)keep, pc, h->max_doc_len);
    if (w.bytes > ws_bytes || L.bytes > out_bytes) { fmi_set_error("fmi_dev_aggregate: workspace/output too small (see fmi_dev_aggregate_sizes)"); return FMI_ERR_ARG; }
    AggView v = make_view(H, (const uint8_t *)d_plan_blob);
    // the sort keys are as wide as THIS index needs (a radix pass per 8 bits: NQ size sorts 37 / 30 bits where the format's maxima are 46 / 37)
    v.pos_bits = bits_for(h->n + 256 + H.max_key_len);
    v.doc_bits = bits_for(h->dev.n_begin);
    const uint64_t N = H.total_occ;
    const uint32_t nq = (uint32_t)H.nq;
    // tools (fmi_dev_debug_marks): marks[1] = 100 * call + stage after every launch, so that a stalled stream names its launch
    static uint32_t call_no = 0;
    const uint32_t call_base = 100u * (++call_no);
    auto mark = [&](uint32_t stage) { if (h->dbg_marks) hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, st, h->dbg_marks + 1, call_base + stage); };
    // stage timing (fmi_dev_agg_timing; measurement passes only): an event after every stage of this call
    const bool timing = h->agg_timing_enabled && h->agg_calls.size() < 64;
    size_t ev_base = 0;
    int ev_next = 0;
    if (timing) {
        ev_base = h->agg_events.size();
        for (int i = 0; i <= FMI_AGG_STAGES; i++) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->agg_events.push_back((void *)e); }
    }
    auto stage_done = [&]() { if (timing && ev_next <= FMI_AGG_STAGES) (void)hipEventRecord((hipEvent_t)h->agg_events[ev_base + ev_next++], st); };
    mark(1);
    HIPCHK(hipMemsetAsync(d_out, 0, L.fixed_bytes, st));
    HIPCHK(hipMemsetAsync(w.top_cnt, 0, nq * 4, st));
    HIPCHK(hipMemsetAsync(w.pool_cursor, 0, 32, st));          // [0]: pool cursor, [4]: error word of the first stage
    if (N) {
        // ---- first stage ----
        uint64_t *ka = w.k0, *kb = w.k1;
        uint32_t *va = w.v0, *vb = w.v1;
        mark(2);
        stage_done();                    // [start]
        hipLaunchKernelGGL(k_agg_locate, dim3(blocks_for(N, 256)), dim3(256), 0, st, h->dev, v, ka, va);
        mark(3);
        stage_done();                    // 0: k_agg_locate
        if ((rc = sort_pairs(w, ka, kb, va, vb, N, v.pos_bits + bits_for(nq - 1), st))) return rc;
        mark(4);
        stage_done();                    // 1: sort by (query, position)
        // (from here on everything is in that order: w.occ_rk = the occurrences' rare keys, w.doc = their documents, w.newflag = their new flags)
        hipLaunchKernelGGL(k_occ_prepare, dim3(blocks_for(N, 256)), dim3(256), 0, st, h->dev, v, ka, va, w.occ_rk, w.doc, w.M, w.state);
        hipLaunchKernelGGL(k_mis, dim3(blocks_for(N, MIS_CHUNK)), dim3(256), 0, st, ka, w.M, va, w.state, w.newflag, (uint32_t)N, (uint32_t)H.max_key_len,
                           w.pool_cursor + 4);
        mark(5);
        stage_done();                    // 2: documents + coverage (k_occ_prepare + k_mis)
        mark(6);
        stage_done();                    // 3: (the sort by (query, document) of rounds 2-5: gone -- the order by position groups by document)
        hipLaunchKernelGGL(k_heads, dim3(blocks_for(N, 256)), dim3(256), 0, st, ka, w.doc, v.pos_bits, w.head, N);
        size_t tb = w.rp_bytes;
        HIPCHK(rocprim::exclusive_scan(w.rp_tmp, tb, w.head, w.eid, 0u, N, rocprim::plus<uint32_t>(), st));
        mark(7);
        hipLaunchKernelGGL(k_entry_starts, dim3(blocks_for(N, 256)), dim3(256), 0, st, w.head, w.eid, w.estart, w.n_entries, N);
        const uint32_t cover_words = (uint32_t)((H.max_u + 31) / 32) + 1;
        stage_done();                    // 4: entry boundaries (k_heads, scan, k_entry_starts)
        // (scratch of the entries' members in processing order: w.tmp32 / w.head / w.state are free here)
        hipLaunchKernelGGL(k_entries, dim3((unsigned)std::min<uint64_t>(blocks_for(N, 4), 4096)), dim3(256), 4 * cover_words * 4, st, v, ka, va, w.doc,
                           w.estart, w.n_entries, w.occ_rk, w.newflag, w.tmp32, w.head, w.state, allow_overlaps, beta, single_key, cover_words, w.ckey, w.cscore,
                           w.ent_nkeys, w.ent_rank, w.ent_first, w.ent_q, w.ent_doc, w.ent_score, w.ent_best);
        mark(8);
        stage_done();                    // 5: k_entries
        // ---- ranking: sorted(first_stage.items(), key=...)[:n_top] per query, as a selection (k_select_top) ----
        if (h->opt.agg_rank_by_sorts == 0) {
            // the streaming passes on SEL_G workgroups per query, the end game on one (k_sel_*)
            SelWork sw{w.sel_mm, w.sel_hist, w.sel_cnt, w.sel_qstart, w.sel_sure, w.sel_maybe, (uint32_t)n_top, nq};
            HIPCHK(hipMemsetAsync(w.sel_mm, 0xFF, (size_t)nq * 8, st));
            HIPCHK(hipMemsetAsync(w.sel_mm + nq, 0, w.sel_zero_bytes, st));
            uint32_t p2 = 1;
            while (p2 < n_top) p2 <<= 1;
            const size_t fixed = (size_t)p2 * 16 + SEL_WAVES * 256 * 4 + 256 * 4 + 64;
            const uint32_t fcache = (uint32_t)std::min<size_t>(SEL_MAYBE_CAP, (160 * 1024 > fixed ? 160 * 1024 - fixed : 0) / 16);
            const size_t flds = std::max(fixed + (size_t)fcache * 16, select_lds_bytes((uint32_t)n_top, 0));
            if (flds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_sel_final, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
            hipLaunchKernelGGL(k_sel_minmax, dim3(nq * SEL_G), dim3(SEL_GT), 0, st, w.n_entries, w.ent_q, w.ent_rank, sw);
            for (uint32_t pass = 0; pass < SEL_PASSES; pass++)
                hipLaunchKernelGGL(k_sel_hist, dim3(nq * SEL_G), dim3(SEL_GT), 0, st, pass, w.ent_rank, w.ent_first, sw);
            hipLaunchKernelGGL(k_sel_compact, dim3(nq * SEL_G), dim3(SEL_GT), 0, st, w.ent_rank, w.ent_first, sw);
            hipLaunchKernelGGL(k_sel_final, dim3(nq), dim3(SEL_WG), flds, st, w.n_entries, w.ent_q, w.ent_rank, w.ent_first, w.ent_doc, w.ent_score, sw, fcache, p2,
                               w.top_doc, w.top_ent, w.top_cnt, L.o.fs_doc, L.o.fs_score, L.o.fs_cnt);
            mark(12);
            stage_done();                    // 6: ranking (k_sel_minmax, k_sel_hist x 2, k_sel_compact, k_sel_final)
        } else if (h->opt.agg_rank_by_sorts == 2) {
            // (the whole selection by ONE workgroup per query: what k_sel_final falls back to for a tie group that overflows; tests run it on its own)
            const uint32_t sel_cache = select_cache_for((uint32_t)n_top);
            const size_t sel_lds = select_lds_bytes((uint32_t)n_top, sel_cache);
            if (sel_lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_select_top, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
            hipLaunchKernelGGL(k_select_top, dim3(nq), dim3(SEL_WG), sel_lds, st, w.n_entries, w.ent_q, w.ent_rank, w.ent_first, w.ent_doc, w.ent_score,
                               (uint32_t)n_top, sel_cache, w.top_doc, w.top_ent, w.top_cnt, L.o.fs_doc, L.o.fs_score, L.o.fs_cnt);
            mark(12);
            stage_done();                    // 6: ranking (k_select_top)
        } else {
        // (the form of rounds 2-5, kept as the checker of the selection: three stable sorts -- first touch, rank key, query -- over every entry slot)
        uint32_t *fa = w.ent_first, *fb = w.tmp32;
        va = w.v0; vb = w.v1;
        hipLaunchKernelGGL(k_pad_entries, dim3(blocks_for(N, 256)), dim3(256), 0, st, w.n_entries, w.ent_first, w.ent_rank, w.ent_q, nq, va, N);
        if ((rc = sort_pairs(w, fa, fb, va, vb, N, 32, st))) return rc;
        mark(9);
        ka = w.k0; kb = w.k1;
        hipLaunchKernelGGL(k_gather<uint64_t>, dim3(blocks_for(N, 256)), dim3(256), 0, st, w.ent_rank, va, ka, N);
        if ((rc = sort_pairs(w, ka, kb, va, vb, N, 64, st))) return rc;
        mark(10);
        uint32_t *qa = w.head, *qb = w.eid;                   // head/eid are free by now
        hipLaunchKernelGGL(k_gather<uint32_t>, dim3(blocks_for(N, 256)), dim3(256), 0, st, w.ent_q, va, qa, N);
        if ((rc = sort_pairs(w, qa, qb, va, vb, N, bits_for(nq), st))) return rc;
        mark(11);
        hipLaunchKernelGGL(k_top_docs, dim3(nq), dim3(256), 0, st, qa, va, w.ent_doc, N, nq, (uint32_t)n_top, w.ent_score, w.top_doc, w.top_ent, w.top_cnt,
                           L.o.fs_doc, L.o.fs_score, L.o.fs_cnt);
        mark(12);
        stage_done();                    // 6: ranking (three stable sorts + k_top_docs)
        }
    } else if (timing) {
        for (int i = 0; i < 8; i++) stage_done();
    }
    // ---- full scoring ----
    HIPCHK(hipMemsetAsync(w.type_dense, 0, (uint64_t)nq * H.vocab * 8, st));
    HIPCHK(hipMemsetAsync(w.tok2local, 0xFF, (uint64_t)nq * H.vocab * 4, st));
    if (H.n_uni) hipLaunchKernelGGL(k_scatter_unigrams, dim3(blocks_for(H.n_uni, 256)), dim3(256), 0, st, v, H.n_uni, w.type_dense);
    if (H.n_tok) hipLaunchKernelGGL(k_scatter_local_ids, dim3(blocks_for(H.n_tok, 256)), dim3(256), 0, st, v, H.n_tok, w.tok2local);
    ScoreParams p{};
    p.allow_overlaps = allow_overlaps; p.single_key_add_unigrams = single_key_add_unigrams; p.unigrams_ignore_free_places = unigrams_ignore_free_places;
    p.beta = beta; p.single_key = single_key; p.shift = shift;
    p.n_top = (uint32_t)n_top; p.t_cap = (uint32_t)h->max_doc_len; p.cover_words = (uint32_t)((H.max_u + 31) / 32) + 1;
    p.cand_cap = AGG_CAND_CAP; p.pick_cap = pc;
    const size_t per_wave = (score_lds_words(p.t_cap, p.cover_words, p.cand_cap) + 1) & ~(size_t)1;
    const size_t lds_bytes = 4 * per_wave * 4;
    if (lds_bytes > 160 * 1024) { fmi_set_error("fmi_dev_aggregate: %zu bytes of LDS per workgroup needed", lds_bytes); return FMI_ERR_ARG; }
    if (lds_bytes > 64 * 1024) {
        HIPCHK(hipFuncSetAttribute((const void *)k_full_score<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIPCHK(hipFuncSetAttribute((const void *)k_full_score<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    OverflowPool pool{w.pool_sk, w.pool_key, w.pool_cursor, (uint32_t)w.pool_cap, w.pool_cursor + 4};
    ScoreOut so = L.o;
    so.stage_id = w.stage_id; so.stage_score = w.stage_score;
    p.per_q = (uint32_t)n_top;
    stage_done();                        // 7: per-query token tables (memsets + scatters)
    hipLaunchKernelGGL(k_full_score<false>, dim3(blocks_for((uint64_t)nq * n_top, 4)), dim3(256), lds_bytes, st, h->dev, v, p, w.top_doc, w.top_cnt,
                       (const uint32_t *)nullptr, w.type_dense, w.tok2local, w.scores, so, pool);
    mark(13);
    stage_done();                        // 8: k_full_score over the ranked documents
    uint32_t rank_p2 = 1;
    while (rank_p2 < n_top) rank_p2 <<= 1;
    if ((size_t)rank_p2 * 12 > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)k_rank_docs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(rank_p2 * 12)));
    hipLaunchKernelGGL(k_rank_docs, dim3(nq), dim3(1024), (size_t)rank_p2 * 12, st, w.scores, w.top_cnt, (uint32_t)n_top, rank_p2, w.order);
    mark(14);
    stage_done();                        // 9: k_rank_docs
    p.per_q = (uint32_t)keep;
    hipLaunchKernelGGL(k_full_score<true>, dim3(blocks_for((uint64_t)nq * keep, 4)), dim3(256), lds_bytes, st, h->dev, v, p, w.top_doc, w.top_cnt,
                       w.order, w.type_dense, w.tok2local, w.scores, so, pool);
    mark(15);
    stage_done();                        // 10: k_full_score again over the caller's top-k (records what goes back)
    HIPCHK(hipGetLastError());
    if (timing) {
        // what the stages processed (read back: this is a measurement pass)
        HIPCHK(hipStreamSynchronize(st));
        uint32_t ne = 0;
        std::vector<uint32_t> cnt(nq);
        if (N) HIPCHK(hipMemcpy(&ne, w.n_entries, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cnt.data(), w.top_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost));
        fmi::AggCall c{N, ne, 0, 0, 0};
        std::vector<uint32_t> docs((size_t)nq * n_top);
        HIPCHK(hipMemcpy(docs.data(), w.top_doc, docs.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < nq; q++) {
            const uint32_t m = std::min<uint32_t>(cnt[q], (uint32_t)n_top);
            c.docs_scored += m; c.docs_kept += std::min<uint64_t>(m, keep);
            for (uint32_t j = 0; j < m; j++) {
                const uint32_t d = docs[(size_t)q * n_top + j];
                if ((uint64_t)d + 1 < h->doc_begin.size()) c.doc_tokens += h->doc_begin[d + 1] - h->doc_begin[d];
            }
        }
        h->agg_calls.push_back(c);
    }
    return FMI_OK;
}

extern "C" int fmi_dev_agg_timing(fmi_t *h, int enable)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    for (void *e : h->agg_events) (void)hipEventDestroy((hipEvent_t)e);
    h->agg_events.clear(); h->agg_calls.clear();
    h->agg_timing_enabled = enable;
    return FMI_OK;
}

extern "C" int fmi_dev_read_agg_timing(fmi_t *h, double *stage_ms, uint64_t *counts, uint64_t *calls_out)
{
    if (!h || !stage_ms || !counts || !calls_out) { fmi_set_error("null argument"); return FMI_ERR_ARG; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < FMI_AGG_STAGES; i++) stage_ms[i] = 0.0;
    for (int i = 0; i < 5; i++) counts[i] = 0;
    for (size_t c = 0; c < h->agg_calls.size(); c++) {
        for (int i = 0; i < FMI_AGG_STAGES; i++) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)h->agg_events[c * (FMI_AGG_STAGES + 1) + i], (hipEvent_t)h->agg_events[c * (FMI_AGG_STAGES + 1) + i + 1]));
            stage_ms[i] += ms;
        }
        const fmi::AggCall &a = h->agg_calls[c];
        counts[0] += a.rows; counts[1] += a.entries; counts[2] += a.docs_scored; counts[3] += a.docs_kept; counts[4] += a.doc_tokens;
    }
    *calls_out = h->agg_calls.size();
    for (void *e : h->agg_events) (void)hipEventDestroy((hipEvent_t)e);
    h->agg_events.clear(); h->agg_calls.clear();
    return FMI_OK;
}
