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
//   k_agg_locate        row -> (text position, document) for every occurrence i (i = the reference's processing
//                       order: keys by descending score, rows ascending); suffix array + boundaries are resident
//   radix sort by (query, position); k_mis: an occurrence is "new" iff no EARLIER-processed new occurrence overlaps
//                       its window [pos - len, pos) -- the greedy maximal independent set of the interval graph in
//                       priority order, resolved cluster by cluster (clusters = connected runs of overlapping
//                       windows, independent of each other) with a parallel fixed point: a vertex is decided once
//                       all its higher-priority neighbours are
//   radix sort by (query, document) (stable: occurrences of a document stay in processing order); k_entries: one
//                       wave per document: first touch, best key, the keys that count once per document
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
// occurrence i -> rare key (binary search over the occurrence offsets), row, text position, document
__global__ __launch_bounds__(256) void k_agg_locate(FmiDev ix, AggView v, uint32_t *occ_rk, uint32_t *doc, uint64_t *key_pos, uint32_t *val)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.total) return;
    uint32_t a = 0, b = v.n_rare;          // last r with rare_occ_off[r] <= i
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (v.rare_occ_off[mid] <= i) a = mid; else b = mid; }
    const uint32_t k = v.rare_key[a];
    const uint64_t row = v.key_lo[k] + (i - v.rare_occ_off[a]);
    const uint64_t pos = sa_at(ix, row);
    occ_rk[i] = a;
    doc[i] = (uint32_t)doc_of(ix, pos);
    key_pos[i] = ((uint64_t)v.key_q[k] << FMI_AGG_POS_BITS) + pos + 256;      // window [pos - len, pos) never goes below 0 after the offset
    val[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_mis_prepare(AggView v, const uint32_t *sorted_val, const uint32_t *occ_rk, uint16_t *M, uint8_t *state)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.total) return;
    const uint32_t m = v.key_len[v.rare_key[occ_rk[sorted_val[j]]]];
    M[j] = (uint16_t)m;
    state[j] = 0;
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
    for (uint32_t j = a + threadIdx.x; j < b; j += blockDim.x) newflag[PRI[j]] = state[j] == ST_NEW;
}

__global__ __launch_bounds__(256) void k_doc_keys(AggView v, const uint32_t *occ_rk, const uint32_t *doc, uint64_t *key_doc, uint32_t *val)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.total) return;
    key_doc[i] = ((uint64_t)v.key_q[v.rare_key[occ_rk[i]]] << 32) | doc[i];
    val[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_heads(const uint64_t *KD, uint32_t *head, uint64_t total)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    head[j] = (j == 0 || KD[j] != KD[j - 1]) ? 1u : 0u;
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
__device__ __forceinline__ void entry_by_wave(const AggView &v, const uint64_t *KD, const uint32_t *ID, const uint32_t *occ_rk, const uint8_t *newflag,
                                              int allow_overlaps, double beta, double single_key, uint32_t cover_words, uint32_t *cover,
                                              uint32_t e, uint32_t s, uint32_t t, uint32_t *ckey, double *cscore, uint32_t *ent_nkeys,
                                              uint64_t *ent_rank, uint32_t *ent_first, uint32_t *ent_q, uint32_t *ent_doc, double *ent_score, uint32_t *ent_best)
{
    const uint32_t lane = threadIdx.x & 63;
    // ---- keys that count: first occurrence of each key that is new (or any, with allow_overlaps) ----
    uint32_t nL = 0, carry = 0xFFFFFFFFu;          // carry = rare key of the last counting occurrence so far
    for (uint32_t base = s; base < t; base += 64) {
        const uint32_t j = base + lane;
        const bool valid = j < t;
        const uint32_t i = valid ? ID[j] : 0;
        const uint32_t r = valid ? occ_rk[i] : 0xFFFFFFFEu;
        const bool c = valid && (allow_overlaps || newflag[i]);
        uint32_t prev = (uint32_t)__shfl_up((int)r, 1);
        if (lane == 0) prev = (base > s) ? occ_rk[ID[base - 1]] : 0xFFFFFFFFu;
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
        const uint64_t kd = KD[s];
        const uint32_t i0 = ID[s], r0 = occ_rk[i0];
        const double best = v.key_score[v.rare_key[r0]];     // keys arrive by descending score: the first one to touch the document
        const double rk = (1.0 - single_key) * (-current) + single_key * (-best);
        ent_nkeys[e] = nL; ent_score[e] = current; ent_best[e] = r0; ent_first[e] = i0;
        ent_q[e] = (uint32_t)(kd >> 32); ent_doc[e] = (uint32_t)kd; ent_rank[e] = f64_order_key(rk);
    }
}

__global__ __launch_bounds__(256) void k_entries(AggView v, const uint64_t *KD, const uint32_t *ID, const uint32_t *estart, const uint32_t *n_entries_p,
                                                 const uint32_t *occ_rk, const uint8_t *newflag, int allow_overlaps, double beta, double single_key,
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
            const uint32_t i0 = ID[s], r0 = occ_rk[i0];
            const bool c = allow_overlaps || newflag[i0];
            const double best = v.key_score[v.rare_key[r0]];
            double current = 0.0;
            if (c) { ckey[s] = r0; cscore[s] = best; current += best; }
            const uint64_t kd = KD[s];
            const double rk = (1.0 - single_key) * (-current) + single_key * (-best);
            ent_nkeys[e] = c ? 1u : 0u; ent_score[e] = current; ent_best[e] = r0; ent_first[e] = i0;
            ent_q[e] = (uint32_t)(kd >> 32); ent_doc[e] = (uint32_t)kd; ent_rank[e] = f64_order_key(rk);
        }
        uint64_t multi = __ballot(have && t - s != 1);
        while (multi) {
            const uint32_t l = (uint32_t)__builtin_ctzll(multi);
            multi &= multi - 1;
            entry_by_wave(v, KD, ID, occ_rk, newflag, allow_overlaps, beta, single_key, cover_words, cover, e0 + l,
                          rfl((uint32_t)__shfl((int)s, (int)l)), rfl((uint32_t)__shfl((int)t, (int)l)), ckey, cscore, ent_nkeys, ent_rank, ent_first,
                          ent_q, ent_doc, ent_score, ent_best);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_gather(const T *src, const uint32_t *idx, T *dst, uint64_t n)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) dst[x] = src[idx[x]];
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

// sorted(results.items(), key=lambda x: -x[1][0]) (keys.py:496): stable, one workgroup per query
__global__ __launch_bounds__(1024) void k_rank_docs(const double *scores, const uint32_t *top_cnt, uint32_t n_top, uint32_t *order)
{
    extern __shared__ uint64_t keys[];
    const uint32_t q = blockIdx.x, cnt = top_cnt[q];
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) keys[i] = f64_order_key(-scores[(uint64_t)q * n_top + i]);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const uint64_t my = keys[i];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < cnt; y++) { const uint64_t o = keys[y]; rank += (o < my) || (o == my && y < i); }
        order[(uint64_t)q * n_top + rank] = i;
    }
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
static constexpr uint32_t AGG_CAND_CAP = 384;          // occurrences of keys per document held in LDS before the pool is used

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

static unsigned bits_for(uint64_t max_value) { unsigned b = 1; while (b < 64 && (max_value >> b)) b++; return b; }

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
    const AggView v = make_view(H, (const uint8_t *)d_plan_blob);
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
        hipLaunchKernelGGL(k_agg_locate, dim3(blocks_for(N, 256)), dim3(256), 0, st, h->dev, v, w.occ_rk, w.doc, ka, va);
        mark(3);
        stage_done();                    // 0: k_agg_locate
        if ((rc = sort_pairs(w, ka, kb, va, vb, N, FMI_AGG_POS_BITS + bits_for(nq - 1), st))) return rc;
        mark(4);
        stage_done();                    // 1: sort by (query, position)
        hipLaunchKernelGGL(k_mis_prepare, dim3(blocks_for(N, 256)), dim3(256), 0, st, v, va, w.occ_rk, w.M, w.state);
        hipLaunchKernelGGL(k_mis, dim3(blocks_for(N, MIS_CHUNK)), dim3(256), 0, st, ka, w.M, va, w.state, w.newflag, (uint32_t)N, (uint32_t)H.max_key_len,
                           w.pool_cursor + 4);
        mark(5);
        stage_done();                    // 2: coverage (k_mis_prepare + k_mis)
        hipLaunchKernelGGL(k_doc_keys, dim3(blocks_for(N, 256)), dim3(256), 0, st, v, w.occ_rk, w.doc, ka, va);
        if ((rc = sort_pairs(w, ka, kb, va, vb, N, 32 + bits_for(nq - 1), st))) return rc;
        mark(6);
        stage_done();                    // 3: k_doc_keys + sort by (query, document)
        hipLaunchKernelGGL(k_heads, dim3(blocks_for(N, 256)), dim3(256), 0, st, ka, w.head, N);
        size_t tb = w.rp_bytes;
        HIPCHK(rocprim::exclusive_scan(w.rp_tmp, tb, w.head, w.eid, 0u, N, rocprim::plus<uint32_t>(), st));
        mark(7);
        hipLaunchKernelGGL(k_entry_starts, dim3(blocks_for(N, 256)), dim3(256), 0, st, w.head, w.eid, w.estart, w.n_entries, N);
        const uint32_t cover_words = (uint32_t)((H.max_u + 31) / 32) + 1;
        stage_done();                    // 4: entry boundaries (k_heads, scan, k_entry_starts)
        hipLaunchKernelGGL(k_entries, dim3((unsigned)std::min<uint64_t>(blocks_for(N, 4), 4096)), dim3(256), 4 * cover_words * 4, st, v, ka, va,
                           w.estart, w.n_entries, w.occ_rk, w.newflag, allow_overlaps, beta, single_key, cover_words, w.ckey, w.cscore,
                           w.ent_nkeys, w.ent_rank, w.ent_first, w.ent_q, w.ent_doc, w.ent_score, w.ent_best);
        mark(8);
        stage_done();                    // 5: k_entries
        // ---- ranking: stable sorts by first touch, then rank key, then query = sorted(first_stage.items(), key=...) ----
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
    if (n_top * 8 > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)k_rank_docs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(n_top * 8)));
    hipLaunchKernelGGL(k_rank_docs, dim3(nq), dim3(1024), n_top * 8, st, w.scores, w.top_cnt, (uint32_t)n_top, w.order);
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
