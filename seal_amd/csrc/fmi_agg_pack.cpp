// Host side of the GPU evidence aggregation: packs the scored keys of a chunk of queries (what
// seal/keys.py:193-309 leaves in `rare_ngrams` / `all_ngrams` / `unigram_scores`) into ONE blob that is copied to the
// GPU with a single transfer -- token ids re-based per query, distinct-token sets per key (repetition(),
// keys.py:186-191), the heap order of keys (keys.py:431 compares (-score, token tuple)), a hash-table trie of the keys
// per query (keys.py:377-384) and the non-zero unigram scores.  No scoring happens here.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "fmi_agg.h"
#include "fmi_internal.h"

struct fmi_agg_plan {
    std::vector<uint8_t> blob;
};

namespace {

struct Builder {
    std::vector<uint8_t> &b;
    explicit Builder(std::vector<uint8_t> &buf) : b(buf) {}
    template <class T>
    uint64_t put(const std::vector<T> &v)
    {
        const uint64_t off = (b.size() + 15) & ~(uint64_t)15;
        b.resize(off + std::max<size_t>(v.size(), 1) * sizeof(T), 0);
        if (!v.empty()) memcpy(b.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
};

struct Slot { uint32_t node, tok, child, key; };

}  // namespace

extern "C" int fmi_agg_pack(uint64_t n_queries, const int64_t *q_key_off, const int64_t *key_tok_off, const int64_t *key_toks,
                            const double *key_score, const uint8_t *key_rare, const uint64_t *key_lo, const uint64_t *key_hi,
                            uint64_t max_hits, uint64_t index_size, const double *const *type_scores, uint64_t vocab,
                            fmi_agg_plan **out)
{
    if (!out || !q_key_off || n_queries == 0 || n_queries > FMI_AGG_MAX_QUERIES) {
        fmi_set_error("fmi_agg_pack: bad argument (1..%u queries per plan)", FMI_AGG_MAX_QUERIES);
        return FMI_ERR_ARG;
    }
    const uint64_t nq = n_queries, nk = (uint64_t)q_key_off[nq];
    if (nk && (!key_tok_off || !key_toks || !key_score || !key_rare || !key_lo || !key_hi)) { fmi_set_error("fmi_agg_pack: null key arrays"); return FMI_ERR_ARG; }
    if (nk >= (1ull << 31)) { fmi_set_error("fmi_agg_pack: too many keys"); return FMI_ERR_ARG; }
    FmiAggHeader H{};
    H.magic = FMI_AGG_MAGIC;
    H.nq = nq; H.n_keys = nk; H.vocab = vocab;
    std::vector<uint32_t> q_key(nq + 1), q_rare(nq + 1, 0), rare_key, key_len(nk), key_q(nk), key_rank(nk), kset_off(nk + 1, 0), kset_ids,
        q_tok(nq + 1, 0), tok_list, q_trie(nq + 1, 0);
    std::vector<uint64_t> rare_occ{0}, lo(nk), uni_flat;
    std::vector<double> score(nk), uni_score;
    std::vector<Slot> trie;
    for (uint64_t q = 0; q <= nq; q++) q_key[q] = (uint32_t)q_key_off[q];
    for (uint64_t k = 0; k < nk; k++) {
        const int64_t m = key_tok_off[k + 1] - key_tok_off[k];
        if (m < 1 || m > (int64_t)FMI_AGG_MAX_KEY_LEN) { fmi_set_error("fmi_agg_pack: key of %lld tokens (1..%u supported)", (long long)m, FMI_AGG_MAX_KEY_LEN); return FMI_ERR_ARG; }
        if (!(key_score[k] > 0.0)) { fmi_set_error("fmi_agg_pack: only keys with a positive score take part (keys.py:378)"); return FMI_ERR_ARG; }
        key_len[k] = (uint32_t)m;
        score[k] = key_score[k];
        lo[k] = key_lo[k];
        H.max_key_len = std::max<uint64_t>(H.max_key_len, (uint64_t)m);
        for (int64_t t = key_tok_off[k]; t < key_tok_off[k + 1]; t++)
            if (key_toks[t] < 0 || key_toks[t] >= (int64_t)0x7FFFFFFF) { fmi_set_error("fmi_agg_pack: token id out of range"); return FMI_ERR_ARG; }
    }
    std::vector<uint32_t> local, order;
    std::vector<int64_t> uniq;
    for (uint64_t q = 0; q < nq; q++) {
        const uint32_t k0 = q_key[q], k1 = q_key[q + 1];
        H.max_q_keys = std::max<uint64_t>(H.max_q_keys, k1 - k0);
        // query-local token ids: the distinct tokens of the query's keys, ascending
        uniq.assign(key_toks + key_tok_off[k0], key_toks + key_tok_off[k1]);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        for (int64_t t : uniq) tok_list.push_back((uint32_t)t);
        q_tok[q + 1] = (uint32_t)tok_list.size();
        H.max_u = std::max<uint64_t>(H.max_u, uniq.size());
        // rare keys in processing order + their occurrence slots; distinct-token sets; query of a key
        for (uint32_t k = k0; k < k1; k++) {
            key_q[k] = (uint32_t)q;
            local.clear();
            for (int64_t t = key_tok_off[k]; t < key_tok_off[k + 1]; t++)
                local.push_back((uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), key_toks[t]) - uniq.begin()));
            std::sort(local.begin(), local.end());
            local.erase(std::unique(local.begin(), local.end()), local.end());
            kset_ids.insert(kset_ids.end(), local.begin(), local.end());
            kset_off[k + 1] = (uint32_t)kset_ids.size();
            if (key_rare[k]) {
                // rows past the end of the index (quirk Q1 can hand out size()+1 as an upper end) have no text position:
                // the reference's locate returns -1 there and its get_doc then fails; they are left out
                const uint64_t hi = std::min<uint64_t>(key_hi[k], index_size);
                const uint64_t cnt = hi > key_lo[k] ? hi - key_lo[k] : 0;
                rare_key.push_back(k);
                rare_occ.push_back(rare_occ.back() + std::min<uint64_t>(cnt, max_hits));
            }
        }
        q_rare[q + 1] = (uint32_t)rare_key.size();
        // heap order of keys.py:431: (-score, token tuple) ascending
        order.resize(k1 - k0);
        std::iota(order.begin(), order.end(), k0);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            if (score[a] != score[b]) return score[a] > score[b];
            return std::lexicographical_compare(key_toks + key_tok_off[a], key_toks + key_tok_off[a + 1],
                                                key_toks + key_tok_off[b], key_toks + key_tok_off[b + 1]);
        });
        for (uint32_t r = 0; r < order.size(); r++) key_rank[order[r]] = r;
        // trie of the query's keys as an open-addressing table over (parent node, token)
        uint64_t edges = (uint64_t)(key_tok_off[k1] - key_tok_off[k0]);
        uint64_t cap = 16;
        while (cap < 2 * edges + 2) cap <<= 1;
        const uint64_t base = trie.size();
        trie.resize(base + cap, Slot{FMI_AGG_TRIE_EMPTY, 0, 0, FMI_AGG_TRIE_EMPTY});
        const uint32_t mask = (uint32_t)(cap - 1);
        uint32_t n_nodes = 1;
        for (uint32_t k = k0; k < k1; k++) {
            uint32_t node = 0;
            Slot *last = nullptr;
            for (int64_t t = key_tok_off[k]; t < key_tok_off[k + 1]; t++) {
                const uint32_t tok = (uint32_t)key_toks[t];
                for (uint32_t i = fmi_agg_trie_hash(node, tok) & mask;; i = (i + 1) & mask) {
                    Slot &s = trie[base + i];
                    if (s.node == FMI_AGG_TRIE_EMPTY) { s.node = node; s.tok = tok; s.child = n_nodes++; last = &s; break; }
                    if (s.node == node && s.tok == tok) { last = &s; break; }
                }
                node = last->child;
            }
            last->key = k;       // keys are distinct within a query (all_ngrams is a dict)
        }
        q_trie[q + 1] = (uint32_t)trie.size();
        // non-zero unigram scores (the dense table is rebuilt on the device)
        if (type_scores && type_scores[q])
            for (uint64_t t = 0; t < vocab; t++)
                if (type_scores[q][t] != 0.0) { uni_flat.push_back(q * vocab + t); uni_score.push_back(type_scores[q][t]); }
    }
    if (trie.size() >= (1ull << 32) || tok_list.size() >= (1ull << 32)) { fmi_set_error("fmi_agg_pack: plan too large"); return FMI_ERR_ARG; }
    H.n_rare = rare_key.size(); H.total_occ = rare_occ.back(); H.n_uni = uni_flat.size(); H.n_trie_slots = trie.size(); H.n_tok = tok_list.size();
    if (H.total_occ >= (1ull << 32) - 2) { fmi_set_error("fmi_agg_pack: more than 2^32 located rows in one plan"); return FMI_ERR_ARG; }
    fmi_agg_plan *plan = new fmi_agg_plan();
    plan->blob.resize(sizeof(FmiAggHeader));
    Builder B(plan->blob);
    H.o_q_key_off = B.put(q_key); H.o_q_rare_off = B.put(q_rare); H.o_rare_key = B.put(rare_key); H.o_rare_occ_off = B.put(rare_occ);
    H.o_key_lo = B.put(lo); H.o_key_len = B.put(key_len); H.o_key_q = B.put(key_q); H.o_key_rank = B.put(key_rank); H.o_key_score = B.put(score);
    H.o_kset_off = B.put(kset_off); H.o_kset_ids = B.put(kset_ids); H.o_q_tok_off = B.put(q_tok); H.o_tok_list = B.put(tok_list);
    H.o_q_trie_off = B.put(q_trie); H.o_trie = B.put(trie); H.o_uni_flat = B.put(uni_flat); H.o_uni_score = B.put(uni_score);
    plan->blob.resize((plan->blob.size() + 15) & ~(size_t)15);
    H.bytes = plan->blob.size();
    memcpy(plan->blob.data(), &H, sizeof(H));
    *out = plan;
    return FMI_OK;
}

extern "C" const void *fmi_agg_plan_blob(const fmi_agg_plan *p, uint64_t *bytes_out)
{
    if (!p) return nullptr;
    if (bytes_out) *bytes_out = p->blob.size();
    return p->blob.data();
}

extern "C" uint64_t fmi_agg_plan_occurrences(const fmi_agg_plan *p) { return p ? ((const FmiAggHeader *)p->blob.data())->total_occ : 0; }
extern "C" void fmi_agg_plan_free(fmi_agg_plan *p) { delete p; }
