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
    // filled by fmi_agg_score_pack: `all_ngrams` of every query in its order (keys.py:305-309) -- src = index of the key
    // in the query's input list, or -(token + 1) for a unigram added by keys.py:274-278 -- and, per table key, the same src
    std::vector<int64_t> ng_off, ng_src, table_src;
    std::vector<double> ng_score;
    std::vector<uint8_t> ng_rare;
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

using SparseScores = std::vector<std::vector<std::pair<uint32_t, double>>>;     // per query: (token, score != 0), ascending tokens

static int agg_pack_impl(uint64_t n_queries, const int64_t *q_key_off, const int64_t *key_tok_off, const int64_t *key_toks,
                         const double *key_score, const uint8_t *key_rare, const uint64_t *key_lo, const uint64_t *key_hi,
                         uint64_t max_hits, uint64_t index_size, const double *const *type_scores, const SparseScores *sparse,
                         uint64_t vocab, fmi_agg_plan **out)
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
        if (sparse) {
            for (const auto &ts : (*sparse)[q]) { uni_flat.push_back(q * vocab + ts.first); uni_score.push_back(ts.second); }
        } else if (type_scores && type_scores[q]) {
            for (uint64_t t = 0; t < vocab; t++)
                if (type_scores[q][t] != 0.0) { uni_flat.push_back(q * vocab + t); uni_score.push_back(type_scores[q][t]); }
        }
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

extern "C" int fmi_agg_pack(uint64_t n_queries, const int64_t *q_key_off, const int64_t *key_tok_off, const int64_t *key_toks,
                            const double *key_score, const uint8_t *key_rare, const uint64_t *key_lo, const uint64_t *key_hi,
                            uint64_t max_hits, uint64_t index_size, const double *const *type_scores, uint64_t vocab,
                            fmi_agg_plan **out)
{
    return agg_pack_impl(n_queries, q_key_off, key_tok_off, key_toks, key_score, key_rare, key_lo, key_hi, max_hits, index_size,
                         type_scores, nullptr, vocab, out);
}

extern "C" const void *fmi_agg_plan_blob(const fmi_agg_plan *p, uint64_t *bytes_out)
{
    if (!p) return nullptr;
    if (bytes_out) *bytes_out = p->blob.size();
    return p->blob.data();
}

extern "C" uint64_t fmi_agg_plan_occurrences(const fmi_agg_plan *p) { return p ? ((const FmiAggHeader *)p->blob.data())->total_occ : 0; }
extern "C" void fmi_agg_plan_free(fmi_agg_plan *p) { delete p; }


// ---------------------------------------------------------------------------
// Key scoring of aggregate_evidence (reference seal/keys.py:207-309) for a chunk of queries + fmi_agg_pack in one call:
// counts -> LM-vs-corpus log-odds (libm log/exp on doubles = python's math module; powers through libm pow = python's
// float `**`), the top-k unigrams by model log-probability and their scores, the best unigrams added as keys, the rare /
// frequent split and the three stable sorts by descending score.  Same values, same orders as the python it replaces
// (seal_amd/keys.py _aggregate_steps), which stays as the checker (tests/test_agg_pack.py) and for what this routine
// does not take (sort_by_length / sort_by_freq, empty keys).
// ---------------------------------------------------------------------------
#include <cmath>
#include <map>

namespace {

inline double py_max0(double o) { return (0.0 > o) ? 0.0 : o; }       // python's max(o, 0.0)

inline double log_odds(double sr, int64_t count, double ntokens, double smoothing)
{
    if (count == 0) return 0.0;
    const double snr = std::log(((double)count + smoothing) / (ntokens + smoothing));
    const double a = sr + std::log(1 - std::exp(snr));
    const double b = snr + std::log(1 - std::exp(sr));
    return a - b;
}

struct Scored { int64_t src; double sco; int64_t count; uint64_t lo, hi; };

}  // namespace

struct QueryScored {          // what scoring one query yields (merged in query order afterwards)
    std::vector<int64_t> ng_src, table_src, t_toks, t_len;
    std::vector<double> ng_score, t_score;
    std::vector<uint8_t> ng_rare, t_rare;
    std::vector<uint64_t> t_lo, t_hi;
    std::vector<std::pair<uint32_t, double>> us;     // non-zero unigram scores
    int err = 0;
};

struct ScoreArgs {
    const int64_t *q_key_off, *key_tok_off, *key_toks;
    const double *key_lm_score;
    const uint64_t *key_lo, *key_hi;
    const double *const *unigram_logprobs;
    uint64_t vocab;
    const int64_t *uni_lo, *uni_hi;
    uint64_t n_uni_table;
    double ntokens, alpha, length_penalty, smoothing;
    int use_fm, add_best;
    int64_t top_k;
    uint64_t max1, max2;
};

static void score_query(const ScoreArgs &A, uint64_t q, QueryScored &R)
{
    const int64_t k0 = A.q_key_off[q], k1 = A.q_key_off[q + 1];
    const int64_t nkeys = k1 - k0;
    double cutoff = 0.0;
    if (!A.use_fm) {
        if (nkeys == 0) { R.err = 1; return; }
        double mn = A.key_lm_score[k0];
        for (int64_t k = k0 + 1; k < k1; k++) if (A.key_lm_score[k] < mn) mn = A.key_lm_score[k];        // python's min()
        cutoff = mn - 0.1;
    }
    // ---- key scores (keys.py:207-234) ----
    std::vector<Scored> scored, rare, freq, all;
    for (int64_t k = k0; k < k1; k++) {
        const int64_t len = A.key_tok_off[k + 1] - A.key_tok_off[k];
        if (len < 1) { R.err = 2; return; }
        const int64_t count = A.key_hi[k] > A.key_lo[k] ? (int64_t)(A.key_hi[k] - A.key_lo[k]) : 0;
        const double f = std::pow(1.0 - A.length_penalty, (double)len - 1.0);
        double sco;
        if (A.use_fm) {
            const double sr_adj = (A.key_lm_score[k] - 1e-10) * f;
            sco = std::pow(py_max0(log_odds(sr_adj, count, A.ntokens, A.smoothing)), A.alpha);
        } else {
            sco = std::pow(py_max0(A.key_lm_score[k] - cutoff) * f, A.alpha);
        }
        if (count == 0) sco = 0.0;
        scored.push_back(Scored{k - k0, sco, count, A.key_lo[k], A.key_hi[k]});
    }
    // ---- unigram scores (keys.py:236-278) ----
    const double *raw = A.unigram_logprobs ? A.unigram_logprobs[q] : nullptr;
    if (raw) {
        const int64_t V = (int64_t)A.vocab;
        int64_t kk = A.top_k < V ? A.top_k : V;
        std::vector<int64_t> best;
        if (kk >= V) { for (int64_t i = 0; i < V; i++) best.push_back(i); }
        else if (kk > 0) {
            std::vector<double> part(raw, raw + V);
            std::nth_element(part.begin(), part.begin() + (V - kk), part.end());
            const double kth = part[(size_t)(V - kk)];
            for (int64_t i = 0; i < V; i++) if (raw[i] > kth) best.push_back(i);
            int64_t room = kk - (int64_t)best.size();
            for (int64_t i = 0; i < V && room > 0; i++) if (raw[i] == kth) { best.push_back(i); room--; }
        }
        std::vector<int64_t> seen_single{0, 1, 2};                       // seen_unigrams of keys.py:236-240
        for (int64_t k = k0; k < k1; k++)
            if (A.key_tok_off[k + 1] - A.key_tok_off[k] == 1) seen_single.push_back(A.key_toks[A.key_tok_off[k]]);
        std::sort(seen_single.begin(), seen_single.end());
        for (int64_t i : best) {
            if (std::binary_search(seen_single.begin(), seen_single.end(), i) || (uint64_t)i >= A.n_uni_table) continue;
            const int64_t cnt = A.uni_hi[i] - A.uni_lo[i];
            if (cnt <= 0) continue;
            double v;
            if (A.use_fm) {
                const double o = log_odds(raw[i], cnt, A.ntokens, A.smoothing);
                v = (o > 0.0) ? o : 0.0;                                   // np.maximum(., 0.0)
            } else {
                v = std::pow(py_max0(raw[i] - cutoff), A.alpha);
            }
            if (v != 0.0) R.us.emplace_back((uint32_t)i, v);
        }
        std::sort(R.us.begin(), R.us.end(), [](const std::pair<uint32_t, double> &a, const std::pair<uint32_t, double> &b) { return a.first < b.first; });
        if (A.add_best) {
            const size_t n_add = scored.size();
            // positives by descending score, ties to the lower id (stable over ascending ids); when there are fewer
            // positives than keys the reference goes on with zero-score ids, which its split then drops (sco == 0.0)
            std::vector<std::pair<uint32_t, double>> nz;
            for (const auto &ts : R.us) if (ts.second > 0) nz.push_back(ts);
            std::stable_sort(nz.begin(), nz.end(), [](const std::pair<uint32_t, double> &a, const std::pair<uint32_t, double> &b) { return a.second > b.second; });
            if (nz.size() > n_add) nz.resize(n_add);
            for (const auto &ts : nz) {
                const int64_t i = ts.first;
                scored.push_back(Scored{-(i + 1), ts.second, A.uni_hi[i] - A.uni_lo[i], (uint64_t)A.uni_lo[i], (uint64_t)A.uni_hi[i]});
            }
        }
    }
    // ---- rare / frequent split with dict semantics (keys.py:280-299), sorts (300-309) ----
    std::map<std::vector<int64_t>, std::pair<int, size_t>> where;          // ngram -> (0 rare / 1 freq, position)
    std::vector<int64_t> ng;
    for (const Scored &sc : scored) {
        if ((uint64_t)sc.count > A.max2 || sc.sco == 0.0) continue;
        if (sc.src >= 0) ng.assign(A.key_toks + A.key_tok_off[k0 + sc.src], A.key_toks + A.key_tok_off[k0 + sc.src + 1]);
        else ng.assign(1, -sc.src - 1);
        const int which = ((uint64_t)sc.count > A.max1 || sc.sco < 0.0) ? 1 : 0;
        std::vector<Scored> &dst = which ? freq : rare;
        auto it = where.find(ng);
        if (it != where.end() && it->second.first == which) { dst[it->second.second].sco = sc.sco; continue; }   // same key again: value replaced, place kept
        where[ng] = {which, dst.size()};
        dst.push_back(sc);
    }
    auto by_score = [](const Scored &a, const Scored &b) { return a.sco > b.sco; };
    std::stable_sort(rare.begin(), rare.end(), by_score);
    std::stable_sort(freq.begin(), freq.end(), by_score);
    const size_t n_rare = rare.size();
    all.assign(rare.begin(), rare.end());
    all.insert(all.end(), freq.begin(), freq.end());
    std::vector<size_t> order(all.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return all[a].sco > all[b].sco; });
    for (size_t i : order) {
        const Scored &sc = all[i];
        const uint8_t is_rare = i < n_rare;
        R.ng_src.push_back(sc.src); R.ng_score.push_back(sc.sco); R.ng_rare.push_back(is_rare);
        if (sc.sco > 0.0) {                                             // a table key (keys.py:378)
            if (sc.src >= 0) R.t_toks.insert(R.t_toks.end(), A.key_toks + A.key_tok_off[k0 + sc.src], A.key_toks + A.key_tok_off[k0 + sc.src + 1]);
            else R.t_toks.push_back(-sc.src - 1);
            R.t_len.push_back(sc.src >= 0 ? A.key_tok_off[k0 + sc.src + 1] - A.key_tok_off[k0 + sc.src] : 1);
            R.t_score.push_back(sc.sco); R.t_rare.push_back(is_rare); R.t_lo.push_back(sc.lo); R.t_hi.push_back(sc.hi);
            R.table_src.push_back(sc.src);
        }
    }
}

#include <thread>

extern "C" int fmi_agg_score_pack(uint64_t n_queries, const int64_t *q_key_off, const int64_t *key_tok_off, const int64_t *key_toks,
                                  const double *key_lm_score, const uint64_t *key_lo, const uint64_t *key_hi,
                                  const double *const *unigram_logprobs, uint64_t vocab, const int64_t *uni_lo, const int64_t *uni_hi,
                                  uint64_t n_uni_table, double ntokens, double alpha, double length_penalty, double smoothing,
                                  int use_fm_index_frequency, int add_best_unigrams_to_ngrams, int64_t use_top_k_unigrams,
                                  uint64_t max_occurrences_1, uint64_t max_occurrences_2, uint64_t index_size, fmi_agg_plan **out)
{
    if (!out || !q_key_off || n_queries == 0 || n_queries > FMI_AGG_MAX_QUERIES) { fmi_set_error("fmi_agg_score_pack: bad argument"); return FMI_ERR_ARG; }
    const uint64_t nq = n_queries;
    const ScoreArgs A{q_key_off, key_tok_off, key_toks, key_lm_score, key_lo, key_hi, unigram_logprobs, vocab, uni_lo, uni_hi, n_uni_table,
                      ntokens, alpha, length_penalty, smoothing, use_fm_index_frequency, add_best_unigrams_to_ngrams, use_top_k_unigrams,
                      max_occurrences_1, max_occurrences_2};
    // queries are independent: the libm-heavy unigram scoring (thousands of log/exp per query) runs on a few threads
    std::vector<QueryScored> R(nq);
    unsigned nt = std::thread::hardware_concurrency() / 2;
    if (const char *e = getenv("SEAL_HOST_THREADS")) nt = (unsigned)atoi(e);
    nt = std::max(1u, std::min<unsigned>(std::min<unsigned>(nt, 8u), (unsigned)nq));
    if (nt == 1) {
        for (uint64_t q = 0; q < nq; q++) score_query(A, q, R[q]);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] { for (uint64_t q = t; q < nq; q += nt) score_query(A, q, R[q]); });
        for (auto &x : th) x.join();
    }
    std::vector<int64_t> t_q_off(nq + 1, 0), t_tok_off{0}, t_toks, ng_off(nq + 1, 0), ng_src, table_src;
    std::vector<double> t_score, ng_score;
    std::vector<uint8_t> t_rare, ng_rare;
    std::vector<uint64_t> t_lo, t_hi;
    SparseScores sparse(nq);
    for (uint64_t q = 0; q < nq; q++) {
        QueryScored &r = R[q];
        if (r.err) { fmi_set_error(r.err == 1 ? "fmi_agg_score_pack: use_fm_index_frequency=False needs at least one key" : "fmi_agg_score_pack: empty key"); return FMI_ERR_ARG; }
        ng_src.insert(ng_src.end(), r.ng_src.begin(), r.ng_src.end());
        ng_score.insert(ng_score.end(), r.ng_score.begin(), r.ng_score.end());
        ng_rare.insert(ng_rare.end(), r.ng_rare.begin(), r.ng_rare.end());
        table_src.insert(table_src.end(), r.table_src.begin(), r.table_src.end());
        t_toks.insert(t_toks.end(), r.t_toks.begin(), r.t_toks.end());
        for (int64_t l : r.t_len) t_tok_off.push_back(t_tok_off.back() + l);
        t_score.insert(t_score.end(), r.t_score.begin(), r.t_score.end());
        t_rare.insert(t_rare.end(), r.t_rare.begin(), r.t_rare.end());
        t_lo.insert(t_lo.end(), r.t_lo.begin(), r.t_lo.end());
        t_hi.insert(t_hi.end(), r.t_hi.begin(), r.t_hi.end());
        sparse[q].swap(r.us);
        ng_off[q + 1] = (int64_t)ng_src.size();
        t_q_off[q + 1] = (int64_t)t_score.size();
    }
    if (t_toks.empty()) t_toks.push_back(0);
    if (t_score.empty()) { t_score.push_back(0.0); t_rare.push_back(0); t_lo.push_back(0); t_hi.push_back(0); }
    fmi_agg_plan *plan = nullptr;
    const int rc = agg_pack_impl(nq, t_q_off.data(), t_tok_off.data(), t_toks.data(), t_score.data(), t_rare.data(), t_lo.data(), t_hi.data(),
                                 max_occurrences_1, index_size, nullptr, unigram_logprobs ? &sparse : nullptr, vocab, &plan);
    if (rc != FMI_OK) return rc;
    plan->ng_off.swap(ng_off); plan->ng_src.swap(ng_src); plan->ng_score.swap(ng_score); plan->ng_rare.swap(ng_rare);
    plan->table_src.swap(table_src);
    *out = plan;
    return FMI_OK;
}

// all_ngrams of query q (null pointers: sizes only); table_src of the whole plan
extern "C" uint64_t fmi_agg_plan_ngrams(const fmi_agg_plan *p, uint64_t q, int64_t *src, double *score, uint8_t *rare)
{
    if (!p || q + 1 >= p->ng_off.size()) return 0;
    const int64_t a = p->ng_off[q], b = p->ng_off[q + 1];
    if (src) memcpy(src, p->ng_src.data() + a, (size_t)(b - a) * 8);
    if (score) memcpy(score, p->ng_score.data() + a, (size_t)(b - a) * 8);
    if (rare) memcpy(rare, p->ng_rare.data() + a, (size_t)(b - a));
    return (uint64_t)(b - a);
}

extern "C" uint64_t fmi_agg_plan_table_src(const fmi_agg_plan *p, int64_t *src)
{
    if (!p) return 0;
    if (src && !p->table_src.empty()) memcpy(src, p->table_src.data(), p->table_src.size() * 8);
    return p->table_src.size();
}
