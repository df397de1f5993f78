// Host-side first stage of evidence aggregation (reference seal/keys.py:311-364)
// over the (position, doc) arrays the GPU locate kernel produced.  This is the
// order-sensitive bookkeeping the reference does in pure Python per matching
// row; the arithmetic is float64 in the reference's operation order (compile
// with -ffp-contract=off), containers reproduce Python's insertion orders:
//   - keys arrive in the reference's processing order (descending score, stable);
//   - rows of a key in ascending row order;
//   - documents are remembered in first-touch order (dict insertion order);
//   - the final ranking is a stable sort.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "fmi_internal.h"

namespace {

struct PosSet {   // open-addressing set of text positions ("covered_points")
    std::vector<int64_t> slot;
    uint64_t mask = 0, used = 0;
    static constexpr int64_t EMPTY = INT64_MIN;
    explicit PosSet(uint64_t expect)
    {
        uint64_t cap = 64;
        while (cap < expect * 2 + 16) cap <<= 1;
        slot.assign(cap, EMPTY);
        mask = cap - 1;
    }
    static uint64_t hash(int64_t v) { uint64_t x = (uint64_t)v * 0x9E3779B97F4A7C15ull; return x ^ (x >> 29); }
    bool has(int64_t v) const
    {
        for (uint64_t i = hash(v) & mask;; i = (i + 1) & mask) {
            if (slot[i] == v) return true;
            if (slot[i] == EMPTY) return false;
        }
    }
    void add(int64_t v)
    {
        if ((used + 1) * 2 > slot.size()) grow();
        for (uint64_t i = hash(v) & mask;; i = (i + 1) & mask) {
            if (slot[i] == v) return;
            if (slot[i] == EMPTY) { slot[i] = v; used++; return; }
        }
    }
    void grow()
    {
        std::vector<int64_t> old;
        old.swap(slot);
        slot.assign(old.size() * 2, EMPTY);
        mask = slot.size() - 1;
        used = 0;
        for (int64_t v : old) if (v != EMPTY) add(v);
    }
};

struct DocEntry {
    int64_t doc;
    double score = 0.0;
    int64_t best_key = -1;
    double best_score = 0.0;
    int64_t last_key = -1;
    std::vector<int32_t> keys;
    std::vector<double> key_scores;
};

struct DocMap {   // doc id -> index into entries, insertion ordered
    std::vector<int64_t> key;
    std::vector<int32_t> val;
    uint64_t mask;
    std::vector<DocEntry> entries;
    explicit DocMap(uint64_t expect)
    {
        uint64_t cap = 64;
        while (cap < expect * 2 + 16) cap <<= 1;
        key.assign(cap, INT64_MIN); val.assign(cap, -1); mask = cap - 1;
    }
    DocEntry &get(int64_t doc)
    {
        if ((entries.size() + 1) * 2 > key.size()) rehash();
        for (uint64_t i = PosSet::hash(doc) & mask;; i = (i + 1) & mask) {
            if (key[i] == doc) return entries[val[i]];
            if (key[i] == INT64_MIN) {
                key[i] = doc; val[i] = (int32_t)entries.size();
                entries.emplace_back();
                entries.back().doc = doc;
                return entries.back();
            }
        }
    }
    void rehash()
    {
        uint64_t cap = key.size() * 2;
        key.assign(cap, INT64_MIN); val.assign(cap, -1); mask = cap - 1;
        for (size_t e = 0; e < entries.size(); e++)
            for (uint64_t i = PosSet::hash(entries[e].doc) & mask;; i = (i + 1) & mask)
                if (key[i] == INT64_MIN) { key[i] = entries[e].doc; val[i] = (int32_t)e; break; }
    }
};

}  // namespace

struct fmi_evidence {
    std::vector<int64_t> doc;
    std::vector<double> score, best_score;
    std::vector<int64_t> best_key;
    std::vector<int64_t> key_off;
    std::vector<int32_t> key_idx;
    std::vector<double> key_score;
};

extern "C" int fmi_first_stage(uint64_t n_keys, const int64_t *key_tok_off, const int64_t *key_toks, const double *key_score,
                               const int64_t *occ_off, const int64_t *pos, const int64_t *doc, int allow_overlaps,
                               double beta, double single_key, uint64_t n_top, fmi_evidence **out)
{
    if (!out || (n_keys && (!key_tok_off || !key_score || !occ_off))) { fmi_set_error("fmi_first_stage: null argument"); return FMI_ERR_ARG; }
    const uint64_t total = n_keys ? (uint64_t)occ_off[n_keys] : 0;
    uint64_t max_len = 1;
    for (uint64_t k = 0; k < n_keys; k++) max_len = std::max<uint64_t>(max_len, (uint64_t)(key_tok_off[k + 1] - key_tok_off[k]));
    PosSet covered(total * std::min<uint64_t>(max_len, 4));
    DocMap docs(total);
    // keys.py:314-350
    for (uint64_t k = 0; k < n_keys; k++) {
        const int64_t m = key_tok_off[k + 1] - key_tok_off[k];
        const double sco = key_score[k];
        for (int64_t r = occ_off[k]; r < occ_off[k + 1]; r++) {
            const int64_t p = pos[r];
            bool is_new = true;
            for (int64_t j = 1; j <= m && is_new; j++) is_new = !covered.has(p - j);   // [tok_end - len, tok_end)
            DocEntry &e = docs.get(doc[r]);
            if (sco > e.best_score) { e.best_key = (int64_t)k; e.best_score = sco; }
            if (is_new) for (int64_t j = m; j >= 1; j--) covered.add(p - j);
            if ((is_new || allow_overlaps) && e.last_key != (int64_t)k) {
                e.last_key = (int64_t)k;
                e.score += sco;
                e.keys.push_back((int32_t)k);
                e.key_scores.push_back(sco);
            }
        }
    }
    // keys.py:352-364: repetition re-weighting, per document in key order
    std::vector<int64_t> cover, tts;
    for (DocEntry &e : docs.entries) {
        cover.clear();
        double current = 0.0;
        for (size_t i = 0; i < e.keys.size(); i++) {
            const int32_t k = e.keys[i];
            tts.assign(key_toks + key_tok_off[k], key_toks + key_tok_off[k + 1]);
            std::sort(tts.begin(), tts.end());
            tts.erase(std::unique(tts.begin(), tts.end()), tts.end());
            double new_sco = e.key_scores[i];
            if (!cover.empty()) {
                uint64_t diff = 0;
                for (int64_t t : tts) if (!std::binary_search(cover.begin(), cover.end(), t)) diff++;
                const double coeff = (1.0 - beta) + ((beta * (double)diff) / (double)tts.size());
                new_sco = coeff * new_sco;
            }
            current += new_sco;
            e.key_scores[i] = new_sco;
            std::vector<int64_t> merged;
            merged.reserve(cover.size() + tts.size());
            std::set_union(cover.begin(), cover.end(), tts.begin(), tts.end(), std::back_inserter(merged));
            cover.swap(merged);
        }
        e.score = current;
    }
    // keys.py:366-367: stable ranking, cut to n_top
    std::vector<uint32_t> order(docs.entries.size());
    std::iota(order.begin(), order.end(), 0u);
    std::vector<double> rank_key(order.size());
    for (size_t i = 0; i < order.size(); i++)
        rank_key[i] = (1.0 - single_key) * (-docs.entries[i].score) + single_key * (-docs.entries[i].best_score);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rank_key[a] < rank_key[b]; });
    if (order.size() > n_top) order.resize(n_top);
    fmi_evidence *ev = new fmi_evidence();
    ev->key_off.push_back(0);
    for (uint32_t i : order) {
        const DocEntry &e = docs.entries[i];
        ev->doc.push_back(e.doc); ev->score.push_back(e.score);
        ev->best_key.push_back(e.best_key); ev->best_score.push_back(e.best_score);
        ev->key_idx.insert(ev->key_idx.end(), e.keys.begin(), e.keys.end());
        ev->key_score.insert(ev->key_score.end(), e.key_scores.begin(), e.key_scores.end());
        ev->key_off.push_back((int64_t)ev->key_idx.size());
    }
    *out = ev;
    return FMI_OK;
}

extern "C" uint64_t fmi_evidence_docs(const fmi_evidence *ev) { return ev ? ev->doc.size() : 0; }
extern "C" uint64_t fmi_evidence_entries(const fmi_evidence *ev) { return ev ? ev->key_idx.size() : 0; }

extern "C" int fmi_evidence_read(const fmi_evidence *ev, int64_t *doc, double *score, int64_t *best_key, double *best_score,
                                 int64_t *key_off, int32_t *key_idx, double *key_score)
{
    if (!ev) { fmi_set_error("null evidence"); return FMI_ERR_ARG; }
    const size_t n = ev->doc.size(), m = ev->key_idx.size();
    if (n) {
        memcpy(doc, ev->doc.data(), n * 8); memcpy(score, ev->score.data(), n * 8);
        memcpy(best_key, ev->best_key.data(), n * 8); memcpy(best_score, ev->best_score.data(), n * 8);
    }
    memcpy(key_off, ev->key_off.data(), (n + 1) * 8);
    if (m) { memcpy(key_idx, ev->key_idx.data(), m * 4); memcpy(key_score, ev->key_score.data(), m * 8); }
    return FMI_OK;
}

extern "C" void fmi_evidence_free(fmi_evidence *ev) { delete ev; }

// LM-vs-corpus log-odds of seal/keys.py:219-227 / 257-262 for many (score, count) pairs.  Python's
// math.log / math.exp are libm's log / exp on doubles, so this loop is bit-identical to the
// reference's per-key python arithmetic (-ffp-contract=off: no fused multiply-add).
#include <cmath>
extern "C" int fmi_log_odds_batch(uint64_t n, const double *sr, const int64_t *count, double ntokens, double smoothing,
                                  double *out)
{
    for (uint64_t i = 0; i < n; i++) {
        if (count[i] == 0) { out[i] = 0.0; continue; }
        const double snr = std::log(((double)count[i] + smoothing) / (ntokens + smoothing));
        const double a = sr[i] + std::log(1 - std::exp(snr));
        const double b = snr + std::log(1 - std::exp(sr[i]));
        out[i] = a - b;
    }
    return FMI_OK;
}
