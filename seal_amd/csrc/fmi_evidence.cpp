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

// ---------------------------------------------------------------------------
// Full-document scoring (reference seal/keys.py:377-494) for the ranked documents of ONE query.
// Same results as the reference's python (float64, its operation order, its container orders):
//   - keys (score > 0) form a trie; every occurrence of every key in a document is a match;
//   - matches are registered per END position in the order the reference's open-match list yields
//     them: odd lengths ascending, then even lengths descending (keys.py:400-416 pops from the end
//     of a list that is rebuilt on every token);
//   - the best single key is the first-registered one with the strictly largest score;
//   - occurrences are taken greedily in the order of the reference's heap: (-score, key tokens
//     lexicographically, score, start, end); a key's score is discounted by the share of its distinct
//     tokens already covered by accepted keys (repetition()), occurrences must not overlap accepted ones;
//   - free positions contribute their token's unigram score once per distinct token, in order of
//     first free occurrence.
// ---------------------------------------------------------------------------
namespace {

struct Trie {
    struct Node { int32_t key = -1; };
    std::vector<Node> nodes;
    std::vector<int64_t> hk;     // open addressing: (node << 32 | ...) needs 64-bit token: use two arrays
    std::vector<int64_t> htok;
    std::vector<int32_t> hnode, hchild;
    uint64_t mask = 0;
    explicit Trie(uint64_t edges)
    {
        uint64_t cap = 64;
        while (cap < edges * 2 + 16) cap <<= 1;
        htok.assign(cap, 0); hnode.assign(cap, -1); hchild.assign(cap, -1); mask = cap - 1;
        nodes.emplace_back();
    }
    static uint64_t h(int32_t node, int64_t tok) { uint64_t x = ((uint64_t)(uint32_t)node << 40) ^ (uint64_t)tok; x *= 0x9E3779B97F4A7C15ull; return x ^ (x >> 31); }
    int32_t child(int32_t node, int64_t tok) const
    {
        for (uint64_t i = h(node, tok) & mask;; i = (i + 1) & mask) {
            if (hnode[i] < 0) return -1;
            if (hnode[i] == node && htok[i] == tok) return hchild[i];
        }
    }
    int32_t add(int32_t node, int64_t tok)
    {
        for (uint64_t i = h(node, tok) & mask;; i = (i + 1) & mask) {
            if (hnode[i] < 0) {
                hnode[i] = node; htok[i] = tok; hchild[i] = (int32_t)nodes.size();
                nodes.emplace_back();
                return hchild[i];
            }
            if (hnode[i] == node && htok[i] == tok) return hchild[i];
        }
    }
};

struct Match { int32_t e, cls, ord, s, key; };

}  // namespace

struct fmi_fullscore {
    std::vector<int64_t> order;          // input doc index, ranked
    std::vector<double> score, best_score;
    std::vector<int64_t> best_key;
    std::vector<int64_t> pick_off;
    std::vector<int64_t> pick_id;        // key index >= 0, or -(token + 1) for a unigram
    std::vector<double> pick_score;
};

extern "C" int fmi_full_score(uint64_t n_keys, const int64_t *key_tok_off, const int64_t *key_toks, const double *key_score,
                              const double *type_scores, uint64_t vocab, uint64_t n_docs, const int64_t *doc_off,
                              const int64_t *doc_toks, int allow_overlaps, double beta, double single_key,
                              int single_key_add_unigrams, int unigrams_ignore_free_places, fmi_fullscore **out)
{
    if (!out || (n_keys && (!key_tok_off || !key_toks || !key_score)) || (n_docs && (!doc_off || !doc_toks))) {
        fmi_set_error("fmi_full_score: null argument");
        return FMI_ERR_ARG;
    }
    const uint64_t total_key_toks = n_keys ? (uint64_t)key_tok_off[n_keys] : 0;
    Trie trie(total_key_toks);
    for (uint64_t k = 0; k < n_keys; k++) {
        int32_t node = 0;
        for (int64_t t = key_tok_off[k]; t < key_tok_off[k + 1]; t++) node = trie.add(node, key_toks[t]);
        trie.nodes[node].key = (int32_t)k;
    }
    // distinct-token sets of the keys (sorted), for repetition()
    std::vector<std::vector<int64_t>> key_set(n_keys);
    for (uint64_t k = 0; k < n_keys; k++) {
        key_set[k].assign(key_toks + key_tok_off[k], key_toks + key_tok_off[k + 1]);
        std::sort(key_set[k].begin(), key_set[k].end());
        key_set[k].erase(std::unique(key_set[k].begin(), key_set[k].end()), key_set[k].end());
    }
    auto key_less = [&](int32_t a, int32_t b) {      // python tuple comparison of the token sequences
        return std::lexicographical_compare(key_toks + key_tok_off[a], key_toks + key_tok_off[a + 1],
                                            key_toks + key_tok_off[b], key_toks + key_tok_off[b + 1]);
    };
    struct DocOut { double score, best_score; int64_t best_key; std::vector<int64_t> ids; std::vector<double> sc; };
    std::vector<DocOut> res(n_docs);
    std::vector<Match> matches;
    std::vector<int32_t> first_seen;                       // keys in registration order
    std::vector<int32_t> seen_stamp(n_keys, -1);
    std::vector<std::vector<std::pair<int32_t, int32_t>>> spans(n_keys);
    struct Cand { double neg; int32_t key; double s; int32_t i, j; };
    std::vector<Cand> cand;
    std::vector<int64_t> cover;
    std::vector<uint8_t> is_free;
    for (uint64_t d = 0; d < n_docs; d++) {
        const int64_t *tok = doc_toks + doc_off[d];
        const int32_t T = (int32_t)(doc_off[d + 1] - doc_off[d]);
        matches.clear();
        for (int32_t s = 0; s < T; s++) {
            int32_t node = 0;
            for (int32_t e = s; e < T; e++) {
                node = trie.child(node, tok[e]);
                if (node < 0) break;
                const int32_t k = trie.nodes[node].key;
                if (k >= 0) {
                    const int32_t len = e - s + 1;
                    matches.push_back(Match{e, (len & 1) ? 0 : 1, (len & 1) ? len : -len, s, k});
                }
            }
        }
        std::sort(matches.begin(), matches.end(), [](const Match &a, const Match &b) {
            if (a.e != b.e) return a.e < b.e;
            if (a.cls != b.cls) return a.cls < b.cls;
            return a.ord < b.ord;
        });
        first_seen.clear();
        for (const Match &m : matches) {
            if (seen_stamp[m.key] != (int32_t)d) { seen_stamp[m.key] = (int32_t)d; spans[m.key].clear(); first_seen.push_back(m.key); }
            spans[m.key].push_back({m.s, m.e + 1});
        }
        DocOut &o = res[d];
        o.best_key = -1; o.best_score = 0.0;
        cand.clear();
        for (int32_t k : first_seen) {
            const double s = key_score[k];
            for (auto &sp : spans[k]) cand.push_back(Cand{-s, k, s, sp.first, sp.second});
            if (-s < -o.best_score) { o.best_key = k; o.best_score = s; }
        }
        std::sort(cand.begin(), cand.end(), [&](const Cand &a, const Cand &b) {
            if (a.neg != b.neg) return a.neg < b.neg;
            if (a.key != b.key) { if (key_less(a.key, b.key)) return true; if (key_less(b.key, a.key)) return false; }
            if (a.s != b.s) return a.s < b.s;
            if (a.i != b.i) return a.i < b.i;
            return a.j < b.j;
        });
        cover.clear();
        is_free.assign((size_t)T, 1);
        int32_t prev = -1;
        auto repetition = [&](const int64_t *set_begin, size_t set_size, double score) {
            if (cover.empty()) return score;
            uint64_t diff = 0;
            for (size_t x = 0; x < set_size; x++) if (!std::binary_search(cover.begin(), cover.end(), set_begin[x])) diff++;
            const double coeff = (1.0 - beta) + ((beta * (double)diff) / (double)set_size);
            return coeff * score;
        };
        for (const Cand &c : cand) {
            const std::vector<int64_t> &ns = key_set[c.key];
            double new_s;
            if (prev == c.key) new_s = o.sc.back();
            else if (ns.empty()) new_s = 0.0;
            else new_s = repetition(ns.data(), ns.size(), c.s);
            if (new_s <= 0.0) continue;
            if (!allow_overlaps) {
                bool all_free = true;
                for (int32_t p = c.i; p < c.j && all_free; p++) all_free = is_free[p];
                if (!all_free) continue;
            }
            if (prev == c.key) {
                o.sc.back() = new_s;
            } else {
                prev = c.key;
                std::vector<int64_t> merged;
                merged.reserve(cover.size() + ns.size());
                std::set_union(cover.begin(), cover.end(), ns.begin(), ns.end(), std::back_inserter(merged));
                cover.swap(merged);
                o.ids.push_back(c.key);
                o.sc.push_back(new_s);
            }
            for (int32_t p = c.i; p < c.j; p++) is_free[p] = 0;
        }
        if (unigrams_ignore_free_places) is_free.assign((size_t)T, 1);
        double single = o.best_score;
        double multi = 0.0;
        for (double v : o.sc) multi += v;
        double uni = 0.0;
        // distinct free tokens in order of first free occurrence (Counter over the free positions)
        std::vector<int64_t> seen_tok;
        for (int32_t p = 0; p < T; p++) {
            if (!is_free[p]) continue;
            const int64_t t = tok[p];
            if (std::find(seen_tok.begin(), seen_tok.end(), t) != seen_tok.end()) continue;
            seen_tok.push_back(t);
            double s = (type_scores && t >= 0 && (uint64_t)t < vocab) ? type_scores[t] : 0.0;
            if (s > 0.0) {
                s = repetition(&t, 1, s);
                if (s != 0.0) { uni += s; o.ids.push_back(-(t + 1)); o.sc.push_back(s); }
            }
        }
        if (single_key_add_unigrams) single += uni;
        multi += uni;
        o.score = (1.0 - single_key) * multi + single_key * single;
    }
    fmi_fullscore *fs = new fmi_fullscore();
    std::vector<uint32_t> order(n_docs);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return -res[a].score < -res[b].score; });
    fs->pick_off.push_back(0);
    for (uint32_t i : order) {
        fs->order.push_back(i); fs->score.push_back(res[i].score);
        fs->best_key.push_back(res[i].best_key); fs->best_score.push_back(res[i].best_score);
        fs->pick_id.insert(fs->pick_id.end(), res[i].ids.begin(), res[i].ids.end());
        fs->pick_score.insert(fs->pick_score.end(), res[i].sc.begin(), res[i].sc.end());
        fs->pick_off.push_back((int64_t)fs->pick_id.size());
    }
    *out = fs;
    return FMI_OK;
}

extern "C" uint64_t fmi_fullscore_docs(const fmi_fullscore *fs) { return fs ? fs->order.size() : 0; }
extern "C" uint64_t fmi_fullscore_entries(const fmi_fullscore *fs) { return fs ? fs->pick_id.size() : 0; }
extern "C" int fmi_fullscore_read(const fmi_fullscore *fs, int64_t *order, double *score, int64_t *best_key, double *best_score,
                                  int64_t *pick_off, int64_t *pick_id, double *pick_score)
{
    if (!fs) { fmi_set_error("null result"); return FMI_ERR_ARG; }
    const size_t n = fs->order.size(), m = fs->pick_id.size();
    if (n) {
        memcpy(order, fs->order.data(), n * 8); memcpy(score, fs->score.data(), n * 8);
        memcpy(best_key, fs->best_key.data(), n * 8); memcpy(best_score, fs->best_score.data(), n * 8);
    }
    memcpy(pick_off, fs->pick_off.data(), (n + 1) * 8);
    if (m) { memcpy(pick_id, fs->pick_id.data(), m * 8); memcpy(pick_score, fs->pick_score.data(), m * 8); }
    return FMI_OK;
}
extern "C" void fmi_fullscore_free(fmi_fullscore *fs) { delete fs; }
