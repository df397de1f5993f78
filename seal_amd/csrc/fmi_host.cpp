// Host side of libsealfm.so: construction, (de)serialisation, upload.
// No query arithmetic lives here -- queries are HIP kernels (fmi_kernels.hip).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "fmi_internal.h"

static thread_local std::string g_err;

void fmi_set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char *fmi_last_error(void) { return g_err.c_str(); }

uint32_t fmi_sb_shift_for(uint64_t n)
{
    const char *e = getenv("SEALFM_FORCE_SB");
    if (e && *e) return (uint32_t)std::max(0, std::min(30, atoi(e)));
    return n >= (1ull << 32) ? FMI_SB_SHIFT : FMI_SB_NONE;
}
extern "C" uint32_t fmi_abi_version(void) { return 1; }

// ---- launch-shape options (fmi_internal.h): built-in choices; fmi_dev_set_option changes one on a handle (tests, tools).  Round 6: the nine
// SEALFM_* environment variables that used to seed them are gone -- every alternative they selected is either a test of a kernel path
// (set through fmi_dev_set_option) or lost its A/B (profiles/ keeps the records) ----
FmiOptions::FmiOptions() {}

int FmiOptions::set(const char *name, int64_t value)
{
    const std::string s(name ? name : "");
    if (s == "constrain_waves") constrain_waves = value;
    else if (s == "leave_early") leave_early = value < 0 ? 1 : value;
    else if (s == "row_first") row_first = value;
    else if (s == "row_first_from") row_first_from = value;
    else if (s == "prefix_tables") prefix_tables = value < 0 ? 1 : value;
    else if (s == "table_grid") table_grid = value;
    else if (s == "topk_narrow") topk_narrow = value;
    else if (s == "topk_legacy") topk_legacy = value < 0 ? 0 : value;
    else if (s == "chain_steps") chain_steps = value < 0 ? 1 : value;
    else if (s == "advance_apart") advance_apart = value < 0 ? 1 : value;
    else if (s == "agg_rank_by_sorts") agg_rank_by_sorts = value < 0 ? 0 : value;
    else if (s == "pt_inject_failure") pt_inject_failure = value < 0 ? 0 : value;
    else return -1;
    return 0;
}

extern "C" int fmi_dev_set_option(fmi_t *h, const char *name, int64_t value)
{
    if (!h || !name) { fmi_set_error("null argument"); return FMI_ERR_ARG; }
    if (h->opt.set(name, value)) { fmi_set_error("fmi_dev_set_option: unknown option '%s'", name); return FMI_ERR_ARG; }
    return FMI_OK;
}

extern "C" int fmi_dev_prefix_table_stats(fmi_t *h, uint64_t *n_tables, uint64_t *n_nodes, uint64_t *n_bytes)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    uint64_t t = 0, n = 0, b = 0;
    for (const FmiPrefixTable &T : h->prefix_tables) {
        if (!T.ok) continue;
        t++; n += T.n_nodes;
        b += T.n_nodes * 16 + (T.vocab + 1) * 8 + T.vocab * 16;
    }
    if (n_tables) *n_tables = t;
    if (n_nodes) *n_nodes = n;
    if (n_bytes) *n_bytes = b;
    return FMI_OK;
}

extern "C" int fmi_create(fmi_t **out)
{
    if (!out) { fmi_set_error("fmi_create: null out"); return FMI_ERR_ARG; }
    *out = new fmi();
    return FMI_OK;
}

void fmi_release_device(fmi *h)
{
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        for (void *p : h->dev_allocs) (void)hipFree(p);
        if (h->ws) (void)hipFree(h->ws);
        if (h->ws_list) (void)hipFree(h->ws_list);
        for (FmiPrefixTable &t : h->prefix_tables) {
            if (t.d_off) (void)hipFree(t.d_off);
            if (t.d_root) (void)hipFree(t.d_root);
            if (t.d_nodes) (void)hipFree(t.d_nodes);
        }
        if (h->sym_bits) (void)hipFree(h->sym_bits);
        if (h->d_probe_counter) (void)hipFree(h->d_probe_counter);
        if (h->service_stream) (void)hipStreamDestroy((hipStream_t)h->service_stream);
        for (void *e : h->ev_start) (void)hipEventDestroy((hipEvent_t)e);
        for (void *e : h->ev_stop) (void)hipEventDestroy((hipEvent_t)e);
        for (void *e : h->agg_events) (void)hipEventDestroy((hipEvent_t)e);
    }
    h->agg_events.clear(); h->agg_calls.clear(); h->call_log.clear();
    h->service_stream = nullptr;
    h->ev_start.clear(); h->ev_stop.clear(); h->ev_used = 0; h->timing_enabled = 0;
    h->dev_allocs.clear();
    h->prefix_tables.clear();
    h->ws = nullptr; h->ws_bytes = 0; h->ws_rows = 0; h->ws_list = nullptr;
    h->sym_bits = nullptr; h->sym_rows = 0; h->sym_row_words = 0;
    h->d_probe_counter = nullptr;
    h->dev = FmiDev{};
    h->device = -1;
    h->dev_bytes = 0;
}

// A second handle on the SAME resident index: shares every device array of `src` (which must outlive it) and owns
// only what a decode / retrieval pipeline mutates -- workspace, incremental constraint state, counters, service
// stream.  One view per concurrent pipeline (seal_amd/retrieval.py runs two query batches at a time on one GPU).
extern "C" int fmi_view_create(const fmi_t *src, fmi_t **out)
{
    if (!src || !out) { fmi_set_error("fmi_view_create: null argument"); return FMI_ERR_ARG; }
    if (src->device < 0) { fmi_set_error("fmi_view_create: the index is not resident on a GPU"); return FMI_ERR_NO_DEVICE; }
    fmi *v = new fmi();
    v->n = src->n; v->max_sym = src->max_sym; v->sigma = src->sigma; v->nblk = src->nblk;
    v->levels = src->levels; v->dlevels = src->dlevels; v->sym_bytes = src->sym_bytes; v->sb_shift = src->sb_shift; v->nsb = src->nsb;
    v->max_doc_len = src->max_doc_len;
    v->device = src->device;
    v->dev = src->dev;              // pointers into src's allocations; dev_allocs stays empty: nothing to free here
    v->dev_bytes = 0;
    *out = v;
    return FMI_OK;
}

extern "C" void fmi_free(fmi_t *h)
{
    if (!h) return;
    fmi_release_device(h);
    delete h;
}

extern "C" uint64_t fmi_size(const fmi_t *h) { return h ? h->n : 0; }
extern "C" uint64_t fmi_sigma(const fmi_t *h) { return h ? h->sigma : 0; }
extern "C" uint64_t fmi_max_symbol(const fmi_t *h) { return h ? h->max_sym : 0; }
extern "C" uint32_t fmi_levels(const fmi_t *h) { return h ? h->levels : 0; }
extern "C" int fmi_device(const fmi_t *h) { return h ? h->device : -1; }
extern "C" uint64_t fmi_device_bytes(const fmi_t *h) { return h ? h->dev_bytes : 0; }

// ---------------------------------------------------------------------------
// Suffix array on the host: prefix doubling.  Round 0 sorts by a 64-bit key
// packing the first 64/bits symbols; later rounds refine only the groups that
// are still tied, by the rank of the suffix h symbols further on.
// The text ends in a unique smallest sentinel (0), so all suffixes are distinct.
// ---------------------------------------------------------------------------
void fmi_host_suffix_array(const uint32_t *text, uint64_t n, uint32_t bits, std::vector<uint64_t> &sa)
{
    sa.resize(n);
    if (n == 0) return;
    if (bits == 0) bits = 1;
    const uint32_t per = std::max<uint32_t>(1, 64 / bits);
    std::vector<std::pair<uint64_t, uint64_t>> keyed(n);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t key = 0;
        for (uint32_t j = 0; j < per; j++) {
            uint64_t s = (i + j < n) ? text[i + j] : 0;
            key = (per * bits >= 64 && j == 0) ? s : ((key << bits) | s);
        }
        keyed[i] = {key, i};
    }
    std::sort(keyed.begin(), keyed.end());
    // rank[i] = index of the first row of i's group
    std::vector<uint64_t> rank(n);
    bool tied = false;
    {
        uint64_t g = 0;
        for (uint64_t j = 0; j < n; j++) {
            if (j > 0 && keyed[j].first != keyed[j - 1].first) g = j;
            else if (j > 0) tied = true;
            sa[j] = keyed[j].second;
            rank[sa[j]] = g;
        }
    }
    keyed.clear(); keyed.shrink_to_fit();
    std::vector<std::pair<uint64_t, uint64_t>> tmp;
    for (uint64_t h = per; tied; h *= 2) {
        tied = false;
        uint64_t j = 0;
        while (j < n) {
            uint64_t g = rank[sa[j]], e = j + 1;
            while (e < n && rank[sa[e]] == g) e++;
            if (e - j > 1) {
                // a tied suffix never contains the sentinel within its first h
                // symbols, so sa[x] + h < n holds for every member
                tmp.resize(e - j);
                for (uint64_t x = j; x < e; x++) tmp[x - j] = {rank[sa[x] + h], sa[x]};
                std::sort(tmp.begin(), tmp.end());
                // ranks of this group are rewritten only after all second keys
                // were read (members may reference each other)
                uint64_t gs = j;
                for (uint64_t x = 0; x < tmp.size(); x++) {
                    sa[j + x] = tmp[x].second;
                }
                std::vector<uint64_t> newrank(tmp.size());
                for (uint64_t x = 0; x < tmp.size(); x++) {
                    if (x > 0 && tmp[x].first != tmp[x - 1].first) gs = j + x;
                    else if (x > 0) tied = true;
                    newrank[x] = gs;
                }
                for (uint64_t x = 0; x < tmp.size(); x++) rank[tmp[x].second] = newrank[x];
            }
            j = e;
        }
    }
}

// ---------------------------------------------------------------------------
// Quirk Q1: the reference starts every search from the
// inclusive interval [0, size()] (seal/index.py:106-107, fm_index.cpp:58-59),
// one row past the end, so sdsl evaluates wt_int::rank(size()+1, c).  In the
// level-concatenated pointerless tree that call reads, at every level, the bit
// that FOLLOWS c's node; the result is occ(c)+1 iff that stray bit equals c's
// own bit on every level, otherwise occ(c).  The stray bit after a node is the
// first bit of the next non-empty node on the same level (nodes are laid out in
// prefix order, empty nodes take no space) or, after the last node of a level,
// the first bit of the next level (0 beyond the last level).  A node's first
// bit belongs to the element of that node that comes first in BWT order.
// All of that is a function of (first occurrence in the BWT of every symbol);
// this routine evaluates it without ever materialising sdsl's layout.
// ---------------------------------------------------------------------------
void fmi_host_q1_table(const uint32_t *bwt, uint64_t n, uint32_t L, uint64_t max_sym,
                       const std::vector<uint64_t> &C, std::vector<uint8_t> &q1)
{
    std::vector<uint64_t> first_pos(max_sym + 1, UINT64_MAX);
    for (uint64_t i = 0; i < n; i++)
        if (first_pos[bwt[i]] == UINT64_MAX) first_pos[bwt[i]] = i;
    fmi_host_q1_from_first_pos(first_pos, L, max_sym, C, q1);
}

void fmi_host_q1_from_first_pos(const std::vector<uint64_t> &first_pos, uint32_t L, uint64_t max_sym,
                                const std::vector<uint64_t> &C, std::vector<uint8_t> &q1)
{
    q1.assign(max_sym + 1, 0);
    std::vector<uint32_t> present;
    for (uint64_t c = 0; c <= max_sym; c++)
        if (C[c + 1] > C[c]) present.push_back((uint32_t)c);
    const size_t P = present.size();
    // per level: for each present symbol, the index of its node among the
    // non-empty nodes of the level, and per node the bit of its first element
    std::vector<std::vector<uint8_t>> node_first_bit(L);
    std::vector<std::vector<uint32_t>> node_of(L, std::vector<uint32_t>(P));
    for (uint32_t k = 0; k < L; k++) {
        const uint32_t sh = L - k;   // node prefix = symbol >> sh (k bits)
        size_t i = 0;
        while (i < P) {
            uint64_t prefix = (sh >= 32) ? 0 : (present[i] >> sh);
            size_t e = i;
            uint64_t best = UINT64_MAX; uint32_t best_sym = 0;
            while (e < P && ((sh >= 32) ? 0 : (present[e] >> sh)) == prefix) {
                if (first_pos[present[e]] < best) { best = first_pos[present[e]]; best_sym = present[e]; }
                e++;
            }
            uint32_t node = (uint32_t)node_first_bit[k].size();
            node_first_bit[k].push_back((uint8_t)((best_sym >> (L - 1 - k)) & 1));
            for (size_t x = i; x < e; x++) node_of[k][x] = node;
            i = e;
        }
    }
    for (size_t x = 0; x < P; x++) {
        bool all_equal = true;
        for (uint32_t k = 0; k < L && all_equal; k++) {
            uint32_t node = node_of[k][x];
            uint8_t stray;
            if (node + 1 < node_first_bit[k].size()) stray = node_first_bit[k][node + 1];
            else if (k + 1 < L) stray = node_first_bit[k + 1][0];
            else stray = 0;
            uint8_t cbit = (uint8_t)((present[x] >> (L - 1 - k)) & 1);
            all_equal = (stray == cbit);
        }
        q1[present[x]] = all_equal ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// BWT -> hex wavelet matrix in the 128-byte block layout (fmi_internal.h) + per-symbol tables.
// ---------------------------------------------------------------------------
void fmi_host_finish_from_bwt(fmi *h, const uint32_t *bwt, uint64_t n)
{
    uint64_t max_sym = 0;
    for (uint64_t i = 0; i < n; i++) max_sym = std::max<uint64_t>(max_sym, bwt[i]);
    uint32_t L = 0;
    while ((max_sym >> L) > 0) L++;   // sdsl: bits::hi(max)+1
    if (L == 0) L = 1;
    h->n = n; h->max_sym = max_sym; h->levels = L;
    h->sym_bytes = (max_sym < 65536) ? 2 : 4;
    // C[c] = #symbols < c
    h->C.assign(max_sym + 2, 0);
    for (uint64_t i = 0; i < n; i++) h->C[bwt[i] + 1]++;
    uint64_t sigma = 0;
    for (uint64_t c = 0; c <= max_sym; c++) { if (h->C[c + 1]) sigma++; h->C[c + 1] += h->C[c]; }
    h->sigma = sigma;
    // hex levels
    const uint32_t D = (L + FMI_DIGIT_BITS - 1) / FMI_DIGIT_BITS;
    h->dlevels = D;
    h->nblk = n / FMI_BLOCK_BITS + 2;
    h->sb_shift = fmi_sb_shift_for(n);
    h->nsb = (h->nblk >> h->sb_shift) + 1;
    h->wm.assign((uint64_t)D * h->nblk * FMI_BLOCK_WORDS, 0);
    h->dbase.assign((size_t)D * FMI_ARITY, 0);
    h->sbase.assign((size_t)D * h->nsb * FMI_ARITY, 0);
    const uint64_t sb_mask = (1ull << h->sb_shift) - 1;
    std::vector<uint32_t> cur(bwt, bwt + n), nxt(n);
    for (uint32_t k = 0; k < D; k++) {
        const uint32_t sh = FMI_DIGIT_BITS * (D - 1 - k);
        uint64_t *lvl = h->wm.data() + (uint64_t)k * h->nblk * FMI_BLOCK_WORDS;
        uint64_t *sb = h->sbase.data() + (size_t)k * h->nsb * FMI_ARITY;
        uint64_t cnt[FMI_ARITY] = {0}, at_sb[FMI_ARITY] = {0};   // digits seen so far / up to the current superblock
        for (uint64_t b = 0; b < h->nblk; b++) {
            uint32_t *blk = reinterpret_cast<uint32_t *>(lvl + b * FMI_BLOCK_WORDS);
            if ((b & sb_mask) == 0) {
                for (uint32_t d = 0; d < FMI_ARITY; d++) { at_sb[d] = cnt[d]; sb[(b >> h->sb_shift) * FMI_ARITY + d] = cnt[d]; }
            }
            for (uint32_t d = 0; d < FMI_ARITY; d++) blk[d] = (uint32_t)(cnt[d] - at_sb[d]);
            const uint64_t p0 = b * FMI_BLOCK_BITS;
            for (uint32_t bit = 0; bit < FMI_BLOCK_BITS && p0 + bit < n; bit++) {
                const uint32_t d = (cur[p0 + bit] >> sh) & (FMI_ARITY - 1);
                for (uint32_t j = 0; j < 4; j++) blk[16 + 4 * j + (bit >> 5)] |= ((d >> j) & 1u) << (bit & 31);
                cnt[d]++;
            }
        }
        uint64_t *db = h->dbase.data() + (size_t)k * FMI_ARITY;
        db[0] = 0;
        for (uint32_t d = 1; d < FMI_ARITY; d++) db[d] = db[d - 1] + cnt[d - 1];
        for (uint64_t s2 = 0; s2 < h->nsb; s2++)
            for (uint32_t d = 0; d < FMI_ARITY; d++) sb[s2 * FMI_ARITY + d] += db[d];
        // stable 16-way partition by digit
        uint64_t o[FMI_ARITY];
        for (uint32_t d = 0; d < FMI_ARITY; d++) o[d] = db[d];
        for (uint64_t i = 0; i < n; i++) nxt[o[(cur[i] >> sh) & (FMI_ARITY - 1)]++] = cur[i];
        cur.swap(nxt);
    }
    // after the last partition equal symbols are contiguous
    h->leaf.assign(max_sym + 1, 0);
    for (uint64_t i = 0; i < n; i++)
        if (i == 0 || cur[i] != cur[i - 1]) h->leaf[cur[i]] = i;
    fmi_host_q1_table(bwt, n, L, max_sym, h->C, h->q1);
}

static void pack_sa(fmi *h, const std::vector<uint64_t> &sa)
{
    const uint64_t n = sa.size();
    h->sa_lo.resize(n);
    bool wide = n > (1ull << 32);
    h->sa_hi.clear();
    if (wide) h->sa_hi.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        h->sa_lo[i] = (uint32_t)sa[i];
        if (wide) h->sa_hi[i] = (uint8_t)(sa[i] >> 32);
    }
}

int fmi_host_build_from_symbols(fmi *h, const uint32_t *text, uint64_t n)
{
    uint64_t max_sym = 0;
    for (uint64_t i = 0; i < n; i++) max_sym = std::max<uint64_t>(max_sym, text[i]);
    uint32_t L = 0;
    while ((max_sym >> L) > 0) L++;
    if (L == 0) L = 1;
    if (L > FMI_MAX_LEVELS) {
        fmi_set_error("alphabet needs %u bits per symbol; this build supports <= %u", L, FMI_MAX_LEVELS);
        return FMI_ERR_UNSUPPORTED;
    }
    std::vector<uint64_t> sa;
    fmi_host_suffix_array(text, n, L, sa);
    h->bwt.resize(n);
    for (uint64_t i = 0; i < n; i++) h->bwt[i] = text[sa[i] ? sa[i] - 1 : n - 1];
    fmi_host_finish_from_bwt(h, h->bwt.data(), n);
    pack_sa(h, sa);
    h->text.resize(n * h->sym_bytes);
    if (h->sym_bytes == 2) {
        uint16_t *t = reinterpret_cast<uint16_t *>(h->text.data());
        for (uint64_t i = 0; i < n; i++) t[i] = (uint16_t)text[i];
    } else {
        memcpy(h->text.data(), text, n * 4);
    }
    h->host_resident = true;
    return FMI_OK;
}

extern "C" int fmi_build(fmi_t *h, const uint64_t *data, uint64_t n_data, int device)
{
    if (!h || (!data && n_data)) { fmi_set_error("fmi_build: null argument"); return FMI_ERR_ARG; }
    fmi_release_device(h);
    std::vector<uint32_t> text(n_data + 1);
    for (uint64_t i = 0; i < n_data; i++) {
        if (data[i] == 0 || data[i] >= (1ull << FMI_MAX_LEVELS)) {
            fmi_set_error("fmi_build: symbol %llu at %llu outside [1, 2^%u)", (unsigned long long)data[i],
                          (unsigned long long)i, FMI_MAX_LEVELS);
            return FMI_ERR_ARG;
        }
        text[i] = (uint32_t)data[i];
    }
    text[n_data] = 0;   // sdsl's construct appends the sentinel
    int rc = fmi_host_build_from_symbols(h, text.data(), n_data + 1);
    if (rc != FMI_OK) return rc;
    if (device >= 0) return fmi_upload(h, device);
    return FMI_OK;
}

extern "C" int fmi_build_from_file(fmi_t *h, const char *path, int width, int device)
{
    if (!h || !path || !(width == 1 || width == 2 || width == 4 || width == 8)) {
        fmi_set_error("fmi_build_from_file: bad argument");
        return FMI_ERR_ARG;
    }
    FILE *f = fopen(path, "rb");
    if (!f) { fmi_set_error("cannot open %s", path); return FMI_ERR_IO; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint64_t cnt = (uint64_t)sz / (uint64_t)width;
    std::vector<uint8_t> raw((size_t)sz);
    if (sz && fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); fmi_set_error("short read %s", path); return FMI_ERR_IO; }
    fclose(f);
    std::vector<uint64_t> data(cnt);
    for (uint64_t i = 0; i < cnt; i++) {
        uint64_t v = 0;
        memcpy(&v, raw.data() + i * width, width);   // little endian host
        data[i] = v;
    }
    return fmi_build(h, data.data(), cnt, device);
}

// device copies of the document boundaries and of the sampled position -> document table (FmiDev::doc_hint)
static int upload_doc_tables(fmi *h)
{
    const std::vector<uint64_t> &b = h->doc_begin;
    const uint64_t n_entries = b.size();
    void *p = nullptr;
    if (hipMalloc(&p, n_entries * 8) != hipSuccess) { fmi_set_error("hipMalloc(doc_begin) failed"); return FMI_ERR_HIP; }
    if (hipMemcpy(p, b.data(), n_entries * 8, hipMemcpyHostToDevice) != hipSuccess) { fmi_set_error("hipMemcpy(doc_begin) failed"); return FMI_ERR_HIP; }
    h->dev_allocs.push_back(p);
    h->dev_bytes += n_entries * 8;
    h->dev.doc_begin = (const uint64_t *)p;
    h->dev.n_begin = n_entries;
    h->dev.doc_hint = nullptr;
    // doc_hint[blk] = bisect_right(b, blk << shift) - 1 for every block that holds a text position (+ one past): one merge
    // pass.  Boundaries must ascend from 0 (index.py:25,50 builds them as a running sum); anything else keeps the full bisect.
    bool ok = n_entries >= 1 && b[0] == 0 && n_entries < 0xffffffffull && h->n > 0;
    for (uint64_t i = 1; ok && i < n_entries; i++) ok = b[i] >= b[i - 1];
    if (ok) {
        const uint64_t nh = ((h->n - 1) >> FMI_DOC_HINT_SHIFT) + 2;
        std::vector<uint32_t> hint(nh);
        uint64_t d = 0;                       // last index with b[d] <= position
        for (uint64_t blk = 0; blk < nh; blk++) {
            const uint64_t pos = blk << FMI_DOC_HINT_SHIFT;
            while (d + 1 < n_entries && b[d + 1] <= pos) d++;
            hint[blk] = (uint32_t)d;
        }
        void *q = nullptr;
        if (hipMalloc(&q, nh * 4) != hipSuccess) { fmi_set_error("hipMalloc(doc_hint) failed"); return FMI_ERR_HIP; }
        if (hipMemcpy(q, hint.data(), nh * 4, hipMemcpyHostToDevice) != hipSuccess) { fmi_set_error("hipMemcpy(doc_hint) failed"); return FMI_ERR_HIP; }
        h->dev_allocs.push_back(q);
        h->dev_bytes += nh * 4;
        h->dev.doc_hint = (const uint32_t *)q;
    }
    return FMI_OK;
}

extern "C" int fmi_set_doc_beginnings(fmi_t *h, const uint64_t *b, uint64_t n_entries)
{
    if (!h || !b || n_entries == 0) { fmi_set_error("fmi_set_doc_beginnings: bad argument"); return FMI_ERR_ARG; }
    h->doc_begin.assign(b, b + n_entries);
    h->max_doc_len = 0;
    for (uint64_t i = 1; i < n_entries; i++) h->max_doc_len = std::max<uint64_t>(h->max_doc_len, b[i] - b[i - 1]);
    if (h->device >= 0) {
        if (hipSetDevice(h->device) != hipSuccess) { fmi_set_error("hipSetDevice failed"); return FMI_ERR_HIP; }
        return upload_doc_tables(h);
    }
    return FMI_OK;
}

// ---------------------------------------------------------------------------
// upload
// ---------------------------------------------------------------------------
template <class T>
static int up(fmi *h, const std::vector<T> &v, const T **out)
{
    *out = nullptr;
    if (v.empty()) return FMI_OK;
    void *p = nullptr;
    size_t bytes = v.size() * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) { fmi_set_error("hipMalloc(%zu) failed", bytes); return FMI_ERR_HIP; }
    if (hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { fmi_set_error("hipMemcpy H2D failed"); return FMI_ERR_HIP; }
    h->dev_allocs.push_back(p);
    h->dev_bytes += bytes;
    *out = (const T *)p;
    return FMI_OK;
}

int fmi_upload(fmi *h, int device)
{
    if (!h->host_resident) { fmi_set_error("fmi_upload: index is not host resident"); return FMI_ERR_STATE; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= device) {
        fmi_set_error("no HIP device %d visible (this library has no CPU query path)", device);
        return FMI_ERR_NO_DEVICE;
    }
    fmi_release_device(h);
    if (hipSetDevice(device) != hipSuccess) { fmi_set_error("hipSetDevice(%d) failed", device); return FMI_ERR_HIP; }
    h->device = device;
    FmiDev d{};
    d.nblk = h->nblk; d.n = h->n; d.max_sym = h->max_sym; d.levels = h->levels; d.dlevels = h->dlevels; d.sym_bytes = h->sym_bytes;
    d.nsb = h->nsb; d.sb_shift = h->sb_shift;
    for (uint32_t k = 0; k < h->dlevels; k++)
        for (uint32_t e = 0; e < FMI_ARITY; e++) d.dbase[k][e] = h->dbase[(size_t)k * FMI_ARITY + e];
    int rc;
    const uint8_t *text8 = nullptr;
    if ((rc = up(h, h->wm, &d.wm)) || (rc = up(h, h->sbase, &d.sbase)) || (rc = up(h, h->C, &d.C)) || (rc = up(h, h->leaf, &d.leaf)) ||
        (rc = up(h, h->q1, &d.q1)) || (rc = up(h, h->sa_lo, &d.sa_lo)) || (rc = up(h, h->sa_hi, &d.sa_hi)) ||
        (rc = up(h, h->text, &text8))) {
        fmi_release_device(h);
        return rc;
    }
    d.text = text8;
    d.doc_begin = nullptr; d.n_begin = 0; d.doc_hint = nullptr;
    h->dev = d;
    if (!h->doc_begin.empty() && (rc = upload_doc_tables(h))) { fmi_release_device(h); return rc; }
    return FMI_OK;
}

extern "C" int fmi_to_device(fmi_t *h, int device)
{
    if (!h) { fmi_set_error("null handle"); return FMI_ERR_ARG; }
    return fmi_upload(h, device);
}

// ---------------------------------------------------------------------------
// on-disk format ".fmi" (little endian):
//   char[8] "SEALFMI4"; u64 n, max_sym, sigma, nblk; u32 levels, sym_bytes; u32 sb_shift, 0;
//   u64 sa_wide(0/1); then arrays, each as u64 byte length + raw bytes, in the
//   order dbase, sbase, C, leaf, q1, wm, sa_lo, sa_hi, text, bwt.  (SEALFMI1..3 were the binary and 4-ary layouts of earlier
//   development snapshots.)
// ---------------------------------------------------------------------------
template <class T>
static bool wr(FILE *f, const std::vector<T> &v)
{
    uint64_t bytes = v.size() * sizeof(T);
    if (fwrite(&bytes, 8, 1, f) != 1) return false;
    return bytes == 0 || fwrite(v.data(), 1, bytes, f) == bytes;
}
template <class T>
static bool rd(FILE *f, std::vector<T> &v)
{
    uint64_t bytes = 0;
    if (fread(&bytes, 8, 1, f) != 1) return false;
    v.resize(bytes / sizeof(T));
    return bytes == 0 || fread(v.data(), 1, bytes, f) == bytes;
}

extern "C" int fmi_save(const fmi_t *h, const char *path)
{
    if (!h || !path) { fmi_set_error("fmi_save: null argument"); return FMI_ERR_ARG; }
    if (!h->host_resident) { fmi_set_error("fmi_save: index has no host copy (built on device without keep_host)"); return FMI_ERR_STATE; }
    FILE *f = fopen(path, "wb");
    if (!f) { fmi_set_error("cannot open %s for writing", path); return FMI_ERR_IO; }
    bool ok = fwrite("SEALFMI4", 1, 8, f) == 8;
    uint64_t hdr[4] = {h->n, h->max_sym, h->sigma, h->nblk};
    uint32_t hdr2[4] = {h->levels, h->sym_bytes, h->sb_shift, 0};
    ok = ok && fwrite(hdr, 8, 4, f) == 4 && fwrite(hdr2, 4, 4, f) == 4;
    ok = ok && wr(f, h->dbase) && wr(f, h->sbase) && wr(f, h->C) && wr(f, h->leaf) && wr(f, h->q1) && wr(f, h->wm) &&
         wr(f, h->sa_lo) && wr(f, h->sa_hi) && wr(f, h->text) && wr(f, h->bwt);
    fclose(f);
    if (!ok) { fmi_set_error("write error on %s", path); return FMI_ERR_IO; }
    return FMI_OK;
}

extern "C" int fmi_load_sdsl(fmi_t **out, const char *path, int device);

extern "C" int fmi_load(fmi_t **out, const char *path, int device)
{
    if (!out || !path) { fmi_set_error("fmi_load: null argument"); return FMI_ERR_ARG; }
    FILE *f = fopen(path, "rb");
    if (!f) { fmi_set_error("cannot open %s", path); return FMI_ERR_IO; }
    char magic[8];
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "SEALFMI4", 8) == 0;
    if (!ok) {
        // not this engine's container: the reference's own files are sdsl-lite serialisations of csa_wt_int<>
        // (fm_index.cpp:186-199); fmi_sdsl.cpp reads those (and refuses anything it cannot vouch for)
        fclose(f);
        return fmi_load_sdsl(out, path, device);
    }
    fmi *h = new fmi();
    uint64_t hdr[4]; uint32_t hdr2[4];
    ok = ok && fread(hdr, 8, 4, f) == 4 && fread(hdr2, 4, 4, f) == 4;
    if (ok) { h->n = hdr[0]; h->max_sym = hdr[1]; h->sigma = hdr[2]; h->nblk = hdr[3]; h->levels = hdr2[0]; h->sym_bytes = hdr2[1]; h->sb_shift = hdr2[2]; }
    ok = ok && rd(f, h->dbase) && rd(f, h->sbase) && rd(f, h->C) && rd(f, h->leaf) && rd(f, h->q1) && rd(f, h->wm) &&
         rd(f, h->sa_lo) && rd(f, h->sa_hi) && rd(f, h->text) && rd(f, h->bwt);
    fclose(f);
    h->dlevels = (h->levels + FMI_DIGIT_BITS - 1) / FMI_DIGIT_BITS;
    h->nsb = ok && h->sb_shift < 64 ? (h->nblk >> h->sb_shift) + 1 : 0;
    if (!ok || h->sbase.size() != (size_t)h->dlevels * h->nsb * FMI_ARITY || h->levels == 0 || h->levels > FMI_MAX_LEVELS || h->dbase.size() != (size_t)h->dlevels * FMI_ARITY ||
        h->wm.size() != (uint64_t)h->dlevels * h->nblk * FMI_BLOCK_WORDS) {
        delete h;
        fmi_set_error("%s is not a SEALFMI4 index", path);
        return FMI_ERR_IO;
    }
    h->host_resident = true;
    if (device >= 0) {
        int rc = fmi_upload(h, device);
        if (rc != FMI_OK) { delete h; return rc; }
    }
    *out = h;
    return FMI_OK;
}

extern "C" const void *fmi_host_array(const fmi_t *h, const char *name, uint64_t *n_out, uint32_t *elem_out)
{
    if (!h || !name || !h->host_resident) return nullptr;
    std::string s(name);
    auto ret = [&](const void *p, uint64_t n, uint32_t e) { if (n_out) *n_out = n; if (elem_out) *elem_out = e; return p; };
    if (s == "sa") return ret(h->sa_lo.data(), h->sa_lo.size(), 4);
    if (s == "sa_hi") return ret(h->sa_hi.data(), h->sa_hi.size(), 1);
    if (s == "bwt") return ret(h->bwt.data(), h->bwt.size(), 4);
    if (s == "text") return ret(h->text.data(), h->n, h->sym_bytes);
    if (s == "C") return ret(h->C.data(), h->C.size(), 8);
    if (s == "leaf") return ret(h->leaf.data(), h->leaf.size(), 8);
    if (s == "q1") return ret(h->q1.data(), h->q1.size(), 1);
    if (s == "dbase") return ret(h->dbase.data(), h->dbase.size(), 8);
    if (s == "sbase") return ret(h->sbase.data(), h->sbase.size(), 8);
    if (s == "wm") return ret(h->wm.data(), h->wm.size(), 8);
    return nullptr;
}
