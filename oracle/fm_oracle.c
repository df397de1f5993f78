/*
 * fm_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the arithmetic
 * behind SEAL's FM-index path, used as the parity checker for the HIP kernels
 * in seal_amd/csrc and as the "port" CPU baseline in bench.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (seal_amd/) never does.
 *
 * PARITY UNPINNED: the arithmetic of the reference lives in the third-party
 * dependency simongog/sdsl-lite (git submodule res/external/sdsl-lite, empty in
 * /root/reference, no pinned SHA recoverable; SEAL-era upstream master ~v2.1.1).
 * The reference holds no tests / golden vectors for this path, and neither sdsl
 * nor SWIG is installed, so the reference cannot be executed here.  What this
 * file does instead:
 *   - restates sdsl's *published* algorithms for the type the reference
 *     instantiates, `csa_wt_int<>` = csa_wt<wt_int<>, 32, 64,
 *     sa_order_sa_sampling<>, isa_sampling<>, int_alphabet<>>
 *     (/root/reference/seal/cpp_modules/fm_index.hpp:12): a level-concatenated
 *     pointerless wavelet tree (wt_int) over the BWT with one rank_support_v
 *     (512-bit superblocks: 64-bit absolute + 7x9-bit relative counts),
 *     SA samples every 32 rows, ISA samples every 64 text positions;
 *   - restates the wrapper fm_index.cpp entry point by entry point (cited
 *     below as "ref cpp:LINE");
 *   - is itself pinned by tests/test_oracle.py against a brute-force suffix
 *     sort + naive substring counting (mathematical facts: counts, SA
 *     positions, doc ids are layout independent) and the SURVEY.md G1/G2
 *     vectors.
 * Everything ABOVE this file is pinned against the reference's own Python, run
 * in this container on top of this file (tests/golden/make_reference_golden.py,
 * tests/test_reference_golden.py): seal/index.py, the logits processor and the
 * whole decode of seal/beam_search.py, aggregate_evidence / rescore_keys /
 * compute_unigram_scores of seal/keys.py.  "Unpinned" is this C layer only.
 * The only layout-dependent behaviour is quirk Q1 (search started from the
 * inclusive upper end r = size(), index.py:106-107, one past the last row, with
 * sdsl's asserts compiled out); it falls out of the faithful wt_int::rank loop
 * below reading one bit past the node at every level.
 *
 * Suffix array construction here is a plain comparison sort (the text ends in
 * a unique smallest sentinel, so every comparison terminates) -- deliberately
 * independent of the product's builders.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint64_t n;          /* csa.size(): text length incl. the appended 0 sentinel */
    uint32_t max_level;  /* wt_int::m_max_level = bits::hi(max symbol)+1 */
    uint64_t sigma;      /* number of distinct symbols incl. sentinel */
    uint64_t max_sym;
    uint64_t tree_bits;  /* n * max_level */
    uint64_t *tree;      /* wt_int::m_tree, level-concatenated bit vector */
    uint64_t *bb;        /* rank_support_v::m_basic_block, 2 words / 512 bits */
    uint64_t *char2comp; /* int_alphabet: symbol -> compact id (0 if absent) */
    uint8_t  *present;
    uint64_t *comp2char;
    uint64_t *C;         /* sigma+1 cumulative counts */
    uint64_t *sa_sample; /* SA[i] for i % 32 == 0  (sa_order_sa_sampling<>, t_dens=32) */
    uint64_t *isa_sample;/* ISA[j] for j % 64 == 0 (isa_sampling<>, t_inv_dens=64) */
} orc_t;

#define SA_DENS 32
#define ISA_DENS 64

/* ---- rank_support_v (sdsl rank_support_v.hpp, restated) ------------------ */
static inline uint64_t tree_rank(const orc_t *o, uint64_t idx)
{
    const uint64_t *p = o->bb + ((idx >> 8) & 0xFFFFFFFFFFFFFFFEULL);
    uint64_t r = p[0] + ((p[1] >> (63 - 9 * ((idx & 0x1FF) >> 6))) & 0x1FF);
    if (idx & 0x3F)
        r += (uint64_t)__builtin_popcountll(o->tree[idx >> 6] & ((1ULL << (idx & 0x3F)) - 1));
    return r;
}
static inline int tree_bit(const orc_t *o, uint64_t idx)
{
    return (int)((o->tree[idx >> 6] >> (idx & 63)) & 1);
}

static void build_rank_support(orc_t *o)
{
    uint64_t nblocks = (o->tree_bits >> 9) + 2;
    o->bb = (uint64_t *)calloc(nblocks * 2, 8);
    uint64_t nwords = (o->tree_bits + 63) / 64;
    /* per-superblock relative counts in parallel, absolute counts by one serial sweep */
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)nblocks; b++) {
        uint64_t rel = 0, packed = 0;
        for (int w = 0; w < 8; w++) {
            uint64_t wi = (uint64_t)b * 8 + w;
            if (w > 0) packed |= rel << (63 - 9 * w);
            if (wi < nwords) rel += (uint64_t)__builtin_popcountll(o->tree[wi]);
        }
        o->bb[2 * b + 1] = packed;
        o->bb[2 * b] = rel;             /* block total for now */
    }
    uint64_t abs_cnt = 0;
    for (uint64_t b = 0; b < nblocks; b++) { uint64_t t = o->bb[2 * b]; o->bb[2 * b] = abs_cnt; abs_cnt += t; }
}

/* ---- wt_int (sdsl wt_int.hpp, restated) ---------------------------------- */
/* rank(i, c): occurrences of c in prefix [0, i).  Faithful to the published
 * loop incl. its behaviour for i == size()+1 (quirk Q1). */
static uint64_t wt_rank(const orc_t *o, uint64_t i, uint64_t c)
{
    if ((1ULL << o->max_level) <= c) return 0;
    uint64_t offset = 0, node_size = o->n;
    uint64_t mask = 1ULL << (o->max_level - 1);
    for (uint32_t k = 0; k < o->max_level && i; ++k) {
        uint64_t ones_before_o = tree_rank(o, offset);
        uint64_t ones_before_i = tree_rank(o, offset + i) - ones_before_o;
        uint64_t ones_before_end = tree_rank(o, offset + node_size) - ones_before_o;
        if (c & mask) {
            offset += node_size - ones_before_end;
            node_size = ones_before_end;
            i = ones_before_i;
        } else {
            node_size = node_size - ones_before_end;
            i = i - ones_before_i;
        }
        offset += o->n;
        mask >>= 1;
    }
    return i;
}

/* inverse_select(i) -> (rank of wt[i] among equal symbols before i, wt[i]) */
static void wt_inverse_select(const orc_t *o, uint64_t i, uint64_t *rank_out, uint64_t *sym_out)
{
    uint64_t c = 0, offset = 0, node_size = o->n;
    for (uint32_t k = 0; k < o->max_level; ++k) {
        uint64_t ones_before_o = tree_rank(o, offset);
        uint64_t ones_before_i = tree_rank(o, offset + i) - ones_before_o;
        uint64_t ones_before_end = tree_rank(o, offset + node_size) - ones_before_o;
        c <<= 1;
        if (tree_bit(o, offset + i)) {
            offset += node_size - ones_before_end;
            node_size = ones_before_end;
            i = ones_before_i;
            c |= 1;
        } else {
            node_size = node_size - ones_before_end;
            i = i - ones_before_i;
        }
        offset += o->n;
    }
    *rank_out = i;
    *sym_out = c;
}

/* recursive left-first expansion => ascending symbol order */
static void wt_interval_symbols_rec(const orc_t *o, uint64_t i, uint64_t j, uint64_t *k,
                                    uint64_t *cs, uint64_t *rank_c_i, uint64_t *rank_c_j,
                                    uint32_t level, uint64_t path, uint64_t node_size, uint64_t offset)
{
    if (level >= o->max_level) {
        rank_c_i[*k] = i;
        rank_c_j[*k] = j;
        cs[(*k)++] = path;
        return;
    }
    uint64_t ones_before_o = tree_rank(o, offset);
    uint64_t ones_before_i = tree_rank(o, offset + i) - ones_before_o;
    uint64_t ones_before_j = tree_rank(o, offset + j) - ones_before_o;
    uint64_t ones_before_end = tree_rank(o, offset + node_size) - ones_before_o;
    if ((j - i) - (ones_before_j - ones_before_i) > 0) {
        wt_interval_symbols_rec(o, i - ones_before_i, j - ones_before_j, k, cs, rank_c_i, rank_c_j,
                                level + 1, path << 1, node_size - ones_before_end, offset + o->n);
    }
    if ((ones_before_j - ones_before_i) > 0) {
        wt_interval_symbols_rec(o, ones_before_i, ones_before_j, k, cs, rank_c_i, rank_c_j,
                                level + 1, (path << 1) | 1, ones_before_end,
                                offset + (node_size - ones_before_end) + o->n);
    }
}

static void wt_interval_symbols(const orc_t *o, uint64_t i, uint64_t j, uint64_t *k,
                                uint64_t *cs, uint64_t *rank_c_i, uint64_t *rank_c_j)
{
    *k = 0;
    if (i == j) return;
    if (i + 1 == j) {
        uint64_t r, c;
        wt_inverse_select(o, i, &r, &c);
        cs[0] = c; rank_c_i[0] = r; rank_c_j[0] = r + 1; *k = 1;
        return;
    }
    wt_interval_symbols_rec(o, i, j, k, cs, rank_c_i, rank_c_j, 0, 0, o->n, 0);
}

/* ---- csa_wt pieces -------------------------------------------------------- */
static inline uint64_t csa_lf(const orc_t *o, uint64_t i)
{
    uint64_t r, c;
    wt_inverse_select(o, i, &r, &c);
    return o->C[o->char2comp[c]] + r;
}

/* csa[i]: walk LF until a sampled row, add the walked distance (mod n) */
static uint64_t csa_sa(const orc_t *o, uint64_t i)
{
    uint64_t off = 0;
    while (i % SA_DENS) { i = csa_lf(o, i); ++off; }
    uint64_t result = o->sa_sample[i / SA_DENS];
    return (result + off < o->n) ? result + off : result + off - o->n;
}

/* csa.isa[j]: nearest text-order sample at/after j, walk LF back */
static uint64_t csa_isa(const orc_t *o, uint64_t j)
{
    uint64_t sp = ((j + ISA_DENS - 1) / ISA_DENS) * ISA_DENS;
    uint64_t row, steps;
    if (sp < o->n) { row = o->isa_sample[sp / ISA_DENS]; steps = sp - j; }
    else { row = o->isa_sample[0]; steps = o->n - j; } /* wrap to text position 0 */
    while (steps--) row = csa_lf(o, row);
    return row;
}

/* sdsl backward_search(csa, l, r, c, l_res, r_res) on inclusive [l, r] */
static void csa_backward_search(const orc_t *o, uint64_t l, uint64_t r, uint64_t c,
                                uint64_t *l_res, uint64_t *r_res)
{
    uint64_t cc = (c <= o->max_sym && o->present[c]) ? o->char2comp[c] : 0;
    if (cc == 0 && c > 0) { *l_res = 1; *r_res = 0; return; }
    uint64_t c_begin = o->C[cc];
    if (l == 0 && r + 1 == o->n) {
        *l_res = c_begin;
        *r_res = o->C[cc + 1] - 1;
    } else {
        *l_res = c_begin + wt_rank(o, l, c);
        *r_res = c_begin + wt_rank(o, r + 1, c) - 1;
    }
}

/* ---- construction --------------------------------------------------------- */
static const uint64_t *g_text;
static int suffix_cmp(const void *a, const void *b)
{
    uint64_t i = *(const uint64_t *)a, j = *(const uint64_t *)b;
    if (i == j) return 0;
    const uint64_t *t = g_text;
    while (t[i] == t[j]) { ++i; ++j; } /* terminates: sentinel is unique */
    return t[i] < t[j] ? -1 : 1;
}

static int orc_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void finish_build(orc_t *o, const uint32_t *bwt)
{
    uint64_t n = o->n;
    uint64_t max_sym = 0;
#pragma omp parallel for reduction(max : max_sym) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) if (bwt[i] > max_sym) max_sym = bwt[i];
    o->max_sym = max_sym;
    uint32_t L = 0;
    while ((max_sym >> L) > 0) L++;
    if (L == 0) L = 1;
    o->max_level = L;
    const int T = orc_threads();
    /* alphabet */
    uint64_t *occ = (uint64_t *)calloc(max_sym + 2, 8);
    {
        uint64_t *part = (uint64_t *)calloc((size_t)T * (max_sym + 1), 8);
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            uint64_t *h = part + (size_t)t * (max_sym + 1);
            uint64_t a = n * (uint64_t)t / T, b = n * (uint64_t)(t + 1) / T;
            for (uint64_t i = a; i < b; i++) h[bwt[i]]++;
        }
        for (int t = 0; t < T; t++) for (uint64_t c = 0; c <= max_sym; c++) occ[c] += part[(size_t)t * (max_sym + 1) + c];
        free(part);
    }
    o->present = (uint8_t *)calloc(max_sym + 1, 1);
    o->char2comp = (uint64_t *)calloc(max_sym + 1, 8);
    uint64_t sigma = 0;
    for (uint64_t c = 0; c <= max_sym; c++) if (occ[c]) { o->present[c] = 1; o->char2comp[c] = sigma++; }
    o->sigma = sigma;
    o->comp2char = (uint64_t *)calloc(sigma, 8);
    o->C = (uint64_t *)calloc(sigma + 1, 8);
    uint64_t acc = 0, cc = 0;
    for (uint64_t c = 0; c <= max_sym; c++) if (occ[c]) { o->comp2char[cc] = c; o->C[cc] = acc; acc += occ[c]; cc++; }
    o->C[sigma] = acc;
    free(occ);
    /* wt_int bit tree: level k holds, for the sequence stably sorted by its top-k
       bits (nodes contiguous in prefix order), bit (L-1-k) of every element.
       Per level: threads own contiguous chunks; bits are flushed a word at a time
       (atomically: chunk and level boundaries share words), then a parallel stable
       counting sort by the top (k+1) bits produces the next level's order. */
    o->tree_bits = n * L;
    o->tree = (uint64_t *)calloc(o->tree_bits / 64 + 3, 8);
    uint32_t *cur = (uint32_t *)malloc(n * 4), *nxt = (uint32_t *)malloc(n * 4);
    memcpy(cur, bwt, n * 4);
    uint64_t *hist = (uint64_t *)malloc(((size_t)T << L) * 8);
    for (uint32_t k = 0; k < L; k++) {
        const uint32_t sh = L - 1 - k;
        const uint64_t nb = 1ULL << (k + 1);
        const int last = (k + 1 == L);
        if (!last) memset(hist, 0, (size_t)T * nb * 8);
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            uint64_t a = n * (uint64_t)t / T, b = n * (uint64_t)(t + 1) / T;
            uint64_t *h = hist + (size_t)t * nb;
            uint64_t word = 0, widx = ((uint64_t)k * n + a) >> 6;
            for (uint64_t i = a; i < b; i++) {
                uint64_t p = (uint64_t)k * n + i;
                if ((p >> 6) != widx) {
                    if (word) __atomic_fetch_or(&o->tree[widx], word, __ATOMIC_RELAXED);
                    word = 0; widx = p >> 6;
                }
                uint32_t v = cur[i];
                word |= (uint64_t)((v >> sh) & 1) << (p & 63);
                if (!last) h[v >> sh]++;
            }
            if (word) __atomic_fetch_or(&o->tree[widx], word, __ATOMIC_RELAXED);
        }
        if (last) break;
        uint64_t run = 0;
        for (uint64_t bkt = 0; bkt < nb; bkt++)
            for (int t = 0; t < T; t++) { uint64_t c = hist[(size_t)t * nb + bkt]; hist[(size_t)t * nb + bkt] = run; run += c; }
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            uint64_t a = n * (uint64_t)t / T, b = n * (uint64_t)(t + 1) / T;
            uint64_t *h = hist + (size_t)t * nb;
            for (uint64_t i = a; i < b; i++) nxt[h[cur[i] >> sh]++] = cur[i];
        }
        uint32_t *tt = cur; cur = nxt; nxt = tt;
    }
    free(hist); free(cur); free(nxt);
    build_rank_support(o);
}

void orc_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* ref cpp:33-41 FMIndex::initialize -> construct_im(index, data, 0): sdsl appends
 * the 0 sentinel, builds SA/BWT/wt/samples.  data must not contain 0. */
orc_t *orc_build(const uint64_t *data, uint64_t n_data)
{
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    uint64_t n = n_data + 1;
    o->n = n;
    uint64_t *text = (uint64_t *)malloc(n * 8);
    memcpy(text, data, n_data * 8);
    text[n_data] = 0;
    uint64_t *sa = (uint64_t *)malloc(n * 8);
    for (uint64_t i = 0; i < n; i++) sa[i] = i;
    g_text = text;
    qsort(sa, n, 8, suffix_cmp);
    uint32_t *bwt = (uint32_t *)malloc(n * 4);
    for (uint64_t i = 0; i < n; i++) bwt[i] = (uint32_t)(sa[i] ? text[sa[i] - 1] : text[n - 1]);
    o->sa_sample = (uint64_t *)calloc(n / SA_DENS + 1, 8);
    o->isa_sample = (uint64_t *)calloc(n / ISA_DENS + 1, 8);
    for (uint64_t i = 0; i < n; i++) {
        if (i % SA_DENS == 0) o->sa_sample[i / SA_DENS] = sa[i];
        if (sa[i] % ISA_DENS == 0) o->isa_sample[sa[i] / ISA_DENS] = i;
    }
    finish_build(o, bwt);
    free(bwt); free(sa); free(text);
    return o;
}

/* bench-only: assemble the same structures from a BWT + samples computed
 * elsewhere (the CPU baseline at index sizes a comparison sort cannot reach). */
orc_t *orc_build_from_bwt(const uint32_t *bwt32, uint64_t n,
                          const uint64_t *sa_sample, const uint64_t *isa_sample)
{
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    o->n = n;
    o->sa_sample = (uint64_t *)malloc((n / SA_DENS + 1) * 8);
    o->isa_sample = (uint64_t *)malloc((n / ISA_DENS + 1) * 8);
    memcpy(o->sa_sample, sa_sample, ((n + SA_DENS - 1) / SA_DENS) * 8);
    memcpy(o->isa_sample, isa_sample, ((n + ISA_DENS - 1) / ISA_DENS) * 8);
    finish_build(o, bwt32);
    return o;
}

void orc_free(orc_t *o)
{
    if (!o) return;
    free(o->tree); free(o->bb); free(o->char2comp); free(o->present);
    free(o->comp2char); free(o->C); free(o->sa_sample); free(o->isa_sample);
    free(o);
}

/* ---- the wrapper, entry point by entry point (fm_index.cpp) --------------- */

/* ref cpp:50-52 */
uint64_t orc_size(const orc_t *o) { return o->n; }
uint64_t orc_sigma(const orc_t *o) { return o->sigma; }
uint32_t orc_max_level(const orc_t *o) { return o->max_level; }

/* ref cpp:67-76: one step on the INCLUSIVE interval [low, high] */
void orc_backward_search_step(const orc_t *o, uint64_t symbol, uint64_t low, uint64_t high, uint64_t out[2])
{
    csa_backward_search(o, low, high, symbol, &out[0], &out[1]);
}

/* ref cpp:55-65: from (0, size()) -- note r = size(), not size()-1 -- returns (l, r+1) */
void orc_backward_search_multi(const orc_t *o, const uint64_t *query, uint64_t len, uint64_t out[2])
{
    uint64_t l = 0, r = o->n;
    for (uint64_t i = 0; i < len; i++) csa_backward_search(o, l, r, query[i], &l, &r);
    out[0] = l; out[1] = r + 1;
}

/* ref cpp:91-109: flat [c0, n0, c1, n1, ...]; fresh sigma-sized scratch per call
 * (the per-call allocation is part of the reference's cost and is kept).
 * out must hold 2*sigma entries; returns the number of entries written. */
uint64_t orc_distinct_count(const orc_t *o, uint64_t low, uint64_t high, uint64_t *out)
{
    if (low == high) return 0;
    uint64_t *cs = (uint64_t *)calloc(o->sigma, 8);
    uint64_t *ri = (uint64_t *)calloc(o->sigma, 8);
    uint64_t *rj = (uint64_t *)calloc(o->sigma, 8);
    uint64_t k = 0;
    wt_interval_symbols(o, low, high, &k, cs, ri, rj);
    for (uint64_t i = 0; i < k; i++) { out[2 * i] = cs[i]; out[2 * i + 1] = rj[i] - ri[i]; }
    free(cs); free(ri); free(rj);
    return 2 * k;
}

/* ref cpp:78-89: symbols only */
uint64_t orc_distinct(const orc_t *o, uint64_t low, uint64_t high, uint64_t *out)
{
    if (low == high) return 0;
    uint64_t *cs = (uint64_t *)calloc(o->sigma, 8);
    uint64_t *ri = (uint64_t *)calloc(o->sigma, 8);
    uint64_t *rj = (uint64_t *)calloc(o->sigma, 8);
    uint64_t k = 0;
    wt_interval_symbols(o, low, high, &k, cs, ri, rj);
    for (uint64_t i = 0; i < k; i++) out[i] = cs[i];
    free(cs); free(ri); free(rj);
    return k;
}

/* ref cpp:111-131: one task per interval (std::async there, OpenMP here), joined
 * in order.  Two-call protocol: out_sizes[i] = entries of interval i; when
 * out != NULL results are written at out + out_offsets[i]. */
void orc_distinct_count_multi(const orc_t *o, uint64_t m, const uint64_t *lows, const uint64_t *highs,
                              uint64_t **bufs, uint64_t *out_sizes)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        uint64_t w = highs[i] > lows[i] ? highs[i] - lows[i] : 0;
        uint64_t cap = 2 * (w < o->sigma ? w : o->sigma) + 2;
        bufs[i] = (uint64_t *)malloc(cap * 8);
        out_sizes[i] = orc_distinct_count(o, lows[i], highs[i], bufs[i]);
    }
}
void orc_free_buf(uint64_t *p) { free(p); }

/* ref cpp:163-167 */
uint64_t orc_locate(const orc_t *o, uint64_t row)
{
    if (row >= o->n) return (uint64_t)-1;
    return csa_sa(o, row);
}

/* ref cpp:169-184: T[end-1], T[end-2], ..., T[begin] */
uint64_t orc_extract_text(const orc_t *o, uint64_t begin, uint64_t end, uint64_t *out)
{
    if (end - begin == 0) return 0;
    uint64_t start = csa_isa(o, end);
    uint64_t r, symbol;
    wt_inverse_select(o, start, &r, &symbol);
    uint64_t k = 0;
    out[k++] = symbol;
    if (end - begin == 1) return k;
    for (uint64_t i = 0; i < end - begin - 1; i++) {
        uint64_t res[2];
        orc_backward_search_step(o, symbol, start, start + 1, res);
        start = res[0];
        wt_inverse_select(o, start, &r, &symbol);
        out[k++] = symbol;
    }
    return k;
}

/* raw access used by tests (csa.bwt[i], csa[i], csa.isa[j]) */
uint64_t orc_bwt(const orc_t *o, uint64_t i) { uint64_t r, c; wt_inverse_select(o, i, &r, &c); return c; }
uint64_t orc_sa(const orc_t *o, uint64_t i) { return csa_sa(o, i); }
uint64_t orc_isa(const orc_t *o, uint64_t j) { return csa_isa(o, j); }
uint64_t orc_rank(const orc_t *o, uint64_t i, uint64_t c) { return wt_rank(o, i, c); }

/* ---- batched drivers for the CPU baseline (bench.py only) ------------------
 * They replay the reference's call pattern over many inputs so that the timing
 * is not dominated by ctypes overhead (the reference pays SWIG overhead there;
 * leaving it out flatters the baseline). */

/* index.py:102-111 get_range per sequence, tokens already shifted; CSR input */
void orc_get_range_batch(const orc_t *o, uint64_t m, const uint64_t *offsets, const uint64_t *tokens,
                         uint64_t *lo_out, uint64_t *hi_out, int threads)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
    for (int64_t s = 0; s < (int64_t)m; s++) {
        uint64_t l = 0, r = o->n;
        for (uint64_t t = offsets[s]; t < offsets[s + 1]; t++) csa_backward_search(o, l, r, tokens[t], &l, &r);
        lo_out[s] = l; hi_out[s] = r + 1;
    }
}

/* keys.py:320-324: locate(row) + bisect_right(beginnings, pos) - 1 */
void orc_locate_bin_batch(const orc_t *o, uint64_t m, const uint64_t *rows,
                          const uint64_t *beginnings, uint64_t nb,
                          uint64_t *pos_out, uint64_t *doc_out, int threads)
{
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        uint64_t pos = orc_locate(o, rows[i]);
        uint64_t lo = 0, hi = nb;
        while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (pos < beginnings[mid]) hi = mid; else lo = mid + 1; }
        pos_out[i] = pos; doc_out[i] = lo - 1;
    }
}

/* beam_search.py:107 get_distinct_count_multi, timing-only variant that keeps
 * just the number of distinct symbols per interval */
void orc_distinct_count_sizes(const orc_t *o, uint64_t m, const uint64_t *lows, const uint64_t *highs,
                              uint64_t *k_out, int threads)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        uint64_t w = highs[i] > lows[i] ? highs[i] - lows[i] : 0;
        uint64_t cap = 2 * (w < o->sigma ? w : o->sigma) + 2;
        uint64_t *buf = (uint64_t *)malloc(cap * 8);
        k_out[i] = orc_distinct_count(o, lows[i], highs[i], buf) / 2;
        free(buf);
    }
}

/* index.py:68-75 get_doc for many documents (extract_text per document), timing-only: keeps lengths */
void orc_extract_batch(const orc_t *o, uint64_t m, const uint64_t *begins, const uint64_t *ends, uint64_t *len_out, int threads)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        uint64_t n = ends[i] - begins[i];
        uint64_t *buf = (uint64_t *)malloc((n + 1) * 8);
        len_out[i] = orc_extract_text(o, begins[i], ends[i], buf);
        free(buf);
    }
}

/* ---- result-returning forms of the drivers above: bench.py replays the recorded operations of a batch through
 * these, times them as the CPU baseline AND compares what they return with what the GPU returned (parity_check) */

/* get_distinct_count_multi (index.py:158-171) per interval, as a bitmap over raw token ids: bit (symbol - shift) of
 * row i set for every distinct symbol > 0 of rows [lows[i], highs[i]); k_out = number of distinct symbols > 0;
 * count_sum_out = sum of their multiplicities */
void orc_distinct_bitmaps(const orc_t *o, uint64_t m, const uint64_t *lows, const uint64_t *highs, uint64_t shift,
                          uint64_t words_per_row, uint32_t *bits_out, uint64_t *k_out, uint64_t *count_sum_out, int threads)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        uint64_t w = highs[i] > lows[i] ? highs[i] - lows[i] : 0;
        uint64_t cap = 2 * (w < o->sigma ? w : o->sigma) + 2;
        uint64_t *buf = (uint64_t *)malloc(cap * 8);
        uint64_t len = orc_distinct_count(o, lows[i], highs[i], buf), k = 0, cs = 0;
        uint32_t *row = bits_out + (uint64_t)i * words_per_row;
        for (uint64_t j = 0; j + 1 < len; j += 2) {
            if (buf[j] == 0) continue;                      /* sentinel: index.py:153,167 */
            uint64_t t = buf[j] - shift;
            if (buf[j] >= shift && (t >> 5) < words_per_row) row[t >> 5] |= 1u << (t & 31);
            k++; cs += buf[j + 1];
        }
        k_out[i] = k; count_sum_out[i] = cs;
        free(buf);
    }
}

/* get_doc (index.py:68-75) for many documents into one flat buffer at the given offsets (symbols, not yet un-shifted) */
void orc_extract_batch_tokens(const orc_t *o, uint64_t m, const uint64_t *begins, const uint64_t *ends,
                              const uint64_t *out_offsets, uint64_t *out, int threads)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)m; i++) orc_extract_text(o, begins[i], ends[i], out + out_offsets[i]);
}

/* ---- sdsl-lite on-disk format of csa_wt_int<> (store_to_file, ref cpp:186-189) -------------------------------------
 * Written from this oracle's own structures so that the product's reader of real SEAL indices (seal_amd/csrc/
 * fmi_sdsl.cpp) can be round-trip tested without sdsl.  PARITY UNPINNED: the layout below is restated from the published
 * sdsl-lite v2.1.x sources (csa_wt::serialize = wavelet_tree, sa_sample, isa_sample, alphabet; wt_int::serialize = size,
 * sigma, tree, tree_rank, tree_select_1, tree_select_0, max_level; int_alphabet = sd_vector m_char (+ stateless rank /
 * select), m_C, m_sigma) and has never been compared with a file sdsl wrote -- the sdsl sources are not in this image.
 * int_vector<w>: u64 size in BITS (+ u8 width when w == 0), then ceil(size/64) little-endian words, elements packed
 * LSB-first. */
static void w_u64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }
static void w_u8(FILE *f, uint8_t v) { fwrite(&v, 1, 1, f); }
static uint8_t hi_bit(uint64_t x) { uint8_t h = 0; while (x >>= 1) h++; return h; }   /* bits::hi, hi(0) = 0 */

static void w_words(FILE *f, const uint64_t *w, uint64_t nbits) { if (nbits) fwrite(w, 8, (nbits + 63) / 64, f); }
static void w_bit_vector(FILE *f, const uint64_t *w, uint64_t nbits) { w_u64(f, nbits); w_words(f, w, nbits); }

static void w_int_vector0(FILE *f, const uint64_t *vals, uint64_t n, uint8_t width)
{
    uint64_t nbits = n * width, nw = (nbits + 63) / 64;
    uint64_t *buf = (uint64_t *)calloc(nw + 1, 8);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t pos = i * width, v = width < 64 ? (vals[i] & ((1ULL << width) - 1)) : vals[i];
        buf[pos >> 6] |= v << (pos & 63);
        if ((pos & 63) + width > 64) buf[(pos >> 6) + 1] |= v >> (64 - (pos & 63));
    }
    w_u64(f, nbits); w_u8(f, width); w_words(f, buf, nbits);
    free(buf);
}

/* rank_support_v<1>::serialize = its int_vector<64> of ((capacity >> 9) + 1) * 2 words */
static void w_rank_support_v(FILE *f, const uint64_t *bits, uint64_t nbits)
{
    uint64_t cap = ((nbits + 63) >> 6) << 6, nblk = (cap >> 9) + 1, nwords = cap >> 6;
    uint64_t *bb = (uint64_t *)calloc(2 * nblk, 8), abs_cnt = 0;
    for (uint64_t b = 0; b < nblk; b++) {
        uint64_t rel = 0, packed = 0;
        for (int w = 0; w < 8; w++) {
            uint64_t wi = b * 8 + w;
            if (w > 0) packed |= rel << (63 - 9 * w);
            if (wi < nwords) rel += (uint64_t)__builtin_popcountll(bits[wi]);
        }
        bb[2 * b] = abs_cnt; bb[2 * b + 1] = packed;
        abs_cnt += rel;
    }
    w_u64(f, 2 * nblk * 64); fwrite(bb, 8, 2 * nblk, f);
    free(bb);
}

/* select_support_mcl<b,1>::serialize: arg count; then (if any) the superblock vector, the mini-or-long indicator and
 * per superblock of 4096 arguments either all positions (span > log^4) or every 64th position relative to the first */
static void w_select_mcl(FILE *f, const uint64_t *bits, uint64_t nbits, int b)
{
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < nbits; i++) cnt += (((bits[i >> 6] >> (i & 63)) & 1) == (uint64_t)b);
    w_u64(f, cnt);
    if (!cnt) return;
    uint64_t cap = ((nbits + 63) >> 6) << 6;
    uint8_t logn = hi_bit(cap) + 1;
    uint64_t logn4 = (uint64_t)logn * logn * logn * logn;
    uint64_t sb = (cnt + 4095) >> 12;
    uint64_t *first = (uint64_t *)calloc(sb, 8);
    uint64_t *pos = (uint64_t *)malloc(4096 * 8);
    uint8_t *is_long = (uint8_t *)calloc(sb, 1);
    int any_long = 0;
    /* pass 1: superblock starts and kinds */
    uint64_t k = 0, s = 0;
    for (uint64_t i = 0; i < nbits; i++) {
        if ((((bits[i >> 6] >> (i & 63)) & 1) != (uint64_t)b)) continue;
        pos[k & 4095] = i; k++;
        if ((k & 4095) == 0 || k == cnt) {
            uint64_t in_sb = (k & 4095) ? (k & 4095) : 4096;
            first[s] = pos[0];
            if (pos[in_sb - 1] - pos[0] > logn4) { is_long[s] = 1; any_long = 1; }
            s++;
        }
    }
    w_int_vector0(f, first, sb, logn);
    if (any_long) {
        uint64_t *mol = (uint64_t *)calloc((sb + 63) / 64 + 1, 8);
        for (uint64_t i = 0; i < sb; i++) if (!is_long[i]) mol[i >> 6] |= 1ULL << (i & 63);    /* bit = "has miniblocks" */
        w_bit_vector(f, mol, sb);
        free(mol);
    } else {
        w_bit_vector(f, NULL, 0);
    }
    /* pass 2: the blocks themselves, in superblock order */
    k = 0; s = 0;
    uint64_t mini[64];
    for (uint64_t i = 0; i < nbits; i++) {
        if ((((bits[i >> 6] >> (i & 63)) & 1) != (uint64_t)b)) continue;
        pos[k & 4095] = i; k++;
        if ((k & 4095) == 0 || k == cnt) {
            uint64_t in_sb = (k & 4095) ? (k & 4095) : 4096;
            if (is_long[s]) {
                uint64_t *all = (uint64_t *)calloc(4096, 8);
                memcpy(all, pos, in_sb * 8);
                w_int_vector0(f, all, 4096, hi_bit(pos[in_sb - 1]) + 1);
                free(all);
            } else {
                memset(mini, 0, sizeof(mini));
                for (uint64_t j = 0; j < in_sb; j += 64) mini[j / 64] = pos[j] - pos[0];
                w_int_vector0(f, mini, 64, hi_bit(pos[in_sb - 1] - pos[0]) + 1);
            }
            s++;
        }
    }
    free(first); free(pos); free(is_long);
}

/* sd_vector<> of a bit vector with ones at ones[0..m) (ascending) over [0, size) */
static void w_sd_vector(FILE *f, const uint64_t *ones, uint64_t m, uint64_t size)
{
    uint8_t logm = hi_bit(m) + 1, logn = hi_bit(size) + 1;
    if (logm == logn) --logm;
    uint8_t wl = logn - logm;
    uint64_t hbits = m + (1ULL << logm);
    uint64_t *high = (uint64_t *)calloc((hbits + 63) / 64 + 1, 8), *low = (uint64_t *)calloc(m + 1, 8);
    uint64_t highpos = 0, last_high = 0;
    for (uint64_t i = 0; i < m; i++) {
        uint64_t cur_high = ones[i] >> wl;
        highpos += cur_high - last_high;
        last_high = cur_high;
        low[i] = wl ? (ones[i] & ((1ULL << wl) - 1)) : 0;
        high[highpos >> 6] |= 1ULL << (highpos & 63);
        highpos++;
    }
    w_u64(f, size); w_u8(f, wl);
    w_int_vector0(f, low, m, wl);
    w_bit_vector(f, high, hbits);
    w_select_mcl(f, high, hbits, 1);
    w_select_mcl(f, high, hbits, 0);
    free(high); free(low);
}

int orc_save_sdsl(const orc_t *o, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    /* wt_int */
    w_u64(f, o->n); w_u64(f, o->sigma);
    w_bit_vector(f, o->tree, o->tree_bits);
    w_rank_support_v(f, o->tree, o->tree_bits);
    w_select_mcl(f, o->tree, o->tree_bits, 1);
    w_select_mcl(f, o->tree, o->tree_bits, 0);
    { uint32_t ml = o->max_level; fwrite(&ml, 4, 1, f); }
    /* sa_order_sa_sampling<>: SA[i], i % 32 == 0; isa_sampling<>: ISA[j], j % 64 == 0; both int_vector<0> of width hi(n)+1 */
    uint8_t w = hi_bit(o->n) + 1;
    w_int_vector0(f, o->sa_sample, (o->n + SA_DENS - 1) / SA_DENS, w);
    w_int_vector0(f, o->isa_sample, (o->n + ISA_DENS - 1) / ISA_DENS, w);
    /* int_alphabet<>: m_char (sd_vector over [0, max_sym]), rank / select supports of an sd_vector hold no data, m_C, m_sigma */
    uint64_t *ones = (uint64_t *)malloc((o->sigma + 1) * 8), m = 0;
    for (uint64_t c = 0; c <= o->max_sym; c++) if (o->present[c]) ones[m++] = c;
    w_sd_vector(f, ones, m, o->max_sym + 1);
    w_int_vector0(f, o->C, o->sigma + 1, w);
    w_u64(f, o->sigma);
    free(ones);
    int bad = ferror(f);
    fclose(f);
    return bad ? -1 : 0;
}
