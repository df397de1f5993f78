// Internal layout shared by the host builder and the gfx950 kernels.
// See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/sealfm.h"

// The BWT is held as a 16-ary ("hex") wavelet matrix: level k stores, for every position of the
// order reached after k stable 16-way partitions, the 4-bit digit (c >> 4*(dlevels-1-k)) & 15 of the
// symbol sitting there.  One level = nblk blocks of 128 bytes (ONE L2 / HBM line) = 32 dwords; block
// b covers the 128 positions [128 b, 128 b + 128):
//   dword  d (d = 0..15)        : digits equal to d in this level in [superblock start, 128 b)
//   dwords 16+4j .. 19+4j       : bit plane P_j (128 bits): bit i = bit j of the digit at position 128 b + i
// and superblock s = b >> sb_shift has one row of sixteen 64-bit words in a side table,
//   sbase[k][s][d] = (positions of level k whose digit is < d) + (digits equal to d before the superblock)
// so that ONE 128-byte line (plus an L2-resident table row) answers "where does position p go on
// level k+1" for all sixteen digits: one probe moves a backward search or an interval-symbols node
// FOUR symbol bits down (BART's 16-bit alphabet = 4 dependent probes).  Texts below 2^32 symbols
// have a single superblock per level (sb_shift = 40: the row is just dbase[k][], kept in kernel
// arguments); longer ones use 2^13 blocks = 2^20 positions per superblock so that the in-block
// counters stay 32-bit.  Every byte of the line is payload or counter: 1 byte per symbol per level.
static constexpr uint32_t FMI_BLOCK_WORDS = 16;  // u64 words
static constexpr uint32_t FMI_BLOCK_BYTES = 128;
static constexpr uint32_t FMI_BLOCK_BITS = 128;  // positions per block
static constexpr uint32_t FMI_BLOCK_SHIFT = 7;
static constexpr uint32_t FMI_SB_SHIFT = 13;     // blocks per superblock (log2) when superblocks are in use
static constexpr uint32_t FMI_SB_NONE = 40;      // sb_shift of a single-superblock index
static constexpr uint32_t FMI_DIGIT_BITS = 4;
static constexpr uint32_t FMI_ARITY = 16;
static constexpr uint32_t FMI_MAX_LEVELS = 17;   // symbols < 2^17 (BART: 50274 < 2^16); node prefixes fit 16 bits
static constexpr uint32_t FMI_MAX_DLEVELS = (FMI_MAX_LEVELS + FMI_DIGIT_BITS - 1) / FMI_DIGIT_BITS;   // 5
static constexpr uint64_t FMI_MAX_N = 1ull << 40;

struct FmiDev {
    const uint64_t *wm;       // [dlevels][nblk][16]
    uint64_t nblk;
    uint64_t n;               // text length incl. sentinel
    uint64_t max_sym;
    uint32_t levels;          // bits per symbol = sdsl's wt_int depth (bits::hi(max)+1); quirk table only
    uint32_t dlevels;         // ceil(levels / 4) digit levels
    uint32_t sym_bytes;       // 2 or 4: width of text[]
    uint64_t dbase[FMI_MAX_DLEVELS][FMI_ARITY];  // [k][d] = positions of level k whose digit is < d (dbase[k][0] = 0)
    const uint64_t *sbase;    // [dlevels][nsb][16] superblock rows (see above); nsb == 1: equals dbase
    uint64_t nsb;             // superblocks per level
    uint32_t sb_shift;        // blocks per superblock (log2), FMI_SB_NONE when nsb == 1
    const uint64_t *C;        // [max_sym+2] number of symbols < c
    const uint64_t *leaf;     // [max_sym+1] start of c's run after the last level
    const uint8_t *q1;        // [max_sym+1] sdsl rank(size()+1, c) - occ(c)  (quirk Q1)
    const uint32_t *sa_lo;    // [n] low 32 bits of SA
    const uint8_t *sa_hi;     // [n] bits 32..39 of SA, or nullptr when n <= 2^32
    const void *text;         // [n] the (reversed-doc) text itself, sym_bytes wide
    const uint64_t *doc_begin;// [n_begin] cumulative doc offsets (index.py beginnings)
    uint64_t n_begin;
    // sampled position -> document table (SURVEY.md 8d "doc binning: 8 B with a sampled pos->doc table"): doc_hint[b] =
    // bisect_right(doc_begin, b << FMI_DOC_HINT_SHIFT) - 1, so the document of a position in block b lies in
    // [doc_hint[b], doc_hint[b + 1]] and the bisect over the 168 MB boundary array shrinks to 0..2 steps; nullptr: full bisect
    const uint32_t *doc_hint;
};
static constexpr uint32_t FMI_DOC_HINT_SHIFT = 7;

// Launch-shape options of the fmi_dev_* constraint / top-2K / aggregation calls: the built-in choice of each, changed per handle with
// fmi_dev_set_option(h, name, value) (-1 = back to the built-in choice) -- how the GPU tests run every kernel path on the same index and how
// tools/ compares two forms on one box.  Not read from the environment (round 6).
struct FmiOptions {
    int64_t constrain_waves = -1;   // 1: one self-contained wave per (row, top digit) instead of workgroups of 8 waves (any alphabet depth takes this form)
    int64_t leave_early = 1;        // 0: the waves of empty items stay in their workgroup
    int64_t row_first = -1;         // 0 / 1: never / always the row-first pair of launches (default: by prefix length)
    int64_t row_first_from = -1;    // prefix length from which a call goes row-first (default 3; 2 from 512 rows on)
    int64_t prefix_tables = 1;      // 0: the first constrained step of a decode through the generic expansion
    int64_t table_grid = -1;        // workgroups of k_constrain_table (default 1024: one resident round)
    int64_t topk_narrow = -1;       // rows of more than this many allowed tokens take the wide-row path of k_row_pick (default 1024)
    int64_t topk_legacy = 0;        // 1: wide rows skip the thread-maxima bound (the exact radix select: the tests' third path)
    int64_t chain_steps = 1;        // 0: fmi_dev_beam_step leaves the rows' chains to the next call (k_constrain_rows) instead of k_beam_advance
    int64_t advance_apart = 1;      // measurement passes: k_beam_advance as two launches (bookkeeping, chains); 0: the product's one launch, timed whole
    int64_t agg_rank_by_sorts = 0;  // tests: fmi_dev_aggregate ranks the first stage 1: with the three full stable sorts of rounds 2-5 (the checker of the selection), 2: with the single-workgroup selection (k_select_top)
    int64_t pt_inject_failure = 0;  // tests: building a prefix table fails after its first allocation (the call must take the generic path)
    FmiOptions();
    int set(const char *name, int64_t value);      // 0, or -1 for an unknown name
};

// Leaf-level node table of one forced prefix P: for every token t, the nodes of the LAST digit level of the
// wavelet matrix below the interval of P + [t] (what the expansion of that interval reaches after its dependent upper levels), and the
// interval itself.  The first constrained step of a decode -- every row's prefix is P + one token -- then reads its rows' node lists
// (contiguous) and streams the leaf level from one flat, evenly cut list: no ramp, no tail (k_constrain_table).
struct FmiPrefixTable {
    std::vector<int64_t> force;
    int64_t shift = 0;
    uint64_t vocab = 0;
    uint64_t *d_off = nullptr;    // [vocab + 1] first node of token t
    uint64_t *d_root = nullptr;   // [vocab][2]   inclusive range (l, r) of P + [t]; (1, 0): empty
    void *d_nodes = nullptr;      // uint4[n_nodes]: two 40-bit positions + 16-bit symbol prefix (pack_node), grouped by token
    uint64_t n_nodes = 0;
    bool ok = false;              // false: not built (too large): the generic path serves such rows
};

struct fmi {
    FmiOptions opt;
    std::vector<FmiPrefixTable> prefix_tables;
    // geometry
    uint64_t n = 0, max_sym = 0, sigma = 0, nblk = 0;
    uint32_t levels = 0, dlevels = 0, sym_bytes = 2, sb_shift = FMI_SB_NONE;
    uint64_t nsb = 1;
    // host-resident arrays (empty when built on device without keep_host)
    std::vector<uint64_t> wm, dbase /* [dlevels][16] */, sbase /* [dlevels][nsb][16] */, C, leaf, doc_begin;
    std::vector<uint8_t> q1, sa_hi;
    std::vector<uint32_t> sa_lo;
    std::vector<uint32_t> bwt;   // kept for tests / hand-over only (not uploaded)
    std::vector<uint8_t> text;   // n * sym_bytes
    uint64_t max_doc_len = 0;    // longest document (fmi_set_doc_beginnings): sizes the LDS of the scoring kernel
    bool host_resident = false;
    // device
    int device = -1;
    FmiDev dev{};
    std::vector<void *> dev_allocs;
    uint64_t dev_bytes = 0;
    // workspace for fmi_dev_* (sized by fmi_dev_reserve)
    uint64_t ws_rows = 0;
    uint64_t ws_seq = 0;      // parity picks the workspace bitmap of the next constraint call
    uint64_t ws_dirty[2] = {0, 0};   // words of each workspace bitmap that its last use may have left non-zero
    // the two symbol-space bitmaps of the table calls (k_constrain_table / k_table_bits, fmi_kernels.hip), one allocation
    void *sym_bits = nullptr;
    uint64_t sym_rows = 0, sym_row_words = 0, sym_dirty_rows[2] = {0, 0};
    int sym_flip = 0;
    // incremental constraint state of fmi_dev_constrained_topk_step (per-row prefix ranges of the last call)
    uint64_t state_tag = 0, state_rows = 0, state_len = 0;
    int state_flip = 0;
    // chained constraint calls (fmi_dev_beam_step): k_beam_advance of step t leaves the ranges / classes / root splits of the rows of
    // step t + 1 in the workspace; chain_* say for which call they are valid, state_base = index of the current call's row 0 in them
    uint64_t chain_tag = 0, chain_len = 0, chain_rows = 0, state_base = 0;
    const uint32_t *last_bits = nullptr;      // the bitmap the last constraint call filled (fmi_dev_last_constraint_bits)
    uint64_t last_bits_rows = 0, last_bits_wpr = 0;
    void *ws = nullptr;
    uint64_t ws_bytes = 0;
    void *ws_list = nullptr;          // list mode of the chained steps: [2][rows][64] positions, [2][rows][64] symbols, [2][rows] lengths
    int bits_prefilled = 0;           // k_beam_advance has put list rows' tokens into the workspace bitmap the next call will fill
    uint64_t *d_probe_counter = nullptr;
    uint64_t *dbg_tstamp = nullptr;   // tools: per-wave realtime stamps of k_constrain (fmi_dev_debug_timestamps)
    uint64_t dbg_tstamp_cap = 0;
    uint32_t *dbg_marks = nullptr;    // tools: host-visible progress words (fmi_dev_debug_marks); [1] = last finished stage of fmi_dev_aggregate
    int probe_count_enabled = 0;
    // optional event timing of k_constrain launches
    int timing_enabled = 0;
    std::vector<void *> ev_start, ev_stop;   // hipEvent_t
    uint64_t ev_used = 0;
    // per-call log of the constraint calls (fmi_dev_call_log): which call (prefix length, rows, launch form), its event time
    // (timing mode) or the blocks its launches loaded (counting mode: read back after every call)
    struct CallRec { uint32_t cur_len, rows, kind; int64_t ev; uint64_t blocks; };
    std::vector<CallRec> call_log;
    int call_log_enabled = 0;
    uint64_t probe_accum[4] = {0, 0, 0, 0};      // counters drained per call, not yet handed to fmi_dev_read_probe_count / _expand_stats
    // stage timing of fmi_dev_aggregate (fmi_dev_agg_timing): per call FMI_AGG_STAGES + 1 events on the call's stream, the located
    // rows / document entries / scored documents of the call (a measurement pass: the entry count is read back after the call)
    int agg_timing_enabled = 0;
    std::vector<void *> agg_events;             // hipEvent_t, (FMI_AGG_STAGES + 1) per logged call
    struct AggCall { uint64_t rows, entries, docs_scored, docs_kept, doc_tokens; };
    std::vector<AggCall> agg_calls;
    // non-blocking stream of the host-buffer API (fmi_<op>): its copies/kernels neither wait for nor
    // stall the caller's (torch's) streams; index arrays are immutable so there is nothing to order
    void *service_stream = nullptr;
};

void fmi_set_error(const char *fmt, ...);
// blocks per superblock (log2) for a text of n symbols; SEALFM_FORCE_SB=<shift> forces superblocks (tests)
uint32_t fmi_sb_shift_for(uint64_t n);

// host builder pieces (fmi_host.cpp)
int fmi_host_build_from_symbols(fmi *h, const uint32_t *text_with_sentinel, uint64_t n);
void fmi_host_suffix_array(const uint32_t *text, uint64_t n, uint32_t bits_per_sym, std::vector<uint64_t> &sa);
void fmi_host_finish_from_bwt(fmi *h, const uint32_t *bwt, uint64_t n);
void fmi_host_q1_table(const uint32_t *bwt, uint64_t n, uint32_t levels, uint64_t max_sym,
                       const std::vector<uint64_t> &C, std::vector<uint8_t> &q1);
void fmi_host_q1_from_first_pos(const std::vector<uint64_t> &first_pos, uint32_t levels, uint64_t max_sym,
                                const std::vector<uint64_t> &C, std::vector<uint8_t> &q1);
int fmi_upload(fmi *h, int device);
void fmi_release_device(fmi *h);
