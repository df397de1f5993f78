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
// b covers the 64 positions [64 b, 64 b + 64):
//   dword  d (d = 0..15)      : low 32 bits of c_d = digits equal to d in this level before position 64 b
//   dwords 16..19, byte d     : bits 32..39 of c_d
//   dwords 20..27 (4 x u64)   : bit planes P0..P3: bit i of P_j = bit j of the digit at position 64 b + i
//   dwords 28..31             : zero
// so ONE 128-byte line answers rank_d(p) for all sixteen digits d: one probe moves a backward search
// or an interval-symbols node FOUR symbol bits down (BART's 16-bit alphabet = 4 dependent probes).
// The structure costs 2 bytes per BWT symbol per level (8 n bytes at 4
// levels): bytes are cheap in 288 GB of HBM, dependent random requests are not.  Positions are < 2^40.
static constexpr uint32_t FMI_BLOCK_WORDS = 16;  // u64 words
static constexpr uint32_t FMI_BLOCK_BYTES = 128;
static constexpr uint32_t FMI_BLOCK_BITS = 64;   // positions per block
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
    const uint64_t *dbase_tab; // the same table in HBM, for per-lane digits
    const uint64_t *C;        // [max_sym+2] number of symbols < c
    const uint64_t *leaf;     // [max_sym+1] start of c's run after the last level
    const uint8_t *q1;        // [max_sym+1] sdsl rank(size()+1, c) - occ(c)  (quirk Q1)
    const uint32_t *sa_lo;    // [n] low 32 bits of SA
    const uint8_t *sa_hi;     // [n] bits 32..39 of SA, or nullptr when n <= 2^32
    const void *text;         // [n] the (reversed-doc) text itself, sym_bytes wide
    const uint64_t *doc_begin;// [n_begin] cumulative doc offsets (index.py beginnings)
    uint64_t n_begin;
};

struct fmi {
    // geometry
    uint64_t n = 0, max_sym = 0, sigma = 0, nblk = 0;
    uint32_t levels = 0, dlevels = 0, sym_bytes = 2;
    // host-resident arrays (empty when built on device without keep_host)
    std::vector<uint64_t> wm, dbase /* [dlevels][16] */, C, leaf, doc_begin;
    std::vector<uint8_t> q1, sa_hi;
    std::vector<uint32_t> sa_lo;
    std::vector<uint32_t> bwt;   // kept for tests / hand-over only (not uploaded)
    std::vector<uint8_t> text;   // n * sym_bytes
    bool host_resident = false;
    // device
    int device = -1;
    FmiDev dev{};
    std::vector<void *> dev_allocs;
    uint64_t dev_bytes = 0;
    // workspace for fmi_dev_* (sized by fmi_dev_reserve)
    uint64_t ws_rows = 0;
    uint64_t ws_seq = 0;      // parity picks the queue-counter pair of the next fused constraint call
    void *ws = nullptr;
    uint64_t ws_bytes = 0;
    uint64_t *d_probe_counter = nullptr;
    int probe_count_enabled = 0;
    // optional event timing of k_expand launches
    int timing_enabled = 0;
    std::vector<void *> ev_start, ev_stop;   // hipEvent_t
    uint64_t ev_used = 0;
    // non-blocking stream of the host-buffer API (fmi_<op>): its copies/kernels neither wait for nor
    // stall the caller's (torch's) streams; index arrays are immutable so there is nothing to order
    void *service_stream = nullptr;
};

void fmi_set_error(const char *fmt, ...);

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
