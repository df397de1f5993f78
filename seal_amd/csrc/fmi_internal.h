// Internal layout shared by the host builder and the gfx950 kernels.
// See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/sealfm.h"

// The BWT is held as a 4-ary ("quad") wavelet matrix: quad level q stores, for every position of
// the order reached after q stable 4-way partitions, the 2-bit digit (c >> 2*(qlevels-1-q)) & 3 of the
// symbol sitting there.  One quad level = nblk blocks of 64 bytes (ONE memory sector) = 4 chunks of
// 16 bytes; block b covers positions [192 b, 192 b + 192) in 3 groups of 64:
//   chunk 0 : group 0      chunk 1 : header      chunk 2 : group 1      chunk 3 : group 2
//   group chunk = { H, L }: bit i of H / L = high / low bit of the digit at position 64*group + i
//   header      = c1, c2, c3 = digits equal to 1 / 2 / 3 in this level before position 192 b + 64
//                 (the boundary between groups 0 and 1), 40 bits each:
//                 w0 = c1 | c2 << 40 (low 24 bits of c2);  w1 = c2 >> 24 | c3 << 16
// rank_d(p) counts from the header towards p: backwards through group 0 when p lies there, forwards
// through groups 1..g otherwise -- the header plus one or two group chunks (2.33 of the 4 chunks on
// average) of ONE 64-byte sector answer rank_d(p) for all four digits d, and one probe moves a
// backward search or an interval-symbols node two symbol bits down.  Positions are < 2^40 (FMI_MAX_N).
static constexpr uint32_t FMI_BLOCK_WORDS = 8;
static constexpr uint32_t FMI_BLOCK_BYTES = 64;
static constexpr uint32_t FMI_BLOCK_BITS = 192;  // positions per block
static constexpr uint32_t FMI_BLOCK_MID = 64;    // header counts refer to this offset inside the block
static constexpr uint32_t FMI_MAX_LEVELS = 17;   // symbols < 2^17 (BART: 50274 < 2^16); node prefixes fit 16 bits
static constexpr uint32_t FMI_MAX_QLEVELS = (FMI_MAX_LEVELS + 1) / 2;
static constexpr uint64_t FMI_MAX_N = 1ull << 40;

struct FmiDev {
    const uint64_t *wm;       // [qlevels][nblk][8]
    uint64_t nblk;
    uint64_t n;               // text length incl. sentinel
    uint64_t max_sym;
    uint32_t levels;          // bits per symbol = sdsl's wt_int depth (bits::hi(max)+1); quirk table only
    uint32_t qlevels;         // (levels + 1) / 2 quad levels
    uint32_t sym_bytes;       // 2 or 4: width of text[]
    uint64_t qbase[FMI_MAX_QLEVELS][4];  // [q][d] = positions of level q whose digit is < d (qbase[q][0] = 0)
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
    uint32_t levels = 0, qlevels = 0, sym_bytes = 2;
    // host-resident arrays (empty when built on device without keep_host)
    std::vector<uint64_t> wm, qbase /* [qlevels][4] */, C, leaf, doc_begin;
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
