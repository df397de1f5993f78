// Internal layout shared by the host builder and the gfx950 kernels.
// See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/sealfm.h"

// One wavelet-matrix level = nblk blocks of 64 bytes:
//   word 0      : number of 1 bits in this level before the block (absolute)
//   words 1..7  : 448 payload bits, bit p of the level lives in block p/448,
//                 word 1 + (p%448)/64, bit p%64
// so that one rank probe touches exactly one 64-byte line.
static constexpr uint32_t FMI_BLOCK_WORDS = 8;
static constexpr uint32_t FMI_BLOCK_BITS = 448;
static constexpr uint32_t FMI_MAX_LEVELS = 17;   // symbols < 2^17 (BART: 50274 < 2^16); node prefixes fit 16 bits

struct FmiDev {
    const uint64_t *wm;       // [levels][nblk][8]
    uint64_t nblk;
    uint64_t n;               // text length incl. sentinel
    uint64_t max_sym;
    uint32_t levels;
    uint32_t sym_bytes;       // 2 or 4: width of text[]
    uint64_t zeros[FMI_MAX_LEVELS];  // zeros per level
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
    uint32_t levels = 0, sym_bytes = 2;
    // host-resident arrays (empty when built on device without keep_host)
    std::vector<uint64_t> wm, zeros, C, leaf, doc_begin;
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
