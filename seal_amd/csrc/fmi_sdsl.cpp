// Reader for the index files the reference writes: sdsl-lite's serialisation of csa_wt_int<> (reference
// seal/cpp_modules/fm_index.cpp:186-199 store_to_file / load_from_file, seal/index.py:186-204 "<path>.fmi"), so that the
// published SEAL indices (README.md:67-69) load into this engine.
//
// sdsl-lite is an un-vendored submodule of the reference (absent from this image), so the layout below is restated from
// its published v2.1.x sources and is PARITY UNPINNED: it round-trips with the writer in oracle/fm_oracle.c
// (orc_save_sdsl, same recollection), it has not yet seen a file sdsl itself wrote.  Every structure is bounds- and
// consistency-checked (sizes that must agree, the whole file consumed, exactly one sentinel in the recovered text), so a
// file laid out differently is REFUSED with a message instead of loading wrong.
//
//   csa_wt::serialize        = wavelet_tree, sa_sample, isa_sample, alphabet
//   wt_int::serialize        = u64 size, u64 sigma, bit_vector tree (level-concatenated, size * max_level bits),
//                              rank_support_v (int_vector<64>), select_support_mcl<1>, select_support_mcl<0>, u32 max_level
//   sa_order_sa_sampling<>   = int_vector<0>: SA[i] for i % 32 == 0;   isa_sampling<> = int_vector<0>: ISA[j] for j % 64 == 0
//   int_alphabet<>           = sd_vector m_char (u64 size, u8 wl, int_vector<0> low, bit_vector high, two select_support_mcl),
//                              its rank/select supports (no data), int_vector<0> m_C, u64 sigma
//   int_vector<w>            = u64 size in bits (+ u8 width if w == 0), ceil(size / 64) words
//
// What is used: the tree (to walk LF), the ISA samples (n/64 independent starting rows), the alphabet (C).  The text is
// recovered with one 64-step LF walk per ISA sample, in parallel; the index proper (suffix array, BWT, wavelet matrix,
// tables) is then built by the engine's own builder from that text -- the suffix array of a text is unique, so this is
// the same index.  The file's REAL bit layout is additionally used for quirk Q1: the table
// rank(size()+1, c) - occ(c) is computed from the file's own tree with sdsl's rank loop and replaces the builder's
// analytic model of it.
#include <cstdlib>
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "fmi_internal.h"

extern "C" int fmi_build(fmi_t *h, const uint64_t *data, uint64_t n_data, int device);
extern "C" int fmi_build_device(fmi_t *h, const uint32_t *d_data, uint64_t n_data, int device, int keep_host);
extern "C" int fmi_create(fmi_t **out);
extern "C" void fmi_free(fmi_t *h);

namespace {

struct Cursor {
    const uint8_t *p, *end;
    bool ok = true;
    uint64_t u64() { uint64_t v = 0; if (end - p < 8) { ok = false; return 0; } memcpy(&v, p, 8); p += 8; return v; }
    uint32_t u32() { uint32_t v = 0; if (end - p < 4) { ok = false; return 0; } memcpy(&v, p, 4); p += 4; return v; }
    uint8_t u8() { if (end - p < 1) { ok = false; return 0; } return *p++; }
    const uint64_t *words(uint64_t nbits)
    {
        const uint64_t bytes = ((nbits + 63) / 64) * 8;
        if ((uint64_t)(end - p) < bytes || nbits > (1ull << 46)) { ok = false; return nullptr; }
        const uint64_t *w = (const uint64_t *)p;
        p += bytes;
        return w;
    }
};

struct IntVec {            // int_vector<0> / int_vector<64> / bit_vector view into the mapped file
    uint64_t bits = 0;
    uint8_t width = 1;
    const uint64_t *w = nullptr;
    uint64_t size() const { return width ? bits / width : 0; }
    uint64_t get(uint64_t i) const
    {
        const uint64_t pos = i * width;
        uint64_t v = w[pos >> 6] >> (pos & 63);
        if ((pos & 63) + width > 64) v |= w[(pos >> 6) + 1] << (64 - (pos & 63));
        return width < 64 ? v & ((1ull << width) - 1) : v;
    }
};

bool read_iv0(Cursor &c, IntVec &v) { v.bits = c.u64(); v.width = c.u8(); if (v.width == 0 || v.width > 64) c.ok = false; v.w = c.ok ? c.words(v.bits) : nullptr; return c.ok; }
bool read_bv(Cursor &c, IntVec &v) { v.bits = c.u64(); v.width = 1; v.w = c.words(v.bits); return c.ok; }

// select_support_mcl<b,1>: nothing of it is needed, but it must be stepped over exactly
bool skip_select(Cursor &c)
{
    const uint64_t cnt = c.u64();
    if (!c.ok) return false;
    if (!cnt) return true;
    IntVec sup, mol, blk;
    if (!read_iv0(c, sup) || !read_bv(c, mol)) return false;
    const uint64_t sb = (cnt + 4095) >> 12;
    if (sup.size() != sb || (mol.bits != 0 && mol.bits != sb)) { c.ok = false; return false; }
    for (uint64_t i = 0; i < sb; i++) if (!read_iv0(c, blk)) return false;
    return true;
}

struct Tree {              // wt_int's level-concatenated tree with a rank directory of our own (one count per 512 bits)
    uint64_t n = 0, bits = 0;
    uint32_t levels = 0;
    const uint64_t *w = nullptr;
    uint64_t nwords = 0;
    std::vector<uint64_t> blk;      // ones before every 512-bit block
    void index()
    {
        nwords = (bits + 63) / 64;
        blk.assign(nwords / 8 + 2, 0);
        uint64_t acc = 0;
        for (uint64_t i = 0; i < nwords; i++) { if ((i & 7) == 0) blk[i >> 3] = acc; acc += (uint64_t)__builtin_popcountll(w[i]); }
        for (uint64_t b = (nwords + 7) / 8; b < blk.size(); b++) blk[b] = acc;
    }
    uint64_t word(uint64_t i) const { return i < nwords ? w[i] : 0; }          // past the end reads as zero bits
    int bit(uint64_t idx) const { return (int)((word(idx >> 6) >> (idx & 63)) & 1); }
    uint64_t rank(uint64_t idx) const                                           // ones in [0, idx)
    {
        uint64_t r = blk[std::min<uint64_t>(idx >> 9, blk.size() - 1)];
        for (uint64_t i = (idx >> 9) << 3; i < (idx >> 6); i++) r += (uint64_t)__builtin_popcountll(word(i));
        if (idx & 63) r += (uint64_t)__builtin_popcountll(word(idx >> 6) & ((1ull << (idx & 63)) - 1));
        return r;
    }
    // sdsl wt_int::inverse_select(i) -> (rank of the symbol at i among equal symbols before i, the symbol)
    void inverse_select(uint64_t i, uint64_t &rank_out, uint64_t &sym_out) const
    {
        uint64_t c = 0, offset = 0, node_size = n;
        for (uint32_t k = 0; k < levels; k++) {
            const uint64_t o0 = rank(offset), oi = rank(offset + i) - o0, oe = rank(offset + node_size) - o0;
            c <<= 1;
            if (bit(offset + i)) { offset += node_size - oe; node_size = oe; i = oi; c |= 1; }
            else { node_size -= oe; i -= oi; }
            offset += n;
        }
        rank_out = i; sym_out = c;
    }
    // sdsl wt_int::rank(i, c) as published, incl. what it returns for i == size() + 1 (quirk Q1)
    uint64_t rank_sym(uint64_t i, uint64_t c) const
    {
        if ((1ull << levels) <= c) return 0;
        uint64_t offset = 0, node_size = n, mask = 1ull << (levels - 1);
        for (uint32_t k = 0; k < levels && i; k++) {
            const uint64_t o0 = rank(offset), oi = rank(offset + i) - o0, oe = rank(offset + node_size) - o0;
            if (c & mask) { offset += node_size - oe; node_size = oe; i = oi; }
            else { node_size -= oe; i -= oi; }
            offset += n;
            mask >>= 1;
        }
        return i;
    }
};

struct Mapped {
    const uint8_t *p = nullptr;
    size_t len = 0;
    ~Mapped() { if (p) munmap((void *)p, len); }
};

}  // namespace

// 1 if the file does not start with this engine's own magic (the caller then tries the sdsl layout)
extern "C" int fmi_load_sdsl(fmi_t **out, const char *path, int device)
{
    if (!out || !path) { fmi_set_error("fmi_load_sdsl: null argument"); return FMI_ERR_ARG; }
    Mapped m;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) { fmi_set_error("cannot open %s", path); return FMI_ERR_IO; }
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 64) { close(fd); fmi_set_error("%s: too short for an sdsl csa_wt_int file", path); return FMI_ERR_IO; }
        m.len = (size_t)st.st_size;
        void *p = mmap(nullptr, m.len, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { fmi_set_error("mmap(%s) failed", path); return FMI_ERR_IO; }
        m.p = (const uint8_t *)p;
    }
    Cursor c{m.p, m.p + m.len};
#define REFUSE(msg) do { fmi_set_error("%s: not an sdsl csa_wt_int<> index as this reader knows the format (%s)", path, msg); return FMI_ERR_IO; } while (0)
    // ---- wt_int ----
    Tree t;
    t.n = c.u64();
    const uint64_t wt_sigma = c.u64();
    IntVec tree, rank_bb;
    if (!read_bv(c, tree)) REFUSE("wavelet tree bits");
    rank_bb.bits = c.u64(); rank_bb.width = 64; rank_bb.w = c.words(rank_bb.bits);
    if (!c.ok || !skip_select(c) || !skip_select(c)) REFUSE("rank/select supports of the wavelet tree");
    t.levels = c.u32();
    if (!c.ok || t.n < 2 || t.n >= FMI_MAX_N || t.levels == 0 || t.levels > FMI_MAX_LEVELS || tree.bits != t.n * t.levels) REFUSE("wavelet tree geometry");
    if (rank_bb.bits != ((((((tree.bits + 63) >> 6) << 6) >> 9) + 1) << 1) * 64) REFUSE("rank_support_v size");
    t.bits = tree.bits; t.w = tree.w;
    // ---- samples ----
    IntVec sa_s, isa_s;
    if (!read_iv0(c, sa_s) || !read_iv0(c, isa_s)) REFUSE("SA / ISA samples");
    if (sa_s.size() != (t.n + 31) / 32 || isa_s.size() != (t.n + 63) / 64) REFUSE("sample counts (SA/32, ISA/64 expected)");
    // ---- int_alphabet ----
    const uint64_t char_size = c.u64();
    const uint8_t wl = c.u8();
    IntVec low, high, Cv;
    if (!c.ok) REFUSE("alphabet header");
    low.bits = c.u64(); low.width = c.u8(); low.w = c.ok ? c.words(low.bits) : nullptr;     // width may be 0 when wl == 0
    if (!c.ok || low.width != wl) REFUSE("alphabet low bits");
    if (!read_bv(c, high) || !skip_select(c) || !skip_select(c) || !read_iv0(c, Cv)) REFUSE("alphabet vectors");
    const uint64_t sigma = c.u64();
    if (!c.ok || c.p != c.end) REFUSE("trailing or missing bytes");
    if (sigma != wt_sigma || Cv.size() != sigma + 1 || char_size == 0 || char_size > (1ull << FMI_MAX_LEVELS)) REFUSE("alphabet size");
    // sd_vector of the occurring symbols: `wl` low bits per entry + a unary-coded high part.  Everything the decode
    // indexes with is checked BEFORE it is used: a corrupt or crafted file must be refused, never read or written out of bounds.
    if (wl > 63 || low.width > 63) REFUSE("alphabet low-bit width");
    if (high.bits > 2 * (char_size + sigma) + 64) REFUSE("alphabet high-bit vector longer than its universe allows");
    std::vector<uint64_t> chars;                 // comp -> symbol
    {
        const uint64_t n_low = wl ? low.bits / wl : 0;           // entries the low-bit vector really holds
        const uint64_t low_words = (low.bits + 63) >> 6;
        uint64_t zeros = 0, i = 0;
        for (uint64_t b = 0; b < high.bits; b++) {
            if ((high.w[b >> 6] >> (b & 63)) & 1) {
                if (chars.size() >= sigma) REFUSE("alphabet lists more symbols than sigma");
                uint64_t lo = 0;
                if (wl) {
                    if (i >= n_low) REFUSE("alphabet high bits name more entries than the low-bit vector holds");
                    const uint64_t bit = i * wl, w0 = bit >> 6, sh = bit & 63;
                    lo = low.w[w0] >> sh;
                    if (sh + wl > 64) {
                        if (w0 + 1 >= low_words) REFUSE("alphabet low bits end inside an entry");
                        lo |= low.w[w0 + 1] << (64 - sh);
                    }
                    lo &= (1ull << wl) - 1;
                }
                if (zeros > (char_size >> wl)) REFUSE("alphabet symbol beyond the character range");
                const uint64_t sym = (zeros << wl) | lo;
                if (sym >= char_size || (!chars.empty() && sym <= chars.back())) REFUSE("alphabet symbols must ascend strictly below the character range");
                chars.push_back(sym);
                i++;
            } else zeros++;
        }
    }
    if (chars.size() != sigma || chars.empty() || chars[0] != 0 || chars.back() >= char_size || Cv.get(0) != 0 || Cv.get(sigma) != t.n || Cv.get(1) != 1)
        REFUSE("alphabet contents (sentinel 0 once, C[sigma] = size)");
    std::vector<uint64_t> C_of(char_size, 0);
    for (uint64_t k = 0; k < sigma; k++) C_of[chars[k]] = Cv.get(k);
    t.index();
    // ---- text: one LF walk of <= 64 steps per ISA sample (+ one from row 0, the sentinel's suffix, for the tail) ----
    const uint64_t n = t.n, ns = isa_s.size();
    std::vector<uint32_t> text(n);
    std::atomic<int> bad{0};
    auto walk = [&](uint64_t row, uint64_t pos, uint64_t stop) {      // fills text[stop .. pos-1] walking back from the suffix at pos
        while (pos > stop) {
            if (row >= n) { bad = 1; return; }
            uint64_t r, sym;
            t.inverse_select(row, r, sym);
            if (sym >= char_size) { bad = 1; return; }
            text[--pos] = (uint32_t)sym;
            row = C_of[sym] + r;
        }
    };
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (n < (1u << 16)) nt = 1;
    {
        std::vector<std::thread> th;
        for (unsigned ti = 0; ti < nt; ti++)
            th.emplace_back([&, ti] {
                for (uint64_t s = 1 + ti; s < ns; s += nt) walk(isa_s.get(s), 64 * s, 64 * (s - 1));
                if (ti == 0) { text[n - 1] = 0; walk(0, n - 1, 64 * (ns - 1)); }     // SA[0] = n - 1: the sentinel sorts first
            });
        for (auto &x : th) x.join();
    }
    uint64_t zeros = 0;
    for (uint64_t i = 0; i + 1 < n; i++) zeros += text[i] == 0;
    if (bad || zeros) REFUSE("the LF walks do not reproduce a text with one final sentinel");
    {   // the file says how often every symbol occurs (alphabet C): the recovered text must agree
        std::vector<uint64_t> hist(char_size, 0);
        for (uint64_t i = 0; i < n; i++) hist[text[i]]++;
        for (uint64_t k = 0; k < sigma; k++) {
            if (hist[chars[k]] != Cv.get(k + 1) - Cv.get(k)) REFUSE("symbol counts of the recovered text differ from the alphabet's C array");
            hist[chars[k]] = 0;
        }
        for (uint64_t v : hist) if (v) REFUSE("the recovered text holds symbols the alphabet does not list");
    }
    // ---- the index itself: the engine's builder over the recovered text ----
    fmi *h = nullptr;
    int rc = fmi_create(&h);
    if (rc) return rc;
    if (device >= 0) {
        // a host copy makes the loaded index saveable in the engine's own container (fmi_save), i.e. turns load + save
        // into a converter; it costs ~9.4 bytes per symbol of host memory, so beyond 2^28 symbols only on request
        const char *e_keep = getenv("SEALFM_KEEP_HOST");
        const int keep_host = e_keep ? atoi(e_keep) != 0 : n <= (1ull << 28);
        uint32_t *d = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&d, std::max<uint64_t>(n - 1, 1) * 4) != hipSuccess) { fmi_free(h); fmi_set_error("hipMalloc for the recovered text failed"); return FMI_ERR_HIP; }
        rc = hipMemcpy(d, text.data(), (n - 1) * 4, hipMemcpyHostToDevice) == hipSuccess ? fmi_build_device(h, d, n - 1, device, keep_host) : FMI_ERR_HIP;
        (void)hipFree(d);
    } else {
        std::vector<uint64_t> data(text.begin(), text.end() - 1);
        rc = fmi_build(h, data.data(), n - 1, -1);
    }
    if (rc) { fmi_free(h); return rc; }
    if (h->n != n || h->levels != t.levels) { fmi_free(h); REFUSE("rebuilt index disagrees with the file on size / levels"); }
    {   // the file's suffix-array samples (every 32nd row) against the rebuilt suffix array
        const uint64_t ns32 = sa_s.size();
        std::vector<uint32_t> lo32(ns32);
        std::vector<uint8_t> hi8(ns32, 0);
        if (h->device >= 0) {
            bool okc = hipMemcpy2D(lo32.data(), 4, h->dev.sa_lo, 32 * 4, 4, ns32, hipMemcpyDeviceToHost) == hipSuccess;
            if (okc && h->dev.sa_hi) okc = hipMemcpy2D(hi8.data(), 1, h->dev.sa_hi, 32, 1, ns32, hipMemcpyDeviceToHost) == hipSuccess;
            if (!okc) { fmi_free(h); fmi_set_error("reading back the suffix-array samples failed"); return FMI_ERR_HIP; }
        } else {
            for (uint64_t i = 0; i < ns32; i++) { lo32[i] = h->sa_lo[32 * i]; if (!h->sa_hi.empty()) hi8[i] = h->sa_hi[32 * i]; }
        }
        for (uint64_t i = 0; i < ns32; i++)
            if ((((uint64_t)hi8[i] << 32) | lo32[i]) != sa_s.get(i)) { fmi_free(h); REFUSE("suffix-array samples differ from the suffix array of the recovered text"); }
    }
    // ---- quirk Q1 from the file's own layout: rank(size() + 1, c) - occ(c) with sdsl's loop over the REAL tree ----
    std::vector<uint8_t> q1(h->max_sym + 1, 0);
    for (uint64_t k = 0; k < sigma; k++) {
        const uint64_t sym = chars[k], occ = Cv.get(k + 1) - Cv.get(k);
        const uint64_t r = t.rank_sym(n + 1, sym);
        q1[sym] = (uint8_t)(r > occ ? 1 : 0);
    }
    if (h->device >= 0 && h->dev.q1) {
        if (hipMemcpy((void *)h->dev.q1, q1.data(), q1.size(), hipMemcpyHostToDevice) != hipSuccess) { fmi_free(h); fmi_set_error("hipMemcpy(q1) failed"); return FMI_ERR_HIP; }
    }
    if (h->host_resident) h->q1 = q1;
    *out = h;
    return FMI_OK;
#undef REFUSE
}
