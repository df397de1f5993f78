// Device-side index construction for corpora the host builder cannot reach
// (NQ: ~2.9e9 symbols).  Produces byte-identical arrays to fmi_host.cpp:
//   suffix array  : prefix doubling; round 0 sorts 64-bit keys packing the first
//                   64/L symbols, round r sorts (rank[i], rank[i+h]) composites,
//                   all with rocPRIM's stable LSD radix sort (double buffered);
//   BWT           : gather;
//   wavelet matrix: per level, one wave per 128-position block builds the 4 bit
//                   words with __ballot, a device scan fills the block counters,
//                   and the stable zero/one partition is a scatter whose
//                   destination is the rank on the level just built;
//   tables        : leaf/occ from the run boundaries after the last partition,
//                   first BWT occurrence per symbol for the Q1 table.
// HBM budget at n = 2.9e9 (u32 indices): ~90 GB peak, all freed before return
// except the index itself.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "fmi_device.h"

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fmi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FMI_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

namespace {

struct Pool {   // frees everything still registered on scope exit
    std::vector<void *> ptrs;
    ~Pool() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <class T> hipError_t alloc(T **out, uint64_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<uint64_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
    void release(void *p) { for (auto &q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; } }
    void keep(void *p) { for (auto &q : ptrs) if (q == p) q = nullptr; }
};

constexpr unsigned TB = 256;
inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + TB - 1) / TB, 1u << 20); }
#define GRID_STRIDE(i, n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (uint64_t)gridDim.x * blockDim.x)

template <typename SymT>
__global__ void k_make_text(const uint32_t *data, uint64_t n, SymT *text)
{
    GRID_STRIDE(i, n) text[i] = (i + 1 < n) ? (SymT)data[i] : (SymT)0;   // sdsl appends the 0 sentinel
}

template <typename SymT, typename IdxT>
__global__ void k_pack_keys(const SymT *text, uint64_t n, uint32_t bits, uint32_t per, uint64_t *keys, IdxT *idx)
{
    GRID_STRIDE(i, n) {
        uint64_t key = 0;
        for (uint32_t j = 0; j < per; j++) {
            uint64_t s = (i + j < n) ? (uint64_t)text[i + j] : 0;
            key = (key << bits) | s;
        }
        keys[i] = key;
        idx[i] = (IdxT)i;
    }
}

// head[j] = j if sorted key j starts a new group else 0 (max-scanned into group starts)
template <typename IdxT>
__global__ void k_heads(const uint64_t *keys, uint64_t n, IdxT *gs)
{
    GRID_STRIDE(j, n) gs[j] = (j == 0 || keys[j] != keys[j - 1]) ? (IdxT)j : (IdxT)0;
}

// two-key form (n >= 2^32: the (rank, rank+h) pair no longer fits one 64-bit key)
template <typename IdxT>
__global__ void k_heads2(const IdxT *sa, const IdxT *rank, uint64_t n, uint64_t h, IdxT *gs)
{
    GRID_STRIDE(j, n) {
        bool head = (j == 0);
        if (!head) {
            const uint64_t p = sa[j], q = sa[j - 1];
            const uint64_t p2 = (p + h < n) ? (uint64_t)rank[p + h] : 0, q2 = (q + h < n) ? (uint64_t)rank[q + h] : 0;
            head = rank[p] != rank[q] || p2 != q2;
        }
        gs[j] = head ? (IdxT)j : (IdxT)0;
    }
}

template <typename IdxT>
__global__ void k_key_of(const IdxT *sa, const IdxT *rank, uint64_t n, uint64_t h, uint64_t *keys)
{
    GRID_STRIDE(j, n) { const uint64_t p = (uint64_t)sa[j] + h; keys[j] = (p < n) ? (uint64_t)rank[p] : 0; }
}

template <typename IdxT>
__global__ void k_scatter_rank(const IdxT *sa, const IdxT *gs, uint64_t n, IdxT *rank, unsigned long long *n_groups)
{
    unsigned long long local = 0;
    GRID_STRIDE(j, n) {
        rank[sa[j]] = gs[j];
        local += (gs[j] == j);
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_groups, local);
}

template <typename IdxT>
__global__ void k_pair_keys(const IdxT *sa, const IdxT *rank, uint64_t n, uint64_t h, uint32_t bits, uint64_t *keys)
{
    GRID_STRIDE(j, n) {
        const uint64_t p = sa[j];
        const uint64_t r2 = (p + h < n) ? rank[p + h] : 0;   // only reached by already-unique suffixes
        keys[j] = ((uint64_t)rank[p] << bits) | r2;
    }
}

template <typename SymT, typename IdxT>
__global__ void k_bwt(const SymT *text, const IdxT *sa, uint64_t n, SymT *bwt)
{
    GRID_STRIDE(j, n) { const uint64_t p = sa[j]; bwt[j] = text[p ? p - 1 : n - 1]; }
}

// one wave per 128-position block of one level: the four bit planes by ballot (two halves); the
// counters are filled in afterwards, one digit class at a time (k_class_count -> exclusive scan ->
// k_class_store)
template <typename SymT>
__global__ __launch_bounds__(256) void k_level_planes(const SymT *cur, uint64_t n, uint32_t sh, uint64_t nblk, uint64_t *lvl)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t b = wave; b < nblk; b += nwaves) {
        uint64_t P[4][2];
        for (uint32_t half = 0; half < 2; half++) {
            const uint64_t p = b * FMI_BLOCK_BITS + 64 * half + lane;
            const uint32_t d = p < n ? (uint32_t)(cur[p] >> sh) & (FMI_ARITY - 1) : 0u;
            P[0][half] = __ballot(d & 1); P[1][half] = __ballot(d & 2); P[2][half] = __ballot(d & 4); P[3][half] = __ballot(d & 8);
        }
        if (lane < 32) {      // the whole 128-byte block, coalesced: counters zero for now, planes at dwords 16..31
            uint32_t v = 0;
            if (lane >= 16) {
                const uint32_t j = (lane - 16) >> 2, w = lane & 3;
                const uint64_t q = j == 0 ? P[0][w >> 1] : (j == 1 ? P[1][w >> 1] : (j == 2 ? P[2][w >> 1] : P[3][w >> 1]));
                v = (w & 1) ? (uint32_t)(q >> 32) : (uint32_t)q;
            }
            reinterpret_cast<uint32_t *>(lvl + b * FMI_BLOCK_WORDS)[lane] = v;
        }
    }
}

// occurrences of digit d inside each block
__global__ void k_class_count(const uint64_t *lvl, uint64_t nblk, uint64_t n, uint32_t d, uint32_t *cnt)
{
    GRID_STRIDE(b, nblk) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(lvl + b * FMI_BLOCK_WORDS);
        const uint64_t begin = b * FMI_BLOCK_BITS;
        // positions past n carry digit 0 in the planes but are not part of the level
        const uint32_t valid = begin >= n ? 0u : (n - begin >= FMI_BLOCK_BITS ? FMI_BLOCK_BITS : (uint32_t)(n - begin));
        uint32_t c = 0;
        for (uint32_t x = 0; x < 4; x++) {
            uint32_t m = valid >= 32 * (x + 1) ? ~0u : (valid <= 32 * x ? 0u : ((1u << (valid - 32 * x)) - 1));
            m &= (d & 1) ? w[16 + x] : ~w[16 + x];
            m &= (d & 2) ? w[20 + x] : ~w[20 + x];
            m &= (d & 4) ? w[24 + x] : ~w[24 + x];
            m &= (d & 8) ? w[28 + x] : ~w[28 + x];
            c += (uint32_t)__popc(m);
        }
        cnt[b] = c;
    }
}

// counter of digit d of every block (relative to its superblock) and the superblock rows
__global__ void k_class_store(const uint64_t *excl, uint64_t nblk, uint32_t d, uint32_t sb_shift, uint64_t *lvl, uint64_t *sb_rows)
{
    GRID_STRIDE(b, nblk) {
        const uint64_t s = b >> sb_shift, first = s << sb_shift;
        reinterpret_cast<uint32_t *>(lvl + b * FMI_BLOCK_WORDS)[d] = (uint32_t)(excl[b] - excl[first]);
        if (b == first) sb_rows[s * FMI_ARITY + d] = excl[b];
    }
}

// sbase rows += dbase[k][d]
__global__ void k_add_dbase(uint64_t *sb_rows, uint64_t nsb, const uint64_t *db)
{
    GRID_STRIDE(i, nsb * FMI_ARITY) sb_rows[i] += db[i & (FMI_ARITY - 1)];
}

template <typename SymT>
__global__ void k_partition(FmiDev ix, uint32_t k, const SymT *cur, SymT *nxt)
{
    GRID_STRIDE(i, ix.n) {
        const SymT v = cur[i];
        nxt[wm_step(ix, k, i, wm_digit(ix, v, k))] = v;
    }
}

template <typename SymT>
__global__ void k_runs(const SymT *sorted, uint64_t n, uint64_t *leaf, uint64_t *occ_end)
{
    GRID_STRIDE(i, n) {
        const SymT v = sorted[i];
        if (i == 0 || sorted[i - 1] != v) leaf[v] = i;
        if (i + 1 == n || sorted[i + 1] != v) occ_end[v] = i + 1;
    }
}

template <typename SymT>
__global__ void k_first_pos(const SymT *bwt, uint64_t n, unsigned long long *first_pos)
{
    GRID_STRIDE(j, n) {
        const SymT v = bwt[j];
        if (j < first_pos[v]) atomicMin(&first_pos[v], (unsigned long long)j);
    }
}

__global__ void k_split_sa(const uint64_t *sa, uint64_t n, uint32_t *lo, uint8_t *hi)
{
    GRID_STRIDE(i, n) { lo[i] = (uint32_t)sa[i]; hi[i] = (uint8_t)(sa[i] >> 32); }
}

template <typename SymT>
__global__ void k_widen(const SymT *in, uint64_t n, uint32_t *out) { GRID_STRIDE(i, n) out[i] = in[i]; }

// BWT (device) -> wavelet matrix levels + C / leaf / q1 tables, on the device; fills the geometry
// fields of `h` and `d`.  Shared by the full builder and by fmi_build_from_bwt_device.
template <typename SymT>
int wavelet_from_bwt(fmi *h, Pool &pool, hipStream_t st, const SymT *bwt, uint64_t n, uint64_t max_sym, uint32_t L, FmiDev &d,
                     uint64_t **wm_out, uint64_t **dC_out, uint64_t **dleaf_out, uint8_t **dq1_out, SymT *consume_bwt = nullptr)
{
    // (consume_bwt: the pool-owned BWT buffer itself becomes the first level's input instead of a copy of it -- 2 B per symbol
    //  less at the peak of a build that is sized to the GPU; it is gone afterwards)
    unsigned long long *first_pos = nullptr;
    HIPCHK(pool.alloc(&first_pos, max_sym + 1));
    HIPCHK(hipMemsetAsync(first_pos, 0xff, (max_sym + 1) * 8, st));
    hipLaunchKernelGGL((k_first_pos<SymT>), dim3(grid_for(n)), dim3(TB), 0, st, bwt, n, first_pos);
    SymT *cur = nullptr, *nxt = nullptr;
    if (consume_bwt) cur = consume_bwt;
    else {
        HIPCHK(pool.alloc(&cur, n));
        HIPCHK(hipMemcpyAsync(cur, bwt, n * sizeof(SymT), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(pool.alloc(&nxt, n));
    const uint64_t nblk = n / FMI_BLOCK_BITS + 2;
    const uint32_t D = (L + FMI_DIGIT_BITS - 1) / FMI_DIGIT_BITS;
    const uint32_t sb_shift = fmi_sb_shift_for(n);
    const uint64_t nsb = (nblk >> sb_shift) + 1;
    uint64_t *wm = nullptr, *excl = nullptr, *sbase_dev = nullptr, *db_dev = nullptr;
    uint32_t *cnt = nullptr;
    HIPCHK(pool.alloc(&wm, (uint64_t)D * nblk * FMI_BLOCK_WORDS));
    HIPCHK(pool.alloc(&excl, nblk + 1)); HIPCHK(pool.alloc(&cnt, nblk + 1));
    HIPCHK(pool.alloc(&sbase_dev, (uint64_t)D * nsb * FMI_ARITY)); HIPCHK(pool.alloc(&db_dev, (uint64_t)FMI_ARITY));
    HIPCHK(hipMemsetAsync(cnt + nblk, 0, 4, st));
    size_t xs_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, xs_bytes, cnt, excl, (uint64_t)0, nblk + 1, rocprim::plus<uint64_t>(), st));
    void *xs_tmp = nullptr;
    HIPCHK(pool.alloc((char **)&xs_tmp, xs_bytes + 256));
    d = FmiDev{};
    d.wm = wm; d.nblk = nblk; d.n = n; d.max_sym = max_sym; d.levels = L; d.dlevels = D; d.sym_bytes = sizeof(SymT) == 2 ? 2 : 4;
    d.sbase = sbase_dev; d.nsb = nsb; d.sb_shift = sb_shift;
    std::vector<uint64_t> dbase((size_t)D * FMI_ARITY, 0);
    for (uint32_t k = 0; k < D; k++) {
        uint64_t *lvl = wm + (uint64_t)k * nblk * FMI_BLOCK_WORDS;
        uint64_t *sb_rows = sbase_dev + (uint64_t)k * nsb * FMI_ARITY;
        hipLaunchKernelGGL((k_level_planes<SymT>), dim3((unsigned)std::min<uint64_t>((nblk + 3) / 4, 1u << 20)), dim3(256), 0, st,
                           cur, n, FMI_DIGIT_BITS * (D - 1 - k), nblk, lvl);
        uint64_t tot[FMI_ARITY];
        for (uint32_t e = 0; e < FMI_ARITY; e++) {
            hipLaunchKernelGGL(k_class_count, dim3(grid_for(nblk)), dim3(TB), 0, st, (const uint64_t *)lvl, nblk, n, e, cnt);
            size_t xb = xs_bytes;
            HIPCHK(rocprim::exclusive_scan(xs_tmp, xb, cnt, excl, (uint64_t)0, nblk + 1, rocprim::plus<uint64_t>(), st));
            hipLaunchKernelGGL(k_class_store, dim3(grid_for(nblk)), dim3(TB), 0, st, (const uint64_t *)excl, nblk, e, sb_shift, lvl, sb_rows);
            HIPCHK(hipMemcpyAsync(&tot[e], excl + nblk, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));   // tot[e] lands in pageable memory; excl / cnt are reused by the next class
        }
        uint64_t *db = dbase.data() + (size_t)k * FMI_ARITY;
        db[0] = 0;
        for (uint32_t e = 1; e < FMI_ARITY; e++) db[e] = db[e - 1] + tot[e - 1];
        for (uint32_t e = 0; e < FMI_ARITY; e++) d.dbase[k][e] = db[e];
        HIPCHK(hipMemcpyAsync(db_dev, db, FMI_ARITY * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_add_dbase, dim3(grid_for(nsb * FMI_ARITY)), dim3(TB), 0, st, sb_rows, nsb, (const uint64_t *)db_dev);
        hipLaunchKernelGGL((k_partition<SymT>), dim3(grid_for(n)), dim3(TB), 0, st, d, k, cur, nxt);
        HIPCHK(hipStreamSynchronize(st));       // db is a host buffer
        std::swap(cur, nxt);
    }
    HIPCHK(hipGetLastError());

    // ---- per-symbol tables --------------------------------------------------
    uint64_t *leaf = nullptr, *occ_end = nullptr;
    HIPCHK(pool.alloc(&leaf, max_sym + 1)); HIPCHK(pool.alloc(&occ_end, max_sym + 1));
    HIPCHK(hipMemsetAsync(leaf, 0, (max_sym + 1) * 8, st));
    HIPCHK(hipMemsetAsync(occ_end, 0, (max_sym + 1) * 8, st));
    hipLaunchKernelGGL((k_runs<SymT>), dim3(grid_for(n)), dim3(TB), 0, st, cur, n, leaf, occ_end);
    std::vector<uint64_t> h_leaf(max_sym + 1), h_end(max_sym + 1), h_first(max_sym + 1);
    HIPCHK(hipMemcpyAsync(h_leaf.data(), leaf, (max_sym + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_end.data(), occ_end, (max_sym + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_first.data(), first_pos, (max_sym + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));

    h->n = n; h->max_sym = max_sym; h->levels = L; h->dlevels = D; h->nblk = nblk; h->sym_bytes = d.sym_bytes;
    h->dbase = dbase;
    h->sb_shift = sb_shift; h->nsb = nsb;
    h->sbase.resize((size_t)D * nsb * FMI_ARITY);
    HIPCHK(hipMemcpy(h->sbase.data(), sbase_dev, h->sbase.size() * 8, hipMemcpyDeviceToHost));
    h->leaf = h_leaf;
    h->C.assign(max_sym + 2, 0);
    uint64_t sigma = 0;
    for (uint64_t c = 0; c <= max_sym; c++) {
        const uint64_t occ = h_end[c] ? h_end[c] - h_leaf[c] : 0;
        if (occ) sigma++;
        h->C[c + 1] = h->C[c] + occ;
    }
    h->sigma = sigma;
    fmi_host_q1_from_first_pos(h_first, L, max_sym, h->C, h->q1);

    // small tables to the device
    uint64_t *dC = nullptr, *dleaf = leaf;
    uint8_t *dq1 = nullptr;
    HIPCHK(pool.alloc(&dC, max_sym + 2)); HIPCHK(pool.alloc(&dq1, max_sym + 1));
    HIPCHK(hipMemcpy(dC, h->C.data(), (max_sym + 2) * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dq1, h->q1.data(), max_sym + 1, hipMemcpyHostToDevice));

    pool.release(cur); pool.release(nxt); pool.release(xs_tmp); pool.release(excl); pool.release(cnt); pool.release(db_dev);
    pool.release(occ_end); pool.release(first_pos);
    *wm_out = wm; *dC_out = dC; *dleaf_out = dleaf; *dq1_out = dq1;
    return FMI_OK;
}

template <typename SymT, typename IdxT>
int build_impl(fmi *h, const uint32_t *d_data, uint64_t n_data, int device, int keep_host, uint64_t max_sym, uint32_t L)
{
    constexpr bool WIDE = sizeof(IdxT) == 8;
    const uint64_t n = n_data + 1;
    Pool pool;
    hipStream_t st = 0;
    SymT *text = nullptr;
    HIPCHK(pool.alloc(&text, n));
    hipLaunchKernelGGL((k_make_text<SymT>), dim3(grid_for(n)), dim3(TB), 0, st, d_data, n, text);

    // ---- suffix array -----------------------------------------------------
    uint64_t *keyA = nullptr, *keyB = nullptr;
    IdxT *idxA = nullptr, *idxB = nullptr, *rank = nullptr;
    unsigned long long *d_groups = nullptr;
    HIPCHK(pool.alloc(&keyA, n)); HIPCHK(pool.alloc(&keyB, n));
    HIPCHK(pool.alloc(&idxA, n)); HIPCHK(pool.alloc(&idxB, n));
    HIPCHK(pool.alloc(&rank, n)); HIPCHK(pool.alloc(&d_groups, 1));
    const uint32_t per = std::max<uint32_t>(1, 64 / L);
    uint32_t nbits = 1;
    while ((n >> nbits) > 0) nbits++;           // bits to hold a rank < n
    hipLaunchKernelGGL((k_pack_keys<SymT, IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, text, n, L, per, keyA, idxA);
    rocprim::double_buffer<uint64_t> dk(keyA, keyB);
    rocprim::double_buffer<IdxT> dv(idxA, idxB);
    size_t tmp_bytes = 0, scan_bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, n, 0u, 64u, st));
    HIPCHK(rocprim::inclusive_scan(nullptr, scan_bytes, (IdxT *)nullptr, (IdxT *)nullptr, n, rocprim::maximum<IdxT>(), st));
    void *tmp = nullptr;
    HIPCHK(pool.alloc((char **)&tmp, std::max(tmp_bytes, scan_bytes) + 256));
    size_t tb = tmp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, 0u, std::min<uint32_t>(64u, per * L), st));
    uint64_t h_step = per;
    bool first = true;
    for (int round = 0;; round++) {
        IdxT *sa = dv.current();
        IdxT *gs = dv.alternate();           // free until the next sort
        if (first || !WIDE) hipLaunchKernelGGL((k_heads<IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, dk.current(), n, gs);
        else hipLaunchKernelGGL((k_heads2<IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, sa, rank, n, h_step / 2, gs);
        first = false;
        size_t sb = scan_bytes;
        HIPCHK(rocprim::inclusive_scan(tmp, sb, gs, gs, n, rocprim::maximum<IdxT>(), st));
        HIPCHK(hipMemsetAsync(d_groups, 0, 8, st));
        // (the heads of a two-key round read the OLD ranks: they are all computed before this scatter)
        hipLaunchKernelGGL((k_scatter_rank<IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, sa, gs, n, rank, d_groups);
        unsigned long long groups = 0;
        HIPCHK(hipMemcpyAsync(&groups, d_groups, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (groups == n) break;
        if (round > 64) { fmi_set_error("suffix array did not converge"); return FMI_ERR_STATE; }
        if (!WIDE) {
            hipLaunchKernelGGL((k_pair_keys<IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, sa, rank, n, h_step, nbits, dk.current());
            tb = tmp_bytes;
            HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, 0u, std::min<uint32_t>(64u, 2 * nbits), st));
        } else {
            // LSD over the pair: stable sort by rank[i+h], then by rank[i]
            hipLaunchKernelGGL((k_key_of<IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, dv.current(), rank, n, h_step, dk.current());
            tb = tmp_bytes;
            HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, 0u, nbits, st));
            hipLaunchKernelGGL((k_key_of<IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, dv.current(), rank, n, (uint64_t)0, dk.current());
            tb = tmp_bytes;
            HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, 0u, nbits, st));
        }
        h_step *= 2;
    }
    IdxT *sa = dv.current();
    pool.release(keyA); pool.release(keyB); pool.release(rank); pool.release(dv.alternate());

    // ---- BWT + wavelet matrix + per-symbol tables ------------------------------
    SymT *bwt = nullptr;
    HIPCHK(pool.alloc(&bwt, n));
    hipLaunchKernelGGL((k_bwt<SymT, IdxT>), dim3(grid_for(n)), dim3(TB), 0, st, text, sa, n, bwt);
    FmiDev d{};
    uint64_t *wm = nullptr, *dC = nullptr, *dleaf = nullptr;
    uint8_t *dq1 = nullptr;
    {
        int rc = wavelet_from_bwt<SymT>(h, pool, st, bwt, n, max_sym, L, d, &wm, &dC, &dleaf, &dq1);
        if (rc) return rc;
    }
    const uint64_t nblk = d.nblk;

    // suffix array in its resident form: low 32 bits (+ bits 32..39 when indices are 64-bit)
    uint32_t *sa_lo_dev = nullptr;
    uint8_t *sa_hi_dev = nullptr;
    if (WIDE) {
        HIPCHK(pool.alloc(&sa_lo_dev, n)); HIPCHK(pool.alloc(&sa_hi_dev, n));
        hipLaunchKernelGGL(k_split_sa, dim3(grid_for(n)), dim3(TB), 0, st, (const uint64_t *)sa, n, sa_lo_dev, sa_hi_dev);
        HIPCHK(hipStreamSynchronize(st));
        pool.release(sa);
    } else {
        sa_lo_dev = (uint32_t *)sa;
    }
    if (keep_host) {
        h->wm.resize((uint64_t)h->dlevels * nblk * FMI_BLOCK_WORDS);
        HIPCHK(hipMemcpy(h->wm.data(), wm, h->wm.size() * 8, hipMemcpyDeviceToHost));
        h->sa_lo.resize(n);
        HIPCHK(hipMemcpy(h->sa_lo.data(), sa_lo_dev, n * 4, hipMemcpyDeviceToHost));
        h->sa_hi.clear();
        if (sa_hi_dev && n > (1ull << 32)) {       // the host format carries sa_hi only past 2^32 rows (fmi_host.cpp pack_sa)
            h->sa_hi.resize(n);
            HIPCHK(hipMemcpy(h->sa_hi.data(), sa_hi_dev, n, hipMemcpyDeviceToHost));
        }
        h->text.resize(n * sizeof(SymT));
        HIPCHK(hipMemcpy(h->text.data(), text, n * sizeof(SymT), hipMemcpyDeviceToHost));
        uint32_t *wide = nullptr;
        HIPCHK(pool.alloc(&wide, n));
        hipLaunchKernelGGL((k_widen<SymT>), dim3(grid_for(n)), dim3(TB), 0, st, bwt, n, wide);
        h->bwt.resize(n);
        HIPCHK(hipMemcpy(h->bwt.data(), wide, n * 4, hipMemcpyDeviceToHost));
        pool.release(wide);
        h->host_resident = true;
    } else {
        h->wm.clear(); h->sa_lo.clear(); h->sa_hi.clear(); h->text.clear(); h->bwt.clear();
        h->host_resident = false;
    }

    // hand the resident arrays over to the index
    d.C = dC; d.leaf = dleaf; d.q1 = dq1; d.sa_lo = sa_lo_dev; d.sa_hi = sa_hi_dev; d.text = text;
    d.doc_begin = nullptr; d.n_begin = 0; d.doc_hint = nullptr;
    for (void *p : {(void *)wm, (void *)d.sbase, (void *)dC, (void *)dleaf, (void *)dq1, (void *)sa_lo_dev, (void *)sa_hi_dev, (void *)text}) {
        if (!p) continue;
        pool.keep(p);
        h->dev_allocs.push_back(p);
    }
    h->dev_bytes = (uint64_t)h->dlevels * nblk * FMI_BLOCK_BYTES + (max_sym + 2) * 8 + (max_sym + 1) * 9 + (uint64_t)h->dlevels * d.nsb * FMI_ARITY * 8 + n * (WIDE ? 5 : 4) + n * sizeof(SymT);
    h->device = device;
    h->dev = d;
    if (!h->doc_begin.empty()) {
        std::vector<uint64_t> b = h->doc_begin;
        return fmi_set_doc_beginnings(h, b.data(), b.size());
    }
    return FMI_OK;
}


// ---------------------------------------------------------------------------
// Suffix array in SLICES, for texts whose prefix-doubling workspace (ranks + double-buffered keys and indices: ~42 B per symbol)
// does not fit the GPU next to the index itself -- BASELINE configs[4]: 1.4e10 symbols = 28 GB of text, 70 GB of suffix array,
// 56 GB of wavelet matrix; prefix doubling would need 590 GB.
//
// The suffixes are cut into slices by the VALUE of their first `per` symbols (splitters = quantiles of a sample of those keys, so
// every slice holds about `slice_rows` suffixes and slice s precedes slice s + 1 in suffix order).  A slice is sorted on its own:
// its positions are collected (one pass over the text per slice), sorted by the key of their first `per` symbols, and then
// refined -- the groups of suffixes that still agree are sorted by their NEXT `per` symbols (stable LSD: by the new key, then
// by the group), `per` symbols deeper every round, until every group is a single suffix.  The text has exactly one sentinel (its
// last symbol, smaller than every other), so two suffixes always differ before either runs off the end: the result is the suffix
// array, byte-identical to the prefix-doubling builder's (tests force small slices on test corpora).  Workspace: ~60 B per
// suffix OF ONE SLICE; rounds = longest repeat / per, two radix sorts each.
// ---------------------------------------------------------------------------
template <typename SymT>
__device__ __forceinline__ uint64_t key_at(const SymT *text, uint64_t n, uint64_t p, uint32_t bits, uint32_t per)
{
    uint64_t key = 0;
    for (uint32_t j = 0; j < per; j++) {
        const uint64_t s = (p + j < n) ? (uint64_t)text[p + j] : 0;
        key = (key << bits) | s;
    }
    return key;
}

template <typename SymT>
__global__ void k_sample_keys(const SymT *text, uint64_t n, uint32_t bits, uint32_t per, uint64_t stride, uint64_t n_samples, uint64_t *keys)
{
    GRID_STRIDE(i, n_samples) keys[i] = key_at(text, n, std::min<uint64_t>(i * stride, n - 1), bits, per);
}

// slice of a key: the number of splitters <= key (splitters ascending, n_split <= 1023)
__device__ __forceinline__ uint32_t slice_of(const uint64_t *split, uint32_t n_split, uint64_t key)
{
    uint32_t a = 0, b = n_split;
    while (a < b) { const uint32_t mid = (a + b) >> 1; if (split[mid] <= key) a = mid + 1; else b = mid; }
    return a;
}

template <typename SymT>
__global__ __launch_bounds__(256) void k_count_slices(const SymT *text, uint64_t n, uint32_t bits, uint32_t per, const uint64_t *split, uint32_t n_split,
                                                      unsigned long long *counts)
{
    __shared__ unsigned int local[1024];
    for (uint32_t i = threadIdx.x; i <= n_split; i += blockDim.x) local[i] = 0;
    __syncthreads();
    GRID_STRIDE(p, n) atomicAdd(&local[slice_of(split, n_split, key_at(text, n, p, bits, per))], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= n_split; i += blockDim.x) if (local[i]) atomicAdd(&counts[i], (unsigned long long)local[i]);
}

// the positions of one slice (key in [lo_key, hi_key), hi_key = 0 with `last`: no upper bound), in any order, with their keys
template <typename SymT>
__global__ __launch_bounds__(256) void k_collect_slice(const SymT *text, uint64_t n, uint32_t bits, uint32_t per, uint64_t lo_key, uint64_t hi_key, int first,
                                                       int last, uint64_t *pos, uint64_t *keys, unsigned long long *cursor)
{
    GRID_STRIDE(p, n) {
        const uint64_t k = key_at(text, n, p, bits, per);
        const bool mine = (first || k >= lo_key) && (last || k < hi_key);
        const uint64_t bal = __ballot(mine);
        if (!bal) continue;
        unsigned long long base = 0;
        const uint32_t lane = threadIdx.x & 63;
        if (lane == (uint32_t)__builtin_ctzll(bal)) base = atomicAdd(cursor, (unsigned long long)__popcll(bal));
        base = __shfl(base, __builtin_ctzll(bal));
        if (mine) {
            const uint64_t at = base + (uint64_t)__popcll(bal & ((1ull << lane) - 1));
            pos[at] = p; keys[at] = k;
        }
    }
}

__global__ void k_iota32(uint32_t *v, uint64_t m) { GRID_STRIDE(j, m) v[j] = (uint32_t)j; }

// group starts after the first sort: head where the key changes
__global__ void k_slice_heads0(const uint64_t *keys, uint64_t m, uint32_t *gs)
{
    GRID_STRIDE(j, m) gs[j] = (j == 0 || keys[j] != keys[j - 1]) ? (uint32_t)j : 0u;
}

// a suffix is unresolved while it shares its group with another one
__global__ void k_slice_unresolved(const uint32_t *gid, uint64_t m, uint8_t *flags)
{
    GRID_STRIDE(j, m) flags[j] = !(gid[j] == (uint32_t)j && (j + 1 == m || gid[j + 1] == (uint32_t)(j + 1)));
}

// key of the `per` symbols at depth `depth` of the given suffixes
template <typename SymT>
__global__ void k_keys_at(const SymT *text, uint64_t n, uint32_t bits, uint32_t per, uint64_t depth, const uint64_t *pos, uint64_t m, uint64_t *keys)
{
    GRID_STRIDE(j, m) keys[j] = key_at(text, n, pos[j] + depth, bits, per);
}

// the re-ordered unresolved suffixes go back to the unresolved slots (ascending: group after group); a slot starts a group where the
// old group or the new key changes (group heads carry their own slot number, the rest 0: the next max-scan fills them in)
__global__ void k_slice_write_back(const uint32_t *act_slot, const uint64_t *apos, const uint32_t *sg, const uint64_t *akey, uint64_t A, uint64_t *pos,
                                   uint32_t *gid)
{
    GRID_STRIDE(j, A) {
        const uint32_t slot = act_slot[j];
        pos[slot] = apos[j];
        gid[slot] = (j == 0 || sg[j] != sg[j - 1] || akey[j] != akey[j - 1]) ? slot : 0u;
    }
}

template <typename T>
__global__ void k_gather_by(const T *src, const uint32_t *idx, T *dst, uint64_t m) { GRID_STRIDE(j, m) dst[j] = src[idx[j]]; }

__global__ void k_store_sa(const uint64_t *pos, uint64_t m, uint64_t base, uint32_t *sa_lo, uint8_t *sa_hi)
{
    GRID_STRIDE(j, m) { sa_lo[base + j] = (uint32_t)pos[j]; if (sa_hi) sa_hi[base + j] = (uint8_t)(pos[j] >> 32); }
}

template <typename SymT>
__global__ void k_bwt_split(const SymT *text, const uint32_t *sa_lo, const uint8_t *sa_hi, uint64_t n, SymT *bwt)
{
    GRID_STRIDE(j, n) { const uint64_t p = (uint64_t)sa_lo[j] | (sa_hi ? (uint64_t)sa_hi[j] << 32 : 0ull); bwt[j] = text[p ? p - 1 : n - 1]; }
}

template <typename SymT>
__global__ void k_max_sym(const SymT *text, uint64_t n, unsigned int *out)
{
    unsigned int local = 0;
    GRID_STRIDE(i, n) local = max(local, (unsigned int)text[i]);
    for (int o = 32; o > 0; o >>= 1) local = max(local, (unsigned int)__shfl_down((int)local, o));
    if ((threadIdx.x & 63) == 0 && local) atomicMax(out, local);
}

template <typename SymT>
int build_sliced_impl(fmi *h, const SymT *text, uint64_t n, int device, uint64_t max_sym, uint32_t L, uint64_t slice_rows, bool text_owned)
{
    Pool pool;
    hipStream_t st = 0;
    const uint32_t per = std::max<uint32_t>(1, 64 / L);
    const bool wide = n > (1ull << 32);
    uint32_t *sa_lo = nullptr;
    uint8_t *sa_hi = nullptr;
    HIPCHK(pool.alloc(&sa_lo, n));
    if (wide) HIPCHK(pool.alloc(&sa_hi, n));
    // ---- splitters: quantiles of the leading keys of ~2^20 evenly spaced suffixes ----
    slice_rows = std::max<uint64_t>(slice_rows, 64);
    uint32_t n_slices = (uint32_t)std::min<uint64_t>((n + slice_rows - 1) / slice_rows, 1000);
    std::vector<uint64_t> split;                     // ascending, distinct
    std::vector<unsigned long long> counts;
    unsigned long long *d_counts = nullptr, *d_cursor = nullptr;
    uint64_t *d_split = nullptr;
    HIPCHK(pool.alloc(&d_counts, 1024)); HIPCHK(pool.alloc(&d_cursor, 1)); HIPCHK(pool.alloc(&d_split, 1024));
    for (int attempt = 0;; attempt++) {
        split.clear();
        if (n_slices > 1) {
            const uint64_t n_samples = std::min<uint64_t>(n, 1ull << 20);
            const uint64_t stride = std::max<uint64_t>(1, n / n_samples);
            uint64_t *sk = nullptr, *sk2 = nullptr;
            HIPCHK(pool.alloc(&sk, n_samples)); HIPCHK(pool.alloc(&sk2, n_samples));
            hipLaunchKernelGGL((k_sample_keys<SymT>), dim3(grid_for(n_samples)), dim3(TB), 0, st, text, n, L, per, stride, n_samples, sk);
            rocprim::double_buffer<uint64_t> db(sk, sk2);
            size_t tb = 0;
            HIPCHK(rocprim::radix_sort_keys(nullptr, tb, db, n_samples, 0u, 64u, st));
            void *tmp = nullptr;
            HIPCHK(pool.alloc((char **)&tmp, tb + 256));
            HIPCHK(rocprim::radix_sort_keys(tmp, tb, db, n_samples, 0u, 64u, st));
            std::vector<uint64_t> hs(n_samples);
            HIPCHK(hipMemcpy(hs.data(), db.current(), n_samples * 8, hipMemcpyDeviceToHost));
            pool.release(sk); pool.release(sk2); pool.release(tmp);
            for (uint32_t q = 1; q < n_slices; q++) {
                const uint64_t v = hs[(uint64_t)q * n_samples / n_slices];
                if (split.empty() || v > split.back()) split.push_back(v);
            }
        }
        counts.assign(split.size() + 1, 0);
        HIPCHK(hipMemsetAsync(d_counts, 0, 1024 * 8, st));
        if (!split.empty()) HIPCHK(hipMemcpyAsync(d_split, split.data(), split.size() * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((k_count_slices<SymT>), dim3(grid_for(n)), dim3(256), 0, st, text, n, L, per, (const uint64_t *)d_split, (uint32_t)split.size(), d_counts);
        HIPCHK(hipMemcpy(counts.data(), d_counts, counts.size() * 8, hipMemcpyDeviceToHost));
        const unsigned long long biggest = *std::max_element(counts.begin(), counts.end());
        if (biggest <= 2 * slice_rows || n_slices >= 1000 || attempt >= 3) {
            if (biggest >= (1ull << 32)) { fmi_set_error("fmi_build_device_sliced: one leading key holds %llu suffixes; a slice is limited to 2^32", biggest); return FMI_ERR_CAPACITY; }
            break;
        }
        n_slices = std::min<uint32_t>(n_slices * 2, 1000);          // skewed keys: cut finer and look again
    }
    const uint64_t cap = *std::max_element(counts.begin(), counts.end());
    // ---- slice workspace (~65 B per suffix of the largest slice) ----
    uint64_t *pos = nullptr, *aposA = nullptr, *aposB = nullptr, *akeyA = nullptr, *akeyB = nullptr;
    uint32_t *gid = nullptr, *act_slot = nullptr, *agidA = nullptr, *agidB = nullptr, *permA = nullptr, *permB = nullptr;
    uint8_t *flags = nullptr;
    HIPCHK(pool.alloc(&pos, cap)); HIPCHK(pool.alloc(&aposA, cap)); HIPCHK(pool.alloc(&aposB, cap)); HIPCHK(pool.alloc(&akeyA, cap)); HIPCHK(pool.alloc(&akeyB, cap));
    HIPCHK(pool.alloc(&gid, cap)); HIPCHK(pool.alloc(&act_slot, cap)); HIPCHK(pool.alloc(&agidA, cap)); HIPCHK(pool.alloc(&agidB, cap));
    HIPCHK(pool.alloc(&permA, cap)); HIPCHK(pool.alloc(&permB, cap)); HIPCHK(pool.alloc(&flags, cap));
    size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    {
        rocprim::double_buffer<uint64_t> k64(akeyA, akeyB), v64(aposA, aposB);
        rocprim::double_buffer<uint32_t> k32(agidA, agidB), v32(permA, permB);
        HIPCHK(rocprim::radix_sort_pairs(nullptr, t1, k64, v64, cap, 0u, 64u, st));
        HIPCHK(rocprim::radix_sort_pairs(nullptr, t2, k64, v32, cap, 0u, 64u, st));
        HIPCHK(rocprim::radix_sort_pairs(nullptr, t3, k32, v32, cap, 0u, 32u, st));
        HIPCHK(rocprim::inclusive_scan(nullptr, t4, (uint32_t *)nullptr, (uint32_t *)nullptr, cap, rocprim::maximum<uint32_t>(), st));
        HIPCHK(rocprim::select(nullptr, t5, rocprim::counting_iterator<uint32_t>(0), (uint8_t *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr, cap, st));
    }
    const size_t tmp_bytes = std::max(std::max(std::max(t1, t2), std::max(t3, t4)), t5) + 256;
    void *tmp = nullptr;
    HIPCHK(pool.alloc((char **)&tmp, tmp_bytes));
    const uint32_t key_bits = std::min<uint32_t>(64u, per * L);
    uint64_t base = 0;
    for (size_t sl = 0; sl < counts.size(); sl++) {
        const uint64_t m = counts[sl];
        if (m == 0) continue;
        // ---- the slice's suffixes, sorted by their first `per` symbols ----
        HIPCHK(hipMemsetAsync(d_cursor, 0, 8, st));
        hipLaunchKernelGGL((k_collect_slice<SymT>), dim3(grid_for(n)), dim3(256), 0, st, text, n, L, per, sl ? split[sl - 1] : 0ull,
                           sl < split.size() ? split[sl] : 0ull, (int)(sl == 0), (int)(sl == split.size()), aposA, akeyA, d_cursor);
        {
            rocprim::double_buffer<uint64_t> dk(akeyA, akeyB), dv(aposA, aposB);
            size_t tb = tmp_bytes;
            HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, m, 0u, key_bits, st));
            HIPCHK(hipMemcpyAsync(pos, dv.current(), m * 8, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_slice_heads0, dim3(grid_for(m)), dim3(TB), 0, st, (const uint64_t *)dk.current(), m, gid);
        }
        uint32_t gbits = 1;
        while (gbits < 32 && (m >> gbits)) gbits++;
        // ---- refinement: the suffixes that still share a group, `per` symbols deeper per round ----
        uint64_t depth = 0;
        for (int round = 0;; round++) {
            size_t sb = tmp_bytes;
            HIPCHK(rocprim::inclusive_scan(tmp, sb, gid, gid, m, rocprim::maximum<uint32_t>(), st));       // gid[j] = slot of j's group head
            hipLaunchKernelGGL(k_slice_unresolved, dim3(grid_for(m)), dim3(TB), 0, st, (const uint32_t *)gid, m, flags);
            sb = tmp_bytes;
            HIPCHK(rocprim::select(tmp, sb, rocprim::counting_iterator<uint32_t>(0), flags, act_slot, d_cursor, m, st));
            unsigned long long A = 0;
            HIPCHK(hipMemcpy(&A, d_cursor, 8, hipMemcpyDeviceToHost));
            if (A == 0) break;
            depth += per;
            if (depth > n || round > 1000000) { fmi_set_error("fmi_build_device_sliced: the suffixes of slice %zu did not separate", sl); return FMI_ERR_STATE; }
            hipLaunchKernelGGL(k_gather_by<uint64_t>, dim3(grid_for(A)), dim3(TB), 0, st, (const uint64_t *)pos, (const uint32_t *)act_slot, aposA, A);
            hipLaunchKernelGGL(k_gather_by<uint32_t>, dim3(grid_for(A)), dim3(TB), 0, st, (const uint32_t *)gid, (const uint32_t *)act_slot, agidA, A);
            hipLaunchKernelGGL((k_keys_at<SymT>), dim3(grid_for(A)), dim3(TB), 0, st, text, n, L, per, depth, (const uint64_t *)aposA, A, akeyA);
            hipLaunchKernelGGL(k_iota32, dim3(grid_for(A)), dim3(TB), 0, st, permA, A);
            // stable LSD over (group, next key): by the next key ...
            uint32_t *perm = permA, *perm_alt = permB;
            {
                rocprim::double_buffer<uint64_t> dk(akeyA, akeyB);
                rocprim::double_buffer<uint32_t> dv(perm, perm_alt);
                size_t tb = tmp_bytes;
                HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, A, 0u, key_bits, st));
                perm = dv.current(); perm_alt = dv.alternate();
            }
            // ... then by the group: the unresolved suffixes of a group are back in the group's own slots, in (next key) order
            hipLaunchKernelGGL(k_gather_by<uint32_t>, dim3(grid_for(A)), dim3(TB), 0, st, (const uint32_t *)agidA, (const uint32_t *)perm, agidB, A);
            uint32_t *sg = nullptr;
            {
                rocprim::double_buffer<uint32_t> dk(agidB, agidA), dv(perm, perm_alt);       // (agidA, the ids in the old order, is free now)
                size_t tb = tmp_bytes;
                HIPCHK(rocprim::radix_sort_pairs(tmp, tb, dk, dv, A, 0u, gbits, st));
                sg = dk.current(); perm = dv.current();
            }
            hipLaunchKernelGGL(k_gather_by<uint64_t>, dim3(grid_for(A)), dim3(TB), 0, st, (const uint64_t *)aposA, (const uint32_t *)perm, aposB, A);
            hipLaunchKernelGGL((k_keys_at<SymT>), dim3(grid_for(A)), dim3(TB), 0, st, text, n, L, per, depth, (const uint64_t *)aposB, A, akeyA);
            hipLaunchKernelGGL(k_slice_write_back, dim3(grid_for(A)), dim3(TB), 0, st, (const uint32_t *)act_slot, (const uint64_t *)aposB, (const uint32_t *)sg,
                               (const uint64_t *)akeyA, (uint64_t)A, pos, gid);
        }
        hipLaunchKernelGGL(k_store_sa, dim3(grid_for(m)), dim3(TB), 0, st, (const uint64_t *)pos, m, base, sa_lo, sa_hi);
        HIPCHK(hipStreamSynchronize(st));
        base += m;
    }
    if (base != n) { fmi_set_error("fmi_build_device_sliced: %llu of %llu suffixes placed", (unsigned long long)base, (unsigned long long)n); return FMI_ERR_STATE; }
    for (void *p : {(void *)pos, (void *)aposA, (void *)aposB, (void *)akeyA, (void *)akeyB, (void *)gid, (void *)act_slot, (void *)agidA, (void *)agidB, (void *)permA,
                    (void *)permB, (void *)flags, tmp}) pool.release(p);

    // ---- BWT, wavelet matrix, tables ----
    SymT *bwt = nullptr;
    HIPCHK(pool.alloc(&bwt, n));
    hipLaunchKernelGGL((k_bwt_split<SymT>), dim3(grid_for(n)), dim3(TB), 0, st, text, (const uint32_t *)sa_lo, (const uint8_t *)sa_hi, n, bwt);
    FmiDev d{};
    uint64_t *wm = nullptr, *dC = nullptr, *dleaf = nullptr;
    uint8_t *dq1 = nullptr;
    {
        int rc = wavelet_from_bwt<SymT>(h, pool, st, bwt, n, max_sym, L, d, &wm, &dC, &dleaf, &dq1, bwt);
        if (rc) return rc;
    }
    h->wm.clear(); h->sa_lo.clear(); h->sa_hi.clear(); h->text.clear(); h->bwt.clear();
    h->host_resident = false;
    d.C = dC; d.leaf = dleaf; d.q1 = dq1; d.sa_lo = sa_lo; d.sa_hi = sa_hi; d.text = text;
    d.doc_begin = nullptr; d.n_begin = 0; d.doc_hint = nullptr;
    for (void *p : {(void *)wm, (void *)d.sbase, (void *)dC, (void *)dleaf, (void *)dq1, (void *)sa_lo, (void *)sa_hi}) {
        if (!p) continue;
        pool.keep(p);
        h->dev_allocs.push_back(p);
    }
    if (text_owned) h->dev_allocs.push_back((void *)text);
    h->dev_bytes = (uint64_t)h->dlevels * d.nblk * FMI_BLOCK_BYTES + (max_sym + 2) * 8 + (max_sym + 1) * 9 + (uint64_t)h->dlevels * d.nsb * FMI_ARITY * 8 +
                   n * (wide ? 5 : 4) + n * sizeof(SymT);
    h->device = device;
    h->dev = d;
    if (!h->doc_begin.empty()) {
        std::vector<uint64_t> b = h->doc_begin;
        return fmi_set_doc_beginnings(h, b.data(), b.size());
    }
    return FMI_OK;
}

}  // namespace

template <typename SymT>
static int bwt_only_impl(fmi *h, const void *d_bwt, uint64_t n, int device, uint64_t max_sym, uint32_t L)
{
    Pool pool;
    FmiDev d{};
    uint64_t *wm = nullptr, *dC = nullptr, *dleaf = nullptr;
    uint8_t *dq1 = nullptr;
    int rc = wavelet_from_bwt<SymT>(h, pool, 0, (const SymT *)d_bwt, n, max_sym, L, d, &wm, &dC, &dleaf, &dq1);
    if (rc) return rc;
    h->wm.clear(); h->sa_lo.clear(); h->sa_hi.clear(); h->text.clear(); h->bwt.clear();
    h->host_resident = false;
    d.C = dC; d.leaf = dleaf; d.q1 = dq1; d.sa_lo = nullptr; d.sa_hi = nullptr; d.text = nullptr;
    for (void *p : {(void *)wm, (void *)d.sbase, (void *)dC, (void *)dleaf, (void *)dq1}) { pool.keep(p); h->dev_allocs.push_back(p); }
    h->dev_bytes = (uint64_t)h->dlevels * d.nblk * FMI_BLOCK_BYTES + (max_sym + 2) * 8 + (max_sym + 1) * 9;
    h->device = device;
    h->dev = d;
    return FMI_OK;
}

extern "C" int fmi_build_from_bwt_device(fmi_t *h, const void *d_bwt, uint64_t n, int sym_bytes, uint64_t max_sym, int device)
{
    if (!h || !d_bwt || n == 0 || !(sym_bytes == 2 || sym_bytes == 4)) { fmi_set_error("fmi_build_from_bwt_device: bad argument"); return FMI_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= device || device < 0) { fmi_set_error("no HIP device %d visible", device); return FMI_ERR_NO_DEVICE; }
    if (n >= (1ull << 40)) { fmi_set_error("positions are 40-bit"); return FMI_ERR_UNSUPPORTED; }
    uint32_t L = 0;
    while ((max_sym >> L) > 0) L++;
    if (L == 0) L = 1;
    if (L > FMI_MAX_LEVELS || (sym_bytes == 2 && max_sym >= 65536)) { fmi_set_error("alphabet too large"); return FMI_ERR_UNSUPPORTED; }
    fmi_release_device(h);
    HIPCHK(hipSetDevice(device));
    return sym_bytes == 2 ? bwt_only_impl<uint16_t>(h, d_bwt, n, device, max_sym, L) : bwt_only_impl<uint32_t>(h, d_bwt, n, device, max_sym, L);
}

extern "C" int fmi_build_device(fmi_t *h, const uint32_t *d_data, uint64_t n_data, int device, int keep_host)
{
    if (!h || (!d_data && n_data)) { fmi_set_error("fmi_build_device: null argument"); return FMI_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= device || device < 0) {
        fmi_set_error("no HIP device %d visible", device);
        return FMI_ERR_NO_DEVICE;
    }
    if (n_data + 1 >= (1ull << 40)) {
        fmi_set_error("fmi_build_device: %llu symbols; positions are 40-bit", (unsigned long long)n_data);
        return FMI_ERR_UNSUPPORTED;
    }
    // 32-bit suffix indices up to 2^32-1 rows, 64-bit (two-pass sort per round) beyond;
    // SEALFM_FORCE_IDX64=1 selects the wide path at any size (tests)
    const char *force = getenv("SEALFM_FORCE_IDX64");
    const bool wide = (n_data + 1 >= (1ull << 32)) || (force && force[0] == '1');
    fmi_release_device(h);
    HIPCHK(hipSetDevice(device));
    // alphabet
    uint32_t *d_max = nullptr;
    size_t rb = 0;
    uint32_t max_sym32 = 0;
    if (n_data) {
        HIPCHK(hipMalloc((void **)&d_max, 4));
        HIPCHK(rocprim::reduce(nullptr, rb, d_data, d_max, (uint32_t)0, n_data, rocprim::maximum<uint32_t>(), 0));
        void *rt = nullptr;
        HIPCHK(hipMalloc(&rt, rb + 256));
        HIPCHK(rocprim::reduce(rt, rb, d_data, d_max, (uint32_t)0, n_data, rocprim::maximum<uint32_t>(), 0));
        HIPCHK(hipMemcpy(&max_sym32, d_max, 4, hipMemcpyDeviceToHost));
        (void)hipFree(rt); (void)hipFree(d_max);
    }
    uint32_t L = 0;
    while (((uint64_t)max_sym32 >> L) > 0) L++;
    if (L == 0) L = 1;
    if (L > FMI_MAX_LEVELS) { fmi_set_error("alphabet needs %u bits per symbol; this build supports <= %u", L, FMI_MAX_LEVELS); return FMI_ERR_UNSUPPORTED; }
    // (a 0 inside the data would collide with the sentinel; the caller owns that contract, as with sdsl)
    if (max_sym32 < 65536)
        return wide ? build_impl<uint16_t, uint64_t>(h, d_data, n_data, device, keep_host, max_sym32, L)
                    : build_impl<uint16_t, uint32_t>(h, d_data, n_data, device, keep_host, max_sym32, L);
    return wide ? build_impl<uint32_t, uint64_t>(h, d_data, n_data, device, keep_host, max_sym32, L)
                : build_impl<uint32_t, uint32_t>(h, d_data, n_data, device, keep_host, max_sym32, L);
}

// Index construction with the suffix array sorted in slices (see build_sliced_impl): for texts whose prefix-doubling workspace does not
// fit the GPU.  `d_text`: the n symbols of the text INCLUDING the final 0 sentinel, 2 or 4 bytes each, in device memory that the CALLER
// owns and keeps alive for the life of the index (it becomes the index's resident text: nothing is copied).  slice_rows: suffixes per
// slice (0: 2^30).
extern "C" int fmi_build_device_sliced(fmi_t *h, const void *d_text, uint64_t n, int sym_bytes, int device, uint64_t slice_rows)
{
    if (!h || !d_text || n < 2 || !(sym_bytes == 2 || sym_bytes == 4)) { fmi_set_error("fmi_build_device_sliced: bad argument"); return FMI_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= device || device < 0) { fmi_set_error("no HIP device %d visible", device); return FMI_ERR_NO_DEVICE; }
    if (n >= (1ull << 40)) { fmi_set_error("fmi_build_device_sliced: %llu symbols; positions are 40-bit", (unsigned long long)n); return FMI_ERR_UNSUPPORTED; }
    fmi_release_device(h);
    HIPCHK(hipSetDevice(device));
    uint32_t last = 1;
    HIPCHK(hipMemcpy(&last, (const uint8_t *)d_text + (n - 1) * sym_bytes, sym_bytes, hipMemcpyDeviceToHost));
    if (sym_bytes == 2) last &= 0xffffu;
    if (last != 0) { fmi_set_error("fmi_build_device_sliced: the text must end with the 0 sentinel"); return FMI_ERR_ARG; }
    unsigned int *d_max = nullptr, max_sym32 = 0;
    HIPCHK(hipMalloc((void **)&d_max, 4));
    HIPCHK(hipMemset(d_max, 0, 4));
    if (sym_bytes == 2) hipLaunchKernelGGL((k_max_sym<uint16_t>), dim3(grid_for(n)), dim3(TB), 0, 0, (const uint16_t *)d_text, n, d_max);
    else hipLaunchKernelGGL((k_max_sym<uint32_t>), dim3(grid_for(n)), dim3(TB), 0, 0, (const uint32_t *)d_text, n, d_max);
    HIPCHK(hipMemcpy(&max_sym32, d_max, 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_max);
    uint32_t L = 0;
    while (((uint64_t)max_sym32 >> L) > 0) L++;
    if (L == 0) L = 1;
    if (L > FMI_MAX_LEVELS) { fmi_set_error("alphabet needs %u bits per symbol; this build supports <= %u", L, FMI_MAX_LEVELS); return FMI_ERR_UNSUPPORTED; }
    if (sym_bytes == 2 && max_sym32 >= 65536) { fmi_set_error("internal: 16-bit symbols above 65535"); return FMI_ERR_STATE; }
    if (slice_rows == 0) slice_rows = 1ull << 30;
    return sym_bytes == 2 ? build_sliced_impl<uint16_t>(h, (const uint16_t *)d_text, n, device, max_sym32, L, slice_rows, false)
                          : build_sliced_impl<uint32_t>(h, (const uint32_t *)d_text, n, device, max_sym32, L, slice_rows, false);
}
