// Device-side primitives shared by the query kernels and the GPU builder.
#pragma once
#include <hip/hip_runtime.h>

#include "fmi_internal.h"

// ---------------------------------------------------------------------------
// device primitives
// ---------------------------------------------------------------------------
struct alignas(16) U64x2 { uint64_t x, y; };

// The chunks of one 128-byte block that a rank probe at one position needs (fmi_internal.h):
// the header and the <= 4 group chunks between the position and the header's reference point.
struct QProbe {
    U64x2 hdr, ch[4];
    uint32_t group, bit;
};

// position -> (block, 64-position group inside the block)
__device__ __forceinline__ uint64_t wm_block_of(const FmiDev &ix, uint64_t p, uint32_t &group)
{
    const uint64_t w = p >> 6;
    uint64_t blk;
    if (ix.n < (1ull << 37)) blk = (uint32_t)w / 7u;      // word index fits 32 bits: one mul_hi instead of a 64-bit divide
    else blk = w / 7;
    group = (uint32_t)(w - blk * 7);
    return blk;
}

// issue the loads of a probe: lanes only touch the chunks they need (per-lane predicated 16-byte
// loads), so a wave-wide probe costs ~3.3 L1 tag look-ups per lane instead of 8
__device__ __forceinline__ void wm_probe_load(const FmiDev &ix, uint32_t q, uint64_t blk, uint32_t group, uint32_t bit, QProbe &pr)
{
    const U64x2 *src = reinterpret_cast<const U64x2 *>(ix.wm + ((uint64_t)q * ix.nblk + blk) * FMI_BLOCK_WORDS);
    pr.group = group; pr.bit = bit;
    pr.hdr = src[3];
    const uint32_t first = group < 3 ? group : 4, nch = group < 3 ? 3 - group : group - 2;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        pr.ch[t] = U64x2{0, 0};
        if (t < nch) pr.ch[t] = src[first + t];
    }
}

// digits equal to 1 / 2 / 3 in the level before the probe's position
__device__ __forceinline__ void wm_probe_counts(const QProbe &pr, uint64_t &r1, uint64_t &r2, uint64_t &r3)
{
    const bool back = pr.group < 3;
    const uint32_t nch = back ? 3 - pr.group : pr.group - 2;
    const uint64_t tail = (1ull << pr.bit) - 1;
    uint32_t nh = 0, nl = 0, nhl = 0;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        uint64_t m = t < nch ? ~0ull : 0ull;
        if (back) { if (t == 0) m = ~tail; }          // positions >= p of the probe's own group
        else if (t + 1 == nch) m = tail;              // positions < p of the probe's own group
        const uint64_t H = pr.ch[t].x & m, Lw = pr.ch[t].y & m;
        nh += (uint32_t)__popcll(H); nl += (uint32_t)__popcll(Lw); nhl += (uint32_t)__popcll(H & Lw);
    }
    const uint64_t w0 = pr.hdr.x, w1 = pr.hdr.y;
    const uint64_t c1 = w0 & 0xffffffffffull, c2 = (w0 >> 40) | ((w1 & 0xffffull) << 24), c3 = w1 >> 16;
    const uint64_t d1 = nl - nhl, d2 = nh - nhl, d3 = nhl;
    r1 = back ? c1 - d1 : c1 + d1;
    r2 = back ? c2 - d2 : c2 + d2;
    r3 = back ? c3 - d3 : c3 + d3;
}

// 64-byte sectors of the block a probe touches: the first always (header), the second iff group >= 3
__device__ __forceinline__ uint32_t wm_probe_sectors(uint32_t group) { return group < 3 ? 1u : 3u; }

// where position p of quad level q goes in level q+1 if its symbol has digit d there
// (p = n maps an exclusive upper bound): ONE 128-byte line.
__device__ __forceinline__ uint64_t wm_step(const FmiDev &ix, uint32_t q, uint64_t p, uint32_t d, uint64_t *sectors)
{
    uint32_t group;
    const uint64_t blk = wm_block_of(ix, p, group);
    QProbe pr;
    wm_probe_load(ix, q, blk, group, (uint32_t)(p & 63), pr);
    uint64_t r1, r2, r3;
    wm_probe_counts(pr, r1, r2, r3);
    if (sectors) *sectors += __popc(wm_probe_sectors(group));
    // q is wave-uniform at every call site: three scalar loads + selects instead of a per-lane table load
    const uint64_t b1 = ix.qbase[q][1], b2 = ix.qbase[q][2], b3 = ix.qbase[q][3];
    return d == 0 ? p - r1 - r2 - r3 : (d == 1 ? b1 + r1 : (d == 2 ? b2 + r2 : b3 + r3));
}

// number of occurrences of symbol c in BWT[0, i), 0 <= i <= n
__device__ __forceinline__ uint64_t wm_rank_sym(const FmiDev &ix, uint64_t c, uint64_t i, uint64_t *sectors)
{
    uint64_t p = i;
    for (uint32_t q = 0; q < ix.qlevels; q++)
        p = wm_step(ix, q, p, (uint32_t)(c >> (2 * (ix.qlevels - 1 - q))) & 3u, sectors);
    return p - ix.leaf[c];
}

// the two ends [lo, hi) of an interval on quad level q -> the four child intervals on level q+1;
// the loads of both ends are issued back to back.  Returns the 64-byte sectors touched.
__device__ __forceinline__ uint32_t wm_children(const FmiDev &ix, uint32_t q, uint64_t lo, uint64_t hi, uint64_t (&clo)[4], uint64_t (&chi)[4])
{
    uint32_t glo, ghi;
    const uint64_t blo = wm_block_of(ix, lo, glo), bhi = wm_block_of(ix, hi, ghi);
    QProbe a, b;
    wm_probe_load(ix, q, blo, glo, (uint32_t)(lo & 63), a);
    wm_probe_load(ix, q, bhi, ghi, (uint32_t)(hi & 63), b);
    uint64_t a1, a2, a3, b1, b2, b3;
    wm_probe_counts(a, a1, a2, a3);
    wm_probe_counts(b, b1, b2, b3);
    const uint64_t q1 = ix.qbase[q][1], q2 = ix.qbase[q][2], q3 = ix.qbase[q][3];
    clo[0] = lo - a1 - a2 - a3; chi[0] = hi - b1 - b2 - b3;
    clo[1] = q1 + a1; chi[1] = q1 + b1;
    clo[2] = q2 + a2; chi[2] = q2 + b2;
    clo[3] = q3 + a3; chi[3] = q3 + b3;
    const uint32_t sa = wm_probe_sectors(glo), sb = wm_probe_sectors(ghi);
    return blo == bhi ? (uint32_t)__popc(sa | sb) : (uint32_t)(__popc(sa) + __popc(sb));
}

// rank_c at two positions at once (the two ends of a backward-search interval): one dependent
// chain instead of two.
__device__ __forceinline__ void wm_rank_sym_pair(const FmiDev &ix, uint64_t c, uint64_t i, uint64_t j, uint64_t &ri, uint64_t &rj,
                                                 uint64_t *sectors)
{
    uint64_t p = i, s = j;
    for (uint32_t q = 0; q < ix.qlevels; q++) {
        const uint32_t d = (uint32_t)(c >> (2 * (ix.qlevels - 1 - q))) & 3u;
        uint64_t clo[4], chi[4];
        const uint32_t sec = wm_children(ix, q, p, s, clo, chi);
        if (sectors) *sectors += sec;
        p = d == 0 ? clo[0] : (d == 1 ? clo[1] : (d == 2 ? clo[2] : clo[3]));
        s = d == 0 ? chi[0] : (d == 1 ? chi[1] : (d == 2 ? chi[2] : chi[3]));
    }
    ri = p - ix.leaf[c]; rj = s - ix.leaf[c];
}
