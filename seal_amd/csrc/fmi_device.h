// Device-side primitives shared by the query kernels and the GPU builder.
#pragma once
#include <hip/hip_runtime.h>

#include "fmi_internal.h"

// ---------------------------------------------------------------------------
// device primitives
// ---------------------------------------------------------------------------
struct alignas(16) U64x2 { uint64_t x, y; };

// The chunks of one 64-byte block that a rank probe at one position needs (fmi_internal.h):
// the header and the one or two group chunks between the position and the header's reference point.
struct QProbe {
    U64x2 hdr, ch[2];
    uint32_t group, bit;
};

// position -> (block, 64-position group inside the block)
__device__ __forceinline__ uint64_t wm_block_of(const FmiDev &ix, uint64_t p, uint32_t &group)
{
    const uint64_t w = p >> 6;
    uint64_t blk;
    if (ix.n < (1ull << 37)) blk = (uint32_t)w / 3u;      // word index fits 32 bits: one mul_hi instead of a 64-bit divide
    else blk = w / 3;
    group = (uint32_t)(w - blk * 3);
    return blk;
}

// issue the loads of a probe: header, the probe's own side of the block, and (group 2 only) one more
// chunk -- per-lane predicated 16-byte loads, 2.33 L1 tag look-ups per lane on average
__device__ __forceinline__ void wm_probe_load(const FmiDev &ix, uint32_t q, uint64_t blk, uint32_t group, uint32_t bit, QProbe &pr)
{
    const U64x2 *src = reinterpret_cast<const U64x2 *>(ix.wm + ((uint64_t)q * ix.nblk + blk) * FMI_BLOCK_WORDS);
    pr.group = group; pr.bit = bit;
    pr.hdr = src[1];
    pr.ch[0] = src[group == 0 ? 0 : 2];         // group 0 (counted backwards) or group 1
    pr.ch[1] = U64x2{0, 0};
    if (group == 2) pr.ch[1] = src[3];
}

// digits equal to 1 / 2 / 3 in the level before the probe's position
__device__ __forceinline__ void wm_probe_counts(const QProbe &pr, uint64_t &r1, uint64_t &r2, uint64_t &r3)
{
    const bool back = pr.group == 0;
    const uint64_t tail = (1ull << pr.bit) - 1;
    // group 0: positions >= p of chunk 0;  group 1: positions < p of chunk 0;  group 2: all of chunk 0, < p of chunk 1
    const uint64_t m0 = back ? ~tail : (pr.group == 1 ? tail : ~0ull);
    const uint64_t m1 = pr.group == 2 ? tail : 0ull;
    const uint64_t H0 = pr.ch[0].x & m0, L0 = pr.ch[0].y & m0, H1 = pr.ch[1].x & m1, L1 = pr.ch[1].y & m1;
    const uint32_t nh = (uint32_t)__popcll(H0) + (uint32_t)__popcll(H1), nl = (uint32_t)__popcll(L0) + (uint32_t)__popcll(L1);
    const uint32_t nhl = (uint32_t)__popcll(H0 & L0) + (uint32_t)__popcll(H1 & L1);
    const uint64_t w0 = pr.hdr.x, w1 = pr.hdr.y;
    const uint64_t c1 = w0 & 0xffffffffffull, c2 = (w0 >> 40) | ((w1 & 0xffffull) << 24), c3 = w1 >> 16;
    const uint64_t d1 = nl - nhl, d2 = nh - nhl, d3 = nhl;
    r1 = back ? c1 - d1 : c1 + d1;
    r2 = back ? c2 - d2 : c2 + d2;
    r3 = back ? c3 - d3 : c3 + d3;
}

// where position p of quad level q goes in level q+1 if its symbol has digit d there
// (p = n maps an exclusive upper bound): ONE 64-byte sector.
__device__ __forceinline__ uint64_t wm_step(const FmiDev &ix, uint32_t q, uint64_t p, uint32_t d, uint64_t *sectors)
{
    uint32_t group;
    const uint64_t blk = wm_block_of(ix, p, group);
    QProbe pr;
    wm_probe_load(ix, q, blk, group, (uint32_t)(p & 63), pr);
    uint64_t r1, r2, r3;
    wm_probe_counts(pr, r1, r2, r3);
    if (sectors) ++*sectors;
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
// the loads of both ends are issued back to back.  Returns the 64-byte sectors (blocks) touched.
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
    return blo == bhi ? 1u : 2u;
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
