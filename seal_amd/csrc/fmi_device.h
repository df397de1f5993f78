// Device-side primitives shared by the query kernels and the GPU builder.
#pragma once
#include <hip/hip_runtime.h>

#include "fmi_internal.h"

// ---------------------------------------------------------------------------
// device primitives over the hex wavelet matrix (layout: fmi_internal.h)
// ---------------------------------------------------------------------------

// One 128-byte block in registers: 7 x global_load_dwordx4 (the eighth chunk is padding).
struct HBlock {
    uint32_t lo[16];   // low 32 bits of c_0..c_15
    uint32_t hi[4];    // packed bits 32..39 of c_0..c_15
    uint64_t P[4];     // bit planes
};

__device__ __forceinline__ const uint32_t *wm_block_ptr(const FmiDev &ix, uint32_t k, uint64_t blk)
{
    return reinterpret_cast<const uint32_t *>(ix.wm + ((uint64_t)k * ix.nblk + blk) * FMI_BLOCK_WORDS);
}

__device__ __forceinline__ void wm_load_block(const FmiDev &ix, uint32_t k, uint64_t blk, HBlock &b)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(wm_block_ptr(ix, k, blk));
    const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
    b.lo[0] = v0.x; b.lo[1] = v0.y; b.lo[2] = v0.z; b.lo[3] = v0.w;
    b.lo[4] = v1.x; b.lo[5] = v1.y; b.lo[6] = v1.z; b.lo[7] = v1.w;
    b.lo[8] = v2.x; b.lo[9] = v2.y; b.lo[10] = v2.z; b.lo[11] = v2.w;
    b.lo[12] = v3.x; b.lo[13] = v3.y; b.lo[14] = v3.z; b.lo[15] = v3.w;
    b.hi[0] = v4.x; b.hi[1] = v4.y; b.hi[2] = v4.z; b.hi[3] = v4.w;
    b.P[0] = (uint64_t)v5.x | ((uint64_t)v5.y << 32); b.P[1] = (uint64_t)v5.z | ((uint64_t)v5.w << 32);
    b.P[2] = (uint64_t)v6.x | ((uint64_t)v6.y << 32); b.P[3] = (uint64_t)v6.z | ((uint64_t)v6.w << 32);
}

// rank_d(p) for all sixteen digits d, p = 64 * block + bit
__device__ __forceinline__ void wm_block_ranks(const HBlock &b, uint32_t bit, uint64_t (&r)[16])
{
    const uint64_t T = (1ull << bit) - 1;
    const uint64_t a[4] = {~b.P[3] & ~b.P[2], ~b.P[3] & b.P[2], b.P[3] & ~b.P[2], b.P[3] & b.P[2]};
    const uint64_t c[4] = {~b.P[1] & ~b.P[0] & T, ~b.P[1] & b.P[0] & T, b.P[1] & ~b.P[0] & T, b.P[1] & b.P[0] & T};
#pragma unroll
    for (uint32_t d = 0; d < 16; d++) {
        const uint64_t base = (uint64_t)b.lo[d] | ((uint64_t)((b.hi[d >> 2] >> (8 * (d & 3))) & 0xffu) << 32);
        r[d] = base + (uint64_t)__popcll(a[d >> 2] & c[d & 3]);
    }
}

// where position p of level k goes in level k+1 if its symbol has digit d there (p = n maps an
// exclusive upper bound): ONE 128-byte line, of which a single digit needs the two plane chunks,
// one counter dword and one counter byte.  The per-level offset table is read from HBM (d is
// per-lane); that load does not depend on the block and is issued alongside it.
__device__ __forceinline__ uint64_t wm_step(const FmiDev &ix, uint32_t k, uint64_t p, uint32_t d)
{
    const uint32_t *w = wm_block_ptr(ix, k, p >> 6);
    const uint64_t base = ix.dbase_tab[k * FMI_ARITY + d];
    const uint4 pa = *reinterpret_cast<const uint4 *>(w + 20), pb = *reinterpret_cast<const uint4 *>(w + 24);
    const uint32_t lo = w[d];
    const uint32_t hi = reinterpret_cast<const uint8_t *>(w)[64 + d];
    const uint64_t P0 = (uint64_t)pa.x | ((uint64_t)pa.y << 32), P1 = (uint64_t)pa.z | ((uint64_t)pa.w << 32);
    const uint64_t P2 = (uint64_t)pb.x | ((uint64_t)pb.y << 32), P3 = (uint64_t)pb.z | ((uint64_t)pb.w << 32);
    uint64_t m = (1ull << (p & 63)) - 1;
    m &= (d & 1) ? P0 : ~P0;
    m &= (d & 2) ? P1 : ~P1;
    m &= (d & 4) ? P2 : ~P2;
    m &= (d & 8) ? P3 : ~P3;
    return base + ((uint64_t)lo | ((uint64_t)hi << 32)) + (uint64_t)__popcll(m);
}

__device__ __forceinline__ uint32_t wm_digit(const FmiDev &ix, uint64_t c, uint32_t k)
{
    return (uint32_t)(c >> (FMI_DIGIT_BITS * (ix.dlevels - 1 - k))) & (FMI_ARITY - 1);
}

// number of occurrences of symbol c in BWT[0, i), 0 <= i <= n
__device__ __forceinline__ uint64_t wm_rank_sym(const FmiDev &ix, uint64_t c, uint64_t i, uint64_t *probes)
{
    uint64_t p = i;
    for (uint32_t k = 0; k < ix.dlevels; k++) p = wm_step(ix, k, p, wm_digit(ix, c, k));
    if (probes) *probes += ix.dlevels;
    return p - ix.leaf[c];
}

// rank_c at two positions at once (the two ends of a backward-search interval): the loads of both
// walks are issued back to back, one dependent chain instead of two.
__device__ __forceinline__ void wm_rank_sym_pair(const FmiDev &ix, uint64_t c, uint64_t i, uint64_t j, uint64_t &ri, uint64_t &rj,
                                                 uint64_t *probes)
{
    uint64_t p = i, s = j;
    for (uint32_t k = 0; k < ix.dlevels; k++) {
        const uint32_t d = wm_digit(ix, c, k);
        const uint64_t p2 = wm_step(ix, k, p, d), s2 = wm_step(ix, k, s, d);
        if (probes) *probes += ((p >> 6) == (s >> 6)) ? 1 : 2;
        p = p2; s = s2;
    }
    ri = p - ix.leaf[c]; rj = s - ix.leaf[c];
}
