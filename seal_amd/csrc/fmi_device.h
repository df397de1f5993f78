// Device-side primitives shared by the query kernels and the GPU builder.
#pragma once
#include <hip/hip_runtime.h>

#include "fmi_internal.h"

// ---------------------------------------------------------------------------
// device primitives
// ---------------------------------------------------------------------------
struct alignas(16) U64x2 { uint64_t x, y; };

// ones in level k before position p (0 <= p <= n): ONE 64-byte line.
__device__ __forceinline__ uint64_t wm_rank1(const FmiDev &ix, uint32_t k, uint64_t p, uint64_t *probes)
{
    const uint64_t w = p >> 6;
    uint64_t blk;
    if (ix.n < (1ull << 37)) blk = (uint32_t)w / 7u;      // word index fits 32 bits: one mul_hi instead of a 64-bit divide
    else blk = w / 7;
    const uint32_t wi = (uint32_t)(w - blk * 7);
    const U64x2 *b = reinterpret_cast<const U64x2 *>(ix.wm + ((uint64_t)k * ix.nblk + blk) * FMI_BLOCK_WORDS);
    const U64x2 v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3];
    const uint64_t words[7] = {v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
    const uint64_t tail = (1ull << (p & 63)) - 1;
    uint64_t r = v0.x;
#pragma unroll
    for (uint32_t j = 0; j < 7; j++) {
        uint64_t m = (j < wi) ? ~0ull : ((j == wi) ? tail : 0ull);
        r += (uint64_t)__popcll(words[j] & m);
    }
    if (probes) ++*probes;
    return r;
}

// number of occurrences of symbol c in BWT[0, i), 0 <= i <= n
__device__ __forceinline__ uint64_t wm_rank_sym(const FmiDev &ix, uint64_t c, uint64_t i, uint64_t *probes)
{
    uint64_t p = i;
    for (uint32_t k = 0; k < ix.levels; k++) {
        const uint64_t r1 = wm_rank1(ix, k, p, probes);
        p = ((c >> (ix.levels - 1 - k)) & 1) ? ix.zeros[k] + r1 : p - r1;
    }
    return p - ix.leaf[c];
}

