// Device-side primitives shared by the query kernels and the GPU builder.
#pragma once
#include <hip/hip_runtime.h>

#include "fmi_internal.h"

// ---------------------------------------------------------------------------
// device primitives over the hex wavelet matrix (layout: fmi_internal.h)
// ---------------------------------------------------------------------------

// Index arrays are immutable while a query kernel runs.  Reading them through the constant address space tells
// the compiler so: a load whose address is wave-uniform (the row's own probes in k_constrain: prefix step, root
// child) becomes a scalar load and everything computed from it scalar ALU -- it no longer takes VALU issue slots
// from the other three waves of the SIMD; a divergent address still compiles to the same global_load.
template <typename T> using cptr = const T __attribute__((address_space(4))) *;
template <typename T> __device__ __forceinline__ cptr<T> as_const(const T *p) { return (cptr<T>)(uintptr_t)p; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// tools (fmi_dev_debug_marks / fmi_dev_mark): a progress word in host-visible memory, written in stream order
static __global__ void k_mark(volatile uint32_t *p, uint32_t v) { *p = v; __threadfence_system(); }

// One 128-byte block in registers: 8 x global_load_dwordx4, every byte used.
struct HBlock {
    uint32_t rel[16];    // digits equal to d between the superblock start and the block
    uint32_t P[4][4];    // bit planes, 128 bits each
};

__device__ __forceinline__ const uint32_t *wm_block_ptr(const FmiDev &ix, uint32_t k, uint64_t blk)
{
    return reinterpret_cast<const uint32_t *>(ix.wm + ((uint64_t)k * ix.nblk + blk) * FMI_BLOCK_WORDS);
}

__device__ __forceinline__ void wm_load_block(const FmiDev &ix, uint32_t k, uint64_t blk, HBlock &b)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(wm_block_ptr(ix, k, blk));
    const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6], v7 = src[7];
    b.rel[0] = v0.x; b.rel[1] = v0.y; b.rel[2] = v0.z; b.rel[3] = v0.w;
    b.rel[4] = v1.x; b.rel[5] = v1.y; b.rel[6] = v1.z; b.rel[7] = v1.w;
    b.rel[8] = v2.x; b.rel[9] = v2.y; b.rel[10] = v2.z; b.rel[11] = v2.w;
    b.rel[12] = v3.x; b.rel[13] = v3.y; b.rel[14] = v3.z; b.rel[15] = v3.w;
    b.P[0][0] = v4.x; b.P[0][1] = v4.y; b.P[0][2] = v4.z; b.P[0][3] = v4.w;
    b.P[1][0] = v5.x; b.P[1][1] = v5.y; b.P[1][2] = v5.z; b.P[1][3] = v5.w;
    b.P[2][0] = v6.x; b.P[2][1] = v6.y; b.P[2][2] = v6.z; b.P[2][3] = v6.w;
    b.P[3][0] = v7.x; b.P[3][1] = v7.y; b.P[3][2] = v7.z; b.P[3][3] = v7.w;
}

// mask of the first `bit` (0..127) positions of a block, dword w
__device__ __forceinline__ uint32_t wm_tail_word(uint32_t bit, uint32_t w)
{
    const int32_t s = (int32_t)bit - (int32_t)(32 * w);
    return s >= 32 ? ~0u : (s <= 0 ? 0u : ((1u << s) - 1));
}

// digits equal to d in [superblock start, 128 * block + bit), for all sixteen d
__device__ __forceinline__ void wm_block_ranks(const HBlock &b, uint32_t bit, uint32_t (&r)[16])
{
#pragma unroll
    for (uint32_t d = 0; d < 16; d++) r[d] = b.rel[d];
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t T = wm_tail_word(bit, w);
        const uint32_t p0 = b.P[0][w], p1 = b.P[1][w], p2 = b.P[2][w], p3 = b.P[3][w];
        const uint32_t a[4] = {~p3 & ~p2, ~p3 & p2, p3 & ~p2, p3 & p2};
        const uint32_t c[4] = {~p1 & ~p0 & T, ~p1 & p0 & T, p1 & ~p0 & T, p1 & p0 & T};
#pragma unroll
        for (uint32_t d = 0; d < 16; d++) r[d] += (uint32_t)__popc(a[d >> 2] & c[d & 3]);
    }
}

// where position p of level k goes in level k+1 if its symbol has digit d there (p = n maps an
// exclusive upper bound): ONE 128-byte line, of which a single digit needs the four plane chunks and
// one counter dword, plus its word of the superblock row (L2-resident; the address depends on p
// only, so that load is issued alongside the block's).
__device__ __forceinline__ uint64_t wm_step(const FmiDev &ix, uint32_t k, uint64_t p, uint32_t d)
{
    const uint64_t blk = p >> FMI_BLOCK_SHIFT;
    const cptr<uint32_t> w = as_const(wm_block_ptr(ix, k, blk));
    const uint64_t base = as_const(ix.sbase)[((uint64_t)k * ix.nsb + (blk >> ix.sb_shift)) * FMI_ARITY + d];
    const cptr<u32x4> wq = (cptr<u32x4>)(w + 16);
    const u32x4 q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3];
    const uint32_t rel = w[d];
    const uint32_t bit = (uint32_t)p & (FMI_BLOCK_BITS - 1);
    const uint32_t P0[4] = {q0.x, q0.y, q0.z, q0.w}, P1[4] = {q1.x, q1.y, q1.z, q1.w};
    const uint32_t P2[4] = {q2.x, q2.y, q2.z, q2.w}, P3[4] = {q3.x, q3.y, q3.z, q3.w};
    uint32_t cnt = rel;
#pragma unroll
    for (uint32_t x = 0; x < 4; x++) {
        uint32_t m = wm_tail_word(bit, x);
        m &= (d & 1) ? P0[x] : ~P0[x];
        m &= (d & 2) ? P1[x] : ~P1[x];
        m &= (d & 4) ? P2[x] : ~P2[x];
        m &= (d & 8) ? P3[x] : ~P3[x];
        cnt += (uint32_t)__popc(m);
    }
    return base + cnt;
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
    return p - as_const(ix.leaf)[c];
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
        if (probes) *probes += ((p >> FMI_BLOCK_SHIFT) == (s >> FMI_BLOCK_SHIFT)) ? 1 : 2;
        p = p2; s = s2;
    }
    const uint64_t lf = as_const(ix.leaf)[c];
    ri = p - lf; rj = s - lf;
}

// ---------------------------------------------------------------------------
// suffix array / text / document boundaries (all resident)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t sa_at(const FmiDev &ix, uint64_t row)
{
    uint64_t v = ix.sa_lo[row];
    if (ix.sa_hi) v |= (uint64_t)ix.sa_hi[row] << 32;
    return v;
}

// bisect_right(beginnings, pos) - 1
__device__ __forceinline__ uint64_t doc_of(const FmiDev &ix, uint64_t pos)
{
    uint64_t lo = 0, hi = ix.n_begin;
    if (ix.doc_hint && pos < ix.n) {
        // every boundary index <= doc_hint[b] is <= pos, every one > doc_hint[b + 1] is > pos: the same search, narrowed
        const uint64_t b = pos >> FMI_DOC_HINT_SHIFT;
        lo = (uint64_t)ix.doc_hint[b] + 1;
        hi = min((uint64_t)ix.doc_hint[b + 1] + 1, ix.n_begin);
    }
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (pos < ix.doc_begin[mid]) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

__device__ __forceinline__ uint64_t text_at(const FmiDev &ix, uint64_t p)
{
    return ix.sym_bytes == 2 ? (uint64_t)((const uint16_t *)ix.text)[p] : (uint64_t)((const uint32_t *)ix.text)[p];
}
