// sealnn_hgemm_nt -- C[M][N] (fp32) = A[M][K] (fp16) x W[N][K]^T (fp16), fp32 accumulation on the gfx950 matrix cores.
//
// What it is for: the linear layers of the BART decode step at M = 300 .. 640 rows (the beams of a batch), run as ONE fp16 product over the
// three split planes of an fp32 operand (seal_amd/split_gemm.py: K = 3 x 1024 or 3 x 4096), reference seal/beam_search.py:231-253 (the
// model forward of every decode step).  hipBLASLt serves these shapes with stream-K kernels tuned for large problems: 14 - 43 us per
// product where the arithmetic is 2 - 6 us (profiles/r5_gemm_shape_table_library.txt), and two of them in flight on different streams can
// wait for each other for ever.  This kernel has no inter-workgroup hand-off at all.
//
// Shape of the kernel (MI355X guide, "canonical CDNA GEMM" + the M = 256 projection notes):
//   * a workgroup of 4 waves (2 x 2) owns a BM x BN tile of C; a wave owns (BM / 2) x (BN / 2) = FM x FN fragments of 16 x 16, each the
//     accumulator of v_mfma_f32_16x16x32_f16 (A fragment: lane l holds row l & 15, the 8 halves of k-chunk l >> 4; B fragment the same of
//     W's row = C's column; D: column l & 15, rows 4 (l >> 4) .. + 3);
//   * both operands are K-contiguous, so a K step of 64 is one 128-byte line per tile row: tiles go global -> LDS with
//     global_load_lds_dwordx4 (16 B per lane, no staging registers), 8 rows per wave instruction, the 16-byte chunks of a row XOR-swizzled
//     with bits 1..3 of the row ON THE SOURCE ADDRESS (the LDS image of an LDS-DMA is lane-linear), so that the sixteen lanes of every
//     ds_read_b128 lane group hit sixteen different 16-byte slots of the 256-byte bank row;
//   * NS LDS stages (2 by default; 3 measured no faster): NS - 1 K steps are in flight while one is computed -- at these heights a workgroup is bound by the latency
//     of its loads, not by their bandwidth --, waited for with a COUNTED s_waitcnt vmcnt and ONE raw s_barrier per K step (never
//     __syncthreads() while an LDS-DMA is in flight: it would drain the prefetch); ALL of the LDS is one array;
//   * rows beyond M / N are clamped on load (they read a valid row) and not stored.
// K must be a multiple of 64 and the operands 16-byte aligned.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "../../include/sealnn.h"
#include "fmi_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t BK = 64;                 // halves per K step = one 128-byte line per tile row
constexpr uint32_t ROW_BYTES = BK * 2;

__device__ __forceinline__ uint32_t swz(uint32_t row, uint32_t chunk) { return chunk ^ ((row >> 1) & 7u); }

// one tile (R rows x 128 B) of a K step, global -> LDS: wave w issues the 8-row pieces w, w + 4, ...
template <uint32_t R, uint32_t AUX = 0>
__device__ __forceinline__ void load_tile(const _Float16 *__restrict__ src, uint32_t row0, uint32_t row_max, uint64_t ld, uint32_t k0,
                                          unsigned char *lds_tile, uint32_t wave, uint32_t lane)
{
#pragma unroll
    for (uint32_t piece = 0; piece < R / 8; piece += 4) {
        const uint32_t p = piece + wave;                       // (R / 8 is a multiple of 4 for R = 64, 128)
        const uint32_t row = p * 8 + (lane >> 3);              // tile row this lane fills
        const uint32_t chunk = swz(row, lane & 7u);            // the logical chunk that lives at physical slot lane & 7 of that row
        uint32_t gr = row0 + row;
        gr = gr < row_max ? gr : row_max - 1;
        const _Float16 *g = src + (uint64_t)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(lds_tile + p * 8 * ROW_BYTES), 16, 0, AUX);
    }
}

__device__ __forceinline__ half8 read_frag(const unsigned char *lds_tile, uint32_t row, uint32_t chunk)
{
    return *reinterpret_cast<const half8 *>(lds_tile + row * ROW_BYTES + swz(row, chunk) * 16);
}

template <uint32_t N>
__device__ __forceinline__ void wait_all_but()      // all LDS-DMA but the newest N have landed (vmcnt is an immediate)
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS = LDS stages: NS - 1 K steps are in flight while one is computed.  NS = 1: one stage, __syncthreads() (the reference form of the tests).
// KG = K groups: KG x 4 waves per workgroup; group g takes the K steps g, g + KG, ... of the tile through LDS stages of its own, and the
// groups' accumulators are summed through the LDS at the end, in group order (deterministic).  A skinny product (a few hundred rows, a
// long K) has too few C tiles for 256 CUs and a 4-wave workgroup waits for the latency of every K step alone: K groups put 8 or 16 waves
// on the CU that overlap one another's waits -- split-K without slabs in memory or a second kernel.
// PAIRS: the operands are the hi / lo planes of a split fp32 operand INTERLEAVED per 32 columns -- a 128-byte line of a row = [hi of 32 columns | lo * 2^11
// (A) resp. lo (W) of the same 32 columns] (sealnn_*_pairs write A, seal_amd/split_gemm.py W) -- and a K step of 64 halves is the three products of
// those 32 columns: hi.hi + hi.lo into one accumulator, (lo 2^11).hi into a second one that joins the first times 2^-11 at the end.  The three-block
// layout [hi | hi | lo 2^11] x [hi | lo | hi 2^-11] feeds the matrix cores the same three products from six tiles of which two are copies: here four
// tiles travel, two thirds of the bytes through the load path that bounds these products (and of the LDS reads), for the same MFMAs.
template <uint32_t BM, uint32_t BN, uint32_t NS, uint32_t KG, bool PAIRS = false>
__global__ __launch_bounds__(256 * KG) void k_hgemm_nt(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, float *__restrict__ C,
                                                  uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t k_per_slice, uint64_t slab_stride,
                                                  uint32_t m_fastest)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr uint32_t TM = BM / 2, TN = BN / 2, FM = TM / 16, FN = TN / 16;
    constexpr uint32_t STAGE = (BM + BN) * ROW_BYTES;
    constexpr uint32_t NL = (BM + BN) / 32;                    // LDS-DMA instructions per wave and stage
    static_assert((NS - 1) * NL <= 48, "vmcnt holds 6 bits");
    static_assert(KG == 1 || (KG - 1) * BM * BN * 4 <= KG * NS * STAGE, "the groups' partial tiles are summed through the stage buffers");
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t grp = wave_all >> 2, wave = wave_all & 3;
    const uint32_t wm = wave >> 1, wn = wave & 1;
    uint32_t tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (m_fastest) {
        // A wide-N product (lm_head: 393 column tiles x 5 row tiles) reads each W tile once per ROW tile; with the column tile as the fastest
        // grid index those five workgroups are 393 launches apart and W comes from memory five times (1.5 GB for a 309 MB matrix).  Here the
        // grid is one-dimensional, the row tile runs fastest, and the index is first remapped so that consecutive logical tiles land on ONE XCD
        // (workgroup i runs on XCD i % 8): the row tiles of a column tile then share that XCD's L2 copy of the W tile.
        const uint32_t tiles_m = (M + BM - 1) / BM, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7u;
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const uint32_t logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tile_n = logical / tiles_m; tile_m = logical - tile_n * tiles_m;
    }
    const uint32_t m0 = tile_m * BM, n0 = tile_n * BN;
    const uint32_t kbeg = blockIdx.z * k_per_slice + grp * BK;       // this group's first K step; its steps are KG * BK apart
    const uint32_t nk = k_per_slice / (BK * KG);
    unsigned char *const lds_grp = lds + grp * NS * STAGE;
    C += (uint64_t)blockIdx.z * slab_stride;

    float4v acc[FM][FN];
    float4v acc2[PAIRS ? FM : 1][PAIRS ? FN : 1];             // (PAIRS) the (lo 2^11) . hi products
#pragma unroll
    for (uint32_t i = 0; i < FM; i++)
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) {
            acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
            if constexpr (PAIRS) acc2[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
        }

    auto issue = [&](uint32_t kt, uint32_t buf) {
        unsigned char *st = lds_grp + buf * STAGE;
        load_tile<BM>(A, m0, M, K, kbeg + kt * BK * KG, st, wave, lane);
        load_tile<BN>(W, n0, N, K, kbeg + kt * BK * KG, st + BM * ROW_BYTES, wave, lane);     // (NOT non-temporal here: five to ten row tiles re-read a W tile through the L2 -- 11.5 -> 13.3 us on the d x d products with the hint)
    };
    auto compute = [&](uint32_t buf) {
        const unsigned char *ta = lds_grp + buf * STAGE, *tb = ta + BM * ROW_BYTES;
        // the fragments of BOTH halves of the K step are asked for before the first MFMA: the second half's reads land behind the first's math
        half8 fa[2][FM], fb[2][FN];
#pragma unroll
        for (uint32_t ks = 0; ks < 2; ks++) {
#pragma unroll
            for (uint32_t i = 0; i < FM; i++) fa[ks][i] = read_frag(ta, wm * TM + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) fb[ks][j] = read_frag(tb, wn * TN + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
        }
        if constexpr (PAIRS) {
#pragma unroll
            for (uint32_t i = 0; i < FM; i++)
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][i], fb[0][j], acc2[i][j], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (uint32_t ks = 0; ks < 2; ks++)
#pragma unroll
                for (uint32_t i = 0; i < FM; i++)
#pragma unroll
                    for (uint32_t j = 0; j < FN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (NS >= 2) {
        // prologue: NS - 1 stages in flight
#pragma unroll
        for (uint32_t s = 0; s + 1 < NS; s++)
            if (s < nk) issue(s, s);
        uint32_t buf = 0;                                    // kt % NS
        for (uint32_t kt = 0; kt < nk; kt++) {
            // stage kt has landed once all but the stages issued after it have: min(NS - 2, nk - 1 - kt) of them
            const uint32_t newer = nk - 1 - kt < NS - 2 ? nk - 1 - kt : NS - 2;
            if (newer == 0) wait_all_but<0>();
            else if (newer == 1) wait_all_but<NL>();
            else if (newer == 2) wait_all_but<2 * NL>();
            else wait_all_but<3 * NL>();
            // ONE barrier per K step: every wave's share of stage kt is in the LDS, and every wave is done reading stage kt - 1 (its
            // ds_reads completed before its last MFMAs were issued), whose buffer the next LDS-DMA refills
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < nk) issue(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
            compute(buf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
    } else {
        for (uint32_t kt = 0; kt < nk; kt++) {
            issue(kt, 0);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    if constexpr (PAIRS) {
#pragma unroll
        for (uint32_t i = 0; i < FM; i++)
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) acc[i][j] += acc2[i][j] * 0x1p-11f;
    }
    if constexpr (KG > 1) {
        // the groups' partial tiles -> group 0, through the (now idle) stage buffers: group g > 0 parks its accumulators as [fragment][lane]
        // float4 (conflict-free 16-byte stores), group 0 adds them in group order
        __syncthreads();
        float4v *park = reinterpret_cast<float4v *>(lds);
        constexpr uint32_t PER_GROUP = 4 * FM * FN * 64;                 // float4 slots of one group: 4 waves x fragments x lanes
        if (grp > 0) {
#pragma unroll
            for (uint32_t i = 0; i < FM; i++)
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) park[(grp - 1) * PER_GROUP + ((wave * FM + i) * FN + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (uint32_t g = 1; g < KG; g++)
#pragma unroll
            for (uint32_t i = 0; i < FM; i++)
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) acc[i][j] += park[(g - 1) * PER_GROUP + ((wave * FM + i) * FN + j) * 64 + lane];
    }
    // D fragment: column lane & 15, rows 4 (lane >> 4) .. + 3
    const bool inside = m0 + BM <= M && n0 + BN <= N;       // (uniform: interior tiles store without bounds tests)
#pragma unroll
    for (uint32_t i = 0; i < FM; i++) {
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) {
            const uint32_t col = n0 + wn * TN + j * 16 + (lane & 15);
            const uint32_t rbase = m0 + wm * TM + i * 16 + (lane >> 4) * 4;
            float *dst = C + (uint64_t)rbase * ldc + col;
            if (inside) {
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) dst[(uint64_t)r * ldc] = acc[i][j][r];
            } else {
#pragma unroll
                for (uint32_t r = 0; r < 4; r++)
                    if (rbase + r < M && col < N) dst[(uint64_t)r * ldc] = acc[i][j][r];
            }
        }
    }
}

// ---- the tall tile: BM = 320 rows (a decode step's 600 rows are TWO row tiles, its 300 rows one) x BN columns, 8 waves ----
// What bounds the 4-wave kernel above at these heights is the CU's load path, not the matrix cores and not HBM: one CU takes ~92 GB/s from the L2
// into its LDS (38 B/clk) whatever the depth of the prefetch, the number of CUs pulling or the form of the load (LDS-DMA or registers:
// tools/glds_rate.hip, profiles/r6_glds_rate.txt), and the part of a stage that comes from HBM -- W, which a decode step streams once -- at
// about a third of that, with everything the wave issued behind it waiting (loads return in order).  Every product of a step fits
// time = bytes through that path / CUs: qkv as 64 x 128 tiles moves 265 MB, the d x d projections as 64 x 64 126 MB, lm_head as 128 x 128 3.1 GB.
// So the tile grows in the only direction these products have left: BM = 320 cuts the bytes per flop to (1/320 + 1/BN) / (1/64 + 1/BN), eight
// waves (4 along M x 2 along N, 80 x BN/2 each: 5 x BN/32 accumulator fragments) put two waves on every SIMD, and the pieces of the K step after
// next are issued between the MFMA rows of this one.  One workgroup per CU; the stages decide BN: 3 stages of (320 + 96) x 128 B are 156 KB of the
// 160, 3 of 320 + 128 do not fit and 2 are not enough (profiles/r6_hgemm_probe_tall.txt).  The grid is one-dimensional, XCD-grouped (consecutive
// logical tiles on one XCD), row tile fastest, then column tile, then K slice: the two row tiles of a W tile share its L2 copy.
// Where a step's cycles go (DBG = 3, s_memtime: profiles/r6_hgemm_tall_step_timeline.txt): the vmcnt wait at its top is 20 - 200 cycles -- the
// stages ARE there in time --, the waves need 1000 - 1550 cycles for 14 fragment reads, 20 MFMAs (340 cycles) and 6 pieces because each piece
// waits for the queue of the load path, and wave 0 then waits 500 cycles at the barrier for wave 7.  Without LDS-DMA a step takes 1200 cycles,
// without reads and MFMAs 1630 (profiles/r6_hgemm_tall_decomposition.txt): the two overlap, and it is the load path that is left.
constexpr uint32_t TALL_BM = 320;
constexpr bool NT_W = true;               // the weights' LDS-DMA pieces carry the non-temporal hint

template <uint32_t BN, uint32_t NS, uint32_t DBG = 0, bool PAIRS = false>      // PAIRS: see k_hgemm_nt;  DBG (probes only): 1 = no LDS-DMA after the prologue, 2 = no fragment reads / MFMAs
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_hgemm_tall(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, float *__restrict__ C,
                                                    uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t k_per_slice, uint64_t slab_stride,
                                                    uint32_t tiles_m, uint32_t tiles_n, const float *__restrict__ ep_bias, uint32_t ep_group, float ep_alpha)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr uint32_t BM = TALL_BM, TM = BM / 4, TN = BN / 2, FM = TM / 16, FN = TN / 16;
    constexpr uint32_t STAGE = (BM + BN) * ROW_BYTES;
    constexpr uint32_t PA = BM / 64, PW = (BN / 8 + 7) / 8, NL = PA + PW;       // LDS-DMA pieces (8 rows x 128 B) per wave and stage: A's, W's
    constexpr uint32_t LAST_W = (BN / 8) % 8;                  // != 0: only waves < LAST_W have a last W piece (BN = 96: twelve pieces over eight waves)
    static_assert(NS >= 2 && NS * STAGE <= 160 * 1024, "the stages must fit the LDS of one CU");
    static_assert((NS - 1) * NL <= 48, "vmcnt holds 6 bits");
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wm = wave >> 1, wn = wave & 1;
    // logical tile of this workgroup: XCD x owns a contiguous range of logical indices (workgroup i runs on XCD i % 8)
    const uint32_t nwg = gridDim.x, q = nwg >> 3, r8 = nwg & 7u;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    uint32_t logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;
    const uint32_t tile_m = logical % tiles_m;
    logical /= tiles_m;
    const uint32_t tile_n = logical % tiles_n, slice = logical / tiles_n;
    const uint32_t m0 = tile_m * BM, n0 = tile_n * BN;
    const uint32_t nk = k_per_slice / BK;
    float *const dbg_out = C;
    const uint64_t t_begin = DBG == 3 ? __builtin_amdgcn_s_memtime() : 0;
    C += (uint64_t)slice * slab_stride;

    // source of every piece this lane issues, at the slice's first K step (a K step later: + 128 bytes).  Piece p of a tile = its rows
    // 8 p .. 8 p + 7; wave w issues pieces w, w + 8, ...; lane l fills row 8 p + (l >> 3), physical 16-byte slot l & 7 = logical chunk
    // (l & 7) ^ ((row >> 1) & 7).  Rows beyond M / N read the last valid row (and are not stored).
    const _Float16 *src[NL];
#pragma unroll
    for (uint32_t i = 0; i < PA; i++) {
        const uint32_t row = (i * 8 + wave) * 8 + (lane >> 3);
        uint32_t gr = m0 + row;
        gr = gr < M ? gr : M - 1;
        src[PW + i] = A + (uint64_t)gr * K + (uint64_t)slice * k_per_slice + swz(row, lane & 7u) * 8;
    }
#pragma unroll
    for (uint32_t i = 0; i < PW; i++) {
        const uint32_t row = ((i * 8 + wave) * 8 + (lane >> 3)) % BN;         // (% BN: the piece a wave >= LAST_W does not have; never issued)
        uint32_t gr = n0 + row;
        gr = gr < N ? gr : N - 1;
        src[i] = W + (uint64_t)gr * K + (uint64_t)slice * k_per_slice + swz(row, lane & 7u) * 8;
    }
    // piece x (0 .. NL - 1; W's first: they come from memory, A's from the L2) of K step kt into stage buffer buf
    auto piece = [&](uint32_t x, uint32_t kt, uint32_t buf) {
        unsigned char *st = lds + buf * STAGE;
        unsigned char *dst = x < PW ? st + BM * ROW_BYTES + (x * 8 + wave) * 8 * ROW_BYTES : st + ((x - PW) * 8 + wave) * 8 * ROW_BYTES;
        if (LAST_W != 0 && x == PW - 1 && wave >= LAST_W) return;
        // W's pieces non-temporal (aux = 2): a decode step streams each weight once; 4 .. 11 % on fc1 / fc2 / lm_head with W from memory, the rest level
        // (MI355X guide, nt-weights: issued -> landed -18 % for weights that one CU reads once)
        if (NT_W && x < PW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[x] + (uint64_t)kt * BK),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[x] + (uint64_t)kt * BK),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };

    float4v acc[FM][FN];
    float4v acc2[PAIRS ? FM : 1][PAIRS ? FN : 1];             // (PAIRS) the (lo 2^11) . hi products
#pragma unroll
    for (uint32_t i = 0; i < FM; i++)
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) {
            acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
            if constexpr (PAIRS) acc2[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
        }

    // prologue: NS - 1 stages in flight
#pragma unroll
    for (uint32_t s = 0; s + 1 < NS; s++)
        if (s < nk) {
#pragma unroll
            for (uint32_t x = 0; x < NL; x++) piece(x, s, s);
        }
    // one K step: MORE = a further stage (kt + NS - 1) exists and is issued between the MFMA rows; NEWER = stages issued after kt and still in flight.
    // The fragment reads are inline assembly with waits counted by hand: left to itself the compiler (which must assume that an LDS-DMA and a
    // ds_read alias) delays the reads to save registers and with them every piece of the next stage, to the END of the step, where the DMA
    // has no MFMAs left to hide behind and the next step waits for it.  Here all 2 (FM + FN) reads go out behind the barrier, the first half's
    // MFMA rows start when lgkmcnt says its fragments are in, and a piece goes out in front of each row.
    uint32_t buf = 0;
    const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)lds);
    const uint32_t frag_swz = (lane >> 4) ^ ((lane >> 1) & 7u);               // chunk (lane >> 4) of K half 0 at row lane & 15; K half 1: ^ 4
    const uint32_t a_off0 = lds_base + (wm * TM + (lane & 15)) * ROW_BYTES + frag_swz * 16, a_off1 = a_off0 ^ 64u;
    const uint32_t b_off0 = lds_base + (BM + wn * TN + (lane & 15)) * ROW_BYTES + frag_swz * 16, b_off1 = b_off0 ^ 64u;
    auto step = [&](uint32_t kt, auto more_tag, auto newer_tag) {
        constexpr bool MORE = decltype(more_tag)::value && DBG != 1;
        constexpr uint32_t NEWER = decltype(newer_tag)::value;
        uint64_t t0 = 0, t1 = 0;
        if constexpr (DBG == 3) t0 = __builtin_amdgcn_s_memtime();
        if (LAST_W != 0 && NEWER != 0 && wave >= LAST_W) wait_all_but<NEWER * (NL - 1)>();
        else wait_all_but<NEWER * NL>();
        if constexpr (DBG == 3) t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (DBG == 3) {
            // (probe) waves 0 and 7 of every workgroup: cycles since the kernel began at the top of the step, after its wait, after the barrier
            const uint64_t t2 = __builtin_amdgcn_s_memtime();
            if ((wave == 0 || wave == 7) && lane == 0 && kt < 60) {
                float *o = dbg_out + ((uint64_t)blockIdx.x * 2 + (wave == 7)) * 192 + kt * 3;
                o[0] = (float)(t0 - t_begin); o[1] = (float)(t1 - t_begin); o[2] = (float)(t2 - t_begin);
            }
        }
        const uint32_t nxt = kt + NS - 1, nbuf = buf == 0 ? NS - 1 : buf - 1;
        const uint32_t va0 = a_off0 + buf * STAGE, va1 = a_off1 + buf * STAGE, vb0 = b_off0 + buf * STAGE, vb1 = b_off1 + buf * STAGE;
        auto issue_all = [&]() {
            if constexpr (MORE) {
#pragma unroll
                for (uint32_t x = 0; x < NL; x++) piece(x, nxt, nbuf);
            }
        };
        auto compute = [&](auto interleave_tag) {
            constexpr bool INTERLEAVE = decltype(interleave_tag)::value;
            half8 fa[2][FM], fb[2][FN];
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0][j]) : "v"(vb0), "n"(j * 16 * ROW_BYTES) : "memory");
#pragma unroll
            for (uint32_t i = 0; i < FM; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[0][i]) : "v"(va0), "n"(i * 16 * ROW_BYTES) : "memory");
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[1][j]) : "v"(vb1), "n"(j * 16 * ROW_BYTES) : "memory");
#pragma unroll
            for (uint32_t i = 0; i < FM; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[1][i]) : "v"(va1), "n"(i * 16 * ROW_BYTES) : "memory");
            static_assert(NL <= 2 * FM, "a piece per MFMA row");
            if constexpr (PAIRS) {
                // three products of the step's 32 columns: hi . hi as soon as the hi fragments are in, then hi . lo and (lo 2^11) . hi
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FM + FN) : "memory");
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) asm volatile("" : "+v"(fb[0][j]));
#pragma unroll
                for (uint32_t i = 0; i < FM; i++) asm volatile("" : "+v"(fa[0][i]));
#pragma unroll
                for (uint32_t i = 0; i < FM; i++) {
                    if constexpr (MORE && INTERLEAVE) {
                        if (i < NL) piece(i, nxt, nbuf);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < FN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) asm volatile("" : "+v"(fb[1][j]));
#pragma unroll
                for (uint32_t i = 0; i < FM; i++) asm volatile("" : "+v"(fa[1][i]));
#pragma unroll
                for (uint32_t i = 0; i < FM; i++) {
                    if constexpr (MORE && INTERLEAVE) {
                        if (FM + i < NL) piece(FM + i, nxt, nbuf);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < FN; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][i], fb[0][j], acc2[i][j], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
            for (uint32_t ks = 0; ks < 2; ks++) {
                // the K half's fragments are in: all reads but the other half's (ks = 0) / all (ks = 1); tied to the registers the MFMAs read
                if (ks == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FM + FN) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) asm volatile("" : "+v"(fb[ks][j]));
#pragma unroll
                for (uint32_t i = 0; i < FM; i++) asm volatile("" : "+v"(fa[ks][i]));
#pragma unroll
                for (uint32_t i = 0; i < FM; i++) {
                    if constexpr (MORE && INTERLEAVE) {
                        if (ks * FM + i < NL) piece(ks * FM + i, nxt, nbuf);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < FN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
                }
            }
            }
        };
        if constexpr (DBG == 2) issue_all();
        else compute(std::true_type{});
        // (tried and dropped, profiles/r6_hgemm_tall_step_timeline_skewed.txt: the two waves of a SIMD taking turns -- waves 0..3 issue all their
        //  pieces first and compute after, waves 4..7 the other way round -- instead of a piece between everybody's MFMA rows: no faster)
        buf = buf + 1 == NS ? 0 : buf + 1;
    };
    using T = std::true_type;
    using F = std::false_type;
    uint32_t kt = 0;
    for (; kt + NS - 1 < nk; kt++) step(kt, T{}, std::integral_constant<uint32_t, NS - 2>{});
    if constexpr (NS == 3) {
        if (kt + 1 < nk) { step(kt, F{}, std::integral_constant<uint32_t, 1>{}); kt++; }
    }
    if (kt < nk) step(kt, F{}, std::integral_constant<uint32_t, 0>{});

    if constexpr (PAIRS) {
#pragma unroll
        for (uint32_t i = 0; i < FM; i++)
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) acc[i][j] += acc2[i][j] * 0x1p-11f;
    }
    if constexpr (DBG == 3) {
        float keep = 0.f;
#pragma unroll
        for (uint32_t i = 0; i < FM; i++)
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) keep += acc[i][j][0];
        if ((wave == 0 || wave == 7) && lane == 0) {
            float *o = dbg_out + ((uint64_t)blockIdx.x * 2 + (wave == 7)) * 192;
            o[180] = (float)(__builtin_amdgcn_s_memtime() - t_begin); o[181] = keep; o[182] = (float)nk;
        }
        return;
    }
    const bool inside = m0 + BM <= M && n0 + BN <= N;
    if (ep_bias) {
        // the product FINISHED in the store (sealnn_hgemm_nt_ep; one slab): C = ep_alpha * acc + ep_bias[row / ep_group][col] -- the output projection of a
        // decode step with final_logits_bias + the per-query logit bias of the row's query (ep_group = beams), instead of a pass over the [rows, vocab] logits
#pragma unroll
        for (uint32_t i = 0; i < FM; i++) {
            const uint32_t rbase = m0 + wm * TM + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t row = rbase + r;
                if (row >= M) continue;
                const float *brow = ep_bias + (uint64_t)(row / ep_group) * N;
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) {
                    const uint32_t col = n0 + wn * TN + j * 16 + (lane & 15);
                    if (col < N) C[(uint64_t)row * ldc + col] = ep_alpha * acc[i][j][r] + brow[col];
                }
            }
        }
        return;
    }
#pragma unroll
    for (uint32_t i = 0; i < FM; i++) {
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) {
            const uint32_t col = n0 + wn * TN + j * 16 + (lane & 15);
            const uint32_t rbase = m0 + wm * TM + i * 16 + (lane >> 4) * 4;
            float *dst = C + (uint64_t)rbase * ldc + col;
            if (inside) {
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) dst[(uint64_t)r * ldc] = acc[i][j][r];
            } else {
#pragma unroll
                for (uint32_t r = 0; r < 4; r++)
                    if (rbase + r < M && col < N) dst[(uint64_t)r * ldc] = acc[i][j][r];
            }
        }
    }
}

struct TallEpilogue { const float *bias = nullptr; uint32_t group = 1; float alpha = 1.f; };

template <uint32_t BN, uint32_t NS, uint32_t DBG = 0, bool PAIRS = false>
int launch_tall(hipStream_t st, const void *A, const void *W, float *C, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t slices, TallEpilogue ep = TallEpilogue())
{
    const uint32_t tiles_m = (M + TALL_BM - 1) / TALL_BM, tiles_n = (N + BN - 1) / BN;
    const size_t lds = (size_t)NS * (TALL_BM + BN) * ROW_BYTES;
    (void)hipFuncSetAttribute((const void *)k_hgemm_tall<BN, NS, DBG, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_hgemm_tall<BN, NS, DBG, PAIRS>), dim3(tiles_m * tiles_n * slices), dim3(512), lds, st, (const _Float16 *)A, (const _Float16 *)W, C, M, N, K, ldc,
                       K / slices, (uint64_t)M * ldc, tiles_m, tiles_n, ep.bias, ep.group ? ep.group : 1u, ep.alpha);
    return hipGetLastError() == hipSuccess ? FMI_OK : FMI_ERR_HIP;
}

template <uint32_t BM, uint32_t BN, uint32_t NS, uint32_t KG, bool PAIRS>
int launch_cfg(hipStream_t st, const void *A, const void *W, float *C, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t slices, uint32_t m_fastest)
{
    const dim3 grid = m_fastest ? dim3(((N + BN - 1) / BN) * ((M + BM - 1) / BM), 1, slices) : dim3((N + BN - 1) / BN, (M + BM - 1) / BM, slices);
    const size_t lds = (size_t)KG * NS * (BM + BN) * ROW_BYTES;
    if ((K / slices) % (BK * KG)) { fmi_set_error("sealnn_hgemm_nt: %u K steps per slice do not split over %u K groups", K / slices / BK, KG); return FMI_ERR_ARG; }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_hgemm_nt<BM, BN, NS, KG, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_hgemm_nt<BM, BN, NS, KG, PAIRS>), grid, dim3(256 * KG), lds, st, (const _Float16 *)A, (const _Float16 *)W, C, M, N, K, ldc, K / slices,
                       (uint64_t)M * ldc, m_fastest);
    return hipGetLastError() == hipSuccess ? FMI_OK : FMI_ERR_HIP;
}

// (the configurations that are instantiated: every tile with 1..3 stages and one K group; the skinny-product forms -- 2 and 4 K groups --
//  for the two small tiles, two stages)
template <uint32_t BM, uint32_t BN, bool PAIRS>
int launch(hipStream_t st, const void *A, const void *W, float *C, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t slices, uint32_t stages,
           uint32_t kgroups, uint32_t mf)
{
    if (kgroups == 1) {
        switch (stages) {
        case 1: return launch_cfg<BM, BN, 1, 1, PAIRS>(st, A, W, C, M, N, K, ldc, slices, mf);
        case 2: return launch_cfg<BM, BN, 2, 1, PAIRS>(st, A, W, C, M, N, K, ldc, slices, mf);
        case 3: return launch_cfg<BM, BN, 3, 1, PAIRS>(st, A, W, C, M, N, K, ldc, slices, mf);
        default: break;
        }
    } else if constexpr (BM * BN <= 128 * 64) {
        if (stages == 2 && kgroups == 2) return launch_cfg<BM, BN, 2, 2, PAIRS>(st, A, W, C, M, N, K, ldc, slices, mf);
        if constexpr (BM * BN <= 64 * 64) {
            if (stages == 2 && kgroups == 4) return launch_cfg<BM, BN, 2, 4, PAIRS>(st, A, W, C, M, N, K, ldc, slices, mf);
        }
    }
    fmi_set_error("sealnn_hgemm_nt: no kernel for %u x %u tiles with %u stages and %u K groups", BM, BN, stages, kgroups);
    return FMI_ERR_ARG;
}

}   // namespace

// config: 0 = pick by shape; else tile (1 = 128 x 128, 2 = 64 x 64, 3 = 128 x 64, 4 = 64 x 128; + 128: row tiles fastest, XCD-grouped; 5 = 320 x 128,
// 6 = 320 x 64: the tall tiles of 8 waves) | stages << 8 (LDS stages 1..3; 0: two) |
// kgroups << 12 (K groups of 4 waves per workgroup: 1, 2 (tiles 2..4), 4 (tile 2); 0: one) | 1 << 29: the operands are hi / lo PAIRS (k_hgemm_nt) |
// slices << 16 (up to 8191; split-K over workgroups: slab s of C
// at C + s * M * ldc, the caller sums the slabs).  Probes and tests pass it explicitly.
static int hgemm_nt(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config, TallEpilogue ep)
{
    if (!a || !w || !c || M == 0 || N == 0) { fmi_set_error("sealnn_hgemm_nt: null / empty operand"); return FMI_ERR_ARG; }
    if (K == 0 || K % BK) { fmi_set_error("sealnn_hgemm_nt: K = %u must be a multiple of %u", K, BK); return FMI_ERR_UNSUPPORTED; }
    if (((uintptr_t)a | (uintptr_t)w) & 15) { fmi_set_error("sealnn_hgemm_nt: operands must be 16-byte aligned"); return FMI_ERR_ARG; }
    const uint32_t mf = (config >> 7) & 1u;          // bit 7 of the tile byte: row tile fastest + XCD remap (wide-N products)
    const bool pairs = (config >> 29) & 1u;          // the operands are hi / lo PAIRS per 32 columns (K = 2 x in_features): three products per K step
    uint32_t tile = config & 0x7f, stages = (config >> 8) & 0xf, kgroups = (config >> 12) & 0xf, slices = ((config >> 16) & 0x1fff) ? ((config >> 16) & 0x1fff) : 1;
    if (stages == 0) stages = 2;
    if (kgroups == 0) kgroups = 1;
    if ((K / BK) % slices) { fmi_set_error("sealnn_hgemm_nt: %u K steps do not split into %u slices", K / BK, slices); return FMI_ERR_ARG; }
    if (tile == 0) {
        // by shape (profiles/r5_hgemm_probe.txt): C tiles of 64 x 64; as many K groups as keep a CU's worth of waves busy when the tiles alone
        // do not fill the chip
        const uint64_t tiles = (uint64_t)((M + 63) / 64) * ((N + 63) / 64);
        tile = 2; stages = 2;
        kgroups = tiles >= 512 ? 1 : (tiles >= 256 ? 2 : 4);
        while (kgroups > 1 && (K / BK) % kgroups) kgroups >>= 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tile >= 5 && tile <= 7) {
        // the tall tile (320 x 128 / 64 / 96, 8 waves, always XCD-grouped); 2 stages, 3 where they fit the LDS (320 x 64 and 320 x 96)
        if (kgroups != 1 || stages < 2 || stages > (tile == 5 ? 2u : 3u)) {
            fmi_set_error("sealnn_hgemm_nt: no tall-tile kernel with %u stages and %u K groups", stages, kgroups);
            return FMI_ERR_ARG;
        }
        if (pairs) {
            if (tile == 5) return launch_tall<128, 2, 0, true>(st, a, w, c, M, N, K, ldc, slices, ep);
            if (tile == 7) return stages == 2 ? launch_tall<96, 2, 0, true>(st, a, w, c, M, N, K, ldc, slices, ep) : launch_tall<96, 3, 0, true>(st, a, w, c, M, N, K, ldc, slices, ep);
            return stages == 2 ? launch_tall<64, 2, 0, true>(st, a, w, c, M, N, K, ldc, slices, ep) : launch_tall<64, 3, 0, true>(st, a, w, c, M, N, K, ldc, slices, ep);
        }
        if (ep.bias) { fmi_set_error("sealnn_hgemm_nt_ep: the finishing store exists for the PAIRS form of the tall tiles"); return FMI_ERR_UNSUPPORTED; }
        if (tile == 5) return launch_tall<128, 2>(st, a, w, c, M, N, K, ldc, slices);
        // (config >> 30, probes only: 1 / 2 = the 320 x 64 kernel without its LDS-DMA / without its reads and MFMAs, 3 = step timestamps instead of C)
        if (tile == 7 && (config >> 30) == 3) return launch_tall<96, 3, 3>(st, a, w, c, M, N, K, ldc, slices);
        if (tile == 6 && (config >> 30) == 3) return launch_tall<64, 3, 3>(st, a, w, c, M, N, K, ldc, slices);
        if (tile == 7) return stages == 2 ? launch_tall<96, 2>(st, a, w, c, M, N, K, ldc, slices) : launch_tall<96, 3>(st, a, w, c, M, N, K, ldc, slices);
        if (config >> 30) return (config >> 30) == 1 ? launch_tall<64, 3, 1>(st, a, w, c, M, N, K, ldc, slices) : launch_tall<64, 3, 2>(st, a, w, c, M, N, K, ldc, slices);
        return stages == 2 ? launch_tall<64, 2>(st, a, w, c, M, N, K, ldc, slices) : launch_tall<64, 3>(st, a, w, c, M, N, K, ldc, slices);
    }
    if (pairs && ep.bias) { fmi_set_error("sealnn_hgemm_nt_ep: the finishing store exists for the PAIRS form of the tall tiles"); return FMI_ERR_UNSUPPORTED; }
    if (pairs) {
        switch (tile) {
        case 1: return launch<128, 128, true>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
        case 2: return launch<64, 64, true>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
        case 3: return launch<128, 64, true>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
        case 4: return launch<64, 128, true>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
        default: fmi_set_error("sealnn_hgemm_nt: unknown tile %u", tile); return FMI_ERR_ARG;
        }
    }
    if (ep.bias) { fmi_set_error("sealnn_hgemm_nt_ep: the finishing store exists for the PAIRS form of the tall tiles"); return FMI_ERR_UNSUPPORTED; }
    switch (tile) {
    case 1: return launch<128, 128, false>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    case 2: return launch<64, 64, false>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    case 3: return launch<128, 64, false>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    case 4: return launch<64, 128, false>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    default: fmi_set_error("sealnn_hgemm_nt: unknown tile %u", tile); return FMI_ERR_ARG;
    }
}

extern "C" int sealnn_hgemm_nt(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config)
{ return hgemm_nt(stream, a, w, c, M, N, K, ldc, config, TallEpilogue()); }

// the same product FINISHED in the kernel's store: C[row][col] = alpha * acc + bias[row / rows_per_bias_row][col] (bias [ceil(M / rows_per_bias_row)][N], fp32).
// One slab, PAIRS form, tall tiles (5..7): the output projection of a decode step (bias = final_logits_bias + the query's logit bias, rows_per_bias_row = beams).
extern "C" int sealnn_hgemm_nt_ep(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config,
                                  const float *bias, uint32_t rows_per_bias_row, float alpha)
{
    if (!bias || rows_per_bias_row == 0) { fmi_set_error("sealnn_hgemm_nt_ep: a bias and >= 1 rows per bias row"); return FMI_ERR_ARG; }
    if ((((config >> 16) & 0x1fff) ? ((config >> 16) & 0x1fff) : 1u) != 1u) { fmi_set_error("sealnn_hgemm_nt_ep: one slab only"); return FMI_ERR_ARG; }
    TallEpilogue ep;
    ep.bias = bias; ep.group = rows_per_bias_row; ep.alpha = alpha;
    return hgemm_nt(stream, a, w, c, M, N, K, ldc, config, ep);
}
