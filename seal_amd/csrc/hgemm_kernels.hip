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

#include "../../include/sealnn.h"
#include "fmi_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t BK = 64;                 // halves per K step = one 128-byte line per tile row
constexpr uint32_t ROW_BYTES = BK * 2;

__device__ __forceinline__ uint32_t swz(uint32_t row, uint32_t chunk) { return chunk ^ ((row >> 1) & 7u); }

// one tile (R rows x 128 B) of a K step, global -> LDS: wave w issues the 8-row pieces w, w + 4, ...
template <uint32_t R>
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
                                         (__attribute__((address_space(3))) void *)(lds_tile + p * 8 * ROW_BYTES), 16, 0, 0);
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
template <uint32_t BM, uint32_t BN, uint32_t NS, uint32_t KG>
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
#pragma unroll
    for (uint32_t i = 0; i < FM; i++)
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](uint32_t kt, uint32_t buf) {
        unsigned char *st = lds_grp + buf * STAGE;
        load_tile<BM>(A, m0, M, K, kbeg + kt * BK * KG, st, wave, lane);
        load_tile<BN>(W, n0, N, K, kbeg + kt * BK * KG, st + BM * ROW_BYTES, wave, lane);
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
#pragma unroll
        for (uint32_t ks = 0; ks < 2; ks++)
#pragma unroll
            for (uint32_t i = 0; i < FM; i++)
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
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

template <uint32_t BM, uint32_t BN, uint32_t NS, uint32_t KG>
int launch_cfg(hipStream_t st, const void *A, const void *W, float *C, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t slices, uint32_t m_fastest)
{
    const dim3 grid = m_fastest ? dim3(((N + BN - 1) / BN) * ((M + BM - 1) / BM), 1, slices) : dim3((N + BN - 1) / BN, (M + BM - 1) / BM, slices);
    const size_t lds = (size_t)KG * NS * (BM + BN) * ROW_BYTES;
    if ((K / slices) % (BK * KG)) { fmi_set_error("sealnn_hgemm_nt: %u K steps per slice do not split over %u K groups", K / slices / BK, KG); return FMI_ERR_ARG; }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_hgemm_nt<BM, BN, NS, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_hgemm_nt<BM, BN, NS, KG>), grid, dim3(256 * KG), lds, st, (const _Float16 *)A, (const _Float16 *)W, C, M, N, K, ldc, K / slices,
                       (uint64_t)M * ldc, m_fastest);
    return hipGetLastError() == hipSuccess ? FMI_OK : FMI_ERR_HIP;
}

// (the configurations that are instantiated: every tile with 1..3 stages and one K group; the skinny-product forms -- 2 and 4 K groups --
//  for the two small tiles, two stages)
template <uint32_t BM, uint32_t BN>
int launch(hipStream_t st, const void *A, const void *W, float *C, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t slices, uint32_t stages,
           uint32_t kgroups, uint32_t mf)
{
    if (kgroups == 1) {
        switch (stages) {
        case 1: return launch_cfg<BM, BN, 1, 1>(st, A, W, C, M, N, K, ldc, slices, mf);
        case 2: return launch_cfg<BM, BN, 2, 1>(st, A, W, C, M, N, K, ldc, slices, mf);
        case 3: return launch_cfg<BM, BN, 3, 1>(st, A, W, C, M, N, K, ldc, slices, mf);
        default: break;
        }
    } else if constexpr (BM * BN <= 128 * 64) {
        if (stages == 2 && kgroups == 2) return launch_cfg<BM, BN, 2, 2>(st, A, W, C, M, N, K, ldc, slices, mf);
        if constexpr (BM * BN <= 64 * 64) {
            if (stages == 2 && kgroups == 4) return launch_cfg<BM, BN, 2, 4>(st, A, W, C, M, N, K, ldc, slices, mf);
        }
    }
    fmi_set_error("sealnn_hgemm_nt: no kernel for %u x %u tiles with %u stages and %u K groups", BM, BN, stages, kgroups);
    return FMI_ERR_ARG;
}

}   // namespace

// config: 0 = pick by shape; else tile (1 = 128 x 128, 2 = 64 x 64, 3 = 128 x 64, 4 = 64 x 128; + 128: row tiles fastest, XCD-grouped) | stages << 8 (LDS stages 1..3; 0: two) |
// kgroups << 12 (K groups of 4 waves per workgroup: 1, 2 (tiles 2..4), 4 (tile 2); 0: one) | slices << 16 (split-K over workgroups: slab s of C
// at C + s * M * ldc, the caller sums the slabs).  Probes and tests pass it explicitly.
extern "C" int sealnn_hgemm_nt(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config)
{
    if (!a || !w || !c || M == 0 || N == 0) { fmi_set_error("sealnn_hgemm_nt: null / empty operand"); return FMI_ERR_ARG; }
    if (K == 0 || K % BK) { fmi_set_error("sealnn_hgemm_nt: K = %u must be a multiple of %u", K, BK); return FMI_ERR_UNSUPPORTED; }
    if (((uintptr_t)a | (uintptr_t)w) & 15) { fmi_set_error("sealnn_hgemm_nt: operands must be 16-byte aligned"); return FMI_ERR_ARG; }
    const uint32_t mf = (config >> 7) & 1u;          // bit 7 of the tile byte: row tile fastest + XCD remap (wide-N products)
    uint32_t tile = config & 0x7f, stages = (config >> 8) & 0xf, kgroups = (config >> 12) & 0xf, slices = (config >> 16) ? (config >> 16) : 1;
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
    switch (tile) {
    case 1: return launch<128, 128>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    case 2: return launch<64, 64>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    case 3: return launch<128, 64>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    case 4: return launch<64, 128>(st, a, w, c, M, N, K, ldc, slices, stages, kgroups, mf);
    default: fmi_set_error("sealnn_hgemm_nt: unknown tile %u", tile); return FMI_ERR_ARG;
    }
}
