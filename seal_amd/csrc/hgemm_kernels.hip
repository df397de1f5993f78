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
//   * two LDS stages: the loads of stage t + 1 are issued before stage t is computed and waited for with a COUNTED s_waitcnt vmcnt -- raw
//     s_barrier, never __syncthreads() while an LDS-DMA is in flight (it would drain the prefetch); ALL of the LDS is one array;
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

template <uint32_t BM, uint32_t BN, bool PIPE>
__global__ __launch_bounds__(256) void k_hgemm_nt(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, float *__restrict__ C,
                                                  uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t k_per_slice, uint64_t slab_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr uint32_t TM = BM / 2, TN = BN / 2, FM = TM / 16, FN = TN / 16;
    constexpr uint32_t STAGE = (BM + BN) * ROW_BYTES;
    constexpr uint32_t NL = (BM + BN) / 32;                    // LDS-DMA instructions per wave and stage
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wm = wave >> 1, wn = wave & 1;
    const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const uint32_t kbeg = blockIdx.z * k_per_slice;
    const uint32_t nk = k_per_slice / BK;
    C += (uint64_t)blockIdx.z * slab_stride;

    float4v acc[FM][FN];
#pragma unroll
    for (uint32_t i = 0; i < FM; i++)
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](uint32_t kt, uint32_t buf) {
        unsigned char *st = lds + buf * STAGE;
        load_tile<BM>(A, m0, M, K, kbeg + kt * BK, st, wave, lane);
        load_tile<BN>(W, n0, N, K, kbeg + kt * BK, st + BM * ROW_BYTES, wave, lane);
    };
    auto compute = [&](uint32_t buf) {
        const unsigned char *ta = lds + buf * STAGE, *tb = ta + BM * ROW_BYTES;
#pragma unroll
        for (uint32_t ks = 0; ks < 2; ks++) {
            half8 fa[FM], fb[FN];
#pragma unroll
            for (uint32_t i = 0; i < FM; i++) fa[i] = read_frag(ta, wm * TM + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
            for (uint32_t j = 0; j < FN; j++) fb[j] = read_frag(tb, wn * TN + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
            for (uint32_t i = 0; i < FM; i++)
#pragma unroll
                for (uint32_t j = 0; j < FN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (PIPE) {
        issue(0, 0);
        for (uint32_t kt = 0; kt < nk; kt++) {
            if (kt + 1 < nk) {
                issue(kt + 1, (kt + 1) & 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");      // all but the newest NL: stage kt has landed (this wave's share)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                        // ... and everybody else's
            asm volatile("" ::: "memory");
            compute(kt & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // stage kt is read: iteration kt + 1 may refill its buffer
            asm volatile("" ::: "memory");
        }
    } else {
        for (uint32_t kt = 0; kt < nk; kt++) {
            issue(kt, 0);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    // D fragment: column lane & 15, rows 4 (lane >> 4) .. + 3
#pragma unroll
    for (uint32_t i = 0; i < FM; i++) {
#pragma unroll
        for (uint32_t j = 0; j < FN; j++) {
            const uint32_t col = n0 + wn * TN + j * 16 + (lane & 15);
            const uint32_t rbase = m0 + wm * TM + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t row = rbase + r;
                if (row < M && col < N) C[(uint64_t)row * ldc + col] = acc[i][j][r];
            }
        }
    }
}

template <uint32_t BM, uint32_t BN>
int launch(hipStream_t st, const void *A, const void *W, float *C, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t slices, bool pipe)
{
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, slices);
    const uint32_t kps = K / slices;
    const size_t lds = (size_t)(pipe ? 2 : 1) * (BM + BN) * ROW_BYTES;
    const uint64_t slab = (uint64_t)M * ldc;
    if (pipe) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_hgemm_nt<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_hgemm_nt<BM, BN, true>), grid, dim3(256), lds, st, (const _Float16 *)A, (const _Float16 *)W, C, M, N, K, ldc, kps, slab);
    } else {
        hipLaunchKernelGGL((k_hgemm_nt<BM, BN, false>), grid, dim3(256), lds, st, (const _Float16 *)A, (const _Float16 *)W, C, M, N, K, ldc, kps, slab);
    }
    return hipGetLastError() == hipSuccess ? FMI_OK : FMI_ERR_HIP;
}

}   // namespace

// config: 0 = pick by shape; else (tile: 1 = 128 x 128, 2 = 64 x 64, 3 = 128 x 64, 4 = 64 x 128) | 0x100: not pipelined (one LDS stage,
// __syncthreads) | slices << 16 (split-K: slab s of C at C + s * M * ldc; the caller sums the slabs).  Probes and tests pass it explicitly.
extern "C" int sealnn_hgemm_nt(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config)
{
    if (!a || !w || !c || M == 0 || N == 0) { fmi_set_error("sealnn_hgemm_nt: null / empty operand"); return FMI_ERR_ARG; }
    if (K == 0 || K % BK) { fmi_set_error("sealnn_hgemm_nt: K = %u must be a multiple of %u", K, BK); return FMI_ERR_UNSUPPORTED; }
    if (((uintptr_t)a | (uintptr_t)w) & 15) { fmi_set_error("sealnn_hgemm_nt: operands must be 16-byte aligned"); return FMI_ERR_ARG; }
    uint32_t tile = config & 0xff, slices = (config >> 16) ? (config >> 16) : 1;
    const bool pipe = !(config & 0x100);
    if ((K / BK) % slices) { fmi_set_error("sealnn_hgemm_nt: %u K steps do not split into %u slices", K / BK, slices); return FMI_ERR_ARG; }
    if (tile == 0) {
        // enough workgroups for 256 CUs first, large tiles (less operand traffic per flop) second
        const uint64_t big = (uint64_t)((M + 127) / 128) * ((N + 127) / 128);
        tile = big >= 120 ? 1 : ((uint64_t)((M + 127) / 128) * ((N + 63) / 64) >= 120 ? 3 : 2);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
    case 1: return launch<128, 128>(st, a, w, c, M, N, K, ldc, slices, pipe);
    case 2: return launch<64, 64>(st, a, w, c, M, N, K, ldc, slices, pipe);
    case 3: return launch<128, 64>(st, a, w, c, M, N, K, ldc, slices, pipe);
    case 4: return launch<64, 128>(st, a, w, c, M, N, K, ldc, slices, pipe);
    default: fmi_set_error("sealnn_hgemm_nt: unknown tile %u", tile); return FMI_ERR_ARG;
    }
}
