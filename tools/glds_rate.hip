// What one CU can pull from the L2 into its LDS per clock, by the form of the load -- the number the tall tile of sealnn_hgemm_nt is built around.
// Every workgroup (8 waves, one per CU: 144 KB of LDS asked for) streams ROWS x 128-byte lines of a matrix with a 6 KB row pitch (the A / W operand of a
// decode-step product) through an LDS ring, `iters` K steps of 128 bytes, with DEPTH steps in flight:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave instruction)
//   mode 1: global_load_dwordx4 into registers, nothing else (the load path alone)
//   mode 2: global_load_dwordx4 into registers + ds_write_b128 (register staging)
//   mode 3: LDS-DMA of HALF lines: a piece = 16 rows x 64 bytes, a K step = 64 bytes of every row (the two halves of a 128-byte line in two steps)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/glds_rate tools/glds_rate.hip && /tmp/glds_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PIECES = 1 KB pieces per wave and K step; DEPTH = K steps in flight; rows of a step = PIECES * 8 waves * 8
template <int MODE, int PIECES, int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const unsigned char *__restrict__ src, uint64_t pitch, uint32_t rows_total, uint32_t iters, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr uint32_t STAGE = PIECES * 8 * 1024;
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // workgroup b reads rows (b * 37 + ...) % rows_total: different workgroups overlap as the tiles of a product do
    const unsigned char *p[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; i++) {
        const uint32_t row = ((blockIdx.x % 24) * 64 + (i * 8 + wave) * 8 + (lane >> 3)) % rows_total;
        p[i] = src + (uint64_t)row * pitch + (lane & 7u) * 16;
        if constexpr (MODE == 3) {
            const uint32_t row2 = ((blockIdx.x % 24) * 64 + (i * 8 + wave) * 16 + (lane >> 2)) % rows_total;
            p[i] = src + (uint64_t)row2 * pitch + (lane & 3u) * 16;
        }
    }
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](uint32_t kt, uint32_t buf, u32x4 (&regs)[PIECES]) {
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            const unsigned char *g = p[i] + (uint64_t)kt * (MODE == 3 ? 64 : 128);
            if constexpr (MODE == 0 || MODE == 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(lds + buf * STAGE + (i * 8 + wave) * 1024), 16, 0, 0);
            else
                regs[i] = *reinterpret_cast<const u32x4 *>(g);
        }
    };
    if constexpr (MODE == 0 || MODE == 3) {
        u32x4 dummy[PIECES];
        for (uint32_t s = 0; s < DEPTH; s++) issue(s, s, dummy);
        uint32_t buf = 0;
        for (uint32_t kt = 0; kt < iters; kt++) {
            wait_vm<(DEPTH - 1) * PIECES>();
            __builtin_amdgcn_s_barrier();
            issue(kt + DEPTH, buf, dummy);                   // (reads past the step count stay inside the row: pitch >= (iters + DEPTH) * 128)
            buf = buf + 1 == DEPTH ? 0 : buf + 1;
        }
        wait_vm<0>();
        __syncthreads();
        acc = *reinterpret_cast<u32x4 *>(lds + threadIdx.x * 16);
    } else {
        // registers: DEPTH = 1 step in flight per wave (PIECES x 4 VGPRs), as a register-staged GEMM has
        u32x4 r[PIECES];
        issue(0, 0, r);
        for (uint32_t kt = 0; kt < iters; kt++) {
            u32x4 nxt[PIECES];
            issue(kt + 1, 0, nxt);
            wait_vm<PIECES>();
            if constexpr (MODE == 2) {
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < PIECES; i++) *reinterpret_cast<u32x4 *>(lds + (kt & 1) * STAGE + (i * 8 + wave) * 1024 + lane * 16) = r[i];
            } else {
#pragma unroll
                for (int i = 0; i < PIECES; i++) acc ^= r[i];
            }
#pragma unroll
            for (int i = 0; i < PIECES; i++) r[i] = nxt[i];
        }
        wait_vm<0>();
        if constexpr (MODE == 2) { __syncthreads(); acc = *reinterpret_cast<u32x4 *>(lds + threadIdx.x * 16); }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int PIECES, int DEPTH>
static int run(const char *what, const unsigned char *src, uint64_t pitch, uint32_t rows, uint32_t iters, uint32_t *sink, int wgs)
{
    const size_t lds = (MODE == 0 || MODE == 3) ? (size_t)DEPTH * PIECES * 8192 : (size_t)2 * PIECES * 8192;
    CHECK(hipFuncSetAttribute((const void *)k_stream<MODE, PIECES, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_stream<MODE, PIECES, DEPTH>), dim3(wgs), dim3(512), 144 * 1024, 0, src, pitch, rows, iters, sink);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL((k_stream<MODE, PIECES, DEPTH>), dim3(wgs), dim3(512), 144 * 1024, 0, src, pitch, rows, iters, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, bytes = (double)wgs * iters * PIECES * 8192;
    printf("%-28s pieces/wave/step %d depth %d (%3zu KB in flight): %7.1f us  %6.2f TB/s  %5.1f GB/s per CU  = %4.1f B/clk/CU at 2.4 GHz\n", what, PIECES, DEPTH,
           (MODE == 0 || MODE == 3) ? lds / 1024 : (size_t)PIECES * 8, us, bytes / us / 1e6, bytes / us / 1e3 / wgs, bytes / us / 1e3 / wgs / 2.4);
    return 0;
}

int main()
{
    const uint64_t pitch = 6144 * 2;            // a [rows, 6144] fp16 matrix: rows 12 KB apart (the K steps read stay inside a row)
    const uint32_t rows = 1536, iters = 64;     // 18.9 MB: resident in the L2s / the memory-side cache after the first pass
    unsigned char *src; uint32_t *sink;
    CHECK(hipMalloc(&src, (size_t)rows * pitch)); CHECK(hipMemset(src, 1, (size_t)rows * pitch));
    CHECK(hipMalloc(&sink, 4));
    for (int wgs : {256, 64}) {
        printf("# %d workgroups of 8 waves, %u K steps of 128 B per row\n", wgs, iters);
        if (run<0, 6, 2>("LDS-DMA", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<0, 6, 3>("LDS-DMA", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<0, 3, 4>("LDS-DMA", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<0, 3, 6>("LDS-DMA", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<0, 2, 8>("LDS-DMA", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<0, 1, 16>("LDS-DMA", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<3, 3, 4>("LDS-DMA, half lines", src, pitch, rows, 2 * iters, sink, wgs)) return 1;
        if (run<3, 3, 6>("LDS-DMA, half lines", src, pitch, rows, 2 * iters, sink, wgs)) return 1;
        if (run<1, 6, 1>("registers, no LDS write", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<1, 12, 1>("registers, no LDS write", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<2, 6, 1>("registers + ds_write_b128", src, pitch, rows, iters, sink, wgs)) return 1;
        if (run<2, 12, 1>("registers + ds_write_b128", src, pitch, rows, iters, sink, wgs)) return 1;
    }
    return 0;
}
