// Calibration micro-benchmark (not part of the product): random aligned gathers
// of G bytes per lane (G = 64 or 128) over a large HBM buffer.
//   - practical ceiling of the access pattern the wavelet-matrix probes have;
//   - calibration of rocprofv3's FETCH_SIZE for this pattern (run under
//     `rocprofv3 --kernel-trace --pmc FETCH_SIZE`): known useful bytes per launch
//     = probes * G.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_calib tools/gather_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

struct alignas(16) V4 { uint32_t a, b, c, d; };

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int G, int DEP>
__global__ __launch_bounds__(256) void k_gather(const V4 *buf, uint64_t n_granules, uint64_t probes_per_thread, uint64_t seed, uint32_t *sink)
{
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state = mix(seed + tid * 0x9E3779B97F4A7C15ULL);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < probes_per_thread; i++) {
        uint64_t g = state & (n_granules - 1);   // n_granules is a power of two
        const V4 *p = buf + g * (G / 16);
        uint32_t local = 0;
#pragma unroll
        for (int j = 0; j < G / 16; j++) { V4 v = p[j]; local += v.a ^ v.b ^ v.c ^ v.d; }
        acc += local;
        // DEP=1: next address depends on the loaded data (a dependent chain, like a tree walk)
        state = mix(state + (DEP ? local : 0) + i);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int G, int DEP>
static void run(const V4 *buf, uint64_t bytes, int blocks, uint64_t ppt, uint32_t *sink)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    uint64_t n_gran = bytes / G;
    hipLaunchKernelGGL((k_gather<G, DEP>), dim3(blocks), dim3(256), 0, 0, buf, n_gran, ppt, 1ull, sink);
    hipEventRecord(a);
    const int reps = 3;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_gather<G, DEP>), dim3(blocks), dim3(256), 0, 0, buf, n_gran, ppt, 7ull + r, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    double useful = (double)blocks * 256 * ppt * G * reps;
    printf("G=%3d dep=%d blocks=%6d probes/thread=%4llu : %8.1f GB/s useful  (%.3f ms/launch, %.1f MB/launch)\n", G, DEP, blocks,
           (unsigned long long)ppt, useful / (ms * 1e-3) / 1e9, ms / reps, useful / reps / 1e6);
}

int main(int argc, char **argv)
{
    uint64_t max_gib = (argc > 1 ? strtoull(argv[1], 0, 10) : 32ull);
    V4 *buf; uint32_t *sink;
    if (hipMalloc(&buf, max_gib << 30) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, max_gib << 30);
    if (argc > 2) {      // fine sweep of the TLB-reach knee: working sets in units of 256 MiB, 128-byte granules only
        for (uint64_t q : {1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128}) {
            if ((q << 28) > (max_gib << 30)) break;
            printf("-- working set %.2f GiB\n", q / 4.0);
            run<128, 0>(buf, q << 28, 4096, 64, sink);
            run<64, 0>(buf, q << 28, 4096, 64, sink);
        }
        return 0;
    }
    for (uint64_t gib = 1; gib <= max_gib; gib *= 2) {     // working-set sweep (TLB reach)
        if (gib != 1 && gib != 8 && gib != max_gib) continue;
        printf("-- working set %llu GiB\n", (unsigned long long)gib);
        for (int blocks : {4096, 16384}) {
            run<64, 0>(buf, gib << 30, blocks, 64, sink);
            run<64, 1>(buf, gib << 30, blocks, 64, sink);
            run<128, 0>(buf, gib << 30, blocks, 64, sink);
        }
    }
    return 0;
}
