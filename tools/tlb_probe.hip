// Experiment (not part of the product): does the way a large buffer is ALLOCATED change the cost of
// random 128-byte requests over it?  The wavelet matrix (11.5 GB at NQ size) is probed at random;
// profiles/r1_gather_calib.txt shows throughput falling 3.6x once the working set passes ~3 GiB, which
// looks like address-translation reach.  Modes: hipMalloc, hipExtMallocWithFlags(hipDeviceMallocContiguous),
// hipMemCreate + hipMemMap at a 1 GiB-aligned reservation.  For each: random-gather throughput and the
// latency of one dependent chain (one lane), at several working-set sizes.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/tlb_probe tools/tlb_probe.hip ; run: /tmp/tlb_probe [GiB=12]
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

__global__ __launch_bounds__(256) void k_gather(const V4 *buf, uint64_t n_lines, uint64_t probes, uint64_t seed, uint32_t *sink)
{
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state = mix(seed + tid * 0x9E3779B97F4A7C15ULL);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < probes; i++) {
        const V4 *p = buf + (state % n_lines) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) { V4 v = p[j]; acc += v.a ^ v.b ^ v.c ^ v.d; }
        state = mix(state + i);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// one lane, `steps` dependent random loads: latency per step
__global__ void k_chain(const V4 *buf, uint64_t n_lines, uint64_t steps, uint64_t seed, uint32_t *sink)
{
    uint64_t state = mix(seed);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < steps; i++) {
        const V4 v = buf[(state % n_lines) * 8];
        acc += v.a;
        state = mix(state + v.a + i);
    }
    sink[1] = acc;
}

static void measure(const char *mode, V4 *buf, uint64_t bytes, uint32_t *sink)
{
    for (uint64_t ws : {(uint64_t)1 << 30, (uint64_t)3 << 30, (uint64_t)6 << 30, bytes}) {
        if (ws > bytes) continue;
        const uint64_t n_lines = ws / 128;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, buf, n_lines, 32ull, 1ull, sink);
        hipEventRecord(a);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, buf, n_lines, 32ull, 7ull + r, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double gbs = 4096.0 * 256 * 32 * 128 * 3 / (ms * 1e-3) / 1e9;
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(1), 0, 0, buf, n_lines, 200ull, 3ull, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(1), 0, 0, buf, n_lines, 4000ull, 5ull, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("%-22s working set %5.1f GiB : %8.1f GB/s random 128-B lines, %7.1f ns per dependent load\n", mode, ws / 1073741824.0, gbs,
               ms * 1e6 / 4000);
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    const uint64_t gib = argc > 1 ? strtoull(argv[1], 0, 10) : 12;
    const uint64_t bytes = gib << 30;
    uint32_t *sink; hipMalloc(&sink, 64);
    {
        V4 *buf = nullptr;
        if (hipMalloc(&buf, bytes) == hipSuccess) { hipMemset(buf, 1, bytes); measure("hipMalloc", buf, bytes, sink); hipFree(buf); }
        else printf("hipMalloc failed\n");
    }
    {
        V4 *buf = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocContiguous);
        if (e == hipSuccess) { hipMemset(buf, 1, bytes); measure("contiguous", buf, bytes, sink); hipFree(buf); }
        else { printf("hipExtMallocWithFlags(Contiguous, %llu GiB) failed: %s\n", (unsigned long long)gib, hipGetErrorString(e)); (void)hipGetLastError(); }
    }
    {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gmin = 0, grec = 0;
        hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
        hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
        printf("hipMem granularity: minimum %zu, recommended %zu\n", gmin, grec);
        hipMemGenericAllocationHandle_t hnd;
        void *va = nullptr;
        hipError_t e = hipMemCreate(&hnd, bytes, &prop, 0);
        if (e == hipSuccess) e = hipMemAddressReserve(&va, bytes, 1ull << 30, nullptr, 0);
        if (e == hipSuccess) e = hipMemMap(va, bytes, 0, hnd, 0);
        hipMemAccessDesc acc{};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        if (e == hipSuccess) e = hipMemSetAccess(va, bytes, &acc, 1);
        if (e == hipSuccess) { printf("hipMemCreate+Map at %p\n", va); hipMemset(va, 1, bytes); measure("hipMemCreate/1GiB-va", (V4 *)va, bytes, sink); }
        else printf("hipMemCreate path failed: %s\n", hipGetErrorString(e));
    }
    return 0;
}
