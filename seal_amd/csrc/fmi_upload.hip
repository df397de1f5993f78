// fmi_dev_kernel_copy: a host <-> device copy done by a KERNEL that reads / writes the (pinned, device-visible) host buffer.
//
// Why not hipMemcpyAsync: a copy of some size goes to a DMA engine, and the engine's queue is shared by every stream of the process.  The
// searcher keeps a decode enqueued ahead whose last command is the copy of its hypotheses to the host; the plan of an aggregation
// (3 MB) sent with hipMemcpyAsync on the index's own stream queued BEHIND that copy and reached the GPU when the decode had ended,
// 25-30 ms after it was enqueued -- with it the whole aggregation, which then ran beside the next rescoring forward instead of beside
// the decode (SEAL_OVERLAP_TIMING=2: "plan of 3110752 bytes on the GPU" 0.07 ms after "decode ends", every batch); and with the plan
// sent by a kernel the copy of the RESULTS back to the host waited the same way (profiles/r5_overlap_timeline.txt).
// A kernel is ordered by its own stream only.  3 MB over the host link is ~60 us.
#include <hip/hip_runtime.h>

#include "../../include/sealfm.h"

#include <algorithm>

#include "fmi_internal.h"

namespace {
template <typename W>
__global__ __launch_bounds__(256) void k_copy_words(W *dst, const W *src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

bool kernel_visible(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}
}  // namespace

extern "C" int fmi_dev_kernel_copy(void *stream, void *dst, const void *src, uint64_t bytes)
{
    if (!dst || !src) { fmi_set_error("fmi_dev_kernel_copy: null argument"); return FMI_ERR_ARG; }
    if (((uintptr_t)dst | (uintptr_t)src | bytes) & 3) { fmi_set_error("fmi_dev_kernel_copy: pointers and size must be multiples of 4"); return FMI_ERR_ARG; }
    if (!bytes) return FMI_OK;
    if (!kernel_visible(dst) || !kernel_visible(src)) {
        fmi_set_error("fmi_dev_kernel_copy: both buffers must be device memory or PINNED host memory (a kernel cannot touch pageable memory)");
        return FMI_ERR_ARG;
    }
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0) {
        const uint64_t n = bytes / 16;
        hipLaunchKernelGGL(k_copy_words<uint4>, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                           (uint4 *)dst, (const uint4 *)src, n);
    } else {
        const uint64_t n = bytes / 4;
        hipLaunchKernelGGL(k_copy_words<uint32_t>, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                           (uint32_t *)dst, (const uint32_t *)src, n);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmi_set_error("fmi_dev_kernel_copy launch failed: %s", hipGetErrorString(e)); return FMI_ERR_HIP; }
    return FMI_OK;
}
